// rv_align.hip -- placeholder until the level-synchronous recursion lands
#include "rv_index.h"
struct Align { int dummy; };
void rv_align_free(rv_index *h) { delete h->al; h->al = nullptr; }
#define NI(name) rv_set_error(name ": not implemented yet"); return -1
extern "C" {
int64_t rv_getmultimums(rv_index *, int, int, int, int64_t *) { NI("rv_getmultimums"); }
int rv_fetch_multi(rv_index *, uint32_t *, int32_t *, int64_t *, uint16_t *, int64_t *) { NI("rv_fetch_multi"); }
int rv_align_begin(rv_index *, int, int) { NI("rv_align_begin"); }
int rv_frontier_size(rv_index *) { NI("rv_frontier_size"); }
int rv_frontier_scan(rv_index *) { NI("rv_frontier_scan"); }
int rv_sub_info(rv_index *, int, rv_sub *) { NI("rv_sub_info"); }
int rv_sub_nodes(rv_index *, int, int64_t *) { NI("rv_sub_nodes"); }
int rv_sub_mums(rv_index *, int, uint32_t *, int32_t *, int64_t *, uint16_t *, int64_t *) { NI("rv_sub_mums"); }
int64_t rv_sub_array(rv_index *, int, int, void *, int64_t) { NI("rv_sub_array"); }
int rv_sub_split(rv_index *, int, uint32_t, int, const int64_t *, const int64_t *, int, const int64_t *, int, const int64_t *, int, const int64_t *, int) { NI("rv_sub_split"); }
int rv_frontier_commit(rv_index *, int32_t *) { NI("rv_frontier_commit"); }
int rv_align_end(rv_index *) { NI("rv_align_end"); }
int rv_align_builtin(rv_index *, int, int, rv_align_stats *) { NI("rv_align_builtin"); }
int64_t rv_anchor_count(rv_index *, int64_t *) { NI("rv_anchor_count"); }
int rv_fetch_anchors(rv_index *, uint32_t *, int64_t *, int64_t *) { NI("rv_fetch_anchors"); }
int rv_set_trace(rv_index *, int) { NI("rv_set_trace"); }
int64_t rv_trace_count(rv_index *) { NI("rv_trace_count"); }
int rv_fetch_trace(rv_index *, rv_trace *, int64_t) { NI("rv_fetch_trace"); }
}

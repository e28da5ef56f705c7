// rv_cascade.h -- host interface of the anchor cascade (rv_cascade.hip)
#pragma once
#include "rv_common.h"
#include <vector>

// device scratch of the cascade, kept between runs (grow-only like every other buffer of a handle)
struct RvCascadeBufs {
    DBuf d[40];
    HBuf hstage, hstage2;              // pinned staging of the multi-sample cascade's results and tables (rv_cascade_multi.hip)
    u32 M = 0, NW = 0;                 // the lists of the last run on this handle (a second attempt starts from them)
    // several sequences per sample (rv_cascade.hip "the lineage of rest sub-indices"): what the first attempt found out on the host,
    // kept for a second attempt on the same lists
    u32 lin_roots = 0, lin_picks = 0, lin_nA = 0, lin_nB = 0;
    int64_t lin_steps = 0; int lin_maxdepth = 0; size_t lin_off[6] = {0, 0, 0, 0, 0, 0};
    // a chain member the match list does not decide: its sequences as (begin, end) pairs and its depth -- the caller makes it the level pipeline's frontier
    std::vector<int64_t> lin_rest; int lin_rest_depth = 0;
    // the cascade's level loop on a stream of the highest priority (RV_CASCADE_PRIO): its kernels are a chain of tiny launches, and on the handle's own
    // stream they queue behind the large kernels of whatever other handles share the hardware queue
    hipStream_t prio_stream = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;
    void release() {
        for (auto &b : d) b.release(); hstage.release(); hstage2.release();
        if (prio_stream) { (void)hipStreamDestroy(prio_stream); prio_stream = nullptr; }
        if (ev_in) { (void)hipEventDestroy(ev_in); ev_in = nullptr; }
        if (ev_out) { (void)hipEventDestroy(ev_out); ev_out = nullptr; }
    }
};

struct RvCascadeIO {
    // where the built-in run collects its anchors and counters (the leaf kernel's output area, rv_leaf.h)
    u32 *anchor_count; u32 anchor_cap; u32 *anchor_l; int64_t *anchor_pos;
    unsigned long long *stats;        // [0] sub-indices visited, [1] anchors, [2] anchored bp, [3] max depth
    u32 *leaf_err;
    u32 stage_cap;
    // level arrays + root table for the sub-indices that are rebuilt from their text
    DBuf *lvSA, *lvLCP, *lvBWT, *roots;
};

struct RvCascadeOut {
    bool done;                         // false: nothing was decided, the caller runs the level pipeline from the top
    int levels;
    int64_t cands, witnesses, children, undecided, rebuilt_ranks;
    int64_t solved, unsolved;          // second attempt: large undecided sub-indices decided from their witnesses / left undecided
    const char *why;                   // done == false: the reason
};

// more than two samples (rv_cascade_multi.hip): the decided part's anchors come back on the host (the level pipeline keeps its anchors
// there), the undecided sub-indices as a frontier for it -- metadata in the layout of rv_frontier_import, arrays in device memory
#define RV_CASM_K 16
struct RvCascadeMultiOut {
    bool done; int levels; int64_t cands, witnesses, children, undecided, rebuilt_ranks, steps; int maxdepth; const char *why;
    int64_t big, big_ranks;                                                  // of the undecided: those rebuilt through global memory (k_casmb_*)
    std::vector<u32> an_l; std::vector<int64_t> an_pos;                      // k members per anchor, ascending
    std::vector<int64_t> meta, node_first, nodes;                           // undecided sub-indices: 6 numbers each; intervals as (begin, end) pairs
    const void *d_sa, *d_lcp, *d_bwt;
};

struct rv_index;
// rv_batch_run (rv_api.hip): the jobs of a batch run the level loops of their anchor cascades as one (rv_cascade_multi.hip)
struct RvBatchGroup;
RvBatchGroup *rv_batch_group_new(int device);
void rv_batch_group_begin(RvBatchGroup *g, int total);      // before the jobs of a run start: how many there are
void rv_batch_group_leave(RvBatchGroup *g);                 // a job that will not come to the rendezvous
void rv_batch_group_free(RvBatchGroup *g);
void rv_batch_group_info(const RvBatchGroup *g, int64_t *out);      // out[0] joint level loops run, out[1] jobs they served
int rv_cascade_multi_run(rv_index *h, RvCascadeBufs &cb, int minl, RvCascadeMultiOut *out);
// danger: large undecided sub-indices are decided from their witnesses (the second attempt, rv_cascade.hip); reuse: the match and
// witness lists of the previous run on this handle are still in cb (same index, same minl)
int rv_cascade_run(rv_index *h, RvCascadeBufs &cb, const RvCascadeIO &io, int minl, RvCascadeOut *out, int danger = 0, int reuse = 0);

// rv_leaf.h -- host interface of the leaf kernel (rv_leaf.hip)
#pragma once
#include "rv_common.h"
#include "../../include/reveal_amd.h"

#ifndef RV_LEAF_N
#define RV_LEAF_N 2048
#endif

// a sub-index of a two-sample alignment with at most RV_LEAF_N ranks and at most one interval per sample
struct RvLeafRoot {
    int64_t off;            // first rank in the current level arrays
    int32_t n, depth;
    int64_t a0, a1, b0, b1; // interval of sample 0 / sample 1 (empty: a0 >= a1)
};

struct RvLeafArgs {
    const RvLeafRoot *roots;
    const sa_t *SA; const lcp_t *LCP; const uint8_t *BWT;   // current level arrays (read only)
    int64_t nsep0;
    int minl;
    u32 lcap;                                                // bound on every LCP value (max LCP of the main index)
    // outputs
    u32 stage_cap;                                           // anchors a workgroup stages in LDS before it writes them out (<= 256; RV_LEAF_ACAP: test hook)
    u32 *anchor_count; u32 anchor_cap; u32 *anchor_l; int64_t *anchor_pos;      // anchor k: length anchor_l[k], members anchor_pos[2k], [2k+1]
    unsigned long long *stats;                               // [0] sub-indices visited, [1] anchors, [2] anchored bp, [3] max depth
    int trace; u32 *trace_count; u32 trace_cap; rv_trace *trace_out;
    u32 *err;
};

int rv_leaf_launch(Workspace &ws, const RvLeafArgs &a, int nroots);
// lower-cases the matched text of the anchors the leaf launches wrote (reveal.c:1230-1234), once, when the run ends
int rv_leaf_lower_launch(Workspace &ws, uint8_t *T, const int64_t *pos, const u32 *len, u32 na);

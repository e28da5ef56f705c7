// rv_decide.h -- device-side decisions of the untraced two-sample recursion (see rv_decide.hip)
#pragma once
#include "rv_common.h"
#include "rv_scan.h"
#include "rv_split.h"

struct RvDecideArgs {
    int nsubs;
    u32 lcap;                       // maximum LCP of the index (longest window in front of a cut)
    const sa_t *nodes;              // [4*nsubs] a0,a1,b0,b1 per sub-index (an absent interval is empty)
    const uint8_t *flags;           // [nsubs] bit 0: finished elsewhere (leaf kernel) -- takes no decision
    const RvPairRec *picks;         // header + one record per sub-index (rank 0xFFFFFFFF = none)
    // label tables (RvLabelTabs), fixed strides: four class intervals and two matched ranges per sub-index
    int *ctab_first, *mtab_first;   // [nsubs+1]
    sa_t *cb, *ce; uint8_t *cc;     // [4*nsubs]
    sa_t *mb, *me;                  // [2*nsubs]
    // split tables (RvSplitArgs)
    u32 *child_n, *child_base, *sub_off;    // [3*nsubs]
    u32 *expect_total;              // [4]
    int *cut_first, *mend_first;    // [nsubs+1]
    sa_t *cut_lo, *cut_hi, *mend_pos;       // [2*nsubs]
    u32 *err;
    u32 ovf_cap;                    // picks[0] carries the scan's overflow count: beyond this the picks are incomplete -> no decisions
    RvBubbleDesc *kid;              // [nsubs] leading child of every sub-index as a bubble descriptor (n = 0: nothing to do)
};

#define RV_DECIDE_MAX_SUBS 65536    // above this the single-block offset scan would take longer than the host round trip it hides
int rv_decide_launch(Workspace &ws, const RvDecideArgs &d);

// ---- more than two samples (untraced built-in run) --------------------------------------------------
// A level whose sub-indices own at most one interval per sample: the built-in picker's match (one member in each of the
// sub-index' samples) and the linear interval model give, per sample slot q of sub-index s,
//     member p in node [b,e):   leading [b,p)   matched [p,p+l)   trailing [p+l,e)          no member: rest [b,e)
// plus rv_frontier_commit's rule that a child which cannot hold another match of minl bases in each of its samples is
// never made (its ranks take the label of matched suffixes).  Fixed strides per sub-index: 2 W class intervals (slot 2q =
// leading or rest, 2q+1 = trailing; empty slots are empty intervals placed in text order, so the tables stay sorted by
// begin for the label look-up), W matched ranges, W cut windows, W "behind the match" positions.
struct RvDecideMultiArgs {
    int nsubs, W;
    int minl, minn;
    u32 lcap;
    const sa_t *nodes;              // [2*W*nsubs] (begin, end) of sample q's interval in sub-index s, (0,0) = none
    const int *want;                // [nsubs] samples of the sub-index = members of a pick
    const u32 *pick_l;              // [nsubs] length of the pick, 0 = none
    const sa_t *pick_pos;           // [W*nsubs] its members (the first want[s])
    const u32 *cand_count; u32 cand_cap;      // the picker's candidate list overflowed (*cand_count > cand_cap): picks incomplete -> no decisions
    int *ctab_first, *mtab_first, *cut_first, *mend_first;      // [nsubs+1]
    sa_t *cb, *ce; uint8_t *cc;     // [2*W*nsubs]
    sa_t *mb, *me;                  // [W*nsubs]
    sa_t *cut_lo, *cut_hi, *mend_pos;       // [W*nsubs]
    u32 *child_n, *child_base, *sub_off;    // [3*nsubs]
    u32 *expect_total;              // [4]
    u32 *err;
    RvBubbleDesc *kid;              // [nsubs] leading child of every sub-index as a bubble descriptor (n = 0: nothing to do)
};
int rv_decide_multi_launch(Workspace &ws, const RvDecideMultiArgs &d);

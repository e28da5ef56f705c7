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

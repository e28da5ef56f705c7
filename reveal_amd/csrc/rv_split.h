// rv_split.h -- host interface of the per-level label / split / bubble kernels
#pragma once
#include "rv_common.h"

#define RV_SPLIT_TILE 2048

// per-sub-index interval tables of one level (device pointers): sub s owns
// entries [ctab_first[s], ctab_first[s+1]) resp. [mtab_first[s], mtab_first[s+1]),
// sorted by begin and non-overlapping inside a sub-index
struct RvLabelTabs {
    const int64_t *sub_start;       // [nsubs+1], last = m
    int            nsubs;
    const int     *tile_sub;        // [ntiles] sub-index that owns rank tile * RV_SPLIT_TILE
    const int     *ctab_first;      // lead / trail / rest intervals
    const sa_t    *cbegin, *cend;
    const uint8_t *ccls;            // 1 lead, 2 trail, 4 rest
    const int     *mtab_first;      // matched ranges [sp, sp+l)
    const sa_t    *mbegin, *mend;
    int            nmatch;          // total number of matched ranges
};

struct RvSplitArgs {
    int64_t ntiles;
    // per tile, class-major [3][ntiles]
    u32 *tile_cnt, *tile_has, *tile_post;   // written by the reduce pass
    u32 *tile_G, *tile_carry;               // exclusive prefixes (k_tile_carry)
    u32 *total;                             // [3] class totals
    // per sub-index of the current frontier
    const int64_t *sub_start;               // [nsubs+1], last = m
    int            nsubs;
    const int     *tile_sub;                // [ntiles] sub-index that owns rank tile * RV_SPLIT_TILE
    const u32     *child_base;              // [nsubs*3] first slot of lead/trail/par child in the next level
    const u32     *child_n;                 // [nsubs*3] expected sizes
    const u32     *sub_off;                 // [nsubs*3] child_base - (class count before the sub); the host knows the child sizes, so it knows this too
    const u32     *expect_total;            // [3] class totals the child sizes add up to (checked against the labels)
    // windows [cut_lo, cut_hi) in front of the cuts of each sub's leading child
    const int     *cut_first;               // [nsubs+1]
    const sa_t    *cut_lo, *cut_hi;
    // positions right behind each sub's matched ranges (sp + l): the BWT byte of a trailing suffix starting there turns lower case
    const int     *mend_first;              // [nsubs+1]
    const sa_t    *mend_pos;
    int            mend_all = 0;            // host-supplied interval lists: a suffix of ANY child may start right behind a matched range
    // outputs
    sa_t  *SA_out;
    lcp_t *LCP_out;
    uint8_t *BWT_out;
    sa_t  *SAi;
    u32   *err;
    // per RV_SPLIT_TILE ranks of the NEXT level (global tiles of the output arrays, preset to 0xFFFFFFFF): a lower bound of the
    // LCP values the LEADING children get there -- the search accelerator of the data-parallel bubble rounds, from the scatter
    u32   *tmin_out = nullptr;
};

struct RvBubbleDesc {
    int64_t off, n;      // leading child's slice of the next-level arrays
    int64_t B;           // cut = begin of a matched interval
    int64_t wlo;         // window [wlo, B)
    int     cut0, cut1;  // this child's windows in cut_lo/cut_hi (for SAi upkeep)
};

// per (child, cut) of the rounds: where the sequential fallback kernel has to start in the sorted active list
// (the data-parallel round sets it past the end when it has done the cut)
struct RvBubbleState {
    int32_t next;
};

// tables of the data-parallel bubble rounds (rv_bubble.hip); per-mover arrays are indexed like `list` (descriptor d at woff[d])
#define RV_PB_CAP 4096      // candidates per (child, cut) the parallel path takes; more -> sequential kernels
struct RvParBubble {
    const int64_t *toff;     // prefix sums of RV_SPLIT_TILE-rank tiles over all descriptors (+1)
    u32 *tmin;               // per GLOBAL tile of the level arrays (rank >> 11): a lower bound of l' over the ranks in it.  Written by
                             // the split (RvSplitArgs::tmin_out), lowered by whatever lowers or moves a value afterwards; never raised,
                             // so a tile may promise a stopper it does not hold (the search then goes on), but never hides one
    u32 *mcnt;               // per descriptor: number of movers (zeroed per level)
    u32 *mrank, *msite;      // movers in discovery order: rank in the child, landing site
    u32 *gcount;             // movers of the running round over all descriptors (reset by the round's last kernel)
    u64 *glist;              // ... (descriptor << 32) | index in the descriptor's list
    u32 *R;                  // mover ranks, ascending
    u32 *Qsite, *QF, *Qt, *Qlcp;   // movers by (site, t): site, final rank, t, LCP value at the final rank
    sa_t *Qs;                // ... suffix
    uint8_t *Qbw, *Qlast;    // ... BWT byte, 1 = last of its site's group
    u32 *tready;             // per tile of the round (toff numbering): the launch number in which the tile was read (k_pb_shift); nullptr =
    u32 epoch;               // the two-pass form (copy-out + scatter).  epoch: this launch's number, never 0, never repeated for a buffer
};

struct RvBubbleArgs {
    RvParBubble par;
    const RvBubbleDesc *desc;
    const int64_t      *woff;     // prefix sums of window widths over all descriptors (+1)
    u32                *cnt;      // per descriptor: number of active ranks found
    u32                *list;     // active ranks, descriptor d at [woff[d], ...)
    uint8_t            *flag;     // one byte per rank of the next level, zero between rounds
    RvBubbleState      *state;    // per descriptor, zeroed per level
    sa_t  *scrSA;                 // scratch of the data-parallel rounds: the (dead) parent-level arrays, indexed like the next level
    lcp_t *scrLCP;
    uint8_t *scrBWT;
    sa_t  *SA;
    lcp_t *LCP;
    uint8_t *BWT;
    sa_t  *SAi;
    const sa_t *cut_lo, *cut_hi;
    u32   *err;
    unsigned long long *dbg;      // RV_LEVEL_LOG: [0] whole-workgroup visits [1] chunks shifted [2] concurrent visits [3] cuts with actives [4] actives
};

// D-label + stable 3-way partition of every split sub-index (D: one scratch byte per rank)
int rv_split_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, uint8_t *D, const uint8_t *BWT, int64_t m, const RvLabelTabs &t, const RvSplitArgs &a,
                    int nsplit);
int rv_lower_launch(Workspace &ws, uint8_t *T, const sa_t *mbegin, const sa_t *mend, const int64_t *mpre, int nmatch, int64_t total);
// tile_sub[t] = the sub-index that holds rank t * RV_SPLIT_TILE (sub_start ascending, nsubs entries)
int rv_tile_sub_launch(Workspace &ws, const int64_t *sub_start, int nsubs, int *tile_sub, int64_t ntiles);
// children up to this many ranks are replayed by a 256-thread workgroup, larger ones by 1024 threads (8 against 2 workgroups per CU: the replay is
// a chain of latencies, so the level is done sooner with more of them resident -- C4 bubble 42.5 -> 40.0 ms from 16 K to 64 K, 41.5 at 256 K, 44.9 at 8 K)
#ifndef RV_BUBBLE_BIG_N
#define RV_BUBBLE_BIG_N 65536
#endif
// children up to this many ranks are bubbled on LDS copies of their arrays (one workgroup, all cuts)
#define RV_BUBBLE_LDS_N0 2048
#define RV_BUBBLE_LDS_N1 4096
#ifndef RV_BUBBLE_LDS_N2
#define RV_BUBBLE_LDS_N2 8192
#endif
#define RV_BUBBLE_LDS_N RV_BUBBLE_LDS_N2
// leading children above this many ranks take the data-parallel rounds (rv_bubble.hip); measured on C2: 16 K -> 485 Mbp/s,
// 256 K -> 541, 512 K -> 555, 1 M -> 550, 2 M -> 511 (below it one workgroup replays the cuts of a child faster than ~18 launches)
#define RV_LEVEL_BUFS 3            // level arrays of the recursion, in rotation (rv_align.hip)
#define RV_BUBBLE_PAR_N 786432
// all cuts of each (non-huge) leading child in one workgroup; descriptors use off, n, cut0, cut1 (cut windows in order)
int rv_bubble_children_launch(Workspace &ws, const RvBubbleArgs &b, const RvBubbleDesc *d_small, int nsmall, const RvBubbleDesc *d_big, int nbig);
int rv_lower_ranges_launch(Workspace &ws, uint8_t *T, const sa_t *mbegin, const sa_t *mend, int nranges);
int rv_bubble_children_dev_launch(Workspace &ws, const RvBubbleArgs &b, const RvBubbleDesc *d_desc, int count, int64_t max_n);
int rv_bubble_children_lds_launch(Workspace &ws, const RvBubbleArgs &b, const RvBubbleDesc *d_lds, const int *count3);
// device-built descriptors (one per sub-index), every size class up to max_n: LDS kernels on ws_lds' stream, one-workgroup kernels on ws_kid's
int rv_bubble_children_dev_classes_launch(Workspace &ws_lds, Workspace &ws_kid, const RvBubbleArgs &b, const RvBubbleDesc *d_desc, int count, int64_t max_n);   // descriptors sorted by size class
// one cut of every child in descriptors [first, first+count): data-parallel (rv_bubble.hip)
// refresh_tmin: an earlier round of this level ran the sequential kernels on some of these children (they do not keep the tile
// bounds): lower the bounds to the values now in place first
int rv_bubble_par_round_launch(Workspace &ws, const RvBubbleArgs &b, int first, int count, int64_t total_window, int64_t total_tiles, bool refresh_tmin);
int rv_bubble_window_launch(Workspace &ws, const RvBubbleArgs &b, int first, int count, int64_t total_window);
// sequential kernels for the (child, cut)s of a round the parallel path left alone (more than RV_PB_CAP candidates)
int rv_bubble_seq_launch(Workspace &ws, const RvBubbleArgs &b, int first, int count);
int rv_sai_level_launch(Workspace &ws, const sa_t *SA, int64_t m, const int64_t *sub_start, int nsubs, sa_t *SAi);

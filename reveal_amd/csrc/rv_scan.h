// rv_scan.h -- scan kernels' host interface (see rv_scan.hip)
#pragma once
#include "rv_common.h"

#ifndef RV_PAIR_TILE
#define RV_PAIR_TILE 1024    // ranks per tile of the pair scan = what one wave scans (sixteen per lane)
#endif
#define RV_TSUB_TILE 2048    // granularity of the tile -> sub-index tables the host ships (== RV_SPLIT_TILE)

// one pairwise MUM: a < b text positions, l = LCP[rank], rank inside the
// scanned (concatenated) array
struct RvPairRec {
    sa_t a, b;
    u32  l;
    u32  rank;
};

#define RV_PAIR_SLOTS 16
// Streams SA/LCP/BWT[0..m) once.  The first RV_PAIR_SLOTS survivors of tile t
// (1024 ranks, rank order) go to slots[t*RV_PAIR_SLOTS ..], further ones to
// ovf[tileovf[t] ..] (*ovf_counter must be zero: rv_pair_compact_launch leaves it so);
// tilecnt[t] = number of survivors, tilecnt[ntile] = 0.  rv_pair_compact_launch packs them densely in rank order given
// tileoff = exclusive scan of tilecnt.
// nsubs > 0: also initialises the tables of the device-side picker (best, picks) that rv_pick_slots_launch fills.
int rv_scan_pair_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, int64_t m, const uint8_t *BWT, sa_t nsep0, int minl,
                        RvPairRec *slots, RvPairRec *ovf, u32 ovf_cap, u32 *ovf_counter, u32 *tilecnt, u32 *tileovf,
                        unsigned long long *best, RvPairRec *picks, int nsubs,
                        hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);   // both given: the kernel's own start / stop (hipExtLaunchKernelGGL)
// built-in picker straight from the slots (no compaction): picks[0] = header {0, overflow count, *err, 0}, picks[1+s] = longest
// record of sub-index s (smallest a on ties), rank 0xFFFFFFFF = none; resets *ovf_counter
int rv_pick_slots_launch(Workspace &ws, const RvPairRec *slots, const RvPairRec *ovf, u32 ovf_cap, const u32 *tilecnt, const u32 *tileovf, int64_t ntile,
                         const int64_t *sub_start, int nsubs, unsigned long long *best, RvPairRec *picks, u32 *ovf_counter, const u32 *err,
                         const int *tile_sub, int64_t ntsub);      // tile_sub (optional): sub-index of the first rank of every RV_TSUB_TILE ranks
// out holds RV_PAIR_HDR header records ({total, overflow count, *err, 0} as u32) followed by the packed records.
// Resets *ovf_counter for the next scan.
#define RV_PAIR_HDR 1
int rv_pair_compact_launch(Workspace &ws, const RvPairRec *slots, const RvPairRec *ovf, const u32 *tilecnt, const u32 *tileovf,
                           const u32 *tileoff, int64_t ntile, RvPairRec *out, u32 out_cap, u32 *ovf_counter, const u32 *err, u32 ovf_cap);

#define RV_MULTI_TILE 512    // ranks per tile of the multi-MUM scan = what one wave scans
struct RvMultiRec { u32 l, n, ub, pad; };
// Multi-MUM scan (getmultimums, reveal.c:436-580).  Records and members of
// tile t (512 ranks) land at rec[tab.x .. +tab.y) / so,pos[tab.z .. +tab.w) in
// the reference's emission order; counters[0..1] must be zeroed.
int rv_scan_multi_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, int64_t m, const uint8_t *BWT, const sa_t *nsep, int nsamples,
                         int minl, int minn, RvMultiRec *rec, uint16_t *so, sa_t *pos, u32 rec_cap, u32 mem_cap, u32 *counters, uint4 *tiletab,
                         const int64_t *sub_start, const int *sub_want, int nsubs);   // sub_want != NULL: keep only matches with n == sub_want[sub of ub]

// getmultimems (reveal.c:292-434) replayed by one wavefront (rv_mems.hip); counts beyond the capacities are still counted
int rv_multimems_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t n, const sa_t *nsep, int nsamples, int minl, int minn,
                        u32 maxlcp, u32 *rec_l, int32_t *rec_c, int64_t *rec_first, uint16_t *so, sa_t *pos,
                        unsigned long long rec_cap, unsigned long long mem_cap, unsigned long long *out);

// Built-in picker for more than two samples: per sub-index the longest match present in every one of its samples
// (ties: smallest minimum position).  pick_l[s] = its length (0 = none), pick_pos[s*nsamples ..] = its members in SA order.
// *cand_count > cand_cap afterwards: the candidate list was too small, grow and rerun.
struct RvMultiCand;
int rv_multi_pick_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, int64_t m, const uint8_t *BWT, const sa_t *nsep, int nsamples, int minl, int minn,
                         const int64_t *sub_start, const int *sub_want, int nsubs, const int *tile_sub /* sub-index of rank t * RV_TSUB_TILE */,
                         unsigned long long *best, u32 *pick_l, sa_t *pick_pos, RvMultiCand *cand, u32 cand_cap, u32 *cand_count,
                         hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);      // both given: start / stop of the streaming kernel itself
#define RV_MULTI_CAND_BYTES 16
// Full matches of a whole index with k samples (2 <= k <= 16; the anchor cascade's root list, rv_cascade_multi.hip): every LCP interval of exactly k
// ranks with a value of minl or more whose members are of k different samples and left-maximal.  Entry i of region r (nregions a power of two, counters
// region_cnt[r * cnt_stride] zeroed by the caller -- 64 words apart they sit in different L2 channels; entries beyond rcap are counted, not stored)
// lives at r * rcap + i: c_len = the value, c_pos[.. * k + s] = the member of sample s.
int rv_full_list_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t n, const sa_t *nsep, int k, u32 minl,
                        u32 *c_len, sa_t *c_pos, u32 rcap, u32 *region_cnt, int nregions, int cnt_stride, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
#define RV_MULTI_REGIONS 64      // the picker's candidate list: regions with a counter each (counter r at word 64 * r, the largest count at word 64 * RV_MULTI_REGIONS)

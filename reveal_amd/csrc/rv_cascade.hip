// rv_cascade.hip -- the anchor cascade: the built-in recursion of a two-sample alignment decided from the
// TOP-LEVEL match list wherever that provably is what the reference's recursion does.
//
// The reference (reveallib/reveal.c:731-1338 with the benchmark callbacks of SURVEY 8(d): longest match of the
// sub-index, ties to the smallest coordinate, linear interval model) scans every sub-index again after every split.
// For related genomes nearly all of those scans only find what the first scan already found: a sub-index C that
// descends from the root X (one interval per sample, [a0,a1) + [b0,b1)) holds the suffixes of X that start inside
// its intervals, cut at the intervals' ends (split reveal.c:582-664 + bubble_sort :666-727), so
//
//   * gap i of X (ranks i-1, i) is a UNIQUE CROSS PAIR iff LCP[i] > LCP[i-1], LCP[i] > LCP[i+1] and the two suffixes
//     start on different sides of nsep[0]; the matches of X (reveal.c:61-85) are its left-maximal unique cross pairs;
//   * W[p], p = SA[j]: the longest prefix suffix p shares with any suffix of X other than its cross-pair partner
//       = max(LCP[j-1], LCP[j+1]) if gap j is such a pair, max(LCP[j], LCP[j+2]) if gap j+1 is, else max(LCP[j], LCP[j+1]);
//   * every match of C longer than Wmax(C) = max W over C's positions is a match of X cut to C -- shifted to start
//     behind the matched text in front of C (whose last base is lower case by then: left-maximal, reveal.c:81-85) and
//     capped at C's ends -- and every such cut match longer than Wmax(C) is a match of C (no third suffix of X shares
//     that many characters with either of its suffixes).
//
// So the choice in C is known from X's match list whenever C's best cut match is longer than Wmax(C); C has no match
// at all when it holds no cut match of minl characters and Wmax(C) < minl, or when one of its intervals is shorter
// than minl.  Whatever is left undecided (a repeat inside the sub-index that is as long as its best match) is rebuilt
// from its own text -- its arrays only depend on the text of its two intervals -- and handed to the leaf kernel
// (rv_leaf.hip), which runs the literal recursion on it.  An undecided sub-index above the leaf kernel's size makes the
// cascade give up before anything is written; the level pipeline (rv_align.hip) then runs as if it had never started.
//
// tools/cascade_proto.py is the same algorithm on the CPU beside the oracle's literal recursion (random inputs with
// SNPs, indels, tandem repeats, N runs, identical copies); tests/test_gpu_cascade.py and tools/fuzz.py run this file
// against the oracle with the cascade on and off.
//
// Level-synchronous like the pipeline it replaces, but on the match list (2.5 x 10^6 records at 2 x 250 Mbp) and the
// witness list (positions with W >= minl) instead of the index: per level three small kernels --
//   k_cas_assign   every live match / witness moves to the child of its sub-index that holds it (or dies) and bids for
//                  that child's best match / raises its Wmax (one atomic per run of equal children in a wave)
//   k_cas_winner   the match that holds a child's best bid writes its cut coordinates
//   k_cas_decide   one thread per sub-index of the level: split (anchor out, children made), ended, or undecided
#include "rv_index.h"
#include "rv_cascade.h"
#include "rv_leaf.h"
#include <algorithm>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

namespace {

constexpr int TB = 256;
constexpr u32 NONE = 0xFFFFFFFFu;
#ifdef RV_SA64
constexpr int KEY_SHIFT = 40;                 // bid = length << 40 | (2^40 - 1 - position): lengths below 2^24, positions below 2^40
#else
constexpr int KEY_SHIFT = 32;
#endif
constexpr u64 KEY_LOW = (1ull << KEY_SHIFT) - 1;

struct CasIv { sa_t a0, a1, b0, b1; };
struct CasRes { sa_t qa, qb; u32 ql, lead, trail, state; };      // state: 0 not decided yet, 1 split, 2 ended
enum { C_NCHILD = 0, C_NUND = 1, C_NWIT = 2, C_ERR = 3, C_MAXN = 4, C_LO = 5, C_HI = 6, C_LEVELS = 7, C_NSOLVED = 8, C_NUNSOLVED = 9, C_NRETRY = 10, C_TICKET = 11 };

// the match (pa, pb, len) of the root cut to a sub-index: start shifted behind the sub-index' begin on both sides, length
// capped at its ends
__device__ inline bool cas_cut(const CasIv &iv, int64_t pa, int64_t pb, int64_t len, int64_t minl, int64_t *qa, int64_t *qb, int64_t *ql) {
    const int64_t ka = (int64_t)iv.a0 - pa, kb = (int64_t)iv.b0 - pb;
    int64_t k = ka > kb ? ka : kb;
    k = k > 0 ? k : 0;
    const int64_t a = pa + k, b = pb + k;
    int64_t l = len - k;
    const int64_t ra = (int64_t)iv.a1 - a, rb = (int64_t)iv.b1 - b;
    l = l < ra ? l : ra;
    l = l < rb ? l : rb;
    *qa = a; *qb = b; *ql = l;
    return l >= minl;
}
__device__ inline u64 cas_key(int64_t qa, int64_t ql) { return ((u64)ql << KEY_SHIFT) | (KEY_LOW - (u64)qa); }

__device__ inline u64 shfl_up64(u64 v, int d) {
    const u32 lo = __shfl_up((u32)v, d, 64), hi = __shfl_up((u32)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}
// atomicMax(dst[child], val) for the active lanes of a wave, one atomic per run of equal children (the lists are sorted by
// position and a sub-index is an interval, so its members sit next to each other), and none when the bid cannot raise the word
__device__ inline void seg_atomic_max64(u64 *__restrict__ dst, u32 child, u64 val, bool active) {
    const int lane = threadIdx.x & 63;
    if (!active) { child = NONE; val = 0; }
    val = rv_wave_seg_max_u64(child, val);
    const u32 nc = __shfl_down(child, 1, 64);
    const bool last = lane == 63 || nc != child;
    if (active && last && val > dst[child]) atomicMax((unsigned long long *)&dst[child], (unsigned long long)val);
}
__device__ inline void seg_atomic_max32(u32 *__restrict__ dst, u32 child, u32 val, bool active) {
    const int lane = threadIdx.x & 63;
    if (!active) { child = NONE; val = 0; }
    val = rv_wave_seg_max_u32(child, val);
    const u32 nc = __shfl_down(child, 1, 64);
    const bool last = lane == 63 || nc != child;
    if (active && last && val > dst[child]) atomicMax(&dst[child], val);
}

// ---- witnesses: positions whose suffix shares minl characters or more with a suffix that is not its cross-pair partner ----
constexpr int WT_ITEMS = 8;
constexpr int WT_TILE = TB * WT_ITEMS;
constexpr int WT_REGIONS = 64;
__global__ __launch_bounds__(TB) void k_cas_witness(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, const uint8_t *__restrict__ BWT, int64_t n,
                                                    u32 minl, sa_t *__restrict__ w_pos, u32 *__restrict__ w_val, u32 *__restrict__ w_rank, u32 cap /* per region */, u32 *__restrict__ counters /* one per region */) {
    __shared__ __attribute__((aligned(16))) u32 sl0[WT_TILE + 8];          // sl = sl0 + 3: LCP of ranks j0-1 .. j0+TILE+1 (0 outside the array)
    __shared__ __attribute__((aligned(16))) uint8_t ss0[WT_TILE + 32];     // ss = ss0 + 15: side bit of ranks j0-1 .. j0+TILE
    u32 *const sl = sl0 + 3;
    uint8_t *const ss = ss0 + 15;
    const int64_t j0 = (int64_t)blockIdx.x * WT_TILE;
    if (j0 + WT_TILE + 2 <= n && sizeof(lcp_t) == 4) {
        // the tile's body in 16-byte loads (its ranks start at a multiple of the tile size), the three ranks around it one by one: loaded
        // entry by entry the kernel ran at 2 TB/s
        const uint4 *L4 = reinterpret_cast<const uint4 *>(LCP + j0);
        uint4 *d4 = reinterpret_cast<uint4 *>(sl + 1);
        for (int k = threadIdx.x; k < WT_TILE / 4; k += TB) d4[k] = L4[k];
        const uint4 *B4 = reinterpret_cast<const uint4 *>(BWT + j0);
        for (int k = threadIdx.x; k < WT_TILE / 16; k += TB) {
            uint4 v = B4[k];
            v.x = (v.x >> 7) & 0x01010101u; v.y = (v.y >> 7) & 0x01010101u; v.z = (v.z >> 7) & 0x01010101u; v.w = (v.w >> 7) & 0x01010101u;
            reinterpret_cast<uint4 *>(ss + 1)[k] = v;
        }
        if (threadIdx.x == 0) { sl[0] = j0 > 0 ? (u32)LCP[j0 - 1] : 0u; ss[0] = j0 > 0 ? (uint8_t)(BWT[j0 - 1] >> 7) : (uint8_t)0; }
        if (threadIdx.x == 1) { sl[WT_TILE + 1] = (u32)LCP[j0 + WT_TILE]; ss[WT_TILE + 1] = (uint8_t)(BWT[j0 + WT_TILE] >> 7); }
        if (threadIdx.x == 2) sl[WT_TILE + 2] = (u32)LCP[j0 + WT_TILE + 1];
    } else {
        for (int k = threadIdx.x; k < WT_TILE + 3; k += TB) { const int64_t j = j0 - 1 + k; sl[k] = (j >= 0 && j < n) ? (u32)LCP[j] : 0u; }
        for (int k = threadIdx.x; k < WT_TILE + 2; k += TB) { const int64_t j = j0 - 1 + k; ss[k] = (j >= 0 && j < n) ? (uint8_t)(BWT[j] >> 7) : (uint8_t)0; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll 1
    for (int r = 0; r < WT_ITEMS; r++) {
        const int k = r * TB + threadIdx.x;
        const int64_t j = j0 + k;
        const u32 lm1 = sl[k], l0 = sl[k + 1], l1 = sl[k + 2], l2 = sl[k + 3];
        const u32 sm1 = ss[k], s0 = ss[k + 1], s1 = ss[k + 2];
        const bool pair0 = (j >= 1) & (l0 > lm1) & (l0 > l1) & (s0 != sm1);             // gap j is a unique cross pair
        const bool pair1 = (j + 1 < n) & (l1 > l0) & (l1 > l2) & (s1 != s0);            // gap j+1 is
        const u32 w = pair0 ? (lm1 > l1 ? lm1 : l1) : pair1 ? (l0 > l2 ? l0 : l2) : (l0 > l1 ? l0 : l1);
        // (the list in WT_REGIONS regions with a counter each -- k_cas_wpack makes it dense: one counter was 10^5 returning atomics on one
        // address at 2 x 250 Mbp, most of the kernel's 1.26 ms)
        const bool hit = (j < n) & (w >= minl);
        const u64 bal = __ballot(hit);
        if (bal) {
            const u32 reg = blockIdx.x & (WT_REGIONS - 1);
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&counters[reg], (u32)__popcll(bal));
            base = (u32)__shfl((int)base, 0, 64);
            if (hit) {
                const u32 i = base + (u32)__popcll(bal & lt);
                if (i < cap) { w_pos[(size_t)reg * cap + i] = SA[j]; w_val[(size_t)reg * cap + i] = w; w_rank[(size_t)reg * cap + i] = (u32)j; }
            }
        }
    }
}
__global__ __launch_bounds__(TB) void k_cas_wpack(const sa_t *__restrict__ src_pos, const u32 *__restrict__ src_val, const u32 *__restrict__ src_rank, u32 rcap,
                                                  const u32 *__restrict__ region_cnt, const u32 *__restrict__ region_off, sa_t *__restrict__ w_pos, u32 *__restrict__ w_val,
                                                  u32 *__restrict__ w_rank) {
    const u32 reg = blockIdx.y, i = blockIdx.x * TB + threadIdx.x;
    if (i >= region_cnt[reg]) return;
    const size_t f = (size_t)reg * rcap + i;
    const u32 o = region_off[reg] + i;
    w_pos[o] = src_pos[f]; w_val[o] = src_val[f]; w_rank[o] = src_rank[f];
}
// the witness list in rank order (the second attempt walks it: k_cas_dwalk)
__global__ __launch_bounds__(TB) void k_cas_wkeys(const u32 *__restrict__ w_rank, u32 NW, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i < NW) { keys[i] = (u64)w_rank[i]; vals[i] = i; }
}
__global__ __launch_bounds__(TB) void k_cas_wgather(const sa_t *__restrict__ src_pos, const u32 *__restrict__ src_val, const u32 *__restrict__ perm, u32 NW,
                                                    sa_t *__restrict__ w_pos, u32 *__restrict__ w_val) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i < NW) { const u32 f = perm[i]; w_pos[i] = src_pos[f]; w_val[i] = src_val[f]; }
}

// ---- the match list, sorted by its first coordinate ----
__global__ __launch_bounds__(TB) void k_cas_keys(const RvPairRec *__restrict__ recs, u32 M, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i < M) { keys[i] = (u64)recs[i].a; vals[i] = i; }
}
__global__ __launch_bounds__(TB) void k_cas_gather(const RvPairRec *__restrict__ recs, const u32 *__restrict__ perm, u32 M, sa_t *__restrict__ c_pa,
                                                   sa_t *__restrict__ c_pb, u32 *__restrict__ c_len, u32 *__restrict__ c_child) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= M) return;
    const RvPairRec r = recs[perm[i]];
    c_pa[i] = r.a; c_pb[i] = r.b; c_len[i] = r.l; c_child[i] = 0u;
}
__global__ void k_cas_init(CasIv *__restrict__ iv, u64 *__restrict__ best, u32 *__restrict__ wmax, int32_t *__restrict__ depth, CasRes *__restrict__ res,
                           u32 *__restrict__ counters, CasIv root, u32 *__restrict__ w_child, u32 nw, u64 *__restrict__ dbest, u32 *__restrict__ dflag, u64 *__restrict__ dceil) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        if (dbest) { dbest[0] = 0; dflag[0] = 0; dceil[0] = 0; }
        iv[0] = root; best[0] = 0; wmax[0] = 0; depth[0] = 0;
        CasRes r; r.qa = 0; r.qb = 0; r.ql = 0; r.lead = NONE; r.trail = NONE; r.state = 0;
        res[0] = r;
        counters[C_NCHILD] = 1; counters[C_NUND] = 0; counters[C_ERR] = 0; counters[C_MAXN] = 0; counters[C_LO] = 0; counters[C_HI] = 1; counters[C_LEVELS] = 0; counters[C_NSOLVED] = 0; counters[C_NUNSOLVED] = 0; counters[C_NRETRY] = 0; counters[C_TICKET] = 0;
    }
    if (i < nw) w_child[i] = 0u;
}

// ---- several sequences per sample: the lineage of "rest" sub-indices ----------------------------------------------------------
// With more than one sequence in a sample the root holds many intervals, and the linear interval model splits it into
// leading = the two touched intervals' left remainders, trailing = their right remainders, rest = every interval the match did not touch.
// The rest child is made of WHOLE sequences again, so the sub-indices with more than one interval per sample form one chain
// R0 = root, R1 = rest(R0), ..., and everything else is a sub-index with one interval per sample -- what the cascade handles.  The
// chain is decided on the host from the root's match list in descending length (ties: smallest first coordinate, as the picker breaks
// them): the choice of Rk is the first match whose two sequences are both still in Rk, provided it is longer than Wmax(Rk) (W over the
// positions of Rk's sequences: the bound of this file's header holds for any union of whole sequences -- nothing of them is cut).  Every
// choice makes an anchor and up to two one-interval-per-sample sub-indices, the ROOTS of the device cascade (k_cas_init_roots); a chain
// member the list does not decide makes the attempt give up.
struct CasRootTabs { const sa_t *beginA, *beginB; const int32_t *pickA, *pickB; const u32 *rootLead, *rootTrail; u32 nA, nB; };
__device__ inline int cas_contig_of(const sa_t *__restrict__ begins, u32 cnt, int64_t pos) {      // last sequence that begins at or in front of pos
    u32 lo = 0, hi = cnt;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if ((int64_t)begins[mid] <= pos) lo = mid + 1; else hi = mid; }
    return (int)lo - 1;
}
__global__ void k_cas_init_roots(CasIv *__restrict__ iv, u64 *__restrict__ best, u32 *__restrict__ wmax, int32_t *__restrict__ depth, CasRes *__restrict__ res,
                                 u32 *__restrict__ counters, const CasIv *__restrict__ roots, const int32_t *__restrict__ rdepth, u32 nroots,
                                 u64 *__restrict__ dbest, u32 *__restrict__ dflag, u64 *__restrict__ dceil) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        counters[C_NCHILD] = nroots; counters[C_NUND] = 0; counters[C_ERR] = 0; counters[C_MAXN] = 0; counters[C_LO] = 0; counters[C_HI] = nroots; counters[C_LEVELS] = 0;
        counters[C_NSOLVED] = 0; counters[C_NUNSOLVED] = 0; counters[C_NRETRY] = 0; counters[C_TICKET] = 0;
    }
    if (i < nroots) {
        if (dbest) { dbest[i] = 0; dflag[i] = 0; dceil[i] = 0; }
        iv[i] = roots[i]; best[i] = 0; wmax[i] = 0; depth[i] = rdepth[i];
        CasRes r; r.qa = 0; r.qb = 0; r.ql = 0; r.lead = NONE; r.trail = NONE; r.state = 0;
        res[i] = r;
    }
}
// the root a match / a witness starts in (NONE: between sequences that were not chosen together, or inside an anchor of the chain)
__global__ __launch_bounds__(TB) void k_cas_assign_roots(const sa_t *__restrict__ c_pa, const sa_t *__restrict__ c_pb, const u32 *__restrict__ c_len, u32 *__restrict__ c_child, u32 M,
                                                         const sa_t *__restrict__ w_pos, u32 *__restrict__ w_child, u32 NW, const CasIv *__restrict__ roots, CasRootTabs t, int64_t minl) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i < M) {
        const int64_t pa = (int64_t)c_pa[i], pb = (int64_t)c_pb[i], len = (int64_t)c_len[i];
        const int ca = cas_contig_of(t.beginA, t.nA, pa), cb = cas_contig_of(t.beginB, t.nB, pb);
        u32 c = NONE;
        if (ca >= 0 && cb >= 0 && t.pickA[ca] >= 0 && t.pickA[ca] == t.pickB[cb]) {
            const u32 k = (u32)t.pickA[ca];
            int64_t qa, qb, ql;
            if (t.rootLead[k] != NONE && cas_cut(roots[t.rootLead[k]], pa, pb, len, minl, &qa, &qb, &ql)) c = t.rootLead[k];
            else if (t.rootTrail[k] != NONE && cas_cut(roots[t.rootTrail[k]], pa, pb, len, minl, &qa, &qb, &ql)) c = t.rootTrail[k];
        }
        c_child[i] = c;
    }
    if (i < NW) {
        const int64_t pos = (int64_t)w_pos[i];
        const bool a_side = t.nB == 0 || pos < (int64_t)t.beginB[0];
        const int ct = a_side ? cas_contig_of(t.beginA, t.nA, pos) : cas_contig_of(t.beginB, t.nB, pos);
        u32 c = NONE;
        if (ct >= 0) {
            const int32_t k = a_side ? t.pickA[ct] : t.pickB[ct];
            if (k >= 0) {
                for (int which = 0; which < 2 && c == NONE; which++) {
                    const u32 r = which ? t.rootTrail[k] : t.rootLead[k];
                    if (r == NONE) continue;
                    const CasIv p = roots[r];
                    if ((pos >= (int64_t)p.a0 && pos < (int64_t)p.a1) || (pos >= (int64_t)p.b0 && pos < (int64_t)p.b1)) c = r;
                }
            }
        }
        w_child[i] = c;
    }
}
__global__ void k_cas_lineage_stats(unsigned long long *__restrict__ stats, unsigned long long steps, unsigned long long maxdepth) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { atomicAdd(&stats[0], steps); atomicMax(&stats[3], maxdepth); }
}
// matches in the picker's order: longest first, ties to the smallest first coordinate
__global__ __launch_bounds__(TB) void k_cas_lkeys(const sa_t *__restrict__ c_pa, const u32 *__restrict__ c_len, u32 M, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i < M) { keys[i] = (((1ull << (64 - KEY_SHIFT)) - 1ull - (u64)c_len[i]) << KEY_SHIFT) | (u64)c_pa[i]; vals[i] = i; }
}
__global__ __launch_bounds__(TB) void k_cas_lgather(const sa_t *__restrict__ c_pa, const sa_t *__restrict__ c_pb, const u32 *__restrict__ c_len, const u32 *__restrict__ perm, u32 M,
                                                    sa_t *__restrict__ o_pa, sa_t *__restrict__ o_pb, u32 *__restrict__ o_len) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i < M) { const u32 f = perm[i]; o_pa[i] = c_pa[f]; o_pb[i] = c_pb[f]; o_len[i] = c_len[f]; }
}

// ---- one level ----
constexpr int CAS_ITEMS = 4;
__global__ __launch_bounds__(TB) void k_cas_assign(const sa_t *__restrict__ c_pa, const sa_t *__restrict__ c_pb, const u32 *__restrict__ c_len, u32 *__restrict__ c_child,
                                                   u32 M, const sa_t *__restrict__ w_pos, const u32 *__restrict__ w_val, u32 *__restrict__ w_child, u32 NW,
                                                   const CasIv *__restrict__ iv, const CasRes *__restrict__ res, u64 *__restrict__ best, u32 *__restrict__ wmax,
                                                   int64_t minl, int first, const u64 *__restrict__ ceil /* second attempt: bids stay below it (0: none) */) {
    // (the grid covers the matches in whole workgroups, then the witnesses.)  A workgroup takes CAS_ITEMS stretches of TB entries, a wave 64 consecutive
    // ones at a time: at the deep levels most entries are dead, and a workgroup per 256 of them was bound by the dispatch of 8 600 workgroups that
    // only read one word each -- 33 us per level however few matches were alive.  Every stretch is a chain of dependent trips to memory (the entry's
    // sub-index, its record, the bid), so more stretches per workgroup cost what fewer workgroups save: 36 / 26.8 / 25.0 us per level with 16 / 8 / 4
    // (same box; issuing all loads of the stretches first: 30 us -- dead entries load too)
    const u32 mblocks = (M + TB * CAS_ITEMS - 1) / (TB * CAS_ITEMS);
    for (int it = 0; it < CAS_ITEMS; it++) {
    if (blockIdx.x < mblocks) {
        const u32 i = (blockIdx.x * CAS_ITEMS + (u32)it) * TB + threadIdx.x;
        u32 c = i < M ? c_child[i] : NONE;
        bool live = c != NONE;
        u64 key = 0;
        if (live) {
            const int64_t pa = (int64_t)c_pa[i], pb = (int64_t)c_pb[i], len = (int64_t)c_len[i];
            int64_t qa, qb, ql;
            if (!first) {
                const CasRes r = res[c];
                const CasIv p = iv[c];
                u32 nc = NONE;
                if (r.state == 1u) {
                    // at most one of the two children holds it: a match that covers the parent's choice would be longer than it
                    CasIv lv; lv.a0 = p.a0; lv.a1 = r.qa; lv.b0 = p.b0; lv.b1 = r.qb;
                    CasIv tv; tv.a0 = (sa_t)((int64_t)r.qa + r.ql); tv.a1 = p.a1; tv.b0 = (sa_t)((int64_t)r.qb + r.ql); tv.b1 = p.b1;
                    if (r.lead != NONE && cas_cut(lv, pa, pb, len, minl, &qa, &qb, &ql)) nc = r.lead;
                    else if (r.trail != NONE && cas_cut(tv, pa, pb, len, minl, &qa, &qb, &ql)) nc = r.trail;
                } else if (r.state == 3u) {      // the same sub-index once more, its best cut match set aside (k_cas_decide)
                    if (cas_cut(p, pa, pb, len, minl, &qa, &qb, &ql)) nc = r.lead;
                }
                c = nc;
                c_child[i] = c;
                live = c != NONE;
            } else {
                live = cas_cut(iv[c], pa, pb, len, minl, &qa, &qb, &ql);
                if (!live) { c = NONE; c_child[i] = NONE; }
            }
            if (live) key = cas_key(qa, ql);
        }
        bool bid = live;
        if (live && ceil) { const u64 ce = ceil[c]; bid = ce == 0 || key < ce; }
        seg_atomic_max64(best, c, key, bid);
    } else {
        const u32 i = ((blockIdx.x - mblocks) * CAS_ITEMS + (u32)it) * TB + threadIdx.x;
        u32 c = i < NW ? w_child[i] : NONE;
        bool live = c != NONE;
        u32 v = 0;
        if (live) {
            const int64_t pos = (int64_t)w_pos[i];
            v = w_val[i];
            if (!first) {
                const CasRes r = res[c];
                const CasIv p = iv[c];
                u32 nc = NONE;
                if (r.state == 1u) {
                    const int64_t ea = (int64_t)r.qa + r.ql, eb = (int64_t)r.qb + r.ql;
                    const bool in_lead = (pos >= p.a0 && pos < r.qa) || (pos >= p.b0 && pos < r.qb);
                    const bool in_trail = (pos >= ea && pos < p.a1) || (pos >= eb && pos < p.b1);
                    nc = in_lead ? r.lead : in_trail ? r.trail : NONE;
                } else if (r.state == 3u) nc = r.lead;
                c = nc;
                w_child[i] = c;
                live = c != NONE;
            } else {
                const CasIv p = iv[c];
                live = (pos >= p.a0 && pos < p.a1) || (pos >= p.b0 && pos < p.b1);
                if (!live) { c = NONE; w_child[i] = NONE; }
            }
        }
        seg_atomic_max32(wmax, c, v, live);
    }
    }
}

__global__ __launch_bounds__(TB) void k_cas_winner(const sa_t *__restrict__ c_pa, const sa_t *__restrict__ c_pb, const u32 *__restrict__ c_len,
                                                   const u32 *__restrict__ c_child, u32 M, const CasIv *__restrict__ iv, CasRes *__restrict__ res,
                                                   const u64 *__restrict__ best, int64_t minl) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= M) return;
    const u32 c = c_child[i];
    if (c == NONE) return;
    int64_t qa, qb, ql;
    if (!cas_cut(iv[c], (int64_t)c_pa[i], (int64_t)c_pb[i], (int64_t)c_len[i], minl, &qa, &qb, &ql)) return;
    if (cas_key(qa, ql) == best[c]) { res[c].qa = (sa_t)qa; res[c].qb = (sa_t)qb; res[c].ql = (u32)ql; }
}

// ---- the second attempt: large undecided sub-indices decided from the witnesses ----------------------------------------------
// A sub-index C the list does not decide (its best cut match, ell characters, is no longer than Wmax(C)) and the leaf kernel cannot
// take.  With theta = ell (minl when C holds no cut match) and D = {p in C: W[p] >= theta, theta characters left before C's end}:
//   * a match of C of theta characters or more that is no cut match has both its suffixes in D (in X a third suffix shares its
//     string, or it would be one of X's matches), and so has every other occurrence of its string inside C;
//   * C's best cut match is a match of C if its suffixes are outside D (no other suffix of X shares ell characters with them);
//     inside D it may have a second occurrence in C -- which then is in D as well;
//   * hence the gaps of theta and more among D's suffixes, cut at C's ends, are those of C's own index: C's choice is the best of
//     the unique cross pairs of D and the cut match if that is outside D; nothing found and no cut match: C has no match.
// D is not sorted: the witness list is in X's rank order, so the prefix a member shares with the members of its sub-index falls with
// their distance in the list -- every member walks outwards for the two longest prefixes it shares (cut at C's ends: a suffix near
// an end may share less than a farther one, the walk goes on behind those; the prefixes themselves come from X's LCP array),
// k_cas_dpick keeps the pairs that are each other's only longest partner.  A cut match inside D that the walk does not confirm (ell was not the length of C's
// best match): C is looked at again in the next level with that match set aside (bids capped below it), up to sixteen times.
constexpr u32 DWALK = 1024;
struct CasDanger { u64 *best; u32 *flag; sa_t *qb; u64 *ceil; u32 *m1, *p1, *m2; u64 *key; const uint8_t *T0; const lcp_t *LCP; const u64 *rank; u32 leaf_n; uint8_t *dflag; const uint8_t *wcls; };
__device__ inline bool cas_is_danger(const CasIv &p, u64 bk, u32 wm, u32 minl, u32 leaf_n, u32 *theta) {
    const int64_t la = (int64_t)p.a1 - p.a0, lb = (int64_t)p.b1 - p.b0;
    const u32 bl = (u32)(bk >> KEY_SHIFT);
    const bool can = (la >= (int64_t)minl) & (lb >= (int64_t)minl);
    const bool split = can & (bl >= minl) & (bl > wm);
    *theta = bl >= minl ? bl : minl;
    return can & !split & (wm >= minl) & ((u64)(la + lb) > (u64)leaf_n);
}
// A byte per sub-index of the level: the second attempt looks at it (cas_is_danger).  The three kernels below run over EVERY witness at every level,
// and nearly all of them sit in sub-indices the match list decides: read per witness, the sub-index' intervals, best bid and Wmax were three random
// sectors each -- 1.4 ms per level for 2 x 10^7 witnesses (2 x 250 Mbp with 2 % repeats: 79 of the run's 173 ms) -- where this byte stays in the L2.
// (A live witness always sits in a sub-index of the current level: k_cas_assign moves it to a child made by the last k_cas_decide or lets it die.)
__global__ __launch_bounds__(TB) void k_cas_dmark(const CasIv *__restrict__ iv, const u64 *__restrict__ best, const u32 *__restrict__ wmax, u32 minl, const u32 *__restrict__ counters, CasDanger dg) {
    const u32 lo = counters[C_LO], hi = counters[C_HI];
    for (u32 id = lo + blockIdx.x * TB + threadIdx.x; id < hi; id += gridDim.x * TB) {
        u32 theta;
        dg.dflag[id] = cas_is_danger(iv[id], best[id], wmax[id], minl, dg.leaf_n, &theta) ? 1 : 0;
    }
}
__global__ __launch_bounds__(TB) void k_cas_dwalk(const sa_t *__restrict__ w_pos, const u32 *__restrict__ w_val, const u32 *__restrict__ w_child, u32 NW,
                                                  const CasIv *__restrict__ iv, const u64 *__restrict__ best, const u32 *__restrict__ wmax, u32 minl, CasDanger dg) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= NW) return;
    if (dg.wcls && dg.wcls[i]) return;          // (a witness of a medium block: k_cas_dwalk_blk walks it in LDS)
    const u32 c = w_child[i];
    if (c == NONE || !dg.dflag[c]) return;      // (m1 / key of such a witness are never read: k_cas_dpick and k_cas_dwrite leave through the same door)
    dg.m1[i] = 0; dg.key[i] = 0;
    const CasIv p = iv[c];
    u32 theta;
    if (!cas_is_danger(p, best[c], wmax[c], minl, dg.leaf_n, &theta)) return;
    const int64_t pos = (int64_t)w_pos[i];
    const int64_t cap = (pos < (int64_t)p.a1 ? (int64_t)p.a1 : (int64_t)p.b1) - pos;
    if (w_val[i] < theta || cap < (int64_t)theta) return;
    u32 b1 = 0, b2 = 0, part = NONE;
    bool over = false;
    // The prefix shared with the suffix k places away in X's order is the smallest LCP value in between.  A rank that is no witness
    // shares less than minl with every suffix but its partner, so nothing behind it shares theta with this suffix: the walk only ever
    // crosses consecutive ranks, and reads X's LCP array instead of the text.
#pragma unroll 1
    for (int dir = -1; dir <= 1; dir += 2) {
        u32 steps = 0;
        u64 rk = dg.rank[i];
        u32 run = 0xFFFFFFFFu;
        for (int64_t k = (int64_t)i + dir; k >= 0 && k < (int64_t)NW; k += dir) {
            const u64 r2 = dg.rank[k];
            if (r2 != rk + (u64)(int64_t)dir) break;
            const u32 g = (u32)dg.LCP[dir > 0 ? r2 : rk];
            run = g < run ? g : run;
            rk = r2;
            if (run < theta) break;
            if (++steps > DWALK) { over = true; break; }
            if (w_child[k] != c || w_val[k] < theta) continue;
            const int64_t pk = (int64_t)w_pos[k];
            const int64_t ck = (pk < (int64_t)p.a1 ? (int64_t)p.a1 : (int64_t)p.b1) - pk;
            if (ck < (int64_t)theta) continue;
            u32 l = (u32)(cap < ck ? cap : ck);
            l = run < l ? run : l;
            if (l > b1) { b2 = b1; b1 = l; part = (u32)k; }
            else if (l > b2) b2 = l;
            if (run <= b2) break;      // (what lies behind shares no more than this)
        }
    }
    if (over) atomicOr(&dg.flag[c], 2u);
    dg.m1[i] = b1 >= theta ? b1 : 0u; dg.p1[i] = part; dg.m2[i] = b2;
}
// ---- the same walk, block by block in LDS -------------------------------------------------------------------------------------
// The witnesses stand in the root's rank order; a walk only ever crosses consecutive ranks joined by LCP values of minl and more, so the list
// falls into BLOCKS no walk leaves -- the suffixes that share a repeat's position: a few for a chance repeat, several hundred for a position of
// a mobile element with hundreds of copies.  Membership in a block never changes; only the sub-index of a witness, theta and the interval ends do.
// In global memory every step of every walk was five dependent reads (rank, LCP, sub-index, value, position): 2 x 10^7 witnesses x hundreds of
// steps x 40 levels = 80 of the 172 ms a 2 x 250 Mbp pair with 2 % repeats took.  A medium block (BLK_MIN < size <= BLK_MAX) is staged once per level
// by ONE workgroup -- sub-index, value, position, LCP with the rank in front -- and its members walk in LDS, step for step what k_cas_dwalk does
// (the same breaks, the same two best partners, the same ties).  Small and large blocks keep the global walk.
constexpr u32 BLK_MIN = 64, BLK_MAX = 1024;
__global__ __launch_bounds__(TB) void k_cas_blk_flags(const u64 *__restrict__ rank, const lcp_t *__restrict__ LCP, u32 NW, u32 minl, u32 *__restrict__ flag) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= NW) return;
    flag[i] = (i == 0 || rank[i] != rank[i - 1] + 1 || (u32)LCP[rank[i]] < minl) ? 1u : 0u;
}
// boff[b] = first witness of block b (bid = exclusive sum of the flags: the block of witness i is bid[i] + flag[i] - 1)
__global__ __launch_bounds__(TB) void k_cas_blk_off(const u32 *__restrict__ flag, const u32 *__restrict__ bid, u32 NW, u32 nb, u32 *__restrict__ boff) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i == 0) boff[nb] = NW;
    if (i < NW && flag[i]) boff[bid[i]] = i;
}
// medium blocks: a flag per block (for the list) and a byte per witness (for the global walk, which leaves them out)
__global__ __launch_bounds__(TB) void k_cas_blk_class(const u32 *__restrict__ flag, const u32 *__restrict__ bid, const u32 *__restrict__ boff, u32 NW, u32 *__restrict__ mflag, uint8_t *__restrict__ wcls) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= NW) return;
    const u32 b = bid[i] + flag[i] - 1u;
    const u32 size = boff[b + 1] - boff[b];
    const bool med = size > BLK_MIN && size <= BLK_MAX;
    wcls[i] = med ? 1 : 0;
    if (flag[i]) mflag[b] = med ? 1u : 0u;
}
__global__ __launch_bounds__(TB) void k_cas_blk_list(const u32 *__restrict__ mflag, const u32 *__restrict__ mid, u32 nb, u32 *__restrict__ mlist) {
    const u32 b = blockIdx.x * TB + threadIdx.x;
    if (b < nb && mflag[b]) mlist[mid[b]] = b;
}
__global__ __launch_bounds__(TB) void k_cas_dwalk_blk(const sa_t *__restrict__ w_pos, const u32 *__restrict__ w_val, const u32 *__restrict__ w_child, const u32 *__restrict__ boff,
                                                      const u32 *__restrict__ mlist, u32 nmed, const CasIv *__restrict__ iv, const u64 *__restrict__ best, const u32 *__restrict__ wmax,
                                                      u32 minl, CasDanger dg) {
    // a workgroup per block (one wavefront per block left the members of a block of 900 fourteen walks each, one after the other, and the LDS of four
    // blocks per workgroup kept the CUs at eight waves: 4.2 ms per level where the global walk took 1.4)
    __shared__ u32 cc[BLK_MAX], vv[BLK_MAX], ll[BLK_MAX];
    __shared__ sa_t pp[BLK_MAX];
    __shared__ u32 s_any;
    const u32 b = mlist[blockIdx.x];
    const u32 s0 = boff[b], size = boff[b + 1] - s0;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    bool any = false;
    for (u32 j = threadIdx.x; j < size; j += TB) {
        const u32 c = w_child[s0 + j];
        cc[j] = c;
        any |= c != NONE && dg.dflag[c];
    }
    if (__ballot(any) && (threadIdx.x & 63) == 0) s_any = 1;
    __syncthreads();
    if (!s_any) return;      // nobody of this block sits in a sub-index the second attempt looks at
    for (u32 j = threadIdx.x; j < size; j += TB) {
        vv[j] = w_val[s0 + j]; pp[j] = w_pos[s0 + j];
        ll[j] = (u32)dg.LCP[dg.rank[s0 + j]];      // common prefix with the rank in front
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < size; j += TB) {
        const u32 c = cc[j];
        if (c == NONE || !dg.dflag[c]) continue;
        const u32 i = s0 + j;
        dg.m1[i] = 0; dg.key[i] = 0;
        const CasIv p = iv[c];
        u32 theta;
        if (!cas_is_danger(p, best[c], wmax[c], minl, dg.leaf_n, &theta)) continue;
        const int64_t pos = (int64_t)pp[j];
        const int64_t cap = (pos < (int64_t)p.a1 ? (int64_t)p.a1 : (int64_t)p.b1) - pos;
        if (vv[j] < theta || cap < (int64_t)theta) continue;
        u32 b1 = 0, b2 = 0, part = NONE;
#pragma unroll 1
        for (int dir = -1; dir <= 1; dir += 2) {
            u32 run = 0xFFFFFFFFu;
            for (int64_t k = (int64_t)j + dir; k >= 0 && k < (int64_t)size; k += dir) {
                const u32 g = ll[dir > 0 ? k : k + 1];
                run = g < run ? g : run;
                if (run < theta) break;
                if (cc[k] != c || vv[k] < theta) continue;
                const int64_t pk = (int64_t)pp[k];
                const int64_t ck = (pk < (int64_t)p.a1 ? (int64_t)p.a1 : (int64_t)p.b1) - pk;
                if (ck < (int64_t)theta) continue;
                u32 l = (u32)(cap < ck ? cap : ck);
                l = run < l ? run : l;
                if (l > b1) { b2 = b1; b1 = l; part = s0 + (u32)k; }
                else if (l > b2) b2 = l;
                if (run <= b2) break;      // (what lies behind shares no more than this)
            }
        }
        dg.m1[i] = b1 >= theta ? b1 : 0u; dg.p1[i] = part; dg.m2[i] = b2;
    }
}
__global__ __launch_bounds__(TB) void k_cas_dpick(const sa_t *__restrict__ w_pos, const u32 *__restrict__ w_val, const u32 *__restrict__ w_child, u32 NW,
                                                  const CasIv *__restrict__ iv, const CasRes *__restrict__ res, const u64 *__restrict__ best, const u32 *__restrict__ wmax,
                                                  u32 minl, CasDanger dg) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= NW) return;
    const u32 c = w_child[i];
    if (c == NONE || !dg.dflag[c]) return;
    const CasIv p = iv[c];
    u32 theta;
    const u64 bk = best[c];
    if (!cas_is_danger(p, bk, wmax[c], minl, dg.leaf_n, &theta)) return;
    const int64_t pos = (int64_t)w_pos[i];
    const int64_t cap = (pos < (int64_t)p.a1 ? (int64_t)p.a1 : (int64_t)p.b1) - pos;
    if (w_val[i] < theta || cap < (int64_t)theta) return;
    // a member of D: is it one of the best cut match's suffixes?
    if ((u32)(bk >> KEY_SHIFT) >= minl) { const CasRes r = res[c]; if (pos == (int64_t)r.qa || pos == (int64_t)r.qb) atomicOr(&dg.flag[c], 1u); }
    const u32 b1 = dg.m1[i];
    if (b1 == 0) return;
    const u32 w = dg.p1[i];
    if (dg.m1[w] != b1 || dg.p1[w] != i || b1 <= dg.m2[i] || b1 <= dg.m2[w]) return;
    const int64_t pw = (int64_t)w_pos[w];
    const bool a_side = pos < (int64_t)p.a1, w_a_side = pw < (int64_t)p.a1;
    if (!a_side || w_a_side) return;      // (the pair reports once, from its suffix in the first sample)
    const int64_t qa = pos, qb = pw;
    if (!(qa == (int64_t)p.a0 || qb == (int64_t)p.b0)) {      // reveal.c:81-85 on the working text: the base in front of an interval that starts behind an anchor is lower case
        const uint8_t ca = dg.T0[qa - 1];
        if (ca == dg.T0[qb - 1] && ca != 'N' && ca != '$' && !(ca >= 'a' && ca <= 'z')) return;
    }
    const u64 key = cas_key(qa, (int64_t)b1);
    dg.key[i] = key;
    if (key > dg.best[c]) atomicMax((unsigned long long *)&dg.best[c], (unsigned long long)key);
}
__global__ __launch_bounds__(TB) void k_cas_dwrite(const sa_t *__restrict__ w_pos, const u32 *__restrict__ w_child, u32 NW, CasDanger dg) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= NW) return;
    const u32 c = w_child[i];
    if (c == NONE || !dg.dflag[c]) return;
    const u64 key = dg.key[i];
    if (key == 0) return;
    if (key == dg.best[c]) dg.qb[c] = w_pos[dg.p1[i]];
}

__global__ __launch_bounds__(TB) void k_cas_decide(CasIv *__restrict__ iv, u64 *__restrict__ best, u32 *__restrict__ wmax, int32_t *__restrict__ depth,
                                                   CasRes *__restrict__ res, u32 minl, u32 *__restrict__ counters, u32 child_cap,
                                                   u32 *__restrict__ und_list, u32 leaf_n, RvCascadeIO io, CasDanger dg, int danger) {
    __shared__ u32 s_cnt[TB / 64][3];      // per wave: children, anchors, undecided entries
    __shared__ u32 s_base[3];
    const u32 lo = counters[C_LO], hi = counters[C_HI];      // the level's sub-indices (k_cas_advance sets the next level's)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (u32 first = lo + blockIdx.x * TB; first < hi; first += gridDim.x * TB) {      // (uniform per workgroup: barriers inside)
        const u32 id = first + threadIdx.x;
        const bool in = id < hi;
        CasIv p; p.a0 = p.a1 = p.b0 = p.b1 = 0;
        u64 bk = 0; u32 wm = 0; int32_t dp = 0;
        CasRes r; r.qa = r.qb = 0; r.ql = 0; r.lead = r.trail = NONE; r.state = 2;
        if (in) { p = iv[id]; bk = best[id]; wm = wmax[id]; dp = depth[id]; r = res[id]; }
        const int64_t la = (int64_t)p.a1 - p.a0, lb = (int64_t)p.b1 - p.b0;
        const u32 bl = (u32)(bk >> KEY_SHIFT);
        // both samples present and room for a match of minl characters in each (otherwise the reference's scan of this sub-index finds nothing)
        const bool can = in & (la >= (int64_t)minl) & (lb >= (int64_t)minl);
        bool split = can & (bl >= minl) & (bl > wm);
        bool und = can & !split & (wm >= minl);
        bool forced = false, retry = false;
        if (danger && und && (u64)(la + lb) > (u64)dg.leaf_n) {
            // the second attempt: what the witnesses of this sub-index say (k_cas_dwalk / k_cas_dpick)
            const u32 fl = dg.flag[id];
            const u64 kh = dg.best[id], kc = (bl >= minl && !(fl & 1u)) ? bk : 0ull;
            if (!(fl & 2u)) {
                if (kh == 0 && kc == 0) {
                    if (bl < minl) und = false;          // no cut match and no pair among the witnesses: nothing to find here
                    else if ((fl >> 8) < 16u) { retry = true; und = false; }      // the cut match is not one of this sub-index and nothing is as long: the next one's turn
                } else {
                    if (kh > kc) { r.qa = (sa_t)(KEY_LOW - (kh & KEY_LOW)); r.qb = dg.qb[id]; r.ql = (u32)(kh >> KEY_SHIFT); }
                    split = true; forced = true; und = false;
                }
            }
            if (!retry) atomicAdd(&counters[und ? C_NUNSOLVED : C_NSOLVED], 1u);
        }
        bool lead = retry, trail = false;
        if (split) {
            if (!forced && r.ql != bl) atomicOr(&counters[C_ERR], 2u);      // (the winner of the bid did not report: cannot happen)
            lead = ((int64_t)r.qa - p.a0) + ((int64_t)r.qb - p.b0) > 0;
            trail = ((int64_t)p.a1 - r.qa - r.ql) + ((int64_t)p.b1 - r.qb - r.ql) > 0;
        }
        // room for the children, the anchor and the undecided entry: one reservation per workgroup each (one per wave was 16 000
        // returning atomics on one address at the widest levels of 2 x 250 Mbp -- 6.7 of the cascade's 12 ms)
        const u64 b_lead = __ballot(lead), b_trail = __ballot(trail), b_split = __ballot(split), b_und = __ballot(und);
        if (retry) atomicAdd(&counters[C_NRETRY], 1u);
        if (lane == 0) { s_cnt[w][0] = (u32)__popcll(b_lead) + (u32)__popcll(b_trail); s_cnt[w][1] = (u32)__popcll(b_split); s_cnt[w][2] = (u32)__popcll(b_und); }
        __syncthreads();
        if (threadIdx.x < 3) {
            u32 tot = 0;
            for (int k = 0; k < TB / 64; k++) tot += s_cnt[k][threadIdx.x];
            u32 *ctr = threadIdx.x == 0 ? &counters[C_NCHILD] : threadIdx.x == 1 ? io.anchor_count : &counters[C_NUND];
            s_base[threadIdx.x] = tot ? atomicAdd(ctr, tot) : 0u;
        }
        __syncthreads();
        u32 base_c = s_base[0], base_a = s_base[1], base_u = s_base[2];
        for (int k = 0; k < w; k++) { base_c += s_cnt[k][0]; base_a += s_cnt[k][1]; base_u += s_cnt[k][2]; }
        if (retry) {
            // the sub-index again, as its own only child, with the bids capped below the match that failed (not a level deeper, not visited twice)
            const u32 slot = base_c + (u32)__popcll(b_lead & lt) + (u32)__popcll(b_trail & lt);
            CasRes nr; nr.qa = 0; nr.qb = 0; nr.ql = 0; nr.lead = NONE; nr.trail = NONE; nr.state = 0;
            if (slot < child_cap) {
                iv[slot] = p; best[slot] = 0; wmax[slot] = 0; depth[slot] = dp; res[slot] = nr;
                dg.best[slot] = 0; dg.flag[slot] = ((dg.flag[id] >> 8) + 1u) << 8; dg.ceil[slot] = bk;
                r.lead = slot; r.trail = NONE;
            } else atomicOr(&counters[C_ERR], 1u);
            r.state = 3;
        } else if (split) {
            u32 slot = base_c + (u32)__popcll(b_lead & lt) + (u32)__popcll(b_trail & lt);
            CasRes nr; nr.qa = 0; nr.qb = 0; nr.ql = 0; nr.lead = NONE; nr.trail = NONE; nr.state = 0;
            if (lead) {
                if (slot < child_cap) {
                    CasIv c; c.a0 = p.a0; c.a1 = r.qa; c.b0 = p.b0; c.b1 = r.qb;
                    iv[slot] = c; best[slot] = 0; wmax[slot] = 0; depth[slot] = dp + 1; res[slot] = nr;
                    if (danger) { dg.best[slot] = 0; dg.flag[slot] = 0; dg.ceil[slot] = 0; }
                    r.lead = slot;
                } else atomicOr(&counters[C_ERR], 1u);
                slot++;
            }
            if (trail) {
                if (slot < child_cap) {
                    CasIv c; c.a0 = (sa_t)((int64_t)r.qa + r.ql); c.a1 = p.a1; c.b0 = (sa_t)((int64_t)r.qb + r.ql); c.b1 = p.b1;
                    iv[slot] = c; best[slot] = 0; wmax[slot] = 0; depth[slot] = dp + 1; res[slot] = nr;
                    if (danger) { dg.best[slot] = 0; dg.flag[slot] = 0; dg.ceil[slot] = 0; }
                    r.trail = slot;
                } else atomicOr(&counters[C_ERR], 1u);
            }
            const u32 as = base_a + (u32)__popcll(b_split & lt);
            if (as < io.anchor_cap) { io.anchor_l[as] = r.ql; io.anchor_pos[2 * (size_t)as] = (int64_t)r.qa; io.anchor_pos[2 * (size_t)as + 1] = (int64_t)r.qb; }
            else atomicOr(&counters[C_ERR], 4u);
            r.state = 1;
        } else {
            r.state = 2;
        }
        if (in) res[id] = r;
        if (und) {
            und_list[base_u + (u32)__popcll(b_und & lt)] = id;
            if ((u64)(la + lb) > (u64)leaf_n) atomicMax(&counters[C_MAXN], (u32)((la + lb) > 0xFFFFFFFFll ? 0xFFFFFFFFll : (la + lb)));
        }
        __syncthreads();      // (s_cnt / s_base are rewritten by the next stretch)
    }
    // The next level's sub-indices are the ones this level has made: the workgroup that finishes last moves the range on.  (Every workgroup read
    // the range when it started, and takes its ticket when it is done: nobody reads the range after the last ticket.  A kernel of its own for
    // this was 4.5 us per level, 53 levels at 2 x 250 Mbp.)
    if (threadIdx.x == 0) {      // (what the workgroups wrote is for the next kernel; the fence orders this workgroup's C_NCHILD reservations -- atomics of other
        __threadfence();         //  waves, behind the barrier above -- in front of its ticket, so the last ticket holder reads the final count)
        if (atomicAdd(&counters[C_TICKET], 1u) == gridDim.x - 1) {
            if (hi > lo) counters[C_LEVELS]++;
            counters[C_LO] = hi; counters[C_HI] = atomicAdd(&counters[C_NCHILD], 0u);
            counters[C_TICKET] = 0;
        }
    }
}
// the run's counters (rv_leaf.h: [0] sub-indices visited, [1] anchors, [2] anchored bp, [3] largest depth) from what the levels left:
// every sub-index made has been visited, except the undecided ones (the leaf kernel counts those itself)
__global__ __launch_bounds__(TB) void k_cas_stats(const u32 *__restrict__ counters, const int32_t *__restrict__ depth, RvCascadeIO io) {
    __shared__ unsigned long long s_bp[TB / 64];
    __shared__ u32 s_md[TB / 64];
    const u32 nchild = counters[C_NCHILD], na = *io.anchor_count;
    unsigned long long bp = 0; u32 md = 0;
    for (u32 i = blockIdx.x * TB + threadIdx.x; i < na; i += gridDim.x * TB) bp += io.anchor_l[i];
    for (u32 i = blockIdx.x * TB + threadIdx.x; i < nchild; i += gridDim.x * TB) { const u32 d = (u32)depth[i]; md = d > md ? d : md; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        bp += ((unsigned long long)(u32)__shfl_down((u32)(bp >> 32), d, 64) << 32) + (unsigned long long)(u32)__shfl_down((u32)bp, d, 64);
        const u32 om = __shfl_down(md, d, 64);
        md = om > md ? om : md;
    }
    if ((threadIdx.x & 63) == 0) { s_bp[threadIdx.x >> 6] = bp; s_md[threadIdx.x >> 6] = md; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < TB / 64; k++) { bp += s_bp[k]; md = s_md[k] > md ? s_md[k] : md; }
        if (bp) atomicAdd(&io.stats[2], bp);
        if (md) atomicMax(&io.stats[3], (unsigned long long)md);
        if (blockIdx.x == 0) { atomicAdd(&io.stats[0], (unsigned long long)(nchild - counters[C_NUND] - counters[C_NRETRY])); atomicAdd(&io.stats[1], (unsigned long long)na); }
    }
}

// ---- undecided sub-indices: rebuilt from their text ----
__global__ __launch_bounds__(TB) void k_cas_sizes(const u32 *__restrict__ und_list, u32 U, const CasIv *__restrict__ iv, u64 *__restrict__ sizes) {
    const u32 u = blockIdx.x * TB + threadIdx.x;
    if (u < U) { const CasIv p = iv[und_list[u]]; sizes[u] = (u64)(((int64_t)p.a1 - p.a0) + ((int64_t)p.b1 - p.b0)); }
    if (u == U) sizes[u] = 0;
}
__global__ __launch_bounds__(TB) void k_cas_roots(const u32 *__restrict__ und_list, u32 U, const CasIv *__restrict__ iv, const int32_t *__restrict__ depth,
                                                  const u64 *__restrict__ offs, RvLeafRoot *__restrict__ roots) {
    const u32 u = blockIdx.x * TB + threadIdx.x;
    if (u >= U) return;
    const u32 id = und_list[u];
    const CasIv p = iv[id];
    RvLeafRoot r;
    r.off = (int64_t)offs[u]; r.n = (int32_t)(((int64_t)p.a1 - p.a0) + ((int64_t)p.b1 - p.b0)); r.depth = depth[id];
    r.a0 = p.a0; r.a1 = p.a1; r.b0 = p.b0; r.b1 = p.b1;
    roots[u] = r;
}

// SA / LCP / BWT of one sub-index from the text of its two intervals (at most RV_LEAF_N suffixes): every
// suffix counts the suffixes in front of it -- byte order of the text, a suffix ending at its interval's end in front of every
// longer one that starts with it, which is where bubble_sort (reveal.c:666-727) puts the suffixes it cuts --, then its common
// prefix with its predecessor with the stops of interface.c:97-114.
constexpr int BN = RV_LEAF_N;
// first x < lim with txt[i + x] != txt[j + x], or lim: eight bytes per step (a repeat's copies agree for hundreds of characters)
__device__ inline int cas_first_diff(const uint8_t *txt, int i, int j, int lim) {
    int x = 0;
    while (x + 8 <= lim) {
        u64 a, b;
        __builtin_memcpy(&a, txt + i + x, 8);
        __builtin_memcpy(&b, txt + j + x, 8);
        if (a != b) return x + (__builtin_ctzll(a ^ b) >> 3);
        x += 8;
    }
    while (x < lim && txt[i + x] == txt[j + x]) x++;
    return x;
}
// (a workgroup takes 256 suffixes of a sub-index: as one workgroup per sub-index, byte by byte, 41 sub-indices of about a thousand suffixes
// inside copies of a 1.5 kb repeat took 40 ms)
__global__ __launch_bounds__(TB) void k_cas_rank(const RvLeafRoot *__restrict__ roots, const uint8_t *__restrict__ T0, uint16_t *__restrict__ ord) {
    __shared__ __attribute__((aligned(8))) uint8_t txt[BN + 24];
    const RvLeafRoot root = roots[blockIdx.x];
    const int la = (int)(root.a1 - root.a0), lb = (int)(root.b1 - root.b0), n = la + lb;
    if ((int)blockIdx.y * TB >= n) return;
    for (int k = threadIdx.x; k < n; k += TB) txt[k] = k < la ? T0[root.a0 + k] : T0[root.b0 + (k - la)];
    __syncthreads();
    const int i = (int)blockIdx.y * TB + threadIdx.x;
    if (i >= n) return;
    const int ri = (i < la ? la : n) - i;
    int cnt = 0;
    // (the first eight bytes from registers: mine once, the other's as a window that moves a byte per step -- rv_cascade_multi.hip k_casm_rank)
    u64 ki = 0, wj = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) { ki |= (u64)txt[i + b] << (8 * b); wj |= (u64)txt[b] << (8 * b); }
    for (int j0 = 0; j0 < n; j0 += 8) {      // (the bytes that enter the window come eight at a time from one aligned load: no step waits for LDS)
        u64 nxt = *reinterpret_cast<const u64 *>(txt + j0 + 8);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int j = j0 + e;
            if (j < n) {
                const int rj = (j < la ? la : n) - j;
                const int lim = ri < rj ? ri : rj;
                const u64 d = ki ^ wj;
                int k = d ? (__builtin_ctzll(d) >> 3) : 8;
                bool j_less;
                if (k < 8 && k < lim) j_less = (u32)((wj >> (8 * k)) & 0xffu) < (u32)((ki >> (8 * k)) & 0xffu);
                else {
                    if (k >= 8 && lim > 8 && j != i) k = 8 + cas_first_diff(txt, i + 8, j + 8, lim - 8);
                    else if (j == i) k = lim;
                    j_less = (k < lim) ? (txt[j + k] < txt[i + k]) : ((rj < ri) | ((rj == ri) & (j < i)));
                }
                cnt += j_less ? 1 : 0;
            }
            wj = (wj >> 8) | (nxt << 56);
            nxt >>= 8;
        }
    }
    ord[root.off + cnt] = (uint16_t)i;
}
__global__ __launch_bounds__(TB) void k_cas_emit(const RvLeafRoot *__restrict__ roots, const uint8_t *__restrict__ T0, const uint16_t *__restrict__ ord, sa_t *__restrict__ SA,
                                                 lcp_t *__restrict__ LCP, uint8_t *__restrict__ BWT, int64_t nsep0, int64_t root_a0, int64_t root_b0,
                                                 const sa_t *__restrict__ beginA, u32 nA, const sa_t *__restrict__ beginB, u32 nB) {
    __shared__ uint8_t txt[BN + 8];
    const RvLeafRoot root = roots[blockIdx.x];
    const int la = (int)(root.a1 - root.a0), lb = (int)(root.b1 - root.b0), n = la + lb;
    if ((int)blockIdx.y * TB >= n) return;
    for (int k = threadIdx.x; k < n; k += TB) txt[k] = k < la ? T0[root.a0 + k] : T0[root.b0 + (k - la)];
    __syncthreads();
    const int r = (int)blockIdx.y * TB + threadIdx.x;
    if (r >= n) return;
    const int i = ord[root.off + r];
    const int ri = (i < la ? la : n) - i;
    u32 l = 0;
    if (r > 0) {
        const int j = ord[root.off + r - 1];
        const int rj = (j < la ? la : n) - j;
        const int lim = ri < rj ? ri : rj;
        int k = 0;
        while (k < lim) { const uint8_t c = txt[i + k]; if (c != txt[j + k] || c == '$' || c == 'N') break; k++; }
        l = (u32)k;
    }
    const int64_t gp = i < la ? root.a0 + i : root.b0 + (i - la);
    uint8_t ch = gp > 0 ? T0[gp - 1] : (uint8_t)'$';
    // the first suffix of an interval that starts behind an anchor: that anchor's last base has been lower-cased (reveal.c:1230-1234)
    // (several sequences per sample: an interval that does not start where its sequence starts begins behind an anchor)
    bool at_seq_start;
    if (beginA) {
        const int c = i < la ? cas_contig_of(beginA, nA, (int64_t)root.a0) : cas_contig_of(beginB, nB, (int64_t)root.b0);
        at_seq_start = i < la ? (c >= 0 && (int64_t)beginA[c] == (int64_t)root.a0) : (c >= 0 && (int64_t)beginB[c] == (int64_t)root.b0);
    } else at_seq_start = i < la ? !(root.a0 > root_a0) : !(root.b0 > root_b0);
    const bool behind_anchor = (i < la ? i == 0 : i == la) && !at_seq_start;
    if (behind_anchor && ch >= 'A' && ch <= 'Z') ch += 32;
    const int64_t o = root.off + r;
    SA[o] = (sa_t)gp; LCP[o] = (lcp_t)l; BWT[o] = (uint8_t)(ch | (gp > nsep0 ? RV_BWT_SIDE : 0u));
}

int bitlen64(u64 x) { int b = 0; while (x) { b++; x >>= 1; } return b; }
double cas_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// the chain of rest sub-indices, decided on the host (see "the lineage of rest sub-indices" above).  -> 0 and *why == nullptr: roots,
// tables and the chain's anchors are in cb.d[30..33]; *why != nullptr: the chain holds a member the list does not decide
static int cas_lineage(rv_index *h, RvCascadeBufs &cb, u32 M, u32 NW, u32 minl, const std::vector<RvIntv> &CA, const std::vector<RvIntv> &CB, const char **why) {
    Workspace &ws = h->ws;
    hipStream_t q = ws.stream;
    *why = nullptr;
    DBuf &k0 = cb.d[0], &k1 = cb.d[1], &v0 = cb.d[2], &v1 = cb.d[3], &bpa = cb.d[4], &bpb = cb.d[5], &blen = cb.d[6], &bwp = cb.d[8], &bwv = cb.d[9];
    DBuf &spa = cb.d[27], &spb = cb.d[28], &slen = cb.d[29], &broots = cb.d[30], &bdepth = cb.d[31], &btabs = cb.d[32], &banch = cb.d[33];
    // ---- the matches in the picker's order, and the witnesses, to the host
    RV_TRY(spa.reserve((size_t)M * sizeof(sa_t))); RV_TRY(spb.reserve((size_t)M * sizeof(sa_t))); RV_TRY(slen.reserve((size_t)M * 4));
    const unsigned mb = (unsigned)ceil_div((int64_t)M, TB);
    hipLaunchKernelGGL(k_cas_lkeys, dim3(mb), dim3(TB), 0, q, (const sa_t *)bpa.as<sa_t>(), (const u32 *)blen.as<u32>(), M, k0.as<u64>(), v0.as<u32>());
    RV_LAUNCH_CHECK();
    int in1 = 0;
    RV_TRY(rv_radix_sort_pairs<u32>(ws, k0.as<u64>(), v0.as<u32>(), k1.as<u64>(), v1.as<u32>(), (int64_t)M, 0, 64, &in1));
    hipLaunchKernelGGL(k_cas_lgather, dim3(mb), dim3(TB), 0, q, (const sa_t *)bpa.as<sa_t>(), (const sa_t *)bpb.as<sa_t>(), (const u32 *)blen.as<u32>(),
                       (const u32 *)(in1 ? v1.as<u32>() : v0.as<u32>()), M, spa.as<sa_t>(), spb.as<sa_t>(), slen.as<u32>());
    RV_LAUNCH_CHECK();
    std::vector<sa_t> ha(M), hb(M), hwp(NW);
    std::vector<u32> hl(M), hwv(NW);
    RV_HIP(hipMemcpyAsync(ha.data(), spa.p, (size_t)M * sizeof(sa_t), hipMemcpyDeviceToHost, q));
    RV_HIP(hipMemcpyAsync(hb.data(), spb.p, (size_t)M * sizeof(sa_t), hipMemcpyDeviceToHost, q));
    RV_HIP(hipMemcpyAsync(hl.data(), slen.p, (size_t)M * 4, hipMemcpyDeviceToHost, q));
    if (NW) {
        RV_HIP(hipMemcpyAsync(hwp.data(), bwp.p, (size_t)NW * sizeof(sa_t), hipMemcpyDeviceToHost, q));
        RV_HIP(hipMemcpyAsync(hwv.data(), bwv.p, (size_t)NW * 4, hipMemcpyDeviceToHost, q));
    }
    RV_HIP(hipStreamSynchronize(q));
    const size_t nA = CA.size(), nB = CB.size();
    auto contig = [](const std::vector<RvIntv> &C, int64_t pos) -> int {
        size_t lo = 0, hi = C.size();
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (C[mid].begin <= pos) lo = mid + 1; else hi = mid; }
        return (int)lo - 1;
    };
    // W over every sequence; the chain's bound = the largest among the sequences still in it
    std::vector<u32> wA(nA, 0), wB(nB, 0);
    const int64_t firstB = CB[0].begin;
    for (u32 i = 0; i < NW; i++) {
        const int64_t pos = (int64_t)hwp[i];
        if (pos < firstB) { const int c = contig(CA, pos); if (c >= 0 && pos < CA[(size_t)c].end) wA[(size_t)c] = std::max(wA[(size_t)c], hwv[i]); }
        else { const int c = contig(CB, pos); if (c >= 0 && pos < CB[(size_t)c].end) wB[(size_t)c] = std::max(wB[(size_t)c], hwv[i]); }
    }
    struct Ent { u32 w; int side; u32 c; };
    std::vector<Ent> byw;
    for (size_t c = 0; c < nA; c++) byw.push_back({wA[c], 0, (u32)c});
    for (size_t c = 0; c < nB; c++) byw.push_back({wB[c], 1, (u32)c});
    std::sort(byw.begin(), byw.end(), [](const Ent &x, const Ent &y) { return x.w > y.w; });
    std::vector<uint8_t> liveA(nA, 1), liveB(nB, 1);
    size_t wptr = 0, nLiveA = nA, nLiveB = nB;
    auto chain_wmax = [&]() -> u32 {
        while (wptr < byw.size() && !(byw[wptr].side ? liveB[byw[wptr].c] : liveA[byw[wptr].c])) wptr++;
        return wptr < byw.size() ? byw[wptr].w : 0u;
    };
    std::vector<CasIv> roots; std::vector<int32_t> rdepth;
    bool stalled = false;
    std::vector<int32_t> pickA(nA, -1), pickB(nB, -1);
    std::vector<u32> rootLead, rootTrail, anl; std::vector<int64_t> anp;
    int32_t depth = 0;
    for (u32 i = 0; i < M && nLiveA && nLiveB; i++) {
        const u32 l = hl[i];
        if (l < minl) break;
        const int64_t a = (int64_t)ha[i], b = (int64_t)hb[i];
        const int ca = contig(CA, a), cb2 = contig(CB, b);
        if (ca < 0 || cb2 < 0 || !liveA[(size_t)ca] || !liveB[(size_t)cb2]) continue;
        if (a + (int64_t)l > CA[(size_t)ca].end || b + (int64_t)l > CB[(size_t)cb2].end) { *why = "a match that leaves its sequence"; return 0; }
        if (l <= chain_wmax()) { stalled = true; break; }      // a repeat as long as the best match between the sequences that are left
        const u32 k = (u32)anl.size();
        if (ws.opt.cascade_log) fprintf(stderr, "cascade: chain member %u chooses %u at %lld / %lld (sequences %d / %d), bound %u\n", k, l, (long long)a, (long long)b, ca, cb2, chain_wmax());
        anl.push_back(l); anp.push_back(a); anp.push_back(b);
        pickA[(size_t)ca] = (int32_t)k; pickB[(size_t)cb2] = (int32_t)k;
        u32 rl = NONE, rt = NONE;
        if ((a - CA[(size_t)ca].begin) + (b - CB[(size_t)cb2].begin) > 0) {
            CasIv c; c.a0 = (sa_t)CA[(size_t)ca].begin; c.a1 = (sa_t)a; c.b0 = (sa_t)CB[(size_t)cb2].begin; c.b1 = (sa_t)b;
            rl = (u32)roots.size(); roots.push_back(c); rdepth.push_back(depth + 1);
        }
        if ((CA[(size_t)ca].end - a - (int64_t)l) + (CB[(size_t)cb2].end - b - (int64_t)l) > 0) {
            CasIv c; c.a0 = (sa_t)(a + (int64_t)l); c.a1 = (sa_t)CA[(size_t)ca].end; c.b0 = (sa_t)(b + (int64_t)l); c.b1 = (sa_t)CB[(size_t)cb2].end;
            rt = (u32)roots.size(); roots.push_back(c); rdepth.push_back(depth + 1);
        }
        rootLead.push_back(rl); rootTrail.push_back(rt);
        liveA[(size_t)ca] = 0; liveB[(size_t)cb2] = 0; nLiveA--; nLiveB--;
        depth++;
    }
    if (anl.empty()) { *why = stalled ? "a repeat as long as the best match of the root" : "no match between two sequences at the top level"; return 0; }
    // the last member of the chain (what no choice touched): visited once more if anything is left; with both samples in it, it must provably hold no
    // match.  A member the list does not decide (a repeat as long as its best match; no match but repeats of minl characters) is not visited here:
    // its sequences go back to the caller, who makes it the level pipeline's frontier (rv_align.hip builtin_cascade)
    int64_t steps = (int64_t)anl.size();
    int maxdepth = depth - 1;
    cb.lin_rest.clear(); cb.lin_rest_depth = depth;
    if (ws.opt.cascade_log) fprintf(stderr, "cascade: chain of %zu choices, %zu + %zu sequences left, stalled %d, bound %u\n", anl.size(), nLiveA, nLiveB, (int)stalled, chain_wmax());
    if (nLiveA + nLiveB > 0) {
        if (stalled || (nLiveA && nLiveB && chain_wmax() >= minl)) {
            for (size_t c = 0; c < nA; c++) if (liveA[c]) { cb.lin_rest.push_back(CA[c].begin); cb.lin_rest.push_back(CA[c].end); }
            for (size_t c = 0; c < nB; c++) if (liveB[c]) { cb.lin_rest.push_back(CB[c].begin); cb.lin_rest.push_back(CB[c].end); }
        } else { steps++; maxdepth = depth; }
    }
    // ---- to the device: the roots, the look-up tables of k_cas_assign_roots, the chain's anchors
    const u32 R = (u32)roots.size(), P = (u32)anl.size();
    RV_TRY(broots.reserve((size_t)std::max<u32>(R, 1) * sizeof(CasIv))); RV_TRY(bdepth.reserve((size_t)std::max<u32>(R, 1) * 4));
    const size_t tab_bytes = (nA + nB) * (sizeof(sa_t) + 4) + (size_t)P * 8 + 64;
    RV_TRY(btabs.reserve(tab_bytes + 64)); RV_TRY(banch.reserve((size_t)P * (4 + 16) + 64));
    std::vector<uint8_t> stage(tab_bytes + 64, 0);
    size_t o = 0;
    auto put = [&](const void *src, size_t bytes) { memcpy(stage.data() + o, src, bytes); const size_t at = o; o += (bytes + 15) / 16 * 16; return at; };
    std::vector<sa_t> bA(nA), bB(nB);
    for (size_t c = 0; c < nA; c++) bA[c] = (sa_t)CA[c].begin;
    for (size_t c = 0; c < nB; c++) bB[c] = (sa_t)CB[c].begin;
    stage.resize((nA + nB) * (sizeof(sa_t) + 4) + (size_t)P * 8 + 16 * 8 + 64);
    const size_t oA = put(bA.data(), nA * sizeof(sa_t)), oB = put(bB.data(), nB * sizeof(sa_t)), opA = put(pickA.data(), nA * 4), opB = put(pickB.data(), nB * 4),
                 oL = put(rootLead.data(), (size_t)P * 4), oT = put(rootTrail.data(), (size_t)P * 4);
    RV_TRY(btabs.reserve(o + 64));
    RV_HIP(hipMemcpyAsync(btabs.p, stage.data(), o, hipMemcpyHostToDevice, q));
    if (R) {
        RV_HIP(hipMemcpyAsync(broots.p, roots.data(), (size_t)R * sizeof(CasIv), hipMemcpyHostToDevice, q));
        RV_HIP(hipMemcpyAsync(bdepth.p, rdepth.data(), (size_t)R * 4, hipMemcpyHostToDevice, q));
    }
    RV_HIP(hipMemcpyAsync(banch.p, anp.data(), (size_t)P * 16, hipMemcpyHostToDevice, q));
    RV_HIP(hipMemcpyAsync((uint8_t *)banch.p + (size_t)P * 16, anl.data(), (size_t)P * 4, hipMemcpyHostToDevice, q));
    RV_HIP(hipStreamSynchronize(q));      // (the vectors above live on this stack frame)
    cb.lin_roots = R; cb.lin_picks = P; cb.lin_nA = (u32)nA; cb.lin_nB = (u32)nB; cb.lin_steps = steps; cb.lin_maxdepth = maxdepth;
    cb.lin_off[0] = oA; cb.lin_off[1] = oB; cb.lin_off[2] = opA; cb.lin_off[3] = opB; cb.lin_off[4] = oL; cb.lin_off[5] = oT;
    return 0;
}

int rv_cascade_run(rv_index *h, RvCascadeBufs &cb, const RvCascadeIO &io, int minl_in, RvCascadeOut *out, int danger, int reuse) {
    memset(out, 0, sizeof *out);
    out->done = false;
    Workspace &ws = h->ws;
    hipStream_t q = ws.stream;
    const int64_t n = h->n;
    const u32 minl = (u32)std::max(minl_in, 1);
    const bool verbose = (ws.opt.cascade_log != 0);
    double tp[6] = {0, 0, 0, 0, 0, 0};
    if (verbose) { (void)hipStreamSynchronize(q); tp[0] = cas_now(); }
#define GIVE_UP(msg) do { out->why = msg; if (verbose) fprintf(stderr, "cascade: gave up: %s\n", msg); return 0; } while (0)
    if (h->nsamples != 2 || h->nsep.size() != 1 || h->nodes.size() < 2) GIVE_UP("not two samples");
    // the sequences of the two samples (empty ones hold no suffix)
    std::vector<RvIntv> CA, CB;
    for (const RvIntv &v : h->nodes) if (v.end > v.begin) (v.begin < h->nsep[0] ? CA : CB).push_back(v);
    std::sort(CA.begin(), CA.end(), [](const RvIntv &x, const RvIntv &y) { return x.begin < y.begin; });
    std::sort(CB.begin(), CB.end(), [](const RvIntv &x, const RvIntv &y) { return x.begin < y.begin; });
    if (CA.empty() || CB.empty()) GIVE_UP("an empty sample");
    const bool chain = CA.size() > 1 || CB.size() > 1;      // several sequences in a sample: the lineage of rest sub-indices first
    if (chain && ws.opt.no_cascade_chain) GIVE_UP("several sequences per sample (switched off)");
    if (n >= ((int64_t)1 << 32) - 2) GIVE_UP("index above 2^32 positions");
#ifdef RV_SA64
    if ((u64)h->maxlcp >= (1ull << 24) || n >= ((int64_t)1 << 40)) GIVE_UP("bid word too narrow");
#endif
    CasIv root; root.a0 = root.a1 = root.b0 = root.b1 = 0;
    root.a0 = (sa_t)CA[0].begin; root.a1 = (sa_t)CA[0].end; root.b0 = (sa_t)CB[0].begin; root.b1 = (sa_t)CB[0].end;
    if (root.a0 >= root.a1 || root.b0 >= root.b1 || (int64_t)root.a1 > h->nsep[0] || (int64_t)root.b0 <= h->nsep[0]) GIVE_UP("an empty sample");

    const sa_t *SA = h->dSA.as<sa_t>(); const lcp_t *LCP = h->dLCP.as<lcp_t>(); const uint8_t *BWT = h->dBWT.as<uint8_t>();

    // ---- the root's matches, packed on the device (the scan of the level pipeline's first level: same kernel, same profile slot)
    const int64_t ntile = ceil_div(n, RV_PAIR_TILE);
    DBuf &bcnt = ws.misc[14], &btab = ws.misc[2], &bslot = ws.misc[3], &bovf = ws.misc[4], &bout = ws.misc[5];
    if (bcnt.cap == 0) { RV_TRY(bcnt.reserve(64)); RV_HIP(hipMemsetAsync(bcnt.p, 0, 64, q)); }
    RV_TRY(btab.reserve((size_t)(ntile + 1) * 3 * sizeof(u32)));
    RV_TRY(bslot.reserve((size_t)ntile * RV_PAIR_SLOTS * sizeof(RvPairRec)));
    if (bovf.cap < 4096 * sizeof(RvPairRec)) RV_TRY(bovf.reserve(4096 * sizeof(RvPairRec)));
    if (bout.cap < 4096 * sizeof(RvPairRec)) RV_TRY(bout.reserve(sizeof(RvPairRec) * (size_t)std::max<int64_t>(4096, n / 64)));
    u32 *tilecnt = btab.as<u32>(), *tileovf = tilecnt + (ntile + 1), *tileoff = tileovf + (ntile + 1);
    u32 M = reuse ? cb.M : 0;
    for (int attempt = 0; !reuse; attempt++) {
        if (attempt == 3) { rv_set_error("cascade: scan buffer sizing failed"); return -1; }
        const size_t ocap = bout.cap / sizeof(RvPairRec) - RV_PAIR_HDR, vcap = bovf.cap / sizeof(RvPairRec);
        hipEvent_t ev_a, ev_b;
        (void)h->prof.attach(RV_K_SCAN_PAIR, (double)n * 8.0, &ev_a, &ev_b);      // 8 B per rank (a 4-byte suffix + a 4-byte LCP value), also for the 64-bit library -- the kernel reads suffixes only where a match may start
        RV_TRY(rv_scan_pair_launch(ws, SA, LCP, n, BWT, (sa_t)h->nsep[0], (int)minl, bslot.as<RvPairRec>(), bovf.as<RvPairRec>(),
                                   (u32)std::min<size_t>(vcap, 0xffffffffu), bcnt.as<u32>(), tilecnt, tileovf, nullptr, nullptr, 0, ev_a, ev_b));
        RV_TRY(rv_exclusive_sum_u32(ws, tilecnt, tileoff, ntile + 1));
        RV_TRY(rv_pair_compact_launch(ws, bslot.as<RvPairRec>(), bovf.as<RvPairRec>(), tilecnt, tileovf, tileoff, ntile, bout.as<RvPairRec>(),
                                      (u32)std::min<size_t>(ocap, 0xffffffffu), bcnt.as<u32>(), nullptr, (u32)std::min<size_t>(vcap, 0xffffffffu)));
        u32 hdr[4];
        RV_TRY(rv_read_back(ws, hdr, bout.p, sizeof hdr));
        const u32 total = hdr[0], novf = hdr[1];
        if (total <= ocap && novf <= vcap) { M = total; break; }
        if (novf > vcap) RV_TRY(bovf.reserve((size_t)novf * sizeof(RvPairRec)));
        if (total > ocap) RV_TRY(bout.reserve(((size_t)total + RV_PAIR_HDR) * sizeof(RvPairRec)));
    }
    if (verbose) tp[1] = cas_now();
    out->cands = M;
    struct ProfSpan { Workspace &w; int id; ~ProfSpan() { w.prof_end(id); } } span{ws, ws.prof_begin(RV_K_CASCADE, 5.0 * (double)n)};      // (bytes: the witness pass over LCP + BWT)
    if (M == 0) GIVE_UP("no match at the top level");
    const RvPairRec *recs = bout.as<RvPairRec>() + RV_PAIR_HDR;
    cb.M = M;

    // ---- buffers
    // a region of the witness list takes every WT_REGIONS-th tile: n / WT_REGIONS + a tile's worth of entries can never overflow (small inputs
    // get that: a low-complexity stretch puts all its witnesses into two or three regions); large inputs a 1024th of n per region
    const int64_t rcap64 = std::min<int64_t>(n / WT_REGIONS + WT_TILE, std::max<int64_t>(16384, n / 1024));
    const u32 wcap = (u32)std::min<int64_t>(rcap64 * WT_REGIONS, 0x7fffffff);
    const int64_t ccap64 = n / (int64_t)minl + 16;      // every anchor covers 2 * minl positions and makes two sub-indices at most
    if (ccap64 >= 0x7fffffff) GIVE_UP("too many sub-indices possible");
    const u32 ccap = (u32)ccap64;
    DBuf &k0 = cb.d[0], &k1 = cb.d[1], &v0 = cb.d[2], &v1 = cb.d[3], &bpa = cb.d[4], &bpb = cb.d[5], &blen = cb.d[6], &bcc = cb.d[7], &bwp = cb.d[8], &bwv = cb.d[9],
         &bwc = cb.d[10], &biv = cb.d[11], &bbest = cb.d[12], &bwm = cb.d[13], &bdep = cb.d[14], &bres = cb.d[15], &bctr = cb.d[16], &bund = cb.d[17], &bsz = cb.d[18];
    RV_TRY(k0.reserve((size_t)M * 8)); RV_TRY(k1.reserve((size_t)M * 8)); RV_TRY(v0.reserve((size_t)M * 4)); RV_TRY(v1.reserve((size_t)M * 4));
    RV_TRY(bpa.reserve((size_t)M * sizeof(sa_t))); RV_TRY(bpb.reserve((size_t)M * sizeof(sa_t))); RV_TRY(blen.reserve((size_t)M * 4)); RV_TRY(bcc.reserve((size_t)M * 4));
    RV_TRY(bwp.reserve((size_t)wcap * sizeof(sa_t))); RV_TRY(bwv.reserve((size_t)wcap * 4)); RV_TRY(bwc.reserve((size_t)wcap * 4));
    RV_TRY(biv.reserve((size_t)ccap * sizeof(CasIv))); RV_TRY(bbest.reserve((size_t)ccap * 8)); RV_TRY(bwm.reserve((size_t)ccap * 4)); RV_TRY(bdep.reserve((size_t)ccap * 4));
    RV_TRY(bres.reserve((size_t)ccap * sizeof(CasRes))); RV_TRY(bctr.reserve(64)); RV_TRY(bund.reserve((size_t)ccap * 4));
    u32 *counters = bctr.as<u32>();
    RV_HIP(hipMemsetAsync(counters, 0, 64, q));

    // ---- witnesses
    DBuf &bwp0 = cb.d[19], &bwv0 = cb.d[20], &bwr = cb.d[21], &bwrk = cb.d[22], &bwrk0 = cb.d[23];
    RV_TRY(bwp0.reserve((size_t)wcap * sizeof(sa_t))); RV_TRY(bwv0.reserve((size_t)wcap * 4)); RV_TRY(bwr.reserve(2 * WT_REGIONS * 4 + 64));
    RV_TRY(bwrk.reserve((size_t)wcap * 4)); RV_TRY(bwrk0.reserve((size_t)wcap * 4));
    u32 hc[16];
    u32 NW = reuse ? cb.NW : 0;
    if (!reuse) {
        u32 *wreg = bwr.as<u32>();
        RV_HIP(hipMemsetAsync(wreg, 0, 2 * WT_REGIONS * 4, q));
        const u32 wrcap = wcap / WT_REGIONS;
        hipLaunchKernelGGL(k_cas_witness, dim3((unsigned)ceil_div(n, WT_TILE)), dim3(TB), 0, q, SA, LCP, BWT, n, minl, bwp0.as<sa_t>(), bwv0.as<u32>(), bwrk0.as<u32>(), wrcap, wreg);
        RV_LAUNCH_CHECK();
        // ---- matches by first coordinate
        {
            const unsigned mb = (unsigned)ceil_div((int64_t)M, TB);
            hipLaunchKernelGGL(k_cas_keys, dim3(mb), dim3(TB), 0, q, recs, M, k0.as<u64>(), v0.as<u32>());
            RV_LAUNCH_CHECK();
            int in1 = 0;
            RV_TRY(rv_radix_sort_pairs<u32>(ws, k0.as<u64>(), v0.as<u32>(), k1.as<u64>(), v1.as<u32>(), (int64_t)M, 0, bitlen64((u64)n), &in1));
            hipLaunchKernelGGL(k_cas_gather, dim3(mb), dim3(TB), 0, q, recs, (const u32 *)(in1 ? v1.as<u32>() : v0.as<u32>()), M, bpa.as<sa_t>(), bpb.as<sa_t>(),
                               blen.as<u32>(), bcc.as<u32>());
            RV_LAUNCH_CHECK();
        }
        u32 hreg[2 * WT_REGIONS];
        RV_TRY(rv_read_back(ws, hreg, wreg, WT_REGIONS * 4));
        u32 wmaxc = 0;
        for (int r = 0; r < WT_REGIONS; r++) { hreg[WT_REGIONS + r] = NW; NW += hreg[r]; wmaxc = std::max(wmaxc, hreg[r]); }
        out->witnesses = NW;
        if (wmaxc > wrcap) GIVE_UP("too many repeat witnesses (a repetitive input)");
        if (NW) {
            RV_HIP(hipMemcpyAsync(wreg + WT_REGIONS, hreg + WT_REGIONS, WT_REGIONS * 4, hipMemcpyHostToDevice, q));
            RV_HIP(hipStreamSynchronize(q));      // (hreg lives on this stack frame)
            hipLaunchKernelGGL(k_cas_wpack, dim3((unsigned)std::max<int64_t>(1, ceil_div((int64_t)wmaxc, TB)), WT_REGIONS), dim3(TB), 0, q, (const sa_t *)bwp0.as<sa_t>(),
                               (const u32 *)bwv0.as<u32>(), (const u32 *)bwrk0.as<u32>(), wrcap, (const u32 *)wreg, (const u32 *)(wreg + WT_REGIONS), bwp.as<sa_t>(), bwv.as<u32>(),
                               bwrk.as<u32>());
            RV_LAUNCH_CHECK();
        }
        cb.NW = NW;
        if (chain) {
            const char *why = nullptr;
            RV_TRY(cas_lineage(h, cb, M, NW, minl, CA, CB, &why));
            if (why) { cb.lin_picks = 0; cb.lin_rest.clear(); GIVE_UP(why); }
        }
    } else {
        if (chain && cb.lin_picks == 0) GIVE_UP("the chain of rest sub-indices was not decided");
        RV_HIP(hipMemsetAsync(bcc.p, 0, (size_t)M * 4, q));      // every match starts in the root again
        out->witnesses = NW;
    }
    // the second attempt walks the witnesses in the root's rank order and keeps a few words per sub-index and per witness
    const sa_t *wp = bwp.as<sa_t>(); const u32 *wv = bwv.as<u32>();
    CasDanger dg; memset(&dg, 0, sizeof dg);
    u32 dwalk_nmed = 0; const u32 *dwalk_boff = nullptr, *dwalk_mlist = nullptr;
    if (danger) {
        if (NW) {
            RV_TRY(k0.reserve((size_t)NW * 8)); RV_TRY(k1.reserve((size_t)NW * 8)); RV_TRY(v0.reserve((size_t)NW * 4)); RV_TRY(v1.reserve((size_t)NW * 4));
            const unsigned wb = (unsigned)ceil_div((int64_t)NW, TB);
            hipLaunchKernelGGL(k_cas_wkeys, dim3(wb), dim3(TB), 0, q, (const u32 *)bwrk.as<u32>(), NW, k0.as<u64>(), v0.as<u32>());
            RV_LAUNCH_CHECK();
            int in1 = 0;
            RV_TRY(rv_radix_sort_pairs<u32>(ws, k0.as<u64>(), v0.as<u32>(), k1.as<u64>(), v1.as<u32>(), (int64_t)NW, 0, bitlen64((u64)n), &in1));
            hipLaunchKernelGGL(k_cas_wgather, dim3(wb), dim3(TB), 0, q, (const sa_t *)bwp.as<sa_t>(), (const u32 *)bwv.as<u32>(), (const u32 *)(in1 ? v1.as<u32>() : v0.as<u32>()), NW,
                               bwp0.as<sa_t>(), bwv0.as<u32>());
            RV_LAUNCH_CHECK();
            wp = bwp0.as<sa_t>(); wv = bwv0.as<u32>();
            dg.rank = in1 ? k1.as<u64>() : k0.as<u64>();
        }
        DBuf &bdc = cb.d[24], &bdw = cb.d[25];
        const size_t per_child = 8 + 8 + 4 + sizeof(sa_t) + 1, per_wit = 4 + 4 + 4 + 8;
        RV_TRY(bdc.reserve((size_t)ccap * per_child + 64)); RV_TRY(bdw.reserve((size_t)std::max<u32>(NW, 1) * per_wit + 64));
        dg.best = bdc.as<u64>(); dg.ceil = dg.best + ccap; dg.qb = (sa_t *)(dg.ceil + ccap); dg.flag = (u32 *)(dg.qb + ccap); dg.dflag = (uint8_t *)(dg.flag + ccap);
        dg.key = bdw.as<u64>(); dg.m1 = (u32 *)(dg.key + std::max<u32>(NW, 1)); dg.p1 = dg.m1 + std::max<u32>(NW, 1); dg.m2 = dg.p1 + std::max<u32>(NW, 1);
        dg.T0 = h->dT0.as<uint8_t>(); dg.LCP = LCP;
        // the blocks of the witness list (k_cas_dwalk_blk): made once, the medium ones listed
        if (NW && !ws.opt.no_dwalk_blocks) {
            DBuf &bfl = cb.d[35], &bbid = cb.d[36], &bboff = cb.d[37], &bml = cb.d[38], &bwcls = cb.d[39];
            RV_TRY(bfl.reserve((size_t)(NW + 1) * 4)); RV_TRY(bbid.reserve((size_t)(NW + 1) * 4)); RV_TRY(bwcls.reserve((size_t)NW + 16));
            const unsigned wb = (unsigned)ceil_div((int64_t)NW, TB);
            hipLaunchKernelGGL(k_cas_blk_flags, dim3(wb), dim3(TB), 0, q, dg.rank, LCP, NW, minl, bfl.as<u32>());
            RV_LAUNCH_CHECK();
            RV_HIP(hipMemsetAsync(bfl.as<u32>() + NW, 0, 4, q));
            RV_TRY(rv_exclusive_sum_u32(ws, bfl.as<u32>(), bbid.as<u32>(), (int64_t)NW + 1));
            u32 nb = 0;
            RV_TRY(rv_read_back(ws, &nb, bbid.as<u32>() + NW, 4));
            RV_TRY(bboff.reserve((size_t)(nb + 2) * 4)); RV_TRY(bml.reserve((size_t)(nb + 2) * 4 * 3));
            u32 *mflag = bml.as<u32>(), *mid = mflag + (nb + 1), *mlist = mid + (nb + 1);
            hipLaunchKernelGGL(k_cas_blk_off, dim3(wb), dim3(TB), 0, q, (const u32 *)bfl.as<u32>(), (const u32 *)bbid.as<u32>(), NW, nb, bboff.as<u32>());
            RV_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_cas_blk_class, dim3(wb), dim3(TB), 0, q, (const u32 *)bfl.as<u32>(), (const u32 *)bbid.as<u32>(), (const u32 *)bboff.as<u32>(), NW, mflag, bwcls.as<uint8_t>());
            RV_LAUNCH_CHECK();
            RV_HIP(hipMemsetAsync(mflag + nb, 0, 4, q));
            RV_TRY(rv_exclusive_sum_u32(ws, mflag, mid, (int64_t)nb + 1));
            RV_TRY(rv_read_back(ws, &dwalk_nmed, mid + nb, 4));
            if (dwalk_nmed) {
                hipLaunchKernelGGL(k_cas_blk_list, dim3((unsigned)ceil_div((int64_t)nb, TB)), dim3(TB), 0, q, (const u32 *)mflag, (const u32 *)mid, nb, mlist);
                RV_LAUNCH_CHECK();
                dg.wcls = bwcls.as<uint8_t>();
                dwalk_boff = bboff.as<u32>(); dwalk_mlist = mlist;
            }
        }
        // every undecided sub-index this way, not only the ones the leaf kernel cannot take: the walk costs less than rebuilding a sub-index that
        // sits inside a repeat (RV_CASCADE_DANGER_MIN: only sub-indices above that size)
        dg.leaf_n = (u32)ws.opt.cascade_danger_min;
    }
    const sa_t *d_beginA = nullptr, *d_beginB = nullptr;
    if (!chain) {
        hipLaunchKernelGGL(k_cas_init, dim3((unsigned)std::max<int64_t>(1, ceil_div((int64_t)NW, TB))), dim3(TB), 0, q, biv.as<CasIv>(), bbest.as<u64>(), bwm.as<u32>(),
                           bdep.as<int32_t>(), bres.as<CasRes>(), counters, root, bwc.as<u32>(), NW, dg.best, dg.flag, dg.ceil);
        RV_LAUNCH_CHECK();
    } else {
        // the roots are the leading / trailing children of the chain's choices; its anchors go in front of the device's
        const u32 R = cb.lin_roots, P = cb.lin_picks;
        if ((int64_t)R + 16 > (int64_t)ccap || P > io.anchor_cap) GIVE_UP("more sequences than the cascade's tables hold");
        const uint8_t *tb = cb.d[32].as<uint8_t>();
        CasRootTabs rt;
        rt.beginA = (const sa_t *)(tb + cb.lin_off[0]); rt.beginB = (const sa_t *)(tb + cb.lin_off[1]);
        rt.pickA = (const int32_t *)(tb + cb.lin_off[2]); rt.pickB = (const int32_t *)(tb + cb.lin_off[3]);
        rt.rootLead = (const u32 *)(tb + cb.lin_off[4]); rt.rootTrail = (const u32 *)(tb + cb.lin_off[5]);
        rt.nA = cb.lin_nA; rt.nB = cb.lin_nB;
        d_beginA = rt.beginA; d_beginB = rt.beginB;
        hipLaunchKernelGGL(k_cas_init_roots, dim3((unsigned)std::max<int64_t>(1, ceil_div((int64_t)R, TB))), dim3(TB), 0, q, biv.as<CasIv>(), bbest.as<u64>(), bwm.as<u32>(),
                           bdep.as<int32_t>(), bres.as<CasRes>(), counters, (const CasIv *)cb.d[30].as<CasIv>(), (const int32_t *)cb.d[31].as<int32_t>(), R, dg.best, dg.flag, dg.ceil);
        RV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_cas_assign_roots, dim3((unsigned)std::max<int64_t>(1, ceil_div((int64_t)std::max(M, NW), TB))), dim3(TB), 0, q, (const sa_t *)bpa.as<sa_t>(),
                           (const sa_t *)bpb.as<sa_t>(), (const u32 *)blen.as<u32>(), bcc.as<u32>(), M, wp, bwc.as<u32>(), NW, (const CasIv *)cb.d[30].as<CasIv>(), rt, (int64_t)minl);
        RV_LAUNCH_CHECK();
        RV_HIP(hipMemcpyAsync(io.anchor_pos, cb.d[33].p, (size_t)P * 16, hipMemcpyDeviceToDevice, q));
        RV_HIP(hipMemcpyAsync(io.anchor_l, (const uint8_t *)cb.d[33].p + (size_t)P * 16, (size_t)P * 4, hipMemcpyDeviceToDevice, q));
        RV_HIP(hipMemcpyAsync(io.anchor_count, &cb.lin_picks, 4, hipMemcpyHostToDevice, q));
    }

    if (verbose) { (void)hipStreamSynchronize(q); tp[2] = cas_now(); }
    // ---- the levels: queued in batches, the level's range of sub-indices lives on the device (k_cas_advance), the host only looks
    // at the counters between batches (a level on an empty range costs its launches, nothing else)
    const unsigned agrid = (unsigned)(ceil_div((int64_t)M, TB * CAS_ITEMS) + ceil_div((int64_t)NW, TB * CAS_ITEMS));
    const int batch = std::max(1, (int)ws.opt.cascade_batch);
    int queued = 0;
    // (RV_CASCADE_PRIO=1: the level loop on a stream of the highest priority, fenced by events against the handle's own -- see rv_cascade_multi.hip)
    struct PrioScope {
        Workspace &w; hipStream_t home; RvCascadeBufs &cb; bool on = false;
        ~PrioScope() { leave(); }
        void leave() {
            if (!on) return;
            on = false;
            (void)hipEventRecord(cb.ev_out, w.stream);
            w.stream = home;
            (void)hipStreamWaitEvent(home, cb.ev_out, 0);
        }
    } prio{ws, q, cb};
    if (ws.opt.cascade_prio) {
        if (!cb.prio_stream) {
            int plo = 0, phi = 0;
            RV_HIP(hipDeviceGetStreamPriorityRange(&plo, &phi));
            RV_HIP(hipStreamCreateWithPriority(&cb.prio_stream, hipStreamNonBlocking, phi));
            RV_HIP(hipEventCreateWithFlags(&cb.ev_in, hipEventDisableTiming));
            RV_HIP(hipEventCreateWithFlags(&cb.ev_out, hipEventDisableTiming));
        }
        RV_HIP(hipEventRecord(cb.ev_in, q));
        RV_HIP(hipStreamWaitEvent(cb.prio_stream, cb.ev_in, 0));
        ws.stream = cb.prio_stream; prio.on = true;
        q = cb.prio_stream;
    }
    for (;;) {
        for (int b = 0; b < batch; b++, queued++) {
            hipLaunchKernelGGL(k_cas_assign, dim3(agrid), dim3(TB), 0, q, (const sa_t *)bpa.as<sa_t>(), (const sa_t *)bpb.as<sa_t>(), (const u32 *)blen.as<u32>(), bcc.as<u32>(), M,
                               wp, wv, bwc.as<u32>(), NW, (const CasIv *)biv.as<CasIv>(), (const CasRes *)bres.as<CasRes>(),
                               bbest.as<u64>(), bwm.as<u32>(), (int64_t)minl, queued == 0 ? 1 : 0, (const u64 *)((danger && NW) ? dg.ceil : nullptr));
            RV_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_cas_winner, dim3((unsigned)ceil_div((int64_t)M, TB)), dim3(TB), 0, q, (const sa_t *)bpa.as<sa_t>(), (const sa_t *)bpb.as<sa_t>(),
                               (const u32 *)blen.as<u32>(), (const u32 *)bcc.as<u32>(), M, (const CasIv *)biv.as<CasIv>(), bres.as<CasRes>(), (const u64 *)bbest.as<u64>(), (int64_t)minl);
            RV_LAUNCH_CHECK();
            if (danger && NW) {
                const unsigned wb = (unsigned)ceil_div((int64_t)NW, TB);
                hipLaunchKernelGGL(k_cas_dmark, dim3(256), dim3(TB), 0, q, (const CasIv *)biv.as<CasIv>(), (const u64 *)bbest.as<u64>(), (const u32 *)bwm.as<u32>(), minl, (const u32 *)counters, dg);
                RV_LAUNCH_CHECK();
                hipLaunchKernelGGL(k_cas_dwalk, dim3(wb), dim3(TB), 0, q, wp, wv, (const u32 *)bwc.as<u32>(), NW, (const CasIv *)biv.as<CasIv>(), (const u64 *)bbest.as<u64>(),
                                   (const u32 *)bwm.as<u32>(), minl, dg);
                RV_LAUNCH_CHECK();
                if (dwalk_nmed) {
                    hipLaunchKernelGGL(k_cas_dwalk_blk, dim3((unsigned)dwalk_nmed), dim3(TB), 0, q, wp, wv, (const u32 *)bwc.as<u32>(), dwalk_boff, dwalk_mlist, dwalk_nmed,
                                       (const CasIv *)biv.as<CasIv>(), (const u64 *)bbest.as<u64>(), (const u32 *)bwm.as<u32>(), minl, dg);
                    RV_LAUNCH_CHECK();
                }
                hipLaunchKernelGGL(k_cas_dpick, dim3(wb), dim3(TB), 0, q, wp, wv, (const u32 *)bwc.as<u32>(), NW, (const CasIv *)biv.as<CasIv>(), (const CasRes *)bres.as<CasRes>(),
                                   (const u64 *)bbest.as<u64>(), (const u32 *)bwm.as<u32>(), minl, dg);
                RV_LAUNCH_CHECK();
                hipLaunchKernelGGL(k_cas_dwrite, dim3(wb), dim3(TB), 0, q, wp, (const u32 *)bwc.as<u32>(), NW, dg);
                RV_LAUNCH_CHECK();
            }
            hipLaunchKernelGGL(k_cas_decide, dim3(256), dim3(TB), 0, q, biv.as<CasIv>(), bbest.as<u64>(), bwm.as<u32>(), bdep.as<int32_t>(),
                               bres.as<CasRes>(), minl, counters, ccap, bund.as<u32>(), (u32)RV_LEAF_N, io, dg, (danger && NW) ? 1 : 0);
            RV_LAUNCH_CHECK();
        }
        RV_TRY(rv_read_back(ws, hc, counters, sizeof hc));
        // (bits 1 and 4: the table of sub-indices / the anchor area is full -- retries of the second attempt take a slot each: not an error of the
        // input, the level pipeline completes such a run)
        if (hc[C_ERR] & ~5u) { rv_set_error("cascade: device error %u", hc[C_ERR]); return -1; }
        if (hc[C_ERR]) GIVE_UP("the cascade's tables are full");
        if (hc[C_MAXN] > (u32)RV_LEAF_N) break;      // an undecided sub-index the leaf kernel cannot take
        if (hc[C_HI] == hc[C_LO]) break;
        if (queued > 1000000) { rv_set_error("cascade: no progress"); return -1; }
    }
    prio.leave(); q = ws.stream;
    const int level = (int)hc[C_LEVELS];
    const u32 hi = hc[C_NCHILD];
    if (verbose) tp[3] = cas_now();
    out->levels = level; out->children = hi; out->solved = hc[C_NSOLVED]; out->unsolved = hc[C_NUNSOLVED];
    const u32 U = hc[C_NUND];
    out->undecided = U;
    if (hc[C_MAXN] > (u32)RV_LEAF_N) {
        // an undecided sub-index the leaf kernel cannot take: nothing of this attempt may stay
        out->why = "an undecided sub-index above the leaf kernel's size";
        if (verbose) fprintf(stderr, "cascade%s: gave up: %s (%u ranks; %u levels, %u sub-indices, %u undecided, %u decided from witnesses, %u not)\n", danger ? " (second attempt)" : "", out->why, hc[C_MAXN], level, hi, U, hc[C_NSOLVED], hc[C_NUNSOLVED]);
        return 0;
    }
    hipLaunchKernelGGL(k_cas_stats, dim3(256), dim3(TB), 0, q, (const u32 *)counters, (const int32_t *)bdep.as<int32_t>(), io);
    RV_LAUNCH_CHECK();
    if (chain) {      // the members of the chain were visited on the host
        hipLaunchKernelGGL(k_cas_lineage_stats, dim3(1), dim3(64), 0, q, io.stats, (unsigned long long)cb.lin_steps, (unsigned long long)cb.lin_maxdepth);
        RV_LAUNCH_CHECK();
    }
    if (U > 0) {
        RV_TRY(bsz.reserve((size_t)(U + 1) * 8));
        u64 *sizes = bsz.as<u64>();
        hipLaunchKernelGGL(k_cas_sizes, dim3((unsigned)ceil_div((int64_t)U + 1, TB)), dim3(TB), 0, q, (const u32 *)bund.as<u32>(), U, (const CasIv *)biv.as<CasIv>(), sizes);
        RV_LAUNCH_CHECK();
        RV_TRY(rv_exclusive_sum_u64(ws, sizes, sizes, (int64_t)U + 1));
        u64 mu = 0;
        RV_TRY(rv_read_back(ws, &mu, sizes + U, 8));
        out->rebuilt_ranks = (int64_t)mu;
        RV_TRY(io.lvSA->reserve((size_t)(mu + 64) * sizeof(sa_t))); RV_TRY(io.lvLCP->reserve((size_t)(mu + 64) * sizeof(lcp_t))); RV_TRY(io.lvBWT->reserve((size_t)mu + 64));
        RV_TRY(io.roots->reserve((size_t)U * sizeof(RvLeafRoot)));
        RvLeafRoot *roots = io.roots->as<RvLeafRoot>();
        hipLaunchKernelGGL(k_cas_roots, dim3((unsigned)ceil_div((int64_t)U, TB)), dim3(TB), 0, q, (const u32 *)bund.as<u32>(), U, (const CasIv *)biv.as<CasIv>(),
                           (const int32_t *)bdep.as<int32_t>(), (const u64 *)sizes, roots);
        RV_LAUNCH_CHECK();
        DBuf &bord = cb.d[26];
        RV_TRY(bord.reserve((size_t)(mu + 64) * 2));
        hipLaunchKernelGGL(k_cas_rank, dim3(U, BN / TB), dim3(TB), 0, q, (const RvLeafRoot *)roots, (const uint8_t *)h->dT0.as<uint8_t>(), bord.as<uint16_t>());
        RV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_cas_emit, dim3(U, BN / TB), dim3(TB), 0, q, (const RvLeafRoot *)roots, (const uint8_t *)h->dT0.as<uint8_t>(), (const uint16_t *)bord.as<uint16_t>(),
                           io.lvSA->as<sa_t>(), io.lvLCP->as<lcp_t>(), io.lvBWT->as<uint8_t>(), h->nsep[0], (int64_t)root.a0, (int64_t)root.b0,
                           d_beginA, cb.lin_nA, d_beginB, cb.lin_nB);
        RV_LAUNCH_CHECK();
        RvLeafArgs la;
        la.roots = roots;
        la.SA = io.lvSA->as<sa_t>(); la.LCP = io.lvLCP->as<lcp_t>(); la.BWT = io.lvBWT->as<uint8_t>();
        la.nsep0 = h->nsep[0]; la.minl = minl_in; la.lcap = h->maxlcp;
        la.stage_cap = io.stage_cap;
        la.anchor_count = io.anchor_count; la.anchor_cap = io.anchor_cap; la.anchor_l = io.anchor_l; la.anchor_pos = io.anchor_pos;
        la.stats = io.stats;
        la.trace = 0; la.trace_count = io.anchor_count + 1; la.trace_cap = 0; la.trace_out = nullptr;
        la.err = io.leaf_err;
        RV_TRY(rv_leaf_launch(ws, la, (int)U));
    }
    if (verbose) {
        (void)hipStreamSynchronize(q); tp[4] = cas_now();
        fprintf(stderr, "cascade%s: %u matches, %u witnesses, %d levels, %u sub-indices, %u undecided (%lld ranks rebuilt), %lld decided from witnesses | ms: scan %.2f witnesses+sort %.2f levels %.2f rebuild+leaf %.2f\n",
                danger ? " (second attempt)" : "", M, NW, level, hi, U, (long long)out->rebuilt_ranks, (long long)out->solved, (tp[1] - tp[0]) * 1e3, (tp[2] - tp[1]) * 1e3, (tp[3] - tp[2]) * 1e3, (tp[4] - tp[3]) * 1e3);
    }
    out->done = true;
    return 0;
#undef GIVE_UP
}

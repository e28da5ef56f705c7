// rv_scan.hip -- MUM scans over (concatenated) SA/LCP arrays on gfx950.
//
//   k_scan_pair  : getmums / getmums_rem predicate   (reveallib/reveal.c:55-116, :119-180)
//   k_scan_multi : getmultimums / getmultimems        (reveallib/reveal.c:436-580, :292-434;
//                  ismultimum :227-259, ismultimem :261-290) in stack-free form
//
// Both run over the whole frontier of the recursion at once: the sub-indices
// of one level are laid out back to back, every sub-index starts with LCP 0,
// and the arrays end with a virtual LCP 0, so a sub-index boundary behaves
// exactly like the array ends of the reference's per-index loops.
//
// The pair scan is the roofline-judged kernel.  SURVEY 8(d) prices it at 8 B per
// rank (4 B SA + 4 B LCP); what it streams is 5 B per rank -- LCP (16-byte
// non-temporal loads) and the BWT byte, whose bit 7 carries the side of the
// separator -- and SA is fetched for the survivors only (no gathers from T).
// bench.py reports both fractions (the 8-B model and the bytes moved).  Output
// is order preserving and free of hot atomics: per-tile slots + a small
// compaction kernel.
//
// The scans for more than two samples (k_full_scan, k_scan_multi) are built on
// the same skeleton: a wave streams 512 ranks, eight per lane, the neighbours'
// values come by shuffle, every per-rank test is a bit operation on 24-rank
// windows, and only the few ranks that pass touch memory again.
#include "rv_common.h"
#include "rv_scan.h"
#include <hip/hip_ext.h>
#include <algorithm>

namespace {

constexpr int TB = 256;
constexpr int PAIR_ITEMS = RV_PAIR_TILE / 64;   // ranks per lane; a tile (RV_PAIR_TILE ranks) is what one wave scans
constexpr int PAIR_TILE = TB * PAIR_ITEMS;      // ranks per workgroup
__device__ inline int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

__device__ inline bool is_lower_c(uint8_t c) { return c >= 'a' && c <= 'z'; }

// left-maximality test of reveal.c:81-85 on the characters in front of the two
// suffixes (ca belongs to the smaller text position; only it is inspected for
// N/$/lower).  The characters come from the BWT level array, '$' standing in
// for "position 0".
__device__ inline bool left_maximal(uint8_t ca, uint8_t cb) {
    return (ca != cb) || ca == 'N' || ca == '$' || is_lower_c(ca);
}

__device__ inline bool lcp_lt(lcp_t v, int minl) {
#ifdef RV_SA64
    return v < (lcp_t)minl;      // unsigned compare, as the reference's uint32 lcp_t
#else
    return v < minl;
#endif
}

// A wave's 1024 LCP values, sixteen consecutive ranks per lane.  Loaded the way the lanes want them -- lane i the 64 bytes at 64 i, four 16-byte
// loads -- every load instruction touches 64 different cache lines for a quarter of each (the pair scan with this pattern: 646 us at 2 x 250 Mbp
// against 620 with eight ranks per lane, its instruction count more than halved).  So the loads are the memory system's: lane i takes 16 bytes at
// 16 i, 1 KB per instruction, and the values change lanes in the wave's corner of LDS (rows padded to 80 bytes: conflict-free 16-byte reads).
// No workgroup barrier: the corner belongs to one wave, whose LDS operations complete in order.
constexpr int LT_ROW = 5;      // uint4 per lane's row (four used)
__device__ inline void wave_lcp16(const lcp_t *__restrict__ LCP, int64_t wfirst, int lane, uint4 *__restrict__ corner, u32 *out /* [16] */) {
    const v4i *src = reinterpret_cast<const v4i *>(LCP + wfirst);
    v4i v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = __builtin_nontemporal_load(src + j * 64 + lane);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r4 = j * 64 + lane;
        corner[(r4 >> 2) * LT_ROW + (r4 & 3)] = make_uint4((u32)v[j].x, (u32)v[j].y, (u32)v[j].z, (u32)v[j].w);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 t = corner[lane * LT_ROW + j];
        out[4 * j] = t.x; out[4 * j + 1] = t.y; out[4 * j + 2] = t.z; out[4 * j + 3] = t.w;
    }
}

// four ranks at once: x = their BWT bytes (bit 7 = side of the separator, RV_BWT_SIDE), y = the bytes of the ranks in front of each.
// -> bit 7 of byte i set iff the two suffixes start on different sides of the separator (not a repeat inside one sample, reveal.c:73) and are
// left-maximal (reveal.c:81-85: the characters differ, or the one in front is N / $ / lower case -- which only decides anything when they are
// equal, and then either of them will do).  (v + 0x7f..: bit 7 of a 7-bit byte set iff it is not zero; no carry leaves such a byte.)
__device__ inline u32 pair_bytes4(u32 x, u32 y) {
    const u32 K = 0x7f7f7f7fu;
    const u32 xc = x & K, yc = y & K;
    const u32 ne = (xc ^ yc) + K;
    const u32 plain = ((xc ^ 0x4e4e4e4eu) + K) & ((xc ^ 0x24242424u) + K);      // neither 'N' nor '$'
    const u32 low = (xc + 0x1f1f1f1fu) & ~(xc + 0x05050505u);                    // 'a' .. 'z'
    return (x ^ y) & (ne | ~plain | low) & 0x80808080u;
}
__device__ inline u32 bits4_of_bytes(u32 e) {      // bit 7 of the four bytes -> bits 0 .. 3
    u32 v = (e >> 7) & 0x01010101u;
    v |= v >> 7;
    v |= v >> 14;
    return v & 0xfu;
}

__global__ __launch_bounds__(TB) void k_scan_pair(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, int64_t m,
                                                  const uint8_t *__restrict__ BWT, sa_t nsep0, int minl,
                                                  RvPairRec *__restrict__ slots, RvPairRec *__restrict__ ovf, u32 ovf_cap,
                                                  u32 *__restrict__ ovf_counter, u32 *__restrict__ tilecnt, u32 *__restrict__ tileovf,
                                                  unsigned long long *__restrict__ best, RvPairRec *__restrict__ picks, int nsubs) {
    static_assert(PAIR_ITEMS == 16, "sixteen ranks per lane: four 16-byte loads of LCP, one of BWT");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // one stretch of TB x 16 ranks per block (persistent blocks walking several tiles measured 25 % slower: the
    // block-wide append at the end of a tile then no longer overlaps with another block's loads)
    const int64_t tile = blockIdx.x;
    const int64_t i0 = tile * PAIR_TILE + (int64_t)threadIdx.x * PAIR_ITEMS;
    if (blockIdx.x == 0 && threadIdx.x == 0) tilecnt[ceil_div_dev(m, RV_PAIR_TILE)] = 0;      // the slot that makes the exclusive scan yield the total
    // tables of the device-side picker that runs right behind this kernel (k_pick_slots1/2), nsubs = 0 otherwise
    for (int64_t s2 = (int64_t)blockIdx.x * TB + threadIdx.x; s2 < nsubs; s2 += (int64_t)gridDim.x * TB) { best[s2] = 0; picks[RV_PAIR_HDR + s2].rank = 0xFFFFFFFFu; }

    // Halo of the wave (the rank in front of its first one, the LCP behind its last one): wave-uniform addresses, issued
    // before the streaming loads.  Loaded by lane 0 / 63 after the shuffles they were a second, dependent memory round
    // trip per wave (measured: the bare load pattern of this kernel streams 6.3 TB/s, the kernel 3.5).
    const int64_t wfirst = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63)) * (int64_t)PAIR_ITEMS + tile * PAIR_TILE;
    const int64_t wnext = wfirst + 64 * PAIR_ITEMS;
    u32 h_lc = 0, h_nlc = 0, h_bw = 0;
    if (wfirst > 0 && wfirst - 1 < m) { h_lc = (u32)LCP[wfirst - 1]; h_bw = BWT[wfirst - 1]; }
    if (wnext < m) h_nlc = (u32)LCP[wnext];

    // What is streamed: LCP (4 B) and the BWT byte, whose bit 7 says on which side of the separator the suffix starts
    // (RV_BWT_SIDE, rv_common.h) -- the predicate of reveal.c:61-85 needs nothing else of SA.  SA is fetched for the
    // survivors only (one in a few hundred ranks).
    __shared__ uint4 s_corner[TB / 64][64 * LT_ROW];
    u32 lc[PAIR_ITEMS + 2], bx[PAIR_ITEMS / 4];      // lc[0] / lc[17]: the ranks in front of and behind my sixteen
    if (wnext <= m) {      // (the whole wave inside the arrays)
        // streamed once: non-temporal 16-byte loads, a kilobyte per instruction (the BWT bytes are sixteen per lane as they lie)
        const v4u32 bb = __builtin_nontemporal_load(reinterpret_cast<const v4u32 *>(BWT + i0));
        bx[0] = bb.x; bx[1] = bb.y; bx[2] = bb.z; bx[3] = bb.w;
        wave_lcp16(LCP, wfirst, lane, s_corner[w], lc + 1);
    } else {
#pragma unroll
        for (int k = 0; k < PAIR_ITEMS / 4; k++) bx[k] = 0;
#pragma unroll
        for (int k = 0; k < PAIR_ITEMS; k++) {
            lc[1 + k] = (i0 + k < m) ? (u32)LCP[i0 + k] : 0u;
            bx[k >> 2] |= ((i0 + k < m) ? (u32)BWT[i0 + k] : 0u) << (8 * (k & 3));
        }
    }
    // neighbours: previous rank's LCP / BWT byte, next rank's LCP (0 past the end)
    lc[0] = (u32)__shfl_up((int)lc[PAIR_ITEMS], 1, 64);
    lc[PAIR_ITEMS + 1] = (u32)__shfl_down((int)lc[1], 1, 64);
    u32 plast = (u32)__shfl_up((int)bx[PAIR_ITEMS / 4 - 1], 1, 64);
    if (lane == 0) { lc[0] = h_lc; plast = h_bw << 24; }
    if (lane == 63) lc[PAIR_ITEMS + 1] = h_nlc;

    // The predicate, branch-free and mostly not per rank.  LCP: "long enough, larger than both neighbours" (unique, reveal.c:76-79) is ONE
    // comparison with the maximum of three (the neighbours and minl - 1).  BWT: the side bit and left-maximality of four ranks at a time
    // (pair_bytes4).  28 vector instructions per rank in the form with a compare chain per rank -- the kernel spent more time issuing them
    // than waiting for memory (SQ_INSTS_VALU, r04_insts_c4.txt) -- about 11 now.
    // Ranks past the end were loaded as zeros and a sub-index' first rank has LCP 0, so neither can pass: no bounds tests are needed here.
#ifdef RV_SA64
    const u32 thr = minl < 0 ? 0xFFFFFFFFu : (minl >= 1 ? (u32)minl - 1u : 0u);      // (unsigned compare with minl, as the reference's uint32 lcp_t: nothing is "long enough" for a negative minl)
#else
    const u32 thr = minl >= 1 ? (u32)minl - 1u : 0u;
#endif
    u32 hit = 0;        // bitmask over my ranks
#pragma unroll
    for (int k = 0; k < PAIR_ITEMS; k++) {
        const u32 a = lc[k], c = lc[k + 2];
        u32 mx = a > c ? a : c;
        mx = mx > thr ? mx : thr;
        hit |= (u32)(lc[k + 1] > mx) << k;
    }
    u32 okb = 0;
#pragma unroll
    for (int k = 0; k < PAIR_ITEMS / 4; k++) {
        const u32 y = __builtin_amdgcn_alignbit(bx[k], k ? bx[k - 1] : plast, 24);      // byte i = the BWT byte of the rank in front of rank 4k + i
        okb |= bits4_of_bytes(pair_bytes4(bx[k], y)) << (4 * k);
    }
    hit &= okb;
    // order-preserving append of the survivors.  A tile is what one wave scans: no workgroup barrier, a wave
    // retires as soon as its own loads are consumed.
    const u32 mine = __popc(hit);
    const u32 inc = rv_wave_incl_sum_u32(mine);
    const u32 tot = (u32)__builtin_amdgcn_readlane((int)inc, 63);
    const int64_t wtile = tile * (TB / 64) + w;
    if (wfirst >= m) return;                     // (a wave entirely past the end)
    // The first RV_PAIR_SLOTS survivors of a tile go to the tile's own slots (no atomics at all in the common case); only a
    // tile with more takes one atomic for room in the overflow array.  k_pair_compact / k_pick_slots read them in tile
    // (= rank) order.
    u32 base = 0;
    if (lane == 0) {
        if (tot > RV_PAIR_SLOTS) base = atomicAdd(ovf_counter, tot - RV_PAIR_SLOTS);
        tilecnt[wtile] = tot;
        tileovf[wtile] = base;
    }
    base = (u32)__shfl((int)base, 0, 64);
    // one trip per survivor of the lane (usually one), not one per rank
    u32 hh = hit, q = inc - mine;                // q: index inside the tile
    while (hh) {
        const int k = __builtin_ctz(hh);
        hh &= hh - 1;
        const sa_t s1 = SA[i0 + k], s0 = SA[i0 + k - 1];      // (a survivor is never the first rank of the arrays: its LCP is 0)
        RvPairRec r;
        r.a = s1 < s0 ? s1 : s0;
        r.b = s1 < s0 ? s0 : s1;
        r.l = (u32)LCP[i0 + k];                  // (from cache; indexing the register copy by k would send it to scratch)
        r.rank = (u32)(i0 + k);
        if (q < RV_PAIR_SLOTS) slots[(size_t)wtile * RV_PAIR_SLOTS + q] = r;
        else { const u32 o = base + (q - RV_PAIR_SLOTS); if (o < ovf_cap) ovf[o] = r; }
        q++;
    }
}

// one thread per (tile, slot): dense, rank-ordered output
// out[0] is a header {total, overflow count, *err, 0} so the host needs a single copy (and a single sync) per level.
__global__ __launch_bounds__(TB) void k_pair_compact(const RvPairRec *__restrict__ slots, const RvPairRec *__restrict__ ovf,
                                                     const u32 *__restrict__ tilecnt, const u32 *__restrict__ tileovf,
                                                     const u32 *__restrict__ tileoff, int64_t ntile, RvPairRec *__restrict__ out, u32 out_cap,
                                                     u32 *__restrict__ ovf_counter, const u32 *__restrict__ err, u32 ovf_cap) {
    const int64_t id = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (id == 0) {
        u32 *hdr = reinterpret_cast<u32 *>(out);
        hdr[0] = tileoff[ntile]; hdr[1] = *ovf_counter; hdr[2] = err ? *err : 0u; hdr[3] = 0;
        *ovf_counter = 0;          // the scan of this launch sequence is done with it: ready for the next scan (no memset per level)
    }
    out += RV_PAIR_HDR;
    const int64_t t = id / RV_PAIR_SLOTS;
    const u32 j = (u32)(id % RV_PAIR_SLOTS);
    if (t >= ntile) return;
    const u32 cnt = tilecnt[t];
    if (j >= cnt) return;
    const u32 o = tileoff[t];
    if (o + j < out_cap) out[o + j] = slots[(size_t)t * RV_PAIR_SLOTS + j];
    const u32 ob = tileovf[t];
    // (a scan whose overflow did not fit is repeated with a larger buffer: what it could not store must not be read either)
    for (u32 q = RV_PAIR_SLOTS + j; q < cnt; q += RV_PAIR_SLOTS)
        if (o + q < out_cap && ob + (q - RV_PAIR_SLOTS) < ovf_cap) out[o + q] = ovf[ob + (q - RV_PAIR_SLOTS)];
}

// ---- built-in picker on the device --------------------------------------------------
// The recursion's built-in callbacks (longest match, ties -> smallest position) need one
// record per sub-index, not every MUM of the level: two passes over the packed records, an atomicMax on
// (l, -a) per sub-index and a gather of the winners.  picks[0] = header, picks[1+s] = winner of sub-index s
// (rank 0xFFFFFFFF = none; the scan kernel initialises the tables).
__device__ inline int sub_of_rank(const int64_t *__restrict__ sub_start, int nsubs, int64_t r) {
    int lo = 0, hi = nsubs;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sub_start[mid] <= r) lo = mid + 1; else hi = mid; }
    return lo - 1;
}
// the same with a tile -> sub-index table (sub-index of the first rank of every RV_TSUB_TILE ranks): two independent loads
// bound the search to the sub-indices that start inside the tile -- usually none or one -- instead of log2(nsubs) dependent ones
__device__ inline int sub_of_rank_t(const int64_t *__restrict__ sub_start, int nsubs, const int *__restrict__ tile_sub, int64_t ntsub, int64_t r) {
    if (!tile_sub) return sub_of_rank(sub_start, nsubs, r);
    const int64_t tt = r / RV_TSUB_TILE;
    int lo = tile_sub[tt] + 1;
    int hi = tt + 1 < ntsub ? tile_sub[tt + 1] + 1 : nsubs;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sub_start[mid] <= r) lo = mid + 1; else hi = mid; }
    return lo - 1;
}
__device__ inline u64 pick_key(const RvPairRec &r) { return ((u64)r.l << 32) | (u64)(0xFFFFFFFFu - (u32)r.a); }

// The same picker straight from the scan's per-tile slots (+ overflow): no tile-count scan, no compaction -- the
// untraced recursion never looks at the packed list.  One thread per (tile, slot); it also walks the tile's share of
// the overflow array.  PASS 1: atomicMax per sub-index; PASS 2: the winners, the header {0, overflow count, *err, 0},
// and the overflow counter back to zero for the next scan.
template <int PASS>
__global__ __launch_bounds__(TB) void k_pick_slots(const RvPairRec *__restrict__ slots, const RvPairRec *__restrict__ ovf, u32 ovf_cap,
                                                   const u32 *__restrict__ tilecnt, const u32 *__restrict__ tileovf, int64_t ntile,
                                                   const int64_t *__restrict__ sub_start, int nsubs, unsigned long long *__restrict__ best,
                                                   RvPairRec *__restrict__ picks, u32 *__restrict__ ovf_counter, const u32 *__restrict__ err,
                                                   const int *__restrict__ tile_sub, int64_t ntsub) {
    const int64_t id = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (PASS == 2 && id == 0) {
        u32 *hdr = reinterpret_cast<u32 *>(picks);
        hdr[0] = 0; hdr[1] = *ovf_counter; hdr[2] = err ? *err : 0u; hdr[3] = 0;
        *ovf_counter = 0;
    }
    const int64_t t = id / RV_PAIR_SLOTS;
    const u32 j = (u32)(id % RV_PAIR_SLOTS);
    const int lane = threadIdx.x & 63;
    const u32 cnt = t < ntile ? tilecnt[t] : 0u;
    const u32 ob = t < ntile ? tileovf[t] : 0u;
    // PASS 1: the wave's best of ONE sub-index stays in registers and is merged over the workgroup at the end (with a handful of
    // sub-indices every wave of the grid aims at the same few words of `best`); a wave that meets further sub-indices sends those on at once
    int st_sub = -1; u64 st_key = 0;
    auto raise = [&](int lsub, u64 v) {      // (only a wave that would raise the maximum goes to the atomic unit)
        if ((unsigned long long)v > __atomic_load_n(&best[lsub], __ATOMIC_RELAXED)) atomicMax(&best[lsub], (unsigned long long)v);
    };
    for (u32 q = j; ; q += RV_PAIR_SLOTS) {
        const bool have = q < cnt && (q < RV_PAIR_SLOTS || ob + (q - RV_PAIR_SLOTS) < ovf_cap);
        RvPairRec r; int sub = -1; u64 key = 0;
        if (have) {
            r = q < RV_PAIR_SLOTS ? slots[(size_t)t * RV_PAIR_SLOTS + q] : ovf[ob + (q - RV_PAIR_SLOTS)];
            sub = sub_of_rank_t(sub_start, nsubs, tile_sub, ntsub, (int64_t)r.rank); key = pick_key(r);
        }
        if (PASS == 1) {
            u64 todo = __ballot(sub >= 0);
            while (todo) {              // one candidate per (wave, sub-index)
                const int leader = (int)__builtin_ctzll(todo);
                const int lsub = __builtin_amdgcn_readlane(sub, leader);
                const bool mine = sub == lsub;
                const u64 v = rv_wave_max_u64(mine ? key : 0);
                if (st_sub < 0 || st_sub == lsub) { st_sub = lsub; st_key = v > st_key ? v : st_key; }
                else if (lane == 0) raise(lsub, v);
                todo &= ~__ballot(mine);
            }
        } else if (have && best[sub] == (unsigned long long)key) {
            picks[RV_PAIR_HDR + sub] = r;
        }
        if (!__any(q + RV_PAIR_SLOTS < cnt)) break;       // (the wave leaves the loop together: the ballots above need all lanes)
    }
    if (PASS == 1) {
        __shared__ int s_sub[TB / 64];
        __shared__ u64 s_key[TB / 64];
        if (lane == 0) { s_sub[threadIdx.x >> 6] = st_sub; s_key[threadIdx.x >> 6] = st_key; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 0; k < TB / 64; k++) {
                const int sb = s_sub[k];
                if (sb < 0) continue;
                u64 v = s_key[k];
                for (int k2 = k + 1; k2 < TB / 64; k2++) if (s_sub[k2] == sb) { v = s_key[k2] > v ? s_key[k2] : v; s_sub[k2] = -1; }
                raise(sb, v);
            }
        }
    }
}

// ---- multi-MUM scan ------------------------------------------------------------
// Stack-free form of the LCP-interval enumeration of getmultimums
// (reveal.c:436-580).  The reference closes an interval (l, lb, ub) when it
// reads LCP[ub+1] < l, deeper intervals first, so its output is ordered by
// (ub ascending, l descending).  Here the thread at rank u emits every
// interval with ub == u: it exists iff LCP[u+1] < LCP[u]; walking left from u
// the interval values are the successive prefix minima of LCP, each > LCP[u+1].
// Only intervals of at most main.nsamples ranks can qualify (reveal.c:477), so
// the walk is cut after that many steps.  ismultimum (reveal.c:227-259) is
// evaluated in place; members are emitted in SA order.
__device__ inline int sample_of_pos(const sa_t *__restrict__ nsep, int nsep_n, sa_t pos) {   // SO[pos], interface.c:116-134
    int lo = 0, hi = nsep_n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (nsep[mid] < pos) lo = mid + 1; else hi = mid; }
    return lo;
}

__device__ inline bool ismultimum_dev(const sa_t *__restrict__ SA, const uint8_t *__restrict__ BWT, const sa_t *__restrict__ nsep, int nsamples,
                                      int64_t lb, int64_t ub) {
    if (nsamples == 2) {
        if ((SA[ub] > nsep[0]) == (SA[lb] > nsep[0])) return false;
    } else if (nsamples <= 64) {
        u64 seen = 0;
        for (int64_t j = lb; j <= ub; j++) {
            const u64 bit = 1ull << sample_of_pos(nsep, nsamples - 1, SA[j]);
            if (seen & bit) return false;
            seen |= bit;
        }
    } else {
        for (int64_t j = lb; j <= ub; j++) {
            const int sj = sample_of_pos(nsep, nsamples - 1, SA[j]);
            for (int64_t k = lb; k < j; k++) if (sample_of_pos(nsep, nsamples - 1, SA[k]) == sj) return false;
        }
    }
    for (int64_t j = lb; j < ub; j++) {       // reveal.c:246-256; BWT holds '$' where SA == 0
        const uint8_t ca = BWT[j] & RV_BWT_CHAR, cb = BWT[j + 1] & RV_BWT_CHAR;
        if (cb == '$' || ca != cb || ca == 'N' || ca == '$' || is_lower_c(ca)) return true;
    }
    return false;
}

// EMIT=false: count records/members of rank u; EMIT=true: write them.
template <bool EMIT>
__device__ inline void multi_walk(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, int64_t m, const uint8_t *__restrict__ BWT,
                                  const sa_t *__restrict__ nsep, int nsamples, int minl, int minn, int64_t u,
                                  u32 &nrec, u32 &nmem, RvMultiRec *rec_out, uint16_t *so_out, sa_t *pos_out, u32 rec_cap, u32 mem_cap,
                                  const int64_t *__restrict__ sub_start, const int *__restrict__ sub_want, int nsubs) {
    nrec = 0; nmem = 0;
    if (u < 1 || u >= m) return;
    const u32 lnext = (u + 1 < m) ? (u32)LCP[u + 1] : 0u;
    u32 cur = (u32)LCP[u];
    if (cur <= lnext) return;
    int64_t p = u;                   // invariant: LCP[p+1..u] >= cur, candidate lb is found by moving p left
    const u32 lmin = (u32)(minl > 1 ? minl : 1);
    u32 rq = 0, mq = 0;
    while (cur > lnext && cur >= lmin) {
        // extend left while LCP[p-1+... ] >= cur : lb = first position (going left) with LCP < cur
        int64_t lb = p - 1;
        while (lb >= 0 && (u32)LCP[lb] >= cur) { lb--; if (u - lb + 1 > nsamples) break; }
        if (lb < 0) break;                                   // cannot happen: LCP of a sub-index' first rank is 0
        const int64_t n = u - lb + 1;
        if (n > nsamples) break;                              // every further interval is larger still
        bool take = n >= minn;
        if (take && sub_want) {      // pre-selection for the built-in picker: only matches present in every sample of the sub-index
            int lo2 = 0, hi2 = nsubs;
            while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (sub_start[mid] <= u) lo2 = mid + 1; else hi2 = mid; }
            const int w = sub_want[lo2 - 1];      // (0: every match of this sub-index; a value no match has: none)
            take = (w == 0) | (n == (int64_t)w);
        }
        if (take && ismultimum_dev(SA, BWT, nsep, nsamples, lb, u)) {
            if (EMIT) {
                if (rq < rec_cap) { RvMultiRec r; r.l = cur; r.n = (u32)n; r.ub = (u32)u; r.pad = 0; rec_out[rq] = r; }
                for (int64_t j = lb; j <= u; j++, mq++)
                    if (mq < mem_cap) { so_out[mq] = (uint16_t)sample_of_pos(nsep, nsamples - 1, SA[j]); pos_out[mq] = SA[j]; }
                rq++;
            } else {
                rq++; mq += (u32)n;
            }
        }
        cur = (u32)LCP[lb];                                  // value of the enclosing interval
        p = lb;
    }
    nrec = rq; nmem = mq;
}

// The enumeration on the pair scan's skeleton: a wave streams 512 ranks (eight per lane, non-temporal 16-byte loads), the rank behind a lane's last
// one comes by shuffle, and the ranks that close an interval at all -- LCP[u] > LCP[u+1], LCP[u] >= minl: one in `nsamples` for related genomes,
// next to none for unrelated ones -- are listed in the wave's corner of LDS.  The lanes then take one listed rank each: a first walk counts its
// records and members, a wave-wide prefix sum and ONE atomic per wave and list give it room in emission order, a second walk (over lines the first
// one left in the cache) writes.  No workgroup barrier; a wave whose ranks close nothing retires after its loads.  (Before: a thread per rank, each
// with its own 4-byte loads and both walks, 256-rank workgroups between two barriers.)
__global__ __launch_bounds__(TB) void k_scan_multi(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, int64_t m,
                                                   const uint8_t *__restrict__ BWT, const sa_t *__restrict__ nsep, int nsamples, int minl, int minn,
                                                   RvMultiRec *__restrict__ rec, uint16_t *__restrict__ so, sa_t *__restrict__ pos,
                                                   u32 rec_cap, u32 mem_cap, u32 *__restrict__ counters, uint4 *__restrict__ tiletab,
                                                   const int64_t *__restrict__ sub_start, const int *__restrict__ sub_want, int nsubs) {
    constexpr int ITEMS = RV_MULTI_TILE / 64;
    static_assert(ITEMS == 8, "eight ranks per lane");
    __shared__ uint16_t s_c[TB / 64][RV_MULTI_TILE], s_ro[TB / 64][RV_MULTI_TILE];
    __shared__ u32 s_mo[TB / 64][RV_MULTI_TILE];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wtile = (int64_t)blockIdx.x * (TB / 64) + w;
    const int64_t wfirst = wtile * RV_MULTI_TILE;
    if (wfirst >= m) return;
    const int64_t i0 = wfirst + (int64_t)lane * ITEMS;
    u32 h_nlc = 0;
    if (wfirst + RV_MULTI_TILE < m) h_nlc = (u32)LCP[wfirst + RV_MULTI_TILE];
    u32 lc[ITEMS + 1];
    if (i0 + ITEMS <= m) {
#pragma unroll
        for (int v4 = 0; v4 < ITEMS / 4; v4++) {
            const v4i c = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(LCP + i0) + v4);
            lc[4 * v4] = (u32)c.x; lc[4 * v4 + 1] = (u32)c.y; lc[4 * v4 + 2] = (u32)c.z; lc[4 * v4 + 3] = (u32)c.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) lc[k] = (i0 + k < m) ? (u32)LCP[i0 + k] : 0u;
    }
    lc[ITEMS] = (u32)__shfl_down((int)lc[0], 1, 64);
    if (lane == 63) lc[ITEMS] = h_nlc;
    const u32 lmin = (u32)(minl > 1 ? minl : 1);
    u32 c8 = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) c8 |= (u32)((lc[k] >= lmin) & (lc[k] > lc[k + 1])) << k;      // (rank 0 and the ranks past the end hold 0)
    const u32 cnt = (u32)__popc(c8);
    const u32 inc = rv_wave_incl_sum_u32(cnt);
    const u32 tot = (u32)__builtin_amdgcn_readlane((int)inc, 63);
    {
        u32 at = inc - cnt, hh = c8;
        while (hh) { const int k = __builtin_ctz(hh); hh &= hh - 1; s_c[w][at++] = (uint16_t)(lane * ITEMS + k); }
    }
    // (the list is read by other lanes of the same wave only: its LDS operations complete in order, the fences keep the compiler from moving them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u32 run_r = 0, run_m = 0, nz = 0;
    for (u32 base = 0, round = 0; base < tot; base += 64, round++) {
        const u32 ci = base + (u32)lane;
        u32 nr = 0, nm = 0;
        if (ci < tot)
            multi_walk<false>(SA, LCP, m, BWT, nsep, nsamples, minl, minn, wfirst + (int64_t)s_c[w][ci], nr, nm, nullptr, nullptr, nullptr, 0, 0, sub_start, sub_want, nsubs);
        const u32 ir = rv_wave_incl_sum_u32(nr), im = rv_wave_incl_sum_u32(nm);
        if (ci < tot) { s_ro[w][ci] = (uint16_t)(run_r + ir - nr); s_mo[w][ci] = run_m + im - nm; }
        nz |= (u32)(nr != 0u) << round;
        run_r += (u32)__builtin_amdgcn_readlane((int)ir, 63); run_m += (u32)__builtin_amdgcn_readlane((int)im, 63);
    }
    u32 rb = 0, mb = 0;
    if (lane == 0) {
        rb = run_r ? atomicAdd(&counters[0], run_r) : 0u;
        mb = run_m ? atomicAdd(&counters[1], run_m) : 0u;
        tiletab[wtile] = make_uint4(rb, run_r, mb, run_m);
    }
    rb = (u32)__shfl((int)rb, 0, 64); mb = (u32)__shfl((int)mb, 0, 64);
    for (u32 base = 0, round = 0; base < tot; base += 64, round++) {
        const u32 ci = base + (u32)lane;
        if (ci < tot && ((nz >> round) & 1u)) {
            const u32 r0 = rb + (u32)s_ro[w][ci], m0 = mb + s_mo[w][ci];      // (written by this very lane)
            u32 a2, b2;
            multi_walk<true>(SA, LCP, m, BWT, nsep, nsamples, minl, minn, wfirst + (int64_t)s_c[w][ci], a2, b2, rec + r0, so + m0, pos + m0,
                             r0 < rec_cap ? rec_cap - r0 : 0u, m0 < mem_cap ? mem_cap - m0 : 0u, sub_start, sub_want, nsubs);
        }
    }
}

}  // namespace


// ---- built-in picker for more than two samples ---------------------------------------------
// The untraced recursion's picker only takes matches present in every sample of their sub-index
// (schemes.py:227): an LCP interval of exactly want = nsamples(sub-index) ranks.  For a fixed size the
// interval ending at rank u is known directly -- lb = u - want + 1, l = min LCP[lb+1..u], valid iff
// LCP[lb] < l > LCP[u+1] -- so no walk through nested intervals, no record lists, no host-side merge:
// pass 1 takes an atomicMax of (l, -min position) per sub-index over the qualifying ranks (and lists
// them), pass 2 lets the winners write their members.
struct RvMultiCand { u32 ub, sub; unsigned long long key; };

__global__ __launch_bounds__(TB) void k_multi_pick1(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, int64_t m, const uint8_t *__restrict__ BWT,
                                                    const sa_t *__restrict__ nsep, int nsamples, int minl, int minn,
                                                    const int64_t *__restrict__ sub_start, const int *__restrict__ sub_want, int nsubs,
                                                    const int *__restrict__ tile_sub,
                                                    unsigned long long *__restrict__ best, u32 *__restrict__ /*pick_l: zeroed by the host*/,
                                                    RvMultiCand *__restrict__ cand, u32 cand_cap, u32 *__restrict__ cand_count) {
    // One block per 2048-rank tile of the host's tile -> sub-index table (so the sub-index of the block's first rank needs
    // no search), eight ranks per thread.  All global reads a tile needs are issued up front -- LCP and BWT of its ranks
    // and of the MP_HALO ranks in front (a candidate interval of up to MP_HALO ranks ends at u), the starts and sample
    // counts of the sub-indices that begin inside it -- and land in LDS; the interval's value, left-maximality and the
    // owning sub-index then cost no further memory round trip.  (One block per 256 ranks with its own search for the
    // first sub-index ran at 13 ps per rank: a chain of three to four round trips per 256 ranks, 2048 blocks in flight.)
    // At the top levels one rank in `nsamples` closes a candidate interval (every conserved position closes the group of
    // its copies) and only left-maximality thins them out; SA is gathered for the few ranks that pass.
    constexpr int MP_HALO = 16, MP_TILE = RV_TSUB_TILE, MP_ITEMS = MP_TILE / TB;
    __shared__ sa_t s_nsep[256];
    __shared__ u32 s_lcp[MP_TILE + MP_HALO + 1];
    __shared__ uint8_t s_bw[MP_TILE + MP_HALO + 1];
    __shared__ short s_ss[MP_TILE + 2];        // starts of the tile's sub-indices relative to u0 (clipped below at -2 * MP_HALO)
    __shared__ uint8_t s_want[MP_TILE + 2];    // their sample counts, 255 = more than MP_HALO (LDS is what limits the tiles in flight)
    const bool lds_sep = nsamples - 1 <= 256;
    if (lds_sep) { for (int k = threadIdx.x; k < nsamples - 1; k += TB) s_nsep[k] = nsep[k]; }
    const sa_t *sep = lds_sep ? s_nsep : nsep;
    const int64_t tt = blockIdx.x, u0 = tt * MP_TILE, ntsub = (m + MP_TILE - 1) / MP_TILE;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < MP_ITEMS; i++) {
        const int at = i * TB + (int)threadIdx.x;
        const int64_t j = u0 + at;
        s_lcp[MP_HALO + at] = j < m ? (u32)LCP[j] : 0u;
        s_bw[MP_HALO + at] = j < m ? (uint8_t)(BWT[j] & RV_BWT_CHAR) : (uint8_t)0;
    }
    if (threadIdx.x <= MP_HALO) {      // MP_HALO ranks in front, one behind
        const int64_t j = threadIdx.x < MP_HALO ? u0 - MP_HALO + threadIdx.x : u0 + MP_TILE;
        const int at = threadIdx.x < MP_HALO ? (int)threadIdx.x : MP_HALO + MP_TILE;
        const bool in = j >= 0 && j < m;
        s_lcp[at] = in ? (u32)LCP[j] : 0u;
        s_bw[at] = in ? (uint8_t)(BWT[j] & RV_BWT_CHAR) : (uint8_t)0;
    }
    const int s0 = tile_sub[tt];                                               // sub-index of rank u0
    const int s1 = tt + 1 < ntsub ? tile_sub[tt + 1] : nsubs - 1;              // ... of the next tile's first rank
    const int nss = s1 - s0 + 1;                                               // sub-indices with ranks in this tile: at most MP_TILE + 1
    for (int k = threadIdx.x; k < nss + 1 && k < MP_TILE + 2; k += TB) {
        const int q = s0 + k;
        const int64_t rel = sub_start[q < nsubs ? q : nsubs] - u0;            // (sub_start[nsubs] = m)
        s_ss[k] = (short)(rel < -2 * MP_HALO ? -2 * MP_HALO : rel > 2 * MP_TILE ? 2 * MP_TILE : (int)rel);
        const int wv = sub_want[q < nsubs ? q : nsubs - 1];
        s_want[k] = (uint8_t)(wv < 0 ? 0 : wv > 254 ? 255 : wv);
    }
    __syncthreads();
    // left-maximality evidence per pair of neighbouring ranks (reveal.c:246-256; BWT holds '$' where SA == 0) as one bit each:
    // a candidate then tests its whole window with a shift and a mask instead of a loop (the kernel is bound by instruction
    // issue: one rank in `nsamples` is a candidate, so every wave walks the slow path)
    __shared__ u64 s_d[(MP_TILE + MP_HALO) / 64 + 2];
    for (int q = threadIdx.x; q < MP_TILE + MP_HALO + 64; q += TB) {
        bool d = false;
        if (q < MP_TILE + MP_HALO) {
            const u32 ca = s_bw[q], cb = s_bw[q + 1];
            d = (cb == '$') | (ca != cb) | (ca == 'N') | (ca == '$') | ((ca - 'a') < 26u);
        }
        const u64 bits = __ballot(d);
        if (lane == 0) s_d[q >> 6] = bits;
    }
    __syncthreads();
    const u32 lmin = (u32)(minl > 1 ? minl : 1);
    // Phase 1, from LDS only: the ranks that close an interval of the wanted size with the right value and left-maximal.
    // They are few (about one in a hundred) but spread over every wave; checking their members' samples right here cost
    // every wave a gather of SA per item, eight items one after the other.  They are listed instead, and in phase 2 one
    // thread per listed rank fetches all its members at once: one more round trip per tile.
    constexpr int MP_LIST = 256;
    __shared__ u32 s_list[MP_LIST][3];     // (LDS index | want << 16, sub-index slot, interval value)
    __shared__ u32 s_nlist;
    if (threadIdx.x == 0) s_nlist = 0;
    __syncthreads();
    for (int i = 0; i < MP_ITEMS; i++) {
        const int me = MP_HALO + i * TB + (int)threadIdx.x;
        const int rel = i * TB + (int)threadIdx.x;
        const int64_t u = u0 + rel;
        bool ok = u >= 1 && u < m;
        u32 cur = 0, nxt = 0;
        if (ok) { cur = s_lcp[me]; nxt = s_lcp[me + 1]; ok = cur > nxt && cur >= lmin; }      // (past the end: stored as 0)
        if (!ok) continue;
        int lo2 = 0, hi2 = nss - 1;                                    // largest k with s_ss[k] <= rel
        while (lo2 < hi2) { const int mid = (lo2 + hi2 + 1) >> 1; if (s_ss[mid] <= rel) lo2 = mid; else hi2 = mid - 1; }
        int want = s_want[lo2];
        if (want == 255) want = sub_want[s0 + lo2];
        ok = want >= minn && want >= 2 && want <= nsamples && rel - want + 1 >= (want > MP_HALO ? sub_start[s0 + lo2] - u0 : (int64_t)s_ss[lo2]);
        if (!ok) continue;
        u32 l = cur;
        bool listed = false;
        if (want <= MP_HALO && nsamples > 2 && nsamples <= 64) {
            const int w0 = me - (want - 1);                               // pairs (w0, w0+1) .. (me-1, me)
            const int off = w0 & 63;
            u64 bits = s_d[w0 >> 6] >> off;
            if (off) bits |= s_d[(w0 >> 6) + 1] << (64 - off);
            ok = (bits & ((1ull << (want - 1)) - 1ull)) != 0;             // left-maximal: what thins the candidates out
            if (ok) {
                // value of the interval [lb, u] = min LCP[lb+1 .. u]; it is an interval iff LCP[lb] < value > LCP[u+1]
                for (int k = 1; k < want - 1; k++) { const u32 v = s_lcp[me - k]; l = v < l ? v : l; }
                ok = l > nxt && l >= lmin && s_lcp[w0] < l;
            }
            if (ok) {
                const u32 at = atomicAdd(&s_nlist, 1u);
                if (at < MP_LIST) { s_list[at][0] = (u32)me | ((u32)want << 16); s_list[at][1] = (u32)lo2; s_list[at][2] = l; listed = true; }
            }
        }
        if (ok && !listed) {       // the general form (more than MP_HALO or 64 samples, or a full list): everything from global memory
            const int64_t lb = u - want + 1;
            const int sub = s0 + lo2;
            l = cur;
            for (int64_t j = lb + 1; j < u; j++) { const u32 v = (u32)LCP[j]; l = v < l ? v : l; }
            ok = l > nxt && l >= lmin && (u32)LCP[lb] < l && ismultimum_dev(SA, BWT, sep, nsamples, lb, u);
            if (ok) {
                sa_t mn = SA[lb];
                for (int64_t j = lb + 1; j <= u; j++) { const sa_t v = SA[j]; mn = v < mn ? v : mn; }
                const unsigned long long key = ((unsigned long long)l << 32) | (unsigned long long)(0xFFFFFFFFu - (u32)mn);
                if (key > __atomic_load_n(&best[sub], __ATOMIC_RELAXED)) {
                    atomicMax(&best[sub], key);
                    const u32 reg = blockIdx.x % RV_MULTI_REGIONS, rcap = cand_cap / RV_MULTI_REGIONS;
                    const u32 q = atomicAdd(&cand_count[reg * 64], 1u);
                    if (q < rcap) { RvMultiCand c; c.ub = (u32)u; c.sub = (u32)sub; c.key = key; cand[(size_t)reg * rcap + q] = c; }
                }
            }
        }
    }
    __syncthreads();
    // Phase 2: the members of every listed rank in one round trip (independent, predicated loads): their samples must all
    // differ (reveal.c:231-244); the smallest position goes into the key.
    const u32 nlist = s_nlist < (u32)MP_LIST ? s_nlist : (u32)MP_LIST;
    for (u32 base = 0; base < nlist; base += TB) {
        const u32 t = base + threadIdx.x;
        bool ok = t < nlist;
        unsigned long long key = 0; int64_t u = 0; int sub = 0;
        if (ok) {
            const u32 e0 = s_list[t][0];
            const int me = (int)(e0 & 0xffffu), want = (int)(e0 >> 16);
            const u32 l = s_list[t][2];
            sub = s0 + (int)s_list[t][1];
            u = u0 + (me - MP_HALO);
            const int64_t lb = u - want + 1;
            sa_t sv[MP_HALO];
#pragma unroll
            for (int k = 0; k < MP_HALO; k++) sv[k] = SA[k < want ? lb + k : u];
            u64 seen = 0; bool distinct = true;
            sa_t mn = sv[0];
#pragma unroll
            for (int k = 0; k < MP_HALO; k++) {
                if (k < want) {
                    const u64 bit = 1ull << sample_of_pos(sep, nsamples - 1, sv[k]);
                    distinct &= !(seen & bit); seen |= bit;
                    mn = sv[k] < mn ? sv[k] : mn;
                }
            }
            key = ((unsigned long long)l << 32) | (unsigned long long)(0xFFFFFFFFu - (u32)mn);
            // Only a rank that would raise the maximum of its sub-index goes to the atomic unit, and only such a rank can be
            // the winner pass 2 looks for (the maximum never falls): the others are not even listed.  At the top levels every
            // candidate of the level aims at the same few words, and one list counter took an atomic from every other wave
            // (160 K atomics on one address: 0.9 ms per level, whatever else the kernel did).
            ok = distinct && key > __atomic_load_n(&best[sub], __ATOMIC_RELAXED);
            if (ok) atomicMax(&best[sub], key);
        }
        // the list is kept in RV_MULTI_REGIONS regions with a counter each (256 B apart: different L2 channels), chosen by block
        const u64 bal = __ballot(ok);
        if (bal) {
            const u32 reg = blockIdx.x % RV_MULTI_REGIONS, rcap = cand_cap / RV_MULTI_REGIONS;
            u32 qb = 0;
            if (lane == 0) qb = atomicAdd(&cand_count[reg * 64], (u32)__popcll(bal));
            qb = __shfl(qb, 0, 64);
            if (ok) {
                const u32 q = qb + (u32)__popcll(bal & ((1ull << lane) - 1ull));
                if (q < rcap) { RvMultiCand c; c.ub = (u32)u; c.sub = (u32)sub; c.key = key; cand[(size_t)reg * rcap + q] = c; }
            }
        }
    }
}
__global__ __launch_bounds__(TB) void k_multi_pick2(const sa_t *__restrict__ SA, const int *__restrict__ sub_want, int W,
                                                    const unsigned long long *__restrict__ best, const RvMultiCand *__restrict__ cand, u32 cand_cap,
                                                    const u32 *__restrict__ cand_count, u32 *__restrict__ pick_l, sa_t *__restrict__ pick_pos) {
    // grid = RV_MULTI_REGIONS x 4 blocks: four blocks per region of the list
    const u32 reg = blockIdx.x / 4, part = blockIdx.x % 4, rcap = cand_cap / RV_MULTI_REGIONS;
    const u32 cnt = cand_count[reg * 64];
    if (part == 0 && threadIdx.x == 0) atomicMax(const_cast<u32 *>(&cand_count[RV_MULTI_REGIONS * 64]), cnt);      // what the host sizes the list by
    const u32 total = cnt < rcap ? cnt : rcap;
    for (u32 q = part * TB + threadIdx.x; q < total; q += 4 * TB) {
        const RvMultiCand c = cand[(size_t)reg * rcap + q];
        if (best[c.sub] != c.key) continue;
        const int want = sub_want[c.sub];
        const int64_t lb = (int64_t)c.ub - want + 1;
        pick_l[c.sub] = (u32)(c.key >> 32);
        for (int k = 0; k < want; k++) pick_pos[(size_t)c.sub * W + k] = SA[lb + k];      // members in SA order, as the reference emits them
    }
}


// ---- full matches by streaming (more than two samples) -------------------------------------
// A FULL match of a sub-index with `want` samples is the LCP interval of exactly `want` ranks [lb, u], lb = u - want + 1:
// value l = min LCP[lb+1 .. u] >= minl, LCP[lb] < l > LCP[u+1], its members' samples all different (reveal.c:231-244) and
// left-maximal (reveal.c:246-256).  This is what the built-in picker takes (schemes.py:227) and what the anchor cascade
// lists at the root, so both scans run on the skeleton below -- the pair scan's: a wave streams 1024 ranks, SIXTEEN per
// lane (non-temporal 16-byte loads: four of LCP, one of BWT), and tests them branch-free on two 32-rank bit windows
// (my sixteen ranks and the sixteen in front of them, which the lane below hands up by shuffle -- lane 0 gets them from
// the wave's halo):
//   G bit j: LCP[j] >= minl                                   -- a full match ending at u needs G set on u-want+2 .. u
//   D bit j: ranks j-1, j are evidence of left-maximality      -- ... and some D set on the same ranks
// together with LCP[u] > LCP[u+1].  "All of the H = want - 1 bits below and at u" is one AND-doubling ladder for all
// sixteen ranks at once (A2 = A1 & A1 << 1, A4 = A2 & A2 << 2, A8 ..., combined by the bits of H); D's bits come from
// the BWT words four characters at a time (byte-parallel compares; every character is ASCII, bit 7 is the carry room).
// About one rank in seven hundred passes at 10 x 5 Mbp.  Those are taken one at a time by the WHOLE wave: lane j reads
// LCP[lb + j] and SA[lb + j] -- lines the wave has just streamed --, the value is a wave minimum, the samples a count of
// the separators below each member (uniform addresses: scalar loads), their census a wave OR.  No LDS, no workgroup
// barrier, no sample array: a wave retires when its own ranks are done.
// The kernel is bound by instruction issue, not by memory, and was written against that: the first form of this skeleton
// (8 ranks per lane, a compare chain per rank and per window, a passing rank's sixteen members walked by its own lane)
// issued ~1 000 vector instructions per wave and took 200 us at 10 x 5 Mbp; before that a workgroup staged 2048 ranks in
// LDS and walked the candidates -- one rank in ten -- in two dense stages between seven barriers: 130 us.
//
// MODE 0 (cascade root, rv_cascade_multi.hip): want = k everywhere; survivors are listed (length, the k positions by sample).
// MODE 1 (built-in picker): want = the sample count of the rank's sub-index, looked up per lane (a lane whose sixteen ranks
//         straddle two sub-indices, or whose sub-index has more than FS_HALO samples, takes the general path rank by rank);
//         survivors raise the maximum of their sub-index and are listed for k_multi_pick2.
constexpr int FS_ITEMS = 16, FS_WTILE = 64 * FS_ITEMS, FS_TILE = (TB / 64) * FS_WTILE, FS_HALO = 16, FS_SEPS = 15;
static_assert(FS_WTILE * 2 == RV_TSUB_TILE, "a wave covers half a tile of the host's tile -> sub-index table");

__device__ inline bool fs_evidence(u32 ca, u32 cb) {      // reveal.c:246-256 on the characters in front of two neighbouring ranks ('$' where SA == 0)
    return (cb == '$') | (ca != cb) | (ca == 'N') | (ca == '$') | ((ca - 'a') < 26u);
}
// the same for four ranks at once: x = their characters (7 bits each, one per byte), y = the characters of the ranks in front of each.
// -> bit 7 of byte i set iff fs_evidence(y_i, x_i).  (v + 0x7f..: bit 7 of a byte set iff the byte is not zero; no carry leaves a 7-bit byte.)
__device__ inline u32 fs_evidence4(u32 x, u32 y) {
    const u32 K = 0x7f7f7f7fu;
    const u32 ne = (x ^ y) + K;                                   // ca != cb
    const u32 eq = ((x ^ 0x24242424u) + K) & ((y ^ 0x4e4e4e4eu) + K) & ((y ^ 0x24242424u) + K);      // none of cb == '$', ca == 'N', ca == '$'
    const u32 low = (y + 0x1f1f1f1fu) & ~(y + 0x05050505u);       // 'a' <= ca <= 'z'
    return (ne | ~eq | low) & 0x80808080u;
}
__device__ inline u32 fs_bits4(u32 e) {      // bit 7 of the four bytes -> bits 0 .. 3
    u32 v = (e >> 7) & 0x01010101u;
    v |= v >> 7;
    v |= v >> 14;
    return v & 0xfu;
}
// bit b of the result: bits b-H+1 .. b of g are all set (H = 0: every bit set).  H < 16.
__device__ inline u32 fs_run(u32 g, int H) {
    const u32 a1 = g, a2 = a1 & (a1 << 1), a4 = a2 & (a2 << 2), a8 = a4 & (a4 << 4);
    u32 acc = 0xFFFFFFFFu; int off = 0;
    if (H & 8) { acc &= a8; off = 8; }
    if (H & 4) { acc &= a4 << off; off += 4; }
    if (H & 2) { acc &= a2 << off; off += 2; }
    if (H & 1) acc &= a1 << off;
    return acc;
}
// minimum / OR over the 16-lane row a lane belongs to (every lane of the row gets it): the four in-row steps of rv_wave_min_u32
__device__ inline u32 rv_row_min_u32(u32 v) {
    int x = (int)v, y;
#define RV_DPP_ROW_(ctrl) y = __builtin_amdgcn_update_dpp(-1, x, ctrl, 0xf, 0xf, false); x = (int)(((u32)y < (u32)x) ? (u32)y : (u32)x);
    RV_DPP_ROW_(0xB1) RV_DPP_ROW_(0x4E) RV_DPP_ROW_(0x141) RV_DPP_ROW_(0x140)      // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
#undef RV_DPP_ROW_
    return (u32)x;
}
__device__ inline u32 rv_row_or_u32(u32 v) {
    int x = (int)v;
#define RV_DPP_ROW_(ctrl) x |= __builtin_amdgcn_update_dpp(0, x, ctrl, 0xf, 0xf, false);
    RV_DPP_ROW_(0xB1) RV_DPP_ROW_(0x4E) RV_DPP_ROW_(0x141) RV_DPP_ROW_(0x140)
#undef RV_DPP_ROW_
    return (u32)x;
}
__device__ inline u32 rv_wave_or_u32(u32 v) {      // every lane gets the OR over the wave
#define RV_STEP_(CTRL, RM, TAKE) v |= rv_dpp_u32<CTRL, RM>(v);
    RV_WAVE_SCAN_STEPS(RV_STEP_)        // (a lane without a source lane sees 0)
#undef RV_STEP_
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

template <int MODE>
__global__ __launch_bounds__(TB) void k_full_scan(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, const uint8_t *__restrict__ BWT, int64_t m,
                                                  const sa_t *__restrict__ nsep, int nsamples, u32 minl, int minn,
                                                  u32 *__restrict__ c_len, sa_t *__restrict__ c_pos, u32 rcap, u32 *__restrict__ region_cnt, int nregions, int cnt_stride,
                                                  const int64_t *__restrict__ sub_start, const int *__restrict__ sub_want, int nsubs, const int *__restrict__ tile_sub,
                                                  unsigned long long *__restrict__ best, RvMultiCand *__restrict__ cand, u32 cand_cap, u32 *__restrict__ cand_count) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wfirst = (int64_t)blockIdx.x * FS_TILE + (int64_t)w * FS_WTILE;
    if (wfirst >= m) return;
    const int64_t i0 = wfirst + (int64_t)lane * FS_ITEMS;
    // halo: the FS_HALO ranks in front of the wave (a rank each for lanes 0 .. 15), the LCP value behind its last rank; issued first
    u32 h_lc = 0, h_bw = 0, h_nlc = 0;
    {
        const int64_t j = wfirst - FS_HALO + lane;
        if (lane < FS_HALO && j >= 0) { h_lc = (u32)LCP[j]; h_bw = (u32)BWT[j]; }      // (the side bit is masked where the byte is used: masking here made the wave wait for this load before it issued the streaming ones)
        if (wfirst + FS_WTILE < m) h_nlc = (u32)LCP[wfirst + FS_WTILE];
    }
    // the first separators, in scalar registers before anything waits for them (the separator array is a DBuf: its allocation reaches 256 bytes beyond
    // the last separator, so these loads stay inside it whatever the sample count; what lies behind the last separator is masked where it is used)
    sa_t sep[FS_SEPS];
#pragma unroll
    for (int q = 0; q < FS_SEPS; q++) sep[q] = nsep[q];
    // the sub-index of the lane's first rank and its sample count (MODE 1)
    int want = nsamples, mine = 0, s_last = 0;
    bool slow = false;
    if (MODE == 1) {
        const int64_t tt = wfirst / RV_TSUB_TILE, ntsub = (m + RV_TSUB_TILE - 1) / RV_TSUB_TILE;
        const int s0 = tile_sub[tt];
        s_last = tt + 1 < ntsub ? tile_sub[tt + 1] : nsubs - 1;
        mine = s0;
        if (s_last > s0) {      // (several sub-indices around this wave's ranks: the largest s with sub_start[s] <= i0)
            int lo = s0, hi = s_last;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sub_start[mid] <= i0) lo = mid; else hi = mid - 1; }
            mine = lo;
            const int64_t nstart = sub_start[mine + 1];      // (sub_start[nsubs] = m)
            slow = nstart < i0 + FS_ITEMS;
        }
        want = sub_want[mine];
        if (want > FS_HALO) slow = true;
        if (!slow && !(want >= minn && want >= 2 && want <= nsamples)) want = 1;      // (no match can have this size: the window test below fails everywhere)
    }
    __shared__ uint4 s_corner[TB / 64][64 * LT_ROW];
    u32 lc[FS_ITEMS + 1], bx[FS_ITEMS / 4];
    if (wfirst + FS_WTILE <= m) {      // (the whole wave inside the arrays)
        const v4u32 bb = __builtin_nontemporal_load(reinterpret_cast<const v4u32 *>(BWT + i0));
        bx[0] = bb.x; bx[1] = bb.y; bx[2] = bb.z; bx[3] = bb.w;
        wave_lcp16(LCP, wfirst, lane, s_corner[w], lc);
    } else {
#pragma unroll
        for (int k = 0; k < FS_ITEMS / 4; k++) bx[k] = 0;
#pragma unroll
        for (int k = 0; k < FS_ITEMS; k++) {
            lc[k] = (i0 + k < m) ? (u32)LCP[i0 + k] : 0u;
            bx[k >> 2] |= ((i0 + k < m) ? (u32)BWT[i0 + k] : 0u) << (8 * (k & 3));
        }
    }
    asm volatile("" : "+v"(h_bw), "+v"(h_lc));      // (the halo's values are first looked at here, behind the streaming loads: see above)
#pragma unroll
    for (int k = 0; k < FS_ITEMS / 4; k++) bx[k] &= 0x7f7f7f7fu;      // RV_BWT_CHAR: the side bit off
    h_bw &= RV_BWT_CHAR;
    lc[FS_ITEMS] = (u32)__shfl_down((int)lc[0], 1, 64);
    if (lane == 63) lc[FS_ITEMS] = h_nlc;
    // per-rank bits of my sixteen ranks: G, closes an interval (c), D
    u32 c16 = 0, g16 = 0;
#pragma unroll
    for (int k = 0; k < FS_ITEMS; k++) {
        const bool g = lc[k] >= minl;
        g16 |= (u32)g << k;
        c16 |= (u32)(g & (lc[k] > lc[k + 1])) << k;
    }
    u32 plast = (u32)__shfl_up((int)bx[FS_ITEMS / 4 - 1], 1, 64);      // the four characters in front of my first rank (the last one matters)
    const u32 hb15 = (u32)__builtin_amdgcn_readlane((int)h_bw, FS_HALO - 1);
    if (lane == 0) plast = hb15 << 24;
    u32 d16 = 0;
#pragma unroll
    for (int k = 0; k < FS_ITEMS / 4; k++) {
        const u32 y = __builtin_amdgcn_alignbit(bx[k], k ? bx[k - 1] : plast, 24);      // byte i = the character in front of rank 4k + i's
        d16 |= fs_bits4(fs_evidence4(bx[k], y)) << (4 * k);
    }
    // the same bits of the 16 ranks in front of my first one
    u32 pw = (u32)__shfl_up((int)(g16 | (d16 << 16)), 1, 64);
    {
        const u32 hprev = (u32)__shfl_up((int)h_bw, 1, 64);
        const u64 HG = __ballot((lane < FS_HALO) & (h_lc >= minl));
        const u64 HD = __ballot((lane < FS_HALO) & (lane > 0) & fs_evidence(hprev, h_bw));
        if (lane == 0) pw = ((u32)HG & 0xffffu) | (((u32)HD & 0xffffu) << 16);
    }
    const u32 G32 = (pw & 0xffffu) | (g16 << 16);
    const u32 D32 = (pw >> 16) | (d16 << 16);
    u32 todo = c16;
    if (!slow) {
        const int H = want - 1;                      // 0 .. FS_HALO - 1
        todo = (((c16 << 16) & fs_run(G32, H) & ~fs_run(~D32, H)) >> 16);      // (H = 0: the second term is everything, its complement nothing)
    }
    const u32 reg = MODE == 0 ? (blockIdx.x & (u32)(nregions - 1)) : (blockIdx.x % RV_MULTI_REGIONS);
    // The ranks that passed, FOUR at a time: every 16-lane row of the wave takes one -- lane j of the row reads LCP[lb + j] and SA[lb + j], the value is
    // a row minimum, the samples' census a row OR (DPP row operations) -- so a wave's passing ranks cost it one memory round trip and one returning atomic
    // per four instead of per rank.  (One at a time by the whole wave: 64 us at 10 x 5 Mbp, and ~11 us for a frontier of a few thousand ranks whose
    // handful of waves had nobody to hide behind.)
    const int row = lane >> 4, j = lane & 15;
    u64 bal = __ballot(todo != 0u);
    while (bal) {
        u64 bb = bal;
        const int s0 = (int)__builtin_ctzll(bb); bb &= bb - 1;
        const int s1 = bb ? (int)__builtin_ctzll(bb) : -1; bb &= bb ? bb - 1 : 0;
        const int s2 = bb ? (int)__builtin_ctzll(bb) : -1; bb &= bb ? bb - 1 : 0;
        const int s3 = bb ? (int)__builtin_ctzll(bb) : -1;
        const int my = row == 0 ? s0 : row == 1 ? s1 : row == 2 ? s2 : s3;      // the lane whose rank my row takes
        const bool valid = my >= 0;
        const int from = valid ? my : 0;
        const u32 td = (u32)__shfl((int)todo, from, 64);
        const int kk = __builtin_ctz(td | (valid ? 0u : 1u));
        const int64_t u = wfirst + (int64_t)from * FS_ITEMS + kk;
        const int wn = MODE == 0 ? nsamples : __shfl(want, from, 64);
        const int sub = MODE == 0 ? 0 : __shfl(mine, from, 64);
        const bool slw = MODE == 1 && __shfl((int)slow, from, 64) != 0;
        const bool fast = valid && !slw;
        const int64_t lb = u - wn + 1;             // >= 0: G is clear at rank 0 and at every sub-index' first rank
        const int64_t at = lb + j;
        const bool mem = fast && j < wn;           // lane j of the row: member j of the interval
        const u32 lv = mem ? (u32)LCP[at] : 0xFFFFFFFFu;
        const u32 nxt = (fast && u + 1 < m) ? (u32)LCP[u + 1] : 0u;
        const sa_t sv = mem ? SA[at] : (sa_t)0;
        const u32 l = rv_row_min_u32(j >= 1 ? lv : 0xFFFFFFFFu);
        const u32 below = (u32)__shfl((int)lv, lane & ~15, 64);
        const bool ok = fast & (l >= minl) & (l > nxt) & (below < l);
        // the members' samples (interface.c:116-134: the separators in front of a position): one pass over the separators, uniform addresses
        int sm = 0;
#pragma unroll
        for (int q = 0; q < FS_SEPS; q++) sm += ((q < nsamples - 1) & (sep[q] < sv)) ? 1 : 0;
        for (int q = FS_SEPS; q < nsamples - 1; q++) sm += nsep[q] < sv ? 1 : 0;
        const u32 census_lo = rv_row_or_u32((mem && sm < 32) ? (1u << sm) : 0u);
        const u32 census_hi = nsamples > 32 ? rv_row_or_u32((mem && sm >= 32) ? (1u << (sm - 32)) : 0u) : 0u;
        const bool good = ok && __popc(census_lo) + __popc(census_hi) == wn;      // all different (reveal.c:231-244)
        if (MODE == 0) {
            const u64 gb = __ballot(good && j == 0);
            if (gb) {
                // (the regions' counters stand cnt_stride words apart: side by side in one cache line their returning atomics -- one per listed
                //  match then, ~71 000 at 10 x 5 Mbp -- queued up in ONE L2 channel at ~3.7 ns each, and the kernel's time was their number, 268 us)
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&region_cnt[(size_t)reg * cnt_stride], (u32)__popcll(gb));
                base = (u32)__builtin_amdgcn_readfirstlane((int)base);
                const u32 i = base + (u32)__popcll(gb & ((1ull << (row * 16)) - 1ull));
                if (good && i < rcap) {
                    const size_t o = (size_t)reg * rcap + i;
                    if (j == 0) c_len[o] = l;
                    if (mem) c_pos[o * (size_t)nsamples + (size_t)sm] = sv;
                }
            }
        } else {
            // Only a rank that would raise the maximum of its sub-index goes to the atomic unit, and only such a rank can be the winner
            // k_multi_pick2 looks for (the maximum never falls): the others are not even listed.
            const u32 mn = rv_row_min_u32(mem ? (u32)sv : 0xFFFFFFFFu);
            const unsigned long long key = ((unsigned long long)l << 32) | (unsigned long long)(0xFFFFFFFFu - mn);
            if (good && j == 0 && key > __atomic_load_n(&best[sub], __ATOMIC_RELAXED)) {
                atomicMax(&best[sub], key);
                const u32 rc = cand_cap / RV_MULTI_REGIONS;
                const u32 q = atomicAdd(&cand_count[reg * 64], 1u);
                if (q < rc) { RvMultiCand c; c.ub = (u32)u; c.sub = (u32)sub; c.key = key; cand[(size_t)reg * rc + q] = c; }
            }
            const int s_last_src = __shfl(s_last, from, 64);
            if (valid && slw && j == 0) {      // the general form, everything from global memory (reveal.c:227-259 by ismultimum_dev), by one lane of the row
                int lo = sub, hi = s_last_src;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sub_start[mid] <= u) lo = mid; else hi = mid - 1; }
                const int sub2 = lo, wn2 = sub_want[sub2];
                const int64_t lb2 = u - wn2 + 1;
                if (wn2 >= minn && wn2 >= 2 && wn2 <= nsamples && lb2 >= sub_start[sub2]) {
                    const u32 nxt2 = u + 1 < m ? (u32)LCP[u + 1] : 0u;
                    u32 l2 = (u32)LCP[u];
                    for (int64_t x = lb2 + 1; x < u; x++) { const u32 v = (u32)LCP[x]; l2 = v < l2 ? v : l2; }
                    if (l2 > nxt2 && l2 >= minl && (u32)LCP[lb2] < l2 && ismultimum_dev(SA, BWT, nsep, nsamples, lb2, u)) {
                        sa_t mn2 = SA[lb2];
                        for (int64_t x = lb2 + 1; x <= u; x++) { const sa_t v = SA[x]; mn2 = v < mn2 ? v : mn2; }
                        const unsigned long long key2 = ((unsigned long long)l2 << 32) | (unsigned long long)(0xFFFFFFFFu - (u32)mn2);
                        if (key2 > __atomic_load_n(&best[sub2], __ATOMIC_RELAXED)) {
                            atomicMax(&best[sub2], key2);
                            const u32 rc = cand_cap / RV_MULTI_REGIONS;
                            const u32 q = atomicAdd(&cand_count[reg * 64], 1u);
                            if (q < rc) { RvMultiCand c; c.ub = (u32)u; c.sub = (u32)sub2; c.key = key2; cand[(size_t)reg * rc + q] = c; }
                        }
                    }
                }
            }
        }
        if (lane == s0 || lane == s1 || lane == s2 || lane == s3) todo &= todo - 1;
        bal = __ballot(todo != 0u);
    }
}

// best / pick_l / the list's counters back to zero in one launch (three memsets were three launches inside the level's scan)
__global__ __launch_bounds__(TB) void k_mp_zero(unsigned long long *__restrict__ best, u32 *__restrict__ pick_l, int nsubs, u32 *__restrict__ cand_count) {
    const int64_t id = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (id < nsubs) { best[id] = 0; pick_l[id] = 0; }
    if (id < RV_MULTI_REGIONS * 64 + 1) cand_count[id] = 0;
}

// (ev_start / ev_stop given: the events ride on the streaming kernel's own dispatch packet, as for the pair scan)
#define RV_FS_LAUNCH(MODE, GRID, ...)                                                                                              \
    do {                                                                                                                           \
        if (ev_start && ev_stop) hipExtLaunchKernelGGL(k_full_scan<MODE>, dim3(GRID), dim3(TB), 0, ws.stream, ev_start, ev_stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(k_full_scan<MODE>, dim3(GRID), dim3(TB), 0, ws.stream, __VA_ARGS__);                               \
    } while (0)

int rv_full_list_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t n, const sa_t *nsep, int k, u32 minl,
                        u32 *c_len, sa_t *c_pos, u32 rcap, u32 *region_cnt, int nregions, int cnt_stride, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (n <= 0) return 0;
    if (k < 2 || k > FS_HALO || (nregions & (nregions - 1))) { rv_set_error("full-match scan: sample count or region count out of range"); return -1; }
    RV_FS_LAUNCH(0, (unsigned)ceil_div(n, FS_TILE), SA, LCP, BWT, n, nsep, k, minl, 2,
                 c_len, c_pos, rcap, region_cnt, nregions, cnt_stride, (const int64_t *)nullptr, (const int *)nullptr, 0, (const int *)nullptr,
                 (unsigned long long *)nullptr, (RvMultiCand *)nullptr, 0u, (u32 *)nullptr);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_multi_pick_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, int64_t m, const uint8_t *BWT, const sa_t *nsep, int nsamples, int minl, int minn,
                         const int64_t *sub_start, const int *sub_want, int nsubs, const int *tile_sub, unsigned long long *best, u32 *pick_l, sa_t *pick_pos,
                         RvMultiCand *cand, u32 cand_cap, u32 *cand_count, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (m <= 0 || nsubs <= 0) return 0;
    hipLaunchKernelGGL(k_mp_zero, dim3((unsigned)ceil_div(std::max<int64_t>(nsubs, RV_MULTI_REGIONS * 64 + 1), TB)), dim3(TB), 0, ws.stream, best, pick_l, nsubs, cand_count);
    RV_LAUNCH_CHECK();
    if (nsamples <= 64 && !ws.opt.scan_v1)
        RV_FS_LAUNCH(1, (unsigned)ceil_div(m, FS_TILE), SA, LCP, BWT, m, nsep, nsamples, (u32)(minl > 1 ? minl : 1), minn,
                     (u32 *)nullptr, (sa_t *)nullptr, 0u, (u32 *)nullptr, 1, 1, sub_start, sub_want, nsubs, tile_sub, best, cand, cand_cap, cand_count);
    else if (ev_start && ev_stop)      // (more than 64 samples: the samples of a match no longer fit a 64-bit census; RV_SCAN_V1: the staged kernel, for comparison)
        hipExtLaunchKernelGGL(k_multi_pick1, dim3((unsigned)ceil_div(m, RV_TSUB_TILE)), dim3(TB), 0, ws.stream, ev_start, ev_stop, 0, SA, LCP, m, BWT, nsep, nsamples, minl, minn,
                              sub_start, sub_want, nsubs, tile_sub, best, pick_l, cand, cand_cap, cand_count);
    else
        hipLaunchKernelGGL(k_multi_pick1, dim3((unsigned)ceil_div(m, RV_TSUB_TILE)), dim3(TB), 0, ws.stream, SA, LCP, m, BWT, nsep, nsamples, minl, minn,
                           sub_start, sub_want, nsubs, tile_sub, best, pick_l, cand, cand_cap, cand_count);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_multi_pick2, dim3(RV_MULTI_REGIONS * 4), dim3(TB), 0, ws.stream, SA, sub_want, nsamples, (const unsigned long long *)best,
                       (const RvMultiCand *)cand, cand_cap, (const u32 *)cand_count, pick_l, pick_pos);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_scan_multi_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, int64_t m, const uint8_t *BWT, const sa_t *nsep, int nsamples,
                         int minl, int minn, RvMultiRec *rec, uint16_t *so, sa_t *pos, u32 rec_cap, u32 mem_cap, u32 *counters, uint4 *tiletab,
                         const int64_t *sub_start, const int *sub_want, int nsubs) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_scan_multi, dim3((unsigned)ceil_div(m, (int64_t)RV_MULTI_TILE * (TB / 64))), dim3(TB), 0, ws.stream, SA, LCP, m, BWT, nsep, nsamples, minl, minn,
                       rec, so, pos, rec_cap, mem_cap, counters, tiletab, sub_start, sub_want, nsubs);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_scan_pair_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, int64_t m, const uint8_t *BWT, sa_t nsep0, int minl,
                        RvPairRec *slots, RvPairRec *ovf, u32 ovf_cap, u32 *ovf_counter, u32 *tilecnt, u32 *tileovf,
                        unsigned long long *best, RvPairRec *picks, int nsubs, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (m <= 0) return 0;
    const int64_t nb = ceil_div(m, PAIR_TILE);
    if (ev_start && ev_stop)      // timed launch: the events ride on the kernel's own dispatch packet (no marker packets around it)
        hipExtLaunchKernelGGL(k_scan_pair, dim3((unsigned)nb), dim3(TB), 0, ws.stream, ev_start, ev_stop, 0, SA, LCP, m, BWT, nsep0, minl, slots, ovf, ovf_cap, ovf_counter,
                              tilecnt, tileovf, best, picks, nsubs);
    else
        hipLaunchKernelGGL(k_scan_pair, dim3((unsigned)nb), dim3(TB), 0, ws.stream, SA, LCP, m, BWT, nsep0, minl, slots, ovf, ovf_cap, ovf_counter, tilecnt, tileovf,
                           best, picks, nsubs);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_pick_slots_launch(Workspace &ws, const RvPairRec *slots, const RvPairRec *ovf, u32 ovf_cap, const u32 *tilecnt, const u32 *tileovf, int64_t ntile,
                          const int64_t *sub_start, int nsubs, unsigned long long *best, RvPairRec *picks, u32 *ovf_counter, const u32 *err,
                          const int *tile_sub, int64_t ntsub) {
    if (ntile <= 0 || nsubs <= 0) return 0;
    const unsigned g = (unsigned)ceil_div(ntile * RV_PAIR_SLOTS, TB);
    hipLaunchKernelGGL(k_pick_slots<1>, dim3(g), dim3(TB), 0, ws.stream, slots, ovf, ovf_cap, tilecnt, tileovf, ntile, sub_start, nsubs, best, picks, ovf_counter, err, tile_sub, ntsub);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pick_slots<2>, dim3(g), dim3(TB), 0, ws.stream, slots, ovf, ovf_cap, tilecnt, tileovf, ntile, sub_start, nsubs, best, picks, ovf_counter, err, tile_sub, ntsub);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_pair_compact_launch(Workspace &ws, const RvPairRec *slots, const RvPairRec *ovf, const u32 *tilecnt, const u32 *tileovf,
                           const u32 *tileoff, int64_t ntile, RvPairRec *out, u32 out_cap, u32 *ovf_counter, const u32 *err, u32 ovf_cap) {
    if (ntile <= 0) return 0;
    hipLaunchKernelGGL(k_pair_compact, dim3((unsigned)ceil_div(ntile * RV_PAIR_SLOTS, TB)), dim3(TB), 0, ws.stream, slots, ovf, tilecnt, tileovf,
                       tileoff, ntile, out, out_cap, ovf_counter, err, ovf_cap);
    RV_LAUNCH_CHECK();
    return 0;
}

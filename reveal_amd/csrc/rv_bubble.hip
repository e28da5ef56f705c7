// rv_bubble.hip -- bubble_sort (reveallib/reveal.c:666-727) of one cut of every large
// leading child of a level, as data-parallel kernels instead of a sequential replay.
//
// Closed form of the reference's loop for one cut B over a child (SA, LCP, n):
// the loop `for i in 0..n` visits ranks in ascending order, and visit i only sees
//   SA[i]   -- still the original value (earlier visits write ranks <= i-1), and
//   LCP[i]  -- the original value, possibly lowered by visit i-1 only.
// So with l'_i = LCP[i] as visit i sees it, t_i = B - SA[i]:
//   visit i is a MOVE      iff SA[i] < B and SA[i] + l'_i > B             (reveal.c:686)
//       then l'_{i+1} = min(l'_i, LCP[i+1])                              (reveal.c:704-708)
//   else a TRUNCATION      iff SA[i] < B, SA[i] + LCP[i+1] > B, LCP[i+1] > l'_i
//       then l'_{i+1} = t_i                                              (reveal.c:714-718)
//   else l'_{i+1} = LCP[i+1].
// A chain of these only runs along consecutive ranks that can act at all.
// A move takes its suffix out (merging the two LCP gaps around it into their minimum)
// and re-inserts it immediately in front of the nearest lower rank k that is not itself
// a mover and has l'_k < t_i (rank 0 stops unconditionally), reveal.c:691-699; the mover
// inherits k's gap and k's gap becomes t_i.  Movers landing in front of the same k end
// up ordered by t ascending, each one's gap being the t of the one before it: a later
// mover with a larger t stops at the gap an earlier one left, one with a smaller t
// passes it (its gap >= t) and lands in front.  The landing site of a mover therefore
// does not depend on the other movers, and the final arrangement is
//     for every non-mover k in rank order:  [movers landing at k, by t ascending], k.
// tests/test_cpu_oracle_golden.py::test_bubble_closed_form checks this restatement against
// the oracle's literal loop on random arrays; the GPU parity tests run both paths.
//
// Kernels of one round (all participating children at once):
//   k_bubble_window (rv_split.hip)  ranks that can act, via the windowed SAi
//   k_pb_runs      the l' chains: movers flagged (2) and listed, LCP patched in place
//   (tile bounds)  per 2048-rank tile of the level arrays a lower bound of l' (search accelerator): written by the split's
//                  scatter, lowered by every kernel below that lowers or moves a value; k_pb_tilemin only refreshes them
//                  after a round that ran the sequential kernels
//   k_pb_search    one wave per mover: its landing site
//   k_pb_rank      one workgroup per child: movers sorted by rank and by (site, t) -> final ranks
//   k_pb_copyout   tiles whose ranks change -> scratch (the dead parent-level arrays)
//   k_pb_scatter   non-movers from scratch to their final rank (+ SAi upkeep)
//   k_pb_movers    movers to their final rank, the site's new gap; flags cleared
// Children whose cut has more than RV_PB_CAP candidates keep the sequential kernels.
#include "rv_common.h"
#include "rv_split.h"

namespace {

constexpr int TB = 256;
constexpr int PT = RV_SPLIT_TILE;          // ranks per tile
constexpr u32 INF = 0xFFFFFFFFu;

__device__ inline int upper_idx64(const int64_t *__restrict__ begins, int n, int64_t pos) {   // last idx with begins[idx] <= pos, or -1
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (begins[mid] <= pos) lo = mid + 1; else hi = mid; }
    return lo - 1;
}

// id-th window slot of the round -> (descriptor, slot inside it)
__device__ inline void slot_of(const RvBubbleArgs &b, int first, int count, int64_t id, int *dd, int64_t *slot) {
    const int64_t g = id + b.woff[first];
    const int d = first + upper_idx64(b.woff + first, count, g);
    *dd = d; *slot = g - b.woff[d];
}
// id-th tile of the round -> (descriptor, tile inside it)
__device__ inline void tile_of(const RvBubbleArgs &b, int first, int count, int64_t id, int *dd, int64_t *ti) {
    const int64_t g = id + b.par.toff[first];
    const int d = first + upper_idx64(b.par.toff + first, count, g);
    *dd = d; *ti = g - b.par.toff[d];
}

__device__ inline bool par_desc(const RvBubbleArgs &b, int dd) {      // does the parallel path own this (child, cut)?
    const u32 c = b.cnt[dd];
    return c > 0 && c <= (u32)RV_PB_CAP;
}

// ---- the l' chains -----------------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_pb_runs(RvBubbleArgs b, int first, int count, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (id >= total) return;
    int dd; int64_t slot;
    slot_of(b, first, count, id, &dd, &slot);
    if (!par_desc(b, dd) || slot >= (int64_t)b.cnt[dd]) return;
    const RvBubbleDesc ds = b.desc[dd];
    const sa_t *SA = b.SA + ds.off;
    lcp_t *LCP = b.LCP + ds.off;
    uint8_t *flag = b.flag + ds.off;
    const int64_t n = ds.n, B = ds.B;
    int64_t i = (int64_t)b.list[b.woff[dd] + slot];
    if (i > 0 && flag[i - 1]) return;              // not the first rank of its chain
    int64_t lp = (int64_t)(u32)LCP[i];             // nothing in front of a chain's head acts: l' = LCP
    for (;;) {
        const int64_t s = (int64_t)SA[i];
        const bool more = i + 1 < n;
        const int64_t ln = more ? (int64_t)(u32)LCP[i + 1] : 0;
        int64_t nlp = ln;
        if (s < B && s + lp > B) {
            if (i == 0) {
                nlp = B - s;                       // x == 0: nothing moves; LCP[1] = t, and min(tmpLCP, t) == t (reveal.c:700-708)
            } else {
                flag[i] = 2;
                const u32 q = atomicAdd(&b.par.mcnt[dd], 1u);
                b.par.mrank[b.woff[dd] + q] = (u32)i;
                b.par.glist[atomicAdd(b.par.gcount, 1u)] = ((u64)(u32)dd << 32) | q;
                nlp = lp < ln ? lp : ln;
            }
            if (more) { LCP[i + 1] = (lcp_t)nlp; atomicMin(&b.par.tmin[(ds.off + i + 1) >> 11], (u32)nlp); }
        } else if (more && s < B && s + ln > B && ln > lp) {
            nlp = B - s;
            LCP[i + 1] = (lcp_t)nlp;
            atomicMin(&b.par.tmin[(ds.off + i + 1) >> 11], (u32)nlp);
        }
        if (!more || !flag[i + 1]) break;
        lp = nlp; i++;
    }
}

// ---- refresh of the tile bounds ---------------------------------------------------------------
// The bounds come from the split and are kept by every kernel of these rounds; the sequential kernels (a cut with more than
// RV_PB_CAP candidates) do not keep them.  After such a round the values in place are folded into the bounds again.
__global__ __launch_bounds__(TB) void k_pb_tilemin(RvBubbleArgs b, int first, int count) {
    int dd; int64_t ti;
    tile_of(b, first, count, blockIdx.x, &dd, &ti);
    const RvBubbleDesc ds = b.desc[dd];
    const lcp_t *LCP = b.LCP + ds.off;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < PT / TB; k++) {
        const int64_t r = ti * PT + (int64_t)k * TB + threadIdx.x;
        const bool in = r < ds.n;
        const u32 key = in ? (u32)((ds.off + r) >> 11) : 0xFFFFFFFFu;
        const u32 val = in ? (r == 0 ? 0u : (u32)LCP[r]) : INF;
        u64 todo = __ballot(in);
        while (todo) {                                   // (a wave's 64 ranks lie in one or two global tiles)
            const int l0 = (int)__builtin_ctzll(todo);
            const u32 k0 = (u32)__builtin_amdgcn_readlane((int)key, l0);
            const bool mine = in && key == k0;
            const u32 v = rv_wave_min_u32(mine ? val : INF);
            if (lane == l0) atomicMin(&b.par.tmin[k0], v);
            todo &= ~__ballot(mine);
        }
    }
}

// ---- landing sites --------------------------------------------------------------------------
// largest rank r in [lo, hi] that stops a mover with threshold t (r == 0, or a non-mover with l' < t); -1 if none.
// The first 256 ranks decide almost every move; after that the scan goes 1024 ranks per step with all of a
// step's loads in flight before the first ballot (one memory round trip per step instead of four).
__device__ inline int64_t wave_scan_down(const lcp_t *__restrict__ LCP, const uint8_t *__restrict__ flag, int64_t hi, int64_t lo, int64_t t) {
    const int lane = threadIdx.x & 63;
    {
        u64 bal[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t r = hi - 64 * k - lane;
            const bool hit = r >= lo && (r == 0 || (flag[r] != 2 && (int64_t)(u32)LCP[r] < t));
            bal[k] = __ballot(hit);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) if (bal[k]) return hi - 64 * k - (int64_t)__builtin_ctzll(bal[k]);
    }
    for (int64_t top = hi - 256; top >= lo; top -= 1024) {
        u32 v[16]; uint8_t f[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int64_t r = top - 64 * k - lane;
            const bool in = r >= lo;
            v[k] = in ? (u32)LCP[r] : INF; f[k] = in ? flag[r] : (uint8_t)2;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int64_t r = top - 64 * k - lane;
            const bool hit = r >= lo && (r == 0 || (f[k] != 2 && (int64_t)v[k] < t));
            const u64 bal = __ballot(hit);
            if (bal) return top - 64 * k - (int64_t)__builtin_ctzll(bal);
        }
    }
    return -1;
}

__global__ __launch_bounds__(TB) void k_pb_search(RvBubbleArgs b) {
    const int lane = threadIdx.x & 63;
    const u32 total = *b.par.gcount;
    const u32 nw = gridDim.x * (TB / 64);
    for (u32 g = blockIdx.x * (TB / 64) + (threadIdx.x >> 6); g < total; g += nw) {      // one wave per mover
        const u64 ent = b.par.glist[g];
        const int dd = (int)(ent >> 32);
        const int64_t slot = (int64_t)(u32)ent;
        const RvBubbleDesc ds = b.desc[dd];
        const lcp_t *LCP = b.LCP + ds.off;
        const uint8_t *flag = b.flag + ds.off;
        const int64_t e = (int64_t)b.par.mrank[b.woff[dd] + slot];
        const int64_t t = ds.B - (int64_t)b.SA[ds.off + e];
        // tiles are those of the level arrays (global rank >> 11); inside the child they start at child rank gt * PT - off
        const int64_t gtile = (ds.off + e) >> 11, gfirst = ds.off >> 11;
        int64_t lo = gtile * PT - ds.off;
        int64_t k = wave_scan_down(LCP, flag, e - 1, lo > 0 ? lo : 0, t);
        int64_t top = gtile - 1;
        while (k < 0) {
            if (top < gfirst) { k = 0; if (lane == 0) atomicOr(b.err, 2u); break; }      // cannot happen: the child's first tile holds its rank 0
            // the nearest lower tile whose bound promises a stopper (the child's first tile always does: LCP of its rank 0 is 0)
            int64_t hit_tile = -1;
            for (int64_t tp = top; tp >= gfirst && hit_tile < 0; tp -= 256) {
                u32 v[4];
#pragma unroll
                for (int c = 0; c < 4; c++) { const int64_t q = tp - 64 * c - lane; v[c] = q >= gfirst ? b.par.tmin[q] : INF; }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const u64 bal = __ballot((int64_t)v[c] < t);
                    if (bal && hit_tile < 0) hit_tile = tp - 64 * c - (int64_t)__builtin_ctzll(bal);
                }
            }
            if (hit_tile < 0) { k = 0; if (lane == 0) atomicOr(b.err, 2u); break; }
            lo = hit_tile * PT - ds.off;
            const int64_t hi = lo + PT - 1;
            k = wave_scan_down(LCP, flag, hi < ds.n - 1 ? hi : ds.n - 1, lo > 0 ? lo : 0, t);
            top = hit_tile - 1;                            // (a bound is only a promise: movers and values that moved on count in it)
        }
        if (lane == 0) b.par.msite[b.woff[dd] + slot] = (u32)k;
    }
}

// ---- final ranks of the movers ---------------------------------------------------------------
template <class K>
__device__ inline void bitonic_lds(K *key, uint16_t *pay, u32 np2, int nt) {
    for (u32 size = 2; size <= np2; size <<= 1)
        for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
            for (u32 k = threadIdx.x; k < np2 / 2; k += nt) {
                const u32 lo = (k / stride) * stride * 2 + (k % stride), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const K x = key[lo], y = key[hi];
                if ((x > y) == up) {
                    key[lo] = y; key[hi] = x;
                    if (pay) { const uint16_t p = pay[lo]; pay[lo] = pay[hi]; pay[hi] = p; }
                }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(TB) void k_pb_rank(RvBubbleArgs b, int first) {
    __shared__ u32 R[RV_PB_CAP];          // mover ranks, sorted
    __shared__ u64 key[RV_PB_CAP];        // (site << 32) | t, sorted
    __shared__ uint16_t idx[RV_PB_CAP];   // list index of the sorted key
    const int dd = first + blockIdx.x;
    if (!par_desc(b, dd)) return;
    const u32 M = b.par.mcnt[dd];
    if (M == 0) return;
    const RvBubbleDesc ds = b.desc[dd];
    const int64_t base = b.woff[dd];
    const sa_t *SA = b.SA + ds.off;
    const lcp_t *LCP = b.LCP + ds.off;
    const uint8_t *BW = b.BWT + ds.off;
    u32 np2 = 1; while (np2 < M) np2 <<= 1;
    for (u32 k = threadIdx.x; k < np2; k += TB) {
        if (k < M) {
            const u32 e = b.par.mrank[base + k];
            const u32 t = (u32)(ds.B - (int64_t)SA[e]);
            R[k] = e;
            key[k] = ((u64)b.par.msite[base + k] << 32) | t;
            idx[k] = (uint16_t)k;
        } else { R[k] = INF; key[k] = ~0ull; idx[k] = 0; }
    }
    __syncthreads();
    bitonic_lds<u32>(R, nullptr, np2, TB);
    bitonic_lds<u64>(key, idx, np2, TB);
    for (u32 q = threadIdx.x; q < M; q += TB) {
        const u32 site = (u32)(key[q] >> 32), t = (u32)key[q];
        const u32 e = b.par.mrank[base + idx[q]];
        int lo = 0, hi = (int)M;                                   // movers with rank < site
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (R[mid] < site) lo = mid + 1; else hi = mid; }
        const bool firstg = q == 0 || (u32)(key[q - 1] >> 32) != site;
        const bool lastg = q + 1 == M || (u32)(key[q + 1] >> 32) != site;
        b.par.R[base + q] = R[q];
        b.par.Qsite[base + q] = site;
        b.par.QF[base + q] = site - (u32)lo + q;
        b.par.Qt[base + q] = t;
        b.par.Qlcp[base + q] = firstg ? (u32)LCP[site] : (u32)key[q - 1];
        b.par.Qs[base + q] = SA[e];
        b.par.Qbw[base + q] = BW[e];
        b.par.Qlast[base + q] = lastg ? 1 : 0;
    }
}

// ---- the permutation ---------------------------------------------------------------------------
// number of entries of the sorted array a[0..M) that are < key, by one wave (64 probes per step)
__device__ inline u32 wave_lower_bound(const u32 *__restrict__ a, u32 M, u32 key) {
    const int lane = threadIdx.x & 63;
    u32 lo = 0, hi = M;
    while (lo < hi) {
        const u32 step = (hi - lo + 63) / 64;
        const u32 p = lo + (u32)lane * step;
        const bool less = p < hi && a[p] < key;
        const u32 c = (u32)__popcll(__ballot(less));
        if (c == 0) { hi = lo; break; }
        const u32 last = lo + (c - 1) * step;           // a[last] < key
        const u32 nxt = last + step;                     // first probe that is >= key (or past the end)
        lo = last + 1;
        hi = nxt < hi ? nxt : hi;
    }
    return lo;
}

struct TileMap { u32 a0, a1, b0, b1; int dirty; };
// movers / landing sites relative to ranks [lo, hi): a = movers with rank < x, b = sites < x
__device__ inline void tile_map(const RvBubbleArgs &b, int dd, u32 M, int64_t lo, int64_t hi, TileMap *tm) {
    if (threadIdx.x < 64) {
        const u32 *R = b.par.R + b.woff[dd], *S = b.par.Qsite + b.woff[dd];
        const u32 a0 = wave_lower_bound(R, M, (u32)lo), a1 = wave_lower_bound(R, M, (u32)hi);
        const u32 b0 = wave_lower_bound(S, M, (u32)lo), b1 = wave_lower_bound(S, M, (u32)hi);
        if (threadIdx.x == 0) { tm->a0 = a0; tm->a1 = a1; tm->b0 = b0; tm->b1 = b1; tm->dirty = (a1 > a0) || (b1 > b0) || (a0 != b0); }
    }
    __syncthreads();
}

__global__ __launch_bounds__(TB) void k_pb_copyout(RvBubbleArgs b, int first, int count) {
    __shared__ TileMap tm;
    int dd; int64_t ti;
    tile_of(b, first, count, blockIdx.x, &dd, &ti);
    if (!par_desc(b, dd)) return;
    const u32 M = b.par.mcnt[dd];
    if (M == 0) return;
    const RvBubbleDesc ds = b.desc[dd];
    const int64_t lo = ti * PT, hi = lo + PT < ds.n ? lo + PT : ds.n;
    tile_map(b, dd, M, lo, hi, &tm);
    if (!tm.dirty) return;
    for (int64_t r = lo + threadIdx.x; r < hi; r += TB) {
        const int64_t g = ds.off + r;
        b.scrSA[g] = b.SA[g]; b.scrLCP[g] = b.LCP[g]; b.scrBWT[g] = b.BWT[g];
    }
}

__global__ __launch_bounds__(TB) void k_pb_scatter(RvBubbleArgs b, int first, int count) {
    __shared__ TileMap tm;
    __shared__ sa_t cw_lo[32], cw_hi[32];
    int dd; int64_t ti;
    tile_of(b, first, count, blockIdx.x, &dd, &ti);
    if (!par_desc(b, dd)) return;
    const u32 M = b.par.mcnt[dd];
    if (M == 0) return;
    const RvBubbleDesc ds = b.desc[dd];
    const int64_t lo = ti * PT, hi = lo + PT < ds.n ? lo + PT : ds.n;
    const int ncw = ds.cut1 - ds.cut0 < 32 ? ds.cut1 - ds.cut0 : 32;
    if ((int)threadIdx.x < ncw) { cw_lo[threadIdx.x] = b.cut_lo[ds.cut0 + threadIdx.x]; cw_hi[threadIdx.x] = b.cut_hi[ds.cut0 + threadIdx.x]; }
    tile_map(b, dd, M, lo, hi, &tm);
    if (!tm.dirty) return;
    const u32 *R = b.par.R + b.woff[dd], *S = b.par.Qsite + b.woff[dd];
    const uint8_t *flag = b.flag + ds.off;
    for (int64_t r = lo + threadIdx.x; r < hi; r += TB) {
        if (flag[r] == 2) continue;                                 // movers are placed by k_pb_movers
        u32 x = tm.a0, y = tm.a1;                                   // movers with rank < r
        while (x < y) { const u32 mid = (x + y) >> 1; if (R[mid] < (u32)r) x = mid + 1; else y = mid; }
        u32 u = tm.b0, v = tm.b1;                                   // sites <= r
        while (u < v) { const u32 mid = (u + v) >> 1; if (S[mid] <= (u32)r) u = mid + 1; else v = mid; }
        const int64_t f = r - (int64_t)x + (int64_t)u;
        const int64_t g = ds.off + r, gf = ds.off + f;
        const sa_t s = b.scrSA[g];
        const lcp_t lv = b.scrLCP[g];
        b.SA[gf] = s; b.LCP[gf] = lv; b.BWT[gf] = b.scrBWT[g];
        if ((gf >> 11) != (g >> 11)) atomicMin(&b.par.tmin[gf >> 11], (u32)lv);
        if (f != r) {                                                // reveal.c:692 SAi[SA[x-1]] = x, kept only where a later cut will look
            bool in = false;
            for (int q = 0; q < ncw && !in; q++) in = s >= cw_lo[q] && s < cw_hi[q];
            for (int q = ds.cut0 + 32; q < ds.cut1 && !in; q++) in = s >= b.cut_lo[q] && s < b.cut_hi[q];
            if (in) b.SAi[s] = (sa_t)f;
        }
    }
}

// Copy-out and scatter in one pass over the ranks that change place: non-movers only move towards higher ranks, by at most the
// number of movers, so a tile writes into itself and into the first ranks of the tile (or, with thousands of movers, tiles) above it.  Every tile reads its ranks into
// registers, says so (tready), and writes once the tile above it has said the same.  Tiles are taken from the top of the round
// down: a workgroup only ever waits for one that was dispatched before it.  Half the traffic of the two-pass form (which moved
// 198 x 10^6 ranks per cut at the top levels of 2 x 250 Mbp: 36 bytes per rank).
__global__ __launch_bounds__(TB) void k_pb_shift(RvBubbleArgs b, int first, int count, int64_t total_tiles) {
    __shared__ TileMap tm;
    __shared__ sa_t cw_lo[32], cw_hi[32];
    __shared__ int s_gave_up;
    if (threadIdx.x == 0) s_gave_up = 0;
    int dd; int64_t ti;
    tile_of(b, first, count, total_tiles - 1 - (int64_t)blockIdx.x, &dd, &ti);
    u32 *ready = b.par.tready + (b.par.toff[dd] + ti);
    const u32 epoch = b.par.epoch;
    const u32 M = par_desc(b, dd) ? b.par.mcnt[dd] : 0u;
    const RvBubbleDesc ds = b.desc[dd];
    const int64_t lo = ti * PT, hi = lo + PT < ds.n ? lo + PT : ds.n;
    bool dirty = false;
    if (M > 0) {
        const int ncw = ds.cut1 - ds.cut0 < 32 ? ds.cut1 - ds.cut0 : 32;
        if ((int)threadIdx.x < ncw) { cw_lo[threadIdx.x] = b.cut_lo[ds.cut0 + threadIdx.x]; cw_hi[threadIdx.x] = b.cut_hi[ds.cut0 + threadIdx.x]; }
        tile_map(b, dd, M, lo, hi, &tm);
        dirty = tm.dirty;
    }
    if (!dirty) {      // nothing of this tile moves, nothing moves into it
        if (threadIdx.x == 0) __hip_atomic_store(ready, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    constexpr int PER = PT / TB;
    sa_t vs[PER]; lcp_t vl[PER]; uint8_t vb[PER], vf[PER];
    const uint8_t *flag = b.flag + ds.off;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int64_t r = lo + threadIdx.x + (int64_t)k * TB;
        if (r < hi) { const int64_t g = ds.off + r; vs[k] = b.SA[g]; vl[k] = b.LCP[g]; vb[k] = b.BWT[g]; vf[k] = flag[r]; }
        else { vs[k] = 0; vl[k] = 0; vb[k] = 0; vf[k] = 2; }
    }
    // The word only says "my loads have returned": nothing another workgroup reads is published with it, so neither side needs a
    // release or an acquire (at device scope those write back / invalidate the L2 of the XCD: 96000 times per launch that was 2.3 x
    // the time of the two-pass form).  The loads are waited for explicitly, the barrier collects the workgroup.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(ready, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the tiles above that are written from here: a rank moves up by (sites at or below it) - (movers below it) <= b1 - a0
        const int64_t top = hi - 1 + (int64_t)(tm.b1 - tm.a0);
        const int64_t t_last = (top < ds.n ? top : ds.n - 1) / PT;
        // (workgroups are dispatched in index order and this one only waits for earlier ones; should that ever not hold, the wait ends
        // after ~a second with the error word set instead of hanging the device)
        for (int64_t t = ti + 1; t <= t_last; t++) {
            u32 spins = 0;
            while (__hip_atomic_load(ready + (t - ti), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 24)) { atomicOr(b.err, 16u); s_gave_up = 1; break; }
            }
        }
    }
    const int ncw = ds.cut1 - ds.cut0 < 32 ? ds.cut1 - ds.cut0 : 32;
    const u32 *R = b.par.R + b.woff[dd], *S = b.par.Qsite + b.woff[dd];
    int64_t fr[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int64_t r = lo + threadIdx.x + (int64_t)k * TB;
        fr[k] = -1;
        if (vf[k] == 2) continue;                                   // movers are placed by k_pb_movers
        u32 x = tm.a0, y = tm.a1;                                   // movers with rank < r
        while (x < y) { const u32 mid = (x + y) >> 1; if (R[mid] < (u32)r) x = mid + 1; else y = mid; }
        u32 u = tm.b0, v = tm.b1;                                   // sites <= r
        while (u < v) { const u32 mid = (u + v) >> 1; if (S[mid] <= (u32)r) u = mid + 1; else v = mid; }
        fr[k] = r - (int64_t)x + (int64_t)u;
    }
    __syncthreads();                                    // thread 0 has seen the tile above read
    if (s_gave_up) return;                              // (never seen; the run fails on the error word -- but nothing is written over ranks nobody has read)
#pragma unroll
    for (int k = 0; k < PER; k++) {
        if (fr[k] < 0) continue;
        const int64_t r = lo + threadIdx.x + (int64_t)k * TB, f = fr[k];
        const int64_t g = ds.off + r, gf = ds.off + f;
        const sa_t s = vs[k];
        b.SA[gf] = s; b.LCP[gf] = vl[k]; b.BWT[gf] = vb[k];
        if ((gf >> 11) != (g >> 11)) atomicMin(&b.par.tmin[gf >> 11], (u32)vl[k]);
        if (f != r) {                                                // reveal.c:692 SAi[SA[x-1]] = x, kept only where a later cut will look
            bool in = false;
            for (int q = 0; q < ncw && !in; q++) in = s >= cw_lo[q] && s < cw_hi[q];
            for (int q = ds.cut0 + 32; q < ds.cut1 && !in; q++) in = s >= b.cut_lo[q] && s < b.cut_hi[q];
            if (in) b.SAi[s] = (sa_t)f;
        }
    }
}

__global__ __launch_bounds__(TB) void k_pb_movers(RvBubbleArgs b, int first, int count, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (id >= total) return;
    if (id == 0) *b.par.gcount = 0;                                  // nobody reads it in this kernel: ready for the next round.  (In front of the
                                                                     // return below: behind it, a round whose first cut had no candidates left its
                                                                     // movers on the list, and the next round's search went over them again)
    int dd; int64_t slot;
    slot_of(b, first, count, id, &dd, &slot);
    if (!par_desc(b, dd)) return;
    const RvBubbleDesc ds = b.desc[dd];
    const int64_t base = b.woff[dd];
    if (slot < (int64_t)b.par.mcnt[dd]) {
        const int64_t F = (int64_t)b.par.QF[base + slot];
        const sa_t s = b.par.Qs[base + slot];
        b.SA[ds.off + F] = s;                                        // reveal.c:700-703
        b.LCP[ds.off + F] = (lcp_t)b.par.Qlcp[base + slot];
        atomicMin(&b.par.tmin[(ds.off + F) >> 11], b.par.Qlcp[base + slot]);
        b.BWT[ds.off + F] = b.par.Qbw[base + slot];
        b.SAi[s] = (sa_t)F;
        if (b.par.Qlast[base + slot] && F + 1 < ds.n) {
            b.LCP[ds.off + F + 1] = (lcp_t)b.par.Qt[base + slot];
            atomicMin(&b.par.tmin[(ds.off + F + 1) >> 11], b.par.Qt[base + slot]);
        }
    }
    if (slot < (int64_t)b.cnt[dd]) b.flag[ds.off + b.list[base + slot]] = 0;
    if (slot == 0) b.state[dd].next = 0x7fffffff;                    // tells the sequential kernels this (child, cut) is done
}

}  // namespace

int rv_bubble_par_round_launch(Workspace &ws, const RvBubbleArgs &b, int first, int count, int64_t total_window, int64_t total_tiles, bool refresh_tmin) {
    if (count <= 0 || total_window <= 0) return 0;
    hipStream_t q = ws.stream;
    const unsigned wb = (unsigned)ceil_div(total_window, TB);
    RV_TRY(rv_bubble_window_launch(ws, b, first, count, total_window));
    hipLaunchKernelGGL(k_pb_runs, dim3(wb), dim3(TB), 0, q, b, first, count, total_window);
    RV_LAUNCH_CHECK();
    if (refresh_tmin || ws.opt.pb_refresh_tmin) {
        hipLaunchKernelGGL(k_pb_tilemin, dim3((unsigned)total_tiles), dim3(TB), 0, q, b, first, count);
        RV_LAUNCH_CHECK();
    }
    {
        const int64_t want = ceil_div(total_window, TB / 64);
        hipLaunchKernelGGL(k_pb_search, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(TB), 0, q, b);
    }
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pb_rank, dim3((unsigned)count), dim3(TB), 0, q, b, first);
    RV_LAUNCH_CHECK();
    if (b.par.tready) {
        hipLaunchKernelGGL(k_pb_shift, dim3((unsigned)total_tiles), dim3(TB), 0, q, b, first, count, total_tiles);
        RV_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_pb_copyout, dim3((unsigned)total_tiles), dim3(TB), 0, q, b, first, count);
        RV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_pb_scatter, dim3((unsigned)total_tiles), dim3(TB), 0, q, b, first, count);
        RV_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_pb_movers, dim3(wb), dim3(TB), 0, q, b, first, count, total_window);
    RV_LAUNCH_CHECK();
    return 0;
}

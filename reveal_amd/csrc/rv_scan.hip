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
// The pair scan is the roofline-judged kernel: a pure stream of 8 B per rank
// (4 B SA + 4 B LCP, 16-byte loads per lane), two byte gathers from T only for
// the ~0.4 % of ranks that survive the LCP tests, and an order-preserving
// append (one atomic per 1024-rank tile, tile table merged on the host).
#include "rv_common.h"
#include "rv_scan.h"

namespace {

constexpr int TB = 256;
constexpr int PAIR_ITEMS = 4;
constexpr int PAIR_TILE = TB * PAIR_ITEMS;   // == RV_PAIR_TILE

__device__ inline bool is_lower_c(uint8_t c) { return c >= 'a' && c <= 'z'; }

// left-maximality test of reveal.c:81-85 (only T[a-1] is inspected for N/$/lower)
__device__ inline bool left_maximal(const uint8_t *__restrict__ T, int64_t a, int64_t b) {
    if (a > 0 && b > 0) {
        const uint8_t ca = T[a - 1], cb = T[b - 1];
        return (ca != cb) || ca == 'N' || ca == '$' || is_lower_c(ca);
    }
    return true;
}

__device__ inline bool lcp_lt(lcp_t v, int minl) {
#ifdef RV_SA64
    return v < (lcp_t)minl;      // unsigned compare, as the reference's uint32 lcp_t
#else
    return v < minl;
#endif
}

__global__ __launch_bounds__(TB) void k_scan_pair(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, int64_t m,
                                                  const uint8_t *__restrict__ T, sa_t nsep0, int minl,
                                                  RvPairRec *__restrict__ out, u32 out_cap, u32 *__restrict__ counter,
                                                  uint2 *__restrict__ tiletab) {
    __shared__ u32 wsum[TB / 64];
    __shared__ u32 s_base;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * PAIR_TILE + (int64_t)threadIdx.x * PAIR_ITEMS;

    sa_t sa[PAIR_ITEMS];
    lcp_t lc[PAIR_ITEMS];
    if (i0 + PAIR_ITEMS <= m) {
#ifndef RV_SA64
        const int4 v = *reinterpret_cast<const int4 *>(SA + i0);
        sa[0] = v.x; sa[1] = v.y; sa[2] = v.z; sa[3] = v.w;
#else
#pragma unroll
        for (int k = 0; k < PAIR_ITEMS; k++) sa[k] = SA[i0 + k];
#endif
        const int4 c = *reinterpret_cast<const int4 *>(LCP + i0);
        lc[0] = (lcp_t)c.x; lc[1] = (lcp_t)c.y; lc[2] = (lcp_t)c.z; lc[3] = (lcp_t)c.w;
    } else {
#pragma unroll
        for (int k = 0; k < PAIR_ITEMS; k++) {
            sa[k] = (i0 + k < m) ? SA[i0 + k] : (sa_t)0;
            lc[k] = (i0 + k < m) ? LCP[i0 + k] : (lcp_t)0;
        }
    }
    // neighbours: previous rank's SA/LCP, next rank's LCP (0 past the end)
#ifdef RV_SA64
    sa_t  psa = (sa_t)__shfl_up((long long)sa[PAIR_ITEMS - 1], 1, 64);
#else
    sa_t  psa = (sa_t)__shfl_up((int)sa[PAIR_ITEMS - 1], 1, 64);
#endif
    lcp_t plc = (lcp_t)__shfl_up((int)lc[PAIR_ITEMS - 1], 1, 64);
    lcp_t nlc = (lcp_t)__shfl_down((int)lc[0], 1, 64);
    if (lane == 0) {
        if (i0 > 0 && i0 - 1 < m) { psa = SA[i0 - 1]; plc = LCP[i0 - 1]; } else { psa = 0; plc = 0; }
    }
    if (lane == 63) nlc = (i0 + PAIR_ITEMS < m) ? LCP[i0 + PAIR_ITEMS] : (lcp_t)0;

    u32 hit = 0;        // bitmask over my 4 ranks
#pragma unroll
    for (int k = 0; k < PAIR_ITEMS; k++) {
        const int64_t i = i0 + k;
        const sa_t  s1 = sa[k], s0 = (k == 0) ? psa : sa[k - 1];
        const lcp_t l = lc[k], lb = (k == 0) ? plc : lc[k - 1];
        lcp_t la = (k == PAIR_ITEMS - 1) ? nlc : lc[k + 1];
        if (i + 1 >= m) la = 0;
        bool ok = (i >= 1) && (i < m) && !lcp_lt(l, minl);
        ok = ok && ((s1 > nsep0) != (s0 > nsep0));          // not a repeat inside one sample
        ok = ok && (lb < l) && (la < l);                    // unique
        if (ok) {
            const sa_t a = s1 < s0 ? s1 : s0, b = s1 < s0 ? s0 : s1;
            ok = left_maximal(T, a, b);
        }
        hit |= ok ? (1u << k) : 0u;
    }
    // order-preserving append of this tile's survivors
    const u32 mine = __popc(hit);
    u32 inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { u32 t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    u32 before = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < TB / 64; k++) { const u32 c = wsum[k]; if (k < w) before += c; tot += c; }
    if (threadIdx.x == 0) {
        u32 base = tot ? atomicAdd(counter, tot) : 0u;
        s_base = base;
        tiletab[blockIdx.x] = make_uint2(base, tot);
    }
    __syncthreads();
    if (mine) {
        u32 q = s_base + before + (inc - mine);
#pragma unroll
        for (int k = 0; k < PAIR_ITEMS; k++) {
            if (hit & (1u << k)) {
                if (q < out_cap) {
                    const sa_t s1 = sa[k], s0 = (k == 0) ? psa : sa[k - 1];
                    RvPairRec r;
                    r.a = s1 < s0 ? s1 : s0;
                    r.b = s1 < s0 ? s0 : s1;
                    r.l = (u32)lc[k];
                    r.rank = (u32)(i0 + k);
                    out[q] = r;
                }
                q++;
            }
        }
    }
}

}  // namespace

int rv_scan_pair_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, int64_t m, const uint8_t *T, sa_t nsep0, int minl,
                        RvPairRec *out, u32 out_cap, u32 *counter, uint2 *tiletab) {
    if (m <= 0) return 0;
    const unsigned nb = (unsigned)ceil_div(m, PAIR_TILE);
    hipLaunchKernelGGL(k_scan_pair, dim3(nb), dim3(TB), 0, ws.stream, SA, LCP, m, T, nsep0, minl, out, out_cap, counter, tiletab);
    RV_LAUNCH_CHECK();
    return 0;
}

// rv_cascade_multi.hip -- the anchor cascade for more than two samples (one sequence each).
//
// The benchmark picker only takes matches present in EVERY sample of the sub-index (schemes.py:227; reveal.c:436-580
// getmultimums + :227-259 ismultimum).  For a descendant C of the root X that still holds all k samples:
//   * R[p] = the longest prefix suffix p shares with another suffix of ITS OWN sample (a repeat inside one genome);
//   * a full match of C (k suffixes, one per sample) longer than Rmax(C) = max R over C's positions is an LCP interval of
//     X with exactly those k members -- an extra member would belong to one of the k samples and raise that member's R --,
//     i.e. a full match of X cut to C (shifted to start behind the matched text in front of C on every sample, capped at
//     C's ends), and every such cut match longer than Rmax(C) is a full match of C.
// So C's choice (longest, ties to the smallest coordinate) is known from X's list of full matches whenever its best cut
// match is longer than Rmax(C); C has no match when it has no cut match of minl characters and Rmax(C) < minl, or when
// one of its intervals is shorter than minl.  A sub-index that lacks a sample (the picker then wants matches of the
// remaining ones), or whose best match is no longer than its repeats, is left undecided: rebuilt from the text of its
// intervals (at most 8192 suffixes) and handed to the level pipeline (rv_align.hip) as its frontier.  A larger
// undecided sub-index makes the cascade give up before anything is kept.
//
// tools/cascade_proto_multi.py is this algorithm on the CPU beside the oracle's literal recursion (1 994 random inputs
// with two to four samples: identical); tests/test_gpu_cascade.py and tools/fuzz.py run this file against the oracle.
// Kernels as in rv_cascade.hip, with k coordinates per match and k intervals per sub-index.
#include "rv_index.h"
#include "rv_cascade.h"
#include "rv_leaf.h"
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int TB = 256;
constexpr u32 NONE = 0xFFFFFFFFu;
#ifdef RV_SA64
constexpr int KEY_SHIFT = 40;
#else
constexpr int KEY_SHIFT = 32;
#endif
constexpr u64 KEY_LOW = (1ull << KEY_SHIFT) - 1;
enum { C_NCHILD = 0, C_NUND = 1, C_NWIT = 2, C_ERR = 3, C_MAXN = 4, C_LO = 5, C_HI = 6, C_LEVELS = 7, C_NCAND = 8, C_NANCH = 9, C_STEPS = 10, C_MAXDEPTH = 11, C_TICKET = 12, C_BIGERR = 13, C_BIGTOT = 14 /* 64 bits */ };

__device__ inline int cm_sample(const sa_t *__restrict__ nsep, int k, sa_t pos) {      // number of separators in front of pos (interface.c:116-134)
    int s = 0;
    for (int q = 0; q < k - 1; q++) s += nsep[q] < pos ? 1 : 0;
    return s;
}
__device__ inline u64 shfl_up64m(u64 v, int d) {
    const u32 lo = __shfl_up((u32)v, d, 64), hi = __shfl_up((u32)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}
__device__ inline void seg_max64(u64 *__restrict__ dst, u32 child, u64 val, bool active) {
    const int lane = threadIdx.x & 63;
    if (!active) { child = NONE; val = 0; }
    val = rv_wave_seg_max_u64(child, val);
    const u32 nc = __shfl_down(child, 1, 64);
    const bool last = lane == 63 || nc != child;
    if (active && last && val > dst[child]) atomicMax((unsigned long long *)&dst[child], (unsigned long long)val);
}
__device__ inline void seg_max32(u32 *__restrict__ dst, u32 child, u32 val, bool active) {
    const int lane = threadIdx.x & 63;
    if (!active) { child = NONE; val = 0; }
    val = rv_wave_seg_max_u32(child, val);
    const u32 nc = __shfl_down(child, 1, 64);
    const bool last = lane == 63 || nc != child;
    if (active && last && val > dst[child]) atomicMax(&dst[child], val);
}

// sample of every rank's suffix (one byte: the walks below read it many times)
// (sixteen ranks per thread, the separators from LDS: a thread per rank was 7.8 x 10^5 waves of two loads and a byte store, 133 us at 10 x 5 Mbp)
constexpr int SO_PER = 16;
__global__ __launch_bounds__(TB) void k_casm_so(const sa_t *__restrict__ SA, int64_t n, const sa_t *__restrict__ nsep, int k, uint8_t *__restrict__ so) {
    __shared__ sa_t s_sep[RV_CASM_K];
    if ((int)threadIdx.x < RV_CASM_K) s_sep[threadIdx.x] = (int)threadIdx.x < k - 1 ? nsep[threadIdx.x] : (sa_t)0;
    __syncthreads();
    const int64_t j0 = ((int64_t)blockIdx.x * TB + threadIdx.x) * SO_PER;
    if (j0 >= n) return;
    if (j0 + SO_PER <= n) {
        sa_t v[SO_PER];
        __builtin_memcpy(v, SA + j0, sizeof v);
        u32 w[SO_PER / 4];
#pragma unroll
        for (int x = 0; x < SO_PER / 4; x++) w[x] = 0;
#pragma unroll
        for (int r = 0; r < SO_PER; r++) {
            u32 sm = 0;
            for (int q = 0; q < k - 1; q++) sm += s_sep[q] < v[r] ? 1u : 0u;
            w[r >> 2] |= sm << (8 * (r & 3));
        }
        __builtin_memcpy(so + j0, w, sizeof w);
    } else {
        for (int64_t j = j0; j < n; j++) { u32 sm = 0; for (int q = 0; q < k - 1; q++) sm += s_sep[q] < SA[j] ? 1u : 0u; so[j] = (uint8_t)sm; }
    }
}

// full matches of the root: the LCP interval of exactly k ranks that ends at rank u (reveal.c:436-580 for n == nsamples).
// A workgroup stages LCP, sample and BWT byte of its 2048 ranks (+ the k - 1 in front, one behind) in LDS: streamed once, the
// per-rank tests read LDS (one thread per rank with its own global reads ran at 0.4 TB/s of its 8 B/rank)
constexpr int MS_ITEMS = 8;
constexpr int MS_TILE = TB * MS_ITEMS;
constexpr int CM_REGIONS = 64;
// (KT: the number of samples when it is one the kernel was built for -- the windows' loops unrolled, their LDS reads at fixed offsets: 332 -> 256 us at
// 10 x 5 Mbp; 0: any number, the loops stay loops -- which is what the compiler's "loop not unrolled" remark would say about that instance)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
template <int KT>
__global__ __launch_bounds__(TB) void k_casm_scan(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, const uint8_t *__restrict__ BWT, const uint8_t *__restrict__ so,
                                                  int64_t n, int k_rt, u32 minl, u32 *__restrict__ c_len, sa_t *__restrict__ c_pos, u32 cap /* per region */, u32 *__restrict__ region_cnt) {
    const int k = KT ? KT : k_rt;
    __shared__ __attribute__((aligned(16))) u32 sl[MS_TILE + RV_CASM_K + 16];          // LCP of ranks u0-HS .. u0+TILE (0 outside the array); HS = the halo rounded up to 16
    __shared__ __attribute__((aligned(16))) uint8_t ss[MS_TILE + RV_CASM_K + 16], sb[MS_TILE + RV_CASM_K + 16];
    __shared__ uint16_t cand[MS_TILE], cand2[MS_TILE];
    __shared__ u32 ncand;
    const int64_t u0 = (int64_t)blockIdx.x * MS_TILE;
    const int H = k - 1;
    constexpr int HS = 16;                       // staged ranks in front of the tile (>= RV_CASM_K - 1, a multiple of the 16-byte loads)
    static_assert(RV_CASM_K - 1 <= HS && MS_TILE % 16 == 0, "halo");
    if (threadIdx.x == 0) ncand = 0;
    if (u0 >= HS && u0 + MS_TILE + 1 <= n && sizeof(lcp_t) == 4) {
        // the tile with its halo in 16-byte loads (ranks u0 - 16 .. u0 + TILE), the one rank behind it on its own
        const uint4 *L4 = reinterpret_cast<const uint4 *>(LCP + (u0 - HS));
        for (int x = threadIdx.x; x < (MS_TILE + HS) / 4; x += TB) reinterpret_cast<uint4 *>(sl)[x] = L4[x];
        const uint4 *S4 = reinterpret_cast<const uint4 *>(so + (u0 - HS)), *B4 = reinterpret_cast<const uint4 *>(BWT + (u0 - HS));
        for (int x = threadIdx.x; x < (MS_TILE + HS) / 16; x += TB) {
            reinterpret_cast<uint4 *>(ss)[x] = S4[x];
            uint4 v = B4[x];
            v.x &= 0x7f7f7f7fu; v.y &= 0x7f7f7f7fu; v.z &= 0x7f7f7f7fu; v.w &= 0x7f7f7f7fu;      // (RV_BWT_CHAR: the side bit off)
            reinterpret_cast<uint4 *>(sb)[x] = v;
        }
        if (threadIdx.x == 0) sl[MS_TILE + HS] = (u32)LCP[u0 + MS_TILE];
    } else {
        for (int x = threadIdx.x; x < MS_TILE + HS + 1; x += TB) { const int64_t j = u0 - HS + x; sl[x] = (j >= 0 && j < n) ? (u32)LCP[j] : 0u; }
        for (int x = threadIdx.x; x < MS_TILE + HS; x += TB) {
            const int64_t j = u0 - HS + x;
            const bool in = j >= 0 && j < n;
            ss[x] = in ? so[j] : (uint8_t)0; sb[x] = in ? (uint8_t)(BWT[j] & RV_BWT_CHAR) : (uint8_t)0;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // the ranks where an interval can end at all (a value of minl or more, larger than the next one): one in k for related samples.  Every rank
    // walking its window made a wave run the long path for its few lanes that needed it; the candidates are listed in LDS and walked densely
    {
        // a thread looks at MS_ITEMS ranks in a row (one reservation per wave: ranked one round of 256 strided ranks at a time, the kernel spent
        // its time on eight ballots and eight returning LDS atomics per wave)
        const int t0 = (int)threadIdx.x * MS_ITEMS;
        u32 v[MS_ITEMS + 1];
#pragma unroll
        for (int i = 0; i <= MS_ITEMS; i++) v[i] = sl[t0 + HS + i];
        u32 mask = 0;
#pragma unroll
        for (int i = 0; i < MS_ITEMS; i++) {
            const int64_t u = u0 + t0 + i;
            mask |= (u32)((u < n) & (u - H >= 0) & (v[i] >= minl) & (v[i] > v[i + 1])) << i;
        }
        const u32 c = (u32)__popc(mask);
        const u32 inc = rv_wave_incl_sum_u32(c);
        u32 base = 0;
        if (lane == 63 && inc) base = atomicAdd(&ncand, inc);
        base = (u32)__shfl((int)base, 63, 64) + inc - c;
#pragma unroll
        for (int i = 0; i < MS_ITEMS; i++) if ((mask >> i) & 1u) cand[base++] = (uint16_t)(t0 + i);
    }
    __syncthreads();
    const u32 nc = ncand;
    // Two dense stages.  First what most candidates fail on -- the window's smallest value (with related samples a candidate closes the k homologues of
    // one position, and some of them have left the others within minl symbols) and the values on either side of it --, which is H reads of LDS;
    // the ones that pass (one in seventy at 10 x 5 Mbp) are listed again, and only they pay the H + 1 reads of the samples and the 2 H of the bytes
    // in front.  (One stage, every test on every candidate: 38 reads of LDS per candidate, 257 us at 10 x 5 Mbp.)
    __syncthreads();      // (ncand has been read by everybody: it counts the second list from here on)
    if (threadIdx.x == 0) ncand = 0;
    __syncthreads();
    for (u32 ci = threadIdx.x; ci < ((nc + 63u) & ~63u); ci += TB) {      // (whole waves: the ballot below)
        bool ok = ci < nc;
        const int t = ok ? (int)cand[ci] : 0;
        const int x = t + HS;
        u32 v = sl[x];
        const u32 nxt = sl[x + 1];
#pragma unroll
        for (int d = 1; d < H; d++) { const u32 y = sl[x - d]; v = y < v ? y : v; }
        ok = ok & (v >= minl) & (v > nxt) & (sl[x - H] < v);
        const u64 bal = __ballot(ok);
        if (bal) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&ncand, (u32)__popcll(bal));
            base = (u32)__shfl((int)base, 0, 64);
            if (ok) cand2[base + (u32)__popcll(bal & lt)] = (uint16_t)t;
        }
    }
    __syncthreads();
    const u32 nc2 = ncand;
    for (u32 ci = threadIdx.x; ci < ((nc2 + 63u) & ~63u); ci += TB) {
        bool ok = ci < nc2;
        const int t = ok ? (int)cand2[ci] : 0;
        const int x = t + HS;
        const int64_t u = u0 + t;
        u32 v = sl[x];
#pragma unroll
        for (int d = 1; d < H; d++) { const u32 y = sl[x - d]; v = y < v ? y : v; }
        u32 seen = 0;
#pragma unroll
        for (int d = 0; d <= H; d++) seen |= 1u << ss[x - d];
        bool mx = false;      // left-maximal (reveal.c:246-256; the BWT byte holds '$' where SA == 0)
#pragma unroll
        for (int d = 1; d <= H; d++) {
            const uint8_t ca = sb[x - d], cb = sb[x - d + 1];
            mx |= (cb == '$') | (ca != cb) | (ca == 'N') | (ca == '$') | ((ca >= 'a') & (ca <= 'z'));
        }
        ok = ok & (__popc(seen) == H + 1) & mx;
        // (the list in CM_REGIONS regions with a counter each: one counter was 50 000 returning atomics on one address, 0.7 of the kernel's 0.9 ms)
        const u64 bal = __ballot(ok);
        if (bal) {
            const u32 reg = blockIdx.x & (CM_REGIONS - 1);
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&region_cnt[reg], (u32)__popcll(bal));
            base = (u32)__shfl((int)base, 0, 64);
            if (ok) {
                const u32 i = base + (u32)__popcll(bal & lt);
                if (i < cap) {
                    const size_t o = (size_t)reg * cap + i;
                    c_len[o] = v;
#pragma unroll
                    for (int d = 0; d <= H; d++) c_pos[o * k + ss[x - d]] = SA[u - d];
                }
            }
        }
    }
}
#pragma clang diagnostic pop

// R: a suffix' longest common prefix with another suffix of its own sample, where it reaches minl -- the nearest such suffix
// above and below in the array, the range minimum of LCP in between; a walk that does not find one within WALK ranks keeps
// the running minimum (an upper bound).  LCP and sample of the tile's ranks and of WALK ranks on either side come through LDS.
constexpr int WALK = 256;
// Only a rank whose RUN -- the ranks around it that are joined by LCP values of minl or more -- holds some sample twice can have such a suffix: both walks
// stay inside the run.  With related genomes a run is the k homologues of one position, all of different samples, so the runs are classified first
// (the rank a run starts at walks it once: ten steps for a tenth of the ranks) and only the ranks of a run that repeats a sample, is longer than
// RUN_LIM or is not seen whole do the two walks of up to WALK ranks -- every rank walking to its neighbours of the same sample ten ranks away was
// 1 360 vector instructions per wave, 0.44 ms at 10 x 5 Mbp.
constexpr int RUN_LIM = 64;
__global__ __launch_bounds__(TB) void k_casm_witness(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, const uint8_t *__restrict__ so, int64_t n, u32 minl,
                                                     sa_t *__restrict__ w_pos, u32 *__restrict__ w_val, u32 cap, u32 *__restrict__ counters) {
    constexpr int SPAN = MS_TILE + 2 * WALK + 1;
    __shared__ u32 sl[SPAN];
    __shared__ uint8_t ss[SPAN];
    __shared__ uint8_t sus[SPAN + 3];      // 1: the rank's run may hold a sample twice
    __shared__ uint16_t s_lead[SPAN];
    __shared__ u32 s_nl;
    const int64_t u0 = (int64_t)blockIdx.x * MS_TILE;
    for (int x = threadIdx.x; x < SPAN; x += TB) {
        const int64_t j = u0 - WALK + x;
        const bool in = j >= 0 && j < n;
        sl[x] = in ? (u32)LCP[j] : 0u; ss[x] = in ? so[j] : (uint8_t)255;
    }
    for (int x = threadIdx.x; x < (SPAN + 3) / 4; x += TB) reinterpret_cast<u32 *>(sus)[x] = 0u;
    if (threadIdx.x == 0) s_nl = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // where the runs start (a value below minl -- or the window's first rank, whose run is not seen whole), listed densely: walked where they stand, a
    // wave ran the longest walk of its lanes once per round of 256 positions, ten rounds
    for (int x0 = 0; x0 < SPAN; x0 += TB) {
        const int x = x0 + (int)threadIdx.x;
        const bool lead = x < SPAN && (x == 0 || sl[x] < minl);
        const u64 bal = __ballot(lead);
        u32 base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_nl, (u32)__popcll(bal));
        base = (u32)__shfl((int)base, 0, 64);
        if (lead) s_lead[base + (u32)__popcll(bal & lt)] = (uint16_t)x;
    }
    __syncthreads();
    const u32 nl = s_nl;
    for (u32 li = threadIdx.x; li < nl; li += TB) {
        const int x = (int)s_lead[li];
        u32 seen = ss[x] < 32 ? 1u << ss[x] : 0u;
        bool bad = x == 0 && sl[0] >= minl;
        int len = 1;
        while (x + len < SPAN && sl[x + len] >= minl) {
            if (len >= RUN_LIM) { bad = true; break; }
            const u32 b = ss[x + len] < 32 ? 1u << ss[x + len] : 0u;
            bad |= (seen & b) != 0u;
            seen |= b; len++;
        }
        bad |= x + len >= SPAN;      // (it may go on behind the window)
        if (bad) {
            int e = x + len;
            while (e < SPAN && sl[e] >= minl) e++;      // (a run cut off at RUN_LIM: all of it)
            for (int y = x; y < e; y++) sus[y] = 1;
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int r0 = 0; r0 < MS_ITEMS; r0++) {
        const int x = r0 * TB + threadIdx.x + WALK;
        const int64_t j = u0 + r0 * TB + threadIdx.x;
        u32 r = 0;
        if (j < n && sus[x]) {
            const uint8_t s = ss[x];
            // upwards: lcp(rank j - d, rank j) = min LCP[j-d+1 .. j]; rank -1 and beyond: LCP 0 (staged), the walk ends there
            u32 mn = 0xFFFFFFFFu; bool open = true;
            for (int d = 1; d <= WALK && open; d++) {
                const u32 y = sl[x - d + 1];
                mn = y < mn ? y : mn;
                if (mn < minl) { mn = 0; open = false; }
                else if (ss[x - d] == s) open = false;
            }
            if (mn >= minl && mn != 0xFFFFFFFFu) r = mn;       // (found, or WALK ranks without one: the running minimum bounds it)
            mn = 0xFFFFFFFFu; open = true;
            for (int d = 1; d <= WALK && open; d++) {          // downwards: lcp(rank j, rank j + d) = min LCP[j+1 .. j+d]
                const u32 y = sl[x + d];
                mn = y < mn ? y : mn;
                if (mn < minl) { mn = 0; open = false; }
                else if (ss[x + d] == s) open = false;
            }
            if (mn >= minl && mn != 0xFFFFFFFFu && mn > r) r = mn;
        }
        const bool hit = r >= minl && r > 0;
        const u64 bal = __ballot(hit);
        if (bal) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&counters[C_NWIT], (u32)__popcll(bal));
            base = (u32)__shfl((int)base, 0, 64);
            if (hit) {
                const u32 o = base + (u32)__popcll(bal & lt);
                if (o < cap) { w_pos[o] = SA[j]; w_val[o] = r; }
            }
        }
    }
}

// the regions' entries as one dense list: key = first coordinate, value = where the entry lives
// the regions' counters, kept 64 words apart while the scan counts into them, side by side for the host and k_casm_keys
__global__ void k_casm_collect(const u32 *__restrict__ spread, int stride, u32 *__restrict__ region_cnt) {
    if (threadIdx.x < (unsigned)CM_REGIONS) region_cnt[threadIdx.x] = spread[(size_t)threadIdx.x * stride];
}
__global__ __launch_bounds__(TB) void k_casm_keys(const sa_t *__restrict__ c_pos, int k, u32 rcap, const u32 *__restrict__ region_cnt, const u32 *__restrict__ region_off,
                                                  u64 *__restrict__ keys, u32 *__restrict__ vals) {
    const u32 reg = blockIdx.y;
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= region_cnt[reg]) return;
    const u32 slot = reg * rcap + i, dense = region_off[reg] + i;
    keys[dense] = (u64)c_pos[(size_t)slot * k]; vals[dense] = slot;
}
__global__ __launch_bounds__(TB) void k_casm_gather(const u32 *__restrict__ len_in, const sa_t *__restrict__ pos_in, const u32 *__restrict__ perm, int k, u32 M,
                                                    u32 *__restrict__ len_out, sa_t *__restrict__ pos_out, u32 *__restrict__ c_child) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= M) return;
    const u32 src = perm[i];
    len_out[i] = len_in[src]; c_child[i] = 0u;
    for (int s = 0; s < k; s++) pos_out[(size_t)i * k + s] = pos_in[(size_t)src * k + s];
}

struct CmTabs {      // per sub-index: k intervals, its best bid, Rmax, depth, state (0 open, 1 split, 2 ended, 3 undecided), children, the chosen match
    sa_t *b, *e; u64 *best; u32 *rmax; int32_t *depth; u32 *state, *lead, *trail; sa_t *q; u32 *ql;
    u32 *job = nullptr;      // the level loops of several jobs as one (rv_batch_*): the job a sub-index belongs to (inherited from its root); NULL otherwise
};
__global__ void k_casm_init(CmTabs t, int k, const sa_t *__restrict__ root_b, const sa_t *__restrict__ root_e, u32 *__restrict__ counters, u32 *__restrict__ w_child, u32 nw) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        for (int s = 0; s < k; s++) { t.b[s] = root_b[s]; t.e[s] = root_e[s]; }
        t.best[0] = 0; t.rmax[0] = 0; t.depth[0] = 0; t.state[0] = 0; t.lead[0] = NONE; t.trail[0] = NONE; t.ql[0] = 0;
        counters[C_NCHILD] = 1; counters[C_NUND] = 0; counters[C_ERR] = 0; counters[C_MAXN] = 0; counters[C_LO] = 0; counters[C_HI] = 1; counters[C_LEVELS] = 0;
        counters[C_NANCH] = 0; counters[C_STEPS] = 0; counters[C_MAXDEPTH] = 0; counters[C_TICKET] = 0;
    }
    if (i < nw) w_child[i] = 0u;
}

// the match cut to the intervals [B_s, E_s): -> its length (< minl: not there), *sh = how far its start moves
// (the k coordinates of a match / the k bounds of a sub-index are loaded as whole rows into registers, every load issued before the first one is used:
// loops of k dependent trips to memory -- and local arrays indexed by a runtime counter, which live in scratch memory -- made a level of ten samples
// 18 + 11 + 6 us however few sub-indices it held)
typedef sa_t CmRow[RV_CASM_K];
__device__ inline void cm_row(const sa_t *__restrict__ src, int k, CmRow &v) {
#pragma unroll
    for (int s = 0; s < RV_CASM_K; s++) v[s] = s < k ? src[s] : (sa_t)0;
}
__device__ inline int64_t cm_cut(const CmRow &p, int k, const CmRow &B, const CmRow &E, int64_t len, int64_t *sh) {
    int64_t k0 = 0;
#pragma unroll
    for (int s = 0; s < RV_CASM_K; s++) if (s < k) { const int64_t d = (int64_t)B[s] - (int64_t)p[s]; k0 = d > k0 ? d : k0; }
    int64_t l = len - k0;
#pragma unroll
    for (int s = 0; s < RV_CASM_K; s++) if (s < k) { const int64_t r = (int64_t)E[s] - ((int64_t)p[s] + k0); l = r < l ? r : l; }
    *sh = k0;
    return l;
}
__global__ __launch_bounds__(TB) void k_casm_assign(const sa_t *__restrict__ c_pos, const u32 *__restrict__ c_len, u32 *__restrict__ c_child, u32 M,
                                                    const sa_t *__restrict__ w_pos, const u32 *__restrict__ w_val, u32 *__restrict__ w_child, u32 NW,
                                                    CmTabs t, int k, const sa_t *__restrict__ nsep, int64_t minl, int first, const uint8_t *__restrict__ w_smp = nullptr) {
    const u32 tid = blockIdx.x * TB + threadIdx.x;
    const u32 mblocks = (M + TB - 1) / TB;
    if (blockIdx.x < mblocks) {
        const u32 i = tid;
        u32 c = i < M ? c_child[i] : NONE;
        bool live = c != NONE;
        u64 key = 0;
        if (live) {
            CmRow p, rb, re, rq;
            cm_row(c_pos + (size_t)i * k, k, p);
            cm_row(t.b + (size_t)c * k, k, rb); cm_row(t.e + (size_t)c * k, k, re);
            if (!first) cm_row(t.q + (size_t)c * k, k, rq);
            const int64_t len = (int64_t)c_len[i];
            int64_t sh = 0, l = -1;
            if (!first) {
                u32 nc = NONE;
                const u32 st = t.state[c], lc = t.lead[c], tc = t.trail[c];
                const int64_t L = (int64_t)t.ql[c];
                if (st == 1u) {
                    if (lc != NONE) {
                        l = cm_cut(p, k, rb, rq, len, &sh);
                        if (l >= minl) nc = lc;
                    }
                    if (nc == NONE && tc != NONE) {
                        CmRow B;
#pragma unroll
                        for (int s = 0; s < RV_CASM_K; s++) B[s] = (sa_t)((int64_t)rq[s] + L);
                        l = cm_cut(p, k, B, re, len, &sh);
                        if (l >= minl) nc = tc;
                    }
                }
                c = nc;
                c_child[i] = c;
                live = c != NONE;
            } else {
                l = cm_cut(p, k, rb, re, len, &sh);
                live = l >= minl;
                if (!live) { c = NONE; c_child[i] = NONE; }
            }
            if (live) key = ((u64)l << KEY_SHIFT) | (KEY_LOW - (u64)((int64_t)p[0] + sh));
        }
        seg_max64(t.best, c, key, live);
    } else {
        const u32 i = tid - mblocks * TB;
        u32 c = i < NW ? w_child[i] : NONE;
        bool live = c != NONE;
        u32 v = 0;
        if (live) {
            const sa_t pos = w_pos[i];
            v = w_val[i];
            const int s = w_smp ? (int)w_smp[i] : cm_sample(nsep, k, pos);      // (several jobs in one loop: every job has separators of its own, the sample was made with the list)
            if (!first) {
                u32 nc = NONE;
                if (t.state[c] == 1u) {
                    const int64_t b = t.b[(size_t)c * k + s], e = t.e[(size_t)c * k + s], q = t.q[(size_t)c * k + s], L = t.ql[c];
                    if ((int64_t)pos >= b && (int64_t)pos < q) nc = t.lead[c];
                    else if ((int64_t)pos >= q + L && (int64_t)pos < e) nc = t.trail[c];
                }
                c = nc;
                w_child[i] = c;
                live = c != NONE;
            } else {
                live = pos >= t.b[(size_t)c * k + s] && pos < t.e[(size_t)c * k + s];
                if (!live) { c = NONE; w_child[i] = NONE; }
            }
        }
        seg_max32(t.rmax, c, v, live);
    }
}

__global__ __launch_bounds__(TB) void k_casm_winner(const sa_t *__restrict__ c_pos, const u32 *__restrict__ c_len, const u32 *__restrict__ c_child, u32 M, CmTabs t, int k, int64_t minl) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= M) return;
    const u32 c = c_child[i];
    if (c == NONE) return;
    CmRow p, rb, re;
    cm_row(c_pos + (size_t)i * k, k, p); cm_row(t.b + (size_t)c * k, k, rb); cm_row(t.e + (size_t)c * k, k, re);
    const u64 bestc = t.best[c];
    int64_t sh;
    const int64_t l = cm_cut(p, k, rb, re, (int64_t)c_len[i], &sh);
    if (l < minl) return;
    if ((((u64)l << KEY_SHIFT) | (KEY_LOW - (u64)((int64_t)p[0] + sh))) == bestc) {
#pragma unroll
        for (int s = 0; s < RV_CASM_K; s++) if (s < k) t.q[(size_t)c * k + s] = (sa_t)((int64_t)p[s] + sh);
        t.ql[c] = (u32)l;
    }
}

__global__ __launch_bounds__(TB) void k_casm_decide(CmTabs t, int k, u32 minl, u32 *__restrict__ counters, u32 child_cap, u32 *__restrict__ und_list, u32 leaf_n,
                                                    u32 *__restrict__ an_l, sa_t *__restrict__ an_pos, u32 an_cap, u32 *__restrict__ an_job = nullptr,
                                                    u32 *__restrict__ job_cnt = nullptr, const u32 *__restrict__ job_abase = nullptr) {
    __shared__ u32 s_cnt[TB / 64][3];
    __shared__ u32 s_base[3];
    const u32 lo = counters[C_LO], hi = counters[C_HI];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (u32 first = lo + blockIdx.x * TB; first < hi; first += gridDim.x * TB) {
        const u32 id = first + threadIdx.x;
        const bool in = id < hi;
        int64_t total = 0, mn = (int64_t)1 << 62, nlong = 0; int empty = 0;
        int64_t nlead = 0, ntrail = 0;
        u32 bl = 0, rm = 0, L = 0; int32_t dp = 0;
        CmRow rb, re, rq;      // the sub-index' row: begins, ends, the chosen match
        if (in) {
            cm_row(t.b + (size_t)id * k, k, rb); cm_row(t.e + (size_t)id * k, k, re); cm_row(t.q + (size_t)id * k, k, rq);
            bl = (u32)(t.best[id] >> KEY_SHIFT); rm = t.rmax[id]; dp = t.depth[id]; L = t.ql[id];
#pragma unroll
            for (int s = 0; s < RV_CASM_K; s++) if (s < k) {
                const int64_t len = (int64_t)re[s] - (int64_t)rb[s];
                total += len; mn = len < mn ? len : mn; empty += len <= 0; nlong += len >= (int64_t)minl;
            }
        }
        // every sample present and room for a match in each: decided from the list; a missing sample changes what the picker wants
        const bool lacking = in && empty > 0;
        const bool can = in && !lacking && mn >= (int64_t)minl;
        const bool split = can && bl >= minl && bl > rm;
        const bool und = (can && !split && rm >= minl) || (lacking && nlong >= 2);
        if (split) {
            if (L != bl) atomicOr(&counters[C_ERR], 2u);
#pragma unroll
            for (int s = 0; s < RV_CASM_K; s++) if (s < k) {
                const int64_t q = rq[s];
                nlead += q - (int64_t)rb[s];
                ntrail += (int64_t)re[s] - q - L;
            }
        }
        const bool lead = split && nlead > 0, trail = split && ntrail > 0;
        const u64 b_lead = __ballot(lead), b_trail = __ballot(trail), b_split = __ballot(split), b_und = __ballot(und);
        if (lane == 0) { s_cnt[w][0] = (u32)__popcll(b_lead) + (u32)__popcll(b_trail); s_cnt[w][1] = (u32)__popcll(b_split); s_cnt[w][2] = (u32)__popcll(b_und); }
        __syncthreads();
        if (threadIdx.x < 3) {
            u32 tot = 0;
            for (int q = 0; q < TB / 64; q++) tot += s_cnt[q][threadIdx.x];
            u32 *ctr = threadIdx.x == 0 ? &counters[C_NCHILD] : threadIdx.x == 1 ? &counters[C_NANCH] : &counters[C_NUND];
            s_base[threadIdx.x] = tot ? atomicAdd(ctr, tot) : 0u;
        }
        __syncthreads();
        u32 base_c = s_base[0], base_a = s_base[1], base_u = s_base[2];
        for (int q = 0; q < w; q++) { base_c += s_cnt[q][0]; base_a += s_cnt[q][1]; base_u += s_cnt[q][2]; }
        if (split) {
            u32 slot = base_c + (u32)__popcll(b_lead & lt) + (u32)__popcll(b_trail & lt);
            u32 lc = NONE, tc = NONE;
            if (lead) {
                if (slot < child_cap) {
#pragma unroll
                    for (int s = 0; s < RV_CASM_K; s++) if (s < k) { t.b[(size_t)slot * k + s] = rb[s]; t.e[(size_t)slot * k + s] = rq[s]; }
                    t.best[slot] = 0; t.rmax[slot] = 0; t.depth[slot] = dp + 1; t.state[slot] = 0; t.lead[slot] = NONE; t.trail[slot] = NONE; t.ql[slot] = 0;
                    if (t.job) t.job[slot] = t.job[id];
                    lc = slot;
                } else atomicOr(&counters[C_ERR], 1u);
                slot++;
            }
            if (trail) {
                if (slot < child_cap) {
#pragma unroll
                    for (int s = 0; s < RV_CASM_K; s++) if (s < k) { t.b[(size_t)slot * k + s] = (sa_t)((int64_t)rq[s] + L); t.e[(size_t)slot * k + s] = re[s]; }
                    t.best[slot] = 0; t.rmax[slot] = 0; t.depth[slot] = dp + 1; t.state[slot] = 0; t.lead[slot] = NONE; t.trail[slot] = NONE; t.ql[slot] = 0;
                    if (t.job) t.job[slot] = t.job[id];
                    tc = slot;
                } else atomicOr(&counters[C_ERR], 1u);
            }
            t.lead[id] = lc; t.trail[id] = tc;
            // (several jobs in one loop: a job's anchors stand together, in its own stretch of the arrays -- the counters 256 B apart, see k_full_scan)
            u32 as = base_a + (u32)__popcll(b_split & lt);
            u32 as_cap = an_cap;
            if (job_cnt) { const u32 jb = t.job[id]; as = job_abase[jb] + atomicAdd(&job_cnt[jb * 64], 1u); as_cap = job_abase[jb + 1]; }
            if (as < as_cap) {
                an_l[as] = L;
                if (an_job) an_job[as] = t.job[id];
#pragma unroll
                for (int s = 0; s < RV_CASM_K; s++) if (s < k) an_pos[(size_t)as * k + s] = rq[s];
            }
            else atomicOr(&counters[C_ERR], 4u);
        }
        if (in) t.state[id] = split ? 1u : (und ? 3u : 2u);
        if (und) {
            und_list[base_u + (u32)__popcll(b_und & lt)] = id;
            if ((u64)total > (u64)leaf_n) {      // (above the size one workgroup rebuilds in LDS: k_casmb_*)
                atomicMax(&counters[C_MAXN], (u32)(total > 0xFFFFFFFFll ? 0xFFFFFFFFll : total));
                atomicAdd((unsigned long long *)&counters[C_BIGTOT], (unsigned long long)total);
            }
        }
        // visited here: everything but the undecided ones (the level pipeline counts those when it takes them)
        const u64 b_vis = __ballot(in && !und);
        u32 md = (in && !und) ? (u32)dp : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const u32 om = __shfl_down(md, d, 64); md = om > md ? om : md; }
        if (lane == 0 && b_vis) {
            atomicAdd(&counters[C_STEPS], (u32)__popcll(b_vis));
            if (md > counters[C_MAXDEPTH]) atomicMax(&counters[C_MAXDEPTH], md);
        }
        __syncthreads();
    }
    // the workgroup that finishes last moves the level's range on (k_cas_decide, rv_cascade.hip)
    if (threadIdx.x == 0) {      // (what the workgroups wrote is for the next kernel; the fence orders this workgroup's C_NCHILD reservations -- atomics of other
        __threadfence();         //  waves, behind the barrier above -- in front of its ticket, so the last ticket holder reads the final count)
        if (atomicAdd(&counters[C_TICKET], 1u) == gridDim.x - 1) {
            if (hi > lo) counters[C_LEVELS]++;
            counters[C_LO] = hi; counters[C_HI] = atomicAdd(&counters[C_NCHILD], 0u);
            counters[C_TICKET] = 0;
        }
    }
}

// ---- undecided sub-indices from their text: up to k intervals, at most BN suffixes, 256 per workgroup (see k_cas_rank in rv_cascade.hip) ----
struct CmRoot { int64_t off; int32_t n, id; };
// first x < lim with txt[i + x] != txt[j + x], or lim: eight bytes per step (related genomes: a suffix agrees with its k - 1
// homologues for a hundred characters and more)
__device__ inline int cm_first_diff(const uint8_t *txt, int i, int j, int lim) {
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
// (counting sort by comparison, one suffix per thread: a workgroup takes 256 suffixes of a sub-index, so a sub-index of n suffixes is
// n / 256 workgroups doing n comparisons per thread -- as ONE workgroup per sub-index the largest of 44 undecided sub-indices of
// 10 x 5 Mbp, 8 000 suffixes whose homologues agree for hundreds of characters, took 30 ms)
constexpr int BN = 8192;
struct CmSlice { int32_t root, first; };
__device__ inline int cm_load_text(const CmRoot &root, const CmTabs &t, int k, const uint8_t *__restrict__ T0, uint8_t *txt, int *seg_lo, int64_t *seg_b, int stride = TB) {
    if (threadIdx.x == 0) {
        int at = 0;
        for (int s = 0; s < k; s++) {
            const int64_t b = t.b[(size_t)root.id * k + s], e = t.e[(size_t)root.id * k + s];
            seg_lo[s] = at; seg_b[s] = b; at += e > b ? (int)(e - b) : 0;
        }
        seg_lo[k] = at;
    }
    __syncthreads();
    const int n = seg_lo[k];
    for (int x = threadIdx.x; x < n; x += stride) {
        int s = 0;
        while (x >= seg_lo[s + 1]) s++;
        txt[x] = T0[seg_b[s] + (x - seg_lo[s])];
    }
    __syncthreads();
    return n;
}
// The ranks of a sub-index' suffixes by sorting instead of counting: a workgroup per sub-index sorts (first six bytes, suffix) words in LDS
// (bitonic: 91 steps for 8192 words), and a suffix is only compared in full with the ones that share its six bytes -- its homologues in the
// other samples and repeats.  Counting compared every suffix with every other: 62 000 vector instructions per wave, 1.35 ms for the 44
// undecided sub-indices (58 000 suffixes) of 10 x 5 Mbp.  The order is the one k_casm_rank counts out: a suffix ends with its interval, the
// shorter of two that agree to the end first, then the one in front; bytes behind the end count as zero in the six.
constexpr int RK_TB = 1024;
__global__ __launch_bounds__(RK_TB) void k_casm_rank_sort(const CmRoot *__restrict__ roots, CmTabs t, int k, const uint8_t *__restrict__ T0, u32 *__restrict__ ord) {
    __shared__ __attribute__((aligned(8))) uint8_t txt[BN + 24];
    __shared__ u64 sk[BN];
    __shared__ int seg_lo[RV_CASM_K + 1];
    __shared__ int64_t seg_b[RV_CASM_K];
    const CmRoot root = roots[blockIdx.x];
    const int n = cm_load_text(root, t, k, T0, txt, seg_lo, seg_b, RK_TB);
    int P = 2;
    while (P < n) P <<= 1;
    for (int i = threadIdx.x; i < P; i += RK_TB) {
        u64 e = ~0ull;
        if (i < n) {
            int si = 0;
            while (i >= seg_lo[si + 1]) si++;
            const int ri = seg_lo[si + 1] - i;
            u64 key = 0;
#pragma unroll
            for (int b = 0; b < 6; b++) key = (key << 8) | (b < ri ? (u64)txt[i + b] : 0ull);
            e = (key << 16) | (u64)i;
        }
        sk[i] = e;
    }
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int x = threadIdx.x; x < P / 2; x += RK_TB) {
                const int lo = 2 * x - (x & (stride - 1)), hi = lo + stride;
                const u64 a = sk[lo], b = sk[hi];
                if ((a > b) == ((lo & size) == 0)) { sk[lo] = b; sk[hi] = a; }
            }
        }
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += RK_TB) {
        const u64 e = sk[r];
        const u64 key = e >> 16;
        const int i = (int)(e & 0xffffu);
        int gs = r, ge = r + 1;
        while (gs > 0 && (sk[gs - 1] >> 16) == key) gs--;
        while (ge < n && (sk[ge] >> 16) == key) ge++;
        int cnt = 0;
        if (ge - gs > 1) {
            int si = 0;
            while (i >= seg_lo[si + 1]) si++;
            const int ri = seg_lo[si + 1] - i;
            for (int m = gs; m < ge; m++) {
                if (m == r) continue;
                const int j = (int)(sk[m] & 0xffffu);
                int sj = 0;
                while (j >= seg_lo[sj + 1]) sj++;
                const int rj = seg_lo[sj + 1] - j;
                const int lim = ri < rj ? ri : rj;
                const int x0 = lim < 6 ? lim : 6;      // (the six bytes agree, zeros behind an end included: both end inside them or neither does)
                const int x = x0 + cm_first_diff(txt, i + x0, j + x0, lim - x0);
                const bool j_less = (x < lim) ? (txt[j + x] < txt[i + x]) : ((rj < ri) | ((rj == ri) & (j < i)));
                cnt += j_less ? 1 : 0;
            }
        }
        ord[root.off + gs + cnt] = (u32)i;
    }
}
__global__ __launch_bounds__(TB) void k_casm_rank(const CmRoot *__restrict__ roots, const CmSlice *__restrict__ slices, CmTabs t, int k, const uint8_t *__restrict__ T0,
                                                  u32 *__restrict__ ord) {
    __shared__ __attribute__((aligned(8))) uint8_t txt[BN + 24];
    __shared__ int seg_lo[RV_CASM_K + 1];
    __shared__ int64_t seg_b[RV_CASM_K];
    const CmSlice sl = slices[blockIdx.x];
    const CmRoot root = roots[sl.root];
    const int n = cm_load_text(root, t, k, T0, txt, seg_lo, seg_b);
    const int i = sl.first + threadIdx.x;
    if (i >= n) return;
    int si = 0;
    while (i >= seg_lo[si + 1]) si++;
    const int ri = seg_lo[si + 1] - i;
    int cnt = 0, sj = 0;
    // the first eight bytes of the two suffixes from registers: mine once, the other's as a window that moves a byte per step (one uniform
    // LDS byte per comparison; two eight-byte reads at unknown alignment per comparison were sixteen byte reads: 2.6 ms at 10 x 5 Mbp)
    u64 ki = 0, wj = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) { ki |= (u64)txt[i + b] << (8 * b); wj |= (u64)txt[b] << (8 * b); }
    for (int j0 = 0; j0 < n; j0 += 8) {      // (the bytes that enter the window come eight at a time from one aligned load: no step waits for LDS)
        u64 nxt = *reinterpret_cast<const u64 *>(txt + j0 + 8);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int j = j0 + e;
            if (j < n) {
                while (j >= seg_lo[sj + 1]) sj++;
                const int rj = seg_lo[sj + 1] - j;
                const int lim = ri < rj ? ri : rj;
                const u64 d = ki ^ wj;
                int x = d ? (__builtin_ctzll(d) >> 3) : 8;
                bool j_less;
                if (x < 8 && x < lim) j_less = (u32)((wj >> (8 * x)) & 0xffu) < (u32)((ki >> (8 * x)) & 0xffu);
                else {
                    if (x >= 8 && lim > 8 && j != i) x = 8 + cm_first_diff(txt, i + 8, j + 8, lim - 8);
                    else if (j == i) x = lim;
                    j_less = (x < lim) ? (txt[j + x] < txt[i + x]) : ((rj < ri) | ((rj == ri) & (j < i)));
                }
                cnt += j_less ? 1 : 0;
            }
            wj = (wj >> 8) | (nxt << 56);
            nxt >>= 8;
        }
    }
    ord[root.off + cnt] = (u32)i;
}
__global__ __launch_bounds__(TB) void k_casm_emit(const CmRoot *__restrict__ roots, const CmSlice *__restrict__ slices, CmTabs t, int k, const uint8_t *__restrict__ T0,
                                                  const u32 *__restrict__ ord, sa_t *__restrict__ SA, lcp_t *__restrict__ LCP, uint8_t *__restrict__ BWT, int64_t nsep0,
                                                  const sa_t *__restrict__ root_b) {
    __shared__ uint8_t txt[BN + 8];
    __shared__ int seg_lo[RV_CASM_K + 1];
    __shared__ int64_t seg_b[RV_CASM_K];
    const CmSlice sl = slices[blockIdx.x];
    const CmRoot root = roots[sl.root];
    const int n = cm_load_text(root, t, k, T0, txt, seg_lo, seg_b);
    const int r = sl.first + threadIdx.x;
    if (r >= n) return;
    const int i = (int)ord[root.off + r];
    int si = 0;
    while (i >= seg_lo[si + 1]) si++;
    const int ri = seg_lo[si + 1] - i;
    u32 l = 0;
    if (r > 0) {
        const int j = (int)ord[root.off + r - 1];
        int sj = 0;
        while (j >= seg_lo[sj + 1]) sj++;
        const int rj = seg_lo[sj + 1] - j;
        const int lim = ri < rj ? ri : rj;
        int x = 0;
        while (x < lim) { const uint8_t c = txt[i + x]; if (c != txt[j + x] || c == '$' || c == 'N') break; x++; }
        l = (u32)x;
    }
    const int64_t gp = seg_b[si] + (i - seg_lo[si]);
    uint8_t ch = gp > 0 ? T0[gp - 1] : (uint8_t)'$';
    const bool behind_anchor = i == seg_lo[si] && seg_b[si] > (int64_t)root_b[si];
    if (behind_anchor && ch >= 'A' && ch <= 'Z') ch += 32;
    const int64_t o = root.off + r;
    SA[o] = (sa_t)gp; LCP[o] = (lcp_t)l; BWT[o] = (uint8_t)(ch | (gp > nsep0 ? RV_BWT_SIDE : 0u));
}
// ---- undecided sub-indices of more than BN suffixes: the same order through global memory ----
// (a sample that lost a kilobase or more to a deletion leaves the other k - 1 samples' kilobases behind as ONE sub-index that lacks a sample: above BN
// ranks from a few hundred bases per sample on; with the simulator's indel model 10 x 5 Mbp holds some twenty of them, and the run used to start again
// at the top in the level pipeline: 135 instead of 8 ms.)  The suffixes of all such sub-indices get a word (sub-index, first `kb` bytes, zeros behind the
// end) and go through one radix sort; a suffix is compared in full only with the ones that share its word -- its homologues and repeats --, on the
// pristine text in HBM, eight bytes a step.  The order is k_casm_rank_sort's.  Low-complexity text (an N run, a tandem array: thousands of suffixes that
// share their first bytes and agree for kilobases) is not for this: a group of more than BIG_GROUP suffixes, or a suffix that has spent BIG_STEPS steps,
// raises a flag, whoever sees the flag stops, and the cascade gives up as it used to -- within a millisecond, not after 10^12 comparisons.
struct CmBig { int64_t off, offL; int32_t n, id; };      // (off: where its arrays start among the rebuilt ones; offL: where its suffixes start in the sort)
constexpr int BIG_GROUP = 256;                 // (homologues: one per sample; repeats inside one sub-index on top)
constexpr int64_t BIG_STEPS = (int64_t)1 << 13;      // eight-byte steps a suffix may spend on its comparisons: 64 KB
__device__ inline u64 cm_ld8(const uint8_t *p) { u64 a; __builtin_memcpy(&a, p, 8); return a; }
__global__ __launch_bounds__(TB) void k_casmb_keys(const CmBig *__restrict__ roots, const CmSlice *__restrict__ slices, CmTabs t, int k, const uint8_t *__restrict__ T0, int kb,
                                                   u64 *__restrict__ keys, u64 *__restrict__ vals) {
    const CmSlice sl = slices[blockIdx.x];
    const CmBig root = roots[sl.root];
    const int64_t i = (int64_t)sl.first + threadIdx.x;
    if (i >= root.n) return;
    int64_t at = 0, gp = 0, ri = 0;
    for (int s = 0; s < k; s++) {
        const int64_t b = (int64_t)t.b[(size_t)root.id * k + s], e = (int64_t)t.e[(size_t)root.id * k + s];
        const int64_t len = e > b ? e - b : 0;
        if (i >= at && i < at + len) { gp = b + (i - at); ri = e - gp; }
        at += len;
    }
    u64 key = 0;
    for (int b = 0; b < kb; b++) key = (key << 8) | (b < ri ? (u64)T0[gp + b] : 0ull);
    keys[root.offL + i] = ((u64)(u32)sl.root << (8 * kb)) | key;
    vals[root.offL + i] = ((u64)ri << 32) | (u64)(u32)gp;
}
__global__ __launch_bounds__(TB) void k_casmb_place(const CmBig *__restrict__ roots, const u64 *__restrict__ keys, const u64 *__restrict__ vals, int64_t mL, int kb,
                                                    const uint8_t *__restrict__ T0, sa_t *__restrict__ SA, u32 *__restrict__ RI, u32 *__restrict__ err) {
    const int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r >= mL) return;
    if (*(volatile u32 *)err) return;
    const u64 key = keys[r];
    const CmBig root = roots[key >> (8 * kb)];
    int64_t gs = r, ge = r + 1;
    while (gs > 0 && r - gs <= BIG_GROUP && keys[gs - 1] == key) gs--;
    while (ge < mL && ge - r <= BIG_GROUP && keys[ge] == key) ge++;
    if (ge - gs > BIG_GROUP) { atomicOr(err, 1u); return; }
    const u64 vi = vals[r];
    const int64_t pi = (int64_t)(u32)vi, ri = (int64_t)(vi >> 32);
    int64_t cnt = 0, steps = 0;
    for (int64_t m = gs; m < ge; m++) {
        if (m == r) continue;
        if (steps > BIG_STEPS || *(volatile u32 *)err) { atomicOr(err, 1u); return; }
        const u64 vj = vals[m];
        const int64_t pj = (int64_t)(u32)vj, rj = (int64_t)(vj >> 32);
        const int64_t lim = ri < rj ? ri : rj;
        int64_t x = lim < kb ? lim : kb;      // (the first kb bytes agree, zeros behind an end included: both end inside them or neither does)
        bool diff = false;
        while (x + 8 <= lim) {
            const u64 a = cm_ld8(T0 + pi + x), b = cm_ld8(T0 + pj + x);
            if (a != b) { x += __builtin_ctzll(a ^ b) >> 3; diff = true; break; }
            x += 8;
            if (++steps > BIG_STEPS) break;
        }
        if (steps > BIG_STEPS) { atomicOr(err, 1u); return; }
        if (!diff) while (x < lim && T0[pi + x] == T0[pj + x]) x++;
        const bool j_less = (x < lim) ? (T0[pj + x] < T0[pi + x]) : ((rj < ri) | ((rj == ri) & (pj < pi)));
        cnt += j_less ? 1 : 0;
    }
    const int64_t o = root.off + (gs - root.offL) + cnt;
    SA[o] = (sa_t)pi; RI[o] = (u32)ri;
}
// bytes of `a` that differ from `b` or are '$' or 'N': the index of the first one, 8 if none
__device__ inline int cm_first_stop(u64 a, u64 b) {
    const u64 lo = 0x0101010101010101ull, hi = 0x8080808080808080ull;
    const u64 d = a ^ b, vn = a ^ (lo * (u64)'N'), vs = a ^ (lo * (u64)'$');
    int x = d ? (__builtin_ctzll(d) >> 3) : 8;
    const u64 zn = (vn - lo) & ~vn & hi, zs = (vs - lo) & ~vs & hi;      // (the lowest flagged byte is exact)
    if (zn) { const int y = __builtin_ctzll(zn) >> 3; x = y < x ? y : x; }
    if (zs) { const int y = __builtin_ctzll(zs) >> 3; x = y < x ? y : x; }
    return x;
}
__global__ __launch_bounds__(TB) void k_casmb_emit(const CmBig *__restrict__ roots, const CmSlice *__restrict__ slices, CmTabs t, int k, const uint8_t *__restrict__ T0,
                                                   const u32 *__restrict__ RI, const sa_t *__restrict__ SA, lcp_t *__restrict__ LCP, uint8_t *__restrict__ BWT, int64_t nsep0,
                                                   const sa_t *__restrict__ root_b, const u32 *__restrict__ err) {
    if (*err) return;      // (k_casmb_place stopped half way: there is no order to read)
    const CmSlice sl = slices[blockIdx.x];
    const CmBig root = roots[sl.root];
    const int64_t r = (int64_t)sl.first + threadIdx.x;
    if (r >= root.n) return;
    const int64_t o = root.off + r;
    const int64_t gp = (int64_t)SA[o], ri = (int64_t)RI[o];
    int64_t l = 0;
    if (r > 0) {
        const int64_t gq = (int64_t)SA[o - 1], rj = (int64_t)RI[o - 1];
        const int64_t lim = ri < rj ? ri : rj;
        int64_t x = 0;
        bool stop = false;
        while (x + 8 <= lim) {
            const int y = cm_first_stop(cm_ld8(T0 + gp + x), cm_ld8(T0 + gq + x));
            x += y;
            if (y < 8) { stop = true; break; }
        }
        if (!stop) while (x < lim) { const uint8_t c = T0[gp + x]; if (c != T0[gq + x] || c == '$' || c == 'N') break; x++; }
        l = x;
    }
    uint8_t ch = gp > 0 ? T0[gp - 1] : (uint8_t)'$';
    bool behind_anchor = false;
    for (int s = 0; s < k; s++) {
        const int64_t b = (int64_t)t.b[(size_t)root.id * k + s];
        behind_anchor |= gp == b && (int64_t)t.e[(size_t)root.id * k + s] > b && b > (int64_t)root_b[s];
    }
    if (behind_anchor && ch >= 'A' && ch <= 'Z') ch += 32;
    LCP[o] = (lcp_t)l; BWT[o] = (uint8_t)(ch | (gp > nsep0 ? RV_BWT_SIDE : 0u));
}
__global__ __launch_bounds__(TB) void k_casm_unlower(uint8_t *__restrict__ T, const uint8_t *__restrict__ T0, const u32 *__restrict__ an_l, const sa_t *__restrict__ an_pos, u32 nranges) {
    const u32 e = (u32)(((int64_t)blockIdx.x * TB + threadIdx.x) >> 6);
    if (e >= nranges) return;
    const int64_t lo = (int64_t)an_pos[e];
    const int64_t l = (int64_t)an_l[e];
    for (int64_t x = threadIdx.x & 63; x < l; x += 64) T[lo + x] = T0[lo + x];
}
__global__ __launch_bounds__(TB) void k_casm_lower(uint8_t *__restrict__ T, const u32 *__restrict__ an_l, const sa_t *__restrict__ an_pos, u32 nranges) {
    const u32 e = (u32)(((int64_t)blockIdx.x * TB + threadIdx.x) >> 6);
    if (e >= nranges) return;
    const int64_t lo = (int64_t)an_pos[e];
    const int64_t l = (int64_t)an_l[e];      // (an_l expanded per range by the caller's indexing: see the launch)
    for (int64_t x = threadIdx.x & 63; x < l; x += 64) { const uint8_t c = T[lo + x]; if (c >= 'A' && c <= 'Z') T[lo + x] = c + 32; }
}
// the rows of the undecided sub-indices side by side (ids, depths, begins, ends): one copy to the host instead of three per row
__global__ __launch_bounds__(TB) void k_casm_rows(const u32 *__restrict__ und, u32 U, CmTabs t, int k, u32 *__restrict__ o_id, int32_t *__restrict__ o_dep,
                                                  sa_t *__restrict__ o_b, sa_t *__restrict__ o_e) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= U * (u32)k) return;
    const u32 x = i / (u32)k, s = i - x * (u32)k, id = und[x];
    o_b[i] = t.b[(size_t)id * k + s]; o_e[i] = t.e[(size_t)id * k + s];
    if (s == 0) { o_id[x] = id; o_dep[x] = t.depth[id]; }
}
__global__ __launch_bounds__(TB) void k_casm_expand_l(const u32 *__restrict__ an_l, u32 na, int k, u32 *__restrict__ out) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i < na * (u32)k) out[i] = an_l[i / (u32)k];
}

int bitlen64m(u64 x) { int b = 0; while (x) { b++; x >>= 1; } return b; }

}  // namespace

// ---- the level loops of several jobs as ONE (rv_batch_*) -------------------------------------------------------------------------
// The cascade's level kernels never touch SA or LCP: they work on a job's list of full matches, its witness list and the table of sub-indices.  A job
// of 5 x 5 Mbp spends a quarter of its kernel time in them -- 40 levels x 3 dependent launches of a few microseconds -- and independent jobs on streams
// of their own hide little of that behind each other (four hardware queues, a tiny kernel waits behind whatever large one shares its queue).  So the
// jobs of a batch meet here: every job builds its index and its two lists as always, on its own stream and host thread; the last to arrive concatenates
// the lists (matches keep their job's coordinates -- a match only ever meets sub-indices of its own job, through c_child), makes one root per job, runs
// the level loop once for all of them -- a sub-index inherits its root's job, an anchor is written with it -- lower-cases every job's anchors, and hands
// each job its anchors and undecided sub-indices; every job then rebuilds those and finishes on its own stream.  Anything unusual (different sample
// counts or minl in the batch, full tables) sends every job back to its own loop.
struct RvBatchJob {
    // what a job brings
    int k = 0; u32 minl = 0; u32 M = 0, NW = 0, ccap = 0, acap = 0;
    const u32 *c_len = nullptr; const sa_t *c_pos = nullptr; const sa_t *w_pos = nullptr; const u32 *w_val = nullptr;
    const sa_t *nsep = nullptr; uint8_t *dT = nullptr;
    std::vector<sa_t> rb, re, sep;
    int64_t big_min = 0, big_root = 0, big_total = 0; bool no_big = false;
    hipStream_t stream = nullptr; hipEvent_t ready = nullptr;
    // what it gets back
    bool ok = false, too_big = false;
    u32 levels = 0, nchild = 0, steps = 0, maxdepth = 0, maxn = 0;
    const u32 *d_anl = nullptr; const sa_t *d_anp = nullptr; u32 na = 0;      // its anchors: a stretch of the group's arrays (device)
    std::vector<u32> und_id; std::vector<int32_t> und_dep; std::vector<sa_t> und_b, und_e;
};
struct RvBatchGroup {
    std::mutex mu; std::condition_variable cv;
    int total = 0, settled = 0;
    std::vector<RvBatchJob *> arrived;
    bool leader_taken = false, done = false;
    DBuf d[20]; HBuf hs, hs2;
    CmTabs t{};
    hipEvent_t ev_done = nullptr;
    int device = 0;
    int64_t joint_runs = 0, joint_jobs = 0;
};

// which job owns entry i of a concatenated list (off: J + 1 ascending offsets)
__device__ inline int cmj_owner(const u32 *__restrict__ off, int J, u32 i) {
    int lo = 0, hi = J;
    while (lo + 1 < hi) { const int mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(TB) void k_cmj_fill(u32 M, u32 NW, const u32 *__restrict__ moff, const u32 *__restrict__ woff, int J, int k, const sa_t *__restrict__ seps /* J x (k - 1) */,
                                                 const sa_t *__restrict__ w_pos, u32 *__restrict__ c_child, u32 *__restrict__ w_child, uint8_t *__restrict__ w_smp) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i < M) c_child[i] = (u32)cmj_owner(moff, J, i);
    if (i < NW) {
        const int j = cmj_owner(woff, J, i);
        w_child[i] = (u32)j;
        w_smp[i] = (uint8_t)cm_sample(seps + (size_t)j * (k - 1), k, w_pos[i]);
    }
}
__global__ void k_cmj_init(CmTabs t, int k, const sa_t *__restrict__ roots_b, const sa_t *__restrict__ roots_e, int J, u32 *__restrict__ counters) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < J) {
        for (int s = 0; s < k; s++) { t.b[(size_t)j * k + s] = roots_b[(size_t)j * k + s]; t.e[(size_t)j * k + s] = roots_e[(size_t)j * k + s]; }
        t.best[j] = 0; t.rmax[j] = 0; t.depth[j] = 0; t.state[j] = 0; t.lead[j] = NONE; t.trail[j] = NONE; t.ql[j] = 0; t.job[j] = (u32)j;
    }
    if (j == 0) {
        counters[C_NCHILD] = (u32)J; counters[C_NUND] = 0; counters[C_ERR] = 0; counters[C_MAXN] = 0; counters[C_LO] = 0; counters[C_HI] = (u32)J; counters[C_LEVELS] = 0;
        counters[C_NANCH] = 0; counters[C_STEPS] = 0; counters[C_MAXDEPTH] = 0; counters[C_TICKET] = 0; counters[C_BIGTOT] = 0; counters[C_BIGTOT + 1] = 0;
    }
}
// per job: sub-indices, the largest depth among all of them, among those that were visited (not undecided)
__global__ __launch_bounds__(TB) void k_cmj_stats(CmTabs t, u32 nchild, u32 *__restrict__ out /* J x 4 */) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= nchild) return;
    const u32 j = t.job[i], dp = (u32)t.depth[i];
    atomicAdd(&out[j * 4], 1u);
    if (dp > out[j * 4 + 1]) atomicMax(&out[j * 4 + 1], dp);
    if (t.state[i] != 3u && dp > out[j * 4 + 2]) atomicMax(&out[j * 4 + 2], dp);
}
__global__ __launch_bounds__(TB) void k_cmj_rows(const u32 *__restrict__ und, u32 U, CmTabs t, int k, u32 *__restrict__ o_id, int32_t *__restrict__ o_dep, u32 *__restrict__ o_job,
                                                 sa_t *__restrict__ o_b, sa_t *__restrict__ o_e) {
    const u32 i = blockIdx.x * TB + threadIdx.x;
    if (i >= U * (u32)k) return;
    const u32 x = i / (u32)k, s = i - x * (u32)k, id = und[x];
    o_b[i] = t.b[(size_t)id * k + s]; o_e[i] = t.e[(size_t)id * k + s];
    if (s == 0) { o_id[x] = id; o_dep[x] = t.depth[id]; o_job[x] = t.job[id]; }
}
// reveal.c:1230-1234 for one job's anchors (its stretch of the group's arrays): a wave per member
__global__ __launch_bounds__(TB) void k_cmj_lower(uint8_t *__restrict__ T, const u32 *__restrict__ an_l, const sa_t *__restrict__ an_pos, u32 na, int k) {
    const u32 e = (u32)(((int64_t)blockIdx.x * TB + threadIdx.x) >> 6);
    if (e >= na * (u32)k) return;
    const int64_t lo = (int64_t)an_pos[e], l = (int64_t)an_l[e / (u32)k];
    for (int64_t x = threadIdx.x & 63; x < l; x += 64) { const uint8_t c = T[lo + x]; if (c >= 'A' && c <= 'Z') T[lo + x] = c + 32; }
}

static int batch_joint_phase(RvBatchGroup *g, Workspace &ws);

// a job arrives with its lists; returns when the joint phase is over (job->ok says whether it served this job)
static void batch_join(RvBatchGroup *g, RvBatchJob *job, Workspace &ws) {
    std::unique_lock<std::mutex> lk(g->mu);
    g->arrived.push_back(job);
    g->settled++;
    for (;;) {
        if (g->done) return;
        if (g->settled >= g->total && !g->leader_taken) {
            g->leader_taken = true;
            lk.unlock();
            if (batch_joint_phase(g, ws) != 0) for (RvBatchJob *x : g->arrived) x->ok = false;      // (everybody falls back to its own loop)
            lk.lock();
            g->done = true;
            g->cv.notify_all();
            return;
        }
        g->cv.wait(lk);
    }
}
void rv_batch_group_leave(RvBatchGroup *g) {
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->mu);
    g->settled++;
    if (g->settled >= g->total) g->cv.notify_all();      // (a job that is waiting takes the joint phase)
}
RvBatchGroup *rv_batch_group_new(int device) { RvBatchGroup *g = new RvBatchGroup(); g->device = device; return g; }
void rv_batch_group_begin(RvBatchGroup *g, int total) {
    std::lock_guard<std::mutex> lk(g->mu);
    g->total = total; g->settled = 0; g->arrived.clear(); g->leader_taken = false; g->done = false;
}
void rv_batch_group_free(RvBatchGroup *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    for (auto &b : g->d) b.release();
    g->hs.release(); g->hs2.release();
    if (g->ev_done) (void)hipEventDestroy(g->ev_done);
    delete g;
}
void rv_batch_group_info(const RvBatchGroup *g, int64_t *out) { out[0] = g->joint_runs; out[1] = g->joint_jobs; }

static int batch_joint_phase(RvBatchGroup *g, Workspace &ws) {
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<RvBatchJob *> &jobs = g->arrived;
    const int J = (int)jobs.size();
    for (RvBatchJob *x : jobs) x->ok = false;
    if (J < 2) return 0;
    const int k = jobs[0]->k; const u32 minl = jobs[0]->minl;
    for (RvBatchJob *x : jobs) if (x->k != k || x->minl != minl) return 0;
    hipStream_t q = ws.stream;      // the leader's stream
    RV_HIP(hipSetDevice(g->device));
    if (!g->ev_done) RV_HIP(hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming));
    std::vector<u32> moff((size_t)J + 1, 0), woff((size_t)J + 1, 0);
    u64 ccap64 = 0, acap64 = 0;
    for (int j = 0; j < J; j++) {
        moff[(size_t)j + 1] = moff[(size_t)j] + jobs[(size_t)j]->M; woff[(size_t)j + 1] = woff[(size_t)j] + jobs[(size_t)j]->NW;
        ccap64 += jobs[(size_t)j]->ccap; acap64 += jobs[(size_t)j]->acap;
        if ((u64)moff[(size_t)j] + jobs[(size_t)j]->M > 0x7fffffffull || (u64)woff[(size_t)j] + jobs[(size_t)j]->NW > 0x7fffffffull) return 0;
    }
    if (ccap64 > 0x7fffffffull || acap64 > 0x7fffffffull) return 0;
    const u32 M = moff[(size_t)J], NW = woff[(size_t)J], ccap = (u32)ccap64, acap = (u32)acap64;
    if (M == 0) return 0;
    DBuf &bcl = g->d[0], &bcp = g->d[1], &bcc = g->d[2], &bwp = g->d[3], &bwv = g->d[4], &bwc = g->d[5], &bws = g->d[6], &btb = g->d[7], &bctr = g->d[8], &bund = g->d[9],
         &banl = g->d[10], &banp = g->d[11], &bsm = g->d[13], &brows = g->d[14], &bst = g->d[15];
    RV_TRY(bcl.reserve((size_t)M * 4)); RV_TRY(bcp.reserve((size_t)M * k * sizeof(sa_t))); RV_TRY(bcc.reserve((size_t)M * 4));
    RV_TRY(bwp.reserve((size_t)(NW + 1) * sizeof(sa_t))); RV_TRY(bwv.reserve((size_t)(NW + 1) * 4)); RV_TRY(bwc.reserve((size_t)(NW + 1) * 4)); RV_TRY(bws.reserve((size_t)NW + 64));
    const size_t per = (size_t)3 * k * sizeof(sa_t) + 8 + 7 * 4;
    RV_TRY(btb.reserve(per * ccap + 256)); RV_TRY(bctr.reserve(256)); RV_TRY(bund.reserve((size_t)ccap * 4));
    RV_TRY(banl.reserve((size_t)acap * 4)); RV_TRY(banp.reserve((size_t)acap * k * sizeof(sa_t)));
    RV_TRY(bst.reserve((size_t)J * 16 + 64));
    CmTabs t;
    {
        uint8_t *p = btb.as<uint8_t>();
        t.best = (u64 *)p; p += (size_t)ccap * 8;
        t.b = (sa_t *)p; p += (size_t)ccap * k * sizeof(sa_t); t.e = (sa_t *)p; p += (size_t)ccap * k * sizeof(sa_t); t.q = (sa_t *)p; p += (size_t)ccap * k * sizeof(sa_t);
        t.rmax = (u32 *)p; p += (size_t)ccap * 4; t.depth = (int32_t *)p; p += (size_t)ccap * 4; t.state = (u32 *)p; p += (size_t)ccap * 4;
        t.lead = (u32 *)p; p += (size_t)ccap * 4; t.trail = (u32 *)p; p += (size_t)ccap * 4; t.ql = (u32 *)p; p += (size_t)ccap * 4; t.job = (u32 *)p;
    }
    g->t = t;
    // small tables in one pinned buffer: list offsets, roots, separators, text pointers, skip flags
    const size_t o_m = 0, o_w = o_m + ((size_t)J + 1) * 4, o_rb = (o_w + ((size_t)J + 1) * 4 + 15) & ~(size_t)15, o_re = o_rb + (size_t)J * k * sizeof(sa_t),
                 o_sep = o_re + (size_t)J * k * sizeof(sa_t), o_ab = (o_sep + (size_t)J * (k - 1) * sizeof(sa_t) + 15) & ~(size_t)15, sm_bytes = o_ab + ((size_t)J + 1) * 4 + 64;
    RV_TRY(g->hs.reserve(sm_bytes)); RV_TRY(bsm.reserve(sm_bytes));
    uint8_t *hp = g->hs.as<uint8_t>();
    memset(hp, 0, sm_bytes);
    memcpy(hp + o_m, moff.data(), ((size_t)J + 1) * 4); memcpy(hp + o_w, woff.data(), ((size_t)J + 1) * 4);
    for (int j = 0; j < J; j++) {
        memcpy(hp + o_rb + (size_t)j * k * sizeof(sa_t), jobs[(size_t)j]->rb.data(), (size_t)k * sizeof(sa_t));
        memcpy(hp + o_re + (size_t)j * k * sizeof(sa_t), jobs[(size_t)j]->re.data(), (size_t)k * sizeof(sa_t));
        memcpy(hp + o_sep + (size_t)j * (k - 1) * sizeof(sa_t), jobs[(size_t)j]->sep.data(), (size_t)(k - 1) * sizeof(sa_t));
    }
    std::vector<u32> abase((size_t)J + 1, 0);      // a job's anchors: its own stretch of the arrays, as long as its own area would have been
    for (int j = 0; j < J; j++) abase[(size_t)j + 1] = abase[(size_t)j] + jobs[(size_t)j]->acap;
    memcpy(hp + o_ab, abase.data(), ((size_t)J + 1) * 4);
    DBuf &bjc = g->d[16];
    RV_TRY(bjc.reserve((size_t)J * 256 + 64));
    RV_HIP(hipMemsetAsync(bjc.p, 0, (size_t)J * 256, q));
    uint8_t *dsm = bsm.as<uint8_t>();
    RV_HIP(hipMemcpyAsync(dsm, hp, sm_bytes, hipMemcpyHostToDevice, q));
    u32 *counters = bctr.as<u32>();
    for (int j = 0; j < J; j++) {      // every job's lists are complete on its own stream: wait for them, then copy them behind each other
        RvBatchJob &x = *jobs[(size_t)j];
        RV_HIP(hipStreamWaitEvent(q, x.ready, 0));
        if (x.M) {
            RV_HIP(hipMemcpyAsync(bcl.as<u32>() + moff[(size_t)j], x.c_len, (size_t)x.M * 4, hipMemcpyDeviceToDevice, q));
            RV_HIP(hipMemcpyAsync(bcp.as<sa_t>() + (size_t)moff[(size_t)j] * k, x.c_pos, (size_t)x.M * k * sizeof(sa_t), hipMemcpyDeviceToDevice, q));
        }
        if (x.NW) {
            RV_HIP(hipMemcpyAsync(bwp.as<sa_t>() + woff[(size_t)j], x.w_pos, (size_t)x.NW * sizeof(sa_t), hipMemcpyDeviceToDevice, q));
            RV_HIP(hipMemcpyAsync(bwv.as<u32>() + woff[(size_t)j], x.w_val, (size_t)x.NW * 4, hipMemcpyDeviceToDevice, q));
        }
    }
    hipLaunchKernelGGL(k_cmj_fill, dim3((unsigned)ceil_div((int64_t)std::max(M, NW), TB)), dim3(TB), 0, q, M, NW, (const u32 *)(dsm + o_m), (const u32 *)(dsm + o_w), J, k,
                       (const sa_t *)(dsm + o_sep), (const sa_t *)bwp.as<sa_t>(), bcc.as<u32>(), bwc.as<u32>(), bws.as<uint8_t>());
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cmj_init, dim3((unsigned)ceil_div((int64_t)J, 64)), dim3(64), 0, q, t, k, (const sa_t *)(dsm + o_rb), (const sa_t *)(dsm + o_re), J, counters);
    RV_LAUNCH_CHECK();
    const unsigned agrid = (unsigned)(ceil_div((int64_t)M, TB) + ceil_div((int64_t)NW, TB));
    const int batch = std::max(1, (int)ws.opt.cascade_batch);
    // (the size above which an undecided sub-index goes through global memory when it is rebuilt: the jobs' own setting -- it only feeds two counters here)
    const u32 big_min = (u32)std::min<int64_t>(jobs[0]->big_min, 0x7fffffff);
    u32 hc[16];
    int queued = 0;
    for (;;) {
        for (int b = 0; b < batch; b++, queued++) {
            hipLaunchKernelGGL(k_casm_assign, dim3(agrid), dim3(TB), 0, q, (const sa_t *)bcp.as<sa_t>(), (const u32 *)bcl.as<u32>(), bcc.as<u32>(), M, (const sa_t *)bwp.as<sa_t>(),
                               (const u32 *)bwv.as<u32>(), bwc.as<u32>(), NW, t, k, (const sa_t *)nullptr, (int64_t)minl, queued == 0 ? 1 : 0, (const uint8_t *)bws.as<uint8_t>());
            RV_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_casm_winner, dim3((unsigned)ceil_div((int64_t)M, TB)), dim3(TB), 0, q, (const sa_t *)bcp.as<sa_t>(), (const u32 *)bcl.as<u32>(), (const u32 *)bcc.as<u32>(), M, t, k,
                               (int64_t)minl);
            RV_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_casm_decide, dim3(192), dim3(TB), 0, q, t, k, minl, counters, ccap, bund.as<u32>(), big_min, banl.as<u32>(), banp.as<sa_t>(), acap, (u32 *)nullptr,
                               bjc.as<u32>(), (const u32 *)(dsm + o_ab));
            RV_LAUNCH_CHECK();
        }
        RV_TRY(rv_read_back(ws, hc, counters, sizeof hc));
        if (hc[C_ERR]) return 0;                         // (full tables, or worse: every job back to its own loop, which reports it)
        if (hc[C_HI] == hc[C_LO]) break;
        if (queued > 1000000) return 0;
    }
    const u32 NCH = hc[C_NCHILD], U = hc[C_NUND];
    // per job: how many sub-indices, how deep
    RV_HIP(hipMemsetAsync(bst.p, 0, (size_t)J * 16, q));
    hipLaunchKernelGGL(k_cmj_stats, dim3((unsigned)ceil_div((int64_t)NCH, TB)), dim3(TB), 0, q, t, NCH, bst.as<u32>());
    RV_LAUNCH_CHECK();
    // the undecided sub-indices' rows, the per-job figures and anchor counts to the host (one pinned buffer, one wait); the anchors themselves stay where
    // they are: every job fetches its own stretch on its own stream afterwards
    const size_t rowbytes = (size_t)U * 12 + (size_t)U * k * sizeof(sa_t) * 2;
    const size_t h_st = 0, h_jc = ((size_t)J * 16 + 15) & ~(size_t)15, h_rows = h_jc + (size_t)J * 256;
    RV_TRY(g->hs2.reserve(h_rows + rowbytes + 64));
    uint8_t *h2 = g->hs2.as<uint8_t>();
    RV_HIP(hipMemcpyAsync(h2 + h_st, bst.p, (size_t)J * 16, hipMemcpyDeviceToHost, q));
    RV_HIP(hipMemcpyAsync(h2 + h_jc, bjc.p, (size_t)J * 256, hipMemcpyDeviceToHost, q));
    if (U) {
        RV_TRY(brows.reserve(rowbytes + 64));
        u32 *d_id = brows.as<u32>(); int32_t *d_dep = (int32_t *)(d_id + U); u32 *d_job = (u32 *)(d_dep + U);
        sa_t *d_b = (sa_t *)(d_job + U), *d_e = d_b + (size_t)U * k;
        hipLaunchKernelGGL(k_cmj_rows, dim3((unsigned)ceil_div((int64_t)U * k, TB)), dim3(TB), 0, q, (const u32 *)bund.as<u32>(), U, t, k, d_id, d_dep, d_job, d_b, d_e);
        RV_LAUNCH_CHECK();
        RV_HIP(hipMemcpyAsync(h2 + h_rows, brows.p, rowbytes, hipMemcpyDeviceToHost, q));
    }
    RV_HIP(hipStreamSynchronize(q));
    const u32 *st = (const u32 *)(h2 + h_st), *jc = (const u32 *)(h2 + h_jc);
    for (int j = 0; j < J; j++) {
        RvBatchJob &x = *jobs[(size_t)j];
        x.nchild = st[(size_t)j * 4]; x.levels = st[(size_t)j * 4 + 1] + 1; x.maxdepth = st[(size_t)j * 4 + 2];
        x.na = jc[(size_t)j * 64];
        if (x.na > x.acap) return 0;      // (its stretch was too short: every job back to its own loop, which reports it)
        x.d_anl = banl.as<u32>() + abase[(size_t)j]; x.d_anp = banp.as<sa_t>() + (size_t)abase[(size_t)j] * k;
        x.und_id.clear(); x.und_dep.clear(); x.und_b.clear(); x.und_e.clear(); x.too_big = false; x.maxn = 0;
    }
    std::vector<unsigned long long> bigtot((size_t)J, 0);
    if (U) {
        const u32 *r_id = (const u32 *)(h2 + h_rows); const int32_t *r_dep = (const int32_t *)(r_id + U); const u32 *r_job = (const u32 *)(r_dep + U);
        const sa_t *r_b = (const sa_t *)(r_job + U), *r_e = r_b + (size_t)U * k;
        for (u32 u2 = 0; u2 < U; u2++) {
            const u32 j = r_job[u2];
            if (j >= (u32)J) return 0;
            RvBatchJob &x = *jobs[j];
            x.und_id.push_back(r_id[u2]); x.und_dep.push_back(r_dep[u2]);
            int64_t sz = 0;
            for (int s2 = 0; s2 < k; s2++) {
                x.und_b.push_back(r_b[(size_t)u2 * k + s2]); x.und_e.push_back(r_e[(size_t)u2 * k + s2]);
                if (r_e[(size_t)u2 * k + s2] > r_b[(size_t)u2 * k + s2]) sz += (int64_t)(r_e[(size_t)u2 * k + s2] - r_b[(size_t)u2 * k + s2]);
            }
            if (sz > x.big_min) { x.maxn = std::max<u32>(x.maxn, (u32)std::min<int64_t>(sz, 0xFFFFFFFFll)); bigtot[j] += (unsigned long long)sz; }
        }
    }
    for (int j = 0; j < J; j++) {
        RvBatchJob &x = *jobs[(size_t)j];
        x.steps = x.nchild - (u32)x.und_id.size();
        x.too_big = x.no_big ? x.maxn > (u32)BN : ((int64_t)x.maxn > x.big_root || (int64_t)bigtot[(size_t)j] > x.big_total);
        // (a job rv_cascade_multi_run gives up on after its loop keeps its text: too large an undecided sub-index, or nothing decided at all)
        const bool nothing = x.und_id.size() == 1 && x.nchild == 1;
        if (x.na && !x.too_big && !nothing) {
            hipLaunchKernelGGL(k_cmj_lower, dim3((unsigned)ceil_div((int64_t)x.na * k * 64, TB)), dim3(TB), 0, q, x.dT, x.d_anl, x.d_anp, x.na, k);
            RV_LAUNCH_CHECK();
        }
    }
    RV_HIP(hipEventRecord(g->ev_done, q));
    for (RvBatchJob *x : jobs) x->ok = true;
    g->joint_runs++; g->joint_jobs += J;
    if (ws.opt.cascade_log)
        fprintf(stderr, "cascade (batch): %d jobs of %d samples in one loop: %u matches, %u witnesses, %u levels, %u sub-indices, %u undecided; %.2f ms from the last arrival to the hand-back\n",
                J, k, M, NW, hc[C_LEVELS], NCH, U, (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count()) * 1e3);
    return 0;
}

int rv_cascade_multi_run(rv_index *h, RvCascadeBufs &cb, int minl_in, RvCascadeMultiOut *out) {
    out->done = false; out->levels = 0; out->cands = out->witnesses = out->children = out->undecided = out->rebuilt_ranks = 0; out->steps = 0; out->maxdepth = 0;
    out->why = nullptr; out->big = out->big_ranks = 0; out->an_l.clear(); out->an_pos.clear(); out->meta.clear(); out->node_first.clear(); out->nodes.clear(); out->d_sa = out->d_lcp = out->d_bwt = nullptr;
    Workspace &ws = h->ws;
    hipStream_t q = ws.stream;
    const int64_t n = h->n;
    const int k = h->nsamples;
    const u32 minl = (u32)std::max(minl_in, 1);
    // a job of a batch that leaves this function before the rendezvous tells the group so (the others would wait for it for ever)
    struct BatchTicket { rv_index *h; bool settled = false; ~BatchTicket() { if (h->batch && !settled && !h->batch_settled) { h->batch_settled = true; rv_batch_group_leave(h->batch); } } } ticket{h};
    const bool verbose = (ws.opt.cascade_log != 0);
#define GIVE_UP(msg) do { out->why = msg; if (verbose) fprintf(stderr, "cascade (%d samples): gave up: %s\n", k, msg); return 0; } while (0)
    if (k < 2 || k > RV_CASM_K) GIVE_UP("sample count outside the cascade's range");      // (two samples: the second attempt of rv_align.hip, see there)
    if ((int)h->nodes.size() != k || (int)h->nsep.size() != k - 1) GIVE_UP("not one sequence per sample");
    if (n >= ((int64_t)1 << 32) - 2) GIVE_UP("index above 2^32 positions");
#ifdef RV_SA64
    if ((u64)h->maxlcp >= (1ull << 24) || n >= ((int64_t)1 << 40)) GIVE_UP("bid word too narrow");
#endif
    std::vector<sa_t> rb((size_t)k), re((size_t)k);
    {
        std::vector<RvIntv> nd(h->nodes.begin(), h->nodes.end());
        std::sort(nd.begin(), nd.end(), [](const RvIntv &x, const RvIntv &y) { return x.begin < y.begin; });
        for (int s = 0; s < k; s++) {
            if (nd[(size_t)s].end <= nd[(size_t)s].begin) GIVE_UP("an empty sample");
            const int64_t lo = s ? h->nsep[(size_t)s - 1] : -1, hi = s < k - 1 ? h->nsep[(size_t)s] : n;
            if (nd[(size_t)s].begin <= lo || nd[(size_t)s].end > hi) GIVE_UP("not one sequence per sample");
            rb[(size_t)s] = (sa_t)nd[(size_t)s].begin; re[(size_t)s] = (sa_t)nd[(size_t)s].end;
        }
    }
    const sa_t *SA = h->dSA.as<sa_t>(); const lcp_t *LCP = h->dLCP.as<lcp_t>(); const uint8_t *BWT = h->dBWT.as<uint8_t>();
    const sa_t *nsep = h->dNsep.as<sa_t>();

    const u32 mcap = (u32)std::min<int64_t>(std::max<int64_t>(1 << 16, n / (int64_t)k / 4), 0x7fffffff);       // full matches: one per k ranks at most, thinned by left-maximality
    const u32 wcap = (u32)std::min<int64_t>(std::max<int64_t>(1 << 16, n / 16), 0x7fffffff);
    const int64_t ccap64 = 2 * n / ((int64_t)minl * k) + 16;
    if (ccap64 >= 0x7fffffff) GIVE_UP("too many sub-indices possible");
    const u32 ccap = (u32)ccap64, acap = ccap / 2 + 8;
    DBuf &bso = cb.d[0], &bcl0 = cb.d[1], &bcp0 = cb.d[2], &bk0 = cb.d[3], &bk1 = cb.d[4], &bv0 = cb.d[5], &bv1 = cb.d[6], &bcl = cb.d[7], &bcp = cb.d[8], &bcc = cb.d[9],
         &bwp = cb.d[10], &bwv = cb.d[11], &bwc = cb.d[12], &btb = cb.d[13], &bctr = cb.d[14], &bund = cb.d[15], &banl = cb.d[16], &banp = cb.d[17], &broot = cb.d[18],
         &bsa = cb.d[19], &blcp = cb.d[20], &bbwt = cb.d[21], &brt = cb.d[22], &bexp = cb.d[23], &brows = cb.d[34];
    RV_TRY(bso.reserve((size_t)n + 64)); RV_TRY(bcl0.reserve((size_t)mcap * 4)); RV_TRY(bcp0.reserve((size_t)mcap * k * sizeof(sa_t)));
    RV_TRY(bwp.reserve((size_t)wcap * sizeof(sa_t))); RV_TRY(bwv.reserve((size_t)wcap * 4)); RV_TRY(bwc.reserve((size_t)wcap * 4));
    constexpr int CNT_STRIDE = 64;      // words between the counters of two regions while the scan counts (256 B: another L2 channel)
    RV_TRY(bctr.reserve(1024 + (size_t)CM_REGIONS * CNT_STRIDE * 4 + 64)); RV_TRY(broot.reserve((size_t)2 * k * sizeof(sa_t)));
    // per sub-index tables, one allocation: b, e, q (k sa_t each), best (u64), rmax, depth, state, lead, trail, ql (u32)
    const size_t per = (size_t)3 * k * sizeof(sa_t) + 8 + 6 * 4;
    RV_TRY(btb.reserve(per * ccap + 256)); RV_TRY(bund.reserve((size_t)ccap * 4)); RV_TRY(banl.reserve((size_t)acap * 4)); RV_TRY(banp.reserve((size_t)acap * k * sizeof(sa_t)));
    CmTabs t;
    {
        uint8_t *p = btb.as<uint8_t>();
        t.best = (u64 *)p; p += (size_t)ccap * 8;
        t.b = (sa_t *)p; p += (size_t)ccap * k * sizeof(sa_t); t.e = (sa_t *)p; p += (size_t)ccap * k * sizeof(sa_t); t.q = (sa_t *)p; p += (size_t)ccap * k * sizeof(sa_t);
        t.rmax = (u32 *)p; p += (size_t)ccap * 4; t.depth = (int32_t *)p; p += (size_t)ccap * 4; t.state = (u32 *)p; p += (size_t)ccap * 4;
        t.lead = (u32 *)p; p += (size_t)ccap * 4; t.trail = (u32 *)p; p += (size_t)ccap * 4; t.ql = (u32 *)p;
    }
    u32 *counters = bctr.as<u32>();
    u32 *region_cnt = counters + 16, *region_off = region_cnt + CM_REGIONS;      // (the match list's regions: counts, then where each starts in the dense list)
    u32 *region_spread = counters + 256;
    RV_HIP(hipMemsetAsync(counters, 0, 1024 + (size_t)CM_REGIONS * CNT_STRIDE * 4, q));
    const u32 rcap = mcap / CM_REGIONS;
    {
        std::vector<sa_t> rr(rb); rr.insert(rr.end(), re.begin(), re.end());
        RV_HIP(hipMemcpyAsync(broot.p, rr.data(), rr.size() * sizeof(sa_t), hipMemcpyHostToDevice, q));
        RV_HIP(hipStreamSynchronize(q));      // (the vector leaves scope)
    }
    const sa_t *d_rb = broot.as<sa_t>(), *d_re = d_rb + k;
    hipLaunchKernelGGL(k_casm_so, dim3((unsigned)ceil_div(n, (int64_t)TB * SO_PER)), dim3(TB), 0, q, SA, n, nsep, k, bso.as<uint8_t>());
    RV_LAUNCH_CHECK();
    {
        hipEvent_t ev_a = nullptr, ev_b = nullptr;      // 8 B per rank; the events ride on the kernel's own dispatch
        int pid = -1;
        if (!ws.opt.scan_v1) (void)h->prof.attach(RV_K_SCAN_MULTI, (double)n * 8.0, &ev_a, &ev_b);
        else pid = h->prof.begin(q, RV_K_SCAN_MULTI, (double)n * 8.0);
#define RV_SCAN_(KT) hipLaunchKernelGGL(k_casm_scan<KT>, dim3((unsigned)ceil_div(n, MS_TILE)), dim3(TB), 0, q, SA, LCP, BWT, (const uint8_t *)bso.as<uint8_t>(), n, k, minl, bcl0.as<u32>(), bcp0.as<sa_t>(), rcap, region_cnt)
        if (!ws.opt.scan_v1) {
            RV_TRY(rv_full_list_launch(ws, SA, LCP, BWT, n, nsep, k, minl, bcl0.as<u32>(), bcp0.as<sa_t>(), rcap, region_spread, CM_REGIONS, CNT_STRIDE, ev_a, ev_b));
            hipLaunchKernelGGL(k_casm_collect, dim3(1), dim3(64), 0, q, (const u32 *)region_spread, CNT_STRIDE, region_cnt);
        }
        else switch (k) {      // (RV_SCAN_V1: the kernel that stages a workgroup's ranks in LDS, for comparison)
            case 3: RV_SCAN_(3); break; case 4: RV_SCAN_(4); break; case 5: RV_SCAN_(5); break; case 6: RV_SCAN_(6); break; case 8: RV_SCAN_(8); break;
            case 10: RV_SCAN_(10); break; case 12: RV_SCAN_(12); break; case 16: RV_SCAN_(16); break; default: RV_SCAN_(0); break;
        }
#undef RV_SCAN_
        RV_LAUNCH_CHECK();
        h->prof.end(q, pid);
    }
    hipLaunchKernelGGL(k_casm_witness, dim3((unsigned)ceil_div(n, MS_TILE)), dim3(TB), 0, q, SA, LCP, (const uint8_t *)bso.as<uint8_t>(), n, minl, bwp.as<sa_t>(), bwv.as<u32>(), wcap, counters);
    RV_LAUNCH_CHECK();
    u32 hc[16];
    RV_TRY(rv_read_back(ws, hc, counters, sizeof hc));
    u32 hreg[2 * CM_REGIONS];
    RV_TRY(rv_read_back(ws, hreg, region_cnt, CM_REGIONS * 4));
    u32 M = 0, rmaxc = 0;
    for (int r = 0; r < CM_REGIONS; r++) { hreg[CM_REGIONS + r] = M; M += hreg[r]; rmaxc = std::max(rmaxc, hreg[r]); }
    const u32 NW = hc[C_NWIT];
    out->cands = M; out->witnesses = NW;
    if (rmaxc > rcap) GIVE_UP("more full matches than the list holds");
    if (NW > wcap) GIVE_UP("too many repeat witnesses (a repetitive input)");
    if (M == 0) GIVE_UP("no full match at the top level");
    struct ProfSpan { Workspace &w; int id; ~ProfSpan() { w.prof_end(id); } } span{ws, ws.prof_begin(RV_K_CASCADE, 5.0 * (double)n)};
    RV_TRY(bk0.reserve((size_t)M * 8)); RV_TRY(bk1.reserve((size_t)M * 8)); RV_TRY(bv0.reserve((size_t)M * 4)); RV_TRY(bv1.reserve((size_t)M * 4));
    RV_TRY(bcl.reserve((size_t)M * 4)); RV_TRY(bcp.reserve((size_t)M * k * sizeof(sa_t))); RV_TRY(bcc.reserve((size_t)M * 4));
    {
        const unsigned mb = (unsigned)ceil_div((int64_t)M, TB);
        RV_HIP(hipMemcpyAsync(region_off, hreg + CM_REGIONS, CM_REGIONS * 4, hipMemcpyHostToDevice, q));
        RV_HIP(hipStreamSynchronize(q));      // (hreg lives on this stack frame)
        hipLaunchKernelGGL(k_casm_keys, dim3((unsigned)std::max<int64_t>(1, ceil_div((int64_t)rmaxc, TB)), CM_REGIONS), dim3(TB), 0, q, (const sa_t *)bcp0.as<sa_t>(), k, rcap,
                           (const u32 *)region_cnt, (const u32 *)region_off, bk0.as<u64>(), bv0.as<u32>());
        RV_LAUNCH_CHECK();
        int in1 = 0;
        RV_TRY(rv_radix_sort_pairs<u32>(ws, bk0.as<u64>(), bv0.as<u32>(), bk1.as<u64>(), bv1.as<u32>(), (int64_t)M, 0, bitlen64m((u64)n), &in1));
        hipLaunchKernelGGL(k_casm_gather, dim3(mb), dim3(TB), 0, q, (const u32 *)bcl0.as<u32>(), (const sa_t *)bcp0.as<sa_t>(), (const u32 *)(in1 ? bv1.as<u32>() : bv0.as<u32>()), k, M,
                           bcl.as<u32>(), bcp.as<sa_t>(), bcc.as<u32>());
        RV_LAUNCH_CHECK();
    }
    // undecided sub-indices of up to big_min ranks are rebuilt by a workgroup in LDS, larger ones through global memory (RV_CASM_BIG_MIN: the test hook
    // that sends smaller ones there, too; RV_CASM_NO_BIG=1: the cascade gives up on them as it did up to round 4)
    const bool no_big = ws.opt.casm_no_big != 0;
    const int64_t big_min = (!no_big && ws.opt.casm_big_min >= 0 && ws.opt.casm_big_min < BN) ? ws.opt.casm_big_min : BN;
    auto too_big = [&](const u32 *c) {
        unsigned long long tot; memcpy(&tot, c + C_BIGTOT, 8);
        return no_big ? c[C_MAXN] > (u32)BN : ((int64_t)c[C_MAXN] > ws.opt.casm_big_root || (int64_t)tot > ws.opt.casm_big_total);
    };
    // ---- a job of a batch (rv_batch_run): the level loop is run once for all jobs by whoever arrives last (batch_joint_phase above)
    bool joint = false;
    RvBatchJob bjob;
    if (h->batch && !h->batch_settled) {
        h->batch_settled = true; ticket.settled = true;
        bjob.k = k; bjob.minl = minl; bjob.M = M; bjob.NW = std::min(NW, wcap); bjob.ccap = ccap; bjob.acap = acap;
        bjob.c_len = bcl.as<u32>(); bjob.c_pos = bcp.as<sa_t>(); bjob.w_pos = bwp.as<sa_t>(); bjob.w_val = bwv.as<u32>();
        bjob.nsep = nsep; bjob.dT = h->dT.as<uint8_t>();
        bjob.rb = rb; bjob.re = re;
        for (int s2 = 0; s2 + 1 < k; s2++) bjob.sep.push_back((sa_t)h->nsep[(size_t)s2]);
        bjob.big_min = big_min; bjob.big_root = ws.opt.casm_big_root; bjob.big_total = ws.opt.casm_big_total; bjob.no_big = no_big;
        bjob.stream = q;
        if (!cb.ev_in) RV_HIP(hipEventCreateWithFlags(&cb.ev_in, hipEventDisableTiming));
        bjob.ready = cb.ev_in;
        RV_HIP(hipEventRecord(cb.ev_in, q));
        batch_join(h->batch, &bjob, ws);
        joint = bjob.ok;
        if (joint) {
            RV_HIP(hipStreamWaitEvent(q, h->batch->ev_done, 0));
            t = h->batch->t;      // (the sub-indices of every job of the batch: this job's are named by the ids it was handed)
            memset(hc, 0, sizeof hc);
            hc[C_LEVELS] = bjob.levels; hc[C_NCHILD] = bjob.nchild; hc[C_NUND] = (u32)bjob.und_id.size(); hc[C_NANCH] = bjob.na;
            hc[C_STEPS] = bjob.steps; hc[C_MAXDEPTH] = bjob.maxdepth; hc[C_MAXN] = bjob.maxn;
        }
    }
    if (!joint) {
    // (RV_CASCADE_PRIO=1: the level loop -- ~120 launches of a few microseconds each, a host round trip every eighth level -- runs on a stream of the highest
    //  priority, fenced by events against the handle's own; restored before the function returns on any path)
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
            int lo = 0, hi = 0;
            RV_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));      // (hi: the numerically smallest = the greatest priority)
            RV_HIP(hipStreamCreateWithPriority(&cb.prio_stream, hipStreamNonBlocking, hi));
            RV_HIP(hipEventCreateWithFlags(&cb.ev_in, hipEventDisableTiming));
            RV_HIP(hipEventCreateWithFlags(&cb.ev_out, hipEventDisableTiming));
        }
        RV_HIP(hipEventRecord(cb.ev_in, q));
        RV_HIP(hipStreamWaitEvent(cb.prio_stream, cb.ev_in, 0));
        ws.stream = cb.prio_stream; prio.on = true;
        q = cb.prio_stream;
    }
    hipLaunchKernelGGL(k_casm_init, dim3((unsigned)std::max<int64_t>(1, ceil_div((int64_t)NW, TB))), dim3(TB), 0, q, t, k, d_rb, d_re, counters, bwc.as<u32>(), NW);
    RV_LAUNCH_CHECK();
    const unsigned agrid = (unsigned)(ceil_div((int64_t)M, TB) + ceil_div((int64_t)NW, TB));
    const int batch = std::max(1, (int)ws.opt.cascade_batch);
    int queued = 0;
    for (;;) {
        for (int b = 0; b < batch; b++, queued++) {
            hipLaunchKernelGGL(k_casm_assign, dim3(agrid), dim3(TB), 0, q, (const sa_t *)bcp.as<sa_t>(), (const u32 *)bcl.as<u32>(), bcc.as<u32>(), M, (const sa_t *)bwp.as<sa_t>(),
                               (const u32 *)bwv.as<u32>(), bwc.as<u32>(), NW, t, k, nsep, (int64_t)minl, queued == 0 ? 1 : 0);
            RV_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_casm_winner, dim3((unsigned)ceil_div((int64_t)M, TB)), dim3(TB), 0, q, (const sa_t *)bcp.as<sa_t>(), (const u32 *)bcl.as<u32>(), (const u32 *)bcc.as<u32>(), M, t, k,
                               (int64_t)minl);
            RV_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_casm_decide, dim3(192), dim3(TB), 0, q, t, k, minl, counters, ccap, bund.as<u32>(), (u32)big_min, banl.as<u32>(), banp.as<sa_t>(), acap);
            RV_LAUNCH_CHECK();
        }
        RV_TRY(rv_read_back(ws, hc, counters, sizeof hc));
        if (hc[C_ERR] & ~5u) { rv_set_error("cascade (multi): device error %u", hc[C_ERR]); return -1; }
        if (hc[C_ERR]) GIVE_UP("the cascade's tables are full");      // (bits 1 and 4: sub-index table / anchor area: the level pipeline completes the run)
        if (too_big(hc)) break;
        if (hc[C_HI] == hc[C_LO]) break;
        if (queued > 1000000) { rv_set_error("cascade (multi): no progress"); return -1; }
    }
    prio.leave(); q = ws.stream;
    }      // (!joint)
    out->levels = (int)hc[C_LEVELS]; out->children = hc[C_NCHILD];
    const u32 U = hc[C_NUND], NA = hc[C_NANCH];
    out->undecided = U;
    if (joint ? bjob.too_big : too_big(hc)) {
        out->why = "an undecided sub-index above the size that is rebuilt from the text";
        if (verbose) fprintf(stderr, "cascade (%d samples): gave up: %s (%u ranks; %u levels, %u sub-indices, %u undecided)\n", k, out->why, hc[C_MAXN], hc[C_LEVELS], hc[C_NCHILD], U);
        return 0;
    }
    if (U == 1 && hc[C_NCHILD] == 1 && n > (int64_t)BN)      // (nothing was decided: rebuilding the root's arrays from its text would only copy the index)
        GIVE_UP("the index' longest match of all samples is no longer than its repeats");
    out->steps = hc[C_STEPS]; out->maxdepth = (int)hc[C_MAXDEPTH];
    // ---- anchors to the host, their text lower-cased (reveal.c:1230-1234); the rows of the undecided sub-indices (ids, depths, begins, ends) gathered on
    // the device (the tables are small next to the index, but only these rows are needed).  Everything through ONE pinned buffer and one wait: copies to
    // and from pageable vectors are staged by the runtime and waited for one by one (five of them: 0.4 ms of a 10 ms step at 10 x 5 Mbp)
    const size_t b_anl = 0, b_anp = b_anl + (((size_t)NA * 4 + 15) & ~(size_t)15), b_rows = b_anp + (((size_t)NA * k * sizeof(sa_t) + 15) & ~(size_t)15);
    const size_t rowbytes = (size_t)U * 8 + (size_t)U * k * sizeof(sa_t) * 2;
    RV_TRY(cb.hstage.reserve(b_rows + rowbytes + 64));
    uint8_t *hs = cb.hstage.as<uint8_t>();
    // (a job of a batch: its anchors are a stretch of the group's arrays, its text has been lower-cased and its rows are on the host already)
    const u32 *d_anl = joint ? bjob.d_anl : (const u32 *)banl.as<u32>();
    const sa_t *d_anp = joint ? bjob.d_anp : (const sa_t *)banp.as<sa_t>();
    if (joint) {
        if (NA) {
            RV_HIP(hipMemcpyAsync(hs + b_anl, d_anl, (size_t)NA * 4, hipMemcpyDeviceToHost, q));
            RV_HIP(hipMemcpyAsync(hs + b_anp, d_anp, (size_t)NA * k * sizeof(sa_t), hipMemcpyDeviceToHost, q));
        }
        if (U) {
            uint8_t *hr = hs + b_rows;
            memcpy(hr, bjob.und_id.data(), (size_t)U * 4); memcpy(hr + (size_t)U * 4, bjob.und_dep.data(), (size_t)U * 4);
            memcpy(hr + (size_t)U * 8, bjob.und_b.data(), (size_t)U * k * sizeof(sa_t)); memcpy(hr + (size_t)U * 8 + (size_t)U * k * sizeof(sa_t), bjob.und_e.data(), (size_t)U * k * sizeof(sa_t));
        }
    } else {
    if (NA) {
        RV_HIP(hipMemcpyAsync(hs + b_anl, banl.p, (size_t)NA * 4, hipMemcpyDeviceToHost, q));
        RV_HIP(hipMemcpyAsync(hs + b_anp, banp.p, (size_t)NA * k * sizeof(sa_t), hipMemcpyDeviceToHost, q));
        RV_TRY(bexp.reserve((size_t)NA * k * 4));
        hipLaunchKernelGGL(k_casm_expand_l, dim3((unsigned)ceil_div((int64_t)NA * k, TB)), dim3(TB), 0, q, (const u32 *)banl.as<u32>(), NA, k, bexp.as<u32>());
        RV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_casm_lower, dim3((unsigned)ceil_div((int64_t)NA * k * 64, TB)), dim3(TB), 0, q, h->dT.as<uint8_t>(), (const u32 *)bexp.as<u32>(), (const sa_t *)banp.as<sa_t>(), NA * (u32)k);
        RV_LAUNCH_CHECK();
    }
    if (U) {
        RV_TRY(brows.reserve(rowbytes + 64));
        u32 *d_id = brows.as<u32>(); int32_t *d_dep = (int32_t *)(d_id + U);
        sa_t *d_b = (sa_t *)(d_dep + U), *d_e = d_b + (size_t)U * k;
        hipLaunchKernelGGL(k_casm_rows, dim3((unsigned)ceil_div((int64_t)U * k, TB)), dim3(TB), 0, q, (const u32 *)bund.as<u32>(), U, t, k, d_id, d_dep, d_b, d_e);
        RV_LAUNCH_CHECK();
        RV_HIP(hipMemcpyAsync(hs + b_rows, brows.p, rowbytes, hipMemcpyDeviceToHost, q));
    }
    }      // (!joint)
    if (NA || U) RV_HIP(hipStreamSynchronize(q));
    // ---- undecided sub-indices: arrays from their text, bookkeeping for the level pipeline
    if (U) {
        const uint8_t *hrows = hs + b_rows;      // (in the order of the ids on the host)
        const u32 *r_id = (const u32 *)hrows; const int32_t *r_dep = (const int32_t *)(r_id + U);
        const sa_t *r_b = (const sa_t *)(r_dep + U), *r_e = r_b + (size_t)U * k;
        std::vector<u32> order(U);
        for (u32 x = 0; x < U; x++) order[x] = x;
        std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return r_id[a] < r_id[b]; });
        std::vector<u32> ids(U);
        std::vector<sa_t> hb((size_t)U * k), he((size_t)U * k);
        std::vector<int32_t> hd(U);
        for (u32 x = 0; x < U; x++) {
            const u32 o = order[x];
            ids[x] = r_id[o]; hd[x] = r_dep[o];
            for (int s2 = 0; s2 < k; s2++) { hb[(size_t)x * k + s2] = r_b[(size_t)o * k + s2]; he[(size_t)x * k + s2] = r_e[(size_t)o * k + s2]; }
        }
        std::vector<CmRoot> roots;      // (up to big_min ranks: a workgroup each, in LDS)
        std::vector<CmBig> bigs;        // (above: through global memory)
        int64_t m = 0, mL = 0;
        out->node_first.assign(1, 0);
        for (u32 x = 0; x < U; x++) {
            int64_t sz = 0; int ns = 0;
            for (int s = 0; s < k; s++) {
                const int64_t b = hb[(size_t)x * k + s], e = he[(size_t)x * k + s];
                if (e > b) { sz += e - b; ns++; out->nodes.push_back(b); out->nodes.push_back(e); }
            }
            out->node_first.push_back((int64_t)out->nodes.size() / 2);
            if (sz > big_min) { bigs.push_back({m, mL, (int32_t)sz, (int32_t)ids[x]}); mL += sz; }
            else roots.push_back({m, (int32_t)sz, (int32_t)ids[x]});
            const int64_t m6[6] = {m, sz, hd[x], ns, 0, -1};
            out->meta.insert(out->meta.end(), m6, m6 + 6);
            m += sz;
        }
        out->rebuilt_ranks = m;
        RV_TRY(bsa.reserve((size_t)(m + 64) * sizeof(sa_t))); RV_TRY(blcp.reserve((size_t)(m + 64) * sizeof(lcp_t))); RV_TRY(bbwt.reserve((size_t)m + 64));
        std::vector<CmSlice> slices, bslices;
        for (size_t x = 0; x < roots.size(); x++)
            for (int f = 0; f < roots[x].n; f += TB) slices.push_back({(int32_t)x, (int32_t)f});
        for (size_t x = 0; x < bigs.size(); x++)
            for (int f = 0; f < bigs[x].n; f += TB) bslices.push_back({(int32_t)x, (int32_t)f});
        const size_t rbytes = roots.size() * sizeof(CmRoot), sbytes = slices.size() * sizeof(CmSlice), gbytes = bigs.size() * sizeof(CmBig), tbytes = bslices.size() * sizeof(CmSlice);
        RV_TRY(brt.reserve(rbytes + sbytes + gbytes + tbytes + 64));
        RV_TRY(bexp.reserve((size_t)(m + 64) * 4));      // (the lower-casing above is done with it: its launch has been waited for)
        CmRoot *d_roots = brt.as<CmRoot>();
        CmSlice *d_slices = (CmSlice *)(d_roots + roots.size());
        CmBig *d_bigs = (CmBig *)(d_slices + slices.size());      // (16-byte records in front, 8-byte ones behind them, 24-byte ones at a multiple of 8)
        CmSlice *d_bslices = (CmSlice *)(d_bigs + bigs.size());
        {      // (tables in one copy from pinned memory, queued in front of the kernels that read them; the staging area's rows have been read)
            RV_TRY(cb.hstage2.reserve(rbytes + sbytes + gbytes + tbytes + 64));
            uint8_t *hp = cb.hstage2.as<uint8_t>();
            if (rbytes) memcpy(hp, roots.data(), rbytes);
            if (sbytes) memcpy(hp + rbytes, slices.data(), sbytes);
            if (gbytes) memcpy(hp + rbytes + sbytes, bigs.data(), gbytes);
            if (tbytes) memcpy(hp + rbytes + sbytes + gbytes, bslices.data(), tbytes);
            RV_HIP(hipMemcpyAsync(d_roots, cb.hstage2.p, rbytes + sbytes + gbytes + tbytes, hipMemcpyHostToDevice, q));
        }
        if (!roots.empty()) {
            if (h->ws.opt.casm_rank_count)      // (test hook: the ranks counted out by comparison, k_casm_rank)
                hipLaunchKernelGGL(k_casm_rank, dim3((unsigned)slices.size()), dim3(TB), 0, q, (const CmRoot *)d_roots, (const CmSlice *)d_slices, t, k, (const uint8_t *)h->dT0.as<uint8_t>(), bexp.as<u32>());
            else
                hipLaunchKernelGGL(k_casm_rank_sort, dim3((unsigned)roots.size()), dim3(RK_TB), 0, q, (const CmRoot *)d_roots, t, k, (const uint8_t *)h->dT0.as<uint8_t>(), bexp.as<u32>());
            RV_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_casm_emit, dim3((unsigned)slices.size()), dim3(TB), 0, q, (const CmRoot *)d_roots, (const CmSlice *)d_slices, t, k, (const uint8_t *)h->dT0.as<uint8_t>(),
                               (const u32 *)bexp.as<u32>(), bsa.as<sa_t>(), blcp.as<lcp_t>(), bbwt.as<uint8_t>(), h->nsep[0], d_rb);
            RV_LAUNCH_CHECK();
        }
        if (!bigs.empty()) {
            int rootbits = 1;
            while (((size_t)1 << rootbits) < bigs.size()) rootbits++;
            const int kb = std::min(6, (64 - rootbits) / 8);
            RV_TRY(bk0.reserve((size_t)(mL + 64) * 8)); RV_TRY(bk1.reserve((size_t)(mL + 64) * 8)); RV_TRY(bv0.reserve((size_t)(mL + 64) * 8)); RV_TRY(bv1.reserve((size_t)(mL + 64) * 8));
            hipLaunchKernelGGL(k_casmb_keys, dim3((unsigned)bslices.size()), dim3(TB), 0, q, (const CmBig *)d_bigs, (const CmSlice *)d_bslices, t, k, (const uint8_t *)h->dT0.as<uint8_t>(), kb,
                               bk0.as<u64>(), bv0.as<u64>());
            RV_LAUNCH_CHECK();
            int in1 = 0;
            RV_TRY(rv_radix_sort_pairs<u64>(ws, bk0.as<u64>(), bv0.as<u64>(), bk1.as<u64>(), bv1.as<u64>(), mL, 0, 8 * kb + rootbits, &in1));
            hipLaunchKernelGGL(k_casmb_place, dim3((unsigned)ceil_div(mL, TB)), dim3(TB), 0, q, (const CmBig *)d_bigs, (const u64 *)(in1 ? bk1.as<u64>() : bk0.as<u64>()),
                               (const u64 *)(in1 ? bv1.as<u64>() : bv0.as<u64>()), mL, kb, (const uint8_t *)h->dT0.as<uint8_t>(), bsa.as<sa_t>(), bexp.as<u32>(), counters + C_BIGERR);
            RV_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_casmb_emit, dim3((unsigned)bslices.size()), dim3(TB), 0, q, (const CmBig *)d_bigs, (const CmSlice *)d_bslices, t, k, (const uint8_t *)h->dT0.as<uint8_t>(),
                               (const u32 *)bexp.as<u32>(), (const sa_t *)bsa.as<sa_t>(), blcp.as<lcp_t>(), bbwt.as<uint8_t>(), h->nsep[0], d_rb, (const u32 *)(counters + C_BIGERR));
            RV_LAUNCH_CHECK();
            u32 berr = 0;
            RV_TRY(rv_read_back(ws, &berr, counters + C_BIGERR, 4));
            if (berr) {      // (a tie group above BIG_GROUP: nothing of this run stays -- the anchors' text goes back to what it was)
                if (NA) {
                    RV_TRY(bcl0.reserve((size_t)NA * k * 4));      // (the candidates' lengths: long gathered into the sorted list)
                    hipLaunchKernelGGL(k_casm_expand_l, dim3((unsigned)ceil_div((int64_t)NA * k, TB)), dim3(TB), 0, q, d_anl, NA, k, bcl0.as<u32>());
                    RV_LAUNCH_CHECK();
                    hipLaunchKernelGGL(k_casm_unlower, dim3((unsigned)ceil_div((int64_t)NA * k * 64, TB)), dim3(TB), 0, q, h->dT.as<uint8_t>(), (const uint8_t *)h->dT0.as<uint8_t>(),
                                       (const u32 *)bcl0.as<u32>(), d_anp, NA * (u32)k);
                    RV_LAUNCH_CHECK();
                    RV_HIP(hipStreamSynchronize(q));
                }
                out->meta.clear(); out->node_first.clear(); out->nodes.clear(); out->rebuilt_ranks = 0;
                GIVE_UP("an undecided sub-index with more suffixes sharing their first bytes than are compared one by one");
            }
            out->big = (int64_t)bigs.size(); out->big_ranks = mL;
        }
        out->d_sa = bsa.p; out->d_lcp = blcp.p; out->d_bwt = bbwt.p;
    }
    if (NA) {      // (behind the launches above: the host copies the anchors out of the staging area while the GPU rebuilds the undecided sub-indices)
        const u32 *pl = (const u32 *)(hs + b_anl); const sa_t *pp = (const sa_t *)(hs + b_anp);
        out->an_l.assign(pl, pl + NA);
        out->an_pos.assign(pp, pp + (size_t)NA * k);
    }
    if (verbose) fprintf(stderr, "cascade (%d samples): %u full matches, %u witnesses, %d levels, %u sub-indices, %u anchors, %u undecided (%lld ranks rebuilt; %lld of them above %lld ranks: %lld)\n", k, M, NW,
                         out->levels, hc[C_NCHILD], NA, U, (long long)out->rebuilt_ranks, (long long)out->big, (long long)big_min, (long long)out->big_ranks);
    out->done = true;
    return 0;
#undef GIVE_UP
}

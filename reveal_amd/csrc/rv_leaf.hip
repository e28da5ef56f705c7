// rv_leaf.hip -- the tail of the recursion inside one workgroup.
//
// The recursion of aligner() (reveallib/reveal.c:731-1338) produces ~10^5
// sub-indices per 10 Mbp, most of a few hundred ranks.  Once a sub-index of a
// two-sample alignment has at most RV_LEAF_N ranks, its whole sub-tree is
// finished here by one workgroup with the arrays in LDS -- one wavefront per sub-index, the waves of the workgroup taking
// sub-indices of the root's tree from a shared stack, no workgroup barrier in between:
//   scan   getmums_rem predicate                      reveal.c:119-180
//   pick   built-in picker (longest full match, ties -> smallest coordinate)
//   split  D-label + stable partition + running-min LCP + lower-casing   reveal.c:1005-1234, 582-664
//   bubble bubble_sort on the leading child, cuts in ascending order     reveal.c:666-727
// Same arithmetic as the level kernels (rv_scan.hip, rv_split.hip), so every
// sub-index has the same SA/LCP as in the reference; only used with the built-in
// callbacks (rv_align_builtin), never when Python callbacks drive the recursion.
#include "rv_common.h"
#include "rv_leaf.h"
#include <type_traits>

namespace {

constexpr int NT = 256;
constexpr int LN = RV_LEAF_N;
constexpr u32 INF = 0xFFFFFFFFu;
constexpr int NW = NT / 64;
constexpr int MAXSTACK = 128;
constexpr int ACAP = 256;            // anchors staged per workgroup (a root of 2048 ranks holds ~5 at minl 20)

typedef std::make_unsigned<sa_t>::type usa_t;      // positions compared in their own width (32 bits in reveallib)
struct Frame { int start, len, depth, buf; int64_t a0, a1, b0, b1; };   // sample-0 interval [a0,a1), sample-1 interval [b0,b1); empty if a0>=a1

__device__ inline bool is_lower_c(uint8_t c) { return c >= 'a' && c <= 'z'; }
__device__ inline bool left_maximal(uint8_t ca, uint8_t cb) { return (ca != cb) || ca == 'N' || ca == '$' || is_lower_c(ca); }

__device__ inline u64 hash_step(u64 acc, u64 i, int64_t v) {      // oracle/reveal_oracle.c ro_hash_step
    u64 x = (u64)v + (i + 1) * 0x9E3779B97F4A7C15ULL;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return acc + x;
}

// ---- wave-level collectives (DPP, rv_common.h) ------------------------------------------
// Every sub-index is processed by ONE wavefront: no workgroup barrier inside the recursion, the four waves of a workgroup
// work on different sub-indices of the same root (disjoint rank ranges of the LDS arrays).
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ inline u64 wave_max_u64(u64 v) { return rv_wave_max_u64(v); }
__device__ inline u64 wave_sum_u64(u64 v) {       // trace mode only
    for (int d = 32; d >= 1; d >>= 1) v += ((u64)__shfl_xor((u32)(v >> 32), d, 64) << 32) | __shfl_xor((u32)v, d, 64);
    return v;
}
__device__ inline u32 lanes_below(u64 mask) { return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u)); }
// the value of the lane below (lane 0: `first`)
__device__ inline u32 from_lane_below(u32 x, u32 first) { return (u32)__builtin_amdgcn_update_dpp((int)first, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }

// Running minimum of the LCP values since the last rank of class 0 / class 1 (reveal.c:582-664 keeps one running minimum per
// child): has = bit k set once a rank of class k was seen, v[k] = minimum since then.  Inclusive scan over the wave.
struct MinSt2 { u32 has, v0, v1; };
__device__ inline MinSt2 ms2_combine(MinSt2 a, MinSt2 b) {      // a, then b
    MinSt2 r; r.has = a.has | b.has;
    r.v0 = (b.has & 1u) ? b.v0 : (a.v0 < b.v0 ? a.v0 : b.v0);
    r.v1 = (b.has & 2u) ? b.v1 : (a.v1 < b.v1 ? a.v1 : b.v1);
    return r;
}
__device__ inline MinSt2 wave_incl_ms2(MinSt2 m) {
    const int lane = threadIdx.x & 63;
#define LF_STEP_(CTRL, RM, TAKE) {                                                                                    \
        MinSt2 t; t.has = rv_dpp_u32<CTRL, RM>(m.has); t.v0 = rv_dpp_u32<CTRL, RM>(m.v0); t.v1 = rv_dpp_u32<CTRL, RM>(m.v1);    \
        const MinSt2 c = ms2_combine(t, m);                                                                           \
        if (TAKE) m = c;                                                                                              \
    }
    RV_WAVE_SCAN_STEPS(LF_STEP_)
#undef LF_STEP_
    return m;
}

__global__ __launch_bounds__(NT) void k_leaf(RvLeafArgs A) {
    // two copies of the arrays: a split reads one and writes the children into the other, a sub-index remembers which one holds it
    __shared__ sa_t  sa2[2][LN];
    __shared__ u32   lc2[2][LN];
    __shared__ uint8_t bw2[2][LN];
    __shared__ uint16_t act[LN];
    __shared__ Frame stack[MAXSTACK];
    __shared__ Frame cur[NW];
#ifdef RV_LEAF_PAD
    __shared__ u32 pad_[RV_LEAF_PAD / 4];      // (tuning: fewer workgroups per CU)
    if (A.minl == -12345) pad_[threadIdx.x] = 0;
#endif
    __shared__ int s_top, s_pending, s_lock;
    // The anchors of the root are collected here and leave with ONE reservation per workgroup: a reservation per anchor was
    // 1.7 x 10^6 returning atomics on one address per run of 2 x 250 Mbp, ~50 ns each at the L2 -- the launches took exactly that long.
    __shared__ u32 an_l[ACAP]; __shared__ sa_t an_a[ACAP], an_b[ACAP];
    __shared__ u32 s_na, s_base;
    __shared__ unsigned long long s_stats[4];

    const RvLeafRoot root = A.roots[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < root.n; i += NT) {
        sa2[0][i] = A.SA[root.off + i]; lc2[0][i] = (u32)A.LCP[root.off + i]; bw2[0][i] = A.BWT[root.off + i] & RV_BWT_CHAR;      /* (the side bit is for the streaming scan; here SA is in LDS) */
    }
    if (tid == 0) {
        Frame f; f.start = 0; f.len = (int)root.n; f.depth = root.depth; f.buf = 0; f.a0 = root.a0; f.a1 = root.a1; f.b0 = root.b0; f.b1 = root.b1;
        cur[0] = f; s_top = 0; s_pending = 1; s_lock = 0; s_na = 0;
        s_stats[0] = s_stats[1] = s_stats[2] = s_stats[3] = 0;
    }
    __syncthreads();                                   // the only workgroup barrier
    const sa_t nsep0 = (sa_t)A.nsep0;
    const u32 minl = A.minl > 0 ? (u32)A.minl : 0u;
    const u32 acap = A.stage_cap < (u32)ACAP ? A.stage_cap : (u32)ACAP;
    u32 my_steps = 0, my_splits = 0, my_maxdepth = 0; u64 my_bp = 0;     // accumulated by lane 0 of every wave
    bool have = wv == 0;
#ifdef RV_LEAF_PROF
    long long pt = clock64(), p_idle = 0, p_scan = 0, p_split = 0, p_bub = 0;
#define LF_PROF(acc) { const long long now_ = clock64(); acc += now_ - pt; pt = now_; }
#else
#define LF_PROF(acc)
#endif

    for (;;) {
        LF_PROF(p_bub)
        if (!have) {
            // take a sub-index from the shared stack, or leave once every sub-index of the root is finished
            int got = 0;
            if (lane == 0) {
                for (;;) {
                    if (__hip_atomic_load(&s_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) { got = -1; break; }
                    if (__hip_atomic_load(&s_top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > 0) {
                        while (atomicCAS(&s_lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const int t = s_top;
                        if (t > 0) { cur[wv] = stack[t - 1]; s_top = t - 1; got = 1; }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        atomicExch(&s_lock, 0);
                        if (got) break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            got = __builtin_amdgcn_readfirstlane(got);
            if (got < 0) break;
        }
        WSYNC();
        const Frame f = cur[wv];
        have = false;
        LF_PROF(p_idle)
        const int b = f.buf;
        sa_t *cs = sa2[b], *ns = sa2[b ^ 1];
        u32 *cl = lc2[b], *nl_ = lc2[b ^ 1];
        uint8_t *cb = bw2[b], *nb = bw2[b ^ 1];
        const int S = f.start, E = f.start + f.len;
        const bool both = f.a0 < f.a1 && f.b0 < f.b1;           // nsamples == 2 (reveal.c:1034-1041)
        if (lane == 0) { my_steps++; if ((u32)f.depth > my_maxdepth) my_maxdepth = (u32)f.depth; }

        // ---- scan (reveal.c:131-159) + picker ------------------------------------------------
        u64 best = 0; sa_t bpart = 0;          // the lane's best candidate and its other member
        u64 hsa = 0, hlc = 0;
        for (int i = S + lane; i < E; i += 64) {
            if (A.trace) { hsa = hash_step(hsa, (u64)(i - S), (int64_t)cs[i]); hlc = hash_step(hlc, (u64)(i - S), (int64_t)cl[i]); }
            if (i == S) continue;
            const u32 l = cl[i];
            if (l < minl) continue;
            const sa_t s1 = cs[i], s0 = cs[i - 1];
            if ((s1 > nsep0) == (s0 > nsep0)) continue;
            const u32 la = (i + 1 < E) ? cl[i + 1] : 0u;
            if (!(cl[i - 1] < l && la < l)) continue;
            const bool ok = s1 < s0 ? left_maximal(cb[i], cb[i - 1]) : left_maximal(cb[i - 1], cb[i]);
            if (!ok) continue;
            const u64 a = (u64)(s1 < s0 ? s1 : s0);
            const u64 key = ((u64)l << 40) | (0xFFFFFFFFFFull - a);           // longest, then smallest position (< 2^40)
            if (key > best) { best = key; bpart = s1 < s0 ? s0 : s1; }
        }
        u64 hm = 0; u32 total_cand = 0;
        if (A.trace) {
            // scan-result hash needs each candidate's ordinal in rank order: second pass with a running count
            u32 run = 0;
            for (int base = S; base < E; base += 64) {
                const int i = base + lane;
                bool ok = false; u32 l = 0; sa_t s1 = 0, s0 = 0;
                if (i < E && i > S) {
                    l = cl[i]; s1 = cs[i]; s0 = cs[i - 1];
                    const u32 la = (i + 1 < E) ? cl[i + 1] : 0u;
                    ok = l >= minl && ((s1 > nsep0) != (s0 > nsep0)) && cl[i - 1] < l && la < l &&
                         (s1 < s0 ? left_maximal(cb[i], cb[i - 1]) : left_maximal(cb[i - 1], cb[i]));
                }
                const u64 mask = __ballot(ok);
                const u32 k = run + lanes_below(mask);
                if (ok) {
                    const int64_t a = (int64_t)(s1 < s0 ? s1 : s0), bb = (int64_t)(s1 < s0 ? s0 : s1);
                    const u64 o = (u64)k * 6;                  // the candidate as the oracle hashes it: l, n = 2, (0, a), (1, b)
                    hm = hash_step(hm, o, (int64_t)l); hm = hash_step(hm, o + 1, 2); hm = hash_step(hm, o + 2, 0);
                    hm = hash_step(hm, o + 3, a); hm = hash_step(hm, o + 4, 1); hm = hash_step(hm, o + 5, bb);
                }
                run += (u32)__popcll(mask);
            }
            total_cand = run;
            hsa = wave_sum_u64(hsa); hlc = wave_sum_u64(hlc); hm = wave_sum_u64(hm);
        }
        const u64 mine = best;
        best = wave_max_u64(best);
        const bool picked = both && best != 0;
        const u32 L = (u32)(best >> 40);
        const int64_t pa = (int64_t)(0xFFFFFFFFFFull - (best & 0xFFFFFFFFFFull));
        int64_t pb = 0;
        if (picked) {
            // the other member of the chosen match: held by the one lane whose candidate won (a position pairs with one rank only)
            const int owner = (int)__builtin_ctzll(__ballot(mine == best));
            const u64 bp = (u64)bpart;
            pb = (int64_t)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(bp >> 32), owner) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)bp, owner));
        }
        if (A.trace && lane == 0) {
            const u32 slot = atomicAdd(A.trace_count, 1u);
            if (slot < A.trace_cap) {
                rv_trace t;
                t.key = f.a0 < f.a1 ? f.a0 : f.b0; t.n = f.len; t.depth = f.depth; t.nsamples = (f.a0 < f.a1) + (f.b0 < f.b1);
                t.nnodes = t.nsamples; t.picked = picked ? 1 : 0; t.nmums = total_cand; t.l = picked ? L : 0; t.mn = picked ? 2 : 0;
                t.sp_min = picked ? pa : 0; t.h_sa = hsa; t.h_lcp = hlc; t.h_mums = hm;
                A.trace_out[slot] = t;
            }
        }
        LF_PROF(p_scan)
        if (!picked) {
            if (lane == 0) atomicSub(&s_pending, 1);
            continue;
        }
        if (lane == 0) {
            my_splits++; my_bp += L;
            const u32 k = atomicAdd(&s_na, 1u);
            if (k < acap) { an_l[k] = L; an_a[k] = (sa_t)pa; an_b[k] = (sa_t)pb; }
            else {                                   // (more anchors than the staging holds: minl of a few bases)
                const u32 slot = atomicAdd(A.anchor_count, 1u);
                if (slot < A.anchor_cap) { A.anchor_l[slot] = L; A.anchor_pos[2 * (size_t)slot] = pa; A.anchor_pos[2 * (size_t)slot + 1] = pb; }
            }
        }
        // (the matched text is lower-cased from the anchor list when the run ends: k_leaf_lower; nothing reads it before)
        // ---- linear graphalign: lead = left remainders, trail = right remainders ------------------
        const int64_t la0 = f.a0, la1 = pa, lb0 = f.b0, lb1 = pb;                     // leading intervals (may be empty)
        const int64_t ta0 = pa + L, ta1 = f.a1, tb0 = pb + L, tb1 = f.b1;             // trailing intervals
        // ---- label + split (reveal.c:1005-1117, 582-664) into the other copy: lead at S, trail right behind it ------
        // A sub-index holds exactly the suffixes of its intervals: the children's sizes follow from the interval lengths.  A lane
        // takes four consecutive ranks, so the wave-wide scans (counts, running minima) run once per 256 ranks.
        const usa_t LA0 = (usa_t)la0, LAn = la1 > la0 ? (usa_t)(la1 - la0) : 0, LB0 = (usa_t)lb0, LBn = lb1 > lb0 ? (usa_t)(lb1 - lb0) : 0;
        const usa_t TA0 = (usa_t)ta0, TAn = ta1 > ta0 ? (usa_t)(ta1 - ta0) : 0, TB0 = (usa_t)tb0, TBn = tb1 > tb0 ? (usa_t)(tb1 - tb0) : 0;
        const u32 nlead = (u32)(LAn + LBn);
        u32 cnt0 = 0, cnt1 = 0;                       // ranks already written to lead / trail
        MinSt2 car; car.has = 0; car.v0 = INF; car.v1 = INF;      // running-minimum carry
        for (int base = S; base < E; base += 4 * 64) {
            const int i0 = base + 4 * lane;
            u32 ev[4]; sa_t pos[4]; uint8_t bo[4]; u32 cls = 0;      // cls: two bits per rank (1 = lead, 2 = trail)
            MinSt2 agg; agg.has = 0; agg.v0 = INF; agg.v1 = INF;
            u32 n01 = 0;                                             // lead count | trail count << 16
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = i0 + r;
                u32 c = 0; ev[r] = INF; pos[r] = 0; bo[r] = 0;
                if (i < E) {
                    pos[r] = cs[i]; bo[r] = cb[i];
                    const usa_t p = (usa_t)pos[r];
                    if ((usa_t)(p - LA0) < LAn || (usa_t)(p - LB0) < LBn) c = 1;
                    else if ((usa_t)(p - TA0) < TAn || (usa_t)(p - TB0) < TBn) c = 2;
                    ev[r] = (i > S) ? cl[i] : INF;      // every rank of a leaf sub-index is labelled (lead, trail or matched): no skipped updates
                    if (c == 2 && (p == TA0 || p == TB0) && bo[r] >= 'A' && bo[r] <= 'Z') bo[r] += 32;   // its left neighbour was just matched
                }
                cls |= c << (2 * r);
                n01 += (c == 1 ? 1u : 0u) + (c == 2 ? 0x10000u : 0u);
                agg.has |= c;
                agg.v0 = c == 1 ? INF : (agg.v0 < ev[r] ? agg.v0 : ev[r]);
                agg.v1 = c == 2 ? INF : (agg.v1 < ev[r] ? agg.v1 : ev[r]);
            }
            const MinSt2 inc = wave_incl_ms2(agg);
            const u32 ninc = rv_wave_incl_sum_u32(n01);
            MinSt2 x; x.has = from_lane_below(inc.has, 0u); x.v0 = from_lane_below(inc.v0, INF); x.v1 = from_lane_below(inc.v1, INF);
            x = ms2_combine(car, x);                   // the state in front of this lane's first rank
            u32 e0 = cnt0 + ((ninc - n01) & 0xFFFFu), e1 = cnt1 + ((ninc - n01) >> 16);
            u32 r0 = x.v0, r1 = x.v1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const u32 c = (cls >> (2 * r)) & 3u;
                if (c == 1) {
                    const u32 v = r0 < ev[r] ? r0 : ev[r];
                    ns[S + e0] = pos[r]; nl_[S + e0] = e0 == 0 ? 0u : v; nb[S + e0] = bo[r];
                    e0++;
                } else if (c == 2) {
                    const u32 v = r1 < ev[r] ? r1 : ev[r];
                    ns[S + nlead + e1] = pos[r]; nl_[S + nlead + e1] = e1 == 0 ? 0u : v; nb[S + nlead + e1] = bo[r];
                    e1++;
                }
                r0 = c == 1 ? INF : (r0 < ev[r] ? r0 : ev[r]);
                r1 = c == 2 ? INF : (r1 < ev[r] ? r1 : ev[r]);
            }
            const u32 ntot = (u32)__builtin_amdgcn_readlane((int)ninc, 63);
            cnt0 += ntot & 0xFFFFu; cnt1 += ntot >> 16;
            MinSt2 tot; tot.has = (u32)__builtin_amdgcn_readlane((int)inc.has, 63); tot.v0 = (u32)__builtin_amdgcn_readlane((int)inc.v0, 63); tot.v1 = (u32)__builtin_amdgcn_readlane((int)inc.v1, 63);
            car = ms2_combine(car, tot);
        }
        WSYNC();
        LF_PROF(p_split)
        const int nl = (int)cnt0, ntr = (int)cnt1;
        if (lane == 0 && (cnt0 != nlead || cnt1 != (u32)(TAn + TBn))) atomicOr(A.err, 8u);      // (a sub-index that is not the suffixes of its intervals)
        const int cdepth = f.depth + 1;
        bool do_lead = nl > 0, do_trail = ntr > 0;
        if (!A.trace) {
            // A child without both samples has nothing to match, and neither has one with an interval shorter than minl (bubble_sort
            // keeps every LCP value inside the intervals): counted as visited (reveal.c:1034-1041 / an empty scan), not scanned
            const int64_t need = minl > 1 ? (int64_t)minl : 1;
            if (do_lead && !(la1 - la0 >= need && lb1 - lb0 >= need)) { do_lead = false; if (lane == 0) { my_steps++; if ((u32)cdepth > my_maxdepth) my_maxdepth = (u32)cdepth; } }
            if (do_trail && !(ta1 - ta0 >= need && tb1 - tb0 >= need)) { do_trail = false; if (lane == 0) { my_steps++; if ((u32)cdepth > my_maxdepth) my_maxdepth = (u32)cdepth; } }
        }
        // ---- bubble_sort on the leading child, cuts in ascending order (reveal.c:666-727); a child nobody scans needs none ----
        for (int cut = 0; cut < 2 && do_lead; cut++) {
            const int64_t B = cut == 0 ? pa : pb;
            const int64_t ib = cut == 0 ? la0 : lb0;
            if (!(ib < B)) continue;                                        // no leading interval ends at this cut
            const int64_t wlo = (B - (int64_t)A.lcap > ib) ? B - (int64_t)A.lcap : ib;
            // actives in rank order
            u32 nact = 0;
            for (int base = 0; base < nl; base += 64) {
                const int e = base + lane;
                bool on = false;
                if (e < nl) {
                    const int64_t p = (int64_t)ns[S + e];
                    if (p >= wlo && p < B) {
                        const int64_t l0 = (int64_t)nl_[S + e], l1 = (e + 1 < nl) ? (int64_t)nl_[S + e + 1] : 0;
                        on = p + l0 > B || p + l1 > B;
                    }
                }
                const u64 mask = __ballot(on);
                if (on) act[S + nact + lanes_below(mask)] = (uint16_t)e;
                nact += (u32)__popcll(mask);
            }
            WSYNC();
            for (u32 ai = 0; ai < nact; ai++) {
                const int e = (int)act[S + ai];
                const int64_t p = (int64_t)ns[S + e], l0 = (int64_t)nl_[S + e];          // (the same address for every lane: one broadcast read)
                if (p < B && p + l0 > B) {
                    const int64_t t = B - p; const uint8_t tB = nb[S + e];
                    // x = largest r <= e with r == 0 or LCP[r] < t
                    int x = 0;
                    for (int hi = e;; hi -= 64) {
                        const int r = hi - lane;
                        const u64 mask = __ballot(r >= 0 && (r == 0 || (int64_t)nl_[S + r] < t));
                        if (mask) { x = hi - (int)__builtin_ctzll(mask); break; }
                    }
                    const u32 lnext = (e < nl - 1) ? nl_[S + e + 1] : 0u;
                    // shift [x, e-1] -> [x+1, e], from the top in pieces of 64: a piece reads below what it writes, the wave reads before it writes
                    for (int hi = e; hi > x; hi -= 64) {
                        const int r = hi - lane;
                        sa_t vs = 0; u32 vl = 0; uint8_t vb = 0;
                        if (r > x) { vs = ns[S + r - 1]; vl = nl_[S + r - 1]; vb = nb[S + r - 1]; }
                        WSYNC();
                        if (r > x) { ns[S + r] = vs; nl_[S + r] = vl; nb[S + r] = vb; }
                        WSYNC();
                    }
                    if (lane == 0) {
                        ns[S + x] = (sa_t)p; nb[S + x] = tB;
                        if (x + 1 < nl) nl_[S + x + 1] = (u32)t;
                        if (e < nl - 1 && l0 < (int64_t)lnext) nl_[S + e + 1] = (u32)l0;
                    }
                } else if (e < nl - 1) {
                    const int64_t l1 = (int64_t)nl_[S + e + 1];
                    if (lane == 0 && p < B && p + l1 > B && l1 > l0) nl_[S + e + 1] = (u32)(B - p);
                }
                WSYNC();
            }
        }
        // ---- children (reveal.c:1296-1324); their order is free: this wave goes on with the smaller one, the larger one goes to
        // the stack for any wave (the stack stays O(waves x log n) deep whatever the shape of the tree) -------------------------
        if (lane == 0) {
            const bool keep_lead = do_lead && (!do_trail || nl <= ntr);      // which child this wave goes on with (if any)
            if (do_lead && do_trail) {
                while (atomicCAS(&s_lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const int t = s_top;
                if (t < MAXSTACK) {
                    Frame *o = &stack[t];                                     // the other child
                    o->start = keep_lead ? S + nl : S; o->len = keep_lead ? ntr : nl; o->depth = cdepth; o->buf = b ^ 1;
                    o->a0 = keep_lead ? ta0 : la0; o->a1 = keep_lead ? ta1 : la1; o->b0 = keep_lead ? tb0 : lb0; o->b1 = keep_lead ? tb1 : lb1;
                    s_top = t + 1; atomicAdd(&s_pending, 1);
                } else atomicOr(A.err, 4u);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                atomicExch(&s_lock, 0);
            }
            if (do_lead || do_trail) {
                Frame *o = &cur[wv];
                o->start = keep_lead ? S : S + nl; o->len = keep_lead ? nl : ntr; o->depth = cdepth; o->buf = b ^ 1;
                o->a0 = keep_lead ? la0 : ta0; o->a1 = keep_lead ? la1 : ta1; o->b0 = keep_lead ? lb0 : tb0; o->b1 = keep_lead ? lb1 : tb1;
            } else {
                atomicSub(&s_pending, 1);
            }
        }
        have = do_lead || do_trail;
    }
    if (lane == 0) {
        atomicAdd(&s_stats[0], (unsigned long long)my_steps); atomicAdd(&s_stats[1], (unsigned long long)my_splits);
        atomicAdd(&s_stats[2], (unsigned long long)my_bp); atomicMax(&s_stats[3], (unsigned long long)my_maxdepth);
#ifdef RV_LEAF_PROF
        atomicAdd(&A.stats[4], (unsigned long long)p_idle); atomicAdd(&A.stats[5], (unsigned long long)p_scan);
        atomicAdd(&A.stats[6], (unsigned long long)p_split); atomicAdd(&A.stats[7], (unsigned long long)p_bub);
#endif
    }
    __syncthreads();                                   // every wave has left the loop: the root is finished
    const u32 na = s_na < acap ? s_na : acap;
    if (tid == 0) {
        s_base = na ? atomicAdd(A.anchor_count, na) : 0u;
        atomicAdd(&A.stats[0], s_stats[0]); atomicAdd(&A.stats[1], s_stats[1]); atomicAdd(&A.stats[2], s_stats[2]); atomicMax(&A.stats[3], s_stats[3]);
    }
    __syncthreads();
    for (u32 k = tid; k < na; k += NT) {
        const size_t slot = (size_t)s_base + k;
        if (slot < A.anchor_cap) { A.anchor_l[slot] = an_l[k]; A.anchor_pos[2 * slot] = (int64_t)an_a[k]; A.anchor_pos[2 * slot + 1] = (int64_t)an_b[k]; }
    }
}

// the matched text of the anchors the leaf launches found, lower-cased when the run ends (reveal.c:1230-1234): four anchors per wave -- sixteen lanes
// per anchor, eight per side, sixteen bytes per lane and step.  (One wave per anchor, half a wave per side and eight bytes per lane: 2 x 10^6 waves of
// three dependent trips to memory each -- length, positions, text -- were 1.2 ms at 2 x 250 Mbp for 1 GB of traffic; a byte per lane 1.07 ms for
// 2 x 10^6 anchors of ~120 bases.)  The anchors of a run cover disjoint text, and whole 16-byte pieces never reach beyond the match.
__device__ inline u64 lower8(u64 x) {
    // 0x20 in every byte of 'A' .. 'Z': bit 7 of (b + 0x3F) is set from 'A' on, bit 7 of (b + 0x25) from '[' on (bytes below 0x80)
    const u64 lo7 = x & 0x7F7F7F7F7F7F7F7Full;
    return x | (((lo7 + 0x3F3F3F3F3F3F3F3Full) & ~(lo7 + 0x2525252525252525ull) & ~x & 0x8080808080808080ull) >> 2);
}
// An anchor longer than LOWER_CAP bytes (near-identical inputs: one anchor of 50 Mbp took its eight lanes 183 ms) is only begun here: it goes
// to a short list, and k_leaf_lower_long shares what is left of it among a whole grid.  (A full list: the anchor is finished here after all.)
constexpr int64_t LOWER_CAP = 16384;
constexpr u32 LOWER_LONG_MAX = 4096;
__global__ __launch_bounds__(NT) void k_leaf_lower(uint8_t *__restrict__ T, const int64_t *__restrict__ pos, const u32 *__restrict__ len, u32 na, u32 *__restrict__ lng) {
    const int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x;
    const u32 e = (u32)(t >> 4);
    if (e >= na) return;
    int64_t l = (int64_t)len[e];
    const int side = (int)(t >> 3) & 1, h = (int)t & 7;
    if (l > LOWER_CAP) {      // (both sides and all eight lanes see the same slot: the first lane of the first side takes it, the others learn it through the wave)
        u32 slot = 0;
        if (side == 0 && h == 0) slot = atomicAdd(&lng[0], 1u);
        slot = (u32)__shfl((int)slot, (int)(threadIdx.x & 63) & ~15, 64);
        if (slot < LOWER_LONG_MAX) {
            if (side == 0 && h == 0) lng[4 + slot] = e;
            l = LOWER_CAP;      // (a multiple of sixteen: the tail loop below does nothing)
        }
    }
    uint8_t *const p = T + pos[2 * (size_t)e + side];
    for (int64_t j = (int64_t)h * 16; j + 16 <= l; j += 128) {
        u64 x[2];
        __builtin_memcpy(x, p + j, 16);
        x[0] = lower8(x[0]); x[1] = lower8(x[1]);
        __builtin_memcpy(p + j, x, 16);
    }
    for (int64_t j = (l & ~(int64_t)15) + h; j < l; j += 8) { const uint8_t ch = p[j]; if (ch >= 'A' && ch <= 'Z') p[j] = ch + 32; }
}
// the rest of the long anchors: every workgroup takes 4 KB pieces of every listed anchor in turn (both sides), sixteen bytes per lane and step
__global__ __launch_bounds__(NT) void k_leaf_lower_long(uint8_t *__restrict__ T, const int64_t *__restrict__ pos, const u32 *__restrict__ len, const u32 *__restrict__ lng) {
    const u32 cnt = lng[0] < LOWER_LONG_MAX ? lng[0] : LOWER_LONG_MAX;
    for (u32 k = 0; k < cnt; k++) {
        const u32 e = lng[4 + k];
        const int64_t l = (int64_t)len[e];
        for (int side = 0; side < 2; side++) {
            uint8_t *const p = T + pos[2 * (size_t)e + side];
            for (int64_t j0 = LOWER_CAP + (int64_t)blockIdx.x * (NT * 16); j0 < l; j0 += (int64_t)gridDim.x * (NT * 16)) {
                const int64_t j = j0 + (int64_t)threadIdx.x * 16;
                if (j + 16 <= l) {
                    u64 x[2];
                    __builtin_memcpy(x, p + j, 16);
                    x[0] = lower8(x[0]); x[1] = lower8(x[1]);
                    __builtin_memcpy(p + j, x, 16);
                } else {
                    for (int64_t i = j; i < l; i++) { const uint8_t ch = p[i]; if (ch >= 'A' && ch <= 'Z') p[i] = ch + 32; }
                }
            }
        }
    }
}

}  // namespace

int rv_leaf_launch(Workspace &ws, const RvLeafArgs &a, int nroots) {
    if (nroots <= 0) return 0;
    hipLaunchKernelGGL(k_leaf, dim3((unsigned)nroots), dim3(NT), 0, ws.stream, a);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_leaf_lower_launch(Workspace &ws, uint8_t *T, const int64_t *pos, const u32 *len, u32 na) {
    if (na == 0) return 0;
    RV_TRY(ws.misc[15].reserve((size_t)(4 + LOWER_LONG_MAX) * 4));
    u32 *lng = ws.misc[15].as<u32>();
    RV_HIP(hipMemsetAsync(lng, 0, 16, ws.stream));
    hipLaunchKernelGGL(k_leaf_lower, dim3((unsigned)ceil_div((int64_t)na * 16, NT)), dim3(NT), 0, ws.stream, T, pos, len, na, lng);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_leaf_lower_long, dim3(1024), dim3(NT), 0, ws.stream, T, pos, len, (const u32 *)lng);
    RV_LAUNCH_CHECK();
    return 0;
}

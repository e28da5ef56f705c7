// rv_leaf.hip -- the tail of the recursion inside one workgroup.
//
// The recursion of aligner() (reveallib/reveal.c:731-1338) produces ~10^5
// sub-indices per 10 Mbp, most of a few hundred ranks.  Once a sub-index of a
// two-sample alignment has at most RV_LEAF_N ranks, its whole sub-tree is
// finished here by one workgroup, depth first, with the arrays in LDS:
//   scan   getmums_rem predicate                      reveal.c:119-180
//   pick   built-in picker (longest full match, ties -> smallest coordinate)
//   split  D-label + stable partition + running-min LCP + lower-casing   reveal.c:1005-1234, 582-664
//   bubble bubble_sort on the leading child, cuts in ascending order     reveal.c:666-727
// Same arithmetic as the level kernels (rv_scan.hip, rv_split.hip), so every
// sub-index has the same SA/LCP as in the reference; only used with the built-in
// callbacks (rv_align_builtin), never when Python callbacks drive the recursion.
#include "rv_common.h"
#include "rv_leaf.h"

namespace {

constexpr int NT = 256;
constexpr int LN = RV_LEAF_N;
constexpr u32 INF = 0xFFFFFFFFu;
constexpr int MAXSTACK = 256;

struct Frame { int start, len, depth; int64_t a0, a1, b0, b1; };   // sample-0 interval [a0,a1), sample-1 interval [b0,b1); empty if a0>=a1

__device__ inline bool is_lower_c(uint8_t c) { return c >= 'a' && c <= 'z'; }
__device__ inline bool left_maximal(uint8_t ca, uint8_t cb) { return (ca != cb) || ca == 'N' || ca == '$' || is_lower_c(ca); }

__device__ inline u64 hash_step(u64 acc, u64 i, int64_t v) {      // oracle/reveal_oracle.c ro_hash_step
    u64 x = (u64)v + (i + 1) * 0x9E3779B97F4A7C15ULL;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return acc + x;
}

// block-wide reductions / scans over one value per thread (NT threads, 4 waves)
__device__ inline u64 block_max_u64(u64 v, u64 *lds) {
    for (int d = 32; d >= 1; d >>= 1) { const u64 o = ((u64)__shfl_down((u32)(v >> 32), d, 64) << 32) | __shfl_down((u32)v, d, 64); v = o > v ? o : v; }
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 r = lds[0];
    for (int k = 1; k < NT / 64; k++) r = lds[k] > r ? lds[k] : r;
    __syncthreads();
    return r;
}
__device__ inline u64 block_sum_u64(u64 v, u64 *lds) {
    for (int d = 32; d >= 1; d >>= 1) v += ((u64)__shfl_down((u32)(v >> 32), d, 64) << 32) | __shfl_down((u32)v, d, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 r = 0;
    for (int k = 0; k < NT / 64; k++) r += lds[k];
    __syncthreads();
    return r;
}
__device__ inline int block_max_int(int v, int *lds) {
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_down(v, d, 64); v = o > v ? o : v; }
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = lds[0];
    for (int k = 1; k < NT / 64; k++) r = lds[k] > r ? lds[k] : r;
    __syncthreads();
    return r;
}
// exclusive prefix sum of one u32 per thread; *total = block total
__device__ inline u32 block_excl_u32(u32 v, u32 *lds, u32 *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    u32 before = 0, tot = 0;
    for (int k = 0; k < NT / 64; k++) { const u32 c = lds[k]; if (k < w) before += c; tot += c; }
    __syncthreads();
    *total = tot;
    return before + inc - v;
}
struct MinSt { u32 has, val; };
__device__ inline MinSt ms_combine(MinSt a, MinSt b) { MinSt r; r.has = a.has | b.has; r.val = b.has ? b.val : (a.val < b.val ? a.val : b.val); return r; }
// exclusive scan of the running-minimum state; *total = block aggregate
__device__ inline MinSt block_excl_ms(MinSt v, MinSt *lds, MinSt *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    MinSt inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        MinSt t; t.has = __shfl_up(inc.has, d, 64); t.val = __shfl_up(inc.val, d, 64);
        if (lane >= d) inc = ms_combine(t, inc);
    }
    MinSt exc; exc.has = __shfl_up(inc.has, 1, 64); exc.val = __shfl_up(inc.val, 1, 64);
    if (lane == 0) { exc.has = 0; exc.val = INF; }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    MinSt before = {0, INF}, tot = {0, INF};
    for (int k = 0; k < NT / 64; k++) { if (k < w) before = ms_combine(before, lds[k]); tot = ms_combine(tot, lds[k]); }
    __syncthreads();
    *total = tot;
    return ms_combine(before, exc);
}

__global__ __launch_bounds__(NT) void k_leaf(RvLeafArgs A) {
    __shared__ sa_t  sa[LN], tsa[LN];
    __shared__ u32   lc[LN], tlc[LN];
    __shared__ uint8_t bw[LN], tbw[LN];
    __shared__ Frame stack[MAXSTACK];
    __shared__ u64 r64[NT / 64];
    __shared__ int ri[NT / 64];
    __shared__ u32 ru[NT / 64];
    __shared__ MinSt rm[NT / 64];
    __shared__ int64_t s_pick[2];
    __shared__ int64_t s_v[4];
    __shared__ u32 act[LN];
    __shared__ int s_sp;

    const RvLeafRoot root = A.roots[blockIdx.x];
    const int tid = threadIdx.x;
    for (int i = tid; i < root.n; i += NT) {
        sa[i] = A.SA[root.off + i]; lc[i] = (u32)A.LCP[root.off + i]; bw[i] = A.BWT[root.off + i] & RV_BWT_CHAR;      /* (the side bit is for the streaming scan; here SA is in LDS) */
    }
    if (tid == 0) {
        Frame f; f.start = 0; f.len = (int)root.n; f.depth = root.depth; f.a0 = root.a0; f.a1 = root.a1; f.b0 = root.b0; f.b1 = root.b1;
        stack[0] = f; s_sp = 1;
    }
    __syncthreads();
    const int64_t nsep0 = A.nsep0;
    u32 my_steps = 0, my_splits = 0, my_maxdepth = 0; u64 my_bp = 0;     // accumulated by thread 0

    while (s_sp > 0) {
        const Frame f = stack[s_sp - 1];
        __syncthreads();
        if (tid == 0) s_sp = s_sp - 1;
        const int S = f.start, E = f.start + f.len;
        const bool both = f.a0 < f.a1 && f.b0 < f.b1;           // nsamples == 2 (reveal.c:1034-1041)
        if (tid == 0) { my_steps++; if ((u32)f.depth > my_maxdepth) my_maxdepth = (u32)f.depth; }

        // ---- scan (reveal.c:131-159) + picker ------------------------------------------------
        u64 best = 0;
        u64 hsa = 0, hlc = 0;
        for (int i = S + tid; i < E; i += NT) {
            if (A.trace) { hsa = hash_step(hsa, (u64)(i - S), (int64_t)sa[i]); hlc = hash_step(hlc, (u64)(i - S), (int64_t)lc[i]); }
            if (i == S) continue;
            const u32 l = lc[i];
            if ((int64_t)l < (int64_t)A.minl) continue;
            const sa_t s1 = sa[i], s0 = sa[i - 1];
            if (((int64_t)s1 > nsep0) == ((int64_t)s0 > nsep0)) continue;
            const u32 la = (i + 1 < E) ? lc[i + 1] : 0u;
            if (!(lc[i - 1] < l && la < l)) continue;
            const bool ok = s1 < s0 ? left_maximal(bw[i], bw[i - 1]) : left_maximal(bw[i - 1], bw[i]);
            if (!ok) continue;
            const u64 a = (u64)(s1 < s0 ? s1 : s0);
            const u64 key = ((u64)l << 40) | (0xFFFFFFFFFFull - a);           // longest, then smallest position (< 2^40)
            if (key > best) best = key;
        }
        u64 hm = 0; u32 total_cand = 0;
        if (A.trace) {
            // scan-result hash needs each candidate's ordinal in rank order: second pass with a running count
            u32 run = 0;
            for (int base = S; base < E; base += NT) {
                const int i = base + tid;
                bool ok = false; u32 l = 0; sa_t s1 = 0, s0 = 0;
                if (i < E && i > S) {
                    l = lc[i]; s1 = sa[i]; s0 = sa[i - 1];
                    const u32 la = (i + 1 < E) ? lc[i + 1] : 0u;
                    ok = (int64_t)l >= (int64_t)A.minl && (((int64_t)s1 > nsep0) != ((int64_t)s0 > nsep0)) && lc[i - 1] < l && la < l &&
                         (s1 < s0 ? left_maximal(bw[i], bw[i - 1]) : left_maximal(bw[i - 1], bw[i]));
                }
                u32 tot;
                const u32 k = run + block_excl_u32(ok ? 1u : 0u, ru, &tot);
                if (ok) {
                    const int64_t a = (int64_t)(s1 < s0 ? s1 : s0), b = (int64_t)(s1 < s0 ? s0 : s1);
                    const int64_t seq[6] = {(int64_t)l, 2, 0, a, 1, b};
                    for (int z = 0; z < 6; z++) hm = hash_step(hm, (u64)k * 6 + z, seq[z]);
                }
                run += tot;
            }
            total_cand = run;
            hsa = block_sum_u64(hsa, r64); hlc = block_sum_u64(hlc, r64); hm = block_sum_u64(hm, r64);
        }
        best = block_max_u64(best, r64);
        const bool picked = both && best != 0;
        const u32 L = (u32)(best >> 40);
        const int64_t pa = (int64_t)(0xFFFFFFFFFFull - (best & 0xFFFFFFFFFFull));
        if (picked) {
            // the partner position b of the chosen match
            for (int i = S + 1 + tid; i < E; i += NT) {
                if (lc[i] != L) continue;
                const sa_t s1 = sa[i], s0 = sa[i - 1];
                if ((int64_t)(s1 < s0 ? s1 : s0) != pa) continue;
                if (((int64_t)s1 > nsep0) == ((int64_t)s0 > nsep0)) continue;
                const u32 la = (i + 1 < E) ? lc[i + 1] : 0u;
                if (lc[i - 1] < L && la < L) s_pick[0] = (int64_t)(s1 < s0 ? s0 : s1);
            }
        }
        __syncthreads();
        const int64_t pb = picked ? s_pick[0] : 0;
        if (A.trace && tid == 0) {
            const u32 slot = atomicAdd(A.trace_count, 1u);
            if (slot < A.trace_cap) {
                rv_trace t;
                t.key = f.a0 < f.a1 ? f.a0 : f.b0; t.n = f.len; t.depth = f.depth; t.nsamples = (f.a0 < f.a1) + (f.b0 < f.b1);
                t.nnodes = t.nsamples; t.picked = picked ? 1 : 0; t.nmums = total_cand; t.l = picked ? L : 0; t.mn = picked ? 2 : 0;
                t.sp_min = picked ? pa : 0; t.h_sa = hsa; t.h_lcp = hlc; t.h_mums = hm;
                A.trace_out[slot] = t;
            }
        }
        if (!picked) { __syncthreads(); continue; }
        if (tid == 0) {
            my_splits++; my_bp += L;
            const u32 slot = atomicAdd(A.anchor_count, 1u);
            if (slot < A.anchor_cap) { A.anchor_l[slot] = L; A.anchor_pos[2 * (size_t)slot] = pa; A.anchor_pos[2 * (size_t)slot + 1] = pb; }
        }
        // ---- linear graphalign: lead = left remainders, trail = right remainders ------------------
        const int64_t la0 = f.a0, la1 = pa, lb0 = f.b0, lb1 = pb;                     // leading intervals (may be empty)
        const int64_t ta0 = pa + L, ta1 = f.a1, tb0 = pb + L, tb1 = f.b1;             // trailing intervals
        // ---- label + split (reveal.c:1005-1117, 582-664) -----------------------------------------
        u32 cnt0 = 0, cnt1 = 0;                   // ranks already written to lead / trail
        MinSt car0 = {0, INF}, car1 = {0, INF};   // running-minimum carries
        for (int base = S; base < E; base += NT) {
            const int i = base + tid;
            int c = -1; u32 ev = INF; sa_t pos = 0; uint8_t bo = 0;
            if (i < E) {
                pos = sa[i]; bo = bw[i];
                const int64_t p = (int64_t)pos;
                if ((p >= la0 && p < la1) || (p >= lb0 && p < lb1)) c = 0;
                else if ((p >= ta0 && p < ta1) || (p >= tb0 && p < tb1)) c = 1;
                ev = (i > S) ? lc[i] : INF;      // every rank of a leaf sub-index is labelled (lead, trail or matched): no skipped updates
                if (c == 1 && (p == ta0 || p == tb0) && bo >= 'A' && bo <= 'Z') bo += 32;   // its left neighbour was just matched
            }
            u32 t0, t1; MinSt a0, a1;
            const u32 e0 = block_excl_u32(c == 0 ? 1u : 0u, ru, &t0);
            const u32 e1 = block_excl_u32(c == 1 ? 1u : 0u, ru, &t1);
            MinSt m0; m0.has = c == 0; m0.val = c == 0 ? INF : ev;
            MinSt m1; m1.has = c == 1; m1.val = c == 1 ? INF : ev;
            MinSt x0 = block_excl_ms(m0, rm, &a0);
            MinSt x1 = block_excl_ms(m1, rm, &a1);
            x0 = ms_combine(car0, x0); x1 = ms_combine(car1, x1);
            if (c == 0) {
                const u32 idx = cnt0 + e0;
                const u32 v = x0.val < ev ? x0.val : ev;
                tsa[S + idx] = pos; tlc[S + idx] = idx == 0 ? 0u : v; tbw[S + idx] = bo;
            } else if (c == 1) {
                const u32 idx = cnt1 + e1;
                const u32 v = x1.val < ev ? x1.val : ev;
                // trail goes behind lead: its final place is known only after the loop -> park it from the top of the range
                tsa[E - 1 - idx] = pos; tlc[E - 1 - idx] = idx == 0 ? 0u : v; tbw[E - 1 - idx] = bo;
            }
            cnt0 += t0; cnt1 += t1;
            car0 = ms_combine(car0, a0); car1 = ms_combine(car1, a1);
        }
        __syncthreads();
        const int nl = (int)cnt0, ntr = (int)cnt1;
        for (int i = tid; i < nl; i += NT) { sa[S + i] = tsa[S + i]; lc[S + i] = tlc[S + i]; bw[S + i] = tbw[S + i]; }
        for (int i = tid; i < ntr; i += NT) { sa[S + nl + i] = tsa[E - 1 - i]; lc[S + nl + i] = tlc[E - 1 - i]; bw[S + nl + i] = tbw[E - 1 - i]; }
        // lower-case the matched text (reveal.c:1230-1234)
        for (int64_t j = tid; j < (int64_t)L; j += NT) {
            uint8_t ch = A.T[pa + j]; if (ch >= 'A' && ch <= 'Z') A.T[pa + j] = ch + 32;
            ch = A.T[pb + j]; if (ch >= 'A' && ch <= 'Z') A.T[pb + j] = ch + 32;
        }
        __syncthreads();
        // ---- bubble_sort on the leading child, cuts in ascending order (reveal.c:666-727) ---------------
        for (int cut = 0; cut < 2 && nl > 0; cut++) {
            const int64_t B = cut == 0 ? pa : pb;
            const int64_t ib = cut == 0 ? la0 : lb0;
            if (!(ib < B)) continue;                                        // no leading interval ends at this cut
            const int64_t wlo = (B - (int64_t)A.lcap > ib) ? B - (int64_t)A.lcap : ib;
            // actives in rank order
            u32 nact = 0;
            for (int base = 0; base < nl; base += NT) {
                const int e = base + tid;
                bool on = false;
                if (e < nl) {
                    const int64_t p = (int64_t)sa[S + e];
                    if (p >= wlo && p < B) {
                        const int64_t l0 = (int64_t)lc[S + e], l1 = (e + 1 < nl) ? (int64_t)lc[S + e + 1] : 0;
                        on = p + l0 > B || p + l1 > B;
                    }
                }
                u32 tot;
                const u32 k = nact + block_excl_u32(on ? 1u : 0u, ru, &tot);
                if (on) act[k] = (u32)e;
                nact += tot;
            }
            __syncthreads();
            for (u32 ai = 0; ai < nact; ai++) {
                const int e = (int)act[ai];
                if (tid == 0) {
                    const int64_t p = (int64_t)sa[S + e], l0 = (int64_t)lc[S + e];
                    int64_t kind = 0;
                    if (p < B && p + l0 > B) kind = 1;
                    else if (e < nl - 1) { const int64_t l1 = (int64_t)lc[S + e + 1]; if (p < B && p + l1 > B && l1 > l0) lc[S + e + 1] = (u32)(B - p); }
                    s_v[0] = kind; s_v[1] = p; s_v[2] = l0; s_v[3] = bw[S + e];
                }
                __syncthreads();
                if (s_v[0] == 1) {
                    const int64_t tS = s_v[1], tL = s_v[2], t = B - tS; const uint8_t tB = (uint8_t)s_v[3];
                    // x = largest r <= e with r == 0 or LCP[r] < t
                    int bestr = -1;
                    for (int r = e - tid; r >= 0; r -= NT) if (r == 0 || (int64_t)lc[S + r] < t) { bestr = r; break; }
                    const int x = block_max_int(bestr, ri);
                    // shift [x, e-1] -> [x+1, e]: read everything, then write
                    sa_t vs[LN / NT]; u32 vl[LN / NT]; uint8_t vb[LN / NT];
#pragma unroll
                    for (int k = 0; k < LN / NT; k++) { const int r = e - k * NT - tid; if (r > x) { vs[k] = sa[S + r - 1]; vl[k] = lc[S + r - 1]; vb[k] = bw[S + r - 1]; } }
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < LN / NT; k++) { const int r = e - k * NT - tid; if (r > x) { sa[S + r] = vs[k]; lc[S + r] = vl[k]; bw[S + r] = vb[k]; } }
                    __syncthreads();
                    if (tid == 0) {
                        sa[S + x] = (sa_t)tS; bw[S + x] = tB;
                        if (x + 1 < nl) lc[S + x + 1] = (u32)t;
                        if (e < nl - 1 && tL < (int64_t)lc[S + e + 1]) lc[S + e + 1] = (u32)tL;
                    }
                }
                __syncthreads();
            }
        }
        // ---- children (reveal.c:1296-1324): trailing first so the leading child is handled next; order is free ----
        if (tid == 0) {
            int sp = s_sp;
            if (ntr > 0 && sp < MAXSTACK) { Frame c; c.start = S + nl; c.len = ntr; c.depth = f.depth + 1; c.a0 = ta0; c.a1 = ta1; c.b0 = tb0; c.b1 = tb1; stack[sp++] = c; }
            if (nl > 0 && sp < MAXSTACK) { Frame c; c.start = S; c.len = nl; c.depth = f.depth + 1; c.a0 = la0; c.a1 = la1; c.b0 = lb0; c.b1 = lb1; stack[sp++] = c; }
            if (sp >= MAXSTACK) atomicOr(A.err, 4u);
            s_sp = sp;
        }
        __syncthreads();
    }
    if (tid == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)my_steps);
        atomicAdd(&A.stats[1], (unsigned long long)my_splits);
        atomicAdd(&A.stats[2], (unsigned long long)my_bp);
        atomicMax(&A.stats[3], (unsigned long long)my_maxdepth);
    }
}

}  // namespace

int rv_leaf_launch(Workspace &ws, const RvLeafArgs &a, int nroots) {
    if (nroots <= 0) return 0;
    hipLaunchKernelGGL(k_leaf, dim3((unsigned)nroots), dim3(NT), 0, ws.stream, a);
    RV_LAUNCH_CHECK();
    return 0;
}

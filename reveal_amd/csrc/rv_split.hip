// rv_split.hip -- one recursion level of aligner() for ALL sub-indices at once:
// D-label, split, lower-casing and bubble_sort (reveallib/reveal.c:1005-1252,
// split :582-664, bubble_sort :666-727) as segmented kernels over the
// concatenated level arrays.
//
//   k_label        D[i] = label of text position SA[i]  (gather form of the
//                  scatter at reveal.c:1024-1116; 1 lead, 2 trail, 4 rest, 3 matched)
//   k_split<false> per 2048-rank tile: class counts + running-min summaries
//   k_tile_carry   exclusive scan of those summaries over tiles (one block)
//   k_seg_offsets  per split sub-index: class counts before its first rank ->
//                  destination offsets of its three children (+ size check)
//   k_split<true>  stable 3-way partition: child SA, child LCP (running minimum
//                  of the parent LCP since the previous rank of the same child,
//                  exactly the minlcp* bookkeeping of reveal.c:636-662 incl.
//                  its `continue` for unlabelled ranks), windowed SAi
//   k_lower        T[j] = tolower(T[j]) over the matched ranges (reveal.c:1230-1234)
//   k_bubble_*     bubble_sort on every leading child, one cut (matched begin)
//                  per round, in the order graphalign listed them
#include "rv_common.h"
#include "rv_split.h"
#include <algorithm>
#include <stdlib.h>

namespace {

constexpr int TB = 256;
constexpr int SP_ITEMS = 8;
constexpr int SP_TILE = TB * SP_ITEMS;      // == RV_SPLIT_TILE
constexpr u32 INF = 0xFFFFFFFFu;

__device__ inline int cls_index(uint8_t d) { return d == 1 ? 0 : d == 2 ? 1 : d == 4 ? 2 : -1; }

// ---- label -------------------------------------------------------------------
template <class P>
__device__ inline int upper_idx(const P *__restrict__ begins, int n, P pos) {   // last idx with begins[idx] <= pos, or -1
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (begins[mid] <= pos) lo = mid + 1; else hi = mid; }
    return lo - 1;
}

// Each sub-index brings its own short, begin-sorted interval list (typically
// <= 3 * nsamples entries), so a rank only searches the list of the sub-index
// it belongs to: one binary search over the sub-index starts per thread, then
// a few compares per rank.
template <class P>
__device__ inline int find_in(const P *__restrict__ begins, int first, int last, P pos) {   // last idx in [first,last) with begins[idx] <= pos, or -1
    if (last - first <= 8) {
        int e = -1;
        for (int k = first; k < last; k++) if (begins[k] <= pos) e = k;
        return e;
    }
    const int r = upper_idx<P>(begins + first, last - first, pos);
    return r < 0 ? -1 : first + r;
}

// number of entries of the ascending array a[0..n) that are <= key, by one wave (64 probes per step)
__device__ inline int wave_count_le(const int64_t *__restrict__ a, int n, int64_t key) {
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int step = (hi - lo + 63) / 64;
        const int p = lo + lane * step;
        const bool le = p < hi && a[p] <= key;
        const int c = (int)__popcll(__ballot(le));
        if (c == 0) { hi = lo; break; }
        const int last = lo + (c - 1) * step;
        const int nxt = last + step;
        lo = last + 1;
        hi = nxt < hi ? nxt : hi;
    }
    return lo;
}

// D-label of text position pos inside sub-index s (gather form of reveal.c:1024-1116)
__device__ inline uint8_t label_of(const RvLabelTabs &t, int s, sa_t pos) {
    uint8_t c = 0;
    int e = find_in<sa_t>(t.cbegin, t.ctab_first[s], t.ctab_first[s + 1], pos);
    if (e >= 0 && pos < t.cend[e]) c = t.ccls[e];
    e = find_in<sa_t>(t.mbegin, t.mtab_first[s], t.mtab_first[s + 1], pos);
    if (e >= 0 && pos < t.mend[e]) c = 3;
    return c;
}

// ---- split -------------------------------------------------------------------
struct MinSt { u32 has, val; };
__device__ inline MinSt ms_combine(MinSt a, MinSt b) {   // a then b
    MinSt r;
    r.has = a.has | b.has;
    r.val = b.has ? b.val : (a.val < b.val ? a.val : b.val);
    return r;
}

#ifdef RV_SA64
typedef u64 usa_t;
#else
typedef u32 usa_t;
#endif

// Thread summaries of SP_ITEMS ranks and their wave-inclusive scans.  dpack = the labels of my ranks (one byte each),
// ev[k] = effective LCP of rank k (INF where the reference skips the min update).  Everything is written with selects:
// the && / if form compiled to an exec-mask branch per term (3500 instructions and 280 branches per thread in the
// counting pass) and made both passes instruction bound.
__device__ inline void split_summaries(u64 dpack, const u32 *ev, u32 *icnt, MinSt *ist) {
    const int lane = threadIdx.x & 63;
    u32 pc = 0, ph = 0;
    u32 v0 = INF, v1 = INF, v2 = INF;
#pragma unroll
    for (int k = 0; k < SP_ITEMS; k++) {
        const u32 d = (u32)(dpack >> (8 * k)) & 0xffu;
        const bool i0 = d == 1u, i1 = d == 2u, i2 = d == 4u;
        const u32 e = ev[k];
        v0 = i0 ? INF : (v0 < e ? v0 : e);
        v1 = i1 ? INF : (v1 < e ? v1 : e);
        v2 = i2 ? INF : (v2 < e ? v2 : e);
        pc += (u32)i0 + ((u32)i1 << 10) + ((u32)i2 << 20);
        ph |= (u32)i0 | ((u32)i1 << 1) | ((u32)i2 << 2);
    }
    // wave-inclusive scan; the three counts (<= 512 each) share one word and the three has-bits another
#ifdef RV_SPLIT_SHFL
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const u32 tc = __shfl_up(pc, dd, 64), th = __shfl_up(ph, dd, 64);
        const u32 t0 = __shfl_up(v0, dd, 64), t1 = __shfl_up(v1, dd, 64), t2 = __shfl_up(v2, dd, 64);
        const bool act = lane >= dd;
        const u32 m0 = t0 < v0 ? t0 : v0, m1 = t1 < v1 ? t1 : v1, m2 = t2 < v2 ? t2 : v2;
        v0 = (act & !(ph & 1u)) ? m0 : v0;
        v1 = (act & !(ph & 2u)) ? m1 : v1;
        v2 = (act & !(ph & 4u)) ? m2 : v2;
        pc += act ? tc : 0u;
        ph |= act ? th : 0u;
    }
#else
    // by DPP (rv_common.h): five words per step through the crossbar of the LDS pipe was a third of both passes' instructions
#define SP_STEP_(CTRL, RM, TAKE) {                                                                                   \
        const u32 tc = rv_dpp_u32<CTRL, RM>(pc), th = rv_dpp_u32<CTRL, RM>(ph);                                      \
        const u32 t0 = rv_dpp_u32<CTRL, RM>(v0), t1 = rv_dpp_u32<CTRL, RM>(v1), t2 = rv_dpp_u32<CTRL, RM>(v2);      \
        const bool act = (TAKE);                                                                                     \
        const u32 m0 = t0 < v0 ? t0 : v0, m1 = t1 < v1 ? t1 : v1, m2 = t2 < v2 ? t2 : v2;                            \
        v0 = (act & !(ph & 1u)) ? m0 : v0;                                                                           \
        v1 = (act & !(ph & 2u)) ? m1 : v1;                                                                           \
        v2 = (act & !(ph & 4u)) ? m2 : v2;                                                                           \
        pc += act ? tc : 0u;                                                                                         \
        ph |= act ? th : 0u;                                                                                         \
    }
    RV_WAVE_SCAN_STEPS(SP_STEP_)
#undef SP_STEP_
#endif
    icnt[0] = pc & 1023u; icnt[1] = (pc >> 10) & 1023u; icnt[2] = pc >> 20;
    ist[0].has = ph & 1u; ist[1].has = (ph >> 1) & 1u; ist[2].has = (ph >> 2) & 1u;
    ist[0].val = v0; ist[1].val = v1; ist[2].val = v2;
}

// labels of a thread's SP_ITEMS ranks from a short interval table held in registers, straight-line: RC class intervals, RM matched ranges
template <int RC, int RM>
__device__ __forceinline__ u64 label_straight(const RvLabelTabs &t, int c0, int nc, int m0, int nm, const sa_t *sav) {
    usa_t tb[RC + RM], tl[RC + RM]; u32 tc[RC + RM];      // (begin, length, class); unused slots have length 0
#pragma unroll
    for (int q = 0; q < RC; q++) {
        const bool in = q < nc;
        const sa_t b = in ? t.cbegin[c0 + q] : (sa_t)0, e = in ? t.cend[c0 + q] : (sa_t)0;
        tb[q] = (usa_t)b; tl[q] = (usa_t)(e - b); tc[q] = in ? (u32)t.ccls[c0 + q] : 0u;
    }
#pragma unroll
    for (int q = 0; q < RM; q++) {
        const bool in = q < nm;
        const sa_t b = in ? t.mbegin[m0 + q] : (sa_t)0, e = in ? t.mend[m0 + q] : (sa_t)0;
        tb[RC + q] = (usa_t)b; tl[RC + q] = (usa_t)(e - b); tc[RC + q] = 3u;
    }
    u64 dpack = 0;
#pragma unroll
    for (int k = 0; k < SP_ITEMS; k++) {
        const usa_t pos = (usa_t)sav[k];
        u32 c = 0;
#pragma unroll
        for (int q = 0; q < RC + RM; q++) c = ((usa_t)(pos - tb[q]) < tl[q]) ? tc[q] : c;      // matched ranges last: they win
        dpack |= (u64)c << (8 * k);
    }
    return dpack;
}

// Pass 1: D-labels of a 2048-rank tile (written for pass 2) and, from them, the tile's class counts and
// running-minimum summaries.  The sub-index of the tile's first rank comes from a host table; a thread whose
// eight ranks lie in one sub-index with a short interval table (two samples: <= 6 class intervals, 2 matched
// ranges) labels them from registers, straight-line; the rest (a sub-index boundary inside the eight ranks,
// many samples) takes the generic loop.
__global__ __launch_bounds__(TB) void k_split_count(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, int64_t m, RvLabelTabs t, RvSplitArgs a,
                                                    uint8_t *__restrict__ D) {
    __shared__ u32   s_cnt[TB / 64][3];
    __shared__ MinSt s_ms[TB / 64][3];
    __shared__ uint8_t s_last[TB / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    const int64_t tlo = tile * SP_TILE;
    const int s_sub0 = t.tile_sub[tile];
    const int64_t j0 = tlo + (int64_t)threadIdx.x * SP_ITEMS;
    int s = s_sub0;
    u32 ev[SP_ITEMS];
    sa_t sav[SP_ITEMS];
    u32 lcv[SP_ITEMS];
    const bool whole = j0 + SP_ITEMS <= m;
    if (whole) {      // vector loads (j0 is a multiple of SP_ITEMS, the level arrays are 16-byte aligned)
        __builtin_memcpy(sav, SA + j0, sizeof sav);
        __builtin_memcpy(lcv, LCP + j0, sizeof lcv);
    } else {
#pragma unroll
        for (int k = 0; k < SP_ITEMS; k++) { const int64_t i = j0 + k; sav[k] = i < m ? SA[i] : (sa_t)0; lcv[k] = i < m ? (u32)LCP[i] : INF; }
    }
    u64 dpack = 0;
    if (j0 < m) {
        int64_t s_end = t.sub_start[s + 1];
        for (int step = 0; j0 >= s_end && step < 4; step++) { s++; s_end = t.sub_start[s + 1]; }
        if (j0 >= s_end) { s += upper_idx<int64_t>(t.sub_start + s, t.nsubs - s, j0); s_end = t.sub_start[s + 1]; }
        const int c0 = t.ctab_first[s], m0 = t.mtab_first[s];
        const int nc = t.ctab_first[s + 1] - c0, nm = t.mtab_first[s + 1] - m0;
        if (whole && j0 + SP_ITEMS <= s_end && nc <= 6 && nm <= 2) {
            // (two samples, one interval each: four class intervals at most -- a quarter fewer tests per rank than the general six)
            dpack = nc <= 4 ? label_straight<4, 2>(t, c0, nc, m0, nm, sav) : label_straight<6, 2>(t, c0, nc, m0, nm, sav);
        } else {
#pragma unroll 1
            for (int k = 0; k < SP_ITEMS; k++) {
                const int64_t i = j0 + k;
                if (i >= m) break;
                while (i >= s_end) { s++; s_end = t.sub_start[s + 1]; }
                dpack |= (u64)label_of(t, s, SA[i]) << (8 * k);
            }
        }
    }
    // label of the rank in front of my first one
    u32 dprev = 0;
    {
        const u32 mylast = (u32)(dpack >> (8 * (SP_ITEMS - 1))) & 0xffu;
        const u32 up = (u32)__shfl_up((int)mylast, 1, 64);
        if (lane == 63) s_last[w] = (uint8_t)mylast;
        __syncthreads();
        if (lane > 0) dprev = up;
        else if (w > 0) dprev = s_last[w - 1];
        else if (j0 > 0 && j0 - 1 < m) {                          // first thread of the tile: label the rank before the tile
            const int sp = (j0 - 1 >= t.sub_start[s_sub0]) ? s_sub0 : s_sub0 - 1;
            dprev = label_of(t, sp, SA[j0 - 1]);
        }
    }
    {
        const u64 dsh = (dpack << 8) | dprev;                      // byte k = label of the rank in front of rank k
#pragma unroll
        for (int k = 0; k < SP_ITEMS; k++) ev[k] = (((u32)(dsh >> (8 * k)) & 0xffu) != 0u) ? lcv[k] : INF;      // (past the end: lcv == INF)
    }
    if (whole) {
        *reinterpret_cast<u64 *>(D + j0) = dpack;
    } else {
        for (int k = 0; k < SP_ITEMS && j0 + k < m; k++) D[j0 + k] = (uint8_t)(dpack >> (8 * k));
    }
    u32 icnt[3]; MinSt ist[3];
    split_summaries(dpack, ev, icnt, ist);
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < 3; c++) { s_cnt[w][c] = icnt[c]; s_ms[w][c] = ist[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        u32 tot = 0; MinSt ms = {0, INF};
        for (int k = 0; k < TB / 64; k++) { tot += s_cnt[k][c]; ms = ms_combine(ms, s_ms[k][c]); }
        a.tile_cnt[(size_t)c * a.ntiles + tile] = tot;
        a.tile_has[(size_t)c * a.ntiles + tile] = ms.has;
        a.tile_post[(size_t)c * a.ntiles + tile] = ms.val;
    }
}

// Pass 2: stable 3-way partition.  A tile's survivors are staged in LDS class by class and leave as
// consecutive stores (one thread per survivor): written straight from the scan order, every store
// instruction touched 64 different cache lines.
struct EmitSub {            // per-sub-index tables of pass 2, in registers (scalars, not arrays: a select between
                            // array elements becomes an indexed load and sends the whole struct to scratch)
    u32 cbase0, cbase1, cbase2, coff0, coff1, coff2, cn0, cn1, cn2;
    int qc0, qc1, qm0, qm1;
    sa_t clo0, clo1, chi0, chi1, mnd0, mnd1;
};
__device__ __forceinline__ void emit_load_sub(const RvSplitArgs &a, int ss, EmitSub &e) {
    e.cbase0 = a.child_base[(size_t)ss * 3]; e.cbase1 = a.child_base[(size_t)ss * 3 + 1]; e.cbase2 = a.child_base[(size_t)ss * 3 + 2];
    e.coff0 = a.sub_off[(size_t)ss * 3]; e.coff1 = a.sub_off[(size_t)ss * 3 + 1]; e.coff2 = a.sub_off[(size_t)ss * 3 + 2];
    e.cn0 = a.child_n[(size_t)ss * 3]; e.cn1 = a.child_n[(size_t)ss * 3 + 1]; e.cn2 = a.child_n[(size_t)ss * 3 + 2];
    e.qc0 = a.cut_first[ss]; e.qc1 = a.cut_first[ss + 1]; e.qm0 = a.mend_first[ss]; e.qm1 = a.mend_first[ss + 1];
    e.clo0 = (e.qc0 < e.qc1) ? a.cut_lo[e.qc0] : (sa_t)0;         e.chi0 = (e.qc0 < e.qc1) ? a.cut_hi[e.qc0] : (sa_t)0;
    e.clo1 = (e.qc0 + 1 < e.qc1) ? a.cut_lo[e.qc0 + 1] : (sa_t)0; e.chi1 = (e.qc0 + 1 < e.qc1) ? a.cut_hi[e.qc0 + 1] : (sa_t)0;
    e.mnd0 = (e.qm0 < e.qm1) ? a.mend_pos[e.qm0] : (sa_t)-1;      e.mnd1 = (e.qm0 + 1 < e.qm1) ? a.mend_pos[e.qm0 + 1] : (sa_t)-1;
}
// one rank of pass 2; d = its label, e = its effective LCP; r0..r2 running minima, n0..n2 global class counts, l0..l2 LDS slots
__device__ __forceinline__ void emit_item(const RvSplitArgs &a, const EmitSub &sb, const u32 coff0, const u32 coff1, const u32 coff2,
                                          const u32 cbase0, const u32 cbase1, const u32 cbase2, const u32 cn0, const u32 cn1, const u32 cn2,
                                          u32 d, u32 e, sa_t sa, uint8_t bo,
                                 u32 &r0, u32 &r1, u32 &r2, u32 &n0, u32 &n1, u32 &n2, u32 &l0, u32 &l1, u32 &l2,
                                 sa_t *o_sa, u32 *o_lcp, u32 *o_np, uint8_t *o_bw) {
    r0 = r0 < e ? r0 : e; r1 = r1 < e ? r1 : e; r2 = r2 < e ? r2 : e;
    const bool i0 = d == 1u, i1 = d == 2u, i2 = d == 4u;
    if (i0 | i1 | i2) {
        const u32 run = i0 ? r0 : i1 ? r1 : r2;
        const u32 ecnt = i0 ? n0 : i1 ? n1 : n2;
        const u32 slot = i0 ? l0 : i1 ? l1 : l2;
        const u32 coff = i0 ? coff0 : i1 ? coff1 : coff2;
        const u32 cbase = i0 ? cbase0 : i1 ? cbase1 : cbase2;
        const u32 cn = i0 ? cn0 : i1 ? cn1 : cn2;
        const u32 np = coff + ecnt;                                  // mod 2^32
        const u32 idx = np - cbase;                                  // rank inside the child
        if (i1 | (a.mend_all != 0)) {   // trailing child (the linear interval model puts nothing else there): the character in front
                    // of a suffix that starts right behind a matched range has just been lower-cased (reveal.c:1230-1234)
            bool hit = (sa == sb.mnd0) | (sa == sb.mnd1);
            for (int q = sb.qm0 + 2; q < sb.qm1 && !hit; q++) hit = sa == a.mend_pos[q];
            const uint8_t ch = bo & RV_BWT_CHAR;                      // (bit 7 = side of the separator, kept)
            if (hit && ch >= 'A' && ch <= 'Z') bo += 32;
        }
        if (idx >= cn) {
            atomicOr(a.err, 1u);                                     // more ranks labelled for this child than its intervals hold
            o_np[slot] = 0xFFFFFFFFu;
        } else {
            o_sa[slot] = sa; o_lcp[slot] = idx == 0 ? 0u : run; o_bw[slot] = bo; o_np[slot] = np;
            if (i0) {   // leading child: publish SAi where bubble_sort will look (windows before its cuts)
                bool hit = ((sa >= sb.clo0) & (sa < sb.chi0)) | ((sa >= sb.clo1) & (sa < sb.chi1));
                for (int q = sb.qc0 + 2; q < sb.qc1 && !hit; q++) hit = sa >= a.cut_lo[q] && sa < a.cut_hi[q];
                if (hit) a.SAi[sa] = (sa_t)idx;
            }
        }
        n0 += i0; n1 += i1; n2 += i2;
        l0 += i0; l1 += i1; l2 += i2;
        r0 = i0 ? INF : r0; r1 = i1 ? INF : r1; r2 = i2 ? INF : r2;
    }
}

__global__ __launch_bounds__(TB) void k_split_emit(const sa_t *__restrict__ SA, const lcp_t *__restrict__ LCP, const uint8_t *__restrict__ D,
                                                   const uint8_t *__restrict__ BWT, int64_t m, RvSplitArgs a) {
    __shared__ u32   s_cnt[TB / 64][3];
    __shared__ MinSt s_ms[TB / 64][3];
    __shared__ sa_t  o_sa[SP_TILE];
    __shared__ u32   o_lcp[SP_TILE], o_np[SP_TILE];
    __shared__ uint8_t o_bw[SP_TILE];
    __shared__ u32   t_key[16], t_val[16];      // tile bounds of this workgroup's output: (output tile, minimum), open addressing
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    const int64_t j0 = tile * SP_TILE + (int64_t)threadIdx.x * SP_ITEMS;
    if (threadIdx.x < 16) { t_key[threadIdx.x] = 0xFFFFFFFFu; t_val[threadIdx.x] = INF; }
    static_assert(SP_ITEMS == 8, "labels / BWT bytes of a thread travel as one 64-bit word");

    u32 ev[SP_ITEMS];             // effective LCP (INF where the reference skips the min update)
    sa_t sa[SP_ITEMS];
    u64 dpack = 0, bpack = 0;
    const u32 dprev = (j0 > 0 && j0 - 1 < m) ? (u32)D[j0 - 1] : 0u;
    const bool whole = j0 + SP_ITEMS <= m;
    {
        u32 lcv[SP_ITEMS];
        if (whole) {
            __builtin_memcpy(sa, SA + j0, sizeof sa);
            __builtin_memcpy(lcv, LCP + j0, sizeof lcv);
            dpack = *reinterpret_cast<const u64 *>(D + j0); bpack = *reinterpret_cast<const u64 *>(BWT + j0);
        } else {
#pragma unroll
            for (int k = 0; k < SP_ITEMS; k++) {
                const int64_t j = j0 + k;
                const bool in = j < m;
                sa[k] = in ? SA[j] : (sa_t)0; lcv[k] = in ? (u32)LCP[j] : INF;
                dpack |= (u64)(in ? D[j] : (uint8_t)0) << (8 * k); bpack |= (u64)(in ? BWT[j] : (uint8_t)0) << (8 * k);
            }
        }
        const u64 dsh = (dpack << 8) | dprev;
#pragma unroll
        for (int k = 0; k < SP_ITEMS; k++) ev[k] = (((u32)(dsh >> (8 * k)) & 0xffu) != 0u) ? lcv[k] : INF;
    }
    u32 icnt[3]; MinSt ist[3];
    split_summaries(dpack, ev, icnt, ist);
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < 3; c++) { s_cnt[w][c] = icnt[c]; s_ms[w][c] = ist[c]; }
    }
    __syncthreads();
    // exclusive prefixes for this thread: tile carry-in, earlier waves, earlier lanes; lds = slot in the staging arrays
    u32 ecnt[3], lds[3], tot[3]; MinSt est[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const u32 g = a.tile_G[(size_t)c * a.ntiles + tile];
        u32 bc = 0; tot[c] = 0;
        MinSt bm; bm.has = 0; bm.val = a.tile_carry[(size_t)c * a.ntiles + tile];
#pragma unroll
        for (int k = 0; k < TB / 64; k++) {
            const bool before = k < w;
            const MinSt comb = ms_combine(bm, s_ms[k][c]);
            bc += before ? s_cnt[k][c] : 0u;
            bm.has = before ? comb.has : bm.has; bm.val = before ? comb.val : bm.val;
            tot[c] += s_cnt[k][c];
        }
        u32 xc = __shfl_up(icnt[c], 1, 64);
        MinSt xm; xm.has = __shfl_up(ist[c].has, 1, 64); xm.val = __shfl_up(ist[c].val, 1, 64);
        if (lane == 0) { xc = 0; xm.has = 0; xm.val = INF; }
        ecnt[c] = g + bc + xc;
        lds[c] = bc + xc;
        est[c] = ms_combine(bm, xm);
    }
    lds[1] += tot[0]; lds[2] += tot[0] + tot[1];
    if (j0 < m) {
        int s = a.tile_sub[tile];
        int64_t s_end = a.sub_start[s + 1];
        for (int step = 0; j0 >= s_end && step < 4; step++) { s++; s_end = a.sub_start[s + 1]; }
        if (j0 >= s_end) { s += upper_idx<int64_t>(a.sub_start + s, a.nsubs - s, j0); s_end = a.sub_start[s + 1]; }
        u32 r0 = est[0].val, r1 = est[1].val, r2 = est[2].val;
        u32 n0 = ecnt[0], n1 = ecnt[1], n2 = ecnt[2], l0 = lds[0], l1 = lds[1], l2 = lds[2];
        EmitSub sb;
        emit_load_sub(a, s, sb);
        if (whole && j0 + SP_ITEMS <= s_end) {        // all my ranks in one sub-index: straight-line
#pragma unroll
            for (int k = 0; k < SP_ITEMS; k++)
                emit_item(a, sb, sb.coff0, sb.coff1, sb.coff2, sb.cbase0, sb.cbase1, sb.cbase2, sb.cn0, sb.cn1, sb.cn2, (u32)(dpack >> (8 * k)) & 0xffu, ev[k], sa[k], (uint8_t)(bpack >> (8 * k)), r0, r1, r2, n0, n1, n2, l0, l1, l2, o_sa, o_lcp, o_np, o_bw);
        } else {
            u32 dp = dprev;
#pragma unroll 1
            for (int k = 0; k < SP_ITEMS; k++) {
                const int64_t j = j0 + k;
                if (j >= m) break;
                if (j >= s_end) { do { s++; s_end = a.sub_start[s + 1]; } while (j >= s_end); emit_load_sub(a, s, sb); }
                const u32 d = (u32)(dpack >> (8 * k)) & 0xffu;
                const u32 e = dp != 0u ? (u32)LCP[j] : INF;              // (re-read: a register array cannot be indexed by k here)
                emit_item(a, sb, sb.coff0, sb.coff1, sb.coff2, sb.cbase0, sb.cbase1, sb.cbase2, sb.cn0, sb.cn1, sb.cn2, d, e, SA[j], (uint8_t)(bpack >> (8 * k)), r0, r1, r2, n0, n1, n2, l0, l1, l2, o_sa, o_lcp, o_np, o_bw);
                dp = d;
            }
        }
    }
    __syncthreads();
    const u32 total = tot[0] + tot[1] + tot[2];
    // Tile bounds (a.tmin_out): a lower bound of the LCP values per RV_SPLIT_TILE ranks of the output arrays.  A thread's slots are
    // 256 ranks apart inside long runs, so its output tile changes every eighth slot at most: it keeps one (tile, minimum) pair
    // in registers and hands it to a 16-entry LDS table (open addressing) when the tile changes and at the end; the table leaves
    // as one global atomic per (workgroup, output tile).  (Wave-wide reductions per 64 slots -- by shuffles or by DPP -- doubled
    // the time of this pass: a chain of cross-lane operations and a single lane's atomics in front of every store.)
    u32 ck = 0xFFFFFFFFu, cv = INF;
    auto flush = [&](u32 key, u32 v) {
        bool done = false;
        for (u32 i = key & 15u, tries = 0; tries < 16 && !done; i = (i + 1) & 15u, tries++) {
            const u32 cur = t_key[i];
            const u32 old = cur == key ? key : atomicCAS(&t_key[i], 0xFFFFFFFFu, key);
            if (old == 0xFFFFFFFFu || old == key) { atomicMin(&t_val[i], v); done = true; }
        }
        if (!done) atomicMin(&a.tmin_out[key], v);
    };
    for (u32 q = threadIdx.x; q < total; q += TB) {
        const u32 np = o_np[q];
        if (np == 0xFFFFFFFFu) continue;
        const u32 lc = o_lcp[q];
        a.SA_out[np] = o_sa[q];
        a.LCP_out[np] = (lcp_t)lc;
        a.BWT_out[np] = o_bw[q];
        if (a.tmin_out && q < tot[0]) {       // (only leading children are bubble-sorted: the staging area holds their ranks first)
            const u32 key = np >> 11;
            if (key != ck) { if (ck != 0xFFFFFFFFu) flush(ck, cv); ck = key; cv = lc; }
            else cv = lc < cv ? lc : cv;
        }
    }
    if (a.tmin_out) {
        if (ck != 0xFFFFFFFFu) flush(ck, cv);
        __syncthreads();
        if (threadIdx.x < 16 && t_key[threadIdx.x] != 0xFFFFFFFFu) atomicMin(&a.tmin_out[t_key[threadIdx.x]], t_val[threadIdx.x]);
    }
}

// exclusive scan over tiles of (count, min-state) for the three classes; one block of NT threads, each thread
// owning CARRY_PER consecutive tiles per pass.  The summaries come in through LDS with lane-contiguous loads
// and leave the same way: a thread reading its own eight consecutive words straight from global memory made
// every load instruction touch 32 cache lines on the one CU this runs on (37 us for 4900 tiles).
constexpr int CARRY_PER = 8;
template <int NT, int NC>
__device__ inline void carry_scan_range(const RvSplitArgs &a, int c0, int64_t t_lo, int64_t t_hi, u32 *s_runc, MinSt *s_runm, u32 (*s_c)[3], MinSt (*s_m)[3],
                                        u32 *s_x, u32 *s_y) {      // s_x, s_y: NT*CARRY_PER words each
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int PASS = NT * CARRY_PER;
    for (int64_t base = t_lo; base < t_hi; base += PASS) {
        const int64_t t0 = base + (int64_t)threadIdx.x * CARRY_PER;
        u32 ix[NC]; MinSt im[NC];
        u32 vc[NC][CARRY_PER]; MinSt vm[NC][CARRY_PER];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            // stage class c: x = count | has << 31 (a tile holds 2048 ranks), y = min value
            __syncthreads();
            for (int k = threadIdx.x; k < PASS; k += NT) {
                const int64_t t = base + k;
                const bool in = t < t_hi;
                s_x[k] = in ? (a.tile_cnt[(size_t)(c0 + c) * a.ntiles + t] | (a.tile_has[(size_t)(c0 + c) * a.ntiles + t] << 31)) : 0u;
                s_y[k] = in ? a.tile_post[(size_t)(c0 + c) * a.ntiles + t] : INF;
            }
            __syncthreads();
            ix[c] = 0; im[c].has = 0; im[c].val = INF;
#pragma unroll
            for (int k = 0; k < CARRY_PER; k++) {
                const u32 x = s_x[threadIdx.x * CARRY_PER + k];
                vc[c][k] = x & 0x7FFFFFFFu; vm[c][k].has = x >> 31; vm[c][k].val = s_y[threadIdx.x * CARRY_PER + k];
                ix[c] += vc[c][k]; im[c] = ms_combine(im[c], vm[c][k]);
            }
        }
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const u32 tc = __shfl_up(ix[c], dd, 64);
                MinSt tm; tm.has = __shfl_up(im[c].has, dd, 64); tm.val = __shfl_up(im[c].val, dd, 64);
                if (lane >= dd) { ix[c] += tc; im[c] = ms_combine(tm, im[c]); }
            }
        }
        if (lane == 63) {
#pragma unroll
            for (int c = 0; c < NC; c++) { s_c[w][c] = ix[c]; s_m[w][c] = im[c]; }
        }
        __syncthreads();
        u32 totc[NC]; MinSt totm[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            u32 bc = s_runc[c]; MinSt bm = s_runm[c];
            totc[c] = s_runc[c]; totm[c] = s_runm[c];
            for (int k = 0; k < NT / 64; k++) {
                if (k < w) { bc += s_c[k][c]; bm = ms_combine(bm, s_m[k][c]); }
                totc[c] += s_c[k][c]; totm[c] = ms_combine(totm[c], s_m[k][c]);
            }
            u32 xc = __shfl_up(ix[c], 1, 64);
            MinSt xm; xm.has = __shfl_up(im[c].has, 1, 64); xm.val = __shfl_up(im[c].val, 1, 64);
            if (lane == 0) { xc = 0; xm.has = 0; xm.val = INF; }
            u32 rc = bc + xc; MinSt rm = ms_combine(bm, xm);          // exclusive prefix in front of my first tile
            __syncthreads();                                           // (the staging arrays are free again)
#pragma unroll
            for (int k = 0; k < CARRY_PER; k++) {
                s_x[threadIdx.x * CARRY_PER + k] = rc; s_y[threadIdx.x * CARRY_PER + k] = rm.val;
                rc += vc[c][k]; rm = ms_combine(rm, vm[c][k]);
            }
            __syncthreads();
            for (int k = threadIdx.x; k < PASS; k += NT) {
                const int64_t t = base + k;
                if (t < t_hi) { a.tile_G[(size_t)(c0 + c) * a.ntiles + t] = s_x[k]; a.tile_carry[(size_t)(c0 + c) * a.ntiles + t] = s_y[k]; }
            }
        }
        (void)t0;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int c = 0; c < NC; c++) { s_runc[c] = totc[c]; s_runm[c] = totm[c]; }
        }
        __syncthreads();
    }
}

// one workgroup per class: the three scans are independent, and one workgroup doing all of them was bound by instruction
// issue on its one CU (~4600 VALU instructions per wave, four waves per SIMD: 23 us for 4900 tiles)
template <int NT>
__global__ __launch_bounds__(NT) void k_tile_carry(RvSplitArgs a) {
    __shared__ u32   s_c[NT / 64][3];
    __shared__ MinSt s_m[NT / 64][3];
    __shared__ u32   s_runc[3];
    __shared__ MinSt s_runm[3];
    __shared__ u32   s_x[NT * CARRY_PER], s_y[NT * CARRY_PER];
    const int c0 = blockIdx.x;
    if (threadIdx.x < 3) { s_runc[threadIdx.x] = 0u; s_runm[threadIdx.x].has = 0; s_runm[threadIdx.x].val = INF; }
    __syncthreads();
    carry_scan_range<NT, 1>(a, c0, 0, a.ntiles, s_runc, s_runm, s_c, s_m, s_x, s_y);
    if (threadIdx.x == 0) {
        a.total[c0] = s_runc[0];
        if (s_runc[0] != a.expect_total[c0]) atomicOr(a.err, 1u);     // the intervals do not cover what they claim
    }
}

// default chunk; RV_CARRY_CH overrides it (tests force the chunked path on small inputs).  One pass of a CARRY_NT-thread workgroup:
// 16 KB of LDS.  (It was 8192 tiles in a 1024-thread workgroup with 64 KB: next to a leaf launch, whose workgroups fill the LDS
// of every CU and are replaced one by one as they finish, such a workgroup waited for the whole launch -- 1 ms at the deep
// levels of 2 x 250 Mbp, 35 us without the neighbour.)
constexpr int CARRY_NT = 256;
constexpr int CARRY_CH = CARRY_NT * CARRY_PER;
__global__ __launch_bounds__(TB) void k_carry_reduce(RvSplitArgs a, int ch, u32 *__restrict__ ch_cnt, MinSt *__restrict__ ch_ms) {
    __shared__ u32   s_c[TB / 64][3];
    __shared__ MinSt s_m[TB / 64][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t lo = (int64_t)blockIdx.x * ch, hi = lo + ch < a.ntiles ? lo + ch : a.ntiles;
    u32 rc[3] = {0, 0, 0}; MinSt rm[3] = {{0, INF}, {0, INF}, {0, INF}};
    // thread t takes a contiguous run of CARRY_CH / TB tiles (order matters for the min-state)
    const int per = (ch + TB - 1) / TB;
    for (int k = 0; k < per; k++) {
        const int64_t t = lo + (int64_t)threadIdx.x * per + k;
        if (t < hi) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                MinSt x; x.has = a.tile_has[(size_t)c * a.ntiles + t]; x.val = a.tile_post[(size_t)c * a.ntiles + t];
                rc[c] += a.tile_cnt[(size_t)c * a.ntiles + t]; rm[c] = ms_combine(rm[c], x);
            }
        }
    }
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const u32 tc = __shfl_up(rc[c], dd, 64);
            MinSt tm; tm.has = __shfl_up(rm[c].has, dd, 64); tm.val = __shfl_up(rm[c].val, dd, 64);
            if (lane >= dd) { rc[c] += tc; rm[c] = ms_combine(tm, rm[c]); }
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < 3; c++) { s_c[w][c] = rc[c]; s_m[w][c] = rm[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        u32 tc = 0; MinSt tm = {0, INF};
        for (int k = 0; k < TB / 64; k++) { tc += s_c[k][c]; tm = ms_combine(tm, s_m[k][c]); }
        ch_cnt[(size_t)blockIdx.x * 3 + c] = tc; ch_ms[(size_t)blockIdx.x * 3 + c] = tm;
    }
}
// exclusive scan of the chunk triples (one wave per class is plenty: a few hundred chunks at most)
__global__ __launch_bounds__(64) void k_carry_chunks(RvSplitArgs a, int nch, u32 *__restrict__ ch_cnt, MinSt *__restrict__ ch_ms) {
    const int c = threadIdx.x;
    if (c >= 3) return;
    u32 rc = 0; MinSt rm = {0, INF};
    for (int k = 0; k < nch; k++) {
        const u32 x = ch_cnt[(size_t)k * 3 + c]; const MinSt y = ch_ms[(size_t)k * 3 + c];
        ch_cnt[(size_t)k * 3 + c] = rc; ch_ms[(size_t)k * 3 + c] = rm;
        rc += x; rm = ms_combine(rm, y);
    }
    a.total[c] = rc;
    if (rc != a.expect_total[c]) atomicOr(a.err, 1u);
}
__global__ __launch_bounds__(CARRY_NT) void k_carry_apply(RvSplitArgs a, int ch, const u32 *__restrict__ ch_cnt, const MinSt *__restrict__ ch_ms) {
    __shared__ u32   s_c[CARRY_NT / 64][3];
    __shared__ MinSt s_m[CARRY_NT / 64][3];
    __shared__ u32   s_runc[3];
    __shared__ MinSt s_runm[3];
    __shared__ u32   s_x[CARRY_NT * CARRY_PER], s_y[CARRY_NT * CARRY_PER];
    const int64_t t_lo = (int64_t)blockIdx.x * ch, t_hi = t_lo + ch < a.ntiles ? t_lo + ch : a.ntiles;
    if (threadIdx.x < 3) { s_runc[threadIdx.x] = ch_cnt[(size_t)blockIdx.x * 3 + threadIdx.x]; s_runm[threadIdx.x] = ch_ms[(size_t)blockIdx.x * 3 + threadIdx.x]; }
    __syncthreads();
    carry_scan_range<CARRY_NT, 3>(a, 0, t_lo, t_hi, s_runc, s_runm, s_c, s_m, s_x, s_y);
}

__global__ __launch_bounds__(TB) void k_lower(uint8_t *__restrict__ T, const sa_t *__restrict__ mbegin, const sa_t *__restrict__ mend,
                                              const int64_t *__restrict__ mpre, int nmatch, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (id >= total) return;
    const int e = upper_idx<int64_t>(mpre, nmatch, id);
    const int64_t pos = (int64_t)mbegin[e] + (id - mpre[e]);
    if (pos < (int64_t)mend[e]) {
        const uint8_t c = T[pos];
        if (c >= 'A' && c <= 'Z') T[pos] = c + 32;
    }
}

// the same, one wave per matched range (device-built tables have no prefix sums over the range lengths)
__global__ __launch_bounds__(TB) void k_lower_ranges(uint8_t *__restrict__ T, const sa_t *__restrict__ mbegin, const sa_t *__restrict__ mend, int nranges) {
    const int e = (int)(((int64_t)blockIdx.x * TB + threadIdx.x) >> 6);
    if (e >= nranges) return;
    const int64_t lo = (int64_t)mbegin[e], hi = (int64_t)mend[e];
    for (int64_t pos = lo + (threadIdx.x & 63); pos < hi; pos += 64) {
        const uint8_t c = T[pos];
        if (c >= 'A' && c <= 'Z') T[pos] = c + 32;
    }
}

// ---- bubble_sort ---------------------------------------------------------------
// Round r handles the r-th matched interval of every split sub-index.
// Window pass: for every text position p in [wlo, B) of a descriptor, look at
// its rank e = SAi[p] in the leading child and keep it if it could act in this
// round (superset of the two `if`s at reveal.c:686 / :714 on the values at the
// start of the round; values of not-yet-visited ranks only decrease during the
// round, so nothing is missed).
__global__ __launch_bounds__(TB) void k_bubble_window(RvBubbleArgs b, int first, int count, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (id >= total) return;
    const int dd = first + upper_idx<int64_t>(b.woff + first, count, id + b.woff[first]);
    const RvBubbleDesc ds = b.desc[dd];
    const int64_t p = ds.wlo + (id + b.woff[first] - b.woff[dd]);
    const int64_t e = (int64_t)b.SAi[p];
    if (e < 0 || e >= ds.n) return;
    const int64_t i = ds.off + e;
    const int64_t sa = (int64_t)b.SA[i];
    if (sa != p) return;
    const int64_t lc = (int64_t)(u32)b.LCP[i];
    const int64_t ln = (e + 1 < ds.n) ? (int64_t)(u32)b.LCP[i + 1] : 0;
    if (sa < ds.B && (sa + lc > ds.B || sa + ln > ds.B)) {
        const u32 slot = atomicAdd(&b.cnt[dd], 1u);
        b.list[b.woff[dd] + slot] = (u32)e;
        b.flag[i] = 1;
    }
}

constexpr int BB_CAP = 4096;

// One visit of the reference's inner loop body (reveal.c:686-721) for rank e of
// the child, executed by the whole workgroup: thread 0 evaluates the two
// conditions on the current values; a move (first branch) walks down from e in
// chunks, looking for its destination x (largest r <= e with r == 0 or
// LCP[r] < t) and shifting [x, e-1] up by one as it goes.
//
// The child's cut windows (for the SAi upkeep of shifted suffixes) are staged
// in LDS once per workgroup: reading them from global memory per shifted rank
// put a dependent load chain on the critical path (measured: 2 ms for ten
// moves in a 5 M-rank child).
constexpr int BB_MAXCUT = 32;
struct CutWin { sa_t lo[BB_MAXCUT], hi[BB_MAXCUT]; int n; };

__device__ inline void sai_upkeep(const RvBubbleArgs &b, const RvBubbleDesc &ds, const CutWin &cw, sa_t pos, int64_t rank) {
    for (int q = 0; q < cw.n; q++)
        if (pos >= cw.lo[q] && pos < cw.hi[q]) { b.SAi[pos] = (sa_t)rank; return; }
    for (int q = ds.cut0 + BB_MAXCUT; q < ds.cut1; q++)           // more windows than fit in LDS (many samples)
        if (pos >= b.cut_lo[q] && pos < b.cut_hi[q]) { b.SAi[pos] = (sa_t)rank; return; }
}

// 16-byte accesses throughout (measured on MI355X: one workgroup shifts
// ~2.7 G ranks/s with dword accesses but streams ~100 GB/s with dwordx4).
// Ranks are handled in aligned groups of four: the search reads LCP[4g..4g+3],
// the shift loads the sources [4g-1..4g+2] (one unaligned dwordx4 per array,
// a dword of bytes for BWT) and stores the group aligned.  A chunk is NT*EG
// groups, top-down; all of a chunk's sources are loaded before its first store.
// (Returns false; the bool is what is left of an earlier version that could hand long moves to grid-wide kernels.)
template <int NT, int EG>
__device__ inline bool bubble_visit_vec(const RvBubbleArgs &b, const RvBubbleDesc &ds, const CutWin &cw, sa_t *SA, lcp_t *LCP, uint8_t *BW, int64_t e,
                                        int64_t *s_v, int *s_max, RvBubbleState *st) {
#ifdef RV_SA64
    typedef longlong4 sa4_t;
#else
    typedef int4 sa4_t;
#endif
    const int64_t n = ds.n, B = ds.B;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // every thread evaluates the two conditions itself (same addresses: one transaction per wave); handing thread 0's result
    // round through LDS cost a barrier per visit.  The arrays are quiescent here: every visit ends with a barrier.
    const int64_t sa_e = (int64_t)SA[e], lc_e = (int64_t)(u32)LCP[e];
    const uint8_t bw_e = BW[e];
    const bool moves = sa_e < B && sa_e + lc_e > B;
    if (!moves && e < n - 1 && threadIdx.x == 0) {
        const int64_t ln = (int64_t)(u32)LCP[e + 1];
        if (sa_e < B && sa_e + ln > B && ln > lc_e) LCP[e + 1] = (lcp_t)(B - sa_e);      // reveal.c:714-718
    }
    if (moves) {                                                                     // reveal.c:686-709
        const int64_t tS = sa_e, tL = lc_e, t = B - tS;
        const uint8_t tB = bw_e;
        // the arrays of a child start at an arbitrary rank of the level arrays: align groups on absolute addresses
        const int64_t skew = (int64_t)((reinterpret_cast<uintptr_t>(LCP) >> 2) & 3);     // LCP + (4g - skew) is 16-byte aligned
        int64_t x = 0;
        int64_t gtop = (e + skew) >> 2;                  // group of rank r: (r + skew) >> 2, ranks 4g-skew .. 4g-skew+3
        bool done = false;
        int step = 0;
        while (!done) {
            if (b.dbg && threadIdx.x == 0) atomicAdd(&b.dbg[1], 1ull);
            const int64_t gl = gtop - (int64_t)NT * EG + 1 > 0 ? gtop - (int64_t)NT * EG + 1 : 0;
            sa4_t vs[EG]; int4 vl[EG]; u32 vb[EG];
            int best = -1;                               // largest rank offset (r - rbase) in this chunk with LCP[r] < t
            const int64_t rbase = 4 * gl - skew;         // lowest rank covered by the chunk (may be < 1)
#pragma unroll
            for (int k = 0; k < EG; k++) {
                const int64_t g = gtop - (int64_t)k * NT - threadIdx.x;
                if (g >= gl) {
                    const int64_t r0 = 4 * g - skew;     // ranks r0..r0+3
                    if (r0 >= 1 && r0 + 3 <= e) {
#ifdef RV_SA64
                        const int4 here = *reinterpret_cast<const int4 *>(LCP + r0);
                        __builtin_memcpy(&vs[k], SA + r0 - 1, sizeof(sa4_t));
                        __builtin_memcpy(&vl[k], LCP + r0 - 1, 16);
                        __builtin_memcpy(&vb[k], BW + r0 - 1, 4);
#else
                        // aligned loads only: this group's ranks r0..r0+3; the source of rank r0 (rank r0-1) is the last
                        // element of group g-1, which the next lane holds (groups run downwards with the lane index)
                        const sa4_t cs = *reinterpret_cast<const sa4_t *>(SA + r0);
                        const int4 here = *reinterpret_cast<const int4 *>(LCP + r0);
                        const u32 cb = *reinterpret_cast<const u32 *>(BW + r0);
                        const bool nb = lane < 63 && g - 1 >= gl && r0 - 4 >= 1;      // neighbour lane holds a full group g-1
                        sa_t ps3 = (sa_t)__shfl_down((int)cs.w, 1, 64);
                        int pl3 = __shfl_down(here.w, 1, 64);
                        u32 pb3 = __shfl_down(cb, 1, 64) >> 24;
                        if (!nb) { ps3 = SA[r0 - 1]; pl3 = (int)LCP[r0 - 1]; pb3 = BW[r0 - 1]; }
                        vs[k].x = ps3; vs[k].y = cs.x; vs[k].z = cs.y; vs[k].w = cs.z;
                        vl[k].x = pl3; vl[k].y = here.x; vl[k].z = here.y; vl[k].w = here.z;
                        vb[k] = (cb << 8) | (pb3 & 0xffu);
#endif
                        const u32 hv[4] = {(u32)here.x, (u32)here.y, (u32)here.z, (u32)here.w};
#pragma unroll
                        for (int j = 3; j >= 0; j--) if ((int64_t)hv[j] < t) { const int o = (int)(r0 + j - rbase); best = o > best ? o : best; break; }
                    } else {                             // ragged group at either end: element-wise
                        sa_t *ps = reinterpret_cast<sa_t *>(&vs[k]); int *pl = reinterpret_cast<int *>(&vl[k]); uint8_t *pb = reinterpret_cast<uint8_t *>(&vb[k]);
                        for (int j = 0; j < 4; j++) {
                            const int64_t r = r0 + j;
                            if (r >= 1 && r <= e) {
                                ps[j] = SA[r - 1]; pl[j] = (int)LCP[r - 1]; pb[j] = BW[r - 1];
                                if ((int64_t)(u32)LCP[r] < t) { const int o = (int)(r - rbase); best = o > best ? o : best; }
                            }
                        }
                    }
                }
            }
            for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_down(best, d, 64); best = o > best ? o : best; }
            // One barrier per chunk (all sources loaded, stop rank known).  The stores below need no barrier of their own:
            // the next chunk reads and writes lower ranks only, so its loads are issued while these stores drain.  The
            // per-wave maxima alternate between two buffers so that a fast wave's next write cannot overtake a slow wave's read.
            int *smx = s_max + (step & 1) * (NT / 64);
            step++;
            if (lane == 0) smx[w] = best;
            __syncthreads();
            int mx = -1;
            for (int k = 0; k < NT / 64; k++) mx = smx[k] > mx ? smx[k] : mx;
            // ranks (stop, min(e, chunk top)] move up by one; rank 0 is a stop by definition
            int64_t stop = (mx >= 0) ? rbase + mx : rbase - 1;
            if (stop < 0) stop = 0;
            if (mx >= 0 || rbase <= 1) done = true;
            if (done && mx < 0) stop = 0;
#pragma unroll
            for (int k = 0; k < EG; k++) {
                const int64_t g = gtop - (int64_t)k * NT - threadIdx.x;
                if (g >= gl) {
                    const int64_t r0 = 4 * g - skew;
                    if (r0 > stop && r0 + 3 <= e) {
                        *reinterpret_cast<sa4_t *>(SA + r0) = vs[k];
                        *reinterpret_cast<int4 *>(LCP + r0) = vl[k];
                        *reinterpret_cast<u32 *>(BW + r0) = vb[k];
                        const sa_t *ps = reinterpret_cast<const sa_t *>(&vs[k]);
                        for (int j = 0; j < 4; j++) sai_upkeep(b, ds, cw, ps[j], r0 + j);
                    } else {
                        const sa_t *ps = reinterpret_cast<const sa_t *>(&vs[k]); const int *pl = reinterpret_cast<const int *>(&vl[k]); const uint8_t *pb = reinterpret_cast<const uint8_t *>(&vb[k]);
                        for (int j = 0; j < 4; j++) {
                            const int64_t r = r0 + j;
                            if (r > stop && r >= 1 && r <= e) {
                                SA[r] = ps[j]; LCP[r] = (lcp_t)pl[j]; BW[r] = pb[j];
                                sai_upkeep(b, ds, cw, ps[j], r);
                            }
                        }
                    }
                }
            }
            if (done) x = stop;
            gtop = gl - 1;
        }
        __threadfence_block();
        __syncthreads();      // (the last chunk's stores to x+1 come before the values written below)
        if (threadIdx.x == 0) {
            SA[x] = (sa_t)tS;
            BW[x] = tB;
            b.SAi[tS] = (sa_t)x;
            if (x + 1 < n) LCP[x + 1] = (lcp_t)t;
            if (e < n - 1 && tL < (int64_t)(u32)LCP[e + 1]) LCP[e + 1] = (lcp_t)tL;
        }
    }
    __threadfence_block();
    __syncthreads();
    return false;
}

// ---- disjoint moves in parallel ---------------------------------------------------
// A visit of rank e reads and writes only ranks [x, e+1] (x = e for a visit that
// does not move), and x itself is found by reading LCP inside that range.  If the
// ranges of a run of consecutive actives are pairwise disjoint *on the current
// arrays*, every visit of the run sees exactly the values it would see in the
// reference's sequential order, so the run can be executed concurrently, one
// thread per active (these moves are short).  A move whose destination is more
// than BB_SCAN ranks away, or whose range touches its predecessor's, ends the run
// and goes through the whole-workgroup visit.
constexpr int BB_SCAN = 32;
template <int CAP> struct ParScratchT { u32 lo[CAP]; uint8_t kind[CAP]; };     // per active: first rank of its range, 0 none / 1 move / 2 truncate / 3 long
typedef ParScratchT<BB_CAP> ParScratch;

// classify actives [from, cnt) on the current arrays
template <int NT, class PS>
__device__ inline void par_classify(const RvBubbleDesc &ds, const sa_t *SA, const lcp_t *LCP, const u32 *lst, u32 from, u32 cnt, PS &ps) {
    const int64_t n = ds.n, B = ds.B;
    for (u32 a = from + threadIdx.x; a < cnt; a += NT) {
        const int64_t e = lst[a];
        const int64_t sa = (int64_t)SA[e], lc = (int64_t)(u32)LCP[e];
        int kind = 0; int64_t lo = e;
        if (sa < B && sa + lc > B) {
            const int64_t t = B - sa;
            kind = 3;
            int64_t r = e;                       // LCP[e] >= t by the condition above
            for (int step = 0; step < BB_SCAN; step++) {
                r--;
                if (r <= 0) { r = 0; kind = 1; break; }
                if ((int64_t)(u32)LCP[r] < t) { kind = 1; break; }
            }
            lo = (kind == 1) ? r : e - BB_SCAN;
            if (e == 0) { kind = 1; lo = 0; }
        } else if (e < n - 1) {
            const int64_t ln = (int64_t)(u32)LCP[e + 1];
            if (sa < B && sa + ln > B && ln > lc) kind = 2;
        }
        ps.lo[a] = (u32)(lo < 0 ? 0 : lo); ps.kind[a] = (uint8_t)kind;
    }
}

// execute actives [from, to) concurrently (their ranges are disjoint, none is long): reveal.c:686-721 per thread
template <int NT, class PS>
__device__ inline void par_execute(const RvBubbleArgs &b, const RvBubbleDesc &ds, const CutWin &cw, sa_t *SA, lcp_t *LCP, uint8_t *BW,
                                   const u32 *lst, u32 from, u32 to, const PS &ps) {
    const int64_t n = ds.n, B = ds.B;
    for (u32 a = from + threadIdx.x; a < to; a += NT) {
        const int64_t e = lst[a];
        const int kind = ps.kind[a];
        if (kind == 2) {
            LCP[e + 1] = (lcp_t)(B - (int64_t)SA[e]);                                     // reveal.c:714-718
        } else if (kind == 1) {
            const int64_t x = ps.lo[a];
            const sa_t tS = SA[e]; const lcp_t tL = LCP[e]; const uint8_t tB = BW[e];
            for (int64_t r = e; r > x; r--) {                                             // reveal.c:691-698
                const sa_t p = SA[r - 1];
                SA[r] = p; LCP[r] = LCP[r - 1]; BW[r] = BW[r - 1];
                sai_upkeep(b, ds, cw, p, r);
            }
            SA[x] = tS; BW[x] = tB;
            b.SAi[tS] = (sa_t)x;
            if (x + 1 < n) LCP[x + 1] = (lcp_t)(B - (int64_t)tS);
            if (e < n - 1 && (int64_t)(u32)tL < (int64_t)(u32)LCP[e + 1]) LCP[e + 1] = tL;
        }
    }
}

// Few actives: one wave per active.  The 64 lanes search the destination 64 ranks
// at a time (up to BB_WSCAN ranks) and, if the ranges of the batch are disjoint,
// shift their own range 64 ranks per step -- no workgroup barrier inside a move.
constexpr int BB_WSCAN = 512;

// classify active a (one wave): kind 0 none / 1 move to lo / 2 truncate / 3 long
__device__ inline void wave_classify(const RvBubbleDesc &ds, const sa_t *SA, const lcp_t *LCP, int64_t e, u32 *lo_out, uint8_t *kind_out) {
    const int lane = threadIdx.x & 63;
    const int64_t n = ds.n, B = ds.B;
    const int64_t sa = (int64_t)SA[e], lc = (int64_t)(u32)LCP[e];
    int kind = 0; int64_t lo = e;
    if (sa < B && sa + lc > B) {
        const int64_t t = B - sa;
        kind = 3; lo = e > BB_WSCAN ? e - BB_WSCAN : 0;
        // all BB_WSCAN ranks below e at once (independent loads: one memory round trip; a loop of 64-rank steps that stopped at the
        // first hit cost one round trip per step, eight for every mover that turns out to be long), then the nearest hit
        lcp_t v[BB_WSCAN / 64];
#pragma unroll
        for (int k = 0; k < BB_WSCAN / 64; k++) {
            const int64_t r = e - 1 - 64 * k - lane;
            v[k] = r > 0 ? LCP[r] : (lcp_t)0;
        }
#pragma unroll
        for (int k = 0; k < BB_WSCAN / 64; k++) {
            const int64_t r = e - 1 - 64 * k - lane;
            const bool hit = r >= 0 && (r == 0 || (int64_t)(u32)v[k] < t);
            const u64 bal = __ballot(hit);
            if (bal && kind == 3) { kind = 1; lo = e - 1 - 64 * k - (int64_t)__builtin_ctzll(bal); }     // lowest lane = largest rank
        }
        if (e == 0) { kind = 1; lo = 0; }
    } else if (e < n - 1) {
        const int64_t ln = (int64_t)(u32)LCP[e + 1];
        if (sa < B && sa + ln > B && ln > lc) kind = 2;
    }
    if (lane == 0) { *lo_out = (u32)lo; *kind_out = (uint8_t)kind; }
}

// execute active a (one wave), kind 1 or 2
__device__ inline void wave_execute(const RvBubbleArgs &b, const RvBubbleDesc &ds, const CutWin &cw, sa_t *SA, lcp_t *LCP, uint8_t *BW,
                                    int64_t e, int kind, int64_t x) {
    const int lane = threadIdx.x & 63;
    const int64_t n = ds.n, B = ds.B;
    if (kind == 2) {
        if (lane == 0) LCP[e + 1] = (lcp_t)(B - (int64_t)SA[e]);
        return;
    }
    if (kind != 1) return;
    const sa_t tS = SA[e]; const lcp_t tL = LCP[e]; const uint8_t tB = BW[e];
    for (int64_t top = e; top > x; top -= 64) {                 // destinations top, top-1, ..., down to x+1
        const int64_t r = top - lane;
        sa_t vs = 0; lcp_t vl = 0; uint8_t vb = 0;
        if (r > x) { vs = SA[r - 1]; vl = LCP[r - 1]; vb = BW[r - 1]; }
        __builtin_amdgcn_wave_barrier();                          // every lane's loads are issued before any store below
        if (r > x) { SA[r] = vs; LCP[r] = vl; BW[r] = vb; sai_upkeep(b, ds, cw, vs, r); }
    }
    if (lane == 0) {
        SA[x] = tS; BW[x] = tB;
        b.SAi[tS] = (sa_t)x;
        if (x + 1 < n) LCP[x + 1] = (lcp_t)(B - (int64_t)tS);
        if (e < n - 1 && (int64_t)(u32)tL < (int64_t)(u32)LCP[e + 1]) LCP[e + 1] = tL;
    }
}

// Visit lst[start .. cnt) in the reference's order.  Returns cnt.
template <int NT, int EL, class PS>
__device__ inline u32 visit_list(const RvBubbleArgs &b, const RvBubbleDesc &ds, const CutWin &cw, sa_t *SA, lcp_t *LCP, uint8_t *BW,
                                 const u32 *lst, u32 start, u32 cnt, PS &ps, int64_t *s_v, int *s_max, u32 *s_first, RvBubbleState *st) {
    u32 cur = start;
    // A handful of actives: classifying them first only adds latency (measured on C2: 12 ms sequential vs 14.5 ms) ->
    // plain sequential visits.  The concurrent path pays off with many actives per cut (closely related samples).
    if (cnt - start <= 32) {
        // a handful of actives: one wave per active, NT/64 at a time
        constexpr int NW = NT / 64;
        const int w = threadIdx.x >> 6;
        while (cur < cnt) {
            const u32 end = cnt - cur > (u32)NW ? cur + NW : cnt;
            if (cur + w < end) wave_classify(ds, SA, LCP, (int64_t)lst[cur + w], &ps.lo[cur + w], &ps.kind[cur + w]);
            if (threadIdx.x == 0) *s_first = end;
            __syncthreads();
            {
                const u32 a = cur + threadIdx.x;
                if (a < end && (ps.kind[a] == 3 || (a > cur && (int64_t)ps.lo[a] <= (int64_t)lst[a - 1] + 1))) atomicMin(s_first, a);
            }
            __syncthreads();
            const u32 f = *s_first;
            if (cur + w < f) wave_execute(b, ds, cw, SA, LCP, BW, (int64_t)lst[cur + w], ps.kind[cur + w], (int64_t)ps.lo[cur + w]);
            __threadfence_block();
            __syncthreads();
            if (f >= end) { cur = end; continue; }
            if (b.dbg && threadIdx.x == 0) atomicAdd(&b.dbg[0], 1ull);
            if (bubble_visit_vec<NT, EL>(b, ds, cw, SA, LCP, BW, (int64_t)lst[f], s_v, s_max, st)) return f;
            cur = f + 1;
        }
        return cnt;
    }
    while (cur < cnt) {
        const u32 end = cnt - cur > (u32)NT ? cur + NT : cnt;       // look one active per thread ahead
        par_classify<NT, PS>(ds, SA, LCP, lst, cur, end, ps);
        if (threadIdx.x == 0) *s_first = end;
        __syncthreads();
        // first active that cannot join the run: long, or its range touches the previous one
        {
            const u32 a = cur + threadIdx.x;
            if (a < end && (ps.kind[a] == 3 || (a > cur && (int64_t)ps.lo[a] <= (int64_t)lst[a - 1] + 1))) atomicMin(s_first, a);
        }
        __syncthreads();
        const u32 f = *s_first;
        par_execute<NT, PS>(b, ds, cw, SA, LCP, BW, lst, cur, f, ps);
        __threadfence_block();
        __syncthreads();
        if (f >= end) { cur = end; continue; }
        if (b.dbg && threadIdx.x == 0) { atomicAdd(&b.dbg[0], 1ull); atomicAdd(&b.dbg[2], (unsigned long long)(f - cur)); }
        if (bubble_visit_vec<NT, EL>(b, ds, cw, SA, LCP, BW, (int64_t)lst[f], s_v, s_max, st)) return f;
        cur = f + 1;
    }
    return cnt;
}

// One workgroup per (leading child, cut).  Visits the active ranks in
// ascending order (the reference's `for i` order; ranks not yet visited never
// move).  Few actives: sort the window pass' list in LDS.  Many (closely
// related samples share long matches across a cut): walk the child's flag
// bytes 4096 ranks at a time, which yields them already ordered.
// NT = 256 for ordinary children, 1024 (8 ranks per thread and step) for the
// few large ones, whose moves travel up to a quarter of the child.
template <int NT, int EL>
__global__ __launch_bounds__(NT) void k_bubble_apply(RvBubbleArgs b, int first) {
    __shared__ u32 lst[BB_CAP];
    __shared__ int64_t s_v[4];      // decision broadcast: kind, tS, tL
    __shared__ int s_max[2 * (NT / 64)];
    __shared__ u32 s_w[NT / 64];
    __shared__ CutWin cw;
    __shared__ ParScratch ps;
    __shared__ u32 s_first;
    const int dd = first + blockIdx.x;
    const u32 cnt = b.cnt[dd];
    if (cnt == 0) return;
    const RvBubbleDesc ds = b.desc[dd];
    {
        const int nc = ds.cut1 - ds.cut0 < BB_MAXCUT ? ds.cut1 - ds.cut0 : BB_MAXCUT;
        if ((int)threadIdx.x < nc) { cw.lo[threadIdx.x] = b.cut_lo[ds.cut0 + threadIdx.x]; cw.hi[threadIdx.x] = b.cut_hi[ds.cut0 + threadIdx.x]; }
        if (threadIdx.x == 0) cw.n = nc;
    }
    sa_t  *SA = b.SA + ds.off;
    lcp_t *LCP = b.LCP + ds.off;
    uint8_t *BW = b.BWT + ds.off;
    uint8_t *flag = b.flag + ds.off;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (cnt <= BB_CAP) {
        RvBubbleState *st = b.state + dd;
        const int32_t start = st->next;
        if (start >= (int32_t)cnt) return;          // done by the data-parallel round (k_pb_movers sets next past the end)
        u32 np2 = 1; while (np2 < cnt) np2 <<= 1;
        for (u32 k = threadIdx.x; k < np2; k += NT) {
            const u32 v = k < cnt ? b.list[b.woff[dd] + k] : 0xFFFFFFFFu;
            lst[k] = v;
            if (k < cnt) flag[v] = 0;
        }
        __syncthreads();
        {
            for (u32 size = 2; size <= np2; size <<= 1)
                for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
                    for (u32 k = threadIdx.x; k < np2 / 2; k += NT) {
                        const u32 lo = (k / stride) * stride * 2 + (k % stride), hi = lo + stride;
                        const bool up = ((lo & size) == 0);
                        const u32 x = lst[lo], y = lst[hi];
                        if ((x > y) == up) { lst[lo] = y; lst[hi] = x; }
                    }
                    __syncthreads();
                }
        }
        u32 ai = (u32)start;
        const u32 stopped = visit_list<NT, EL, ParScratch>(b, ds, cw, SA, LCP, BW, lst, ai, cnt, ps, s_v, s_max, &s_first, st);
        if (threadIdx.x == 0) st->next = (int32_t)stopped;
        return;
    }
    constexpr int FL = BB_CAP / NT;     // flag bytes per thread and chunk
    for (int64_t base = 0; base < ds.n; base += BB_CAP) {
        const int64_t r0 = base + (int64_t)threadIdx.x * FL;
        u32 bits = 0;
        for (int k = 0; k < FL; k++) if (r0 + k < ds.n && flag[r0 + k]) { bits |= 1u << k; flag[r0 + k] = 0; }
        const u32 mine = __popc(bits);
        u32 inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        u32 before = 0, tot = 0;
        for (int k = 0; k < NT / 64; k++) { const u32 c = s_w[k]; if (k < w) before += c; tot += c; }
        u32 q = before + inc - mine;
        for (int k = 0; k < FL; k++) if (bits & (1u << k)) lst[q++] = (u32)(r0 + k);
        __syncthreads();
        (void)visit_list<NT, EL, ParScratch>(b, ds, cw, SA, LCP, BW, lst, 0, tot, ps, s_v, s_max, &s_first, nullptr);
        __syncthreads();
    }
    if (threadIdx.x == 0) b.state[dd].next = 0x7fffffff;
}

// ---- all cuts of one leading child in one workgroup ---------------------------------------
// For children up to RV_BUBBLE_HUGE_N ranks.  The workgroup walks the child's cuts in
// graphalign's order; per cut it runs the window pass itself (the window is at most
// max-LCP positions), sorts the actives and visits them.  Children advance through
// their cuts independently of each other, instead of every round waiting for the
// slowest child of the level.
template <int NT, int EL>
__global__ __launch_bounds__(NT) void k_bubble_child(RvBubbleArgs b, const RvBubbleDesc *__restrict__ cdesc, int64_t max_n, int64_t min_n = 0) {
    __shared__ u32 lst[BB_CAP];
    __shared__ int64_t s_v[4];
    __shared__ int s_max[2 * (NT / 64)];
    __shared__ u32 s_w[NT / 64];
    __shared__ CutWin cw;
    __shared__ ParScratch ps;
    __shared__ u32 s_first, s_cnt;
    RvBubbleDesc ds = cdesc[blockIdx.x];
    if (ds.n <= min_n || ds.n > max_n) return;      // device-built descriptor arrays hold one entry per sub-index: empty ones, children of another size class, children the rounds take
    {
        const int nc = ds.cut1 - ds.cut0 < BB_MAXCUT ? ds.cut1 - ds.cut0 : BB_MAXCUT;
        if ((int)threadIdx.x < nc) { cw.lo[threadIdx.x] = b.cut_lo[ds.cut0 + threadIdx.x]; cw.hi[threadIdx.x] = b.cut_hi[ds.cut0 + threadIdx.x]; }
        if (threadIdx.x == 0) cw.n = nc;
    }
    sa_t  *SA = b.SA + ds.off;
    lcp_t *LCP = b.LCP + ds.off;
    uint8_t *BW = b.BWT + ds.off;
    uint8_t *flag = b.flag + ds.off;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // RV_LEVEL_LOG: where the slowest child of the launch spends its time (100 MHz clock; maxima over the workgroups)
    const unsigned long long tk0 = b.dbg ? wall_clock64() : 0ull;
    unsigned long long t_find = 0, t_visit = 0, n_act = 0;
    for (int q = ds.cut0; q < ds.cut1; q++) {
        const int64_t B = (int64_t)b.cut_hi[q], wlo = (int64_t)b.cut_lo[q];
        if (wlo >= B) continue;                                   // uniform
        ds.B = B;
        const unsigned long long tc0 = b.dbg ? wall_clock64() : 0ull;
        if (ds.n <= 4 * BB_CAP) {
            // Small child: the ranks that can act are found by reading the child itself, 4096 ranks at a time -- already in
            // visiting order, no SAi gathers over the window, no flags, no sort.  (With many samples a level has tens of
            // thousands of (child, cut) pairs with a few dozen actives each; the window pass and the sort were most of
            // their 150 us.)  A chunk is examined after the chunks below it have been visited: visits only lower values of
            // ranks not yet visited, and the test is the one the reference would evaluate at that moment anyway.
            constexpr int FL = BB_CAP / NT;
            for (int64_t base = 0; base < ds.n; base += BB_CAP) {
                const int64_t r0 = base + (int64_t)threadIdx.x * FL;
                u32 bits = 0;
#pragma unroll
                for (int k = 0; k < FL; k++) {
                    const int64_t r = r0 + k;
                    if (r < ds.n) {
                        const int64_t p = (int64_t)SA[r];
                        const int64_t lc = (int64_t)(u32)LCP[r];
                        const int64_t ln = (r + 1 < ds.n) ? (int64_t)(u32)LCP[r + 1] : 0;
                        if ((p >= wlo) & (p < B) & ((p + lc > B) | (p + ln > B))) bits |= 1u << k;
                    }
                }
                const u32 mine = __popc(bits);
                u32 inc = mine;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
                if (lane == 63) s_w[w] = inc;
                __syncthreads();
                u32 before = 0, tot = 0;
                for (int k = 0; k < NT / 64; k++) { const u32 c = s_w[k]; if (k < w) before += c; tot += c; }
                u32 qq = before + inc - mine;
                for (int k = 0; k < FL; k++) if (bits & (1u << k)) lst[qq++] = (u32)(r0 + k);
                __syncthreads();
                if (tot) {
                    if (b.dbg && threadIdx.x == 0) { atomicAdd(&b.dbg[3], 1ull); atomicAdd(&b.dbg[4], (unsigned long long)tot); }
                    (void)visit_list<NT, EL, ParScratch>(b, ds, cw, SA, LCP, BW, lst, 0, tot, ps, s_v, s_max, &s_first, nullptr);
                }
                __threadfence_block();
                __syncthreads();
            }
            continue;
        }
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        // window pass (same superset test as k_bubble_window)
        for (int64_t p = wlo + threadIdx.x; p < B; p += NT) {
            const int64_t e = (int64_t)b.SAi[p];
            if (e < 0 || e >= ds.n) continue;
            if ((int64_t)SA[e] != p) continue;
            const int64_t lc = (int64_t)(u32)LCP[e];
            const int64_t ln = (e + 1 < ds.n) ? (int64_t)(u32)LCP[e + 1] : 0;
            if (p + lc > B || p + ln > B) {
                const u32 k = atomicAdd(&s_cnt, 1u);
                if (k < BB_CAP) lst[k] = (u32)e;
                flag[e] = 1;
            }
        }
        __syncthreads();
        const u32 cnt = s_cnt;
        if (b.dbg) t_find += wall_clock64() - tc0;
        if (cnt == 0) continue;
        if (b.dbg && threadIdx.x == 0) { atomicAdd(&b.dbg[3], 1ull); atomicAdd(&b.dbg[4], (unsigned long long)cnt); }
        const unsigned long long tc1 = b.dbg ? wall_clock64() : 0ull;
        n_act += cnt;
        // Order of the visits = ascending rank.  Few actives: sort the list.  Many actives in a small child: walking the
        // child's flag bytes yields them already ordered and costs less than ~60 bitonic stages (many samples: hundreds of
        // actives per cut, ten cuts per child).
        if (cnt <= BB_CAP && !(ds.n <= 4 * BB_CAP && cnt > 128)) {
            u32 np2 = 1; while (np2 < cnt) np2 <<= 1;
            for (u32 k = threadIdx.x; k < np2; k += NT) { if (k < cnt) flag[lst[k]] = 0; else lst[k] = 0xFFFFFFFFu; }
            __syncthreads();
            for (u32 size = 2; size <= np2; size <<= 1)
                for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
                    for (u32 k = threadIdx.x; k < np2 / 2; k += NT) {
                        const u32 lo = (k / stride) * stride * 2 + (k % stride), hi = lo + stride;
                        const bool up = ((lo & size) == 0);
                        const u32 x = lst[lo], y = lst[hi];
                        if ((x > y) == up) { lst[lo] = y; lst[hi] = x; }
                    }
                    __syncthreads();
                }
            (void)visit_list<NT, EL, ParScratch>(b, ds, cw, SA, LCP, BW, lst, 0, cnt, ps, s_v, s_max, &s_first, nullptr);
        } else {
            constexpr int FL = BB_CAP / NT;
            for (int64_t base = 0; base < ds.n; base += BB_CAP) {
                const int64_t r0 = base + (int64_t)threadIdx.x * FL;
                u32 bits = 0;
                for (int k = 0; k < FL; k++) if (r0 + k < ds.n && flag[r0 + k]) { bits |= 1u << k; flag[r0 + k] = 0; }
                const u32 mine = __popc(bits);
                u32 inc = mine;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
                if (lane == 63) s_w[w] = inc;
                __syncthreads();
                u32 before = 0, tot = 0;
                for (int k = 0; k < NT / 64; k++) { const u32 c = s_w[k]; if (k < w) before += c; tot += c; }
                u32 qq = before + inc - mine;
                for (int k = 0; k < FL; k++) if (bits & (1u << k)) lst[qq++] = (u32)(r0 + k);
                __syncthreads();
                (void)visit_list<NT, EL, ParScratch>(b, ds, cw, SA, LCP, BW, lst, 0, tot, ps, s_v, s_max, &s_first, nullptr);
                __syncthreads();
            }
        }
        __threadfence_block();
        __syncthreads();
        if (b.dbg) t_visit += wall_clock64() - tc1;
    }
    if (b.dbg && threadIdx.x == 0) {
        atomicMax(&b.dbg[5], ((wall_clock64() - tk0) << 24) | (n_act & 0xFFFFFFull)); atomicMax(&b.dbg[6], t_find); atomicMax(&b.dbg[7], t_visit);
    }
}

// The same, for children that fit into LDS (<= RV_BUBBLE_LDS_N ranks): SA / LCP / BWT are copied in once, every cut of
// the child is replayed on the copies, and they are written back at the end.  The classify / shift loops of a visit are
// chains of dependent loads and stores; in global memory each link costs a memory round trip (~1 us), which made a cut
// of a small child cost ~150 us however little it moved -- with many samples a level has tens of thousands of such cuts.
// The windowed SAi is not needed here (the actives are found by reading the child) and is not kept up.
template <int NT, int EL, int N, int CH>
__global__ __launch_bounds__(NT) void k_bubble_child_lds(RvBubbleArgs b, const RvBubbleDesc *__restrict__ cdesc, int64_t min_n = 0, int64_t max_n = (int64_t)1 << 62) {
    __shared__ __attribute__((aligned(16))) sa_t sSA[N];
    __shared__ __attribute__((aligned(16))) lcp_t sLCP[N];
    __shared__ __attribute__((aligned(16))) uint8_t sBW[N];
    __shared__ u32 lst[CH];
    __shared__ int64_t s_v[4];
    __shared__ int s_max[2 * (NT / 64)];
    __shared__ u32 s_w[NT / 64];
    __shared__ CutWin cw;
    __shared__ ParScratchT<CH> ps;
    __shared__ u32 s_first;
    RvBubbleDesc ds = cdesc[blockIdx.x];
    if (ds.n <= min_n || ds.n > max_n) return;   // (device-built descriptor arrays: one entry per sub-index, every size class looks at all of them)
    const int cut0 = ds.cut0, cut1 = ds.cut1;
    ds.cut0 = ds.cut1 = 0;                       // no SAi upkeep inside the visits
    if (threadIdx.x == 0) cw.n = 0;
    sa_t  *gSA = b.SA + ds.off;
    lcp_t *gLCP = b.LCP + ds.off;
    uint8_t *gBW = b.BWT + ds.off;
    const int n = (int)ds.n;
    for (int r = threadIdx.x; r < n; r += NT) { sSA[r] = gSA[r]; sLCP[r] = gLCP[r]; sBW[r] = gBW[r]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int q = cut0; q < cut1; q++) {
        const int64_t B = (int64_t)b.cut_hi[q], wlo = (int64_t)b.cut_lo[q];
        if (wlo >= B) continue;                                   // uniform
        ds.B = B;
        constexpr int FL = CH / NT;
        static_assert(FL >= 1 && FL <= 32, "one bit per rank of a thread");
        for (int base = 0; base < n; base += CH) {
            const int r0 = base + (int)threadIdx.x * FL;
            u32 bits = 0;
#pragma unroll
            for (int k = 0; k < FL; k++) {
                const int r = r0 + k;
                if (r < n) {
                    const int64_t p = (int64_t)sSA[r];
                    const int64_t lc = (int64_t)(u32)sLCP[r];
                    const int64_t ln = (r + 1 < n) ? (int64_t)(u32)sLCP[r + 1] : 0;
                    if ((p >= wlo) & (p < B) & ((p + lc > B) | (p + ln > B))) bits |= 1u << k;
                }
            }
            const u32 mine = __popc(bits);
            u32 inc = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
            if (lane == 63) s_w[w] = inc;
            __syncthreads();
            u32 before = 0, tot = 0;
            for (int k = 0; k < NT / 64; k++) { const u32 c = s_w[k]; if (k < w) before += c; tot += c; }
            u32 qq = before + inc - mine;
            for (int k = 0; k < FL; k++) if (bits & (1u << k)) lst[qq++] = (u32)(r0 + k);
            __syncthreads();
            if (tot) (void)visit_list<NT, EL, ParScratchT<CH> >(b, ds, cw, sSA, sLCP, sBW, lst, 0, tot, ps, s_v, s_max, &s_first, nullptr);
            __syncthreads();
        }
    }
    for (int r = threadIdx.x; r < n; r += NT) { gSA[r] = sSA[r]; gLCP[r] = sLCP[r]; gBW[r] = sBW[r]; }
}

// SAi[SA[i]] = rank inside the owning sub-index (materialised on demand for the getter)
__global__ __launch_bounds__(TB) void k_sai_level(const sa_t *__restrict__ SA, int64_t m, const int64_t *__restrict__ sub_start, int nsubs, sa_t *__restrict__ SAi) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= m) return;
    const int s = upper_idx<int64_t>(sub_start, nsubs, i);
    SAi[SA[i]] = (sa_t)(i - sub_start[s]);
}

}  // namespace

int rv_split_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, uint8_t *D, const uint8_t *BWT, int64_t m, const RvLabelTabs &t, const RvSplitArgs &a,
                    int nsplit) {
    if (m <= 0 || nsplit <= 0) return 0;
    const unsigned nt = (unsigned)a.ntiles;
    hipLaunchKernelGGL(k_split_count, dim3(nt), dim3(TB), 0, ws.stream, SA, LCP, m, t, a, D);
    RV_LAUNCH_CHECK();
    const int ch = (ws.opt.carry_ch > 0 ? (int)ws.opt.carry_ch : CARRY_CH);
    // up to four passes of one small workgroup per class (one launch instead of three: 2 x 5 Mbp has 24 such levels); the
    // 1024-thread form with its 66 KB of LDS took one pass there, but had to wait for room next to a leaf launch (0.3-0.6 ms)
    if (a.ntiles <= 4 * (int64_t)ch) {
        hipLaunchKernelGGL(k_tile_carry<256>, dim3(3), dim3(256), 0, ws.stream, a);
        RV_LAUNCH_CHECK();
    } else {
        const int nch = (int)ceil_div(a.ntiles, ch);
        DBuf &buf = ws.scan_tmp[3];
        RV_TRY(buf.reserve((size_t)nch * 3 * (sizeof(u32) + sizeof(MinSt)) + 64));
        MinSt *ch_ms = buf.as<MinSt>();
        u32 *ch_cnt = (u32 *)(ch_ms + (size_t)nch * 3);
        hipLaunchKernelGGL(k_carry_reduce, dim3((unsigned)nch), dim3(TB), 0, ws.stream, a, ch, ch_cnt, ch_ms);
        RV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_carry_chunks, dim3(1), dim3(64), 0, ws.stream, a, nch, ch_cnt, ch_ms);
        RV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_carry_apply, dim3((unsigned)nch), dim3(CARRY_NT), 0, ws.stream, a, ch, (const u32 *)ch_cnt, (const MinSt *)ch_ms);
        RV_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_split_emit, dim3(nt), dim3(TB), 0, ws.stream, SA, LCP, (const uint8_t *)D, BWT, m, a);
    RV_LAUNCH_CHECK();
    return 0;
}

// tile -> the sub-index of its first rank (what rv_frontier_commit used to fill on the host: one entry per 2048 ranks, 0.4 ms per level at 2 x 250 Mbp)
__global__ __launch_bounds__(TB) void k_tile_sub(const int64_t *__restrict__ ss, int nsubs, int *__restrict__ tsub, int64_t ntiles) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= ntiles) return;
    const int s = upper_idx<int64_t>(ss, nsubs, t * (int64_t)RV_SPLIT_TILE);
    tsub[t] = s < 0 ? 0 : s;
}
int rv_tile_sub_launch(Workspace &ws, const int64_t *sub_start, int nsubs, int *tile_sub, int64_t ntiles) {
    if (ntiles <= 0) return 0;
    hipLaunchKernelGGL(k_tile_sub, dim3((unsigned)ceil_div(ntiles, TB)), dim3(TB), 0, ws.stream, sub_start, nsubs, tile_sub, ntiles);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_lower_launch(Workspace &ws, uint8_t *T, const sa_t *mbegin, const sa_t *mend, const int64_t *mpre, int nmatch, int64_t total) {
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_lower, dim3((unsigned)ceil_div(total, TB)), dim3(TB), 0, ws.stream, T, mbegin, mend, mpre, nmatch, total);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_bubble_window_launch(Workspace &ws, const RvBubbleArgs &b, int first, int count, int64_t total_window) {
    if (count <= 0 || total_window <= 0) return 0;
    hipLaunchKernelGGL(k_bubble_window, dim3((unsigned)ceil_div(total_window, TB)), dim3(TB), 0, ws.stream, b, first, count, total_window);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_bubble_seq_launch(Workspace &ws, const RvBubbleArgs &b, int first, int count) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL((k_bubble_apply<1024, 4>), dim3((unsigned)count), dim3(1024), 0, ws.stream, b, first);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_lower_ranges_launch(Workspace &ws, uint8_t *T, const sa_t *mbegin, const sa_t *mend, int nranges) {
    if (nranges <= 0) return 0;
    hipLaunchKernelGGL(k_lower_ranges, dim3((unsigned)ceil_div((int64_t)nranges * 64, TB)), dim3(TB), 0, ws.stream, T, mbegin, mend, nranges);
    RV_LAUNCH_CHECK();
    return 0;
}

// one descriptor per sub-index (built on the device): entries with n <= 0 or n > max_n are skipped
int rv_bubble_children_dev_launch(Workspace &ws, const RvBubbleArgs &b, const RvBubbleDesc *d_desc, int count, int64_t max_n) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL((k_bubble_child<1024, 4>), dim3((unsigned)count), dim3(1024), 0, ws.stream, b, d_desc, max_n);
    RV_LAUNCH_CHECK();
    return 0;
}

// one device-built descriptor per sub-index, all size classes: the LDS kernels on ws_lds, the one-workgroup kernels on ws_kid
// (each launch covers every descriptor and leaves those of another class at once); children above max_n are left to the rounds
int rv_bubble_children_dev_classes_launch(Workspace &ws_lds, Workspace &ws_kid, const RvBubbleArgs &b, const RvBubbleDesc *d_desc, int count, int64_t max_n) {
    if (count <= 0) return 0;
    const unsigned g = (unsigned)count;
    hipLaunchKernelGGL((k_bubble_child_lds<128, 1, RV_BUBBLE_LDS_N0, 1024>), dim3(g), dim3(128), 0, ws_lds.stream, b, d_desc, (int64_t)0, (int64_t)RV_BUBBLE_LDS_N0);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_bubble_child_lds<256, 1, RV_BUBBLE_LDS_N1, 1024>), dim3(g), dim3(256), 0, ws_lds.stream, b, d_desc, (int64_t)RV_BUBBLE_LDS_N0, (int64_t)RV_BUBBLE_LDS_N1);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_bubble_child_lds<256, 1, RV_BUBBLE_LDS_N2, 2048>), dim3(g), dim3(256), 0, ws_lds.stream, b, d_desc, (int64_t)RV_BUBBLE_LDS_N1, (int64_t)RV_BUBBLE_LDS_N2);
    RV_LAUNCH_CHECK();
    if (max_n > RV_BUBBLE_LDS_N2) {
        hipLaunchKernelGGL((k_bubble_child<256, 1>), dim3(g), dim3(256), 0, ws_kid.stream, b, d_desc, std::min<int64_t>(max_n, RV_BUBBLE_BIG_N), (int64_t)RV_BUBBLE_LDS_N2);
        RV_LAUNCH_CHECK();
    }
    if (max_n > RV_BUBBLE_BIG_N) {
        hipLaunchKernelGGL((k_bubble_child<1024, 4>), dim3(g), dim3(1024), 0, ws_kid.stream, b, d_desc, max_n, (int64_t)RV_BUBBLE_BIG_N);
        RV_LAUNCH_CHECK();
    }
    return 0;
}

int rv_bubble_children_lds_launch(Workspace &ws, const RvBubbleArgs &b, const RvBubbleDesc *d_lds, const int *count) {
    // three size classes (RV_BUBBLE_LDS_N0/1/2 ranks), descriptors back to back: smaller children take less LDS, so more
    // of them run per CU (28 / 46 / 92 KB per workgroup)
    int first = 0;
    if (count[0] > 0) { hipLaunchKernelGGL((k_bubble_child_lds<128, 1, RV_BUBBLE_LDS_N0, 1024>), dim3((unsigned)count[0]), dim3(128), 0, ws.stream, b, d_lds + first); RV_LAUNCH_CHECK(); }
    first += count[0];
    if (count[1] > 0) { hipLaunchKernelGGL((k_bubble_child_lds<256, 1, RV_BUBBLE_LDS_N1, 1024>), dim3((unsigned)count[1]), dim3(256), 0, ws.stream, b, d_lds + first); RV_LAUNCH_CHECK(); }
    first += count[1];
    if (count[2] > 0) { hipLaunchKernelGGL((k_bubble_child_lds<256, 1, RV_BUBBLE_LDS_N2, 2048>), dim3((unsigned)count[2]), dim3(256), 0, ws.stream, b, d_lds + first); RV_LAUNCH_CHECK(); }
    return 0;
}

int rv_bubble_children_launch(Workspace &ws, const RvBubbleArgs &b, const RvBubbleDesc *d_small, int nsmall, const RvBubbleDesc *d_big, int nbig) {
    if (nsmall > 0) {
        hipLaunchKernelGGL((k_bubble_child<256, 1>), dim3((unsigned)nsmall), dim3(256), 0, ws.stream, b, d_small, (int64_t)1 << 62);
        RV_LAUNCH_CHECK();
    }
    if (nbig > 0) {
        hipLaunchKernelGGL((k_bubble_child<1024, 4>), dim3((unsigned)nbig), dim3(1024), 0, ws.stream, b, d_big, (int64_t)1 << 62);
        RV_LAUNCH_CHECK();
    }
    return 0;
}

int rv_sai_level_launch(Workspace &ws, const sa_t *SA, int64_t m, const int64_t *sub_start, int nsubs, sa_t *SAi) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(k_sai_level, dim3((unsigned)ceil_div(m, TB)), dim3(TB), 0, ws.stream, SA, m, sub_start, nsubs, SAi);
    RV_LAUNCH_CHECK();
    return 0;
}

// rv_decide.hip -- the built-in callbacks of the untraced two-sample recursion, on the device.
//
// For a level whose sub-indices all own at most one interval per sample, the built-in picker
// (longest MUM of the sub-index) and the linear interval model of graphalign
// reduce to a closed rule per sub-index: with nodes [a0,a1) / [b0,b1) and the pick (a, b, l)
//     leading  = [a0,a) + [b0,b)      trailing = [a+l,a1) + [b+l,b1)      matched = [a,a+l) + [b,b+l)
// so the tables the split kernels need (reveal.c:1005-1117 labels, child offsets, cut windows) can
// be written by a kernel right behind the picker, and the split of the level starts without
// waiting for the host.  The host still receives the picks and rebuilds the same decisions for
// its own bookkeeping (next frontier, bubble descriptors, anchors); it does that while the split
// kernels run instead of in front of them.  Both sides use the same formulas; the child layout
// ((sub-index, class) order, empty children take no room) is the one of rv_frontier_commit.
#include "rv_common.h"
#include "rv_split.h"
#include "rv_scan.h"
#include "rv_decide.h"

namespace {

constexpr int TB = 256;

__device__ inline void decide_one(const RvDecideArgs &d, int s) {
    if (s > d.nsubs) return;
    if (s == d.nsubs) {      // closing entries of the fixed-stride index arrays
        d.ctab_first[s] = 4 * s; d.mtab_first[s] = 2 * s; d.cut_first[s] = 2 * s; d.mend_first[s] = 2 * s;
        return;
    }
    const sa_t a0 = d.nodes[4 * (size_t)s], a1 = d.nodes[4 * (size_t)s + 1], b0 = d.nodes[4 * (size_t)s + 2], b1 = d.nodes[4 * (size_t)s + 3];
    const RvPairRec rec = d.picks[RV_PAIR_HDR + s];
    const bool complete = reinterpret_cast<const u32 *>(d.picks)[1] <= d.ovf_cap;      // (the host reruns the scan with a larger overflow buffer otherwise)
    bool have = complete && rec.rank != 0xFFFFFFFFu && !(d.flags[s] & 1) && a0 < a1 && b0 < b1;
    sa_t a = 0, b = 0; sa_t l = 0;
    if (have) {
        a = rec.a; b = rec.b; l = (sa_t)rec.l;
        if (!(a0 <= a && a + l <= a1 && b0 <= b && b + l <= b1)) { atomicOr(d.err, 4u); have = false; }
    }
    d.ctab_first[s] = 4 * s; d.mtab_first[s] = 2 * s; d.cut_first[s] = 2 * s; d.mend_first[s] = 2 * s;
    sa_t *cb = d.cb + 4 * (size_t)s, *ce = d.ce + 4 * (size_t)s;
    uint8_t *cc = d.cc + 4 * (size_t)s;
    sa_t *mb = d.mb + 2 * (size_t)s, *me = d.me + 2 * (size_t)s;
    sa_t *clo = d.cut_lo + 2 * (size_t)s, *chi = d.cut_hi + 2 * (size_t)s, *mend = d.mend_pos + 2 * (size_t)s;
    u32 *cn = d.child_n + 3 * (size_t)s;
    if (!have) {
        for (int k = 0; k < 4; k++) { cb[k] = 0; ce[k] = 0; cc[k] = 0; }
        for (int k = 0; k < 2; k++) { mb[k] = 0; me[k] = 0; clo[k] = 0; chi[k] = 0; mend[k] = (sa_t)-1; }
        cn[0] = cn[1] = cn[2] = 0;
        return;
    }
    // class intervals in text order (sample 0 lies in front of sample 1); empty ones match nothing
    cb[0] = a0;    ce[0] = a;  cc[0] = 1;
    cb[1] = a + l; ce[1] = a1; cc[1] = 2;
    cb[2] = b0;    ce[2] = b;  cc[2] = 1;
    cb[3] = b + l; ce[3] = b1; cc[3] = 2;
    mb[0] = a; me[0] = a + l; mb[1] = b; me[1] = b + l;
    cn[0] = (u32)((a - a0) + (b - b0));
    cn[1] = (u32)((a1 - a - l) + (b1 - b - l));
    cn[2] = 0;
    // windows in front of the cuts (rv_frontier_commit: a cut with no leading interval in front of it has an empty window)
    const sa_t lcap = (sa_t)d.lcap;
    clo[0] = a > a0 ? (a - lcap > a0 ? a - lcap : a0) : a; chi[0] = a;
    clo[1] = b > b0 ? (b - lcap > b0 ? b - lcap : b0) : b; chi[1] = b;
    mend[0] = a + l; mend[1] = b + l;
}

__global__ __launch_bounds__(TB) void k_decide(RvDecideArgs d) { decide_one(d, blockIdx.x * TB + threadIdx.x); }

// child offsets: running offset over (sub-index, class) and the class totals in front of every sub-index.
// One block; sub-index counts beyond its reach use rv_decide_offsets_large.
template <bool FUSED>
__global__ __launch_bounds__(1024) void k_decide_offsets(RvDecideArgs d) {
    __shared__ u32 s_w[16][3];
    __shared__ u32 s_run[3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (FUSED) {      // few sub-indices: the decisions themselves in the same (single) workgroup -- one launch less per level
        for (int s = threadIdx.x; s <= d.nsubs; s += 1024) decide_one(d, s);
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x < 3) s_run[threadIdx.x] = 0;
    __syncthreads();
    for (int base = 0; base < d.nsubs; base += 1024) {
        const int s = base + threadIdx.x;
        u32 c0 = 0, c1 = 0;
        if (s < d.nsubs) { c0 = d.child_n[3 * (size_t)s]; c1 = d.child_n[3 * (size_t)s + 1]; }
        const u32 i0 = rv_wave_incl_sum_u32(c0), i1 = rv_wave_incl_sum_u32(c1);      // (DPP: this one workgroup is a chain of latencies, 50 rounds at 50000 sub-indices)
        if (lane == 63) { s_w[w][0] = i0; s_w[w][1] = i1; }
        __syncthreads();
        u32 b0 = s_run[0], b1 = s_run[1], t0 = 0, t1 = 0;
        for (int k = 0; k < 16; k++) { if (k < w) { b0 += s_w[k][0]; b1 += s_w[k][1]; } t0 += s_w[k][0]; t1 += s_w[k][1]; }
        if (s < d.nsubs) {
            const u32 g0 = b0 + i0 - c0, g1 = b1 + i1 - c1;            // class counts in front of sub-index s
            const u32 lead_base = g0 + g1, trail_base = lead_base + c0;   // (sub-index, class) order, rest children are empty here
            d.child_base[3 * (size_t)s] = lead_base; d.child_base[3 * (size_t)s + 1] = trail_base; d.child_base[3 * (size_t)s + 2] = trail_base + c1;
            d.sub_off[3 * (size_t)s] = lead_base - g0; d.sub_off[3 * (size_t)s + 1] = trail_base - g1; d.sub_off[3 * (size_t)s + 2] = trail_base + c1;
            // the leading child as a bubble descriptor (all its cuts in one workgroup, rv_split.hip)
            RvBubbleDesc kd; kd.off = (int64_t)lead_base; kd.B = 0; kd.wlo = 0; kd.cut0 = 2 * s; kd.cut1 = 2 * s + 2;
            const bool win = d.cut_lo[2 * (size_t)s] < d.cut_hi[2 * (size_t)s] || d.cut_lo[2 * (size_t)s + 1] < d.cut_hi[2 * (size_t)s + 1];
            kd.n = win ? (int64_t)c0 : 0;
            d.kid[s] = kd;
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_run[0] += t0; s_run[1] += t1; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { d.expect_total[0] = s_run[0]; d.expect_total[1] = s_run[1]; d.expect_total[2] = 0; d.expect_total[3] = 0; }
}

// ---- more than two samples ---------------------------------------------------------------------------
// the member of the pick that lies inside [b, e), or -1
__device__ inline int64_t member_in(const RvDecideMultiArgs &d, int s, int want, sa_t b, sa_t e, sa_t l) {
    int64_t p = -1;
    for (int k = 0; k < want; k++) {
        const sa_t x = d.pick_pos[(size_t)s * d.W + k];
        if (x >= b && x + l <= e) p = (int64_t)x;
    }
    return p;
}

__global__ __launch_bounds__(TB) void k_decide_multi(RvDecideMultiArgs d) {
    const int s = blockIdx.x * TB + threadIdx.x;
    if (s > d.nsubs) return;
    const int W = d.W;
    d.ctab_first[s] = 2 * W * s; d.mtab_first[s] = W * s; d.cut_first[s] = W * s; d.mend_first[s] = W * s;
    if (s == d.nsubs) return;
    const sa_t *nd = d.nodes + (size_t)2 * W * s;
    sa_t *cb = d.cb + (size_t)2 * W * s, *ce = d.ce + (size_t)2 * W * s;
    uint8_t *cc = d.cc + (size_t)2 * W * s;
    sa_t *mb = d.mb + (size_t)W * s, *me = d.me + (size_t)W * s;
    sa_t *clo = d.cut_lo + (size_t)W * s, *chi = d.cut_hi + (size_t)W * s, *mend = d.mend_pos + (size_t)W * s;
    u32 *cn = d.child_n + (size_t)3 * s;
    const bool complete = *d.cand_count <= d.cand_cap;
    const sa_t l = complete ? (sa_t)d.pick_l[s] : (sa_t)0;
    const int want = d.want[s];
    bool have = l > 0;
    // pass 1: sizes and the "cannot hold another match" rule (child_is_dead of rv_frontier_commit, one interval per sample)
    const int64_t need = d.minl > 1 ? d.minl : 1;
    const int nsmin = d.minn > 2 ? d.minn : 2;
    int64_t n0 = 0, n1 = 0, n2 = 0;
    int c0 = 0, c1 = 0, c2 = 0, members = 0;
    bool sh0 = false, sh1 = false, sh2 = false;
    if (have) {
        for (int q = 0; q < W; q++) {
            const sa_t b = nd[2 * q], e = nd[2 * q + 1];
            if (b >= e) continue;
            const int64_t p = member_in(d, s, want, b, e, l);
            if (p >= 0) {
                members++;
                const int64_t ll = p - b, tl = (int64_t)e - p - l;
                if (ll > 0) { n0 += ll; c0++; sh0 |= ll < need; }
                if (tl > 0) { n1 += tl; c1++; sh1 |= tl < need; }
            } else {
                n2 += (int64_t)e - b; c2++; sh2 |= ((int64_t)e - b) < need;
            }
        }
        if (members != want) { atomicOr(d.err, 4u); have = false; }      // a member outside the intervals of its sub-index
    }
    if (!have) {
        for (int k = 0; k < 2 * W; k++) { cb[k] = 0; ce[k] = 0; cc[k] = 0; }
        for (int k = 0; k < W; k++) { mb[k] = 0; me[k] = 0; clo[k] = 0; chi[k] = 0; mend[k] = (sa_t)-1; }
        cn[0] = cn[1] = cn[2] = 0;
        return;
    }
    const bool dead0 = c0 > 0 && (sh0 || c0 < nsmin), dead1 = c1 > 0 && (sh1 || c1 < nsmin), dead2 = c2 > 0 && (sh2 || c2 < nsmin);
    cn[0] = dead0 ? 0u : (u32)n0; cn[1] = dead1 ? 0u : (u32)n1; cn[2] = dead2 ? 0u : (u32)n2;
    const uint8_t k0 = dead0 ? 3 : 1, k1 = dead1 ? 3 : 2, k2 = dead2 ? 3 : 4;
    // pass 2: the tables, in text order (samples lie one behind the other); empty slots are empty intervals at the running position
    const sa_t lcap = (sa_t)d.lcap;
    sa_t cur = 0, curm = 0;
    for (int q = 0; q < W; q++) {
        const sa_t b = nd[2 * q], e = nd[2 * q + 1];
        if (b >= e) {
            cb[2 * q] = cur; ce[2 * q] = cur; cc[2 * q] = 0; cb[2 * q + 1] = cur; ce[2 * q + 1] = cur; cc[2 * q + 1] = 0;
            mb[q] = curm; me[q] = curm; clo[q] = 0; chi[q] = 0; mend[q] = (sa_t)-1;
            continue;
        }
        const int64_t pp = member_in(d, s, want, b, e, l);
        if (pp >= 0) {
            const sa_t p = (sa_t)pp;
            cb[2 * q] = b; ce[2 * q] = p; cc[2 * q] = k0;
            cb[2 * q + 1] = p + l; ce[2 * q + 1] = e; cc[2 * q + 1] = k1;
            mb[q] = p; me[q] = p + l; curm = p + l;
            clo[q] = p > b ? (p - lcap > b ? p - lcap : b) : p; chi[q] = p;
            mend[q] = p + l;
        } else {
            cb[2 * q] = b; ce[2 * q] = e; cc[2 * q] = k2;
            cb[2 * q + 1] = e; ce[2 * q + 1] = e; cc[2 * q + 1] = 0;
            mb[q] = curm > b ? curm : b; me[q] = mb[q]; curm = mb[q];
            clo[q] = 0; chi[q] = 0; mend[q] = (sa_t)-1;
        }
        cur = e;
    }
}

// child offsets for three classes: running offset over (sub-index, class), class totals in front of every sub-index; one block
__global__ __launch_bounds__(1024) void k_decide_offsets3(RvDecideMultiArgs d) {
    __shared__ u32 s_w[16][3];
    __shared__ u32 s_run[3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 3) s_run[threadIdx.x] = 0;
    __syncthreads();
    for (int base = 0; base < d.nsubs; base += 1024) {
        const int s = base + threadIdx.x;
        u32 c[3] = {0, 0, 0};
        if (s < d.nsubs) { c[0] = d.child_n[3 * (size_t)s]; c[1] = d.child_n[3 * (size_t)s + 1]; c[2] = d.child_n[3 * (size_t)s + 2]; }
        u32 i[3] = {c[0], c[1], c[2]};
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
#pragma unroll
            for (int k = 0; k < 3; k++) { const u32 t = __shfl_up(i[k], dd, 64); if (lane >= dd) i[k] += t; }
        }
        if (lane == 63) { s_w[w][0] = i[0]; s_w[w][1] = i[1]; s_w[w][2] = i[2]; }
        __syncthreads();
        u32 g[3], tot[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            u32 b = s_run[k], t = 0;
            for (int v = 0; v < 16; v++) { if (v < w) b += s_w[v][k]; t += s_w[v][k]; }
            g[k] = b + i[k] - c[k];                                   // class-k ranks in front of sub-index s
            tot[k] = t;
        }
        if (s < d.nsubs) {
            const u32 lead_base = g[0] + g[1] + g[2], trail_base = lead_base + c[0], rest_base = trail_base + c[1];
            d.child_base[3 * (size_t)s] = lead_base; d.child_base[3 * (size_t)s + 1] = trail_base; d.child_base[3 * (size_t)s + 2] = rest_base;
            d.sub_off[3 * (size_t)s] = lead_base - g[0]; d.sub_off[3 * (size_t)s + 1] = trail_base - g[1]; d.sub_off[3 * (size_t)s + 2] = rest_base - g[2];
            RvBubbleDesc kd; kd.off = (int64_t)lead_base; kd.B = 0; kd.wlo = 0; kd.cut0 = d.W * s; kd.cut1 = d.W * s + d.W;
            bool win = false;
            for (int q = 0; q < d.W; q++) win |= d.cut_lo[(size_t)d.W * s + q] < d.cut_hi[(size_t)d.W * s + q];
            kd.n = win ? (int64_t)c[0] : 0;
            d.kid[s] = kd;
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_run[0] += tot[0]; s_run[1] += tot[1]; s_run[2] += tot[2]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { d.expect_total[0] = s_run[0]; d.expect_total[1] = s_run[1]; d.expect_total[2] = s_run[2]; d.expect_total[3] = 0; }
}

}  // namespace

int rv_decide_multi_launch(Workspace &ws, const RvDecideMultiArgs &d) {
    if (d.nsubs <= 0) return 0;
    hipLaunchKernelGGL(k_decide_multi, dim3((unsigned)ceil_div((int64_t)d.nsubs + 1, TB)), dim3(TB), 0, ws.stream, d);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_decide_offsets3, dim3(1), dim3(1024), 0, ws.stream, d);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_decide_launch(Workspace &ws, const RvDecideArgs &d) {
    if (d.nsubs <= 0) return 0;
    if (d.nsubs <= 4096) {
        hipLaunchKernelGGL(k_decide_offsets<true>, dim3(1), dim3(1024), 0, ws.stream, d);
        RV_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_decide, dim3((unsigned)ceil_div((int64_t)d.nsubs + 1, TB)), dim3(TB), 0, ws.stream, d);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_decide_offsets<false>, dim3(1), dim3(1024), 0, ws.stream, d);
    RV_LAUNCH_CHECK();
    return 0;
}

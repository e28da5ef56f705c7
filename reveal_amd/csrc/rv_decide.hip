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
        u32 i0 = c0, i1 = c1;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const u32 t0 = __shfl_up(i0, dd, 64), t1 = __shfl_up(i1, dd, 64);
            if (lane >= dd) { i0 += t0; i1 += t1; }
        }
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

}  // namespace

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

// rv_mems.hip -- getmultimems (reveallib/reveal.c:292-434, ismultimem :261-290).
//
// Unlike getmultimums, whose intervals hold at most nsamples ranks and close independently of each other
// (k_scan_multi enumerates them without a stack), the reference's MEM enumeration carries an order dependence:
// a popped interval that qualifies as a multi-MEM but covers fewer than `minn` samples leaves the loop body through
// `continue` (reveal.c:340-342) and thereby skips `lb = i_lb` (:362), so the interval pushed next starts where the
// *last interval that did not take that exit* started -- which in turn depended on the intervals popped before it.
// The left bounds are defined by induction over the rank order.  First version, exact by construction: ONE wavefront
// replays the stack machine (LCP streamed through LDS, the stack in LDS with a global spill area, every lane holding
// the same state), and uses its 64 lanes where the reference loops over an interval's members: the sample census,
// the left-maximality test and the output of the members.  Measured 0.33 us per rank (a single wavefront issues a
// dependent instruction every ~10 cycles: that, not memory, is the bound); no caller of the reference uses this method
// (SURVEY.md 8(f) N1), so exactness came before speed.
#include "rv_common.h"
#include "rv_scan.h"

namespace {

constexpr int LCP_CHUNK = 2048;      // ranks staged per refill
constexpr int RING = 2 * LCP_CHUNK;  // SA / sample / BWT of the current and the previous chunk stay in LDS: an interval ends right behind the scan position
constexpr int ST_LDS = 4096;         // stack entries kept in LDS; deeper ones live in global memory

__device__ inline bool is_lower_c(uint8_t c) { return c >= 'a' && c <= 'z'; }

__device__ inline int sample_of_pos(const sa_t *__restrict__ nsep, int nsep_n, sa_t pos) {   // SO[pos], interface.c:116-134
    int lo = 0, hi = nsep_n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (nsep[mid] < pos) lo = mid + 1; else hi = mid; }
    return lo;
}

struct MemsArgs {
    const sa_t *SA; const lcp_t *LCP; const uint8_t *BWT; int64_t n;
    const sa_t *nsep; int nsamples, minl, minn;
    u32 *g_lcp; int64_t *g_lb; int64_t g_cap;            // stack entries ST_LDS.. (global spill)
    u32 *rec_l; int32_t *rec_c; int64_t *rec_first;       // per record: length, samples covered, first member
    uint16_t *so; sa_t *pos;
    unsigned long long rec_cap, mem_cap;
    unsigned long long *out;                              // [0] records, [1] members, [2] error bits
};

__global__ __launch_bounds__(64) void k_multimems_seq(MemsArgs A) {
    __shared__ u32 s_lcp[LCP_CHUNK];
    __shared__ sa_t r_sa[RING];
    __shared__ uint8_t r_so[RING], r_bw[RING];
    __shared__ u32 st_lcp[ST_LDS];
    __shared__ int64_t st_lb[ST_LDS];
    const int lane = threadIdx.x;
    unsigned long long nrec = 0, nmem = 0, err = 0;
    int64_t depth = 0;
    u32 top_lcp = 0; int64_t top_lb = 0;                  // stack[depth], kept in registers
    const int nsep_n = A.nsamples - 1;
    int64_t win_lo = 0;                                   // ranks [win_lo, chunk end) are in the ring

    // reveal.c:323-363 body for the interval (l, lb, ub) just popped; true = the reference's `continue`
    auto close = [&](u32 l, int64_t lb, int64_t ub) -> bool {
        const int64_t cnt = ub - lb + 1;
#ifdef RV_SA64
        if (l < (u32)A.minl) return false;
#else
        if ((int)l < A.minl) return false;
#endif
        if (cnt < (int64_t)A.minn) return false;
        if (l == 0) return false;                         // ismultimem: `if (l>0)` else 0
        // sample census (reveal.c:266-277) and left-maximality (:279-287), 64 members at a time
        u64 seen = 0; bool maximal = false;               // seen: wave-uniform mask of the samples met so far
        const bool in_lds = lb >= win_lo;
        for (int64_t j0 = lb; j0 <= ub; j0 += 64) {
            const int64_t j = j0 + lane;
            int my = -1;
            if (j <= ub) {
                if (A.nsamples > 2) my = in_lds ? (int)r_so[j & (RING - 1)] : sample_of_pos(A.nsep, nsep_n, A.SA[j]);
                if (j < ub) {
                    const uint8_t ca = in_lds ? r_bw[j & (RING - 1)] : (uint8_t)(A.BWT[j] & RV_BWT_CHAR), cb = in_lds ? r_bw[(j + 1) & (RING - 1)] : (uint8_t)(A.BWT[j + 1] & RV_BWT_CHAR);      // '$' stands for "position 0" (SA == 0)
                    maximal |= (cb == '$') | (ca != cb) | (ca == 'N') | (ca == '$') | is_lower_c(ca);
                }
            }
            // one ballot per sample (a cross-lane OR of 64-bit masks is twelve LDS-crossbar shuffles: it was most of the kernel's time)
            if (A.nsamples > 2)
                for (int sx = 0; sx < A.nsamples; sx++) if (__ballot(my == sx)) seen |= 1ull << sx;
        }
        if (!__any(maximal)) return false;
        const int cc = A.nsamples == 2 ? 1 : __popcll(seen);      // two samples: flag_so[a == b]++ : exactly one counter is positive
        if (cc < A.minn) return true;
        if (nrec < A.rec_cap && lane == 0) { A.rec_l[nrec] = l; A.rec_c[nrec] = cc; A.rec_first[nrec] = (int64_t)nmem; }
        if (nrec < A.rec_cap) {
            for (int64_t j0 = lb; j0 <= ub; j0 += 64) {
                const int64_t j = j0 + lane;
                const unsigned long long o = nmem + (unsigned long long)(j - lb);
                if (j <= ub && o < A.mem_cap) {
                    const sa_t p = in_lds ? r_sa[j & (RING - 1)] : A.SA[j];
                    A.so[o] = in_lds ? (uint16_t)r_so[j & (RING - 1)] : (uint16_t)(A.nsamples > 2 ? sample_of_pos(A.nsep, nsep_n, p) : (p > A.nsep[0] ? 1 : 0));
                    A.pos[o] = p;
                }
            }
        }
        nmem += (unsigned long long)cnt;
        nrec++;
        return false;
    };
    auto push = [&](u32 l, int64_t lb) {
        // the old top goes to memory, the new one stays in registers
        if (depth < ST_LDS) { if (lane == 0) { st_lcp[depth] = top_lcp; st_lb[depth] = top_lb; } }
        else if (depth - ST_LDS < A.g_cap) { if (lane == 0) { A.g_lcp[depth - ST_LDS] = top_lcp; A.g_lb[depth - ST_LDS] = top_lb; } }
        else err |= 1;
        depth++;
        top_lcp = l; top_lb = lb;      // (one wavefront: its LDS accesses are served in program order, no barrier needed before the next pop)
    };
    auto pop = [&]() {
        depth--;
        if (depth < ST_LDS) { top_lcp = st_lcp[depth]; top_lb = st_lb[depth]; }
        else if (depth - ST_LDS < A.g_cap) { top_lcp = A.g_lcp[depth - ST_LDS]; top_lb = A.g_lb[depth - ST_LDS]; }
    };

    const sa_t sep0 = A.nsep[0];
    for (int64_t base = 0; base < A.n; base += LCP_CHUNK) {      // chunk = ranks [base, base + LCP_CHUNK)
        __syncthreads();
        for (int k = lane; k < LCP_CHUNK; k += 64) {
            const int64_t r = base + k;
            if (r < A.n) {
                const sa_t p = A.SA[r];
                s_lcp[k] = (u32)A.LCP[r];
                r_sa[r & (RING - 1)] = p; r_bw[r & (RING - 1)] = A.BWT[r] & RV_BWT_CHAR;
                r_so[r & (RING - 1)] = (uint8_t)(A.nsamples > 2 ? sample_of_pos(A.nsep, nsep_n, p) : (p > sep0 ? 1 : 0));
            } else s_lcp[k] = 0u;
        }
        __syncthreads();
        win_lo = base >= LCP_CHUNK ? base - LCP_CHUNK : 0;
        const int64_t lim = A.n - base < LCP_CHUNK ? A.n - base : LCP_CHUNK;
        u32 vreg = 0;                                     // 64 LCP values at a time in one register, read by lane index
        for (int64_t k = (base == 0 ? 1 : 0); k < lim; k++) {
            const int64_t i = base + k;
            if ((k & 63) == 0 || k == 1) vreg = s_lcp[(k & ~(int64_t)63) + lane];
            const u32 v = (u32)__builtin_amdgcn_readlane((int)vreg, (int)(k & 63));
            int64_t lb = i - 1;
            while (v < top_lcp) {                         // reveal.c:322
                const u32 i_lcp = top_lcp; const int64_t i_lb = top_lb;
                pop();
                if (close(i_lcp, i_lb, i - 1)) continue;  // the quirk: lb keeps its value
                lb = i_lb;
            }
            if (v > top_lcp) push(v, lb);                 // reveal.c:365-389
        }
    }
    for (;;) {                                            // reveal.c:391-428: what is still open ends at n-1
        const u32 i_lcp = top_lcp; const int64_t i_lb = top_lb;
        (void)close(i_lcp, i_lb, A.n - 1);
        if (depth == 0) break;
        pop();
    }
    if (lane == 0) { A.out[0] = nrec; A.out[1] = nmem; A.out[2] = err; }
}

}  // namespace

// -> l, c (samples covered), first member of every multi-MEM in the reference's order; members (so, pos) back to back.
// out[0..2] = records, members, error bits; counts beyond the capacities are still counted (the caller grows and repeats).
int rv_multimems_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t n, const sa_t *nsep, int nsamples, int minl, int minn,
                        u32 *g_lcp, int64_t *g_lb, int64_t g_cap, u32 *rec_l, int32_t *rec_c, int64_t *rec_first, uint16_t *so, sa_t *pos,
                        unsigned long long rec_cap, unsigned long long mem_cap, unsigned long long *out) {
    MemsArgs A;
    A.SA = SA; A.LCP = LCP; A.BWT = BWT; A.n = n; A.nsep = nsep; A.nsamples = nsamples; A.minl = minl; A.minn = minn;
    A.g_lcp = g_lcp; A.g_lb = g_lb; A.g_cap = g_cap; A.rec_l = rec_l; A.rec_c = rec_c; A.rec_first = rec_first; A.so = so; A.pos = pos;
    A.rec_cap = rec_cap; A.mem_cap = mem_cap; A.out = out;
    hipLaunchKernelGGL(k_multimems_seq, dim3(1), dim3(64), 0, ws.stream, A);
    RV_LAUNCH_CHECK();
    return 0;
}

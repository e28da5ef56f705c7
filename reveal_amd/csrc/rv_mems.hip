// rv_mems.hip -- getmultimems (reveallib/reveal.c:292-434, ismultimem :261-290).
//
// Unlike getmultimums, whose intervals hold at most nsamples ranks and close independently of each other
// (k_scan_multi enumerates them without a stack), the reference's MEM enumeration carries an order dependence:
// a popped interval that qualifies as a multi-MEM but covers fewer than `minn` samples leaves the loop body through
// `continue` (reveal.c:340-342) and thereby skips `lb = i_lb` (:362), so the interval pushed next starts where the
// *last interval that did not take that exit* started -- which in turn depended on the intervals popped before it.
// The left bounds are defined by induction over the rank order.
//
// Rounds 1-4: ONE wavefront replayed the stack machine over the whole index (0.33 us per rank: 165 s for 5 x 10^8 ranks).
// Round 5: the induction never crosses a rank whose LCP value is below minl.  An interval below minl returns from the loop body
// at its first test (reveal.c:325): no record, never the `continue`; and an interval of minl and more inherits its left bound only
// from intervals popped at the rank it is pushed at -- all of them deeper than itself, so of minl and more as well.  A RUN -- a maximal
// stretch of ranks with LCP >= max(minl, 1): the occurrences of one minl-mer -- is therefore a stack machine of its own, started on
// an empty stack at its first rank and emptied by the first value below minl behind it, and the reference's output is the runs'
// outputs in rank order.  Runs are short (a minl-mer of related genomes occurs once per sample) and there are millions of them:
//   k_mems_runs<false>   a thread per 32 ranks replays every run that starts there (stack of 24 entries in registers / scratch)
//                        and counts its records and members; a run of more than 2048 ranks, or deeper than the stack, goes to a list
//   k_mems_long<false>   a wavefront per listed run: the old kernel's machine (LCP through LDS, stack in LDS with a spill area, 64 lanes
//                        share an interval's members), started at the run's first rank; counts
//   exclusive sums of the tiles' counts = where each tile's records and members go
//   k_mems_runs<true>, k_mems_long<true>   the same again, writing
// Exact by the same construction as before: tests/test_gpu_align.py getmultimems cases, the golden vectors, random inputs against the oracle.
#include "rv_common.h"
#include "rv_scan.h"

namespace {

constexpr int TB = 256;
constexpr int LCP_CHUNK = 2048;      // ranks staged per refill
constexpr int RING = 2 * LCP_CHUNK;  // SA / sample / BWT of the current and the previous chunk stay in LDS: an interval ends right behind the scan position
constexpr int ST_LDS = 4096;         // stack entries kept in LDS; deeper ones live in global memory
constexpr int MR = 32;               // ranks per thread of k_mems_runs
constexpr int ST_THREAD = 24;        // a thread's stack
constexpr int64_t RUN_MAX = 2048;    // ranks a thread follows a run for

__device__ inline bool is_lower_c(uint8_t c) { return c >= 'a' && c <= 'z'; }

__device__ inline int sample_of_pos(const sa_t *__restrict__ nsep, int nsep_n, sa_t pos) {   // SO[pos], interface.c:116-134
    int lo = 0, hi = nsep_n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (nsep[mid] < pos) lo = mid + 1; else hi = mid; }
    return lo;
}

struct MemsArgs {
    const sa_t *SA; const lcp_t *LCP; const uint8_t *BWT; int64_t n;
    const sa_t *nsep; int nsamples, minl, minn;
    u32 minl_e;                                           // max(minl, 1): what a run's LCP values reach
    u32 *g_lcp; int64_t *g_lb; int64_t g_cap;            // stack entries ST_LDS.. (global spill), g_cap per workgroup of k_mems_long
    u32 *rec_l; int32_t *rec_c; int64_t *rec_first;       // per record: length, samples covered, first member
    uint16_t *so; sa_t *pos;
    unsigned long long rec_cap, mem_cap;
    unsigned long long *out;                              // [0] records, [1] members, [2] error bits, [3] listed runs
    u64 *tile_rec, *tile_mem; int64_t ntiles;             // counts, then (exclusive sums) bases per tile of MR ranks
    int64_t *long_s; u64 *long_rec, *long_mem; u32 long_cap; u32 nlong;      // the listed runs: first rank (sorted by the host before the second pass), counts / bases
};

// ---- a run per thread ----------------------------------------------------------------------------------------------------------------------------
// the run whose first LCP value of minl and more stands at rank s (members from rank s - 1 on).  WRITE = false: counts its records and members;
// true: writes them from (nrec, nmem) on.  -> false: too long or too deep for a thread (nothing counted, nothing written)
template <bool WRITE>
__device__ bool run_thread(const MemsArgs &A, int64_t s, unsigned long long &nrec, unsigned long long &nmem) {
    u32 st_l[ST_THREAD]; int64_t st_b[ST_THREAD];
    int depth = 0;
    const int nsep_n = A.nsamples - 1;
    unsigned long long r0 = nrec, m0 = nmem;
    auto close = [&](u32 l, int64_t lb, int64_t ub) -> bool {      // reveal.c:323-363 for an interval of minl and more; true = the reference's `continue`
        const int64_t cnt = ub - lb + 1;
        if (cnt < (int64_t)A.minn) return false;
        u64 seen = 0; bool maximal = false;
        uint8_t ca = (uint8_t)(A.BWT[lb] & RV_BWT_CHAR);
        for (int64_t j = lb; j <= ub; j++) {
            if (A.nsamples > 2) seen |= 1ull << sample_of_pos(A.nsep, nsep_n, A.SA[j]);
            if (j < ub) {
                const uint8_t cb = (uint8_t)(A.BWT[j + 1] & RV_BWT_CHAR);      // '$' stands for "position 0" (SA == 0)
                maximal |= (cb == '$') | (ca != cb) | (ca == 'N') | (ca == '$') | is_lower_c(ca);
                ca = cb;
            }
        }
        if (!maximal) return false;
        const int cc = A.nsamples == 2 ? 1 : __popcll(seen);      // two samples: flag_so[a == b]++ : exactly one counter is positive
        if (cc < A.minn) return true;
        if (WRITE) {
            if (r0 < A.rec_cap) {
                A.rec_l[r0] = l; A.rec_c[r0] = cc; A.rec_first[r0] = (int64_t)m0;
                for (int64_t j = lb; j <= ub; j++) {
                    const unsigned long long o = m0 + (unsigned long long)(j - lb);
                    if (o < A.mem_cap) {
                        const sa_t p = A.SA[j];
                        A.so[o] = (uint16_t)(A.nsamples > 2 ? sample_of_pos(A.nsep, nsep_n, p) : (p > A.nsep[0] ? 1 : 0));
                        A.pos[o] = p;
                    }
                }
            }
        }
        m0 += (unsigned long long)cnt; r0++;
        return false;
    };
    for (int64_t i = s;; i++) {
        if (i - s > RUN_MAX) return false;
        const u32 v = i < A.n ? (u32)A.LCP[i] : 0u;
        const bool open = i < A.n && v >= A.minl_e;          // the run goes on
        int64_t lb = i - 1;
        while (depth > 0 && v < st_l[depth - 1]) {            // reveal.c:322
            depth--;
            const u32 i_lcp = st_l[depth]; const int64_t i_lb = st_b[depth];
            if (close(i_lcp, i_lb, i - 1)) continue;          // the quirk: lb keeps its value
            lb = i_lb;
        }
        if (!open) break;                                     // (everything of minl and more has been popped: v is below all of it)
        if (depth == 0 || v > st_l[depth - 1]) {              // reveal.c:365-389
            if (depth == ST_THREAD) return false;
            st_l[depth] = v; st_b[depth] = lb; depth++;
        }
    }
    nrec = r0; nmem = m0;
    return true;
}

template <bool WRITE>
__global__ __launch_bounds__(TB) void k_mems_runs(MemsArgs A) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= A.ntiles) return;
    const int64_t lo = t * MR > 1 ? t * MR : 1, hi = (t + 1) * MR < A.n ? (t + 1) * MR : A.n;
    unsigned long long nrec = WRITE ? A.tile_rec[t] : 0ull, nmem = WRITE ? A.tile_mem[t] : 0ull;
    u32 prev = lo > 1 ? (u32)A.LCP[lo - 1] : 0u;
    for (int64_t i = lo; i < hi; i++) {
        const u32 v = (u32)A.LCP[i];
        if (v >= A.minl_e && (i == 1 || prev < A.minl_e)) {
            if (WRITE) {
                // a run the count pass listed for the wavefront machine is not walked again (it would write up to 2048 ranks' records that
                // k_mems_long rewrites at the same places, only to fail where it failed before): its counts are known -- where it goes, and past it
                u32 a = 0, b = A.nlong;
                while (a < b) { const u32 mid = (a + b) >> 1; if (A.long_s[mid] < i) a = mid + 1; else b = mid; }
                if (a < A.nlong && A.long_s[a] == i) {
                    const u64 cr = A.long_rec[a], cm = A.long_mem[a];
                    A.long_rec[a] = nrec; A.long_mem[a] = nmem;
                    nrec += cr; nmem += cm;
                } else {
                    (void)run_thread<true>(A, i, nrec, nmem);
                }
            } else if (!run_thread<false>(A, i, nrec, nmem)) {
                const u32 x = (u32)atomicAdd(&A.out[3], 1ull);
                if (x < A.long_cap) A.long_s[x] = i;
            }
        }
        prev = v;
    }
    if (!WRITE) { A.tile_rec[t] = nrec; A.tile_mem[t] = nmem; }
}

// ---- a run per wavefront: the machine of rounds 1-4, started at a run's first rank -------------------------------------------------------------
template <bool WRITE>
__global__ __launch_bounds__(64) void k_mems_long(MemsArgs A) {
    __shared__ u32 s_lcp[LCP_CHUNK];
    __shared__ sa_t r_sa[RING];
    __shared__ uint8_t r_so[RING], r_bw[RING];
    __shared__ u32 st_lcp[ST_LDS];
    __shared__ int64_t st_lb[ST_LDS];
    const int lane = threadIdx.x;
    const int nsep_n = A.nsamples - 1;
    u32 *const g_lcp = A.g_lcp + (size_t)blockIdx.x * (size_t)A.g_cap;
    int64_t *const g_lb = A.g_lb + (size_t)blockIdx.x * (size_t)A.g_cap;
    for (u32 run = blockIdx.x; run < A.nlong; run += gridDim.x) {
    const int64_t s = A.long_s[run];
    unsigned long long nrec = WRITE ? A.long_rec[run] : 0ull, nmem = WRITE ? A.long_mem[run] : 0ull, err = 0;
    const unsigned long long rec0 = nrec, mem0 = nmem;
    int64_t depth = 0;
    u32 top_lcp = 0; int64_t top_lb = 0;                  // stack[depth], kept in registers; entry 0 = (0, 0): below every run
    int64_t win_lo = 0;                                   // ranks [win_lo, chunk end) are in the ring

    // reveal.c:323-363 body for the interval (l, lb, ub) just popped; true = the reference's `continue`
    auto close = [&](u32 l, int64_t lb, int64_t ub) -> bool {
        const int64_t cnt = ub - lb + 1;
        if (l < A.minl_e) return false;
        if (cnt < (int64_t)A.minn) return false;
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
        if (WRITE) {
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
        }
        nmem += (unsigned long long)cnt;
        nrec++;
        return false;
    };
    auto push = [&](u32 l, int64_t lb) {
        // the old top goes to memory, the new one stays in registers
        if (depth < ST_LDS) { if (lane == 0) { st_lcp[depth] = top_lcp; st_lb[depth] = top_lb; } }
        else if (depth - ST_LDS < A.g_cap) { if (lane == 0) { g_lcp[depth - ST_LDS] = top_lcp; g_lb[depth - ST_LDS] = top_lb; } }
        else err |= 1;
        depth++;
        top_lcp = l; top_lb = lb;      // (one wavefront: its LDS accesses are served in program order, no barrier needed before the next pop)
    };
    auto pop = [&]() {
        depth--;
        if (depth < ST_LDS) { top_lcp = st_lcp[depth]; top_lb = st_lb[depth]; }
        else if (depth - ST_LDS < A.g_cap) { top_lcp = g_lcp[depth - ST_LDS]; top_lb = g_lb[depth - ST_LDS]; }
    };

    const sa_t sep0 = A.nsep[0];
    const int64_t base0 = (s / LCP_CHUNK) * LCP_CHUNK;
    bool done = false;
    for (int64_t base = base0; base < A.n && !done; base += LCP_CHUNK) {      // chunk = ranks [base, base + LCP_CHUNK)
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
        win_lo = base > base0 ? base - LCP_CHUNK : base0;      // (the chunk in front of the run's first one was never staged)
        const int64_t lim = A.n - base < LCP_CHUNK ? A.n - base : LCP_CHUNK;
        const int64_t k0 = base == base0 ? s - base0 : 0;
        u32 vreg = 0;                                     // 64 LCP values at a time in one register, read by lane index
        for (int64_t k = k0; k < lim; k++) {
            const int64_t i = base + k;
            if ((k & 63) == 0 || k == k0) vreg = s_lcp[(k & ~(int64_t)63) + lane];
            const u32 v = (u32)__builtin_amdgcn_readlane((int)vreg, (int)(k & 63));
            int64_t lb = i - 1;
            while (v < top_lcp) {                         // reveal.c:322
                const u32 i_lcp = top_lcp; const int64_t i_lb = top_lb;
                pop();
                if (close(i_lcp, i_lb, i - 1)) continue;  // the quirk: lb keeps its value
                lb = i_lb;
            }
            if (v < A.minl_e) { done = true; break; }     // the run is over: what it opened has been popped
            if (v > top_lcp) push(v, lb);                 // reveal.c:365-389
        }
    }
    if (!done)                                            // reveal.c:391-428: what is still open ends at n-1
        while (depth > 0) {
            const u32 i_lcp = top_lcp; const int64_t i_lb = top_lb;
            pop();
            (void)close(i_lcp, i_lb, A.n - 1);
        }
    if (lane == 0) {
        if (!WRITE) { A.long_rec[run] = nrec - rec0; A.long_mem[run] = nmem - mem0; }
        if (err) atomicOr(&A.out[2], err);
    }
    __syncthreads();
    }
}

__global__ void k_mems_totals(MemsArgs A) {      // the sums behind the last tile
    A.out[0] = A.tile_rec[A.ntiles]; A.out[1] = A.tile_mem[A.ntiles];
}
__global__ void k_mems_add_long(MemsArgs A) {    // a listed run's counts into the tile it starts in
    const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= A.nlong) return;
    const int64_t t = A.long_s[x] / MR;
    atomicAdd((unsigned long long *)&A.tile_rec[t], (unsigned long long)A.long_rec[x]);
    atomicAdd((unsigned long long *)&A.tile_mem[t], (unsigned long long)A.long_mem[x]);
}

}  // namespace

// -> l, c (samples covered), first member of every multi-MEM in the reference's order; members (so, pos) back to back.
// out[0..2] = records, members, error bits; counts beyond the capacities are still counted (the caller grows and repeats).
int rv_multimems_launch(Workspace &ws, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t n, const sa_t *nsep, int nsamples, int minl, int minn,
                        u32 maxlcp, u32 *rec_l, int32_t *rec_c, int64_t *rec_first, uint16_t *so, sa_t *pos,
                        unsigned long long rec_cap, unsigned long long mem_cap, unsigned long long *out) {
    hipStream_t q = ws.stream;
    MemsArgs A;
    A.SA = SA; A.LCP = LCP; A.BWT = BWT; A.n = n; A.nsep = nsep; A.nsamples = nsamples; A.minl = minl; A.minn = minn;
    A.minl_e = (u32)std::max(minl, 1);
    A.rec_l = rec_l; A.rec_c = rec_c; A.rec_first = rec_first; A.so = so; A.pos = pos;
    A.rec_cap = rec_cap; A.mem_cap = mem_cap; A.out = out;
    A.ntiles = ceil_div(n, MR);
    DBuf &btile = ws.misc[16], &blong = ws.misc[17], &bst = ws.misc[11];
    RV_TRY(btile.reserve((size_t)(A.ntiles + 1) * 16 + 64));
    A.tile_rec = btile.as<u64>(); A.tile_mem = A.tile_rec + (A.ntiles + 1);
    // the list of runs for the wavefront machine: runs beyond 2048 ranks (n / 2048 at most) and runs deeper than a thread's stack.  Some 25 ranks
    // are enough for the latter (tandem arrays, homopolymers: up to n / 26 runs), so a count pass that finds more than the list holds is repeated
    // with a list of that size
    A.long_cap = (u32)std::min<int64_t>(n / 64 + 1024, 0x7fffffff);
    A.nlong = 0; A.g_lcp = nullptr; A.g_lb = nullptr; A.g_cap = 0;
    const unsigned grid = (unsigned)ceil_div(A.ntiles, TB);
    unsigned long long res[4] = {0, 0, 0, 0};
    for (int attempt = 0; ; attempt++) {
        RV_TRY(blong.reserve((size_t)A.long_cap * 24 + 64));
        A.long_s = blong.as<int64_t>(); A.long_rec = (u64 *)(A.long_s + A.long_cap); A.long_mem = A.long_rec + A.long_cap;
        RV_HIP(hipMemsetAsync(out, 0, 32, q));
        RV_HIP(hipMemsetAsync(A.tile_rec + A.ntiles, 0, 8, q)); RV_HIP(hipMemsetAsync(A.tile_mem + A.ntiles, 0, 8, q));
        hipLaunchKernelGGL(k_mems_runs<false>, dim3(grid), dim3(TB), 0, q, A);
        RV_LAUNCH_CHECK();
        RV_TRY(rv_read_back(ws, res, out, sizeof res));
        if (res[3] <= A.long_cap) break;
        if (attempt || res[3] > 0x7fffffffull) { rv_set_error("getmultimems: more long runs than the list holds"); return -1; }
        A.long_cap = (u32)res[3];
    }
    A.nlong = (u32)res[3];
    unsigned lgrid = 0;
    if (A.nlong) {
        // the listed runs in rank order (the second pass looks its runs up), then their counts by the wavefront machine
        std::vector<int64_t> ls(A.nlong);
        RV_HIP(hipMemcpy(ls.data(), A.long_s, (size_t)A.nlong * 8, hipMemcpyDeviceToHost));
        std::sort(ls.begin(), ls.end());
        RV_HIP(hipMemcpy(A.long_s, ls.data(), (size_t)A.nlong * 8, hipMemcpyHostToDevice));
        A.g_cap = (int64_t)maxlcp + 16;      // the stack holds strictly increasing LCP values
        lgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(A.nlong, 1024), ((int64_t)1 << 24) / A.g_cap));
        RV_TRY(bst.reserve((size_t)lgrid * (size_t)A.g_cap * 12 + 64));
        A.g_lb = bst.as<int64_t>(); A.g_lcp = (u32 *)(A.g_lb + (size_t)lgrid * (size_t)A.g_cap);
        hipLaunchKernelGGL(k_mems_long<false>, dim3(lgrid), dim3(64), 0, q, A);
        RV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_mems_add_long, dim3((unsigned)ceil_div((int64_t)A.nlong, TB)), dim3(TB), 0, q, A);
        RV_LAUNCH_CHECK();
    }
    RV_TRY(rv_exclusive_sum_u64(ws, A.tile_rec, A.tile_rec, A.ntiles + 1));
    RV_TRY(rv_exclusive_sum_u64(ws, A.tile_mem, A.tile_mem, A.ntiles + 1));
    hipLaunchKernelGGL(k_mems_totals, dim3(1), dim3(1), 0, q, A);
    RV_LAUNCH_CHECK();
    RV_TRY(rv_read_back(ws, res, out, 24));
    if (res[2]) return 0;                                             // (the caller reports it)
    if (res[0] > rec_cap || res[1] > mem_cap) return 0;               // (the caller grows its arrays and comes again)
    hipLaunchKernelGGL(k_mems_runs<true>, dim3(grid), dim3(TB), 0, q, A);
    RV_LAUNCH_CHECK();
    if (A.nlong) {
        hipLaunchKernelGGL(k_mems_long<true>, dim3(lgrid), dim3(64), 0, q, A);
        RV_LAUNCH_CHECK();
    }
    return 0;
}

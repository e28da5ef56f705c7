// rv_common.h -- shared declarations of the reveal_amd HIP library (gfx950 only).
//
// Index-width switch exactly like the reference's two modules
// (reveallib/reveal.h:7-13): default = reveallib (int32 SA, int32 LCP),
// -DRV_SA64 = reveallib64 (int64 SA, uint32 LCP).
#pragma once
#include <mutex>
#include <thread>
#include <vector>
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stddef.h>
#include <vector>
#include <string>

#ifdef RV_SA64
typedef int64_t  sa_t;
typedef uint32_t lcp_t;
#else
typedef int32_t  sa_t;
typedef int32_t  lcp_t;
#endif

typedef unsigned long long u64;
typedef unsigned int       u32;

void rv_set_error(const char *fmt, ...);

#define RV_HIP(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            rv_set_error("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
            return -1;                                                                         \
        }                                                                                      \
    } while (0)
#define RV_TRY(x)                                                                              \
    do {                                                                                       \
        int r_ = (x);                                                                          \
        if (r_ != 0) return r_;                                                                \
    } while (0)
// ---- switches (test hooks, diagnostics, A/B paths).  The library never reads the process environment: a handle starts with the
// defaults below and rv_set_option() (include/reveal_amd.h) changes one; the Python layer applies RV_* environment variables
// when it makes a handle (reveal_amd/_index.py).  name = the historical environment spelling.
#define RV_OPTION_LIST(X) \
    X(bubble_par_min, "RV_BUBBLE_PAR_MIN", -1) \
    X(presel_log, "RV_PRESEL_LOG", 0) \
    X(presel_host, "RV_PRESEL_HOST", 0) \
    X(leaf_acap, "RV_LEAF_ACAP", 256) \
    X(bubble_lds_always, "RV_BUBBLE_LDS_ALWAYS", 0) \
    X(no_early_bubble, "RV_NO_EARLY_BUBBLE", 0) \
    X(early_bubble_many, "RV_EARLY_BUBBLE_MANY", 0) \
    X(no_early_split, "RV_NO_EARLY_SPLIT", 0) \
    X(keep_dead, "RV_KEEP_DEAD", 0) \
    X(bubble_no_lds, "RV_BUBBLE_NO_LDS", 0) \
    X(bubble_no_join, "RV_BUBBLE_NO_JOIN", 0) \
    X(bubble_no_merge, "RV_BUBBLE_NO_MERGE", 0) \
    X(tables_memcpy, "RV_TABLES_MEMCPY", 0) \
    X(bubble_parent_scratch, "RV_BUBBLE_PARENT_SCRATCH", 0) \
    X(level_log, "RV_LEVEL_LOG", 0) \
    X(pb_two_pass, "RV_PB_TWO_PASS", 0) \
    X(no_leaf, "RV_NO_LEAF", 0) \
    X(leaf_prof, "RV_LEAF_PROF", 0) \
    X(no_cascade, "RV_NO_CASCADE", 0) \
    X(cascade_second, "RV_CASCADE_SECOND", 0) \
    X(cascade_danger, "RV_CASCADE_DANGER", 1) \
    X(cascade_second_off, "RV_CASCADE_SECOND_OFF", 0) \
    X(sync_block, "RV_SYNC_BLOCK", 0) \
    X(pb_refresh_tmin, "RV_PB_REFRESH_TMIN", 0) \
    X(cascade_log, "RV_CASCADE_LOG", 0) \
    X(cascade_danger_min, "RV_CASCADE_DANGER_MIN", 0) \
    X(cascade_batch, "RV_CASCADE_BATCH", 8) \
    X(lcp_by_rank, "RV_LCP_BY_RANK", 0) \
    X(no_tiny_sa, "RV_NO_TINY_SA", 0) \
    X(no_short_alphabet, "RV_NO_SHORT_ALPHABET", 0) \
    X(no_fused_lcp, "RV_NO_FUSED_LCP", 0) \
    X(no_diag, "RV_NO_DIAG", 0) \
    X(no_packed_text, "RV_NO_PACKED_TEXT", 0) \
    X(no_heads_fusion, "RV_NO_HEADS_FUSION", 0) \
    X(no_twin_collapse, "RV_NO_TWIN_COLLAPSE", 0) \
    X(no_far_twins, "RV_NO_FAR_TWINS", 0) \
    X(no_lcp_list, "RV_NO_LCP_LIST", 0) \
    X(no_text_jump, "RV_NO_TEXT_JUMP", 0) \
    X(far_table, "RV_FAR_TABLE", 0) \
    X(no_slow_class, "RV_NO_SLOW_CLASS", 0) \
    X(no_dwalk_blocks, "RV_NO_DWALK_BLOCKS", 0) \
    X(no_pub_twins, "RV_NO_PUB_TWINS", 0) \
    X(sa_no_text, "RV_SA_NO_TEXT", 0) \
    X(text_mode, "RV_TEXT_MODE", -1) \
    X(carry_ch, "RV_CARRY_CH", -1) \
    X(rs_bits, "RV_RS_BITS", 8) \
    X(rs_xcd, "RV_RS_XCD", 1) \
    X(rs_cnt16, "RV_RS_CNT16", 1) \
    X(rs_no_digit_bytes, "RV_RS_NO_DIGIT_BYTES", 0) \
    X(diag_table, "RV_DIAG_TABLE", -1) \
    X(no_cascade_chain, "RV_NO_CASCADE_CHAIN", 0) \
    X(casm_rank_count, "RV_CASM_RANK_COUNT", 0) \
    X(casm_big_min, "RV_CASM_BIG_MIN", -1) \
    X(casm_no_big, "RV_CASM_NO_BIG", 0) \
    X(casm_big_root, "RV_CASM_BIG_ROOT", 1 << 22) \
    X(casm_big_total, "RV_CASM_BIG_TOTAL", 1 << 26) \
    X(lock_any, "RV_LOCK_ANY", 0) \
    X(presel_dev_min, "RV_PRESEL_DEV_MIN", 65536) \
    X(scan_v1, "RV_SCAN_V1", 0) \
    X(pick_threads, "RV_PICK_THREADS", 0) \
    X(cascade_prio, "RV_CASCADE_PRIO", 0)
struct RvOptions {
#define RV_X_(f, name, def) int64_t f = def;
    RV_OPTION_LIST(RV_X_)
#undef RV_X_
    int64_t *find(const char *name) {
#define RV_X_(f, nm, def) if (strcmp(name, nm) == 0) return &f;
        RV_OPTION_LIST(RV_X_)
#undef RV_X_
        return nullptr;
    }
};
extern int g_rv_launch_trace;      // process-wide (rv_api.hip): rv_set_launch_trace(), not a switch of a handle
// RV_LAUNCH_TRACE=1 (diagnostics): print the source line of every kernel launch and wait for it, so that a GPU memory fault
// (which aborts the process) names the kernel behind it
static inline bool rv_launch_trace_on() { return g_rv_launch_trace != 0; }
#define RV_LAUNCH_CHECK()                                                                      \
    do {                                                                                       \
        RV_HIP(hipGetLastError());                                                             \
        if (rv_launch_trace_on()) {                                                            \
            fprintf(stderr, "launch %s:%d\n", __FILE__, __LINE__);                             \
            fflush(stderr);                                                                    \
            RV_HIP(hipDeviceSynchronize());                                                    \
        }                                                                                      \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// BWT level arrays: one byte per rank = the character in front of the suffix ('$' for text position 0); all of them are
// ASCII, so bit 7 is free: it says on which side of the first sample separator the suffix starts (reveal.c:73).  The pair
// scan then needs LCP and this byte only (5 B per rank instead of 9) and fetches SA for its few survivors.  The byte travels
// with its rank through split, bubble_sort and the frontier hand-off; readers of the character mask the bit.
#define RV_BWT_SIDE 0x80u
#define RV_BWT_CHAR 0x7fu

#ifdef __HIPCC__
// wave-wide minimum of a u32, broadcast to every lane, without LDS traffic: DPP row operations inside the four 16-lane rows,
// row broadcasts across them, the total read from lane 63.  (Six __shfl_xor steps are six ds_bpermute round trips through the
// LDS pipeline, ~100 cycles each and serialised by their dependence: fine once per tile, ruinous once per 64 ranks.)
__device__ inline u32 rv_wave_min_u32(u32 v) {
    const int inf = -1;      // 0xFFFFFFFF: what a lane without a source lane sees
    int x = (int)v, y;
#define RV_DPP_MIN(ctrl, rmask)                                                        \
    y = __builtin_amdgcn_update_dpp(inf, x, ctrl, rmask, 0xf, false);                  \
    x = (int)(((u32)y < (u32)x) ? (u32)y : (u32)x);
    RV_DPP_MIN(0xB1, 0xf)     // quad_perm [1,0,3,2]
    RV_DPP_MIN(0x4E, 0xf)     // quad_perm [2,3,0,1]
    RV_DPP_MIN(0x141, 0xf)    // row_half_mirror
    RV_DPP_MIN(0x140, 0xf)    // row_mirror: every lane of a row holds the row's minimum
    RV_DPP_MIN(0x142, 0xa)    // row_bcast:15 into rows 1 and 3
    RV_DPP_MIN(0x143, 0xc)    // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's minimum
#undef RV_DPP_MIN
    return (u32)__builtin_amdgcn_readlane(x, 63);
}
#endif

#ifdef __HIPCC__
// One step of a wave-wide inclusive scan by DPP: the value of the lane `shift` lanes below inside a 16-lane row (row_shr), of
// lane 15 of the previous row (row_bcast:15 into rows 1 and 3), of lane 31 (row_bcast:31 into rows 2 and 3).  `take` says
// whether this lane has such a source: Hillis-Steele inside the rows (shifts 1, 2, 4, 8), then the two broadcasts --
// six DPP moves per scanned word where __shfl_up needs six trips through the LDS crossbar.
template <int CTRL, int ROWMASK> __device__ inline u32 rv_dpp_u32(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWMASK, 0xf, false); }
#define RV_WAVE_SCAN_STEPS(STEP)                                                     \
    STEP(0x111, 0xf, (lane & 15) >= 1)                                               \
    STEP(0x112, 0xf, (lane & 15) >= 2)                                               \
    STEP(0x114, 0xf, (lane & 15) >= 4)                                               \
    STEP(0x118, 0xf, (lane & 15) >= 8)                                               \
    STEP(0x142, 0xa, (lane & 31) >= 16)                                              \
    STEP(0x143, 0xc, lane >= 32)
// maximum over the wave (every lane gets it)
__device__ inline u64 rv_wave_max_u64(u64 v) {
#define RV_STEP_(CTRL, RM, TAKE) { const u64 t = ((u64)rv_dpp_u32<CTRL, RM>((u32)(v >> 32)) << 32) | rv_dpp_u32<CTRL, RM>((u32)v); v = t > v ? t : v; }
    RV_WAVE_SCAN_STEPS(RV_STEP_)        // (a lane without a source lane sees 0: the identity)
#undef RV_STEP_
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, 63);
}
// inclusive maximum over the run of equal `seg` values a lane belongs to (runs are stretches of consecutive lanes): lane i gets the maximum of the
// values of the lanes of its run at or below i.  The same six DPP steps as the scans: a lane takes its source lane's value when that lane belongs
// to the same run (then every lane in between does).  (Written with __shfl_up this was eighteen dependent trips through the LDS crossbar per call.)
__device__ inline u64 rv_wave_seg_max_u64(u32 seg, u64 v) {
    const int lane = threadIdx.x & 63;
    u32 lo = (u32)v, hi = (u32)(v >> 32);
#define RV_STEP_(CTRL, RM, TAKE) { const u32 os = rv_dpp_u32<CTRL, RM>(seg), ol = rv_dpp_u32<CTRL, RM>(lo), oh = rv_dpp_u32<CTRL, RM>(hi); \
                                   const bool up = (TAKE) && os == seg && (oh > hi || (oh == hi && ol > lo)); lo = up ? ol : lo; hi = up ? oh : hi; }
    RV_WAVE_SCAN_STEPS(RV_STEP_)
#undef RV_STEP_
    return ((u64)hi << 32) | lo;
}
__device__ inline u32 rv_wave_seg_max_u32(u32 seg, u32 v) {
    const int lane = threadIdx.x & 63;
#define RV_STEP_(CTRL, RM, TAKE) { const u32 os = rv_dpp_u32<CTRL, RM>(seg), ov = rv_dpp_u32<CTRL, RM>(v); v = ((TAKE) && os == seg && ov > v) ? ov : v; }
    RV_WAVE_SCAN_STEPS(RV_STEP_)
#undef RV_STEP_
    return v;
}
// inclusive prefix sum over the wave
__device__ inline u32 rv_wave_incl_sum_u32(u32 v) {
    const int lane = threadIdx.x & 63;
#define RV_STEP_(CTRL, RM, TAKE) { const u32 t = rv_dpp_u32<CTRL, RM>(v); v += (TAKE) ? t : 0u; }
    RV_WAVE_SCAN_STEPS(RV_STEP_)
#undef RV_STEP_
    return v;
}
#endif

// When a device allocation fails: give back the SA-build scratch of every handle that is not inside a construct() right now
// (it is kept between calls so that a benchmark step does not pay for hipMalloc -- 160 GB at 2.2 x 10^9 positions in the 64-bit
// library, more than the recursion's level arrays find room beside).  true = something was released (rv_api.hip).
bool rv_oom_trim();

// Wait for an event by polling: the wake-up of a sleeping hipEventSynchronize costs tens of microseconds, and the waits on the path
// (a few counters between launches) are 5-15 us long.  The first ~20 us spin (pause instruction: the sibling hyper-thread keeps its
// issue slots), after that the thread yields between polls, and after ~200 us it sleeps in the runtime -- a host application that
// runs many handles does not burn a core per handle on a long wait (bench.py --jobs).
inline hipError_t rv_event_wait(hipEvent_t ev) {
    hipError_t e;
    for (int spins = 0; (e = hipEventQuery(ev)) == hipErrorNotReady; spins++) {
        if (spins < 64) { __builtin_ia32_pause(); continue; }
        if (spins < 640) { std::this_thread::yield(); continue; }
        return hipEventSynchronize(ev);
    }
    return e;
}

// Grow-only device buffer.
struct DBuf {
    void  *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        // (the growth margin saves re-allocations while inputs of similar size follow each other; on the arrays of a large index -- 256 MB and
        //  more -- an eighth was 8.5 bytes per text position of device memory)
        size_t want = bytes + (bytes >= ((size_t)256 << 20) ? bytes / 64 : bytes / 8) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            if (rv_oom_trim()) e = hipMalloc(&p, want);
            if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); want = bytes + 256; e = hipMalloc(&p, want); }      // without the growth margin
        }
        if (e != hipSuccess) { p = nullptr; rv_set_error("%s:%d hipMalloc(%zu bytes): %s", __FILE__, __LINE__, want, hipGetErrorString(e)); return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

// Pinned host staging buffer (grow-only).
// Streams and pinned host buffers outlive the handle that made them.  `reveal refine` builds an index per bubble -- a few hundred bases,
// 0.9 ms of construct + recursion -- and a handle's three streams and four pinned buffers cost 6 ms to create and destroy
// (hipStreamCreateWithFlags 0.65-2 ms, hipStreamDestroy 0.96 ms, hipHostFree 0.23 ms each: rocprofv3 --hip-trace of 30 such alignments).
// Idle ones wait here for the next handle on the same device; nothing is given back at process exit.
struct RvPools {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> streams;
    struct Pin { int dev; void *p; size_t cap; };
    std::vector<Pin> pinned;
    size_t pinned_bytes = 0;
};
inline RvPools &rv_pools() { static RvPools *p = new RvPools(); return *p; }
inline hipError_t rv_stream_get(hipStream_t *s) {
    int dev = 0; (void)hipGetDevice(&dev);
    {
        RvPools &P = rv_pools();
        std::lock_guard<std::mutex> g(P.mu);
        for (size_t k = P.streams.size(); k-- > 0;)
            if (P.streams[k].first == dev) { *s = P.streams[k].second; P.streams.erase(P.streams.begin() + (ptrdiff_t)k); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
inline void rv_stream_put(hipStream_t s) {      // (the caller has set the stream's device)
    if (!s) return;
    (void)hipStreamSynchronize(s);
    int dev = 0; (void)hipGetDevice(&dev);
    RvPools &P = rv_pools();
    {
        std::lock_guard<std::mutex> g(P.mu);
        if (P.streams.size() < 64) { P.streams.push_back({dev, s}); return; }
    }
    (void)hipStreamDestroy(s);
}
inline hipError_t rv_pinned_get(size_t bytes, void **p, size_t *cap) {
    int dev = 0; (void)hipGetDevice(&dev);
    {
        RvPools &P = rv_pools();
        std::lock_guard<std::mutex> g(P.mu);
        size_t best = (size_t)-1;
        for (size_t k = 0; k < P.pinned.size(); k++)
            if (P.pinned[k].dev == dev && P.pinned[k].cap >= bytes && P.pinned[k].cap <= 4 * bytes + (1u << 20) && (best == (size_t)-1 || P.pinned[k].cap < P.pinned[best].cap)) best = k;
        if (best != (size_t)-1) { *p = P.pinned[best].p; *cap = P.pinned[best].cap; P.pinned_bytes -= *cap; P.pinned.erase(P.pinned.begin() + (ptrdiff_t)best); return hipSuccess; }
    }
    const hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) *cap = bytes;
    return e;
}
inline void rv_pinned_put(void *p, size_t cap) {
    if (!p) return;
    int dev = 0; (void)hipGetDevice(&dev);
    RvPools &P = rv_pools();
    {
        std::lock_guard<std::mutex> g(P.mu);
        if (cap <= ((size_t)64 << 20) && P.pinned_bytes + cap <= ((size_t)512 << 20) && P.pinned.size() < 256) { P.pinned.push_back({dev, p, cap}); P.pinned_bytes += cap; return; }
    }
    (void)hipHostFree(p);
}

struct HBuf {
    void  *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        // (kernels read and write pinned buffers directly -- picks, table staging: nothing may be in flight when one is replaced)
        if (p) { (void)hipDeviceSynchronize(); rv_pinned_put(p, cap); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 2 + 4096;
        RV_HIP(rv_pinned_get(want, &p, &cap));
        return 0;
    }
    void release() { if (p) rv_pinned_put(p, cap); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

// Scratch slots used by the primitives (one set per index handle, one stream).
struct Workspace {
    hipStream_t stream = nullptr;
    RvOptions opt;         // the owning handle's switches (rv_set_option)
    DBuf scan_tmp[4];      // block sums of the multi-level scan
    DBuf rs_hist;          // radix sort: per-block digit histograms
    DBuf rs_digits;        // radix sort: the next pass' digit of every key, a byte each
    DBuf misc[20];
    DBuf sa[32];           // SA-build scratch, kept between construct() calls
    HBuf hpin;             // pinned landing zone of rv_read_back
    hipEvent_t ev_rb = nullptr;
    // kernel-class timing of the handle that owns this workspace (RvProf, rv_index.h), for code that only sees the workspace
    void *prof_ctx = nullptr;
    int (*prof_begin_fn)(void *, hipStream_t, int, double) = nullptr;
    void (*prof_end_fn)(void *, hipStream_t, int) = nullptr;
    int prof_begin(int k, double bytes) { return prof_begin_fn ? prof_begin_fn(prof_ctx, stream, k, bytes) : -1; }
    void prof_end(int id) { if (prof_end_fn && id >= 0) prof_end_fn(prof_ctx, stream, id); }
    bool sa_in_use = false; // inside rv_build_sa / rv_build_lcp: the scratch below must stay
    size_t trim_sa() {      // releases the SA-build scratch; -> bytes given back
        size_t got = 0;
        for (auto &b : sa) { got += b.cap; b.release(); }
        return got;
    }
    void release() {
        hpin.release();
        if (ev_rb) { (void)hipEventDestroy(ev_rb); ev_rb = nullptr; }
        for (auto &b : scan_tmp) b.release();
        rs_hist.release(); rs_digits.release();
        for (auto &b : misc) b.release();
        for (auto &b : sa) b.release();
    }
};

std::mutex &rv_trim_mutex();      // rv_api.hip: guards the list of live handles and every change of sa_in_use
struct SaScratchInUse {
    Workspace &w; bool was;
    SaScratchInUse(Workspace &x) : w(x) { std::lock_guard<std::mutex> g(rv_trim_mutex()); was = w.sa_in_use; w.sa_in_use = true; }
    ~SaScratchInUse() { std::lock_guard<std::mutex> g(rv_trim_mutex()); w.sa_in_use = was; }
};

// ---- primitives (rv_prims.hip) ---------------------------------------------
// out[i] = sum_{j<i} in[j]  (in may alias out); n up to 2^40.
int rv_exclusive_sum_u32(Workspace &ws, const u32 *in, u32 *out, int64_t n);
int rv_exclusive_sum_u64(Workspace &ws, const u64 *in, u64 *out, int64_t n);
// out[i] = max_{j<=i} in[j]
int rv_inclusive_max_u32(Workspace &ws, const u32 *in, u32 *out, int64_t n);
int rv_inclusive_max_u64(Workspace &ws, const u64 *in, u64 *out, int64_t n);

// Stable LSD radix sort of (64-bit key, 32/64-bit value) pairs on key bits
// [bit_lo, bit_hi).  Ping-pongs between (k0,v0) and (k1,v1); *result_in_1
// tells where the sorted data ended up.
// radix passes rv_radix_sort_pairs makes over `nbits` key bits (8- or 10-bit digits: ws.opt.rs_bits)
int rv_radix_passes(const Workspace &ws, int nbits);
template <class V>
int rv_radix_sort_pairs(Workspace &ws, u64 *k0, V *v0, u64 *k1, V *v1, int64_t n,
                        int bit_lo, int bit_hi, int *result_in_1);

// A few bytes the host needs before it can go on (counts that size the next launch): device -> pinned memory -> dst, waiting on
// an event by polling.  A pageable destination is staged by the runtime and hipStreamSynchronize sleeps: ~35 us per read
// against ~12 us this way, five reads per construct().
int rv_read_back(Workspace &ws, void *dst, const void *dsrc, size_t bytes);

// copy of a small table from pinned host memory by a kernel (bytes rounded up to 16: both buffers must have that room)
int rv_h2d_copy(Workspace &ws, const void *pinned_src, void *dst, size_t bytes);

// ---- construct (rv_construct.hip) ------------------------------------------
struct RvSaStats {
    int    sigma, bits, k0;     // alphabet size, bits per symbol, symbols in the first key
    int    rounds;              // doubling rounds after the initial sort
    int64_t sorted_elems;       // sum of elements pushed through the radix sort
    int    radix_passes;
    int    diag_table;          // two samples: the hint and the twins' leaving follow piecewise diagonals from seeds (k_diag_bits_tab)
    int64_t far_pairs;          // tied pairs of partners ordered from the diagonal's marks (k_far_twins)
    int64_t lcp_list;           // ranks whose LCP / BWT came from the text after the doubling rounds (k_lcp_list)
};
// SA of T[0..n) (device pointers).  T must be readable up to n+15 (zero padded).
// LCP / BWT / d_maxlcp given: the build also leaves LCP (interface.c:97-114), the BWT bytes (side_sep as for rv_build_lcp) and
// the largest LCP whenever the first key and the text round finish the order (*fused_done = true); otherwise rv_build_lcp is due
int rv_build_sa(Workspace &ws, const uint8_t *T, int64_t n, sa_t *SA, RvSaStats *st,
                lcp_t *LCP = nullptr, uint8_t *BWT = nullptr, sa_t side_sep = 0, u32 *d_maxlcp = nullptr, bool *fused_done = nullptr,
                const int64_t *seps = nullptr, int nseps = 0);      // seps: every sample separator (nsep), for the diagonal hint
int rv_build_inverse(Workspace &ws, const sa_t *SA, sa_t *SAi, int64_t n);
// the same for an SA that came from a file: fails unless it is a permutation of 0..n-1
int rv_build_inverse_checked(Workspace &ws, const sa_t *SA, sa_t *SAi, int64_t n);
// LCP with the reference's stop characters (interface.c:97-114); also returns max LCP.
// by_rank: every rank from scratch (one thread per rank); otherwise text order with Kasai's carry (PHI scatter, no inverse needed)
// side_sep: text position of the first sample separator, nsep[0] (bit RV_BWT_SIDE of a BWT byte = the suffix starts behind
// it), or the largest sa_t for a single sample (the bit stays clear)
int rv_build_lcp(Workspace &ws, const uint8_t *T, const sa_t *SA, bool by_rank, lcp_t *LCP, int64_t n, u32 *d_maxlcp, uint8_t *BWT, sa_t side_sep);
// BWT only (when LCP came from a file)
int rv_build_bwt(Workspace &ws, const uint8_t *T, const sa_t *SA, int64_t n, uint8_t *BWT, sa_t side_sep);

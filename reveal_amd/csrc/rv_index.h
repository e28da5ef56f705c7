// rv_index.h -- the index handle behind the C ABI (include/reveal_amd.h).
#pragma once
#include "rv_common.h"
#include "rv_scan.h"
#include "../../include/reveal_amd.h"
#include <utility>

struct RvIntv { int64_t begin, end; };

// HIP-event profiler for kernel classes (bench.py's roofline figure).
struct RvProf {
    bool on = false;
    u32 mask = 0xFFFFFFFFu;           // kernel classes that are timed (an event pair costs the stream a few microseconds per span)
    struct Span { hipEvent_t a, b; int k; double bytes; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    int64_t launches[RV_K_COUNT] = {0};
    double ms[RV_K_COUNT] = {0};
    double bytes[RV_K_COUNT] = {0};
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    int begin(hipStream_t s, int k, double nbytes) {
        if (!on || !((mask >> k) & 1u)) return -1;
        Span sp; sp.a = get(); sp.b = get(); sp.k = k; sp.bytes = nbytes;
        (void)hipEventRecord(sp.a, s);
        spans.push_back(sp);
        return (int)spans.size() - 1;
    }
    void end(hipStream_t s, int id) { if (id >= 0) (void)hipEventRecord(spans[id].b, s); }
    // algorithmic bytes of a class whose spans are opened before the figure is known (bubble_sort: the leading children's sizes come with the commit)
    void credit(int k, double nbytes) { if (on && ((mask >> k) & 1u)) bytes[k] += nbytes; }
    // span whose two events are handed to hipExtLaunchKernelGGL: start / stop of that one kernel, no packets of their own
    int attach(int k, double nbytes, hipEvent_t *ea, hipEvent_t *eb) {
        *ea = *eb = nullptr;
        if (!on || !((mask >> k) & 1u)) return -1;
        Span sp; sp.a = get(); sp.b = get(); sp.k = k; sp.bytes = nbytes;
        spans.push_back(sp);
        *ea = sp.a; *eb = sp.b;
        return (int)spans.size() - 1;
    }
    void resolve() {   // caller has synchronised the stream
        for (auto &sp : spans) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) { ms[sp.k] += t; launches[sp.k]++; bytes[sp.k] += sp.bytes; }
            pool.push_back(sp.a); pool.push_back(sp.b);
        }
        spans.clear();
    }
    void reset() { resolve(); for (int k = 0; k < RV_K_COUNT; k++) { launches[k] = 0; ms[k] = 0; bytes[k] = 0; } }
    void release() { resolve(); for (auto e : pool) (void)hipEventDestroy(e); pool.clear(); }
};

// One sub-index of the current recursion level (host side bookkeeping; the
// ranks live in the level arrays at [off, off+n)).
struct RvSub {
    int64_t off = 0, n = 0;
    int depth = 0, nsamples = 0, parent = -1, kind = 0;
    std::vector<RvIntv> nodes;         // sorted by begin
    // scan result (host copies), CSR
    int64_t mum_first = 0, nmums = 0;  // range in the level-wide match arrays
    // decision
    bool has_split = false;
    u32 l = 0;
    std::vector<int64_t> sp;
    std::vector<RvIntv> lead, trail, match, rest;
};

// The assembled text on the host (n chars + NUL).  Pageable while small -- `reveal refine` builds an index per bubble of a few hundred
// bases --, page-locked (hipHostMalloc) from RV_TEXT_PIN_MIN bytes on: the text's way into HBM is then ONE copy by the DMA engine
// straight from where addsequence wrote it (through two pinned chunks filled by host threads it was 10 ms of four cores' memcpy per
// 500 MB; a stream of inputs keeps the cores for assembling the next text).  No zero-fill on growth: the bytes are written once.
struct HostText {
    static constexpr size_t RV_TEXT_PIN_MIN = (size_t)8 << 20;
    char *p = nullptr;
    size_t n = 0, cap = 0;
    bool pinned = false;
    HostText() = default;
    HostText(const HostText &) = delete;
    HostText &operator=(const HostText &o) {      // (rv_clone)
        if (this != &o && resize(o.n) == 0 && o.n) memcpy(p, o.p, o.n);
        return *this;
    }
    ~HostText() { release(); }
    void release() {
        if (p) { if (pinned) (void)hipHostFree(p); else free(p); }
        p = nullptr; n = cap = 0; pinned = false;
    }
    int reserve(size_t want) {
        if (want <= cap) return 0;
        size_t grow = want < RV_TEXT_PIN_MIN ? std::max<size_t>(want * 2, 256) : want + want / 16 + 4096;
        char *q = nullptr;
        bool pin = false;
        if (grow >= RV_TEXT_PIN_MIN) {
            void *v = nullptr;
            if (hipHostMalloc(&v, grow, hipHostMallocDefault) == hipSuccess) { q = (char *)v; pin = true; }
            else (void)hipGetLastError();      // (no page-locked memory to be had: pageable, the copy is staged by the runtime)
        }
        if (!q) q = (char *)malloc(grow);
        if (!q) { rv_set_error("out of host memory for %zu bytes of text", grow); return -1; }
        if (n) memcpy(q, p, n);
        if (p) { if (pinned) (void)hipHostFree(p); else free(p); }
        p = q; cap = grow; pinned = pin;
        return 0;
    }
    int resize(size_t m) { if (reserve(m) != 0) return -1; n = m; return 0; }
    void push_back(char c) { if (resize(n + 1) == 0) p[n - 1] = c; }
    char *data() { return p; }
    const char *data() const { return p; }
    size_t size() const { return n; }
    char &operator[](size_t k) { return p[k]; }
    const char &operator[](size_t k) const { return p[k]; }
};

struct rv_index {
    int device = 0;
    Workspace ws;
    RvProf prof;
    // ---- host text assembly (interface.c:18-95)
    HostText T;                        // n chars + NUL
    std::vector<int64_t> nsep;
    std::vector<int64_t> nsep_dev;     // what dNsep holds
    std::vector<RvIntv> nodes;
    int nsamples = 0;
    int64_t n = 0, nT = 0;
    int rc = 0;
    bool constructed = false, main_arrays_freed = false;
    bool sai_valid = false;            // dSAi holds the inverse of the main SA (made on demand: rv_need_sai)
    // ---- device state
    DBuf dT, dT0, dSA, dSAi, dLCP, dBWT, dNsep;   // dT0 = pristine text, dT = working copy (lower-cased by align)
    bool text_dirty = true;
    bool text_only = false;            // a worker of a divided alignment: text and shared inverse in HBM, no main SA / LCP
    HBuf hscan;                        // pinned staging for the scan records
    HBuf hupload;                      // two pinned chunks for the text's way into HBM (rv_upload)
    hipEvent_t ev_picks = nullptr;     // recorded behind the picker kernels: the host waits for this, not for the stream
    size_t scan_guess = 4096;
    DBuf ps[7];                        // rv_set_preselect on two samples: the longest matches of every sub-index chosen on the device (pair_topk, rv_api.hip)
    u32 maxlcp = 0;
    RvSaStats sa_stats{};
    // ---- scan results of the main index (getmums / getmultimums)
    std::vector<u32> m_l; std::vector<int64_t> m_a, m_b;
    std::vector<u32> mm_l; std::vector<int32_t> mm_n; std::vector<int64_t> mm_off, mm_pos; std::vector<uint16_t> mm_so;
    // ---- recursion state (rv_align.hip)
    struct Align *al = nullptr;
    struct RvBatchGroup *batch = nullptr;      // rv_batch_run: the group this handle's job meets in the middle of its anchor cascade (rv_cascade_multi.hip)
    bool batch_settled = false;                // ... it has arrived there, or has let the group know that it will not
    // ---- where the built-in run delivers its anchors (rv_set_result_buffers): the caller's arrays, page-locked while they are set
    struct ResultBufs {
        uint32_t *l = nullptr; int64_t *off = nullptr, *pos = nullptr;
        int64_t l_cap = 0, off_cap = 0, pos_cap = 0;
        bool direct = false;           // the last run's leaf / cascade anchors are already in them (rv_fetch_anchors copies the rest)
    } rb;
};

// scan of SA/LCP[0..m) -> host records in rank order (rv_api.hip)
// d_err (optional): device word copied into the scan header and returned through err_out (deferred error check of the previous commit)
// after_pick (optional): called once the picker kernels and the copy of the picks are queued, before the host waits for them; it
// may queue work that needs nothing but the picks on the device (the level's split with device-side decisions).  *redo is set
// when the picks had to be computed a second time (candidate list grown): whatever the hook queued saw incomplete picks
int rv_run_multi_pick(rv_index *h, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t m, int minl, int minn,
                      const int64_t *d_sub_start, const int *d_sub_want, int nsubs, const int *d_tile_sub, std::vector<u32> &pick_l, std::vector<sa_t> &pick_pos,
                      int (*after_pick)(rv_index *) = nullptr, bool *redo = nullptr);
int rv_run_pair_scan(rv_index *h, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t m, int minl, std::vector<RvPairRec> &out,
                     const u32 *d_err, u32 *err_out, const int64_t *d_sub_start, int nsubs,   // d_sub_start != NULL: only the best record per sub-index
                     int (*after_pick)(rv_index *), bool use_hook,
                     const int *d_tile_sub = nullptr,                                        // ... tile -> sub-index table of the level (optional, speeds the picker's look-ups)
                     const int64_t *d_presel_start = nullptr, int presel_subs = 0, int64_t presel = 0);   // rv_set_preselect: of every sub-index (starts on the device, presel_subs + 1 of them) only the `presel` longest records come back                        // ... and a hook called once the picker kernels are queued

// text, shared inverse and separators in HBM without an index (rv_api.hip); maxlcp = window size of bubble_sort
int rv_text_only(rv_index *h, u32 maxlcp);

// The inverse suffix array of the main index in HBM (interface.c:236-238).  Neither construct (LCP goes through PHI) nor the
// untraced built-in recursion (the split writes the windows of the shared inverse that bubble_sort reads) needs it, so it is
// made when somebody asks: the SAi getter, copy(), the detached-index steps, and align() with callbacks / tracing (rv_api.hip).
int rv_need_sai(rv_index *h);

// rv_align.hip
void rv_align_free(rv_index *h);

// rv_api.hip -- C ABI (include/reveal_amd.h): index lifetime, text assembly,
// construct, getters, top-level scans, profiling and primitive self-tests.
// The recursion lives in rv_align.hip.
#include "rv_index.h"
#include "rv_cascade.h"
#include <limits>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <limits.h>
#include <algorithm>
#include <thread>

static thread_local char g_err[1024] = "";

void rv_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// live handles, for rv_oom_trim (rv_common.h).  rv_trim_mutex() guards the list and every change of a handle's sa_in_use flag
// (SaScratchInUse takes it): while a trim holds it no handle can enter construct(), so the scratch of a handle found idle stays
// idle until it has been released.  Host threads driving one handle each (bench.py --jobs) meet here and nowhere else.
static std::vector<rv_index *> &live_handles() { static std::vector<rv_index *> v; return v; }
std::mutex &rv_trim_mutex() { static std::mutex *m = new std::mutex(); return *m; }
bool rv_oom_trim() {
    size_t got = 0;
    bool log = false;
    int caller_dev = 0;
    const bool have_dev = hipGetDevice(&caller_dev) == hipSuccess;
    {
        std::lock_guard<std::mutex> g(rv_trim_mutex());
        for (rv_index *h : live_handles()) {
            log = log || h->ws.opt.cascade_log != 0;
            if (h->ws.sa_in_use) continue;
            bool any = false;
            for (auto &b : h->ws.sa) any = any || b.p;
            if (!any) continue;
            (void)hipSetDevice(h->device);
            (void)hipDeviceSynchronize();          // nothing may still be reading what is released
            got += h->ws.trim_sa();
        }
    }
    if (have_dev) (void)hipSetDevice(caller_dev);      // the retry in DBuf::reserve allocates on the caller's device, not on the last handle's
    if (log && got) fprintf(stderr, "reveal_amd: device memory short, released %.1f GB of SA-build scratch\n", got / 1e9);
    return got > 0;
}

int g_rv_launch_trace = 0;

extern "C" {

const char *rv_last_error(void) { return g_err; }
int rv_abi_version(void) { return 2; }      /* 2: rv_picker_info writes six values (round 6) */
int rv_sa_bits(void) { return (int)sizeof(sa_t) * 8; }
int rv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

rv_index *rv_new(int device) {
    int nd = rv_device_count();
    if (nd <= 0) { rv_set_error("no HIP device visible: reveal_amd has no CPU fallback"); return nullptr; }
    if (device < 0 || device >= nd) { rv_set_error("device %d out of range (%d visible)", device, nd); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { rv_set_error("hipSetDevice(%d) failed", device); return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { rv_set_error("hipGetDeviceProperties failed"); return nullptr; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rv_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return nullptr;
    }
    rv_index *h = new rv_index();
    h->device = device;
    if (rv_stream_get(&h->ws.stream) != hipSuccess) {
        rv_set_error("hipStreamCreate failed");
        delete h;
        return nullptr;
    }
    h->T.push_back('\0');
    h->ws.prof_ctx = &h->prof;
    h->ws.prof_begin_fn = [](void *c, hipStream_t st, int k, double b) { return ((RvProf *)c)->begin(st, k, b); };
    h->ws.prof_end_fn = [](void *c, hipStream_t st, int id) { ((RvProf *)c)->end(st, id); };
    { std::lock_guard<std::mutex> g(rv_trim_mutex()); live_handles().push_back(h); }
    return h;
}

void rv_free(rv_index *h) {
    if (!h) return;
    { std::lock_guard<std::mutex> g(rv_trim_mutex()); auto &v = live_handles(); v.erase(std::remove(v.begin(), v.end(), h), v.end()); }
    (void)hipSetDevice(h->device);
    if (h->ws.stream) (void)hipStreamSynchronize(h->ws.stream);
    h->rb.direct = false;      // (nothing will fetch the last run's anchors any more: no copy to the staging buffer)
    (void)rv_set_result_buffers(h, nullptr, 0, nullptr, 0, nullptr, 0);
    rv_align_free(h);
    h->prof.release();
    h->dT.release(); h->dT0.release(); h->dSA.release(); h->dSAi.release(); h->dLCP.release(); h->dBWT.release(); h->dNsep.release();
    h->ws.release();
    h->hscan.release();
    for (auto &b : h->ps) b.release();
    h->hupload.release();
    if (h->ev_picks) { (void)hipEventDestroy(h->ev_picks); h->ev_picks = nullptr; }
    if (h->ws.stream) rv_stream_put(h->ws.stream);
    delete h;
}

/* switches of one handle (rv_common.h RV_OPTION_LIST): test hooks, diagnostics, A/B paths.  The library itself never looks at the
 * process environment. */
int rv_set_option(rv_index *h, const char *name, int64_t value) {
    if (!h || !name) { rv_set_error("rv_set_option: null argument"); return -1; }
    int64_t *f = h->ws.opt.find(name);
    if (!f) { rv_set_error("rv_set_option: unknown option %s", name); return -1; }
    *f = value;
    return 0;
}
/* diagnostics of the whole process, not of a handle: every kernel launch of the library prints its source line and is waited for, so that a
 * GPU memory fault (which aborts the process) names the kernel behind it.  Returns the previous setting. */
int rv_set_launch_trace(int on) {
    const int was = g_rv_launch_trace;
    g_rv_launch_trace = on != 0;
    return was;
}
int rv_get_option(rv_index *h, const char *name, int64_t *value) {
    if (!h || !name || !value) { rv_set_error("rv_get_option: null argument"); return -1; }
    const int64_t *f = h->ws.opt.find(name);
    if (!f) { rv_set_error("rv_get_option: unknown option %s", name); return -1; }
    *value = *f;
    return 0;
}
int rv_option_count(void) {
    int c = 0;
#define RV_X_(f, nm, def) c++;
    RV_OPTION_LIST(RV_X_)
#undef RV_X_
    return c;
}
const char *rv_option_name(int k) {
    static const char *const names[] = {
#define RV_X_(f, nm, def) nm,
        RV_OPTION_LIST(RV_X_)
#undef RV_X_
    };
    return k >= 0 && k < rv_option_count() ? names[k] : nullptr;
}

/* copy() (interface.c:432-470) of a main index: an independent handle with its own text (incl. the lower-case marks of
 * the working copy), SA, SAi, LCP.  The reference also resets depth / file names of the source and hands its SO to nobody
 * (self->SO = NULL, interface.c:462): not imitated. */
rv_index *rv_clone(rv_index *h) {
    if (!h->constructed || h->main_arrays_freed) { rv_set_error("Index not yet constructed."); return nullptr; }
    if (rv_need_sai(h)) return nullptr;
    rv_index *c = rv_new(h->device);
    if (!c) return nullptr;
    try { c->T = h->T; c->nsep = h->nsep; c->nodes = h->nodes; }
    catch (...) { rv_set_error("copy of the index: out of host memory"); rv_free(c); return nullptr; }
    if (c->T.size() != h->T.size()) { rv_free(c); return nullptr; }      // (HostText::resize failed and said why)
    c->nsamples = h->nsamples; c->n = h->n; c->nT = h->nT; c->rc = h->rc;
    c->maxlcp = h->maxlcp; c->sa_stats = h->sa_stats; c->text_dirty = true;
    c->ws.opt = h->ws.opt;
    const int64_t n = h->n;
    hipStream_t q = c->ws.stream;
    bool ok = rv_upload(c) == 0;
    ok = ok && c->dT.reserve((size_t)n + 64) == 0 && c->dSA.reserve((size_t)(n + 64) * sizeof(sa_t)) == 0 && c->dSAi.reserve((size_t)(n + 64) * sizeof(sa_t)) == 0
            && c->dLCP.reserve((size_t)(n + 64) * sizeof(lcp_t)) == 0 && c->dBWT.reserve((size_t)n + 64) == 0 && c->dNsep.reserve((h->nsep.size() + 1) * sizeof(sa_t)) == 0;
    ok = ok && hipStreamSynchronize(h->ws.stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->dT.p, h->dT.p, (size_t)n + 64, hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->dSA.p, h->dSA.p, (size_t)n * sizeof(sa_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->dSAi.p, h->dSAi.p, (size_t)n * sizeof(sa_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->dLCP.p, h->dLCP.p, (size_t)n * sizeof(lcp_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->dBWT.p, h->dBWT.p, (size_t)n, hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->dNsep.p, h->dNsep.p, (h->nsep.size() + 1) * sizeof(sa_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipStreamSynchronize(q) == hipSuccess;
    if (!ok) { rv_set_error("copy of the index failed"); rv_free(c); return nullptr; }
    c->nsep_dev = h->nsep_dev;
    c->constructed = true; c->sai_valid = true;
    return c;
}

/* interface.c:18-49 */
int rv_add_sample(rv_index *h) {
    if (h->nsamples > 0) h->nsep.push_back(h->n - 1);
    h->nsamples++;
    h->constructed = false; h->text_only = false;      /* the device arrays describe the text as it was: every getter and scan needs a new construct() */
    return 0;
}

/* interface.c:51-95 */
int rv_add_sequence(rv_index *h, const char *seq, int64_t len, int64_t *begin, int64_t *end) {
    if (len < 0 || !seq) { rv_set_error("addsequence: bad sequence"); return -1; }
#ifndef RV_SA64
    if ((uint64_t)h->n + (uint64_t)(len + 1) + 1 > (uint64_t)INT_MAX) {
        rv_set_error("Total amount of sequence too large, use \"reveal <subcommand> --64\" to use 64 bit suffix arrays instead.");
        return -1;
    }
#endif
    // one pass: copy into the (page-locked, when large) host text and test for non-ASCII bytes on the way -- bit 7 of the BWT bytes
    // carries the sample side (rv_common.h) --, block by block so that the test reads what the copy just left in the cache; four threads
    // above 4 MB (250 Mbp: 60 ms -> 12 ms; a stream of inputs is assembled while the GPU works on the one before)
    const int64_t s = h->n;
    if ((size_t)(h->n + len + 2) > h->T.cap) (void)hipSetDevice(h->device);      // (a large text is page-locked memory; 10^7 segments of a graph each find the room there)
    if (h->T.resize((size_t)(h->n + len + 2)) != 0) return -1;
    char *dst = h->T.data() + h->n;
    auto copy_check = [dst, seq](int64_t lo, int64_t hi) -> uint64_t {
        uint64_t acc = 0;
        const int64_t B = 32768;
        for (int64_t at = lo; at < hi; at += B) {
            const int64_t m = std::min(B, hi - at);
            memcpy(dst + at, seq + at, (size_t)m);
            int64_t k = 0;
            for (; k + 8 <= m; k += 8) { uint64_t w; memcpy(&w, dst + at + k, 8); acc |= w; }
            for (; k < m; k++) acc |= (uint64_t)(uint8_t)dst[at + k];
        }
        return acc;
    };
    uint64_t acc = 0;
    if (len >= ((int64_t)4 << 20)) {
        const int nt = 4;
        uint64_t part[nt] = {0, 0, 0, 0};
        std::thread th[nt];
        bool started[nt] = {false, false, false, false};
        for (int t = 1; t < nt; t++) {
            const int64_t lo = len / nt * t, hi = t + 1 == nt ? len : len / nt * (t + 1);
            // (no exception may leave through the C ABI: a thread the system will not start -- std::system_error at the thread limit -- has its
            //  part copied by this one)
            try { th[t] = std::thread([&, t, lo, hi]() { part[t] = copy_check(lo, hi); }); started[t] = true; }
            catch (...) { part[t] = copy_check(lo, hi); }
        }
        part[0] = copy_check(0, len / nt);
        for (int t = 1; t < nt; t++) if (started[t]) th[t].join();
        for (int t = 0; t < nt; t++) acc |= part[t];
    } else {
        acc = copy_check(0, len);
    }
    if (acc & 0x8080808080808080ull) {
        (void)h->T.resize((size_t)h->n + 1);
        h->T[(size_t)h->n] = '\0';
        rv_set_error("addsequence: the sequence contains non-ASCII bytes");
        return -1;
    }
    h->T[(size_t)(h->n + len)] = '$';
    h->T[(size_t)(h->n + len + 1)] = '\0';
    h->n += len + 1;
    h->text_dirty = true;
    h->constructed = false; h->text_only = false;      /* SA / LCP / BWT in HBM are sized for the old text (the getters would read past them) */
    if (begin) *begin = s;
    if (end) *end = h->n - 1;
    try { h->nodes.push_back(RvIntv{s, h->n - 1}); }
    catch (...) { rv_set_error("addsequence: out of host memory"); return -1; }
    return 0;
}

/* `count` sequences of the current sample at once: text = the sequences with a '$' behind each (total bytes), lens their lengths -- what count calls of
 * rv_add_sequence would leave, with one copy (the segments of a graph file: rv_graph_adopt) */
int rv_add_sequences(rv_index *h, const char *text, int64_t total, const int64_t *lens, int64_t count) {
    if (!h || !text || total < 0 || count < 0 || (count && !lens)) { rv_set_error("rv_add_sequences: bad arguments"); return -1; }
    int64_t sum = 0;
    for (int64_t k = 0; k < count; k++) { if (lens[k] < 0) { rv_set_error("rv_add_sequences: negative length"); return -1; } sum += lens[k] + 1; }
    if (sum != total) { rv_set_error("rv_add_sequences: lengths and text do not agree"); return -1; }
#ifndef RV_SA64
    if ((uint64_t)h->n + (uint64_t)total + 1 > (uint64_t)INT_MAX) {
        rv_set_error("Total amount of sequence too large, use \"reveal <subcommand> --64\" to use 64 bit suffix arrays instead.");
        return -1;
    }
#endif
    if ((size_t)(h->n + total + 1) > h->T.cap) (void)hipSetDevice(h->device);
    if (h->T.resize((size_t)(h->n + total + 1)) != 0) return -1;
    char *dst = h->T.data() + h->n;
    uint64_t acc = 0;
    {
        const int nt = total >= ((int64_t)4 << 20) ? 4 : 1;
        std::vector<uint64_t> part((size_t)nt, 0);
        auto work = [&](int t) {
            const int64_t lo = total / nt * t, hi = t + 1 == nt ? total : total / nt * (t + 1);
            memcpy(dst + lo, text + lo, (size_t)(hi - lo));
            uint64_t a = 0; int64_t k = lo;
            for (; k + 8 <= hi; k += 8) { uint64_t w; memcpy(&w, text + k, 8); a |= w; }
            for (; k < hi; k++) a |= (uint64_t)(uint8_t)text[k];
            part[(size_t)t] = a;
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) { try { th.emplace_back(work, t); } catch (...) { work(t); } }
        work(0);
        for (auto &x : th) x.join();
        for (uint64_t a : part) acc |= a;
    }
    if (acc & 0x8080808080808080ull) {
        (void)h->T.resize((size_t)h->n + 1);
        h->T[(size_t)h->n] = '\0';
        rv_set_error("addsequence: the sequence contains non-ASCII bytes");
        return -1;
    }
    try {
        h->nodes.reserve(h->nodes.size() + (size_t)count);
        int64_t at = h->n;
        for (int64_t k = 0; k < count; k++) {
            if (dst[at - h->n + lens[k]] != '$') { rv_set_error("rv_add_sequences: no '$' behind sequence %lld", (long long)k); (void)h->T.resize((size_t)h->n + 1); h->T[(size_t)h->n] = '\0'; return -1; }
            h->nodes.push_back(RvIntv{at, at + lens[k]});
            at += lens[k] + 1;
        }
    } catch (...) { rv_set_error("addsequence: out of host memory"); return -1; }
    h->T[(size_t)(h->n + total)] = '\0';
    h->n += total;
    h->text_dirty = true;
    h->constructed = false; h->text_only = false;
    return 0;
}

/* Forget the text and the samples; keep every allocation (host text, device arrays, SA-build scratch, streams): the handle is ready for
 * the next input's addsample / addsequence.  No counterpart in the reference, whose callers make a new index object per input -- which
 * here would allocate ~86 B per position of device memory again.  Result arrays set by rv_set_result_buffers stay set. */
int rv_reset(rv_index *h) {
    if (!h) { rv_set_error("rv_reset: null handle"); return -1; }
    RV_HIP(hipSetDevice(h->device));
    if (h->al) (void)rv_align_end(h);
    RV_HIP(hipStreamSynchronize(h->ws.stream));
    if (h->T.resize(1) != 0) return -1;
    h->T[0] = '\0';
    h->n = 0; h->nT = 0; h->nsamples = 0; h->rc = 0;
    h->nsep.clear(); h->nodes.clear();
    h->constructed = false; h->main_arrays_freed = false; h->sai_valid = false; h->text_only = false; h->text_dirty = true;
    h->maxlcp = 0;
    h->m_l.clear(); h->m_a.clear(); h->m_b.clear();
    h->mm_l.clear(); h->mm_n.clear(); h->mm_off.clear(); h->mm_pos.clear(); h->mm_so.clear();
    return 0;
}

/* Optional hint: the text will hold this many bytes (sequences + one separator each): the host buffer is made once instead of growing
 * sequence by sequence. */
int rv_reserve_text(rv_index *h, int64_t bytes) {
    if (!h || bytes < 0) { rv_set_error("rv_reserve_text: bad argument"); return -1; }
    (void)hipSetDevice(h->device);
    return h->T.reserve((size_t)bytes + 2);
}

int64_t rv_n(const rv_index *h) { return h->n; }
int rv_nsamples(const rv_index *h) { return h->nsamples; }
int rv_nnodes(const rv_index *h) { return (int)h->nodes.size(); }
int rv_node_list(const rv_index *h, int64_t *begin_end) {
    if (!h || !begin_end) { rv_set_error("rv_node_list: bad arguments"); return -1; }
    for (size_t k = 0; k < h->nodes.size(); k++) { begin_end[2 * k] = h->nodes[k].begin; begin_end[2 * k + 1] = h->nodes[k].end; }
    return 0;
}

/* interface.c:136-158 */
static char comp_of(char ch) {
    static const char up[] = "TVGHEFCDIJMLKNOPQYSAABWXRZ";
    const unsigned char c = (unsigned char)ch;
    if (c >= 'A' && c <= 'Z') return up[c - 'A'];
    if (c >= 'a' && c <= 'z') return (char)(up[c - 'a'] + 32);
    if (c == 96) return 64;
    return ch;
}
static void revcomp(char *T, int64_t n) {
    for (int64_t i = 0; i < n >> 1; ++i) {
        const char c0 = comp_of(T[i]), c1 = comp_of(T[n - 1 - i]);
        T[i] = c1; T[n - 1 - i] = c0;
    }
    if (n & 1) T[n >> 1] = comp_of(T[n >> 1]);
}

static int read_raw(const char *path, void *dst, size_t bytes) {
    FILE *f = fopen(path, "rb");
    if (!f) { rv_set_error("cannot open %s", path); return -1; }
    size_t got = fread(dst, 1, bytes, f);
    fclose(f);
    if (got != bytes) { rv_set_error("%s: expected %zu bytes, got %zu", path, bytes, got); return -1; }
    return 0;
}
static int write_raw(const char *path, const void *src, size_t bytes) {
    FILE *f = fopen(path, "wb");
    if (!f) { rv_set_error("cannot write %s", path); return -1; }
    const size_t put = fwrite(src, 1, bytes, f);
    const int rc = fclose(f);
    if (put != bytes || rc != 0) { rv_set_error("%s: short write (%zu of %zu bytes)", path, put, bytes); return -1; }
    return 0;
}

/* Text -> HBM (pristine copy, zero padded so word-wise readers may run past the
 * end).  construct() calls it when the host text changed; calling it up front
 * keeps the PCIe copy out of a timed construct(). */
int rv_upload(rv_index *h) {
    RV_HIP(hipSetDevice(h->device));
    if (!h->text_dirty) return 0;
    const int64_t n = h->n;
    RV_TRY(h->dT0.reserve((size_t)n + 64));
    hipStream_t q = h->ws.stream;
    RV_HIP(hipMemsetAsync(h->dT0.p, 0, 64, q));
    RV_HIP(hipMemsetAsync(h->dT0.as<uint8_t>() + n, 0, 64, q));
    const size_t CH = (size_t)32 << 20;
    if ((size_t)n < 2 * CH || h->T.pinned) {      // (a page-locked text: one copy by the DMA engine, no host thread touches it)
        RV_HIP(hipMemcpyAsync(h->dT0.p, h->T.data(), (size_t)n, hipMemcpyHostToDevice, q));
    } else {
        // A large text through two pinned chunks: several host threads fill one while the other is on the wire.  (Straight from the
        // pageable vector the runtime stages it itself, single-threaded: 500 MB took 38-52 ms, 10-13 GB/s.)
        HBuf &pin = h->hupload;
        RV_TRY(pin.reserve(2 * CH));
        struct Events { hipEvent_t e[2] = {nullptr, nullptr}; ~Events() { for (auto x : e) if (x) (void)hipEventDestroy(x); } } evs;      // (destroyed on every way out)
        hipEvent_t *ev = evs.e;
        for (int k = 0; k < 2; k++) RV_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        bool used[2] = {false, false};
        int slot = 0;
        for (size_t at = 0; at < (size_t)n; at += CH, slot ^= 1) {
            const size_t len = std::min(CH, (size_t)n - at);
            char *dst = pin.as<char>() + (size_t)slot * CH;
            if (used[slot]) RV_HIP(hipEventSynchronize(ev[slot]));
            const int nt = 4;
            std::thread th[nt];
            for (int t = 1; t < nt; t++) {
                const size_t lo = len / nt * t, hi = t + 1 == nt ? len : len / nt * (t + 1);
                th[t] = std::thread([=]() { memcpy(dst + lo, h->T.data() + at + lo, hi - lo); });
            }
            memcpy(dst, h->T.data() + at, len / nt);
            for (int t = 1; t < nt; t++) th[t].join();
            RV_HIP(hipMemcpyAsync(h->dT0.as<char>() + at, dst, len, hipMemcpyHostToDevice, q));
            RV_HIP(hipEventRecord(ev[slot], q));
            used[slot] = true;
        }
        RV_HIP(hipStreamSynchronize(q));
    }
    RV_HIP(hipStreamSynchronize(q));
    h->text_dirty = false;
    return 0;
}

int rv_upload_again(rv_index *h) {
    h->text_dirty = true;
    return rv_upload(h);
}

/* interface.c:160-291 */
int rv_construct(rv_index *h, int rc, const char *safile, const char *lcpfile, int cache) {
    RV_HIP(hipSetDevice(h->device));
    if (rc == 1) {
        if (h->nsep.empty()) { rv_set_error("construct(rc=1) needs at least two samples"); return -1; }
        h->rc = 1;
        revcomp(h->T.data() + h->nsep[0], h->n - h->nsep[0]);
        h->text_dirty = true;
    } else {
        h->rc = 0;
    }
    if (h->n == 0) { rv_set_error("No text to index."); return -1; }
    const int64_t n = h->n;
    // ranks, child sizes and scan records are carried as u32 inside the library (rv_scan.h, rv_split.h), whatever the width of
    // sa_t and however SA is obtained (built, or read from a file)
    if (n >= ((int64_t)1 << 32) - 2) { rv_set_error("index of %lld positions: this build carries ranks in 32 bits (n < 2^32 - 2)", (long long)n); return -1; }
    if (cache == 1) RV_TRY(write_raw(".reveal.t", h->T.data(), (size_t)n));
    if (h->al) (void)rv_align_end(h);        /* a new construct ends any recursion in flight; its device scratch is kept */
    h->nT = n;
    hipStream_t q = h->ws.stream;
    // working copy of the (HBM-resident) text: align() lower-cases it in place
    RV_TRY(rv_upload(h));
    RV_TRY(h->dT.reserve((size_t)n + 64));
    RV_HIP(hipMemcpyAsync(h->dT.p, h->dT0.p, (size_t)n + 64, hipMemcpyDeviceToDevice, q));
    RV_TRY(h->dSA.reserve((size_t)(n + 64) * sizeof(sa_t)));
    RV_TRY(h->dSAi.reserve((size_t)(n + 64) * sizeof(sa_t)));
    RV_TRY(h->dLCP.reserve((size_t)(n + 64) * sizeof(lcp_t)));
    RV_TRY(h->dBWT.reserve((size_t)n + 64));
    memset(&h->sa_stats, 0, sizeof h->sa_stats);
    RV_TRY(h->ws.misc[0].reserve(64));
    u32 *d_max = h->ws.misc[0].as<u32>();
    const sa_t side_sep = !h->nsep.empty() ? (sa_t)h->nsep[0] : std::numeric_limits<sa_t>::max();      // (RV_BWT_SIDE, rv_common.h; getmums tests against nsep[0] whatever the number of samples, reveal.c:73)
    const bool lcp_from_file = lcpfile && lcpfile[0];
    bool lcp_done = false;
    SaScratchInUse in_use(h->ws);
    if (!safile || !safile[0]) {
        int id = h->prof.begin(q, RV_K_SA_SORT, 5.0 * (double)n);
        if (lcp_from_file) RV_TRY(rv_build_sa(h->ws, h->dT.as<uint8_t>(), n, h->dSA.as<sa_t>(), &h->sa_stats));
        else RV_TRY(rv_build_sa(h->ws, h->dT.as<uint8_t>(), n, h->dSA.as<sa_t>(), &h->sa_stats, h->dLCP.as<lcp_t>(), h->dBWT.as<uint8_t>(), side_sep, d_max, &lcp_done,
                                 h->nsep.data(), (int)h->nsep.size()));
        h->prof.end(q, id);
    } else {
        std::vector<sa_t> tmp((size_t)n);
        RV_TRY(read_raw(safile, tmp.data(), (size_t)n * sizeof(sa_t)));
        RV_HIP(hipMemcpyAsync(h->dSA.p, tmp.data(), (size_t)n * sizeof(sa_t), hipMemcpyHostToDevice, q));
        RV_HIP(hipStreamSynchronize(q));
    }
    h->sai_valid = false;
    if (safile && safile[0]) { RV_TRY(rv_build_inverse_checked(h->ws, h->dSA.as<sa_t>(), h->dSAi.as<sa_t>(), n)); h->sai_valid = true; }      // (the check of an untrusted SA needs it anyway)
    if (!lcp_from_file) {
        if (!lcp_done) {       // (SA from a file, or an order the first key + text round did not finish)
            int id = h->prof.begin(q, RV_K_LCP, 13.0 * (double)n);
            RV_TRY(rv_build_lcp(h->ws, h->dT.as<uint8_t>(), h->dSA.as<sa_t>(), false, h->dLCP.as<lcp_t>(), n, d_max, h->dBWT.as<uint8_t>(), side_sep));
            h->prof.end(q, id);
        }
        RV_TRY(rv_read_back(h->ws, &h->maxlcp, d_max, 4));
    } else {
        std::vector<lcp_t> tmp((size_t)n);
        RV_TRY(read_raw(lcpfile, tmp.data(), (size_t)n * sizeof(lcp_t)));
        u32 mx = 0;
        for (int64_t i = 0; i < n; i++) if ((u32)tmp[(size_t)i] > mx) mx = (u32)tmp[(size_t)i];
        h->maxlcp = mx;
        RV_HIP(hipMemcpyAsync(h->dLCP.p, tmp.data(), (size_t)n * sizeof(lcp_t), hipMemcpyHostToDevice, q));
        RV_TRY(rv_build_bwt(h->ws, h->dT.as<uint8_t>(), h->dSA.as<sa_t>(), n, h->dBWT.as<uint8_t>(), side_sep));
        RV_HIP(hipStreamSynchronize(q));
    }
    if (cache == 1) {
        std::vector<sa_t> sa((size_t)n);
        std::vector<lcp_t> lc((size_t)n);
        RV_HIP(hipMemcpy(sa.data(), h->dSA.p, (size_t)n * sizeof(sa_t), hipMemcpyDeviceToHost));
        RV_HIP(hipMemcpy(lc.data(), h->dLCP.p, (size_t)n * sizeof(lcp_t), hipMemcpyDeviceToHost));
        RV_TRY(write_raw(".reveal.sa", sa.data(), (size_t)n * sizeof(sa_t)));
        RV_TRY(write_raw(".reveal.lcp", lc.data(), (size_t)n * sizeof(lcp_t)));
    }
    {   // sample separators for the multi-sample scans (SO is derived from them on the fly)
        if (h->nsep_dev != h->nsep || !h->dNsep.p) {      // (a synchronous copy: only when the separators changed)
            std::vector<sa_t> ns(h->nsep.size() + 1, 0);
            for (size_t k = 0; k < h->nsep.size(); k++) ns[k] = (sa_t)h->nsep[k];
            RV_TRY(h->dNsep.reserve(ns.size() * sizeof(sa_t)));
            RV_HIP(hipMemcpy(h->dNsep.p, ns.data(), ns.size() * sizeof(sa_t), hipMemcpyHostToDevice));
            h->nsep_dev = h->nsep;
        }
    }
    h->constructed = true;
    h->main_arrays_freed = false;
    h->text_only = false;
    return 0;
}

int64_t rv_get_array(rv_index *h, int which, void *out, int64_t cap) {
    (void)hipSetDevice(h->device);
    const int64_t n = h->n;
    switch (which) {
    case RV_T:
        if (cap < n) { rv_set_error("buffer too small"); return -1; }
        if (h->constructed || h->text_only) {
            if (hipMemcpy(out, h->dT.p, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) { rv_set_error("D2H failed"); return -1; }
        } else {
            memcpy(out, h->T.data(), (size_t)n);
        }
        return n;
    case RV_SA: case RV_SAI: case RV_LCP: {
        if (!h->constructed || (which != RV_SAI && h->main_arrays_freed)) { rv_set_error("Index not yet constructed."); return -2; }
        if (which == RV_SAI && rv_need_sai(h)) return -2;
        if (cap < h->nT && which == RV_SAI) { rv_set_error("buffer too small"); return -1; }
        if (cap < n && which != RV_SAI) { rv_set_error("buffer too small"); return -1; }
        const void *src = which == RV_SA ? h->dSA.p : which == RV_SAI ? h->dSAi.p : h->dLCP.p;
        const size_t esz = which == RV_LCP ? sizeof(lcp_t) : sizeof(sa_t);
        const int64_t cnt = which == RV_SAI ? h->nT : n;
        (void)hipStreamSynchronize(h->ws.stream);
        if (hipMemcpy(out, src, (size_t)cnt * esz, hipMemcpyDeviceToHost) != hipSuccess) { rv_set_error("D2H failed"); return -1; }
        return cnt;
    }
    case RV_SO: {   /* build_SO, interface.c:116-134 */
        if (!h->constructed || h->nsamples <= 2) { rv_set_error("SO not available."); return -2; }
        if (cap < h->nT) { rv_set_error("buffer too small"); return -1; }
        uint16_t *so = (uint16_t *)out;
        int64_t j = 0;
        for (int i = 0; i < h->nsamples; i++) {
            const int64_t hi = (i == h->nsamples - 1) ? h->nT - 1 : h->nsep[(size_t)i];
            for (; j <= hi; j++) so[j] = (uint16_t)i;
        }
        return h->nT;
    }
    case RV_NSEP:
        if (cap < (int64_t)h->nsep.size()) { rv_set_error("buffer too small"); return -1; }
        for (size_t k = 0; k < h->nsep.size(); k++) ((int64_t *)out)[k] = h->nsep[k];
        return (int64_t)h->nsep.size();
    case RV_NODES:
        if (cap < (int64_t)h->nodes.size() * 2) { rv_set_error("buffer too small"); return -1; }
        for (size_t k = 0; k < h->nodes.size(); k++) { ((int64_t *)out)[2 * k] = h->nodes[k].begin; ((int64_t *)out)[2 * k + 1] = h->nodes[k].end; }
        return (int64_t)h->nodes.size() * 2;
    }
    rv_set_error("unknown array id %d", which);
    return -1;
}

}  // extern "C"

int rv_need_sai(rv_index *h) {
    if (h->sai_valid) return 0;
    if (!h->constructed || h->main_arrays_freed) { rv_set_error("the inverse suffix array was not requested before align() consumed the main index"); return -1; }
    RV_HIP(hipSetDevice(h->device));
    RV_TRY(rv_build_inverse(h->ws, h->dSA.as<sa_t>(), h->dSAi.as<sa_t>(), h->n));
    h->sai_valid = true;
    return 0;
}

int rv_text_only(rv_index *h, u32 maxlcp) {
    RV_HIP(hipSetDevice(h->device));
    if (h->n == 0) { rv_set_error("No text to index."); return -1; }
    if (h->nsep.empty()) { rv_set_error("a worker index needs at least two samples"); return -1; }
    const int64_t n = h->n;
    hipStream_t q = h->ws.stream;
    h->rc = 0; h->nT = n;
    RV_TRY(rv_upload(h));
    RV_TRY(h->dT.reserve((size_t)n + 64));
    RV_HIP(hipMemcpyAsync(h->dT.p, h->dT0.p, (size_t)n + 64, hipMemcpyDeviceToDevice, q));
    RV_TRY(h->dSAi.reserve((size_t)(n + 64) * sizeof(sa_t)));      // written by every split in front of its cuts before bubble_sort reads there
    std::vector<sa_t> ns(h->nsep.size() + 1, 0);
    for (size_t k = 0; k < h->nsep.size(); k++) ns[k] = (sa_t)h->nsep[k];
    RV_TRY(h->dNsep.reserve(ns.size() * sizeof(sa_t)));
    RV_HIP(hipMemcpy(h->dNsep.p, ns.data(), ns.size() * sizeof(sa_t), hipMemcpyHostToDevice));
    h->nsep_dev = h->nsep;
    RV_HIP(hipStreamSynchronize(q));
    h->maxlcp = maxlcp;
    h->constructed = false; h->main_arrays_freed = true; h->text_only = true; h->sai_valid = false;
    return 0;
}

// ---------------------------------------------------------------------------
// pair scan driver: scan kernel -> scan of the tile counts -> compaction ->
// one D2H copy of the dense, rank-ordered records
// ---------------------------------------------------------------------------
// rv_set_preselect on two samples (reveal/schemes.py:240, 287-289): of every sub-index the `presel` longest matches, of equal
// lengths the later emitted, in emission order -- chosen on the device, so that the others never cross into host memory (2 x 10^6 records of
// 16 bytes at the top level of 2 x 250 Mbp).  The dense records are in rank order = emission order: (sub-index, length) keys with the record
// number as value through the library's stable radix sort; inside a sub-index the last `presel` of the sorted run stay (equal lengths keep their
// emission order in a stable sort: the later ones are the last); flags -> offsets -> the kept records in their old order.
namespace {
__global__ __launch_bounds__(256) void k_ps_keys(const RvPairRec *__restrict__ recs, u32 total, const int64_t *__restrict__ ss, int ns, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const int64_t r = (int64_t)recs[i].rank;
    int lo = 0, hi = ns;                                      // the last sub-index that starts at or before r (ss[0] = 0, ss[ns] = the level's size)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ss[mid] <= r) lo = mid; else hi = mid; }
    keys[i] = ((u64)(u32)lo << 32) | (u64)recs[i].l;
    vals[i] = i;
}
__global__ __launch_bounds__(256) void k_ps_mark(const u64 *__restrict__ skeys, const u32 *__restrict__ svals, u32 total, u32 presel, u32 *__restrict__ flag) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= total) return;
    const u64 next = ((skeys[p] >> 32) + 1ull) << 32;          // the first key of the next sub-index
    u32 lo = p, hi = total;                                   // skeys[lo] < next; the first position at or above `next` ends up in hi
    while (hi - lo > 1) { const u32 mid = lo + ((hi - lo) >> 1); if (skeys[mid] < next) lo = mid; else hi = mid; }
    flag[svals[p]] = (hi - p <= presel) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_ps_emit(const RvPairRec *__restrict__ recs, const u32 *__restrict__ flag, const u32 *__restrict__ off, u32 total, RvPairRec *__restrict__ out) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < total && flag[i]) out[off[i]] = recs[i];
}
}  // namespace
static int pair_topk(rv_index *h, const RvPairRec *recs, u32 total, const int64_t *d_ss, int ns, int64_t presel, std::vector<RvPairRec> &out) {
    hipStream_t q = h->ws.stream;
    DBuf *b = h->ps;
    RV_TRY(b[0].reserve((size_t)total * 8 + 64)); RV_TRY(b[1].reserve((size_t)total * 8 + 64));
    RV_TRY(b[2].reserve((size_t)total * 4 + 64)); RV_TRY(b[3].reserve((size_t)total * 4 + 64));
    RV_TRY(b[4].reserve(((size_t)total + 1) * 4 + 64)); RV_TRY(b[5].reserve(((size_t)total + 1) * 4 + 64));
    RV_TRY(b[6].reserve((size_t)total * sizeof(RvPairRec) + 64));
    const unsigned grid = (unsigned)ceil_div((int64_t)total, 256);
    hipLaunchKernelGGL(k_ps_keys, dim3(grid), dim3(256), 0, q, recs, total, d_ss, ns, b[0].as<u64>(), b[2].as<u32>());
    RV_LAUNCH_CHECK();
    int segbits = 1;
    while (segbits < 31 && ((int64_t)1 << segbits) < (int64_t)ns) segbits++;
    int in1 = 0;
    RV_TRY(rv_radix_sort_pairs<u32>(h->ws, b[0].as<u64>(), b[2].as<u32>(), b[1].as<u64>(), b[3].as<u32>(), (int64_t)total, 0, 32 + segbits, &in1));
    RV_HIP(hipMemsetAsync(b[4].p, 0, ((size_t)total + 1) * 4, q));
    const u32 keep = (u32)std::min<int64_t>(presel, 0xffffffffll);
    hipLaunchKernelGGL(k_ps_mark, dim3(grid), dim3(256), 0, q, (const u64 *)(in1 ? b[1].p : b[0].p), (const u32 *)(in1 ? b[3].p : b[2].p), total, keep, b[4].as<u32>());
    RV_LAUNCH_CHECK();
    RV_TRY(rv_exclusive_sum_u32(h->ws, b[4].as<u32>(), b[5].as<u32>(), (int64_t)total + 1));
    hipLaunchKernelGGL(k_ps_emit, dim3(grid), dim3(256), 0, q, recs, (const u32 *)b[4].p, (const u32 *)b[5].p, total, b[6].as<RvPairRec>());
    RV_LAUNCH_CHECK();
    u32 cnt = 0;
    RV_TRY(rv_read_back(h->ws, &cnt, b[5].as<u32>() + total, 4));
    if (cnt > total) { rv_set_error("preselect: more records kept than scanned"); return -1; }
    out.resize(cnt);
    if (cnt) RV_TRY(rv_read_back(h->ws, out.data(), b[6].p, (size_t)cnt * sizeof(RvPairRec)));
    return 0;
}

int rv_run_pair_scan(rv_index *h, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t m, int minl, std::vector<RvPairRec> &out,
                     const u32 *d_err, u32 *err_out, const int64_t *d_sub_start, int nsubs, int (*after_pick)(rv_index *), bool use_hook,
                     const int *d_tile_sub, const int64_t *d_presel_start, int presel_subs, int64_t presel) {
    out.clear();
    if (err_out) *err_out = 0;
    if (m <= 1) {
        if (d_err && err_out) { RV_HIP(hipMemcpyAsync(err_out, d_err, 4, hipMemcpyDeviceToHost, h->ws.stream)); RV_HIP(hipStreamSynchronize(h->ws.stream)); }
        return 0;
    }
    if (h->nsep.empty()) { rv_set_error("pairwise scan needs at least two samples"); return -1; }
    hipStream_t q = h->ws.stream;
    const int64_t ntile = ceil_div(m, RV_PAIR_TILE);
    DBuf &bcnt = h->ws.misc[14], &btab = h->ws.misc[2], &bslot = h->ws.misc[3], &bovf = h->ws.misc[4], &bout = h->ws.misc[5];   // misc[14]: the pair scan's own counter (self-resetting)
    if (bcnt.cap == 0) { RV_TRY(bcnt.reserve(64)); RV_HIP(hipMemsetAsync(bcnt.p, 0, 64, q)); }      // counters return to zero by themselves afterwards
    RV_TRY(btab.reserve((size_t)(ntile + 1) * 3 * sizeof(u32)));
    RV_TRY(bslot.reserve((size_t)ntile * RV_PAIR_SLOTS * sizeof(RvPairRec)));
    if (bovf.cap < 4096 * sizeof(RvPairRec)) RV_TRY(bovf.reserve(4096 * sizeof(RvPairRec)));
    if (bout.cap < 4096 * sizeof(RvPairRec)) RV_TRY(bout.reserve(sizeof(RvPairRec) * (size_t)std::max<int64_t>(4096, m / 64)));
    u32 *tilecnt = btab.as<u32>(), *tileovf = tilecnt + (ntile + 1), *tileoff = tileovf + (ntile + 1);
    for (int attempt = 0; attempt < 3; attempt++) {
        const size_t ocap = bout.cap / sizeof(RvPairRec) - RV_PAIR_HDR, vcap = bovf.cap / sizeof(RvPairRec);
        // picker tables; the picks go straight to pinned host memory (the kernels write them over PCIe: a few KB), so the level's
        // round trip needs no copy command, only the wait for the stream
        DBuf &bbest = h->ws.misc[12];
        RvPairRec *picks = nullptr;
        if (d_sub_start) {
            RV_TRY(bbest.reserve((size_t)nsubs * 8));
            RV_TRY(h->hscan.reserve((size_t)(nsubs + RV_PAIR_HDR) * sizeof(RvPairRec)));
            picks = h->hscan.as<RvPairRec>();
        }
        hipEvent_t ev_a, ev_b;      /* 8 B/rank (12 B in the 64-bit build); the BWT byte is not counted */
        (void)h->prof.attach(RV_K_SCAN_PAIR, (double)m * 8.0, &ev_a, &ev_b);      // 8 B per rank (a 4-byte suffix + a 4-byte LCP value), also for the 64-bit library -- the kernel reads suffixes only where a match may start
        RV_TRY(rv_scan_pair_launch(h->ws, SA, LCP, m, BWT, (sa_t)h->nsep[0], minl, bslot.as<RvPairRec>(), bovf.as<RvPairRec>(),
                                   (u32)std::min<size_t>(vcap, 0xffffffffu), bcnt.as<u32>(), tilecnt, tileovf,
                                   bbest.as<unsigned long long>(), picks, d_sub_start ? nsubs : 0, ev_a, ev_b));
        if (d_sub_start) {
            // the built-in picker only wants the best record of each sub-index: pick on the device straight from the slots
            RV_TRY(rv_pick_slots_launch(h->ws, bslot.as<RvPairRec>(), bovf.as<RvPairRec>(), (u32)std::min<size_t>(vcap, 0xffffffffu), tilecnt, tileovf, ntile,
                                        d_sub_start, nsubs, bbest.as<unsigned long long>(), picks, bcnt.as<u32>(), d_err, d_tile_sub, ceil_div(m, RV_TSUB_TILE)));
            // The one host round trip of a level.  The host waits for the picks only (an event behind the picker kernels), not for
            // the stream: work that needs nothing but the picks on the device (the level's split, rv_decide.hip) is queued first
            // and runs while the host works.  Spinning instead of sleeping in a synchronize: its wake-up costs tens of
            // microseconds, 33 times per alignment.
            if (!h->ev_picks) RV_HIP(hipEventCreateWithFlags(&h->ev_picks, hipEventDisableTiming));
            RV_HIP(hipEventRecord(h->ev_picks, q));
            if (use_hook && after_pick) RV_TRY(after_pick(h));
            if (h->ws.opt.sync_block) { RV_HIP(hipEventSynchronize(h->ev_picks)); }
            else { const hipError_t qe = rv_event_wait(h->ev_picks); if (qe != hipSuccess) { rv_set_error("stream: %s", hipGetErrorString(qe)); return -1; } }
            const u32 *hdr = h->hscan.as<u32>();
            const u32 novf = hdr[1];
            if (err_out) *err_out = hdr[2];
            if (novf <= vcap) {
                const RvPairRec *src = h->hscan.as<RvPairRec>() + RV_PAIR_HDR;
                for (int s2 = 0; s2 < nsubs; s2++) if (src[s2].rank != 0xFFFFFFFFu) out.push_back(src[s2]);
                return 0;
            }
            RV_TRY(bovf.reserve((size_t)novf * sizeof(RvPairRec)));
            continue;
        }
        RV_TRY(rv_exclusive_sum_u32(h->ws, tilecnt, tileoff, ntile + 1));
        RV_TRY(rv_pair_compact_launch(h->ws, bslot.as<RvPairRec>(), bovf.as<RvPairRec>(), tilecnt, tileovf, tileoff, ntile, bout.as<RvPairRec>(),
                                      (u32)std::min<size_t>(ocap, 0xffffffffu), bcnt.as<u32>(), d_err, (u32)std::min<size_t>(vcap, 0xffffffffu)));
        // one copy: header + as many records as the previous scan produced (record counts shrink level by level).  With a pre-selection cap
        // the copy brings RV_PRESEL_DEV_MIN records at most: a level with more is capped on the device (pair_topk) and its records never cross
        // into host memory -- the header that decides this arrives with the same copy (it used to be a round trip of its own per level)
        const bool presel_on = d_presel_start && presel > 0 && presel_subs > 0;
        size_t guess = std::min<size_t>(ocap, h->scan_guess);
        if (presel_on) guess = std::min<size_t>(guess, (size_t)std::max<int64_t>(h->ws.opt.presel_dev_min, 1));
        RV_TRY(h->hscan.reserve((guess + RV_PAIR_HDR) * sizeof(RvPairRec)));
        RV_HIP(hipMemcpyAsync(h->hscan.p, bout.p, (guess + RV_PAIR_HDR) * sizeof(RvPairRec), hipMemcpyDeviceToHost, q));
        RV_HIP(hipStreamSynchronize(q));
        const u32 *hdr = h->hscan.as<u32>();
        const u32 total = hdr[0], novf = hdr[1];
        if (err_out) *err_out = hdr[2];
        if (presel_on && total <= ocap && novf <= vcap && (int64_t)total >= h->ws.opt.presel_dev_min && (int64_t)total > presel) {
            RV_TRY(pair_topk(h, bout.as<RvPairRec>() + RV_PAIR_HDR, total, d_presel_start, presel_subs, presel, out));
            h->scan_guess = (size_t)total + total / 16 + 64;
            return 0;
        }
        if (total <= ocap && novf <= vcap) {
            out.resize(total);
            const RvPairRec *src = h->hscan.as<RvPairRec>() + RV_PAIR_HDR;
            const size_t have = std::min<size_t>(total, guess);
            if (have) memcpy(out.data(), src, have * sizeof(RvPairRec));
            if (total > have) RV_HIP(hipMemcpy(out.data() + have, bout.as<RvPairRec>() + RV_PAIR_HDR + have, (total - have) * sizeof(RvPairRec), hipMemcpyDeviceToHost));
            h->scan_guess = (size_t)total + total / 16 + 64;
            return 0;
        }
        if (novf > vcap) RV_TRY(bovf.reserve((size_t)novf * sizeof(RvPairRec)));
        if (total > ocap) RV_TRY(bout.reserve(((size_t)total + RV_PAIR_HDR) * sizeof(RvPairRec)));
    }
    rv_set_error("pair scan: output buffer sizing failed");
    return -1;
}

// multi-MUM scan driver (same tile-ordered merge as the pair scan)
int rv_run_multi_scan(rv_index *h, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t m, int minl, int minn, int mems,
                      std::vector<u32> &l, std::vector<int32_t> &n, std::vector<int64_t> &off, std::vector<uint16_t> &so,
                      std::vector<int64_t> &pos, std::vector<int64_t> *ub_out,
                      const int64_t *d_sub_start, const int *d_sub_want, int nsubs) {
    l.clear(); n.clear(); off.assign(1, 0); so.clear(); pos.clear();
    if (ub_out) ub_out->clear();
    if (h->nsamples < 2) { rv_set_error("multi scan needs at least two samples"); return -1; }
    if (mems) {
        /* reveal.c:292-434, a stack machine per run of LCP values of minl and more (rv_mems.hip).  The sample census of an interval is a 64-bit mask there. */
        if (h->nsamples > 64) { rv_set_error("getmultimems: more than 64 samples not supported yet"); return -1; }
        if (m <= 1) return 0;
        hipStream_t q = h->ws.stream;
        DBuf &brec = h->ws.misc[13], &bso = h->ws.misc[6], &bpos = h->ws.misc[7];
        size_t rcap = (size_t)std::max<int64_t>(4096, m / 16), mcap = (size_t)std::max<int64_t>(8192, m / 2);
        for (int attempt = 0; attempt < 2; attempt++) {
            RV_TRY(brec.reserve(64 + rcap * 16 + 64));
            RV_TRY(bso.reserve(mcap * 2 + 64));
            RV_TRY(bpos.reserve(mcap * sizeof(sa_t) + 64));
            uint8_t *rb = brec.as<uint8_t>();
            unsigned long long *d_out = (unsigned long long *)rb;
            int64_t *rec_first = (int64_t *)(rb + 64); u32 *rec_l = (u32 *)(rec_first + rcap); int32_t *rec_c = (int32_t *)(rec_l + rcap);
            int id = h->prof.begin(q, RV_K_SCAN_MULTI, (double)m * (sizeof(sa_t) + sizeof(lcp_t)));
            RV_TRY(rv_multimems_launch(h->ws, SA, LCP, BWT, m, h->dNsep.as<sa_t>(), h->nsamples, minl, minn, (u32)std::min<int64_t>(h->maxlcp, 0xfffffff0ll), rec_l, rec_c, rec_first,
                                       bso.as<uint16_t>(), bpos.as<sa_t>(), rcap, mcap, d_out));
            h->prof.end(q, id);
            unsigned long long res[3] = {0, 0, 0};
            RV_TRY(rv_read_back(h->ws, res, d_out, sizeof res));
            if (res[2]) { rv_set_error("getmultimems: interval stack deeper than the largest LCP value"); return -1; }
            if (res[0] <= rcap && res[1] <= mcap) {
                const size_t nr = (size_t)res[0], nm = (size_t)res[1];
                std::vector<int64_t> first(nr); std::vector<int32_t> cc(nr); std::vector<uint16_t> rso(nm); std::vector<sa_t> rpos(nm);
                l.resize(nr);
                if (nr) {
                    RV_HIP(hipMemcpy(first.data(), rec_first, nr * 8, hipMemcpyDeviceToHost));
                    RV_HIP(hipMemcpy(l.data(), rec_l, nr * 4, hipMemcpyDeviceToHost));
                    RV_HIP(hipMemcpy(cc.data(), rec_c, nr * 4, hipMemcpyDeviceToHost));
                }
                if (nm) {
                    RV_HIP(hipMemcpy(rso.data(), bso.p, nm * 2, hipMemcpyDeviceToHost));
                    RV_HIP(hipMemcpy(rpos.data(), bpos.p, nm * sizeof(sa_t), hipMemcpyDeviceToHost));
                }
                n.assign(cc.begin(), cc.end());
                off.assign(first.begin(), first.end()); off.push_back((int64_t)nm);
                so.assign(rso.begin(), rso.end());
                pos.assign(rpos.begin(), rpos.end());
                return 0;
            }
            rcap = std::max<size_t>(rcap, (size_t)res[0]); mcap = std::max<size_t>(mcap, (size_t)res[1]);
        }
        rv_set_error("getmultimems: output buffer sizing failed");
        return -1;
    }
    if (m <= 1) return 0;
    hipStream_t q = h->ws.stream;
    const int64_t ntile = ceil_div(m, RV_MULTI_TILE);
    DBuf &bcnt = h->ws.misc[1], &btab = h->ws.misc[2], &brec = h->ws.misc[8], &bso = h->ws.misc[6], &bpos = h->ws.misc[7];
    RV_TRY(bcnt.reserve(64));
    RV_TRY(btab.reserve((size_t)ntile * sizeof(uint4)));
    if (brec.cap < 4096 * sizeof(RvMultiRec)) RV_TRY(brec.reserve(sizeof(RvMultiRec) * (size_t)std::max<int64_t>(4096, m / 32)));
    if (bso.cap < 8192 * 2) { RV_TRY(bso.reserve(2 * (size_t)std::max<int64_t>(8192, m / 8))); }
    RV_TRY(bpos.reserve((bso.cap / 2) * sizeof(sa_t)));
    std::vector<uint4> tab((size_t)ntile);
    for (int attempt = 0; attempt < 2; attempt++) {
        const size_t rcap = brec.cap / sizeof(RvMultiRec), mcap = std::min(bso.cap / 2, bpos.cap / sizeof(sa_t));
        RV_HIP(hipMemsetAsync(bcnt.p, 0, 8, q));
        int id = h->prof.begin(q, RV_K_SCAN_MULTI, (double)m * (sizeof(sa_t) + sizeof(lcp_t)));
        RV_TRY(rv_scan_multi_launch(h->ws, SA, LCP, m, BWT, h->dNsep.as<sa_t>(), h->nsamples, minl, minn,
                                    brec.as<RvMultiRec>(), bso.as<uint16_t>(), bpos.as<sa_t>(), (u32)std::min<size_t>(rcap, 0xffffffffu),
                                    (u32)std::min<size_t>(mcap, 0xffffffffu), bcnt.as<u32>(), btab.as<uint4>(), d_sub_start, d_sub_want, nsubs));
        h->prof.end(q, id);
        u32 tot[2] = {0, 0};
        RV_HIP(hipMemcpyAsync(tot, bcnt.p, 8, hipMemcpyDeviceToHost, q));
        RV_HIP(hipMemcpyAsync(tab.data(), btab.p, (size_t)ntile * sizeof(uint4), hipMemcpyDeviceToHost, q));
        RV_HIP(hipStreamSynchronize(q));
        if (tot[0] <= rcap && tot[1] <= mcap) {
            std::vector<RvMultiRec> rr(tot[0]);
            std::vector<uint16_t> rso(tot[1]);
            std::vector<sa_t> rpos(tot[1]);
            if (tot[0]) RV_HIP(hipMemcpy(rr.data(), brec.p, (size_t)tot[0] * sizeof(RvMultiRec), hipMemcpyDeviceToHost));
            if (tot[1]) {
                RV_HIP(hipMemcpy(rso.data(), bso.p, (size_t)tot[1] * 2, hipMemcpyDeviceToHost));
                RV_HIP(hipMemcpy(rpos.data(), bpos.p, (size_t)tot[1] * sizeof(sa_t), hipMemcpyDeviceToHost));
            }
            l.reserve(tot[0]); n.reserve(tot[0]); off.reserve(tot[0] + 1); so.reserve(tot[1]); pos.reserve(tot[1]);
            for (int64_t t = 0; t < ntile; t++) {
                const uint4 e = tab[(size_t)t];
                for (u32 k = 0; k < e.y; k++) {
                    const RvMultiRec &r = rr[e.x + k];
                    l.push_back(r.l); n.push_back((int32_t)r.n);
                    if (ub_out) ub_out->push_back((int64_t)r.ub);
                }
                for (u32 k = 0; k < e.w; k++) { so.push_back(rso[e.z + k]); pos.push_back((int64_t)rpos[e.z + k]); }
            }
            int64_t acc = 0;
            for (size_t k = 0; k < n.size(); k++) { acc += n[k]; off.push_back(acc); }
            return 0;
        }
        if (tot[0] > rcap) RV_TRY(brec.reserve((size_t)tot[0] * sizeof(RvMultiRec)));
        if (tot[1] > mcap) { RV_TRY(bso.reserve((size_t)tot[1] * 2)); RV_TRY(bpos.reserve((size_t)tot[1] * sizeof(sa_t))); }
    }
    rv_set_error("multi scan: output buffer sizing failed");
    return -1;
}

// built-in picker of the untraced recursion, more than two samples: one (l, members) per sub-index, picked on the device
int rv_run_multi_pick(rv_index *h, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t m, int minl, int minn,
                      const int64_t *d_sub_start, const int *d_sub_want, int nsubs, const int *d_tile_sub, std::vector<u32> &pick_l, std::vector<sa_t> &pick_pos,
                      int (*after_pick)(rv_index *), bool *redo) {
    pick_l.assign((size_t)nsubs, 0); pick_pos.clear();
    if (m <= 1 || nsubs <= 0) return 0;
    hipStream_t q = h->ws.stream;
    const int W = h->nsamples;
    DBuf &bbest = h->ws.misc[12], &bl = h->ws.misc[13], &bpos = h->ws.misc[7], &bcand = h->ws.misc[8], &bcnt = h->ws.misc[1];
    RV_TRY(bbest.reserve((size_t)nsubs * 8)); RV_TRY(bl.reserve((size_t)nsubs * 4)); RV_TRY(bpos.reserve((size_t)nsubs * W * sizeof(sa_t)));
    RV_TRY(bcnt.reserve((RV_MULTI_REGIONS * 64 + 16) * 4));
    if (bcand.cap < 65536 * RV_MULTI_CAND_BYTES) RV_TRY(bcand.reserve((size_t)std::max<int64_t>(65536, m / 64) * RV_MULTI_CAND_BYTES));
    for (int attempt = 0; attempt < 2; attempt++) {
        const size_t ccap = bcand.cap / RV_MULTI_CAND_BYTES;
        hipEvent_t ev_a, ev_b;      // the streaming kernel alone, as for the pair scan (8 B per rank)
        (void)h->prof.attach(RV_K_SCAN_MULTI, (double)m * 8.0, &ev_a, &ev_b);
        RV_TRY(rv_multi_pick_launch(h->ws, SA, LCP, m, BWT, h->dNsep.as<sa_t>(), W, minl, minn, d_sub_start, d_sub_want, nsubs, d_tile_sub,
                                    bbest.as<unsigned long long>(), bl.as<u32>(), bpos.as<sa_t>(), (RvMultiCand *)bcand.p,
                                    (u32)std::min<size_t>(ccap, 0xffffffffu), bcnt.as<u32>(), ev_a, ev_b));
        u32 ncand = 0;
        pick_pos.resize((size_t)nsubs * W);
        {   // the level's one round trip: three arrays into pinned memory, one polled event (pageable destinations are staged copy by copy)
            const size_t b1 = (size_t)nsubs * 4, b2 = (size_t)nsubs * W * sizeof(sa_t), o1 = 64, o2 = (o1 + b1 + 63) & ~(size_t)63;
            RV_TRY(h->ws.hpin.reserve(o2 + b2 + 64));
            uint8_t *hp = h->ws.hpin.as<uint8_t>();
            if (!h->ws.ev_rb) RV_HIP(hipEventCreateWithFlags(&h->ws.ev_rb, hipEventDisableTiming));
            RV_HIP(hipMemcpyAsync(hp, bcnt.as<u32>() + RV_MULTI_REGIONS * 64, 4, hipMemcpyDeviceToHost, q));      // the fullest region of the list
            RV_HIP(hipMemcpyAsync(hp + o1, bl.p, b1, hipMemcpyDeviceToHost, q));
            RV_HIP(hipMemcpyAsync(hp + o2, bpos.p, b2, hipMemcpyDeviceToHost, q));
            RV_HIP(hipEventRecord(h->ws.ev_rb, q));
            if (attempt == 0 && after_pick) RV_TRY(after_pick(h));      // (queued behind the copies: the host gets the picks first)
            const hipError_t qe = rv_event_wait(h->ws.ev_rb);
            if (qe != hipSuccess) { rv_set_error("stream: %s", hipGetErrorString(qe)); return -1; }
            memcpy(&ncand, hp, 4); memcpy(pick_l.data(), hp + o1, b1); memcpy(pick_pos.data(), hp + o2, b2);
        }
        if ((size_t)ncand <= ccap / RV_MULTI_REGIONS) return 0;
        if (redo) *redo = true;
        RV_TRY(bcand.reserve(((size_t)ncand + ncand / 2 + 64) * RV_MULTI_REGIONS * RV_MULTI_CAND_BYTES));
    }
    rv_set_error("multi picker: candidate buffer sizing failed");
    return -1;
}

extern "C" {

/* reveal.c:436-580 / 292-434 */
int64_t rv_getmultimums(rv_index *h, int minlength, int minn, int mems, int64_t *members) {
    if (!h->constructed || h->main_arrays_freed) { rv_set_error("Index not yet constructed."); return -2; }
    if (h->nsamples <= 2) {
        /* two samples, getmultimems: every qualifying interval counts one "sample" (the flag_so index trick, reveal.c:268) and
         * leaves through `continue` when minn >= 2: an empty list.  (With minn < 2 the reference dereferences the missing SO.) */
        if (mems && h->nsamples == 2 && minn >= 2) { h->mm_l.clear(); h->mm_n.clear(); h->mm_off.assign(1, 0); h->mm_so.clear(); h->mm_pos.clear(); if (members) *members = 0; return 0; }
        rv_set_error("getmultimums needs more than two samples (SO not available)"); return -1;
    }
    (void)hipSetDevice(h->device);
    if (rv_run_multi_scan(h, h->dSA.as<sa_t>(), h->dLCP.as<lcp_t>(), h->dBWT.as<uint8_t>(), h->n, minlength, minn, mems, h->mm_l, h->mm_n, h->mm_off, h->mm_so, h->mm_pos, nullptr, nullptr, nullptr, 0)) return -1;
    if (members) *members = (int64_t)h->mm_pos.size();
    return (int64_t)h->mm_l.size();
}

int rv_fetch_multi(rv_index *h, uint32_t *l, int32_t *n, int64_t *off, uint16_t *so, int64_t *pos) {
    memcpy(l, h->mm_l.data(), h->mm_l.size() * 4);
    memcpy(n, h->mm_n.data(), h->mm_n.size() * 4);
    memcpy(off, h->mm_off.data(), h->mm_off.size() * 8);
    memcpy(so, h->mm_so.data(), h->mm_so.size() * 2);
    memcpy(pos, h->mm_pos.data(), h->mm_pos.size() * 8);
    return 0;
}

/* reveal.c:55-116 */
int64_t rv_getmums(rv_index *h, int minl) {
    if (!h->constructed || h->main_arrays_freed) { rv_set_error("Index not yet constructed."); return -2; }
    (void)hipSetDevice(h->device);
    std::vector<RvPairRec> recs;
    if (rv_run_pair_scan(h, h->dSA.as<sa_t>(), h->dLCP.as<lcp_t>(), h->dBWT.as<uint8_t>(), h->n, minl, recs, nullptr, nullptr, nullptr, 0, nullptr, false) != 0) return -1;
    h->m_l.resize(recs.size()); h->m_a.resize(recs.size()); h->m_b.resize(recs.size());
    for (size_t k = 0; k < recs.size(); k++) {
        int64_t b = recs[k].b;
        if (h->rc == 1) b = h->nsep[0] + (h->nT - b - (int64_t)recs[k].l);     /* reveal.c:98-100 */
        h->m_l[k] = recs[k].l; h->m_a[k] = recs[k].a; h->m_b[k] = b;
    }
    return (int64_t)recs.size();
}

int rv_fetch_mums(rv_index *h, uint32_t *l, int64_t *a, int64_t *b, int64_t cap) {
    if (cap < (int64_t)h->m_l.size()) { rv_set_error("buffer too small"); return -1; }
    for (size_t k = 0; k < h->m_l.size(); k++) { l[k] = h->m_l[k]; a[k] = h->m_a[k]; b[k] = h->m_b[k]; }
    return 0;
}

int rv_prof_enable(rv_index *h, int on) {      /* 0 off; 1 every class; otherwise bit k+1 selects class k */
    h->prof.on = on != 0;
    h->prof.mask = (on & 1) ? 0xFFFFFFFFu : ((u32)on >> 1);
    return 0;
}
int rv_prof_reset(rv_index *h) { (void)hipStreamSynchronize(h->ws.stream); h->prof.reset(); return 0; }
int rv_prof_get(rv_index *h, int k, int64_t *launches, double *ms, double *bytes) {
    if (k < 0 || k >= RV_K_COUNT) { rv_set_error("bad kernel id"); return -1; }
    (void)hipStreamSynchronize(h->ws.stream);
    h->prof.resolve();
    if (launches) *launches = h->prof.launches[k];
    if (ms) *ms = h->prof.ms[k];
    if (bytes) *bytes = h->prof.bytes[k];
    return 0;
}
int rv_sa_stats(rv_index *h, int *sigma, int *bits, int *k0, int *rounds, int64_t *sorted_elems, int *radix_passes) {
    if (sigma) *sigma = h->sa_stats.sigma;
    if (bits) *bits = h->sa_stats.bits;
    if (k0) *k0 = h->sa_stats.k0;
    if (rounds) *rounds = h->sa_stats.rounds;
    if (sorted_elems) *sorted_elems = h->sa_stats.sorted_elems;
    if (radix_passes) *radix_passes = h->sa_stats.radix_passes;
    return 0;
}
/* 1: the last construct() of two samples followed piecewise diagonals from seeds (the samples had left their fixed diagonal: indels) */
int rv_sa_diag_table(rv_index *h) { return h->sa_stats.diag_table; }
/* what finished the suffixes the first key and the text round left tied in the last construct(): out[0] tied pairs of partners ordered from the
 * diagonal's marks (k_far_twins), out[1] ranks whose LCP / BWT came from the text after the doubling rounds (k_lcp_list) */
int rv_sa_tail(rv_index *h, int64_t *out) { if (!h || !out) return -1; out[0] = h->sa_stats.far_pairs; out[1] = h->sa_stats.lcp_list; return 0; }

/* ---- the node's practical HBM ceiling (SURVEY 8(d) "Roofline": measured copy-kernel bandwidth beside the 8 TB/s spec) ---- */
}  // extern "C"
namespace {
typedef int bw_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_bw_read(const bw_v4 *__restrict__ a, int64_t n16, unsigned *sink) {
    bw_v4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) acc ^= __builtin_nontemporal_load(a + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) sink[0] = 1;      // (never true for the fill pattern: keeps the loads alive)
}
__global__ __launch_bounds__(256) void k_bw_copy(const bw_v4 *__restrict__ a, bw_v4 *__restrict__ b, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
}  // namespace
extern "C" {
/* ---- a batch of independent alignments (include/reveal_amd.h "rv_batch"; the reference's counterpart is a shell script of independent `reveal rem`
 * commands, reveal/align.py:27-54) ---------------------------------------------------------------------------------------------------------------- */
struct rv_batch {
    std::vector<rv_index *> hs;
    RvBatchGroup *g = nullptr;
    std::vector<std::string> errs;
};
rv_batch *rv_batch_new(void) {
    try { return new rv_batch(); } catch (...) { rv_set_error("rv_batch_new: out of host memory"); return nullptr; }
}
int rv_batch_add(rv_batch *b, rv_index *h) {
    if (!b || !h) { rv_set_error("rv_batch_add: null argument"); return -1; }
    if (!b->hs.empty() && b->hs[0]->device != h->device) { rv_set_error("rv_batch_add: the handles of a batch live on one device"); return -1; }
    try { b->hs.push_back(h); } catch (...) { rv_set_error("rv_batch_add: out of host memory"); return -1; }
    return 0;
}
void rv_batch_free(rv_batch *b) {
    if (!b) return;
    rv_batch_group_free(b->g);
    delete b;
}
int rv_batch_info(const rv_batch *b, int64_t *out) {
    out[0] = out[1] = 0;
    if (b && b->g) rv_batch_group_info(b->g, out);
    return 0;
}
/* construct (when asked) + rv_align_builtin of every handle, each on a host thread and on its handle's stream; status[i] = what the calls of handle i
 * returned (0 = fine), stats[i] its statistics.  Handles with more than two samples meet in the middle of their anchor cascades and run the level loops as one
 * set of launches (rv_cascade_multi.hip); their results are those of rv_align_builtin called on each. */
int rv_batch_run(rv_batch *b, int minl, int minn, int construct, rv_align_stats *stats, int *status) {
    if (!b || b->hs.empty()) { rv_set_error("rv_batch_run: empty batch"); return -1; }
    const int n = (int)b->hs.size();
    try {
        b->errs.assign((size_t)n, std::string());
        if (!b->g) b->g = rv_batch_group_new(b->hs[0]->device);
        int members = 0;
        for (rv_index *h : b->hs) {
            const bool in = h->nsamples > 2 && !h->ws.opt.no_cascade;      // (what may reach the rendezvous at all; every early exit on the way there lets the group know)
            h->batch = in ? b->g : nullptr; h->batch_settled = false;
            members += in ? 1 : 0;
        }
        rv_batch_group_begin(b->g, members);
        std::vector<int> rc((size_t)n, 0);
        auto work = [&](int i) {
            rv_index *h = b->hs[(size_t)i];
            int r = 0;
            if (hipSetDevice(h->device) != hipSuccess) { rv_set_error("hipSetDevice failed"); r = -1; }
            if (r == 0 && construct) r = rv_construct(h, 0, "", "", 0);
            if (r == 0) r = rv_align_builtin(h, minl, minn, stats ? &stats[i] : nullptr);
            if (h->batch && !h->batch_settled) { h->batch_settled = true; rv_batch_group_leave(h->batch); }      // (it never got to the rendezvous)
            if (r != 0) b->errs[(size_t)i] = rv_last_error();      // (the error text is this thread's)
            rc[(size_t)i] = r;
        };
        std::vector<std::thread> th;
        int started = n;
        for (int i = 1; i < n; i++) {
            try { th.emplace_back(work, i); }
            catch (...) { started = i; break; }      // (no thread to be had: the rest one after the other below -- outside the group, which must not wait for them)
        }
        if (started < n) for (int i = started; i < n; i++) { rv_index *h = b->hs[(size_t)i]; if (h->batch) { h->batch_settled = true; rv_batch_group_leave(h->batch); h->batch = nullptr; } }
        work(0);
        for (auto &t : th) t.join();
        for (int i = started; i < n; i++) work(i);
        int bad = -1;
        for (int i = 0; i < n; i++) { b->hs[(size_t)i]->batch = nullptr; if (status) status[i] = rc[(size_t)i]; if (rc[(size_t)i] != 0 && bad < 0) bad = i; }
        if (bad >= 0) { rv_set_error("rv_batch_run: job %d: %s", bad, b->errs[(size_t)bad].c_str()); return -1; }
        return 0;
    } catch (const std::exception &e) {
        for (rv_index *h : b->hs) h->batch = nullptr;
        rv_set_error("rv_batch_run: %s", e.what());
        return -1;
    }
}

/* device memory for the frontier hand-off between processes (include/reveal_amd.h) */
void *rv_dev_alloc(int device, int64_t bytes) {
    void *p = nullptr;
    if (bytes < 0 || hipSetDevice(device) != hipSuccess) { rv_set_error("rv_dev_alloc: bad device or size"); return nullptr; }
    const hipError_t e = hipMalloc(&p, (size_t)std::max<int64_t>(bytes, 256));
    if (e != hipSuccess) { (void)hipGetLastError(); rv_set_error("rv_dev_alloc(%lld bytes): %s", (long long)bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
int rv_dev_free(int device, void *p) {
    if (!p) return 0;
    RV_HIP(hipSetDevice(device));
    RV_HIP(hipFree(p));
    return 0;
}
int rv_ipc_export(int device, const void *p, uint8_t handle[64]) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    RV_HIP(hipSetDevice(device));
    hipIpcMemHandle_t hd;
    RV_HIP(hipIpcGetMemHandle(&hd, const_cast<void *>(p)));
    memcpy(handle, &hd, 64);
    return 0;
}
void *rv_ipc_open(int device, const uint8_t handle[64]) {
    if (hipSetDevice(device) != hipSuccess) { rv_set_error("rv_ipc_open: bad device"); return nullptr; }
    hipIpcMemHandle_t hd;
    memcpy(&hd, handle, 64);
    void *p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); rv_set_error("rv_ipc_open: %s", hipGetErrorString(e)); return nullptr; }
    return p;
}
int rv_ipc_close(int device, void *p) {
    if (!p) return 0;
    RV_HIP(hipSetDevice(device));
    RV_HIP(hipIpcCloseMemHandle(p));
    return 0;
}
int rv_dev_copy(int device, void *dst, const void *src, int64_t bytes) {
    if (bytes <= 0) return 0;
    RV_HIP(hipSetDevice(device));
    RV_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDefault));
    // (a copy between two device addresses may return before it is done; the library's own streams do not wait for the null stream: one run in four of the
    //  three-process hand-off test imported segments that were still arriving)
    RV_HIP(hipStreamSynchronize(nullptr));
    return 0;
}

int rv_measure_bandwidth(int device, int64_t bytes, int iters, double *read_gbs, double *copy_gbs) {
    if (bytes < (1 << 20) || iters < 1) { rv_set_error("rv_measure_bandwidth: bytes >= 1 MiB and iters >= 1 needed"); return -1; }
    RV_HIP(hipSetDevice(device));
    const int64_t n16 = bytes / 16;
    void *a = nullptr, *b = nullptr; unsigned *sink = nullptr;
    hipStream_t q = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    RV_HIP(hipMalloc(&a, (size_t)n16 * 16)); RV_HIP(hipMalloc(&b, (size_t)n16 * 16)); RV_HIP(hipMalloc((void **)&sink, 64));
    RV_HIP(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    RV_HIP(hipEventCreate(&e0)); RV_HIP(hipEventCreate(&e1));
    RV_HIP(hipMemsetAsync(a, 1, (size_t)n16 * 16, q)); RV_HIP(hipMemsetAsync(b, 2, (size_t)n16 * 16, q));
    const unsigned grid = 256 * 16;      // 16 workgroups per CU, grid-stride
    double out[2] = {0, 0};
    for (int pass = 0; pass < 2; pass++) {
        for (int r = -1; r < iters; r++) {       // (r = -1: warm-up)
            if (r == 0) RV_HIP(hipEventRecord(e0, q));
            if (pass == 0) hipLaunchKernelGGL(k_bw_read, dim3(grid), dim3(256), 0, q, (const bw_v4 *)a, n16, sink);
            else hipLaunchKernelGGL(k_bw_copy, dim3(grid), dim3(256), 0, q, (const bw_v4 *)a, (bw_v4 *)b, n16);
        }
        RV_HIP(hipEventRecord(e1, q));
        RV_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        RV_HIP(hipEventElapsedTime(&ms, e0, e1));
        out[pass] = (double)n16 * 16 * (pass ? 2 : 1) * iters / ((double)ms * 1e6);      // GB/s; the copy counts read + write
    }
    if (read_gbs) *read_gbs = out[0];
    if (copy_gbs) *copy_gbs = out[1];
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(q);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
    return 0;
}

/* ---- primitive self-tests (host buffers in, host buffers out) -------------- */
static int test_ws(Workspace &ws) {
    if (rv_device_count() <= 0) { rv_set_error("no HIP device"); return -1; }
    RV_HIP(hipSetDevice(0));
    RV_HIP(hipStreamCreateWithFlags(&ws.stream, hipStreamNonBlocking));
    return 0;
}
static void test_ws_done(Workspace &ws) { ws.release(); (void)hipStreamDestroy(ws.stream); }

static int test_scan(const uint32_t *in, uint32_t *out, int64_t n, bool maxscan) {
    Workspace ws;
    RV_TRY(test_ws(ws));
    DBuf a, b;
    int r = 0;
    if (a.reserve((size_t)n * 4 + 64) || b.reserve((size_t)n * 4 + 64)) r = -1;
    if (!r && hipMemcpy(a.p, in, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) r = -1;
    if (!r) r = maxscan ? rv_inclusive_max_u32(ws, a.as<u32>(), b.as<u32>(), n) : rv_exclusive_sum_u32(ws, a.as<u32>(), b.as<u32>(), n);
    if (!r && hipStreamSynchronize(ws.stream) != hipSuccess) { rv_set_error("sync failed: %s", hipGetErrorString(hipGetLastError())); r = -1; }
    if (!r && hipMemcpy(out, b.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) r = -1;
    a.release(); b.release();
    test_ws_done(ws);
    return r;
}
int rv_test_exclusive_sum_u32(const uint32_t *in, uint32_t *out, int64_t n) { return test_scan(in, out, n, false); }
int rv_test_inclusive_max_u32(const uint32_t *in, uint32_t *out, int64_t n) { return test_scan(in, out, n, true); }

int rv_test_radix_sort(uint64_t *keys, uint32_t *vals, int64_t n, int bit_lo, int bit_hi) {
    Workspace ws;
    RV_TRY(test_ws(ws));
    DBuf k0, k1, v0, v1;
    int r = 0, in1 = 0;
    if (k0.reserve((size_t)n * 8 + 64) || k1.reserve((size_t)n * 8 + 64) || v0.reserve((size_t)n * 4 + 64) || v1.reserve((size_t)n * 4 + 64)) r = -1;
    if (!r && hipMemcpy(k0.p, keys, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) r = -1;
    if (!r && hipMemcpy(v0.p, vals, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) r = -1;
    if (!r) r = rv_radix_sort_pairs<u32>(ws, k0.as<u64>(), v0.as<u32>(), k1.as<u64>(), v1.as<u32>(), n, bit_lo, bit_hi, &in1);
    if (!r && hipStreamSynchronize(ws.stream) != hipSuccess) { rv_set_error("sync failed: %s", hipGetErrorString(hipGetLastError())); r = -1; }
    if (!r && hipMemcpy(keys, in1 ? k1.p : k0.p, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) r = -1;
    if (!r && hipMemcpy(vals, in1 ? v1.p : v0.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) r = -1;
    k0.release(); k1.release(); v0.release(); v1.release();
    test_ws_done(ws);
    return r;
}

/* Timing of the radix sort on keys made on the device (tools/ubench/radix_time.py): n pairs, the low `bits` bits of the keys sorted.
 * dist 0: uniform bits; 1: the digits of a base-5 number of 17 symbols drawn from {1..4} (the first keys of a DNA text); 2: constant.
 * flags: bit 0 = 10-bit digits, bit 1 = XCD-aware tile order, bit 2 = 16-bit wave counters, bit 3 = no digit bytes.  ms[it] = duration of the whole sort by HIP
 * events on the sort's own stream; *bad = adjacent pairs out of order (keys, then original index: stability) after the last one. */
}
namespace {
__device__ inline u64 rt_mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ __launch_bounds__(256) void k_rt_fill(u64 *keys, u32 *vals, int64_t n, int bits, int dist) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 r = rt_mix((u64)i * 0x9E3779B97F4A7C15ull + 12345);
    u64 k = 0;
    if (dist == 0) k = r;
    else if (dist == 1) { u64 r2 = rt_mix(r); for (int j = 0; j < 17; j++) { k = k * 5 + 1 + (r2 & 3); r2 >>= 2; } }
    const u64 mask = bits >= 64 ? ~0ull : (1ull << bits) - 1;
    keys[i] = (k & mask) | (r & ~mask & 0x00ffffffffffffffull);      // (payload bits above the sort key, as the SA build has them)
    vals[i] = (u32)i;
}
__global__ __launch_bounds__(256) void k_rt_check(const u64 *keys, const u32 *vals, int64_t n, int bits, unsigned long long *bad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 >= n) return;
    const u64 mask = bits >= 64 ? ~0ull : (1ull << bits) - 1;
    const u64 a = keys[i] & mask, b = keys[i + 1] & mask;
    if (a > b || (a == b && vals[i] > vals[i + 1])) atomicAdd(bad, 1ull);
}
}
extern "C" {
int rv_test_radix_time(int64_t n, int bits, int dist, int flags, int iters, double *ms, int64_t *bad) {
    Workspace ws;
    RV_TRY(test_ws(ws));
    ws.opt.rs_bits = (flags & 1) ? 10 : 8; ws.opt.rs_xcd = (flags & 2) ? 1 : 0; ws.opt.rs_cnt16 = (flags & 4) ? 1 : 0; ws.opt.rs_no_digit_bytes = (flags & 8) ? 1 : 0;
    DBuf k0, k1, v0, v1, db;
    int r = 0, in1 = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (k0.reserve((size_t)n * 8 + 64) || k1.reserve((size_t)n * 8 + 64) || v0.reserve((size_t)n * 4 + 64) || v1.reserve((size_t)n * 4 + 64) || db.reserve(64)) r = -1;
    if (!r && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) r = -1;
    for (int it = 0; !r && it < iters; it++) {
        hipLaunchKernelGGL(k_rt_fill, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ws.stream, k0.as<u64>(), v0.as<u32>(), n, bits, dist);
        (void)hipEventRecord(e0, ws.stream);
        r = rv_radix_sort_pairs<u32>(ws, k0.as<u64>(), v0.as<u32>(), k1.as<u64>(), v1.as<u32>(), n, 0, bits, &in1);
        (void)hipEventRecord(e1, ws.stream);
        if (!r && hipEventSynchronize(e1) != hipSuccess) { rv_set_error("sync failed: %s", hipGetErrorString(hipGetLastError())); r = -1; }
        float t = 0;
        if (!r) { (void)hipEventElapsedTime(&t, e0, e1); ms[it] = t; }
    }
    if (!r) {
        (void)hipMemsetAsync(db.p, 0, 8, ws.stream);
        hipLaunchKernelGGL(k_rt_check, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ws.stream, (const u64 *)(in1 ? k1.p : k0.p), (const u32 *)(in1 ? v1.p : v0.p), n, bits,
                           db.as<unsigned long long>());
        unsigned long long b = 0;
        if (hipMemcpyAsync(&b, db.p, 8, hipMemcpyDeviceToHost, ws.stream) != hipSuccess || hipStreamSynchronize(ws.stream) != hipSuccess) r = -1;
        *bad = (int64_t)b;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    k0.release(); k1.release(); v0.release(); v1.release(); db.release();
    test_ws_done(ws);
    return r;
}

}  // extern "C"

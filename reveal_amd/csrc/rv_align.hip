// rv_align.hip -- the recursion of align()/aligner() (reveallib/interface.c:293-415,
// reveallib/reveal.c:731-1338), level-synchronous.
//
// The reference pops one sub-index at a time (LIFO, reveal.c:21-25), scans it,
// asks Python which match to split on, labels / splits / bubble-sorts it and
// pushes up to three children.  Children cover disjoint text and only read
// text fixed by their ancestors, so the order is free: here every level of the
// recursion tree is one batch -- the sub-indices lie back to back in two
// ping-pong (SA, LCP) level arrays in HBM, one scan launch covers the whole
// frontier, the host (or the Python callbacks) decides per sub-index, one
// label/split/bubble pipeline produces the next level.
#include "rv_index.h"
#include "rv_split.h"
#include <string.h>
#include <algorithm>
#include <chrono>

namespace {

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

u64 hash_step(u64 acc, u64 i, int64_t v) {      // same as oracle/reveal_oracle.c ro_hash_step
    u64 x = (u64)v + (i + 1) * 0x9E3779B97F4A7C15ULL;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return acc + x;
}

// host staging of several small tables into one upload
struct Packer {
    std::vector<uint8_t> buf;
    size_t add(const void *p, size_t bytes) {
        size_t off = (buf.size() + 15) & ~(size_t)15;
        buf.resize(off + bytes);
        if (bytes) memcpy(buf.data() + off, p, bytes);
        return off;
    }
    template <class T> size_t addv(const std::vector<T> &v) { return add(v.data(), v.size() * sizeof(T)); }
    size_t reserve(size_t bytes) { size_t off = (buf.size() + 15) & ~(size_t)15; buf.resize(off + bytes, 0); return off; }
};

}  // namespace

struct Align {
    int minl = 0, minn = 0;
    bool multi = false;
    int level = 0;
    DBuf lvSA[2], lvLCP[2], lvBWT[2];
    int cur = 0;                 // which level buffer holds the frontier (level > 0)
    int64_t m = 0;               // ranks in the frontier
    std::vector<RvSub> subs;
    bool scanned = false;
    // level-wide scan result, CSR
    std::vector<u32> ml; std::vector<int32_t> mn; std::vector<int64_t> moff, mpos; std::vector<uint16_t> mso;
    // device scratch
    DBuf dD, dTab, dTile, dList, dFlag;
    // results of rv_align_builtin
    std::vector<u32> an_l; std::vector<int64_t> an_off, an_pos;
    bool trace_on = false;
    std::vector<rv_trace> trace;
    rv_align_stats st{};
    void release() {
        for (int k = 0; k < 2; k++) { lvSA[k].release(); lvLCP[k].release(); lvBWT[k].release(); }
        dD.release(); dTab.release(); dTile.release(); dList.release(); dFlag.release();
    }
};

void rv_align_free(rv_index *h) {
    if (h->al) { h->al->release(); delete h->al; h->al = nullptr; }
}

static const sa_t *cur_sa(rv_index *h) { Align *a = h->al; return a->level == 0 ? h->dSA.as<sa_t>() : a->lvSA[a->cur].as<sa_t>(); }
static const uint8_t *cur_bwt(rv_index *h) { Align *a = h->al; return a->level == 0 ? h->dBWT.as<uint8_t>() : a->lvBWT[a->cur].as<uint8_t>(); }
static const lcp_t *cur_lcp(rv_index *h) { Align *a = h->al; return a->level == 0 ? h->dLCP.as<lcp_t>() : a->lvLCP[a->cur].as<lcp_t>(); }

static int sample_of(const rv_index *h, int64_t pos) {       /* SO[pos], interface.c:116-134 */
    return (int)(std::lower_bound(h->nsep.begin(), h->nsep.end(), pos) - h->nsep.begin());
}

/* child sample count, reveal.c:1028-1042 */
static int count_samples(const rv_index *h, const std::vector<RvIntv> &iv) {
    if (h->nsamples > 2) {
        std::vector<int> seen;
        for (auto &x : iv) { int s = sample_of(h, x.begin); if (std::find(seen.begin(), seen.end(), s) == seen.end()) seen.push_back(s); }
        return (int)seen.size();
    }
    bool f0 = false, f1 = false;
    for (auto &x : iv) { if (x.begin < h->nsep[0]) f0 = true; if (x.begin > h->nsep[0]) f1 = true; }
    return (int)f0 + (int)f1;
}

static int need_align(rv_index *h) {
    if (!h->al) { rv_set_error("align not started (rv_align_begin)"); return -1; }
    return 0;
}

extern "C" {

int rv_align_begin(rv_index *h, int minl, int minn) {
    if (!h->constructed || h->main_arrays_freed) { rv_set_error("Index not yet constructed, alignment stopped."); return -1; }
    if (h->nsamples < 2) { rv_set_error("align needs at least two samples"); return -1; }
    RV_HIP(hipSetDevice(h->device));
    rv_align_free(h);
    Align *a = h->al = new Align();
    a->minl = minl; a->minn = minn;
    a->multi = h->nsamples > 2;
    a->m = h->n;
    RvSub root;
    root.off = 0; root.n = h->n; root.depth = 0; root.nsamples = h->nsamples; root.parent = -1; root.kind = 0;
    root.nodes = h->nodes;
    std::sort(root.nodes.begin(), root.nodes.end(), [](const RvIntv &x, const RvIntv &y) { return x.begin < y.begin; });
    a->subs.push_back(std::move(root));
    return 0;
}

int rv_frontier_size(rv_index *h) { return h->al ? (int)h->al->subs.size() : 0; }

int rv_align_end(rv_index *h) {
    if (h->al) { h->al->subs.clear(); h->al->scanned = false; }
    return 0;
}

int rv_set_trace(rv_index *h, int on) {
    if (!h->al) { h->al = new Align(); }
    h->al->trace_on = on != 0;
    return 0;
}

}  // extern "C"

int rv_run_multi_scan(rv_index *h, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t m, int minl, int minn, int mems,
                      std::vector<u32> &l, std::vector<int32_t> &n, std::vector<int64_t> &off, std::vector<uint16_t> &so,
                      std::vector<int64_t> &pos, std::vector<int64_t> *ub_out);

extern "C" {

/* reveal.c:802-822 for every sub-index of the frontier */
int rv_frontier_scan(rv_index *h) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    RV_HIP(hipSetDevice(h->device));
    const double t0 = now_s();
    a->ml.clear(); a->mn.clear(); a->moff.assign(1, 0); a->mso.clear(); a->mpos.clear();
    for (auto &s : a->subs) { s.mum_first = 0; s.nmums = 0; }
    if (!a->multi) {
        std::vector<RvPairRec> recs;
        RV_TRY(rv_run_pair_scan(h, cur_sa(h), cur_lcp(h), cur_bwt(h), a->m, a->minl, recs));
        const size_t nr = recs.size();
        a->ml.resize(nr); a->mn.assign(nr, 2); a->moff.resize(nr + 1); a->mso.resize(2 * nr); a->mpos.resize(2 * nr);
        size_t si = 0;
        for (size_t k = 0; k < nr; k++) {                     /* (l, 2, ((0,a),(1,b)))  reveal.c:166-170 */
            a->ml[k] = recs[k].l; a->moff[k] = (int64_t)(2 * k);
            a->mso[2 * k] = 0; a->mpos[2 * k] = recs[k].a;
            a->mso[2 * k + 1] = 1; a->mpos[2 * k + 1] = recs[k].b;
            while (si < a->subs.size() && (int64_t)recs[k].rank >= a->subs[si].off + a->subs[si].n) si++;
            if (si >= a->subs.size()) { rv_set_error("scan record outside the frontier"); return -1; }
            RvSub &s = a->subs[si];
            if (s.nmums == 0) s.mum_first = (int64_t)k;
            s.nmums++;
        }
        a->moff[nr] = (int64_t)(2 * nr);
    } else {
        std::vector<int64_t> ub;
        RV_TRY(rv_run_multi_scan(h, cur_sa(h), cur_lcp(h), cur_bwt(h), a->m, a->minl, a->minn, 0, a->ml, a->mn, a->moff, a->mso, a->mpos, &ub));
        size_t si = 0;
        for (size_t k = 0; k < a->ml.size(); k++) {
            while (si < a->subs.size() && ub[k] >= a->subs[si].off + a->subs[si].n) si++;
            if (si >= a->subs.size()) { rv_set_error("scan record outside the frontier"); return -1; }
            RvSub &s = a->subs[si];
            if (s.nmums == 0) s.mum_first = (int64_t)k;
            s.nmums++;
        }
    }
    a->scanned = true;
    a->st.scanned_ranks += a->m;
    a->st.t_scan += now_s() - t0;
    return 0;
}

int rv_sub_info(rv_index *h, int s, rv_sub *out) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    if (s < 0 || s >= (int)a->subs.size()) { rv_set_error("sub-index %d out of range", s); return -1; }
    const RvSub &x = a->subs[s];
    memset(out, 0, sizeof *out);
    out->n = x.n; out->depth = x.depth; out->nsamples = x.nsamples; out->nnodes = (int32_t)x.nodes.size();
    out->parent = x.parent; out->kind = x.kind; out->nmums = x.nmums;
    out->nmembers = x.nmums ? a->moff[(size_t)(x.mum_first + x.nmums)] - a->moff[(size_t)x.mum_first] : 0;
    return 0;
}

int rv_sub_nodes(rv_index *h, int s, int64_t *be) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    if (s < 0 || s >= (int)a->subs.size()) { rv_set_error("sub-index %d out of range", s); return -1; }
    const RvSub &x = a->subs[s];
    for (size_t k = 0; k < x.nodes.size(); k++) { be[2 * k] = x.nodes[k].begin; be[2 * k + 1] = x.nodes[k].end; }
    return 0;
}

int rv_sub_mums(rv_index *h, int s, uint32_t *l, int32_t *n, int64_t *off, uint16_t *so, int64_t *pos) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    if (s < 0 || s >= (int)a->subs.size()) { rv_set_error("sub-index %d out of range", s); return -1; }
    const RvSub &x = a->subs[s];
    const int64_t base = x.nmums ? a->moff[(size_t)x.mum_first] : 0;
    for (int64_t k = 0; k < x.nmums; k++) {
        const size_t g = (size_t)(x.mum_first + k);
        l[k] = a->ml[g]; n[k] = a->mn[g]; off[k] = a->moff[g] - base;
        for (int64_t q = a->moff[g]; q < a->moff[g + 1]; q++) { so[q - base] = a->mso[(size_t)q]; pos[q - base] = a->mpos[(size_t)q]; }
    }
    off[x.nmums] = x.nmums ? a->moff[(size_t)(x.mum_first + x.nmums)] - base : 0;
    return 0;
}

static int upload_sub_starts(rv_index *h, DBuf &buf) {
    Align *a = h->al;
    std::vector<int64_t> st(a->subs.size() + 1);
    for (size_t k = 0; k < a->subs.size(); k++) st[k] = a->subs[k].off;
    st[a->subs.size()] = a->m;
    RV_TRY(buf.reserve(st.size() * 8));
    RV_HIP(hipMemcpyAsync(buf.p, st.data(), st.size() * 8, hipMemcpyHostToDevice, h->ws.stream));
    RV_HIP(hipStreamSynchronize(h->ws.stream));
    return 0;
}

int64_t rv_sub_array(rv_index *h, int s, int which, void *out, int64_t cap) {
    if (need_align(h)) return -1;
    Align *a = h->al;
    (void)hipSetDevice(h->device);
    if (which == RV_SAI) {     /* the shared inverse: rank inside the owning sub-index (reveal.c:597,609,630) */
        if (cap < h->nT) { rv_set_error("buffer too small"); return -1; }
        if (a->level > 0) {
            if (upload_sub_starts(h, h->ws.misc[4])) return -1;
            if (rv_sai_level_launch(h->ws, cur_sa(h), a->m, h->ws.misc[4].as<int64_t>(), (int)a->subs.size(), h->dSAi.as<sa_t>())) return -1;
        }
        (void)hipStreamSynchronize(h->ws.stream);
        if (hipMemcpy(out, h->dSAi.p, (size_t)h->nT * sizeof(sa_t), hipMemcpyDeviceToHost) != hipSuccess) { rv_set_error("D2H failed"); return -1; }
        return h->nT;
    }
    if (s < 0 || s >= (int)a->subs.size()) { rv_set_error("sub-index %d out of range", s); return -1; }
    const RvSub &x = a->subs[s];
    if (cap < x.n) { rv_set_error("buffer too small"); return -1; }
    (void)hipStreamSynchronize(h->ws.stream);
    hipError_t e;
    if (which == RV_SA) e = hipMemcpy(out, cur_sa(h) + x.off, (size_t)x.n * sizeof(sa_t), hipMemcpyDeviceToHost);
    else if (which == RV_LCP) e = hipMemcpy(out, cur_lcp(h) + x.off, (size_t)x.n * sizeof(lcp_t), hipMemcpyDeviceToHost);
    else { rv_set_error("rv_sub_array: bad array id"); return -1; }
    if (e != hipSuccess) { rv_set_error("D2H failed"); return -1; }
    return x.n;
}

int rv_sub_split(rv_index *h, int s, uint32_t l, int nsp, const int64_t *sp,
                 const int64_t *lead, int nlead, const int64_t *trail, int ntrail,
                 const int64_t *match, int nmatch, const int64_t *rest, int nrest) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    if (s < 0 || s >= (int)a->subs.size()) { rv_set_error("sub-index %d out of range", s); return -1; }
    RvSub &x = a->subs[s];
    auto fill = [](std::vector<RvIntv> &v, const int64_t *p, int n) {
        v.resize((size_t)n);
        for (int k = 0; k < n; k++) { v[(size_t)k].begin = p[2 * k]; v[(size_t)k].end = p[2 * k + 1]; }
    };
    for (int k = 0; k < nsp; k++)
        if (sp[k] < 0 || sp[k] + (int64_t)l > h->nT) { rv_set_error("match outside the text"); return -1; }
    x.has_split = true; x.l = l;
    x.sp.assign(sp, sp + nsp);
    fill(x.lead, lead, nlead); fill(x.trail, trail, ntrail); fill(x.match, match, nmatch); fill(x.rest, rest, nrest);
    for (auto *v : {&x.lead, &x.trail, &x.rest, &x.match})
        for (auto &iv : *v)
            if (iv.begin < 0 || iv.end > h->nT || iv.begin > iv.end) { rv_set_error("interval outside the text"); x.has_split = false; return -1; }
    return 0;
}

/* reveal.c:1005-1252 for every decided sub-index; children -> next frontier */
int rv_frontier_commit(rv_index *h, int32_t *children) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    RV_HIP(hipSetDevice(h->device));
    hipStream_t q = h->ws.stream;
    const double t0 = now_s();
    const int ns = (int)a->subs.size();
    if (children) for (int k = 0; k < 3 * ns; k++) children[k] = -1;
    std::vector<int> split_subs;
    for (int s = 0; s < ns; s++) if (a->subs[s].has_split) split_subs.push_back(s);
    a->scanned = false;
    if (split_subs.empty()) { a->subs.clear(); a->m = 0; return 0; }

    // ---- interval tables ---------------------------------------------------------
    struct Ent { int64_t b, e; uint8_t c; };
    std::vector<Ent> cls;
    std::vector<RvIntv> mt;
    for (int s : split_subs) {
        const RvSub &x = a->subs[s];
        for (auto &iv : x.lead) if (iv.end > iv.begin) cls.push_back({iv.begin, iv.end, 1});
        for (auto &iv : x.trail) if (iv.end > iv.begin) cls.push_back({iv.begin, iv.end, 2});
        for (auto &iv : x.rest) if (iv.end > iv.begin) cls.push_back({iv.begin, iv.end, 4});
        for (int64_t p : x.sp) if (x.l) mt.push_back({p, p + (int64_t)x.l});
    }
    std::sort(cls.begin(), cls.end(), [](const Ent &x, const Ent &y) { return x.b < y.b; });
    std::sort(mt.begin(), mt.end(), [](const RvIntv &x, const RvIntv &y) { return x.begin < y.begin; });
    for (size_t k = 1; k < cls.size(); k++)
        if (cls[k].b < cls[k - 1].e) { rv_set_error("graphalign returned overlapping intervals [%lld,%lld) / [%lld,%lld)", (long long)cls[k - 1].b, (long long)cls[k - 1].e, (long long)cls[k].b, (long long)cls[k].e); return -1; }
    std::vector<sa_t> cb(cls.size()), ce(cls.size()), mb(mt.size()), me(mt.size());
    std::vector<uint8_t> cc(cls.size());
    std::vector<int64_t> mpre(mt.size() + 1, 0);
    for (size_t k = 0; k < cls.size(); k++) { cb[k] = (sa_t)cls[k].b; ce[k] = (sa_t)cls[k].e; cc[k] = cls[k].c; }
    for (size_t k = 0; k < mt.size(); k++) { mb[k] = (sa_t)mt[k].begin; me[k] = (sa_t)mt[k].end; mpre[k + 1] = mpre[k] + (mt[k].end - mt[k].begin); }

    // ---- next level layout + child bookkeeping (reveal.c:1136-1207) -------------------
    std::vector<RvSub> next;
    std::vector<u32> child_base((size_t)ns * 3, 0), child_n((size_t)ns * 3, 0);
    std::vector<int64_t> sub_start((size_t)ns + 1);
    for (int s = 0; s < ns; s++) sub_start[(size_t)s] = a->subs[s].off;
    sub_start[(size_t)ns] = a->m;
    std::vector<int> cut_first((size_t)ns + 1, 0), mend_first((size_t)ns + 1, 0);
    std::vector<sa_t> cut_lo, cut_hi, mend_pos;
    std::vector<std::vector<RvBubbleDesc>> rounds;
    int64_t running = 0;
    const int64_t lcap = (int64_t)h->maxlcp;
    {
        size_t si = 0;
        for (int s = 0; s < ns; s++) {
            cut_first[(size_t)s] = (int)cut_lo.size();
            mend_first[(size_t)s] = (int)mend_pos.size();
            if (si >= split_subs.size() || split_subs[si] != s) continue;
            si++;
            RvSub &x = a->subs[s];
            for (int64_t p : x.sp) mend_pos.push_back((sa_t)(p + (int64_t)x.l));
            auto isort = [](std::vector<RvIntv> &v) { std::sort(v.begin(), v.end(), [](const RvIntv &p, const RvIntv &r) { return p.begin < r.begin; }); };
            std::vector<RvIntv> *lists[3] = {&x.lead, &x.trail, &x.rest};
            int64_t lead_off = 0, lead_n = 0;
            for (int c = 0; c < 3; c++) {
                int64_t cn = 0;
                for (auto &iv : *lists[c]) cn += iv.end - iv.begin;
                child_base[(size_t)s * 3 + c] = (u32)running;
                child_n[(size_t)s * 3 + c] = (u32)cn;
                if (c == 0) { lead_off = running; lead_n = cn; }
                if (cn > 0) {
                    RvSub ch;
                    ch.off = running; ch.n = cn; ch.depth = x.depth + 1; ch.parent = s; ch.kind = c + 1;
                    ch.nodes = *lists[c];
                    ch.nsamples = count_samples(h, ch.nodes);
                    isort(ch.nodes);
                    if (children) children[3 * s + c] = (int32_t)next.size();
                    next.push_back(std::move(ch));
                    running += cn;
                }
            }
            // windows in front of this sub's cuts, in the order graphalign listed the matched intervals
            if (lead_n > 0) {
                for (size_t r = 0; r < x.match.size(); r++) {
                    const int64_t B = x.match[r].begin;
                    int64_t lo = B;
                    for (auto &iv : x.lead) if (iv.end == B && iv.begin < B) { lo = std::max(iv.begin, B - lcap); break; }
                    cut_lo.push_back((sa_t)lo); cut_hi.push_back((sa_t)B);
                }
                const int c0 = cut_first[(size_t)s], c1 = (int)cut_lo.size();
                for (size_t r = 0; r < x.match.size(); r++) {
                    const int64_t B = x.match[r].begin, lo = (int64_t)cut_lo[(size_t)c0 + r];
                    if (lo >= B) continue;
                    if (rounds.size() <= r) rounds.resize(r + 1);
                    RvBubbleDesc d; d.off = lead_off; d.n = lead_n; d.B = B; d.wlo = lo; d.cut0 = c0; d.cut1 = c1;
                    rounds[r].push_back(d);
                }
            }
        }
        cut_first[(size_t)ns] = (int)cut_lo.size();
        mend_first[(size_t)ns] = (int)mend_pos.size();
    }
    const int64_t m_next = running;
    if (m_next >= ((int64_t)1 << 32)) { rv_set_error("level larger than 2^32 ranks not supported yet"); return -1; }
    std::vector<RvBubbleDesc> descs;
    std::vector<int> round_first, round_small;
    for (auto &r : rounds) {      // per round: ordinary children first, then the large ones (bigger workgroups)
        std::stable_partition(r.begin(), r.end(), [](const RvBubbleDesc &d) { return d.n <= RV_BUBBLE_BIG_N; });
        int nsmall = 0;
        for (auto &d : r) nsmall += d.n <= RV_BUBBLE_BIG_N;
        round_first.push_back((int)descs.size()); round_small.push_back(nsmall);
        descs.insert(descs.end(), r.begin(), r.end());
    }
    round_first.push_back((int)descs.size());
    std::vector<int64_t> woff(descs.size() + 1, 0);
    for (size_t k = 0; k < descs.size(); k++) woff[k + 1] = woff[k] + (descs[k].B - descs[k].wlo);

    // ---- one upload for all the tables ---------------------------------------------------
    const int64_t ntiles = ceil_div(a->m, RV_SPLIT_TILE);
    Packer pk;
    const size_t o_cb = pk.addv(cb), o_ce = pk.addv(ce), o_cc = pk.addv(cc), o_mb = pk.addv(mb), o_me = pk.addv(me), o_mpre = pk.addv(mpre);
    const size_t o_ss = pk.addv(sub_start), o_cbase = pk.addv(child_base), o_cn = pk.addv(child_n), o_cf = pk.addv(cut_first);
    const size_t o_clo = pk.addv(cut_lo), o_chi = pk.addv(cut_hi), o_split = pk.addv(split_subs);
    const size_t o_desc = pk.addv(descs), o_woff = pk.addv(woff), o_mf = pk.addv(mend_first), o_mp = pk.addv(mend_pos);
    const size_t o_suboff = pk.reserve((size_t)ns * 3 * 4), o_total = pk.reserve(16), o_err = pk.reserve(16), o_bcnt = pk.reserve(descs.size() * 4 + 4);
    RV_TRY(a->dTab.reserve(pk.buf.size() + 64));
    RV_HIP(hipMemcpyAsync(a->dTab.p, pk.buf.data(), pk.buf.size(), hipMemcpyHostToDevice, q));
    uint8_t *tb = a->dTab.as<uint8_t>();
    RV_TRY(a->dD.reserve((size_t)a->m + 64));
    RV_TRY(a->dTile.reserve((size_t)ntiles * 3 * 5 * 4 + 64));
    RV_TRY(a->dList.reserve((size_t)woff.back() * 4 + 64));
    const int nxt = (a->level == 0) ? 0 : (a->cur ^ 1);
    RV_TRY(a->lvSA[nxt].reserve((size_t)(m_next + 64) * sizeof(sa_t)));
    RV_TRY(a->lvLCP[nxt].reserve((size_t)(m_next + 64) * sizeof(lcp_t)));
    RV_TRY(a->lvBWT[nxt].reserve((size_t)m_next + 64));

    RvLabelTabs lt;
    lt.cbegin = (const sa_t *)(tb + o_cb); lt.cend = (const sa_t *)(tb + o_ce); lt.ccls = tb + o_cc; lt.ncls = (int)cls.size();
    lt.mbegin = (const sa_t *)(tb + o_mb); lt.mend = (const sa_t *)(tb + o_me); lt.nmatch = (int)mt.size();
    int id = h->prof.begin(q, RV_K_LABEL, (double)a->m * (sizeof(sa_t) + 1));
    RV_TRY(rv_label_launch(h->ws, cur_sa(h), a->m, lt, a->dD.as<uint8_t>()));
    h->prof.end(q, id);

    RvSplitArgs sa;
    u32 *tiles = a->dTile.as<u32>();
    sa.ntiles = ntiles;
    sa.tile_cnt = tiles; sa.tile_has = tiles + 3 * ntiles; sa.tile_post = tiles + 6 * ntiles;
    sa.tile_G = tiles + 9 * ntiles; sa.tile_carry = tiles + 12 * ntiles;
    sa.total = (u32 *)(tb + o_total);
    sa.sub_start = (const int64_t *)(tb + o_ss); sa.nsubs = ns;
    sa.child_base = (const u32 *)(tb + o_cbase); sa.child_n = (const u32 *)(tb + o_cn); sa.sub_off = (u32 *)(tb + o_suboff);
    sa.cut_first = (const int *)(tb + o_cf); sa.cut_lo = (const sa_t *)(tb + o_clo); sa.cut_hi = (const sa_t *)(tb + o_chi);
    sa.mend_first = (const int *)(tb + o_mf); sa.mend_pos = (const sa_t *)(tb + o_mp);
    sa.SA_out = a->lvSA[nxt].as<sa_t>(); sa.LCP_out = a->lvLCP[nxt].as<lcp_t>(); sa.BWT_out = a->lvBWT[nxt].as<uint8_t>(); sa.SAi = h->dSAi.as<sa_t>();
    sa.err = (u32 *)(tb + o_err);
    id = h->prof.begin(q, RV_K_SPLIT, (double)a->m * (2 * (sizeof(sa_t) + sizeof(lcp_t) + 1)) + (double)m_next * (sizeof(sa_t) + sizeof(lcp_t)));
    RV_TRY(rv_split_launch(h->ws, cur_sa(h), cur_lcp(h), a->dD.as<uint8_t>(), cur_bwt(h), a->m, sa, (const int *)(tb + o_split), (int)split_subs.size()));
    h->prof.end(q, id);
    RV_TRY(rv_lower_launch(h->ws, h->dT.as<uint8_t>(), lt.mbegin, lt.mend, (const int64_t *)(tb + o_mpre), lt.nmatch, mpre.back()));
    const double t1 = now_s();

    // ---- bubble_sort rounds (reveal.c:1250-1252, :666-727) -----------------------------------
    RvBubbleArgs ba;
    ba.desc = (const RvBubbleDesc *)(tb + o_desc); ba.woff = (const int64_t *)(tb + o_woff);
    ba.cnt = (u32 *)(tb + o_bcnt); ba.list = a->dList.as<u32>();
    RV_TRY(a->dFlag.reserve((size_t)m_next + 64));
    RV_HIP(hipMemsetAsync(a->dFlag.p, 0, (size_t)m_next + 64, q));
    ba.flag = a->dFlag.as<uint8_t>();
    ba.SA = sa.SA_out; ba.LCP = sa.LCP_out; ba.BWT = sa.BWT_out; ba.SAi = sa.SAi; ba.cut_lo = sa.cut_lo; ba.cut_hi = sa.cut_hi; ba.err = sa.err;
    id = h->prof.begin(q, RV_K_BUBBLE, 0.0);
    for (size_t r = 0; r + 1 < round_first.size(); r++) {
        const int first = round_first[r], count = round_first[r + 1] - first;
        RV_TRY(rv_bubble_round_launch(h->ws, ba, first, round_small[r], count - round_small[r], woff[(size_t)(first + count)] - woff[(size_t)first]));
    }
    h->prof.end(q, id);
    u32 err = 0;
    RV_HIP(hipMemcpyAsync(&err, tb + o_err, 4, hipMemcpyDeviceToHost, q));
    RV_HIP(hipStreamSynchronize(q));
    if (err & 1u) { rv_set_error("split: the intervals returned by graphalign do not partition the sub-index (child size mismatch)"); return -1; }

    if (a->level == 0) h->main_arrays_freed = true;      /* reveal.c:1279-1284 */
    a->level++;
    a->cur = nxt;
    a->m = m_next;
    a->subs = std::move(next);
    a->st.t_split += t1 - t0;
    a->st.t_bubble += now_s() - t1;
    return 0;
}

/* ---- the whole recursion with the built-in benchmark callbacks ------------------- */
int rv_align_builtin(rv_index *h, int minl, int minn, rv_align_stats *out) {
    const bool trace_on = h->al && h->al->trace_on;
    RV_TRY(rv_align_begin(h, minl, minn));
    Align *a = h->al;
    a->trace_on = trace_on;
    memset(&a->st, 0, sizeof a->st);
    a->an_l.clear(); a->an_off.assign(1, 0); a->an_pos.clear(); a->trace.clear();
    std::vector<sa_t> hsa; std::vector<lcp_t> hlcp;
    while (!a->subs.empty()) {
        RV_TRY(rv_frontier_scan(h));
        const double t0 = now_s();
        if (a->trace_on) {
            hsa.resize((size_t)a->m); hlcp.resize((size_t)a->m);
            RV_HIP(hipMemcpy(hsa.data(), cur_sa(h), (size_t)a->m * sizeof(sa_t), hipMemcpyDeviceToHost));
            RV_HIP(hipMemcpy(hlcp.data(), cur_lcp(h), (size_t)a->m * sizeof(lcp_t), hipMemcpyDeviceToHost));
        }
        a->st.levels++;
        for (size_t s = 0; s < a->subs.size(); s++) {
            RvSub &x = a->subs[s];
            a->st.steps++;
            if (x.depth > a->st.maxdepth) a->st.maxdepth = x.depth;
            rv_trace tr;
            if (a->trace_on) {
                memset(&tr, 0, sizeof tr);
                tr.key = x.nodes.empty() ? -1 : x.nodes[0].begin;
                tr.n = x.n; tr.depth = x.depth; tr.nsamples = x.nsamples; tr.nnodes = (int32_t)x.nodes.size(); tr.nmums = x.nmums;
                u64 h1 = 0, h2 = 0, h3 = 0, c3 = 0;
                for (int64_t i = 0; i < x.n; i++) { h1 = hash_step(h1, (u64)i, (int64_t)hsa[(size_t)(x.off + i)]); h2 = hash_step(h2, (u64)i, (int64_t)hlcp[(size_t)(x.off + i)]); }
                for (int64_t k = x.mum_first; k < x.mum_first + x.nmums; k++) {
                    h3 = hash_step(h3, c3++, (int64_t)a->ml[(size_t)k]); h3 = hash_step(h3, c3++, a->mn[(size_t)k]);
                    for (int64_t qq = a->moff[(size_t)k]; qq < a->moff[(size_t)k + 1]; qq++) { h3 = hash_step(h3, c3++, a->mso[(size_t)qq]); h3 = hash_step(h3, c3++, a->mpos[(size_t)qq]); }
                }
                tr.h_sa = h1; tr.h_lcp = h2; tr.h_mums = h3;
            }
            // picker: longest match present in every sample of the sub-index, ties -> smallest minimum coordinate
            int64_t best = -1, bmin = 0; u32 bl = 0;
            for (int64_t k = x.mum_first; k < x.mum_first + x.nmums; k++) {
                if (a->mn[(size_t)k] != x.nsamples) continue;
                int64_t mnp = a->mpos[(size_t)a->moff[(size_t)k]];
                for (int64_t qq = a->moff[(size_t)k] + 1; qq < a->moff[(size_t)k + 1]; qq++) mnp = std::min(mnp, a->mpos[(size_t)qq]);
                if (best < 0 || a->ml[(size_t)k] > bl || (a->ml[(size_t)k] == bl && mnp < bmin)) { best = k; bl = a->ml[(size_t)k]; bmin = mnp; }
            }
            if (best >= 0) {
                // graphalign, linear interval model
                const int64_t q0 = a->moff[(size_t)best], q1 = a->moff[(size_t)best + 1];
                const int nm = (int)(q1 - q0);
                std::vector<int64_t> sp(a->mpos.begin() + q0, a->mpos.begin() + q1);
                std::vector<int64_t> lead, trail, match, rest;
                std::vector<uint8_t> touched(x.nodes.size(), 0);
                std::sort(sp.begin(), sp.end());
                for (int k = 0; k < nm; k++) {
                    const int64_t p = sp[(size_t)k];
                    size_t lo = 0, hi = x.nodes.size();
                    while (lo < hi) { size_t mid = (lo + hi) / 2; if (x.nodes[mid].begin <= p) lo = mid + 1; else hi = mid; }
                    if (lo == 0 || p >= x.nodes[lo - 1].end || p + (int64_t)bl > x.nodes[lo - 1].end) { rv_set_error("match at %lld is not inside an interval of its sub-index", (long long)p); return -1; }
                    const RvIntv iv = x.nodes[lo - 1];
                    touched[lo - 1] = 1;
                    if (p > iv.begin) { lead.push_back(iv.begin); lead.push_back(p); }
                    if (p + (int64_t)bl < iv.end) { trail.push_back(p + bl); trail.push_back(iv.end); }
                    match.push_back(p); match.push_back(p + bl);
                }
                for (size_t k = 0; k < x.nodes.size(); k++) if (!touched[k]) { rest.push_back(x.nodes[k].begin); rest.push_back(x.nodes[k].end); }
                RV_TRY(rv_sub_split(h, (int)s, bl, nm, sp.data(), lead.data(), (int)lead.size() / 2, trail.data(), (int)trail.size() / 2,
                                    match.data(), (int)match.size() / 2, rest.data(), (int)rest.size() / 2));
                a->an_l.push_back(bl);
                a->an_pos.insert(a->an_pos.end(), sp.begin(), sp.end());
                a->an_off.push_back((int64_t)a->an_pos.size());
                a->st.splits++; a->st.anchored_bp += bl;
                if (a->trace_on) { tr.picked = 1; tr.l = bl; tr.mn = nm; tr.sp_min = sp[0]; }
            }
            if (a->trace_on) a->trace.push_back(tr);
        }
        a->st.t_host += now_s() - t0;
        RV_TRY(rv_frontier_commit(h, nullptr));
    }
    if (out) *out = a->st;
    return 0;
}

int64_t rv_anchor_count(rv_index *h, int64_t *members) {
    if (need_align(h)) return -1;
    if (members) *members = (int64_t)h->al->an_pos.size();
    return (int64_t)h->al->an_l.size();
}
int rv_fetch_anchors(rv_index *h, uint32_t *l, int64_t *off, int64_t *pos) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    memcpy(l, a->an_l.data(), a->an_l.size() * 4);
    memcpy(off, a->an_off.data(), a->an_off.size() * 8);
    memcpy(pos, a->an_pos.data(), a->an_pos.size() * 8);
    return 0;
}
int64_t rv_trace_count(rv_index *h) { return h->al ? (int64_t)h->al->trace.size() : 0; }
int rv_fetch_trace(rv_index *h, rv_trace *out, int64_t cap) {
    RV_TRY(need_align(h));
    if (cap < (int64_t)h->al->trace.size()) { rv_set_error("buffer too small"); return -1; }
    memcpy(out, h->al->trace.data(), h->al->trace.size() * sizeof(rv_trace));
    return 0;
}

}  // extern "C"

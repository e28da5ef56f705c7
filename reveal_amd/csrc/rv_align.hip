// rv_align.hip -- the recursion of align()/aligner() (reveallib/interface.c:293-415,
// reveallib/reveal.c:731-1338), level-synchronous.
//
// The reference pops one sub-index at a time (LIFO, reveal.c:21-25), scans it,
// asks Python which match to split on, labels / splits / bubble-sorts it and
// pushes up to three children.  Children cover disjoint text and only read
// text fixed by their ancestors, so the order is free: here every level of the
// recursion tree is one batch -- the sub-indices lie back to back in two
// ping-pong (SA, LCP, BWT) level arrays in HBM, one scan launch covers the
// whole frontier, the host (or the Python callbacks) decides per sub-index, one
// label/split/bubble pipeline produces the next level.
//
// Host bookkeeping is flat (CSR arrays per level, reused across levels): the
// recursion visits ~10^5 sub-indices per 10 Mbp, so nothing here allocates per
// sub-index.
#include "rv_index.h"
#include "rv_split.h"
#include "rv_decide.h"
static_assert(RV_TSUB_TILE == RV_SPLIT_TILE, "one tile -> sub-index table serves the split passes and the multi-sample picker");
#include "rv_leaf.h"
#include "rv_cascade.h"
#include <string.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <atomic>
#include "rv_graphrem.h"

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

// host staging of several small tables into one upload.  Pinned memory: the copy is queued and the call returns (a
// pageable source is staged synchronously, ~10-20 us per level).  The buffer is rewritten only after a stream
// synchronisation (the next level's scan), so the queued copy has always finished by then.
struct Packer {
    uint8_t *p = nullptr;
    size_t used = 0, cap = 0;
    void clear() { used = 0; }
    const uint8_t *data() const { return p; }
    size_t size() const { return used; }
    void grow(size_t need) {
        if (need <= cap) return;
        size_t want = need + need / 2 + 4096;
        uint8_t *q = nullptr;
        size_t got = want;
        const bool was_pageable = pageable;
        bool now_pageable = false;
        if (rv_pinned_get(want, (void **)&q, &got) != hipSuccess) { q = (uint8_t *)malloc(want); got = want; now_pageable = true; }
        if (p) { (void)hipDeviceSynchronize(); memcpy(q, p, used); pageable = was_pageable; release_ptr(); }      // (a queued copy kernel may still read the old buffer)
        p = q; cap = got; pageable = now_pageable;
    }
    size_t add(const void *src, size_t bytes) {
        const size_t off = (used + 15) & ~(size_t)15;
        grow(off + bytes);
        if (off > used) memset(p + used, 0, off - used);
        if (bytes) memcpy(p + off, src, bytes);
        used = off + bytes;
        return off;
    }
    template <class T> size_t addv(const std::vector<T> &v) { return add(v.data(), v.size() * sizeof(T)); }
    size_t reserve(size_t bytes) {
        const size_t off = (used + 15) & ~(size_t)15;
        grow(off + bytes);
        memset(p + used, 0, off + bytes - used);
        used = off + bytes;
        return off;
    }
    bool pageable = false;
    void release_ptr() { if (p) { if (pageable) free(p); else rv_pinned_put(p, cap); } p = nullptr; }
    void release() { release_ptr(); cap = used = 0; }
};

// One recursion level: sub-index s owns ranks [off[s], off[s]+n[s]) of the level
// arrays and the intervals nodes[node_first[s] .. node_first[s+1]) (sorted by begin).
struct Level {
    int64_t m = 0;
    std::vector<int64_t> off, n;
    std::vector<int32_t> depth, nsamples, parent, kind;
    std::vector<int64_t> node_first;
    std::vector<RvIntv>  nodes;
    int size() const { return (int)off.size(); }
    void clear() { m = 0; off.clear(); n.clear(); depth.clear(); nsamples.clear(); parent.clear(); kind.clear(); node_first.assign(1, 0); nodes.clear(); }
};

// Decisions of one level (mumpicker + graphalign results), appended in any order.
struct Decisions {
    std::vector<int32_t> of_sub;          // per sub: decision index or -1
    std::vector<int32_t> sub;             // per decision
    std::vector<u32>     l;
    std::vector<int64_t> sp_first, lead_first, trail_first, rest_first, match_first;   // CSR, size ndec+1
    std::vector<int64_t> sp;
    std::vector<u32>     spl;             // length of the matched range at sp[k] (the match length; splitindex / extract: any)
    bool                 host_lists = false;   // some decision carries interval lists from the caller (not the built-in linear model)
    bool                 drop_lead = false;    // the leading class of the (only) decision is dropped like matched ranks, no child is made (builtin_cascade: the chain's consumed sequences)
    std::vector<RvIntv>  lead, trail, rest, match;     // lead/trail/rest sorted by begin per decision; match in the given order
    int size() const { return (int)sub.size(); }
    void reset(int nsubs) {
        of_sub.assign((size_t)nsubs, -1);
        sub.clear(); l.clear();
        sp_first.assign(1, 0); lead_first.assign(1, 0); trail_first.assign(1, 0); rest_first.assign(1, 0); match_first.assign(1, 0);
        sp.clear(); spl.clear(); lead.clear(); trail.clear(); rest.clear(); match.clear();
        host_lists = false; drop_lead = false;
    }
    void close() {
        sp_first.push_back((int64_t)sp.size()); lead_first.push_back((int64_t)lead.size()); trail_first.push_back((int64_t)trail.size());
        rest_first.push_back((int64_t)rest.size()); match_first.push_back((int64_t)match.size());
    }
};

inline bool intv_less(const RvIntv &x, const RvIntv &y) { return x.begin < y.begin; }

}  // namespace

struct Align {
    int minl = 0, minn = 0;
    bool multi = false;
    int level = 0;
    // three level buffers in rotation: the split of level k writes (k+1) % 3 while the leaf launch of level k-1 (second stream,
    // ~200 us) may still be reading (k-1) % 3 -- with two buffers the deep levels, shorter than a leaf launch, each waited for it
    DBuf lvSA[RV_LEVEL_BUFS], lvLCP[RV_LEVEL_BUFS], lvBWT[RV_LEVEL_BUFS];
    int cur = 0;                 // which level buffer holds the frontier (level > 0)
    Level lv, nx;                // current frontier / the one being built
    Decisions dec;
    bool scanned = false;
    std::vector<uint8_t> skip_scan;   // per sub of the current level: its matches come from the caller (skipmums, reveal.c:802,830-837), the scan's are not wanted
    bool full_only = false;      // multi scan pre-selection (built-in picker, no trace)
    const u32 *d_err = nullptr;  // error word of the last commit, checked with the next scan's copy
    // scan result of the level: pair records in rank order, or CSR for the multi scan
    std::vector<RvPairRec> recs;
    std::vector<u32> ml; std::vector<int32_t> mn; std::vector<int64_t> moff, mpos; std::vector<uint16_t> mso;
    std::vector<int64_t> mum_first, nmums;       // per sub
    // rv_set_picker: 0 = the benchmark picker (longest match in every sample), 1 = the reference's default picker in C++ (rv_pick_chain, rv_chain.hip).
    // Its seeds (schemes.py:321-332; reveal.c:1157, 1180 hands them to the children as skipmums) wait here: per sub-index of the current level the
    // list it was seeded with (empty: it is scanned), and the two lists its own decision leaves for its leading / trailing child
    int picker = 0; rv_picker_args pargs{};
    struct SeedList {
        std::vector<u32> l; std::vector<int32_t> n; std::vector<int64_t> off; std::vector<uint16_t> so; std::vector<int64_t> pos, score;
        size_t size() const { return l.size(); }
        void clear() { l.clear(); n.clear(); off.assign(1, 0); so.clear(); pos.clear(); score.clear(); }
        void push(u32 l_, int32_t n_, const uint16_t *so_, const int64_t *pos_, int members, int64_t sc) {
            if (off.empty()) off.assign(1, 0);
            l.push_back(l_); n.push_back(n_); score.push_back(sc);
            so.insert(so.end(), so_, so_ + members); pos.insert(pos.end(), pos_, pos_ + members); off.push_back((int64_t)pos.size());
        }
    };
    std::vector<SeedList> seeds_cur, seeds_lead, seeds_trail;
    // rv_set_graph_picker (kind 2): graph inputs -- the picker and graphalign of `reveal rem` for graphs (rv_graphrem.hip) on the caller's rv_graph, called per
    // sub-index in the reference's order; per sub-index of the current level its left / right graph node (rem.py:318-382: graphalign hands the children theirs)
    rv_graph *ggraph = nullptr;
    // rv_set_replay_graph (picker kind 1): the anchors' surgery follows the run level by level on a host thread of its own (rv_graph.hip RvReplayFeed)
    rv_graph *replay_graph = nullptr; RvReplayFeed *feed = nullptr; size_t fed = 0;
    std::vector<RvGraphIv> g_left, g_right, g_newleft, g_newright;
    RvGraphAlignOut g_out;
    int64_t picker_calls = 0, picker_seeded = 0, picker_ns = 0, picker_list_ns = 0, galign_ns = 0;
    // pre-selection for the Python callbacks (rv_set_preselect): record numbers handed out per sub, in emission order
    int64_t presel = 0; bool presel_on = false;
    bool pick_filter = false;      // presel_on for the native pickers' sake: the device-side filter alone, no per-sub-index selection on the host
    int64_t presel_d2h = 0;            // records the scans of this alignment copied to the host while pre-selection was on (RV_PRESEL_LOG)
    std::vector<int64_t> sel, sel_first, sel_tmp;
    // device scratch
    DBuf dD, dTab, dTile, dList, dFlag, dPar, dDbg, scrSA, scrLCP, scrBWT;
    RvCascadeBufs cas;           // scratch of the anchor cascade (rv_cascade.hip)
    RvCascadeOut cas_out{};      // what the last built-in run's cascade did
    DBuf dPbReady; u32 pb_epoch = 0;   // k_pb_shift: per tile of a round, the number of the launch that read it
    DBuf dNextTsub;              // tile -> sub-index of the level being written (rv_tile_sub_launch)
    DBuf dTmin;                  // per RV_SPLIT_TILE ranks of the level being written: lower bound of its LCP values (split -> bubble rounds)
    HBuf hLeafRoots[2], hLeafOut;   // pinned staging: roots per ping-pong slot; counters + anchors of the leaf launches at the end of a run
    DBuf dLeaf, dLeafRoots[2];   // leaf kernel outputs (counters, stats, anchors, trace) and its per-level root tables
    hipStream_t leaf_stream = nullptr, leaf_stream2 = nullptr;   // leaf launches overlap the level pipeline -- and, on alternating streams, each other:
                                                                 // at the deep levels a launch (~180 us) takes as long as a level, and one stream in order made them the critical path
    hipStream_t bub_stream = nullptr, bub_stream2 = nullptr;      // LDS-resident / one-workgroup bubble kernels run here, beside the main stream's rounds
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
    hipEvent_t ev_ready = nullptr, ev_leaf[RV_LEVEL_BUFS + 1] = {}, ev_roots[2] = {nullptr, nullptr};      // ev_leaf[RV_LEVEL_BUFS]: the leaf launch of level 0 (reads the main arrays)
    bool roots_inflight[2] = {false, false};
    bool flag_clean = false;     // dFlag is all zero
    // device-side decisions (rv_decide.hip): tables of the NEXT level shipped with a commit, state of the early split
    DBuf dDec, dErr, dTab0;
    bool next_dev_ok = false, cur_dev_ok = false, early_done = false, early_bubble = false, use_leaf = false;
    int64_t par_min_cur = RV_BUBBLE_PAR_N;
    const sa_t *d_next_nodes = nullptr; const uint8_t *d_next_flags = nullptr; const int *d_next_tsub2 = nullptr;
    std::vector<sa_t> next_nodes; std::vector<uint8_t> next_flags;
    RvLabelTabs e_lt; RvSplitArgs e_sa;
    bool leaf_pending[RV_LEVEL_BUFS + 1] = {};   // a leaf launch may still be reading level buffer k
    size_t leaf_anchor_cap = 0, leaf_trace_cap = 0;
    std::vector<uint8_t> leaf_done;   // per sub of the current level: handed to the leaf kernel
    std::vector<RvLeafRoot> leaf_roots[2];
    Packer pk;
    std::vector<int> stamp; int epoch = 0;        // count_samples scratch
    // reusable host tables of commit()
    std::vector<sa_t> cb, ce, mb, me, cut_lo, cut_hi, mend_pos;
    std::vector<uint8_t> cc;
    std::vector<int> ctab_first, mtab_first, cut_first, mend_first, split_subs;
    std::vector<u32> sub_off_h;
    std::vector<int> tile_sub;
    std::vector<int64_t> mpre, sub_start, woff, toff, next_ss;
    const int64_t *d_next_ss = nullptr;      // device copy of the next level's sub-index starts (inside dTab)
    const int *d_next_want = nullptr;        // ... and of its sub-indices' sample counts
    const int *d_next_tsub = nullptr;        // ... and its tile -> sub-index table (more than two samples)
    std::vector<int> next_tsub;
    std::vector<u32> pick_l; std::vector<sa_t> pick_pos;
    std::vector<u32> child_base, child_n;
    std::vector<RvBubbleDesc> descs;
    std::vector<std::vector<RvBubbleDesc>> rounds;
    std::vector<RvBubbleDesc> kids_small, kids_big, kids_lds;
    struct Kid { int64_t off, n; int c0, c1; int64_t m0, wsum; };
    std::vector<Kid> kid_tmp;
    // results of rv_align_builtin
    std::vector<u32> an_l; std::vector<int64_t> an_off, an_pos;
    bool trace_on = false;
    std::vector<rv_trace> trace;
    rv_align_stats st{};
    double lg[8] = {0}, lgx[8] = {0};          // RV_LEVEL_LOG: host time stamps inside the current level
    // rv_align_builtin split into set-up / levels / collection, so that a frontier can be handed to other devices in between
    int leaf_flip = 0;
    bool leaf_launch_due = false, hook_early = false;   // the level's leaf launch waits until the level's scan / split kernels are queued
    size_t last_leaf_count = 0;
    bool running = false;        // a built-in run is between its set-up and its collection
    u32 *lf_counters = nullptr; unsigned long long *lf_stats = nullptr; u32 *lf_l = nullptr; int64_t *lf_pos = nullptr; rv_trace *lf_tr = nullptr;
    size_t leaf_na = 0;          // anchors of the leaf launches of the last run: they stay in the pinned staging buffer (hLeafOut: pos[2 na], l[na]) until fetched
    void release() {
        // (the side streams may still be writing pinned buffers that go back to the process-wide pool below: after an aborted run nothing else waits for them)
        if (leaf_stream) (void)hipStreamSynchronize(leaf_stream);
        if (leaf_stream2) (void)hipStreamSynchronize(leaf_stream2);
        if (bub_stream) { (void)hipStreamSynchronize(bub_stream); (void)hipStreamSynchronize(bub_stream2); }
        for (int k = 0; k < RV_LEVEL_BUFS; k++) { lvSA[k].release(); lvLCP[k].release(); lvBWT[k].release(); }
        scrSA.release(); scrLCP.release(); scrBWT.release(); cas.release();
        dTmin.release(); dPbReady.release(); dNextTsub.release(); dD.release(); dTab.release(); dTile.release(); dList.release(); dFlag.release(); dPar.release(); dDbg.release(); pk.release(); dDec.release(); dErr.release(); dTab0.release(); hLeafRoots[0].release(); hLeafRoots[1].release(); hLeafOut.release(); dLeaf.release(); dLeafRoots[0].release(); dLeafRoots[1].release();
        if (leaf_stream) { rv_stream_put(leaf_stream); leaf_stream = nullptr; }
        if (leaf_stream2) { rv_stream_put(leaf_stream2); leaf_stream2 = nullptr; }
        if (bub_stream) { rv_stream_put(bub_stream); bub_stream = nullptr; rv_stream_put(bub_stream2); bub_stream2 = nullptr; (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join); (void)hipEventDestroy(ev_join2); ev_fork = ev_join = ev_join2 = nullptr; }
        if (ev_ready) { (void)hipEventDestroy(ev_ready); ev_ready = nullptr; }
        for (int k = 0; k <= RV_LEVEL_BUFS; k++) if (ev_leaf[k]) { (void)hipEventDestroy(ev_leaf[k]); ev_leaf[k] = nullptr; }
        for (int k = 0; k < 2; k++) if (ev_roots[k]) { (void)hipEventDestroy(ev_roots[k]); ev_roots[k] = nullptr; }
    }
};

void rv_align_free(rv_index *h) {
    if (h->al && h->al->feed) { (void)rv_replay_feed_finish(h->al->feed); h->al->feed = nullptr; }
    if (h->al) { h->al->release(); delete h->al; h->al = nullptr; }
}

// leading children above this many ranks take the data-parallel bubble rounds.  Measured crossovers: two samples (one cut per
// sample; C2: 256 K 758, 384 K - 512 K 777, 768 K - 1 M 789, 1.5 M 724 Mbp/s; no difference at 2 x 50 Mbp) 768 K; more samples
// (a cut per sample and child; C3: 184 ms per step against 195 at 512 K) 256 K
// the device-side error word of the recursion (bits: 1 split sizes, 2 bubble search found no landing site, 4 device-side decision outside its
// intervals, 16 a tile of the one-pass shift waited too long for its neighbour) -> message
static int report_dev_err(u32 err) {
    if (err & 1u) { rv_set_error("split: the intervals returned by graphalign do not partition the sub-index (child size mismatch)"); return -1; }
    if (err & 4u) { rv_set_error("device-side decision: a picked match does not lie inside the intervals of its sub-index"); return -1; }
    if (err & 2u) { rv_set_error("bubble_sort (data-parallel rounds): a mover found no landing site"); return -1; }
    if (err & 16u) { rv_set_error("bubble_sort (data-parallel rounds): a tile waited in vain for the tile above it (workgroups not dispatched in index order?)"); return -1; }
    if (err) { rv_set_error("device-side error word %u", err); return -1; }
    return 0;
}

static int64_t bubble_par_default(bool multi) { return multi ? (int64_t)262144 : (int64_t)RV_BUBBLE_PAR_N; }

static const sa_t *cur_sa(rv_index *h) { Align *a = h->al; return a->level == 0 ? h->dSA.as<sa_t>() : a->lvSA[a->cur].as<sa_t>(); }
static const lcp_t *cur_lcp(rv_index *h) { Align *a = h->al; return a->level == 0 ? h->dLCP.as<lcp_t>() : a->lvLCP[a->cur].as<lcp_t>(); }
static const uint8_t *cur_bwt(rv_index *h) { Align *a = h->al; return a->level == 0 ? h->dBWT.as<uint8_t>() : a->lvBWT[a->cur].as<uint8_t>(); }

static int sample_of(const rv_index *h, int64_t pos) {       /* SO[pos], interface.c:116-134 */
    return (int)(std::lower_bound(h->nsep.begin(), h->nsep.end(), pos) - h->nsep.begin());
}

/* child sample count, reveal.c:1028-1042 */
static int count_samples(rv_index *h, const RvIntv *iv, size_t cnt) {
    Align *a = h->al;
    if (h->nsamples > 2) {
        // the intervals are sorted by begin, so their samples are non-decreasing: one walk along the sample boundaries
        bool sorted = true;
        for (size_t k = 1; k < cnt && sorted; k++) sorted = iv[k - 1].begin <= iv[k].begin;
        if (sorted) {
            int ns = 0, last = -1; size_t sp = 0;
            for (size_t k = 0; k < cnt; k++) {
                while (sp < h->nsep.size() && h->nsep[sp] < iv[k].begin) sp++;
                if ((int)sp != last) { ns++; last = (int)sp; }
            }
            return ns;
        }
        if ((int)a->stamp.size() < h->nsamples) a->stamp.assign((size_t)h->nsamples, 0);
        const int ep = ++a->epoch;
        int ns = 0;
        for (size_t k = 0; k < cnt; k++) { const int s = sample_of(h, iv[k].begin); if (a->stamp[(size_t)s] != ep) { a->stamp[(size_t)s] = ep; ns++; } }
        return ns;
    }
    bool f0 = false, f1 = false;
    for (size_t k = 0; k < cnt; k++) { if (iv[k].begin < h->nsep[0]) f0 = true; if (iv[k].begin > h->nsep[0]) f1 = true; }
    return (int)f0 + (int)f1;
}

/* can the sub-index made of these intervals (sorted by begin) still hold a match of minl bases present in each of its samples? */
static bool child_is_dead(const rv_index *h, const RvIntv *iv, size_t cnt, int minl, int minn) {
    const int64_t need = std::max(minl, 1);
    int ns = 0; size_t sp = 0; int last = -1; int64_t best = 0;
    for (size_t k = 0; k < cnt; k++) {
        if (k && iv[k].begin < iv[k - 1].begin) return false;          // (not sorted: no shortcut)
        while (sp < h->nsep.size() && h->nsep[sp] < iv[k].begin) sp++;
        if ((int)sp != last) {
            if (last >= 0 && best < need) return true;
            ns++; last = (int)sp; best = 0;
        }
        best = std::max(best, iv[k].end - iv[k].begin);
    }
    if (last >= 0 && best < need) return true;
    return ns < std::max(minn, 2);
}

static int need_align(rv_index *h) {
    if (!h->al) { rv_set_error("align not started (rv_align_begin)"); return -1; }
    return 0;
}
static int need_sub(rv_index *h, int s) {
    RV_TRY(need_align(h));
    if (s < 0 || s >= h->al->lv.size()) { rv_set_error("sub-index %d out of range", s); return -1; }
    return 0;
}

// record one decision; lead/trail/rest are sorted here, match keeps its order (reveal.c:673-674)
static int add_decision(rv_index *h, int s, u32 l, const int64_t *sp, int nsp,
                        const RvIntv *lead, int nlead, const RvIntv *trail, int ntrail,
                        const RvIntv *match, int nmatch, const RvIntv *rest, int nrest, bool ranges_from_match = false) {
    Decisions &d = h->al->dec;
    if (d.of_sub[(size_t)s] >= 0) { rv_set_error("sub-index %d already has a decision", s); return -1; }
    d.of_sub[(size_t)s] = d.size();
    d.sub.push_back(s); d.l.push_back(l);
    if (ranges_from_match) d.host_lists = true;
    if (ranges_from_match) {      // splitindex / extract: the matched ranges are the matching intervals themselves (reveal.c:1618-1621)
        std::vector<RvIntv> mm(match, match + nmatch);
        std::sort(mm.begin(), mm.end(), [](const RvIntv &x, const RvIntv &y) { return x.begin < y.begin; });
        for (const RvIntv &iv : mm) { d.sp.push_back(iv.begin); d.spl.push_back((u32)(iv.end - iv.begin)); }
    } else {
        d.sp.insert(d.sp.end(), sp, sp + nsp);
        std::sort(d.sp.end() - nsp, d.sp.end());
        d.spl.insert(d.spl.end(), (size_t)nsp, l);
    }
    auto put = [](std::vector<RvIntv> &v, const RvIntv *p, int n, bool sorted) {
        const size_t at = v.size();
        v.insert(v.end(), p, p + n);
        if (sorted && n > 1 && !std::is_sorted(v.begin() + at, v.end(), intv_less)) std::sort(v.begin() + at, v.end(), intv_less);
    };
    put(d.lead, lead, nlead, true); put(d.trail, trail, ntrail, true); put(d.rest, rest, nrest, true); put(d.match, match, nmatch, false);
    d.close();
    return 0;
}

extern "C" {

static int align_begin(rv_index *h, int minl, int minn, bool need_sai);
/* the callback protocol: sub-indices are visible to the caller, the shared inverse (reveal.c:597,609,630) is a getter of theirs */
int rv_align_begin(rv_index *h, int minl, int minn) { return align_begin(h, minl, minn, true); }

static int align_begin(rv_index *h, int minl, int minn, bool need_sai) {
    if (!h->constructed || h->main_arrays_freed) { rv_set_error("Index not yet constructed, alignment stopped."); return -1; }
    if (h->nsamples < 2) { rv_set_error("align needs at least two samples"); return -1; }
    RV_HIP(hipSetDevice(h->device));
    if (need_sai) RV_TRY(rv_need_sai(h));
    const bool keep_trace = h->al && h->al->trace_on;
    if (!h->al) h->al = new Align();         // device scratch of an earlier run is reused
    Align *a = h->al;
    a->trace_on = keep_trace;
    a->minl = minl; a->minn = minn;
    a->multi = h->nsamples > 2;
    a->level = 0; a->cur = 0; a->scanned = false; a->d_err = nullptr; a->full_only = false; a->flag_clean = false;
    a->next_dev_ok = a->cur_dev_ok = a->early_done = a->early_bubble = false; a->use_leaf = false;
    a->presel_on = a->presel > 0;
    a->par_min_cur = (h->ws.opt.bubble_par_min >= 0 ? h->ws.opt.bubble_par_min : bubble_par_default(a->multi));
    RV_TRY(a->dErr.reserve(64));
    RV_HIP(hipMemsetAsync(a->dErr.p, 0, 64, h->ws.stream));
    memset(&a->st, 0, sizeof a->st);
    a->lv.clear();
    a->lv.m = h->n;
    a->lv.off.push_back(0); a->lv.n.push_back(h->n); a->lv.depth.push_back(0); a->lv.nsamples.push_back(h->nsamples);
    a->lv.parent.push_back(-1); a->lv.kind.push_back(0);
    a->lv.nodes = h->nodes;
    std::sort(a->lv.nodes.begin(), a->lv.nodes.end(), intv_less);
    a->lv.node_first.push_back((int64_t)a->lv.nodes.size());
    a->dec.reset(1);
    a->skip_scan.assign(1, 0);
    return 0;
}

int rv_frontier_size(rv_index *h) { return h->al ? h->al->lv.size() : 0; }

int rv_align_end(rv_index *h) {
    if (h->al) {
        if (h->al->presel_on && h->ws.opt.presel_log)
            fprintf(stderr, "preselect: %lld match records copied to the host by the scans of this alignment (%s)\n", (long long)h->al->presel_d2h,
                    h->ws.opt.presel_host ? "every match, filtered on the host" : "the scan kernel keeps only matches present in every sample of their sub-index");
        h->al->presel_d2h = 0;
        h->al->lv.clear(); h->al->scanned = false;
    }
    return 0;
}

int rv_set_trace(rv_index *h, int on) {
    if (!h->al) h->al = new Align();
    h->al->trace_on = on != 0;
    return 0;
}

/* What schemes.graphmumpicker keeps of a scan before it chains (schemes.py:227 the matches present in every sample of the
 * sub-index; :240, :245-247, :287-289 of those the `maxmums` longest -- two stable sorts, so of equal lengths the later
 * emitted ones stay), applied inside the library so that only those cross into Python.  0 = off (the reference's lists). */
int rv_set_picker(rv_index *h, int kind, const rv_picker_args *args) {
    if (!h || kind < 0 || kind > 1 || (kind == 1 && !args)) { rv_set_error("rv_set_picker: bad arguments"); return -1; }
    if (kind == 1 && (args->gcmodel < 0 || args->gcmodel > 2)) { rv_set_error("rv_set_picker: gap cost model 0 (sumofpairs), 1 (star-avg) or 2 (star-med)"); return -1; }
    if (!h->al) h->al = new Align();
    h->al->picker = kind;
    h->al->ggraph = nullptr;
    if (kind == 1) h->al->pargs = *args;
    return 0;
}
/* Picker kind 2: graph inputs.  `g` is the caller's rv_graph of the inputs (rv_graph_import) -- it stays the caller's, is changed by the run (graphalign's
 * surgery per chosen match) and is the alignment graph afterwards (rv_graph_finish / _prune / _gfa).  g == NULL: picker off. */
int rv_set_graph_picker(rv_index *h, rv_graph *g, const rv_picker_args *args) {
    if (!h || (g && !args)) { rv_set_error("rv_set_graph_picker: bad arguments"); return -1; }
    if (g && (args->gcmodel < 0 || args->gcmodel > 2)) { rv_set_error("rv_set_graph_picker: gap cost model 0 (sumofpairs), 1 (star-avg) or 2 (star-med)"); return -1; }
    if (!h->al) h->al = new Align();
    h->al->picker = g ? 2 : 0;
    h->al->ggraph = g;
    if (g) h->al->pargs = *args;
    return 0;
}
/* With picker kind 1: g = rv_graph_replay_begin's graph of the same sequences; the next whole run (rv_align_builtin) applies every level's anchors to it on a host
 * thread while the GPU works on the next level, and returns with g as rv_graph_replay would have left it (the replay was a third of `reveal rem` on five genomes
 * of 5 Mbp, all of it after the run).  One run; g = NULL: off.  Runs that stop at a frontier (rv_align_builtin_until) do not feed it. */
int rv_set_replay_graph(rv_index *h, rv_graph *g) {
    if (!h) { rv_set_error("rv_set_replay_graph: null handle"); return -1; }
    if (!h->al) h->al = new Align();
    if (h->al->feed) { (void)rv_replay_feed_finish(h->al->feed); h->al->feed = nullptr; }
    h->al->replay_graph = g;
    return 0;
}
int rv_picker_info(const rv_index *h, int64_t *out) {
    if (!h || !out) return -1;
    out[0] = h->al ? h->al->picker : 0; out[1] = h->al ? h->al->picker_calls : 0; out[2] = h->al ? h->al->picker_seeded : 0;
    out[3] = h->al ? h->al->picker_ns : 0; out[4] = h->al ? h->al->picker_list_ns : 0; out[5] = h->al ? h->al->galign_ns : 0;
    return 0;
}
int rv_set_preselect(rv_index *h, int64_t maxmums) {
    if (maxmums < 0) { rv_set_error("rv_set_preselect: maxmums must not be negative"); return -1; }
    if (!h->al) h->al = new Align();
    h->al->presel = maxmums;
    return 0;
}

}  // extern "C"

int rv_run_multi_scan(rv_index *h, const sa_t *SA, const lcp_t *LCP, const uint8_t *BWT, int64_t m, int minl, int minn, int mems,
                      std::vector<u32> &l, std::vector<int32_t> &n, std::vector<int64_t> &off, std::vector<uint16_t> &so,
                      std::vector<int64_t> &pos, std::vector<int64_t> *ub_out,
                      const int64_t *d_sub_start, const int *d_sub_want, int nsubs);

extern "C" {

static int early_split(rv_index *h);
static int early_split_multi(rv_index *h);

/* the leaf kernel of the current level (roots prepared by builtin_levels) on its own stream */
static int leaf_launch(rv_index *h) {
    Align *a = h->al;
    a->leaf_launch_due = false;
    const int leaf_flip = a->leaf_flip;
    std::vector<RvLeafRoot> &roots = a->leaf_roots[leaf_flip];
    DBuf &droots = a->dLeafRoots[leaf_flip];
    hipStream_t ls = leaf_flip ? a->leaf_stream2 : a->leaf_stream;
    const int slot = (a->level == 0) ? RV_LEVEL_BUFS : a->cur;
    if (a->leaf_pending[slot]) RV_HIP(hipEventSynchronize(a->ev_leaf[slot]));     // (cannot happen: commit waits first)
    RV_TRY(droots.reserve(roots.size() * sizeof(RvLeafRoot)));
    // pinned staging (one per ping-pong slot): a pageable copy would make the host wait for the leaf stream to drain
    HBuf &hroots = a->hLeafRoots[leaf_flip];
    if (a->roots_inflight[leaf_flip]) RV_HIP(hipEventSynchronize(a->ev_roots[leaf_flip]));     // copy of two levels ago (long done)
    RV_TRY(hroots.reserve(roots.size() * sizeof(RvLeafRoot)));
    memcpy(hroots.p, roots.data(), roots.size() * sizeof(RvLeafRoot));
    RV_HIP(hipMemcpyAsync(droots.p, hroots.p, roots.size() * sizeof(RvLeafRoot), hipMemcpyHostToDevice, ls));
    RV_HIP(hipEventRecord(a->ev_roots[leaf_flip], ls));
    a->roots_inflight[leaf_flip] = true;
    RvLeafArgs la;
    la.roots = droots.as<RvLeafRoot>();
    la.SA = cur_sa(h); la.LCP = cur_lcp(h); la.BWT = cur_bwt(h);
    la.nsep0 = h->nsep[0]; la.minl = a->minl; la.lcap = h->maxlcp;
    la.stage_cap = (u32)h->ws.opt.leaf_acap;
    la.anchor_count = a->lf_counters; la.anchor_cap = (u32)a->leaf_anchor_cap; la.anchor_l = a->lf_l; la.anchor_pos = a->lf_pos;
    la.stats = a->lf_stats;
    la.trace = a->trace_on ? 1 : 0; la.trace_count = a->lf_counters + 1; la.trace_cap = (u32)a->leaf_trace_cap; la.trace_out = a->lf_tr;
    la.err = a->lf_counters + 2;
    {
        Workspace lw; lw.stream = ls;
        RV_TRY(rv_leaf_launch(lw, la, (int)roots.size()));
    }
    RV_HIP(hipEventRecord(a->ev_leaf[slot], ls));
    a->leaf_pending[slot] = true;
    a->leaf_flip ^= 1;
    return 0;
}

/* called by the pair scan once its kernels and the picker's are queued, before the host waits for the picks */
static int level_hook(rv_index *h) {
    Align *a = h->al;
    if (a->hook_early) RV_TRY(early_split(h));
    if (a->leaf_launch_due) RV_TRY(leaf_launch(h));
    return 0;
}

/* The split of the current level (reveal.c:1005-1252 without lower-casing and bubble_sort), queued right behind the picker
 * kernels with decisions taken on the device (rv_decide.hip): it runs while the host receives the picks and rebuilds the
 * same decisions for its own bookkeeping.  rv_frontier_commit then finds it done. */
static int early_split(rv_index *h) {
    Align *a = h->al;
    hipStream_t q = h->ws.stream;
    const Level &lv = a->lv;
    const int ns = lv.size();
    const int64_t m = lv.m, ntiles = ceil_div(m, RV_SPLIT_TILE);
    const int nxt = (a->level == 0) ? 0 : (a->cur + 1) % RV_LEVEL_BUFS;
    RV_TRY(a->dD.reserve((size_t)m + 64));
    RV_TRY(a->dTile.reserve((size_t)ntiles * 3 * 5 * 4 + 64));
    RV_TRY(a->lvSA[nxt].reserve((size_t)(m + 64) * sizeof(sa_t)));
    RV_TRY(a->lvLCP[nxt].reserve((size_t)(m + 64) * sizeof(lcp_t)));
    RV_TRY(a->lvBWT[nxt].reserve((size_t)m + 64));
    // a leaf launch of an earlier level may still be reading the buffer the split writes (ping-pong)
    if (a->leaf_pending[nxt]) { RV_HIP(hipStreamWaitEvent(q, a->ev_leaf[nxt], 0)); a->leaf_pending[nxt] = false; }
    const size_t S = (size_t)ns;
    size_t bytes = 0;
    auto take = [&](size_t b) { const size_t o = (bytes + 15) & ~(size_t)15; bytes = o + b; return o; };
    const size_t o_cb = take(4 * S * sizeof(sa_t)), o_ce = take(4 * S * sizeof(sa_t)), o_cc = take(4 * S), o_ctf = take((S + 1) * 4);
    const size_t o_mb = take(2 * S * sizeof(sa_t)), o_me = take(2 * S * sizeof(sa_t)), o_mtf = take((S + 1) * 4);
    const size_t o_cn = take(3 * S * 4), o_cbase = take(3 * S * 4), o_soff = take(3 * S * 4), o_exp = take(16), o_tot = take(16);
    const size_t o_cf = take((S + 1) * 4), o_mf = take((S + 1) * 4), o_clo = take(2 * S * sizeof(sa_t)), o_chi = take(2 * S * sizeof(sa_t)), o_mp = take(2 * S * sizeof(sa_t));
    const size_t o_kid = take(S * sizeof(RvBubbleDesc));
    RV_TRY(a->dDec.reserve(bytes + 64));
    uint8_t *db = a->dDec.as<uint8_t>();
    RvDecideArgs d;
    d.nsubs = ns; d.lcap = h->maxlcp;
    d.nodes = a->d_next_nodes; d.flags = a->d_next_flags; d.picks = h->hscan.as<RvPairRec>();
    d.ctab_first = (int *)(db + o_ctf); d.mtab_first = (int *)(db + o_mtf);
    d.cb = (sa_t *)(db + o_cb); d.ce = (sa_t *)(db + o_ce); d.cc = db + o_cc; d.mb = (sa_t *)(db + o_mb); d.me = (sa_t *)(db + o_me);
    d.child_n = (u32 *)(db + o_cn); d.child_base = (u32 *)(db + o_cbase); d.sub_off = (u32 *)(db + o_soff); d.expect_total = (u32 *)(db + o_exp);
    d.cut_first = (int *)(db + o_cf); d.mend_first = (int *)(db + o_mf);
    d.cut_lo = (sa_t *)(db + o_clo); d.cut_hi = (sa_t *)(db + o_chi); d.mend_pos = (sa_t *)(db + o_mp);
    d.err = a->dErr.as<u32>();
    d.ovf_cap = (u32)std::min<size_t>(h->ws.misc[4].cap / sizeof(RvPairRec), 0xffffffffu);      // (the scan's overflow buffer)
    d.kid = (RvBubbleDesc *)(db + o_kid);
    RV_TRY(rv_decide_launch(h->ws, d));
    RvLabelTabs &lt = a->e_lt;
    lt.sub_start = a->d_next_ss; lt.nsubs = ns; lt.tile_sub = a->d_next_tsub2;
    lt.ctab_first = d.ctab_first; lt.cbegin = d.cb; lt.cend = d.ce; lt.ccls = d.cc;
    lt.mtab_first = d.mtab_first; lt.mbegin = d.mb; lt.mend = d.me; lt.nmatch = 2 * ns;
    RvSplitArgs &sa = a->e_sa;
    u32 *tiles = a->dTile.as<u32>();
    sa.ntiles = ntiles;
    sa.tile_cnt = tiles; sa.tile_has = tiles + 3 * ntiles; sa.tile_post = tiles + 6 * ntiles;
    sa.tile_G = tiles + 9 * ntiles; sa.tile_carry = tiles + 12 * ntiles;
    sa.total = (u32 *)(db + o_tot);
    sa.sub_start = lt.sub_start; sa.nsubs = ns; sa.tile_sub = lt.tile_sub;
    sa.child_base = d.child_base; sa.child_n = d.child_n; sa.sub_off = d.sub_off; sa.expect_total = d.expect_total;
    sa.cut_first = d.cut_first; sa.cut_lo = d.cut_lo; sa.cut_hi = d.cut_hi;
    sa.mend_first = d.mend_first; sa.mend_pos = d.mend_pos;
    sa.SA_out = a->lvSA[nxt].as<sa_t>(); sa.LCP_out = a->lvLCP[nxt].as<lcp_t>(); sa.BWT_out = a->lvBWT[nxt].as<uint8_t>(); sa.SAi = h->dSAi.as<sa_t>();
    sa.err = a->dErr.as<u32>();
    {   // tile bounds for the data-parallel bubble rounds: only a level that can still have a child above their threshold needs them
        int64_t big0 = 0;
        for (int s2 = 0; s2 < ns; s2++) big0 = std::max<int64_t>(big0, lv.n[(size_t)s2]);
        sa.tmin_out = nullptr;
        if (big0 > a->par_min_cur) {
            RV_TRY(a->dTmin.reserve((size_t)(m / RV_SPLIT_TILE + 2) * 4));      // (the next level is not larger than this one)
            RV_HIP(hipMemsetAsync(a->dTmin.p, 0xFF, (size_t)(m / RV_SPLIT_TILE + 2) * 4, q));
            sa.tmin_out = a->dTmin.as<u32>();
        }
    }
    int id = h->prof.begin(q, RV_K_SPLIT, (double)m * (2 * (sizeof(sa_t) + sizeof(lcp_t) + 2)) + (double)m * (sizeof(sa_t) + sizeof(lcp_t) + 1));
    RV_TRY(rv_split_launch(h->ws, cur_sa(h), cur_lcp(h), a->dD.as<uint8_t>(), cur_bwt(h), m, lt, sa, 1));
    h->prof.end(q, id);
    a->early_done = true;
    // Few sub-indices (above a few thousand the LDS kernels on their own stream are the better choice): lower-casing and the bubble of every leading child the rounds
    // do not take follow at once, too -- the whole level except those rounds is then queued before the host has seen the picks.
    a->early_bubble = false;
    int64_t biggest = 0;
    for (int s2 = 0; s2 < ns; s2++) biggest = std::max<int64_t>(biggest, lv.n[(size_t)s2]);
    // (a level that still has a sub-index above the rounds' threshold keeps the host-built mix of rounds and joined children)
    // (ns > 4096 through the size-class launches below: tried at 2 x 250 Mbp, 304 against 300 ms -- five launches over every
    // descriptor cost more GPU time than the hidden host time is worth; RV_EARLY_BUBBLE_MANY=1 switches it on)
    if (biggest <= a->par_min_cur && !h->ws.opt.bubble_lds_always && !h->ws.opt.no_early_bubble && (ns <= 4096 || h->ws.opt.early_bubble_many)) {
        RV_TRY(rv_lower_ranges_launch(h->ws, h->dT.as<uint8_t>(), d.mb, d.me, 2 * ns));
        {
            const void *before = a->dFlag.p;
            RV_TRY(a->dFlag.reserve((size_t)m + 64));
            if (!a->flag_clean || a->dFlag.p != before) { RV_HIP(hipMemsetAsync(a->dFlag.p, 0, a->dFlag.cap, q)); a->flag_clean = true; }
        }
        RvBubbleArgs ba;
        memset(&ba, 0, sizeof ba);
        ba.flag = a->dFlag.as<uint8_t>();
        ba.SA = sa.SA_out; ba.LCP = sa.LCP_out; ba.BWT = sa.BWT_out; ba.SAi = sa.SAi; ba.cut_lo = d.cut_lo; ba.cut_hi = d.cut_hi; ba.err = a->dErr.as<u32>();
        id = h->prof.begin(q, RV_K_BUBBLE, 0.0);
        if (ns <= 4096) {
            RV_TRY(rv_bubble_children_dev_launch(h->ws, ba, d.kid, ns, a->par_min_cur));
        } else {
            // thousands of sub-indices (the deep levels of large inputs): the size-class kernels, LDS-resident ones included, each on
            // its own stream -- the host-built launches of rv_frontier_commit, minus the wait for the host's tables
            if (!a->bub_stream) {
                RV_HIP(rv_stream_get(&a->bub_stream));
                RV_HIP(rv_stream_get(&a->bub_stream2));
                RV_HIP(hipEventCreateWithFlags(&a->ev_fork, hipEventDisableTiming));
                RV_HIP(hipEventCreateWithFlags(&a->ev_join, hipEventDisableTiming));
                RV_HIP(hipEventCreateWithFlags(&a->ev_join2, hipEventDisableTiming));
            }
            RV_HIP(hipEventRecord(a->ev_fork, q));
            RV_HIP(hipStreamWaitEvent(a->bub_stream, a->ev_fork, 0));
            RV_HIP(hipStreamWaitEvent(a->bub_stream2, a->ev_fork, 0));
            Workspace wl, wk; wl.stream = a->bub_stream; wk.stream = a->bub_stream2;
            RV_TRY(rv_bubble_children_dev_classes_launch(wl, wk, ba, d.kid, ns, a->par_min_cur));
            RV_HIP(hipEventRecord(a->ev_join, a->bub_stream));
            RV_HIP(hipEventRecord(a->ev_join2, a->bub_stream2));
            RV_HIP(hipStreamWaitEvent(q, a->ev_join, 0));
            RV_HIP(hipStreamWaitEvent(q, a->ev_join2, 0));
        }
        h->prof.end(q, id);
        a->early_bubble = true;
    }
    return 0;
}

/* The same for more than two samples: decisions of the built-in callbacks on the device (rv_decide.hip, k_decide_multi) right behind
 * the multi-sample picker, then the level's split -- it runs while the host receives the picks and rebuilds the same decisions
 * for its own bookkeeping (thousands of sub-indices per level with ten samples: 30 of 163 ms at 10 x 5 Mbp with no kernel running). */
static int early_split_multi(rv_index *h) {
    Align *a = h->al;
    hipStream_t q = h->ws.stream;
    const Level &lv = a->lv;
    const int ns = lv.size(), W = h->nsamples;
    const int64_t m = lv.m, ntiles = ceil_div(m, RV_SPLIT_TILE);
    const int nxt = (a->level == 0) ? 0 : (a->cur + 1) % RV_LEVEL_BUFS;
    RV_TRY(a->dD.reserve((size_t)m + 64));
    RV_TRY(a->dTile.reserve((size_t)ntiles * 3 * 5 * 4 + 64));
    RV_TRY(a->lvSA[nxt].reserve((size_t)(m + 64) * sizeof(sa_t)));
    RV_TRY(a->lvLCP[nxt].reserve((size_t)(m + 64) * sizeof(lcp_t)));
    RV_TRY(a->lvBWT[nxt].reserve((size_t)m + 64));
    const size_t S = (size_t)ns;
    size_t bytes = 0;
    auto take = [&](size_t b) { const size_t o = (bytes + 15) & ~(size_t)15; bytes = o + b; return o; };
    const size_t o_cb = take(2 * W * S * sizeof(sa_t)), o_ce = take(2 * W * S * sizeof(sa_t)), o_cc = take(2 * W * S), o_ctf = take((S + 1) * 4);
    const size_t o_mb = take(W * S * sizeof(sa_t)), o_me = take(W * S * sizeof(sa_t)), o_mtf = take((S + 1) * 4);
    const size_t o_cn = take(3 * S * 4), o_cbase = take(3 * S * 4), o_soff = take(3 * S * 4), o_exp = take(16), o_tot = take(16);
    const size_t o_cf = take((S + 1) * 4), o_mf = take((S + 1) * 4), o_clo = take(W * S * sizeof(sa_t)), o_chi = take(W * S * sizeof(sa_t)), o_mp = take(W * S * sizeof(sa_t));
    const size_t o_kid = take(S * sizeof(RvBubbleDesc));
    RV_TRY(a->dDec.reserve(bytes + 64));
    uint8_t *db = a->dDec.as<uint8_t>();
    RvDecideMultiArgs d;
    d.kid = (RvBubbleDesc *)(db + o_kid);
    d.nsubs = ns; d.W = W; d.minl = a->minl; d.minn = a->minn; d.lcap = h->maxlcp;
    d.nodes = a->d_next_nodes; d.want = a->d_next_want;
    // the picker's device buffers (rv_run_multi_pick, rv_api.hip)
    d.pick_l = h->ws.misc[13].as<u32>(); d.pick_pos = h->ws.misc[7].as<sa_t>();
    d.cand_count = h->ws.misc[1].as<u32>() + RV_MULTI_REGIONS * 64;
    d.cand_cap = (u32)std::min<size_t>(h->ws.misc[8].cap / RV_MULTI_CAND_BYTES / RV_MULTI_REGIONS, 0xffffffffu);
    d.ctab_first = (int *)(db + o_ctf); d.mtab_first = (int *)(db + o_mtf); d.cut_first = (int *)(db + o_cf); d.mend_first = (int *)(db + o_mf);
    d.cb = (sa_t *)(db + o_cb); d.ce = (sa_t *)(db + o_ce); d.cc = db + o_cc; d.mb = (sa_t *)(db + o_mb); d.me = (sa_t *)(db + o_me);
    d.cut_lo = (sa_t *)(db + o_clo); d.cut_hi = (sa_t *)(db + o_chi); d.mend_pos = (sa_t *)(db + o_mp);
    d.child_n = (u32 *)(db + o_cn); d.child_base = (u32 *)(db + o_cbase); d.sub_off = (u32 *)(db + o_soff); d.expect_total = (u32 *)(db + o_exp);
    d.err = a->dErr.as<u32>();
    RV_TRY(rv_decide_multi_launch(h->ws, d));
    RvLabelTabs &lt = a->e_lt;
    lt.sub_start = a->d_next_ss; lt.nsubs = ns; lt.tile_sub = a->d_next_tsub;
    lt.ctab_first = d.ctab_first; lt.cbegin = d.cb; lt.cend = d.ce; lt.ccls = d.cc;
    lt.mtab_first = d.mtab_first; lt.mbegin = d.mb; lt.mend = d.me; lt.nmatch = W * ns;
    RvSplitArgs &sa = a->e_sa;
    u32 *tiles = a->dTile.as<u32>();
    sa.ntiles = ntiles;
    sa.tile_cnt = tiles; sa.tile_has = tiles + 3 * ntiles; sa.tile_post = tiles + 6 * ntiles;
    sa.tile_G = tiles + 9 * ntiles; sa.tile_carry = tiles + 12 * ntiles;
    sa.total = (u32 *)(db + o_tot);
    sa.sub_start = lt.sub_start; sa.nsubs = ns; sa.tile_sub = lt.tile_sub;
    sa.child_base = d.child_base; sa.child_n = d.child_n; sa.sub_off = d.sub_off; sa.expect_total = d.expect_total;
    sa.cut_first = d.cut_first; sa.cut_lo = d.cut_lo; sa.cut_hi = d.cut_hi;
    sa.mend_first = d.mend_first; sa.mend_pos = d.mend_pos; sa.mend_all = 0;
    sa.SA_out = a->lvSA[nxt].as<sa_t>(); sa.LCP_out = a->lvLCP[nxt].as<lcp_t>(); sa.BWT_out = a->lvBWT[nxt].as<uint8_t>(); sa.SAi = h->dSAi.as<sa_t>();
    sa.err = a->dErr.as<u32>();
    {   // tile bounds for the data-parallel bubble rounds (see early_split)
        int64_t big0 = 0;
        for (int s2 = 0; s2 < ns; s2++) big0 = std::max<int64_t>(big0, lv.n[(size_t)s2]);
        sa.tmin_out = nullptr;
        if (big0 > a->par_min_cur) {
            RV_TRY(a->dTmin.reserve((size_t)(m / RV_SPLIT_TILE + 2) * 4));
            RV_HIP(hipMemsetAsync(a->dTmin.p, 0xFF, (size_t)(m / RV_SPLIT_TILE + 2) * 4, q));
            sa.tmin_out = a->dTmin.as<u32>();
        }
    }
    int id = h->prof.begin(q, RV_K_SPLIT, (double)m * (2 * (sizeof(sa_t) + sizeof(lcp_t) + 2)) + (double)m * (sizeof(sa_t) + sizeof(lcp_t) + 1));
    RV_TRY(rv_split_launch(h->ws, cur_sa(h), cur_lcp(h), a->dD.as<uint8_t>(), cur_bwt(h), m, lt, sa, 1));
    h->prof.end(q, id);
    a->early_done = true;
    a->early_bubble = false;
    // No sub-index above the rounds' threshold: lower-casing and the bubble of every leading child follow at once, from the
    // device-built descriptors (all size classes: LDS kernels and one-workgroup kernels on their own streams, as the host-built
    // launches of rv_frontier_commit) -- the whole level is queued before the host has seen the picks.
    int64_t biggest = 0;
    for (int s2 = 0; s2 < ns; s2++) biggest = std::max<int64_t>(biggest, lv.n[(size_t)s2]);
    if (biggest <= a->par_min_cur && !h->ws.opt.no_early_bubble) {
        RV_TRY(rv_lower_ranges_launch(h->ws, h->dT.as<uint8_t>(), d.mb, d.me, W * ns));
        {
            const void *before = a->dFlag.p;
            RV_TRY(a->dFlag.reserve((size_t)m + 64));
            if (!a->flag_clean || a->dFlag.p != before) { RV_HIP(hipMemsetAsync(a->dFlag.p, 0, a->dFlag.cap, q)); a->flag_clean = true; }
        }
        RvBubbleArgs ba;
        memset(&ba, 0, sizeof ba);
        ba.flag = a->dFlag.as<uint8_t>();
        ba.SA = sa.SA_out; ba.LCP = sa.LCP_out; ba.BWT = sa.BWT_out; ba.SAi = sa.SAi; ba.cut_lo = d.cut_lo; ba.cut_hi = d.cut_hi; ba.err = a->dErr.as<u32>();
        if (!a->bub_stream) {
            RV_HIP(rv_stream_get(&a->bub_stream));
            RV_HIP(rv_stream_get(&a->bub_stream2));
            RV_HIP(hipEventCreateWithFlags(&a->ev_fork, hipEventDisableTiming));
            RV_HIP(hipEventCreateWithFlags(&a->ev_join, hipEventDisableTiming));
            RV_HIP(hipEventCreateWithFlags(&a->ev_join2, hipEventDisableTiming));
        }
        id = h->prof.begin(q, RV_K_BUBBLE, 0.0);
        RV_HIP(hipEventRecord(a->ev_fork, q));
        RV_HIP(hipStreamWaitEvent(a->bub_stream, a->ev_fork, 0));
        RV_HIP(hipStreamWaitEvent(a->bub_stream2, a->ev_fork, 0));
        Workspace wl, wk; wl.stream = a->bub_stream; wk.stream = a->bub_stream2;
        RV_TRY(rv_bubble_children_dev_classes_launch(wl, wk, ba, d.kid, ns, a->par_min_cur));
        RV_HIP(hipEventRecord(a->ev_join, a->bub_stream));
        RV_HIP(hipEventRecord(a->ev_join2, a->bub_stream2));
        RV_HIP(hipStreamWaitEvent(q, a->ev_join, 0));
        RV_HIP(hipStreamWaitEvent(q, a->ev_join2, 0));
        h->prof.end(q, id);
        a->early_bubble = true;
    }
    return 0;
}

/* rv_set_preselect: per sub-index the record numbers that go to mumpicker.  No match in every sample (schemes.py:229-232
 * goes on with segment() over all of them): the whole list, uncapped. */
static void build_preselection(Align *a) {
    const int ns = a->lv.size();
    a->sel.clear(); a->sel_first.assign((size_t)ns + 1, 0);
    std::vector<int64_t> &tmp = a->sel_tmp;
    for (int s = 0; s < ns; s++) {
        const int64_t first = a->mum_first[(size_t)s], cnt = a->nmums[(size_t)s];
        tmp.clear();
        if (a->multi) {
            const int32_t want = a->lv.nsamples[(size_t)s];
            for (int64_t k = first; k < first + cnt; k++) if (a->mn[(size_t)k] == want) tmp.push_back(k);
        } else
            for (int64_t k = first; k < first + cnt; k++) tmp.push_back(k);
        if (tmp.empty()) for (int64_t k = first; k < first + cnt; k++) tmp.push_back(k);
        else if ((int64_t)tmp.size() > a->presel) {
            const size_t drop = tmp.size() - (size_t)a->presel;
            auto below = [a](int64_t x, int64_t y) {
                const u32 lx = a->multi ? a->ml[(size_t)x] : a->recs[(size_t)x].l, ly = a->multi ? a->ml[(size_t)y] : a->recs[(size_t)y].l;
                return lx < ly || (lx == ly && x < y);
            };
            std::nth_element(tmp.begin(), tmp.begin() + (ptrdiff_t)drop, tmp.end(), below);
            tmp.erase(tmp.begin(), tmp.begin() + (ptrdiff_t)drop);
            std::sort(tmp.begin(), tmp.end());
        }
        a->sel.insert(a->sel.end(), tmp.begin(), tmp.end());
        a->sel_first[(size_t)s + 1] = (int64_t)a->sel.size();
    }
}

/* reveal.c:802-822 for every sub-index of the frontier */
int rv_frontier_scan(rv_index *h) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    RV_HIP(hipSetDevice(h->device));
    const double t0 = now_s();
    const int ns = a->lv.size();
    a->mum_first.assign((size_t)ns, 0); a->nmums.assign((size_t)ns, 0);
    bool all_skipped = ns > 0 && (int)a->skip_scan.size() == ns && !a->full_only;
    for (int s2 = 0; s2 < ns && all_skipped; s2++) all_skipped = a->skip_scan[(size_t)s2] != 0;
    if (all_skipped) {       // every sub-index of the level brings its own matches: no launch at all (the deferred error word is read here instead)
        a->recs.clear(); a->ml.clear(); a->mn.clear(); a->moff.assign(1, 0); a->mso.clear(); a->mpos.clear();
        if (a->d_err) {
            u32 err = 0;
            RV_HIP(hipMemcpyAsync(&err, a->dErr.p, 4, hipMemcpyDeviceToHost, h->ws.stream));
            RV_HIP(hipStreamSynchronize(h->ws.stream));
            RV_TRY(report_dev_err(err));
        }
    } else if (!a->multi) {
        u32 err = 0;
        // built-in picker without tracing: the device keeps only the record the picker would take.  The sub-index starts of this
        // level came with the previous commit's table upload (a pageable H2D copy here would wait for the stream to drain and
        // expose the launch latency of the whole scan).
        const int64_t *d_ss = (a->full_only && (a->level > 0 || a->cur_dev_ok)) ? a->d_next_ss : nullptr;
        a->early_done = false; a->early_bubble = false;
        const bool early = d_ss && a->cur_dev_ok && !h->ws.opt.no_early_split;
        a->hook_early = early;
        if (early) RV_TRY(h->hscan.reserve((size_t)(ns + RV_PAIR_HDR) * sizeof(RvPairRec)));      // (the hook needs the final address of the picks)
        // rv_set_preselect with Python callbacks: the `presel` longest matches of every sub-index are chosen on the device when the level holds
        // many (RV_PRESEL_DEV_MIN records; the host caps what arrives otherwise -- build_preselection, which leaves a chosen list as it is)
        const int64_t *d_ps = nullptr;
        if (a->presel_on && !a->full_only && ns > 0 && !h->ws.opt.presel_host) {
            Packer &pk = a->pk;
            // (the staging buffer still feeds the previous commit's table upload -- a kernel that reads this pinned memory, queued and not waited for:
            //  rewriting it here raced with that kernel.  Seen once in ~15 000 cases of tools/fuzz_preselect.py as "child size mismatch" or a memory fault,
            //  in the round-4 code as in this one)
            RV_HIP(hipStreamSynchronize(h->ws.stream));
            pk.clear();
            std::vector<int64_t> ss(a->lv.off); ss.push_back(a->lv.m);
            const size_t o1 = pk.addv(ss);
            DBuf &buf = h->ws.misc[10];
            RV_TRY(buf.reserve(pk.size() + 64));
            RV_HIP(hipMemcpyAsync(buf.p, pk.data(), pk.size(), hipMemcpyHostToDevice, h->ws.stream));
            d_ps = (const int64_t *)(buf.as<uint8_t>() + o1);
        }
        RV_TRY(rv_run_pair_scan(h, cur_sa(h), cur_lcp(h), cur_bwt(h), a->lv.m, a->minl, a->recs, a->dErr.as<u32>(), &err, d_ss, ns, level_hook, early || a->leaf_launch_due,
                                (d_ss && a->cur_dev_ok) ? a->d_next_tsub2 : nullptr, d_ps, ns, a->presel));
        if (a->presel_on && !a->full_only) a->presel_d2h += (int64_t)a->recs.size();
        RV_TRY(report_dev_err(err));
        int si = 0;
        for (size_t k = 0; k < a->recs.size(); k++) {
            const int64_t r = (int64_t)a->recs[k].rank;
            while (si < ns && r >= a->lv.off[(size_t)si] + a->lv.n[(size_t)si]) si++;
            if (si >= ns) { rv_set_error("scan record outside the frontier"); return -1; }
            if (a->nmums[(size_t)si] == 0) a->mum_first[(size_t)si] = (int64_t)k;
            a->nmums[(size_t)si]++;
        }
    } else {
        if (a->full_only && a->level > 0) {
            // built-in picker without tracing: the device returns, per sub-index, the match the picker would take (the tables
            // came with the previous commit's upload)
            a->early_done = false; a->early_bubble = false;
            const bool early = a->cur_dev_ok && !h->ws.opt.no_early_split;
            bool redo = false;
            RV_TRY(rv_run_multi_pick(h, cur_sa(h), cur_lcp(h), cur_bwt(h), a->lv.m, a->minl, a->minn, a->d_next_ss, a->d_next_want, ns, a->d_next_tsub, a->pick_l, a->pick_pos,
                                     early ? early_split_multi : nullptr, &redo));
            if (redo) a->early_done = false;      // (the early split saw an overflowed candidate list and decided nothing: the commit splits again)
            a->ml.clear(); a->mn.clear(); a->moff.assign(1, 0); a->mso.clear(); a->mpos.clear();
            const int W = h->nsamples;
            for (int s2 = 0; s2 < ns; s2++) {
                if (a->pick_l[(size_t)s2] == 0) continue;
                const int want = a->lv.nsamples[(size_t)s2];
                a->mum_first[(size_t)s2] = (int64_t)a->ml.size(); a->nmums[(size_t)s2] = 1;
                a->ml.push_back(a->pick_l[(size_t)s2]); a->mn.push_back(want);
                for (int k = 0; k < want; k++) {
                    const int64_t p = (int64_t)a->pick_pos[(size_t)s2 * W + k];
                    a->mso.push_back((uint16_t)sample_of(h, p)); a->mpos.push_back(p);
                }
                a->moff.push_back((int64_t)a->mpos.size());
            }
        } else {
        std::vector<int64_t> ub;
        const int64_t *d_ss = nullptr; const int *d_want = nullptr;
        if (a->full_only) {       // (level 0: the device tables of the picker are not there yet) keep only matches present in every sample of their sub-index
            Packer &pk = a->pk;
            pk.clear();
            std::vector<int64_t> ss(a->lv.off); ss.push_back(a->lv.m);
            const size_t o1 = pk.addv(ss), o2 = pk.addv(a->lv.nsamples);
            DBuf &buf = h->ws.misc[10];
            RV_TRY(buf.reserve(pk.size() + 64));
            RV_HIP(hipMemcpyAsync(buf.p, pk.data(), pk.size(), hipMemcpyHostToDevice, h->ws.stream));
            d_ss = (const int64_t *)(buf.as<uint8_t>() + o1); d_want = (const int *)(buf.as<uint8_t>() + o2);
        }
        const bool dev_filter = a->presel_on && !a->full_only && ns > 0 && !h->ws.opt.presel_host;
        if (dev_filter) {
            // rv_set_preselect: the picker only ever sees the matches present in every sample of their sub-index (schemes.py:227) --
            // the scan kernel drops the others, they never cross into host memory.  A sub-index without such a match keeps its whole list
            // (schemes.py:229-232): those sub-indices, and only those, are scanned a second time without the filter.
            Packer &pk = a->pk;
            RV_HIP(hipStreamSynchronize(h->ws.stream));      // (the previous commit's upload reads this buffer: see the pair branch above)
            pk.clear();
            std::vector<int64_t> ss(a->lv.off); ss.push_back(a->lv.m);
            const size_t o1 = pk.addv(ss), o2 = pk.addv(a->lv.nsamples);
            DBuf &buf = h->ws.misc[10];
            RV_TRY(buf.reserve(pk.size() + 64));
            RV_HIP(hipMemcpyAsync(buf.p, pk.data(), pk.size(), hipMemcpyHostToDevice, h->ws.stream));
            RV_TRY(rv_run_multi_scan(h, cur_sa(h), cur_lcp(h), cur_bwt(h), a->lv.m, a->minl, a->minn, 0, a->ml, a->mn, a->moff, a->mso, a->mpos, &ub,
                                     (const int64_t *)(buf.as<uint8_t>() + o1), (const int *)(buf.as<uint8_t>() + o2), ns));
            a->presel_d2h += (int64_t)a->ml.size();
            // which sub-indices came back empty?
            std::vector<int32_t> want2((size_t)ns, -1);
            {
                std::vector<char> has((size_t)ns, 0);
                int si = 0;
                for (size_t k = 0; k < a->ml.size(); k++) {
                    while (si < ns && ub[k] >= a->lv.off[(size_t)si] + a->lv.n[(size_t)si]) si++;
                    if (si >= ns) { rv_set_error("scan record outside the frontier"); return -1; }
                    has[(size_t)si] = 1;
                }
                bool any = false;
                for (int s2 = 0; s2 < ns; s2++)
                    if (!has[(size_t)s2] && !(s2 < (int)a->skip_scan.size() && a->skip_scan[(size_t)s2])) { want2[(size_t)s2] = 0; any = true; }
                if (any) {
                    std::vector<u32> l2; std::vector<int32_t> n2; std::vector<int64_t> off2, pos2, ub2; std::vector<uint16_t> so2;
                    pk.clear();
                    const size_t p1 = pk.addv(ss), p2 = pk.addv(want2);
                    RV_HIP(hipStreamSynchronize(h->ws.stream));      // (the first upload has been read)
                    RV_TRY(buf.reserve(pk.size() + 64));
                    RV_HIP(hipMemcpyAsync(buf.p, pk.data(), pk.size(), hipMemcpyHostToDevice, h->ws.stream));
                    RV_TRY(rv_run_multi_scan(h, cur_sa(h), cur_lcp(h), cur_bwt(h), a->lv.m, a->minl, a->minn, 0, l2, n2, off2, so2, pos2, &ub2,
                                             (const int64_t *)(buf.as<uint8_t>() + p1), (const int *)(buf.as<uint8_t>() + p2), ns));
                    a->presel_d2h += (int64_t)l2.size();
                    // a sub-index' records come from one of the two passes: merge by the records' upper rank
                    std::vector<u32> ml; std::vector<int32_t> mn; std::vector<int64_t> moff(1, 0), mpos, mub; std::vector<uint16_t> mso;
                    size_t ia = 0, ib = 0;
                    while (ia < a->ml.size() || ib < l2.size()) {
                        const bool from_a = ib >= l2.size() || (ia < a->ml.size() && ub[ia] < ub2[ib]);
                        if (from_a) {
                            ml.push_back(a->ml[ia]); mn.push_back(a->mn[ia]); mub.push_back(ub[ia]);
                            for (int64_t q2 = a->moff[ia]; q2 < a->moff[ia + 1]; q2++) { mso.push_back(a->mso[(size_t)q2]); mpos.push_back(a->mpos[(size_t)q2]); }
                            ia++;
                        } else {
                            ml.push_back(l2[ib]); mn.push_back(n2[ib]); mub.push_back(ub2[ib]);
                            for (int64_t q2 = off2[ib]; q2 < off2[ib + 1]; q2++) { mso.push_back(so2[(size_t)q2]); mpos.push_back(pos2[(size_t)q2]); }
                            ib++;
                        }
                        moff.push_back((int64_t)mpos.size());
                    }
                    a->ml.swap(ml); a->mn.swap(mn); a->moff.swap(moff); a->mso.swap(mso); a->mpos.swap(mpos); ub.swap(mub);
                }
            }
        } else {
        RV_TRY(rv_run_multi_scan(h, cur_sa(h), cur_lcp(h), cur_bwt(h), a->lv.m, a->minl, a->minn, 0, a->ml, a->mn, a->moff, a->mso, a->mpos, &ub,
                                 d_ss, d_want, ns));
        if (a->presel_on && !a->full_only) a->presel_d2h += (int64_t)a->ml.size();
        }
        int si = 0;
        for (size_t k = 0; k < a->ml.size(); k++) {
            while (si < ns && ub[k] >= a->lv.off[(size_t)si] + a->lv.n[(size_t)si]) si++;
            if (si >= ns) { rv_set_error("scan record outside the frontier"); return -1; }
            if (a->nmums[(size_t)si] == 0) a->mum_first[(size_t)si] = (int64_t)k;
            a->nmums[(size_t)si]++;
        }
        }
    }
    for (int s2 = 0; s2 < ns && s2 < (int)a->skip_scan.size(); s2++)
        if (a->skip_scan[(size_t)s2]) { a->nmums[(size_t)s2] = 0; a->mum_first[(size_t)s2] = 0; }
    a->scanned = true;
    if (a->presel_on && !a->full_only && !a->pick_filter) build_preselection(a);
    a->st.scanned_ranks += a->lv.m;
    a->st.t_scan += now_s() - t0;
    return 0;
}

/* Sub-index s of the new frontier was seeded by its parent's mumpicker (non-empty skipmums, reveal.c:1157, 1180): the reference
 * does not scan such an index (reveal.c:802 `if (PyList_Size(idx->skipmums)==0)` ... else :830-837 uses the list).  To be called
 * between the commit that created s and the next rv_frontier_scan; the scan then reports no matches for s, and a level made
 * of seeded sub-indices only is not scanned at all. */
int rv_sub_skip_scan(rv_index *h, int s) {
    RV_TRY(need_sub(h, s));
    Align *a = h->al;
    if (a->scanned) { rv_set_error("rv_sub_skip_scan after rv_frontier_scan"); return -1; }
    if ((int)a->skip_scan.size() != a->lv.size()) a->skip_scan.assign((size_t)a->lv.size(), 0);
    a->skip_scan[(size_t)s] = 1;
    return 0;
}

int rv_sub_info(rv_index *h, int s, rv_sub *out) {
    RV_TRY(need_sub(h, s));
    Align *a = h->al;
    memset(out, 0, sizeof *out);
    out->n = a->lv.n[(size_t)s]; out->depth = a->lv.depth[(size_t)s]; out->nsamples = a->lv.nsamples[(size_t)s];
    out->nnodes = (int32_t)(a->lv.node_first[(size_t)s + 1] - a->lv.node_first[(size_t)s]);
    out->parent = a->lv.parent[(size_t)s]; out->kind = a->lv.kind[(size_t)s];
    if (a->scanned && a->presel_on && !a->full_only) {
        out->nmums = a->sel_first[(size_t)s + 1] - a->sel_first[(size_t)s];
        if (!a->multi) out->nmembers = 2 * out->nmums;
        else for (int64_t q = a->sel_first[(size_t)s]; q < a->sel_first[(size_t)s + 1]; q++) { const size_t g = (size_t)a->sel[(size_t)q]; out->nmembers += a->moff[g + 1] - a->moff[g]; }
    } else if (a->scanned) {
        out->nmums = a->nmums[(size_t)s];
        if (!a->multi) out->nmembers = 2 * out->nmums;
        else out->nmembers = out->nmums ? a->moff[(size_t)(a->mum_first[(size_t)s] + out->nmums)] - a->moff[(size_t)a->mum_first[(size_t)s]] : 0;
    }
    return 0;
}

int rv_sub_nodes(rv_index *h, int s, int64_t *be) {
    RV_TRY(need_sub(h, s));
    Align *a = h->al;
    int64_t k = 0;
    for (int64_t q = a->lv.node_first[(size_t)s]; q < a->lv.node_first[(size_t)s + 1]; q++, k++) { be[2 * k] = a->lv.nodes[(size_t)q].begin; be[2 * k + 1] = a->lv.nodes[(size_t)q].end; }
    return 0;
}

int rv_sub_mums(rv_index *h, int s, uint32_t *l, int32_t *n, int64_t *off, uint16_t *so, int64_t *pos) {
    RV_TRY(need_sub(h, s));
    Align *a = h->al;
    if (!a->scanned) { rv_set_error("rv_sub_mums before rv_frontier_scan"); return -1; }
    if (a->presel_on && !a->full_only) {                      /* the pre-selected records only (rv_set_preselect) */
        int64_t k = 0, w = 0;
        for (int64_t q = a->sel_first[(size_t)s]; q < a->sel_first[(size_t)s + 1]; q++, k++) {
            const size_t g = (size_t)a->sel[(size_t)q];
            off[k] = w;
            if (!a->multi) {
                const RvPairRec &r = a->recs[g];
                l[k] = r.l; n[k] = 2;
                so[w] = 0; pos[w++] = r.a; so[w] = 1; pos[w++] = r.b;
            } else {
                l[k] = a->ml[g]; n[k] = a->mn[g];
                for (int64_t m = a->moff[g]; m < a->moff[g + 1]; m++) { so[w] = a->mso[(size_t)m]; pos[w++] = a->mpos[(size_t)m]; }
            }
        }
        off[k] = w;
        return 0;
    }
    const int64_t first = a->mum_first[(size_t)s], cnt = a->nmums[(size_t)s];
    if (!a->multi) {                                          /* (l, 2, ((0,a),(1,b)))  reveal.c:166-170 */
        for (int64_t k = 0; k < cnt; k++) {
            const RvPairRec &r = a->recs[(size_t)(first + k)];
            l[k] = r.l; n[k] = 2; off[k] = 2 * k;
            so[2 * k] = 0; pos[2 * k] = r.a; so[2 * k + 1] = 1; pos[2 * k + 1] = r.b;
        }
        off[cnt] = 2 * cnt;
        return 0;
    }
    const int64_t base = cnt ? a->moff[(size_t)first] : 0;
    for (int64_t k = 0; k < cnt; k++) {
        const size_t g = (size_t)(first + k);
        l[k] = a->ml[g]; n[k] = a->mn[g]; off[k] = a->moff[g] - base;
        for (int64_t q = a->moff[g]; q < a->moff[g + 1]; q++) { so[q - base] = a->mso[(size_t)q]; pos[q - base] = a->mpos[(size_t)q]; }
    }
    off[cnt] = cnt ? a->moff[(size_t)(first + cnt)] - base : 0;
    return 0;
}

int64_t rv_sub_array(rv_index *h, int s, int which, void *out, int64_t cap) {
    if (need_align(h)) return -1;
    Align *a = h->al;
    (void)hipSetDevice(h->device);
    if (which == RV_SAI) {     /* the shared inverse: rank inside the owning sub-index (reveal.c:597,609,630) */
        if (cap < h->nT) { rv_set_error("buffer too small"); return -1; }
        if (a->level > 0) {
            std::vector<int64_t> st(a->lv.off);
            st.push_back(a->lv.m);
            DBuf &buf = h->ws.misc[9];
            if (buf.reserve(st.size() * 8)) return -1;
            if (hipMemcpy(buf.p, st.data(), st.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { rv_set_error("H2D failed"); return -1; }
            if (rv_sai_level_launch(h->ws, cur_sa(h), a->lv.m, buf.as<int64_t>(), a->lv.size(), h->dSAi.as<sa_t>())) return -1;
        }
        (void)hipStreamSynchronize(h->ws.stream);
        if (hipMemcpy(out, h->dSAi.p, (size_t)h->nT * sizeof(sa_t), hipMemcpyDeviceToHost) != hipSuccess) { rv_set_error("D2H failed"); return -1; }
        return h->nT;
    }
    if (need_sub(h, s)) return -1;
    const int64_t off = a->lv.off[(size_t)s], n = a->lv.n[(size_t)s];
    if (cap < n) { rv_set_error("buffer too small"); return -1; }
    (void)hipStreamSynchronize(h->ws.stream);
    hipError_t e;
    if (which == RV_SA) e = hipMemcpy(out, cur_sa(h) + off, (size_t)n * sizeof(sa_t), hipMemcpyDeviceToHost);
    else if (which == RV_LCP) e = hipMemcpy(out, cur_lcp(h) + off, (size_t)n * sizeof(lcp_t), hipMemcpyDeviceToHost);
    else { rv_set_error("rv_sub_array: bad array id"); return -1; }
    if (e != hipSuccess) { rv_set_error("D2H failed"); return -1; }
    return n;
}

int rv_sub_split(rv_index *h, int s, uint32_t l, int nsp, const int64_t *sp,
                 const int64_t *lead, int nlead, const int64_t *trail, int ntrail,
                 const int64_t *match, int nmatch, const int64_t *rest, int nrest) {
    RV_TRY(need_sub(h, s));
    for (int k = 0; k < nsp; k++)
        if (sp[k] < 0 || sp[k] + (int64_t)l > h->nT) { rv_set_error("match outside the text"); return -1; }
    const int64_t *lists[4] = {lead, trail, match, rest};
    const int cnts[4] = {nlead, ntrail, nmatch, nrest};
    for (int q = 0; q < 4; q++)
        for (int k = 0; k < cnts[q]; k++)
            if (lists[q][2 * k] < 0 || lists[q][2 * k + 1] > h->nT || lists[q][2 * k] > lists[q][2 * k + 1]) { rv_set_error("interval outside the text"); return -1; }
    static_assert(sizeof(RvIntv) == 2 * sizeof(int64_t), "RvIntv layout");
    RV_TRY(add_decision(h, s, l, sp, nsp, (const RvIntv *)lead, nlead, (const RvIntv *)trail, ntrail, (const RvIntv *)match, nmatch, (const RvIntv *)rest, nrest));
    h->al->dec.host_lists = true;
    return 0;
}

// Tables the scan and the device-side decisions of a level need; the commit in front of the level ships them with its own
// upload (a frontier import ships them itself): sub-index starts, and
//  - more than two samples: tile -> sub-index table (the multi-sample picker looks sub-indices up per candidate)
//  - two samples, untraced: the level's split can be decided on the device if each of its sub-indices owns at most one
//    interval per sample (rv_decide.hip): node intervals, "finished by the leaf kernel" flags, tiles.
static void prep_level_tables(rv_index *h, const Level &nx, int64_t m_next, bool tsub_on_device = false) {
    Align *a = h->al;
    const int nsn = nx.size();
    a->next_ss.assign(nx.off.begin(), nx.off.end()); a->next_ss.push_back(m_next);
    a->next_dev_ok = a->full_only && nsn > 0 && nsn <= RV_DECIDE_MAX_SUBS && m_next > 0;
    if (a->next_dev_ok && a->multi) {
        // more than two samples: one (begin, end) slot per sample and sub-index; a sub-index with two intervals of one sample
        // (multi-contig inputs) sends the level to the host path
        const int W = h->nsamples;
        a->next_dev_ok = W <= 64 && !h->ws.opt.keep_dead && !a->trace_on;
        if (a->next_dev_ok) {
            a->next_nodes.assign((size_t)nsn * 2 * W, 0); a->next_flags.assign((size_t)nsn, 0);
            for (int s2 = 0; s2 < nsn && a->next_dev_ok; s2++) {
                sa_t *nd = a->next_nodes.data() + (size_t)s2 * 2 * W;
                size_t sp = 0;
                for (int64_t k = nx.node_first[(size_t)s2]; k < nx.node_first[(size_t)s2 + 1]; k++) {
                    const RvIntv iv = nx.nodes[(size_t)k];
                    while (sp < h->nsep.size() && h->nsep[sp] < iv.begin) sp++;         // (sorted by begin: the sample index only grows)
                    if (iv.end <= iv.begin) continue;
                    if (nd[2 * sp] < nd[2 * sp + 1]) { a->next_dev_ok = false; break; }      // a second interval of this sample
                    nd[2 * sp] = (sa_t)iv.begin; nd[2 * sp + 1] = (sa_t)iv.end;
                }
            }
        }
    } else if (a->next_dev_ok) {
        a->next_nodes.assign((size_t)nsn * 4, 0); a->next_flags.assign((size_t)nsn, 0);
        const int64_t sep = h->nsep[0];
        for (int s2 = 0; s2 < nsn && a->next_dev_ok; s2++) {
            const int64_t nf = nx.node_first[(size_t)s2], nn = nx.node_first[(size_t)s2 + 1] - nf;
            if (nn < 1 || nn > 2) { a->next_dev_ok = false; break; }
            sa_t *nd = a->next_nodes.data() + (size_t)s2 * 4;
            for (int64_t k = 0; k < nn; k++) {
                const RvIntv iv = nx.nodes[(size_t)(nf + k)];
                if (iv.begin < sep) { if (nd[0] < nd[1]) a->next_dev_ok = false; nd[0] = (sa_t)iv.begin; nd[1] = (sa_t)iv.end; }
                else if (iv.begin > sep) { if (nd[2] < nd[3]) a->next_dev_ok = false; nd[2] = (sa_t)iv.begin; nd[3] = (sa_t)iv.end; }
                else a->next_dev_ok = false;
            }
            // the leaf kernel takes every sub-index of at most RV_LEAF_N ranks here (same rule as builtin_levels)
            a->next_flags[(size_t)s2] = (a->use_leaf && nx.n[(size_t)s2] <= RV_LEAF_N) ? 1 : 0;
        }
    }
    a->next_tsub.clear();
    if ((a->multi || a->next_dev_ok) && !tsub_on_device) {
        const int64_t ntn = ceil_div(m_next, RV_SPLIT_TILE);
        a->next_tsub.resize((size_t)ntn);
        int s2 = 0;
        for (int64_t t = 0; t < ntn; t++) {
            const int64_t r = t * RV_SPLIT_TILE;
            while (s2 + 1 < nsn && a->next_ss[(size_t)s2 + 1] <= r) s2++;
            a->next_tsub[(size_t)t] = s2;
        }
    }
}

/* reveal.c:1005-1252 for every decided sub-index; children -> next frontier */
int rv_frontier_commit(rv_index *h, int32_t *children) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    RV_HIP(hipSetDevice(h->device));
    hipStream_t q = h->ws.stream;
    const double t0 = now_s();
    const Level &lv = a->lv;
    Level &nx = a->nx;
    const Decisions &dc = a->dec;
    const int ns = lv.size();
    if (children) for (int k = 0; k < 3 * ns; k++) children[k] = -1;
    a->scanned = false;
    nx.clear();
    if (dc.size() == 0) { a->lv.clear(); a->dec.reset(0); return 0; }

    // ---- per-sub tables, next level layout, child bookkeeping (reveal.c:1136-1207) ----------------
    a->cb.clear(); a->ce.clear(); a->cc.clear(); a->mb.clear(); a->me.clear(); a->mpre.assign(1, 0);
    a->ctab_first.assign((size_t)ns + 1, 0); a->mtab_first.assign((size_t)ns + 1, 0);
    a->cut_first.assign((size_t)ns + 1, 0); a->mend_first.assign((size_t)ns + 1, 0);
    a->cut_lo.clear(); a->cut_hi.clear(); a->mend_pos.clear(); a->split_subs.clear();
    a->child_base.assign((size_t)ns * 3, 0); a->child_n.assign((size_t)ns * 3, 0);
    a->sub_start.resize((size_t)ns + 1);
    for (auto &r : a->rounds) r.clear();
    a->kids_small.clear(); a->kids_big.clear(); a->kids_lds.clear();
    const int64_t lds_n = h->ws.opt.bubble_no_lds ? 0 : RV_BUBBLE_LDS_N;
    int64_t running = 0;
    const int64_t lcap = (int64_t)h->maxlcp;
    // leading children above this many ranks take the data-parallel bubble rounds (RV_BUBBLE_PAR_MIN: test hook)
    const int64_t par_min = (h->ws.opt.bubble_par_min >= 0 ? h->ws.opt.bubble_par_min : bubble_par_default(a->multi));
    struct Ent { int64_t b, e; uint8_t c; };
    std::vector<Ent> ent;
    a->kid_tmp.clear();
    bool any_par = false;
    int64_t window_sum = 0;
    const bool split_tabs = !a->early_done;       // the split is still to be launched, from the tables built here
    const bool lower_tabs = !a->early_bubble;     // ... and so is the lower-casing of the matched ranges
    for (int s = 0; s < ns; s++) {
        a->sub_start[(size_t)s] = lv.off[(size_t)s];
        a->ctab_first[(size_t)s] = (int)a->cb.size(); a->mtab_first[(size_t)s] = (int)a->mb.size();
        a->cut_first[(size_t)s] = (int)a->cut_lo.size(); a->mend_first[(size_t)s] = (int)a->mend_pos.size();
        const int d = dc.of_sub[(size_t)s];
        if (d < 0) continue;
        a->split_subs.push_back(s);
        const RvIntv *lists[3] = {dc.lead.data() + dc.lead_first[(size_t)d], dc.trail.data() + dc.trail_first[(size_t)d], dc.rest.data() + dc.rest_first[(size_t)d]};
        const size_t cnts[3] = {(size_t)(dc.lead_first[(size_t)d + 1] - dc.lead_first[(size_t)d]), (size_t)(dc.trail_first[(size_t)d + 1] - dc.trail_first[(size_t)d]),
                                (size_t)(dc.rest_first[(size_t)d + 1] - dc.rest_first[(size_t)d])};
        // Untraced runs with the built-in picker and more than two samples: a child in which some sample owns no interval of
        // minl bases can never hold a match present in all its samples (a match lies inside one interval per sample), so
        // the picker would return () for it and it would leave after one more level of being split off, bubble-sorted,
        // scanned and book-kept for nothing -- about every second sub-index of a run is such a leaf.  Its ranks get the
        // label of matched suffixes (dropped by split, minima updated as for any labelled rank: reveal.c:640-662 treats
        // them like members of another child), the child is never made.  Anchors and text are unaffected; the number of
        // sub-indices visited is smaller than the reference's.
        bool dead[3] = {false, false, false};
        if (a->multi && a->full_only && !dc.host_lists && !h->ws.opt.keep_dead)
            for (int c = 0; c < 3; c++) dead[c] = cnts[c] > 0 && child_is_dead(h, lists[c], cnts[c], a->minl, a->minn);
        if (dc.drop_lead) dead[0] = cnts[0] > 0;
        // class table of this sub: lead/trail/rest merged by begin (the split's tables: not needed when the split of this level ran
        // already, from the same tables built on the device -- early_split; those decisions come from the built-in picker, whose
        // intervals need no checking)
        if (split_tabs) {
            ent.clear();
            static const uint8_t cls_of[3] = {1, 2, 4};
            for (int c = 0; c < 3; c++) for (size_t k = 0; k < cnts[c]; k++) if (lists[c][k].end > lists[c][k].begin) ent.push_back({lists[c][k].begin, lists[c][k].end, dead[c] ? (uint8_t)3 : cls_of[c]});
            if (!std::is_sorted(ent.begin(), ent.end(), [](const Ent &x, const Ent &y) { return x.b < y.b; }))
                std::sort(ent.begin(), ent.end(), [](const Ent &x, const Ent &y) { return x.b < y.b; });
            for (size_t k = 0; k < ent.size(); k++) {
                if (k && ent[k].b < ent[k - 1].e) { rv_set_error("graphalign returned overlapping intervals [%lld,%lld) / [%lld,%lld)", (long long)ent[k - 1].b, (long long)ent[k - 1].e, (long long)ent[k].b, (long long)ent[k].e); return -1; }
                a->cb.push_back((sa_t)ent[k].b); a->ce.push_back((sa_t)ent[k].e); a->cc.push_back(ent[k].c);
            }
        }
        for (int64_t k = dc.sp_first[(size_t)d]; k < dc.sp_first[(size_t)d + 1]; k++) {       // sorted
            const int64_t p = dc.sp[(size_t)k];
            const u32 l = dc.spl[(size_t)k];
            if (l && lower_tabs) { a->mb.push_back((sa_t)p); a->me.push_back((sa_t)(p + l)); a->mpre.push_back(a->mpre.back() + l); }
            if (split_tabs) a->mend_pos.push_back((sa_t)(p + (int64_t)l));
        }
        // children
        int64_t lead_off = 0, lead_n = 0;
        for (int c = 0; c < 3; c++) {
            int64_t cn = 0;
            if (!dead[c]) for (size_t k = 0; k < cnts[c]; k++) cn += lists[c][k].end - lists[c][k].begin;
            a->child_base[(size_t)s * 3 + c] = (u32)running;
            a->child_n[(size_t)s * 3 + c] = (u32)cn;
            if (c == 0) { lead_off = running; lead_n = cn; }
            if (cn > 0) {
                if (children) children[3 * s + c] = nx.size();
                nx.off.push_back(running); nx.n.push_back(cn); nx.depth.push_back(lv.depth[(size_t)s] + 1);
                nx.nsamples.push_back(count_samples(h, lists[c], cnts[c]));
                nx.parent.push_back(s); nx.kind.push_back(c + 1);
                nx.nodes.insert(nx.nodes.end(), lists[c], lists[c] + cnts[c]);
                nx.node_first.push_back((int64_t)nx.nodes.size());
                running += cn;
            }
        }
        // windows in front of this sub's cuts, in the order graphalign listed the matched intervals
        if (lead_n > 0) {
            const int64_t m0 = dc.match_first[(size_t)d], m1 = dc.match_first[(size_t)d + 1];
            const int c0 = (int)a->cut_lo.size();
            for (int64_t r = m0; r < m1; r++) {
                const int64_t B = dc.match[(size_t)r].begin;
                int64_t lo = B;
                {   // the leading interval that ends at this cut (the list is sorted and non-overlapping: binary search on the ends)
                    size_t x = 0, y = cnts[0];
                    while (x < y) { const size_t mid = (x + y) / 2; if (lists[0][mid].end < B) x = mid + 1; else y = mid; }
                    if (x < cnts[0] && lists[0][x].end == B && lists[0][x].begin < B) lo = std::max(lists[0][x].begin, B - lcap);
                }
                a->cut_lo.push_back((sa_t)lo); a->cut_hi.push_back((sa_t)B);
            }
            const int c1 = (int)a->cut_lo.size();
            bool any = false;
            int64_t wsum = 0;
            for (int q = c0; q < c1; q++) { any = any || a->cut_lo[(size_t)q] < a->cut_hi[(size_t)q]; wsum += (int64_t)a->cut_hi[(size_t)q] - (int64_t)a->cut_lo[(size_t)q]; }
            if (any) {
                a->kid_tmp.push_back({lead_off, lead_n, c0, c1, m0, wsum});
                if (lead_n > par_min) any_par = true;
                window_sum += wsum;
            }
        }
    }
    // Leading children above par_min ranks take the data-parallel rounds.  Once a level runs those rounds anyway, the small
    // children join them as long as there are few of them (measured: with thousands of small children per level the
    // one-workgroup-per-child kernel is the cheaper way, C3/C4): their kernel would only add its own latency in front.
    const bool all_par = any_par && window_sum <= ((int64_t)1 << 20) && !h->ws.opt.bubble_no_join && !a->early_bubble;
    // The LDS kernels pay off by throughput (thousands of small children per level: many samples, or very large inputs).  A
    // few hundred small children ride along with the larger ones for free (measured on C2: 612 vs 597 Mbp/s).
    size_t lds_candidates = 0;
    for (const auto &kd : a->kid_tmp) lds_candidates += kd.n <= lds_n;
    // (more than two samples: many cuts per child, no leaf kernel -- the LDS kernel is always the better one for small children)
    const bool use_lds = lds_candidates > 1024 || (lds_candidates > 0 && (a->multi || h->ws.opt.bubble_lds_always));
    for (const auto &kd : a->kid_tmp) {
        if (a->early_bubble && kd.n <= par_min) continue;      // bubbled already, right behind the early split
        if ((!all_par && kd.n <= par_min) || (use_lds && kd.n <= lds_n)) {             // every cut of this child in one workgroup, sequentially
            RvBubbleDesc bd; bd.off = kd.off; bd.n = kd.n; bd.B = 0; bd.wlo = 0; bd.cut0 = kd.c0; bd.cut1 = kd.c1;
            ((use_lds && kd.n <= lds_n) ? a->kids_lds : kd.n <= RV_BUBBLE_BIG_N ? a->kids_small : a->kids_big).push_back(bd);
            continue;
        }
        for (int q = kd.c0; q < kd.c1; q++) {          // one cut per round, data-parallel (rv_bubble.hip)
            const int64_t B = (int64_t)a->cut_hi[(size_t)q], lo = (int64_t)a->cut_lo[(size_t)q];
            if (lo >= B) continue;
            const size_t r = (size_t)(q - kd.c0);
            if (a->rounds.size() <= r) a->rounds.resize(r + 1);
            RvBubbleDesc bd; bd.off = kd.off; bd.n = kd.n; bd.B = B; bd.wlo = lo; bd.cut0 = kd.c0; bd.cut1 = kd.c1;
            a->rounds[r].push_back(bd);
        }
    }
    a->sub_start[(size_t)ns] = lv.m;
    a->ctab_first[(size_t)ns] = (int)a->cb.size(); a->mtab_first[(size_t)ns] = (int)a->mb.size();
    a->cut_first[(size_t)ns] = (int)a->cut_lo.size(); a->mend_first[(size_t)ns] = (int)a->mend_pos.size();
    nx.m = running;
    const int64_t m_next = running;
    if (m_next >= ((int64_t)1 << 32)) { rv_set_error("level larger than 2^32 ranks not supported yet"); return -1; }
    a->descs.clear();
    std::vector<int> round_first;
    std::vector<uint8_t> round_seq;      // some window of the round exceeds what the parallel path takes: also launch the sequential kernels
    for (auto &r : a->rounds) {
        round_first.push_back((int)a->descs.size());
        bool seq = false;
        for (auto &x : r) seq = seq || (x.B - x.wlo) > RV_PB_CAP;
        round_seq.push_back(seq ? 1 : 0);
        a->descs.insert(a->descs.end(), r.begin(), r.end());
    }
    round_first.push_back((int)a->descs.size());
    a->woff.assign(a->descs.size() + 1, 0);
    a->toff.assign(a->descs.size() + 1, 0);
    for (size_t k = 0; k < a->descs.size(); k++) {
        a->woff[k + 1] = a->woff[k] + (a->descs[k].B - a->descs[k].wlo);
        a->toff[k + 1] = a->toff[k] + ceil_div(a->descs[k].n, RV_SPLIT_TILE);
    }

    a->lg[0] = now_s() - t0;      // tables built
    // ---- one upload for all the tables ---------------------------------------------------
    const int64_t ntiles = ceil_div(lv.m, RV_SPLIT_TILE);
    a->tile_sub.resize(split_tabs ? (size_t)ntiles : 0);
    if (split_tabs) {
        int s2 = 0;
        for (int64_t t = 0; t < ntiles; t++) {
            const int64_t r = t * RV_SPLIT_TILE;
            while (s2 + 1 < ns && a->sub_start[(size_t)s2 + 1] <= r) s2++;
            a->tile_sub[(size_t)t] = s2;
        }
    }
    a->lgx[0] = now_s() - t0;     // tile -> sub-index table
    Packer &pk = a->pk;
    pk.clear();
    const size_t o_tsub = pk.addv(a->tile_sub);
    const size_t o_cb = pk.addv(a->cb), o_ce = pk.addv(a->ce), o_cc = pk.addv(a->cc), o_mb = pk.addv(a->mb), o_me = pk.addv(a->me), o_mpre = pk.addv(a->mpre);
    // (tables only the split reads stay at home when it has run already: ~100 bytes per sub-index, 50000 sub-indices per level at 2 x 250 Mbp)
    static const std::vector<int> no_int; static const std::vector<u32> no_u32; static const std::vector<int64_t> no_i64;
    const size_t o_ctf = pk.addv(split_tabs ? a->ctab_first : no_int), o_mtf = pk.addv(split_tabs ? a->mtab_first : no_int);
    const size_t o_ss = pk.addv(split_tabs ? a->sub_start : no_i64), o_cbase = pk.addv(split_tabs ? a->child_base : no_u32), o_cn = pk.addv(split_tabs ? a->child_n : no_u32);
    const size_t o_cf = pk.addv(split_tabs ? a->cut_first : no_int);
    const size_t o_clo = pk.addv(a->cut_lo), o_chi = pk.addv(a->cut_hi);
    const size_t o_desc = pk.addv(a->descs), o_woff = pk.addv(a->woff), o_toff = pk.addv(a->toff), o_mf = pk.addv(split_tabs ? a->mend_first : no_int), o_mp = pk.addv(a->mend_pos);
    // one launch instead of two when a level has few small children next to large ones (the kernels would run one after the
    // other on the stream; in a 1024-thread workgroup a small child simply finishes early)
    if (!a->kids_big.empty() && a->kids_small.size() <= 512 && !h->ws.opt.bubble_no_merge) { a->kids_big.insert(a->kids_big.end(), a->kids_small.begin(), a->kids_small.end()); a->kids_small.clear(); }
    int lds_count[3] = {0, 0, 0};
    {   // LDS children by size class (stable: the order inside a class does not matter)
        auto cls = [](const RvBubbleDesc &x) { return x.n <= RV_BUBBLE_LDS_N0 ? 0 : x.n <= RV_BUBBLE_LDS_N1 ? 1 : 2; };
        std::stable_sort(a->kids_lds.begin(), a->kids_lds.end(), [&](const RvBubbleDesc &x, const RvBubbleDesc &y) { return cls(x) < cls(y); });
        for (auto &x : a->kids_lds) lds_count[cls(x)]++;
        // few children: one launch with the largest configuration (occupancy does not matter, a launch does)
        if (a->kids_lds.size() <= 512) { lds_count[2] = (int)a->kids_lds.size(); lds_count[0] = lds_count[1] = 0; }
    }
    const size_t o_ks = pk.addv(a->kids_small), o_kb = pk.addv(a->kids_big), o_kl = pk.addv(a->kids_lds);
    a->lgx[1] = now_s() - t0;     // packed
    prep_level_tables(h, nx, m_next, true);
    a->lgx[2] = now_s() - t0;     // tables of the next level
    size_t o_ntsub = 0, o_nnodes = 0, o_nflags = 0, o_ntsub2 = 0;
    if (a->multi) o_ntsub = pk.addv(a->next_tsub);
    if (a->next_dev_ok) { o_nnodes = pk.addv(a->next_nodes); o_nflags = pk.addv(a->next_flags); o_ntsub2 = a->multi ? o_ntsub : pk.addv(a->next_tsub); }
    const size_t o_nss = pk.addv(a->next_ss), o_nwant = pk.addv(nx.nsamples);
    a->sub_off_h.assign(split_tabs ? (size_t)ns * 3 : 0, 0);
    {   // SURVEY 8(a) A12 / reveal.c:666-727: bubble_sort looks at every rank of a leading child (4 B SA + 4 B LCP; the BWT byte travels with them here).
        // The spans of the class are opened where the kernels are queued (the early ones before the host knows the children's sizes): its bytes come from here.
        double lead_ranks = 0;
        for (int s = 0; s < ns; s++) lead_ranks += (double)a->child_n[(size_t)s * 3];
        h->prof.credit(RV_K_BUBBLE, lead_ranks * (double)(sizeof(sa_t) + sizeof(lcp_t) + 1));
    }
    u32 class_total[4] = {0, 0, 0, 0};
    if (split_tabs)
        for (int s = 0; s < ns; s++) for (int c = 0; c < 3; c++) { a->sub_off_h[(size_t)s * 3 + c] = a->child_base[(size_t)s * 3 + c] - class_total[c]; class_total[c] += a->child_n[(size_t)s * 3 + c]; }
    const size_t o_suboff = pk.addv(a->sub_off_h), o_expect = pk.add(class_total, sizeof class_total), o_total = pk.reserve(16), o_bcnt = pk.reserve(a->descs.size() * 4 + 4), o_mcnt = pk.reserve(a->descs.size() * 4 + 4), o_gcnt = pk.reserve(16);
    const size_t o_bstate = pk.reserve((a->descs.size() + 1) * sizeof(RvBubbleState));
    RV_TRY(a->dTab.reserve(pk.size() + 64));
    a->lgx[3] = now_s() - t0;     // offsets packed
    if (pk.pageable || h->ws.opt.tables_memcpy) RV_HIP(hipMemcpyAsync(a->dTab.p, pk.data(), pk.size(), hipMemcpyHostToDevice, q));
    else { pk.grow(pk.size() + 16); RV_TRY(rv_h2d_copy(h->ws, pk.data(), a->dTab.p, pk.size())); }
    a->lg[1] = now_s() - t0;      // upload issued
    uint8_t *tb = a->dTab.as<uint8_t>();
    a->d_next_ss = (const int64_t *)(tb + o_nss); a->d_next_want = (const int *)(tb + o_nwant); a->d_next_tsub = (const int *)(tb + o_ntsub);
    a->d_next_nodes = (const sa_t *)(tb + o_nnodes); a->d_next_flags = tb + o_nflags; a->d_next_tsub2 = (const int *)(tb + o_ntsub2);
    if (a->multi || a->next_dev_ok) {      // the next level's tile -> sub-index table, from its sub-index starts (queued behind the upload)
        const int64_t ntn = ceil_div(m_next, RV_SPLIT_TILE);
        RV_TRY(a->dNextTsub.reserve((size_t)ntn * 4 + 64));
        RV_TRY(rv_tile_sub_launch(h->ws, a->d_next_ss, nx.size(), a->dNextTsub.as<int>(), ntn));
        a->d_next_tsub = a->d_next_tsub2 = a->dNextTsub.as<int>();
    }
    RV_TRY(a->dD.reserve((size_t)lv.m + 64));
    RV_TRY(a->dTile.reserve((size_t)ntiles * 3 * 5 * 4 + 64));
    RV_TRY(a->dList.reserve((size_t)a->woff.back() * 4 + 64));
    const int nxt = (a->level == 0) ? 0 : (a->cur + 1) % RV_LEVEL_BUFS;
    RV_TRY(a->lvSA[nxt].reserve((size_t)(m_next + 64) * sizeof(sa_t)));
    RV_TRY(a->lvLCP[nxt].reserve((size_t)(m_next + 64) * sizeof(lcp_t)));
    RV_TRY(a->lvBWT[nxt].reserve((size_t)m_next + 64));

    // A leaf launch of an earlier level may still be reading the buffer this commit writes (ping-pong), and the long-move
    // path of this commit scribbled over the current (parent) buffer when that was its scratch: order after them.
    {
        const int cur_id = (a->level == 0) ? RV_LEVEL_BUFS : a->cur;            // RV_LEVEL_BUFS = the main arrays
        (void)cur_id;
        const int wait_ids[2] = {nxt, h->ws.opt.bubble_parent_scratch && !a->descs.empty() ? cur_id : -1};
        for (int k = 0; k < 2; k++) {
            const int id = wait_ids[k];
            if (id < 0) continue;
            const int slot = id;
            if (a->leaf_pending[slot]) { RV_HIP(hipStreamWaitEvent(q, a->ev_leaf[slot], 0)); a->leaf_pending[slot] = false; }
        }
    }
    RvLabelTabs lt;
    lt.sub_start = (const int64_t *)(tb + o_ss); lt.nsubs = ns; lt.tile_sub = (const int *)(tb + o_tsub);
    lt.ctab_first = (const int *)(tb + o_ctf); lt.cbegin = (const sa_t *)(tb + o_cb); lt.cend = (const sa_t *)(tb + o_ce); lt.ccls = tb + o_cc;
    lt.mtab_first = (const int *)(tb + o_mtf); lt.mbegin = (const sa_t *)(tb + o_mb); lt.mend = (const sa_t *)(tb + o_me); lt.nmatch = (int)a->mb.size();
    RvSplitArgs sa;
    u32 *tiles = a->dTile.as<u32>();
    sa.ntiles = ntiles;
    sa.tile_cnt = tiles; sa.tile_has = tiles + 3 * ntiles; sa.tile_post = tiles + 6 * ntiles;
    sa.tile_G = tiles + 9 * ntiles; sa.tile_carry = tiles + 12 * ntiles;
    sa.total = (u32 *)(tb + o_total);
    sa.sub_start = lt.sub_start; sa.nsubs = ns; sa.tile_sub = lt.tile_sub;
    sa.child_base = (const u32 *)(tb + o_cbase); sa.child_n = (const u32 *)(tb + o_cn); sa.sub_off = (const u32 *)(tb + o_suboff); sa.expect_total = (const u32 *)(tb + o_expect);
    sa.cut_first = (const int *)(tb + o_cf); sa.cut_lo = (const sa_t *)(tb + o_clo); sa.cut_hi = (const sa_t *)(tb + o_chi);
    sa.mend_first = (const int *)(tb + o_mf); sa.mend_pos = (const sa_t *)(tb + o_mp);
    sa.mend_all = dc.host_lists ? 1 : 0;
    sa.SA_out = a->lvSA[nxt].as<sa_t>(); sa.LCP_out = a->lvLCP[nxt].as<lcp_t>(); sa.BWT_out = a->lvBWT[nxt].as<uint8_t>(); sa.SAi = h->dSAi.as<sa_t>();
    sa.err = a->dErr.as<u32>();      // persistent for the alignment: an early split (rv_decide.hip) runs before this upload exists
    // (a level whose split went out behind the picker has its span there: none here, or its bytes would count twice)
    int id = a->early_done ? -1 : h->prof.begin(q, RV_K_SPLIT, (double)lv.m * (2 * (sizeof(sa_t) + sizeof(lcp_t) + 2)) + (double)m_next * (sizeof(sa_t) + sizeof(lcp_t) + 1));
    if (!a->early_done) {     // (otherwise queued behind the picker already, with the same tables built on the device)
        if (!a->descs.empty()) {      // tile bounds: the level has data-parallel bubble rounds
            RV_TRY(a->dTmin.reserve((size_t)(lv.m / RV_SPLIT_TILE + 2) * 4));
            RV_HIP(hipMemsetAsync(a->dTmin.p, 0xFF, (size_t)(lv.m / RV_SPLIT_TILE + 2) * 4, q));
            sa.tmin_out = a->dTmin.as<u32>();
        }
        RV_TRY(rv_split_launch(h->ws, cur_sa(h), cur_lcp(h), a->dD.as<uint8_t>(), cur_bwt(h), lv.m, lt, sa, (int)a->split_subs.size()));
    }
    h->prof.end(q, id);
    if (!a->early_bubble) {
        // many short ranges (the deep levels: 10^5 matches of ~100 bases): a wave per range; few long ones: a thread per base, which looks
        // its range up by a binary search over the prefix sums (17 dependent loads per base at 10^5 ranges: 0.2 ms per level at C4)
        if (lt.nmatch > 256 && a->mpre.back() / lt.nmatch <= 512) RV_TRY(rv_lower_ranges_launch(h->ws, h->dT.as<uint8_t>(), lt.mbegin, lt.mend, lt.nmatch));
        else RV_TRY(rv_lower_launch(h->ws, h->dT.as<uint8_t>(), lt.mbegin, lt.mend, (const int64_t *)(tb + o_mpre), lt.nmatch, a->mpre.back()));
    }
    const double t1 = now_s();
    a->lg[2] = t1 - t0;           // label/split/lower enqueued

    // ---- bubble_sort rounds (reveal.c:1250-1252, :666-727) -----------------------------------
    if (!a->descs.empty() || !a->kids_small.empty() || !a->kids_big.empty() || !a->kids_lds.empty()) {
        RvBubbleArgs ba;
        ba.desc = (const RvBubbleDesc *)(tb + o_desc); ba.woff = (const int64_t *)(tb + o_woff);
        ba.cnt = (u32 *)(tb + o_bcnt); ba.list = a->dList.as<u32>();
        {   // every bubble kernel leaves the flag bytes it set at zero again: one memset per alignment (and per reallocation) instead of per level
            const void *before = a->dFlag.p;
            RV_TRY(a->dFlag.reserve((size_t)m_next + 64));
            if (!a->flag_clean || a->dFlag.p != before) { RV_HIP(hipMemsetAsync(a->dFlag.p, 0, a->dFlag.cap, q)); a->flag_clean = true; }
        }
        ba.flag = a->dFlag.as<uint8_t>();
        ba.SA = sa.SA_out; ba.LCP = sa.LCP_out; ba.BWT = sa.BWT_out; ba.SAi = sa.SAi; ba.cut_lo = sa.cut_lo; ba.cut_hi = sa.cut_hi; ba.err = sa.err;
        ba.state = (RvBubbleState *)(tb + o_bstate);
        ba.dbg = nullptr;
        if (h->ws.opt.level_log) { if (!a->dDbg.p) { RV_TRY(a->dDbg.reserve(64)); RV_HIP(hipMemsetAsync(a->dDbg.p, 0, 64, q)); } ba.dbg = a->dDbg.as<unsigned long long>(); }
        // the parent level is dead once split has run (at level 0 these are the main SA/LCP/BWT, which the
        // reference frees at this point, reveal.c:1279-1284): scratch for the grid-wide long moves
        ba.scrSA = const_cast<sa_t *>(cur_sa(h)); ba.scrLCP = const_cast<lcp_t *>(cur_lcp(h)); ba.scrBWT = const_cast<uint8_t *>(cur_bwt(h));
        if (!a->descs.empty() && !h->ws.opt.bubble_parent_scratch) {
            // ... but this level's leaf launch (second stream, ~180 us) still reads them, and waiting for it left the main
            // stream idle for ~40 us at every level that has both leaf sub-indices and data-parallel rounds: own scratch
            // (9 B per rank of the next level; RV_BUBBLE_PARENT_SCRATCH=1 = the parent arrays and the wait, as before)
            RV_TRY(a->scrSA.reserve((size_t)(m_next + 64) * sizeof(sa_t)));
            RV_TRY(a->scrLCP.reserve((size_t)(m_next + 64) * sizeof(lcp_t)));
            RV_TRY(a->scrBWT.reserve((size_t)m_next + 64));
            ba.scrSA = a->scrSA.as<sa_t>(); ba.scrLCP = a->scrLCP.as<lcp_t>(); ba.scrBWT = a->scrBWT.as<uint8_t>();
        }
        if (!a->descs.empty()) {
            const size_t W = (size_t)a->woff.back() + 16, TT = (size_t)a->toff.back() + 16;
            RV_TRY(a->dPar.reserve(TT * 4 + W * (8 + 7 * 4 + sizeof(sa_t) + 2) + 256));
            uint8_t *pb = a->dPar.as<uint8_t>();
            ba.par.toff = (const int64_t *)(tb + o_toff);
            ba.par.mcnt = (u32 *)(tb + o_mcnt);
            ba.par.gcount = (u32 *)(tb + o_gcnt);
            ba.par.glist = (u64 *)pb; pb += W * 8;
            ba.par.Qs = (sa_t *)pb; pb += W * sizeof(sa_t);
            ba.par.tmin = a->dTmin.as<u32>(); pb += TT * 4;
            ba.par.mrank = (u32 *)pb; pb += W * 4;
            ba.par.msite = (u32 *)pb; pb += W * 4;
            ba.par.R = (u32 *)pb; pb += W * 4;
            ba.par.Qsite = (u32 *)pb; pb += W * 4;
            ba.par.QF = (u32 *)pb; pb += W * 4;
            ba.par.Qt = (u32 *)pb; pb += W * 4;
            ba.par.Qlcp = (u32 *)pb; pb += W * 4;
            ba.par.Qbw = pb; pb += W;
            ba.par.Qlast = pb;
            ba.par.tready = nullptr; ba.par.epoch = 0;
            if (!h->ws.opt.pb_two_pass) {      // (test hook: copy-out + scatter as two kernels through the scratch arrays)
                const size_t before = a->dPbReady.cap;
                RV_TRY(a->dPbReady.reserve(TT * 4 + 64));
                if (a->dPbReady.cap != before) RV_HIP(hipMemsetAsync(a->dPbReady.p, 0, a->dPbReady.cap, q));      // launch numbers start at 1
                ba.par.tready = a->dPbReady.as<u32>();
            }
        }
        id = h->prof.begin(q, RV_K_BUBBLE, 0.0);
        // Three independent groups of work: children that fit into LDS, children replayed in one workgroup each, and the
        // data-parallel rounds of the largest ones.  A kernel boundary on one stream is a barrier, so each extra group goes
        // to its own stream (fork / join by events): the level's bubble time is the longest group, not their sum.
        bool forked = false, forked2 = false;
        if (!a->bub_stream) {
            RV_HIP(rv_stream_get(&a->bub_stream));
            RV_HIP(rv_stream_get(&a->bub_stream2));
            RV_HIP(hipEventCreateWithFlags(&a->ev_fork, hipEventDisableTiming));
            RV_HIP(hipEventCreateWithFlags(&a->ev_join, hipEventDisableTiming));
            RV_HIP(hipEventCreateWithFlags(&a->ev_join2, hipEventDisableTiming));
        }
        const bool have_kids = !a->kids_small.empty() || !a->kids_big.empty();
        const int groups = (int)!a->kids_lds.empty() + (int)have_kids + (int)!a->descs.empty();
        if (groups > 1) RV_HIP(hipEventRecord(a->ev_fork, q));
        if (!a->kids_lds.empty()) {
            const bool side = groups > 1;
            Workspace lw; lw.stream = side ? a->bub_stream : q;
            if (side) RV_HIP(hipStreamWaitEvent(a->bub_stream, a->ev_fork, 0));
            RV_TRY(rv_bubble_children_lds_launch(lw, ba, (const RvBubbleDesc *)(tb + o_kl), lds_count));
            if (side) { RV_HIP(hipEventRecord(a->ev_join, a->bub_stream)); forked = true; }
        }
        if (have_kids) {
            // (two samples, a few hundred children: the fork/join costs what it saves -- measured on C2)
            const bool side = !a->descs.empty() && (a->multi || a->kids_small.size() + a->kids_big.size() > 1024);      // the rounds stay on the main stream
            Workspace lw; lw.stream = side ? a->bub_stream2 : q;
            if (side) RV_HIP(hipStreamWaitEvent(a->bub_stream2, a->ev_fork, 0));
            RV_TRY(rv_bubble_children_launch(lw, ba, (const RvBubbleDesc *)(tb + o_ks), (int)a->kids_small.size(), (const RvBubbleDesc *)(tb + o_kb), (int)a->kids_big.size()));
            if (side) { RV_HIP(hipEventRecord(a->ev_join2, a->bub_stream2)); forked2 = true; }
        }
        bool seq_before = false;      // an earlier round of this level ran the sequential kernels: the tile bounds have to be refreshed
        for (size_t r = 0; r + 1 < round_first.size(); r++) {
            const int first = round_first[r], count = round_first[r + 1] - first;
            ba.par.epoch = ++a->pb_epoch;
            if (a->pb_epoch == 0xFFFFFFFFu) { a->pb_epoch = 0; RV_HIP(hipMemsetAsync(a->dPbReady.p, 0, a->dPbReady.cap, q)); }
            RV_TRY(rv_bubble_par_round_launch(h->ws, ba, first, count, a->woff[(size_t)(first + count)] - a->woff[(size_t)first],
                                              a->toff[(size_t)(first + count)] - a->toff[(size_t)first], seq_before));
            if (round_seq[r]) { RV_TRY(rv_bubble_seq_launch(h->ws, ba, first, count)); seq_before = true; }
        }
        if (forked2) RV_HIP(hipStreamWaitEvent(q, a->ev_join2, 0));
        if (forked) RV_HIP(hipStreamWaitEvent(q, a->ev_join, 0));
        h->prof.end(q, id);
    }
    if (!a->multi && m_next > 1) {
        a->d_err = a->dErr.as<u32>();      // pair mode: the next scan's single copy brings the error word along
    } else {
        u32 err = 0;
        RV_HIP(hipMemcpyAsync(&err, a->dErr.p, 4, hipMemcpyDeviceToHost, q));
        RV_HIP(hipStreamSynchronize(q));
        RV_TRY(report_dev_err(err));
    }

    if (a->level == 0) h->main_arrays_freed = true;      /* reveal.c:1279-1284 */
    a->level++;
    a->cur = nxt;
    a->cur_dev_ok = a->next_dev_ok; a->early_done = false; a->early_bubble = false;
    std::swap(a->lv, a->nx);
    a->dec.reset(a->lv.size());
    a->skip_scan.assign((size_t)a->lv.size(), 0);
    a->st.t_split += t1 - t0;
    a->st.t_bubble += now_s() - t1;
    return 0;
}

/* ---- the whole recursion with the built-in benchmark callbacks ------------------- */
// output buffers of the leaf kernel (two samples only): sub-indices of at most RV_LEAF_N ranks finish on the GPU in one launch per level
static int builtin_leaf_setup(rv_index *h) {
    Align *a = h->al;
    hipStream_t q = h->ws.stream;
    a->use_leaf = !a->multi && !h->ws.opt.no_leaf && a->picker == 0;
    a->leaf_flip = 0;
    if (!a->use_leaf) return 0;
    a->leaf_anchor_cap = (size_t)(h->nT / std::max(a->minl, 1)) + 1024;
    a->leaf_trace_cap = a->trace_on ? 2 * a->leaf_anchor_cap + 1024 : 0;
    const size_t bytes = 256 + a->leaf_anchor_cap * (4 + 8 + 8) + a->leaf_trace_cap * sizeof(rv_trace) + 64;
    RV_TRY(a->dLeaf.reserve(bytes));
    uint8_t *base = a->dLeaf.as<uint8_t>();
    a->lf_counters = (u32 *)base;                          // [0] anchors, [1] trace records, [2] error bits
    a->lf_stats = (unsigned long long *)(base + 64);
    a->lf_pos = (int64_t *)(base + 256); a->lf_l = (u32 *)(a->lf_pos + 2 * a->leaf_anchor_cap);
    a->lf_tr = (rv_trace *)(base + 256 + a->leaf_anchor_cap * 20 + ((8 - (a->leaf_anchor_cap * 20) % 8) % 8));
    RV_HIP(hipMemsetAsync(base, 0, 256, q));
    if (!a->leaf_stream) {
        // (tried: the largest LDS class and the 1024-thread replays of the bubble on streams of their own, four side streams instead of two,
        // and the lower-casing queued behind their fork -- 267.4 / 267.9 / 265.4 ms for both / one / neither at C4: no gain)
        // (tried twice: lowest stream priority for the leaf launches -- 306 against 302 ms at C4 while the launches were bound by their
        // atomics, 282 against 277 ms after that)
        RV_HIP(rv_stream_get(&a->leaf_stream));
        RV_HIP(rv_stream_get(&a->leaf_stream2));
        RV_HIP(hipEventCreateWithFlags(&a->ev_ready, hipEventDisableTiming));
        for (int k = 0; k <= RV_LEVEL_BUFS; k++) RV_HIP(hipEventCreateWithFlags(&a->ev_leaf[k], hipEventDisableTiming));
        for (int k = 0; k < 2; k++) RV_HIP(hipEventCreateWithFlags(&a->ev_roots[k], hipEventDisableTiming));
    }
    for (int k = 0; k <= RV_LEVEL_BUFS; k++) a->leaf_pending[k] = false;
    a->roots_inflight[0] = a->roots_inflight[1] = false;
    return 0;
}

static int builtin_setup(rv_index *h, int minl, int minn) {
    // (an untraced built-in run never hands a sub-index out: the inverse stays unmade unless somebody asked for it before)
    RV_TRY(align_begin(h, minl, minn, h->al && h->al->trace_on));
    Align *a = h->al;
    a->full_only = !a->trace_on && a->picker == 0;      // (the chain picker looks at every match of a sub-index: the scans hand their whole lists to the host)
    a->an_l.clear(); a->an_off.assign(1, 0); a->an_pos.clear(); a->trace.clear(); a->leaf_na = 0;
    a->seeds_cur.clear(); a->seeds_lead.clear(); a->seeds_trail.clear(); a->picker_calls = a->picker_seeded = 0; a->picker_ns = a->picker_list_ns = a->galign_ns = 0;
    if (a->picker == 1) {
        if ((int)h->nodes.size() != h->nsamples) { rv_set_error("the native picker (rv_set_picker) takes one sequence per sample: %d sequences in %d samples", (int)h->nodes.size(), h->nsamples); return -1; }
        if (h->rc) { rv_set_error("the native picker (rv_set_picker) with construct(rc=1) is not supported"); return -1; }
    }
    // The reference's picker looks at the matches present in every sample of the sub-index and at nothing else, unless there is none (schemes.py:227-232): the scan
    // drops the others on the device (the filter of rv_set_preselect without its cap, which --trim forbids: trim_overlap runs on the filtered list, in front of the
    // cap) -- of the 1.8 x 10^6 matches of five 5 Mbp genomes that every level's scan found again, copied to the host and sorted into its sub-indices.
    if (a->picker != 0 && a->multi && !a->trace_on) { a->presel_on = true; a->pick_filter = true; }
    else a->pick_filter = false;
    if (a->picker == 2) {
        if (!a->ggraph) { rv_set_error("rv_set_graph_picker: no graph"); return -1; }
        if (h->rc) { rv_set_error("the native picker (rv_set_graph_picker) with construct(rc=1) is not supported"); return -1; }
        a->g_left.assign((size_t)a->lv.size(), RvGraphIv{-1, -1}); a->g_right.assign((size_t)a->lv.size(), RvGraphIv{-1, -1});      // (the root: no left / right node)
    }
    const bool use_leaf = !a->multi && !h->ws.opt.no_leaf && a->picker == 0;
    a->use_leaf = use_leaf;
    // level 0 of an untraced two-sample run: ship the tables the device-side picker and decisions need (what a commit ships for
    // the levels after it), so the first level takes the same path as the others
    if (!a->multi && a->full_only && !a->lv.nodes.empty() && a->lv.nodes.size() <= 2 && h->n < ((int64_t)1 << 32)) {
        sa_t nd[4] = {0, 0, 0, 0};
        bool ok = true;
        for (const RvIntv &iv : a->lv.nodes) {
            if (iv.begin < h->nsep[0]) { if (nd[0] < nd[1]) ok = false; nd[0] = (sa_t)iv.begin; nd[1] = (sa_t)iv.end; }
            else if (iv.begin > h->nsep[0]) { if (nd[2] < nd[3]) ok = false; nd[2] = (sa_t)iv.begin; nd[3] = (sa_t)iv.end; }
            else ok = false;
        }
        if (ok) {
            Packer &pk = a->pk;
            pk.clear();
            const int64_t ss[2] = {0, h->n};
            const uint8_t fl[16] = {(uint8_t)((use_leaf && h->n <= RV_LEAF_N) ? 1 : 0)};
            a->next_tsub.assign((size_t)ceil_div(h->n, RV_SPLIT_TILE), 0);
            const size_t o1 = pk.add(ss, sizeof ss), o2 = pk.add(nd, sizeof nd), o3 = pk.add(fl, sizeof fl), o4 = pk.addv(a->next_tsub);
            RV_TRY(a->dTab0.reserve(pk.size() + 64));
            if (pk.pageable) RV_HIP(hipMemcpyAsync(a->dTab0.p, pk.data(), pk.size(), hipMemcpyHostToDevice, h->ws.stream));
            else { pk.grow(pk.size() + 16); RV_TRY(rv_h2d_copy(h->ws, pk.data(), a->dTab0.p, pk.size())); }
            // (the staging buffer is reused by the first commit, which comes after the host has waited for the first scan's result)
            const uint8_t *t0b = a->dTab0.as<uint8_t>();
            a->d_next_ss = (const int64_t *)(t0b + o1); a->d_next_nodes = (const sa_t *)(t0b + o2); a->d_next_flags = t0b + o3; a->d_next_tsub2 = (const int *)(t0b + o4);
            a->cur_dev_ok = true;
        }
    }
    RV_TRY(builtin_leaf_setup(h));
    a->running = true;
    return 0;
}

// rv_set_picker(1): the chain picker's call for one sub-index of a level (schemes.py:197-361 by rv_pick_chain).  The calls of a level are independent of
// each other -- each reads its sub-index' match list and intervals, and leaves a choice and two seed lists of its own -- so a level with enough work runs
// them on a few host threads (the reference serialises its Python callbacks behind one mutex, reveal.c:779; here the picker is C++ and needs none):
// 230 765 calls, 0.88 of the 1.34 s of a 5 x 5 Mbp recursion, were made one after the other between the level's scan and its commit.
struct PickScratch {
    std::vector<u32> pk_l, pk_sl; std::vector<int32_t> pk_n, pk_sn; std::vector<int64_t> pk_off, pk_mpos, pk_sb, pk_ib, pk_ie, pk_soff, pk_spos, pk_ssc;
    std::vector<uint16_t> pk_mso, pk_sso; std::vector<uint8_t> pk_srt;
};
struct PickRes {
    int rc = -9;                       // rv_pick_chain's return value (-9: not called)
    u32 l = 0; int members = 0;
    std::vector<uint16_t> so; std::vector<int64_t> pos;
    double t_list = 0, t_pick = 0;
    std::string err;
};
static void pick_one(rv_index *h, int s, PickScratch &X, PickRes &R, RvGraphIv gleft = RvGraphIv{-1, -1}, RvGraphIv gright = RvGraphIv{-1, -1}) {
    Align *a = h->al;
    const Level &lv = a->lv;
    const int W = h->nsamples, want = lv.nsamples[(size_t)s];
    const RvIntv *nodes = lv.nodes.data() + lv.node_first[(size_t)s];
    const size_t nn = (size_t)(lv.node_first[(size_t)s + 1] - lv.node_first[(size_t)s]);
    const int64_t first = a->mum_first[(size_t)s], cnt = a->nmums[(size_t)s];
    Align::SeedList &sl = a->seeds_lead[(size_t)s], &st = a->seeds_trail[(size_t)s];
    R.so.assign((size_t)W, 0); R.pos.assign((size_t)W, 0);
    // the list as rv_sub_mums hands it out
    const double tp0 = now_s();
    X.pk_l.clear(); X.pk_n.clear(); X.pk_off.assign(1, 0); X.pk_mso.clear(); X.pk_mpos.clear();
    if (!a->multi) {
        for (int64_t k = first; k < first + cnt; k++) {
            const RvPairRec &r = a->recs[(size_t)k];
            X.pk_l.push_back(r.l); X.pk_n.push_back(2);
            X.pk_mso.push_back(0); X.pk_mpos.push_back((int64_t)r.a); X.pk_mso.push_back(1); X.pk_mpos.push_back((int64_t)r.b);
            X.pk_off.push_back((int64_t)X.pk_mpos.size());
        }
    } else {
        for (int64_t k = first; k < first + cnt; k++) {
            X.pk_l.push_back(a->ml[(size_t)k]); X.pk_n.push_back(a->mn[(size_t)k]);
            for (int64_t qq = a->moff[(size_t)k]; qq < a->moff[(size_t)k + 1]; qq++) { X.pk_mso.push_back(a->mso[(size_t)qq]); X.pk_mpos.push_back(a->mpos[(size_t)qq]); }
            X.pk_off.push_back((int64_t)X.pk_mpos.size());
        }
    }
    X.pk_sb.assign((size_t)W, 0); X.pk_ib.assign((size_t)W, -1); X.pk_ie.assign((size_t)W, -1);
    if (a->picker == 1) for (int q2 = 0; q2 < W; q2++) X.pk_sb[(size_t)q2] = h->nodes[(size_t)q2].begin;
    for (size_t k = 0; k < nn && a->picker == 1; k++) {
        const int sm = sample_of(h, nodes[k].begin);
        if (X.pk_ib[(size_t)sm] >= 0) { R.rc = -1; R.err = "the native picker takes one interval per sample and sub-index"; return; }
        X.pk_ib[(size_t)sm] = nodes[k].begin; X.pk_ie[(size_t)sm] = nodes[k].end;
    }
    const size_t scap = X.pk_l.size(), mcap = X.pk_mpos.size();
    X.pk_sl.resize(scap); X.pk_sn.resize(scap); X.pk_soff.resize(scap + 1); X.pk_sso.resize(std::max<size_t>(mcap, 1)); X.pk_spos.resize(std::max<size_t>(mcap, 1));
    X.pk_ssc.resize(scap); X.pk_srt.resize(scap);
    rv_picker_out po;
    memset(&po, 0, sizeof po);
    po.pick_so = R.so.data(); po.pick_pos = R.pos.data(); po.member_cap = W;
    po.seed_cap = (int64_t)scap; po.seed_member_cap = (int64_t)std::max<size_t>(mcap, 1);
    po.seed_l = X.pk_sl.data(); po.seed_n = X.pk_sn.data(); po.seed_off = X.pk_soff.data(); po.seed_so = X.pk_sso.data(); po.seed_pos = X.pk_spos.data();
    po.seed_score = X.pk_ssc.data(); po.seed_right = X.pk_srt.data();
    const double tp1 = now_s();
    auto graph_pick = [&]() -> int {      // (host containers grow in there: no exception may leave through the C ABI)
        try { return rv_graph_do_pick(a->ggraph, &a->pargs, want, (int64_t)X.pk_l.size(), X.pk_l.data(), X.pk_n.data(), X.pk_off.data(), X.pk_mso.data(), X.pk_mpos.data(), gleft, gright, a->minl, &po); }
        catch (const std::exception &e) { rv_set_error("graph picker: %s", e.what()); return -1; }
        catch (...) { rv_set_error("graph picker failed"); return -1; }
    };
    const int pr = a->picker == 2
        ? graph_pick()
        : rv_pick_chain(&a->pargs, want, (int64_t)X.pk_l.size(), X.pk_l.data(), X.pk_n.data(), X.pk_off.data(), X.pk_mso.data(), X.pk_mpos.data(), W,
                        X.pk_sb.data(), X.pk_ib.data(), X.pk_ie.data(), a->minl, &po);
    R.t_pick = now_s() - tp1; R.t_list = tp1 - tp0;
    R.rc = pr;
    if (pr < 0) { R.err = rv_last_error(); return; }      // (the error text is this thread's: the caller sets it again on its own)
    if (pr == 1) {
        R.l = po.pick_l; R.members = po.pick_members;
        for (int64_t k = 0; k < po.nleft + po.nright; k++)
            (X.pk_srt[(size_t)k] ? st : sl).push(X.pk_sl[(size_t)k], X.pk_sn[(size_t)k], X.pk_sso.data() + X.pk_soff[(size_t)k], X.pk_spos.data() + X.pk_soff[(size_t)k],
                                                 (int)(X.pk_soff[(size_t)k + 1] - X.pk_soff[(size_t)k]), X.pk_ssc[(size_t)k]);
    }
}
// every not-seeded sub-index of the level with matches; on several threads when the level holds enough of them (RV_PICK_THREADS: 0 = up to eight, 1 = none)
static void pick_level(rv_index *h, std::vector<PickRes> &res) {
    Align *a = h->al;
    const int ns = a->lv.size();
    res.assign((size_t)ns, PickRes());
    a->seeds_lead.assign((size_t)ns, Align::SeedList()); a->seeds_trail.assign((size_t)ns, Align::SeedList());
    std::vector<int> todo;
    int64_t work = 0;
    for (int s = 0; s < ns; s++) {
        if (s < (int)a->seeds_cur.size() && a->seeds_cur[(size_t)s].size() > 0) continue;      // (seeded: the middle of its list, in the level loop)
        if (a->nmums[(size_t)s] <= 0) continue;
        todo.push_back(s); work += a->nmums[(size_t)s];
    }
    int nt = (int)h->ws.opt.pick_threads;
    if (nt <= 0) nt = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    nt = (int)std::min<size_t>((size_t)nt, todo.size());
    if (work < 4096) nt = 1;
    if (nt <= 1) {
        PickScratch X;
        for (int s : todo) pick_one(h, s, X, res[(size_t)s]);
        return;
    }
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        PickScratch X;
        for (;;) {
            const size_t i = next.fetch_add(16);      // (sixteen sub-indices a turn: the deep levels hold tens of thousands of small ones)
            if (i >= todo.size()) return;
            for (size_t j = i; j < std::min(todo.size(), i + 16); j++) pick_one(h, todo[j], X, res[(size_t)todo[j]]);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) {
        try { th.emplace_back(worker); } catch (...) { break; }      // (no thread to be had: the others, or this one alone, do the work)
    }
    worker();
    for (auto &t : th) t.join();
}

// levels of the recursion until the frontier is empty, or (stop_subs > 0) until it holds at least stop_subs sub-indices
static int builtin_levels(rv_index *h, int stop_subs) {
    Align *a = h->al;
    std::vector<sa_t> hsa; std::vector<lcp_t> hlcp;
    std::vector<int64_t> sp;
    std::vector<RvIntv> lead, trail, match, rest;
    std::vector<uint8_t> touched;
    std::vector<int64_t> pk_pos; std::vector<uint16_t> pk_so;
    std::vector<PickRes> pres; PickScratch gpx;
    const bool use_leaf = a->use_leaf;
    hipStream_t q = h->ws.stream;
    int &leaf_flip = a->leaf_flip;
    const bool level_log = (h->ws.opt.level_log != 0);      // diagnostics: per-level wall time (adds a sync per level)
    while (a->lv.size() > 0) {
        if (stop_subs > 0 && a->level > 0 && a->lv.size() >= stop_subs) break;      // hand-off point (rv_frontier_export)
        const double tl0 = level_log ? now_s() : 0.0;
        const int log_ns = a->lv.size(); const int64_t log_m = a->lv.m; const int log_level = a->level;
        size_t log_leaf = 0;
        if (use_leaf) {
            const Level &lv0 = a->lv;
            a->leaf_done.assign((size_t)lv0.size(), 0);
            std::vector<RvLeafRoot> &roots = a->leaf_roots[leaf_flip];
            roots.clear();
            for (int s = 0; s < lv0.size(); s++) {
                if (lv0.n[(size_t)s] > RV_LEAF_N) continue;
                const int64_t nf = lv0.node_first[(size_t)s], nn = lv0.node_first[(size_t)s + 1] - nf;
                if (nn < 1 || nn > 2) continue;
                RvLeafRoot r; r.off = lv0.off[(size_t)s]; r.n = (int32_t)lv0.n[(size_t)s]; r.depth = lv0.depth[(size_t)s];
                r.a0 = r.a1 = r.b0 = r.b1 = 0;
                bool ok = true;
                for (int64_t k = 0; k < nn && ok; k++) {
                    const RvIntv iv = lv0.nodes[(size_t)(nf + k)];
                    if (iv.begin < h->nsep[0]) { if (r.a0 < r.a1) ok = false; r.a0 = iv.begin; r.a1 = iv.end; }
                    else if (iv.begin > h->nsep[0]) { if (r.b0 < r.b1) ok = false; r.b0 = iv.begin; r.b1 = iv.end; }
                    else ok = false;
                }
                if (!ok) continue;
                a->leaf_done[(size_t)s] = 1;
                roots.push_back(r);
            }
            if (!roots.empty()) {
                // second stream: starts once the level arrays are complete (the event is recorded here, in front of this level's
                // kernels), runs beside this level's scan / split / bubble.  The launch itself (staging of the roots, copy,
                // kernel, events: ~15 us of host time) is issued after the level's own kernels are queued, while the host would
                // otherwise only wait for the picks -- issued first, it left the main stream idle that long at every level.
                RV_HIP(hipEventRecord(a->ev_ready, q));
                RV_HIP(hipStreamWaitEvent(leaf_flip ? a->leaf_stream2 : a->leaf_stream, a->ev_ready, 0));
                a->leaf_launch_due = true;
                log_leaf = roots.size();
                if ((size_t)lv0.size() == roots.size()) {       // nothing left for the level path
                    RV_TRY(leaf_launch(h));
                    a->st.levels++;
                    a->lv.clear();
                    break;
                }
            }
        }
        const double tl_leaf = level_log ? now_s() : 0.0;
        RV_TRY(rv_frontier_scan(h));
        if (a->leaf_launch_due) RV_TRY(leaf_launch(h));      // (scan paths without the hook)
        const double t0 = now_s();
        const Level &lv = a->lv;
        if (a->trace_on) {
            hsa.resize((size_t)lv.m); hlcp.resize((size_t)lv.m);
            RV_HIP(hipMemcpy(hsa.data(), cur_sa(h), (size_t)lv.m * sizeof(sa_t), hipMemcpyDeviceToHost));
            RV_HIP(hipMemcpy(hlcp.data(), cur_lcp(h), (size_t)lv.m * sizeof(lcp_t), hipMemcpyDeviceToHost));
        }
        a->st.levels++;
        const int ns = lv.size();
        if (a->picker == 1) pick_level(h, pres);
        if (a->picker == 2) {
            // graph inputs: a pick reads node offsets graphalign's surgery of the calls before it may have set, so the calls are made here, one sub-index after the
            // other in the reference's order (pick, then graphalign)
            a->seeds_lead.assign((size_t)ns, Align::SeedList()); a->seeds_trail.assign((size_t)ns, Align::SeedList());
            a->g_newleft.assign((size_t)ns, RvGraphIv{-1, -1}); a->g_newright.assign((size_t)ns, RvGraphIv{-1, -1});
            if ((int)a->g_left.size() != ns || (int)a->g_right.size() != ns) { rv_set_error("graph picker: the level's left / right nodes are missing"); return -1; }
            pres.assign(1, PickRes());
        }
        for (int s = 0; s < ns; s++) {
            if (use_leaf && a->leaf_done[(size_t)s]) continue;         // finished (with its whole sub-tree) by the leaf kernel
            a->st.steps++;
            if (lv.depth[(size_t)s] > a->st.maxdepth) a->st.maxdepth = lv.depth[(size_t)s];
            const RvIntv *nodes = lv.nodes.data() + lv.node_first[(size_t)s];
            const size_t nn = (size_t)(lv.node_first[(size_t)s + 1] - lv.node_first[(size_t)s]);
            const int64_t first = a->mum_first[(size_t)s], cnt = a->nmums[(size_t)s];
            rv_trace tr;
            if (a->trace_on) {
                memset(&tr, 0, sizeof tr);
                tr.key = nn ? nodes[0].begin : -1;
                tr.n = lv.n[(size_t)s]; tr.depth = lv.depth[(size_t)s]; tr.nsamples = lv.nsamples[(size_t)s]; tr.nnodes = (int32_t)nn; tr.nmums = cnt;
                u64 h1 = 0, h2 = 0, h3 = 0, c3 = 0;
                for (int64_t i = 0; i < lv.n[(size_t)s]; i++) { h1 = hash_step(h1, (u64)i, (int64_t)hsa[(size_t)(lv.off[(size_t)s] + i)]); h2 = hash_step(h2, (u64)i, (int64_t)hlcp[(size_t)(lv.off[(size_t)s] + i)]); }
                for (int64_t k = first; k < first + cnt; k++) {
                    if (!a->multi) {
                        const RvPairRec &r = a->recs[(size_t)k];
                        const int64_t seq[6] = {(int64_t)r.l, 2, 0, (int64_t)r.a, 1, (int64_t)r.b};
                        for (int z = 0; z < 6; z++) h3 = hash_step(h3, c3++, seq[z]);
                    } else {
                        h3 = hash_step(h3, c3++, (int64_t)a->ml[(size_t)k]); h3 = hash_step(h3, c3++, a->mn[(size_t)k]);
                        for (int64_t qq = a->moff[(size_t)k]; qq < a->moff[(size_t)k + 1]; qq++) { h3 = hash_step(h3, c3++, a->mso[(size_t)qq]); h3 = hash_step(h3, c3++, a->mpos[(size_t)qq]); }
                    }
                }
                tr.h_sa = h1; tr.h_lcp = h2; tr.h_mums = h3;
            }
            // picker: longest match present in every sample of the sub-index, ties -> smallest minimum coordinate
            int64_t best = -1, bmin = 0; u32 bl = 0;
            const int want = lv.nsamples[(size_t)s];
            bool chain_pick = false; int pk_members = 0;
            if (a->picker != 0) {
                // the reference's default picker (schemes.py:197-361) in C++: rv_pick_chain on the sub-index' whole list (pick_level, above) -- or, for a
                // sub-index its parent seeded, the middle of that list (schemes.py:349-354) and its two halves for the children
                const int W = h->nsamples;
                pk_so.assign((size_t)W, 0); pk_pos.assign((size_t)W, 0);
                int members = 0;
                Align::SeedList &sl = a->seeds_lead[(size_t)s], &st = a->seeds_trail[(size_t)s];
                a->picker_calls++;
                if (s < (int)a->seeds_cur.size() && a->seeds_cur[(size_t)s].size() > 0) {
                    const Align::SeedList &sd = a->seeds_cur[(size_t)s];
                    const size_t cnt2 = sd.size(), mid = cnt2 / 2;
                    a->picker_seeded++;
                    sl.clear(); st.clear();
                    bl = sd.l[mid]; members = (int)(sd.off[mid + 1] - sd.off[mid]);
                    for (int q2 = 0; q2 < members; q2++) { pk_so[(size_t)q2] = sd.so[(size_t)sd.off[mid] + q2]; pk_pos[(size_t)q2] = sd.pos[(size_t)sd.off[mid] + q2]; }
                    for (size_t k = 0; k < cnt2; k++) {
                        if (k == mid) continue;
                        (k < mid ? sl : st).push(sd.l[k], sd.n[k], sd.so.data() + sd.off[k], sd.pos.data() + sd.off[k], (int)(sd.off[k + 1] - sd.off[k]), sd.score[k]);
                    }
                    chain_pick = true;
                } else if (cnt > 0) {
                    if (a->picker == 2) { pres[0] = PickRes(); pick_one(h, s, gpx, pres[0], a->g_left[(size_t)s], a->g_right[(size_t)s]); }
                    const PickRes &R = pres[a->picker == 2 ? 0 : (size_t)s];
                    a->picker_ns += (int64_t)(R.t_pick * 1e9); a->picker_list_ns += (int64_t)(R.t_list * 1e9);
                    if (R.rc == -9) { rv_set_error("the native picker was not run for sub-index %d", s); return -1; }
                    if (R.rc < 0) { rv_set_error("%s", R.err.c_str()); return -1; }
                    if (R.rc == 1) {
                        bl = R.l; members = R.members; chain_pick = true;
                        for (int q2 = 0; q2 < members; q2++) { pk_so[(size_t)q2] = R.so[(size_t)q2]; pk_pos[(size_t)q2] = R.pos[(size_t)q2]; }
                    }
                }
                if (chain_pick) { sp.assign(pk_pos.begin(), pk_pos.begin() + members); best = 0; pk_members = members; }
            } else
            if (!a->multi) {
                if (want == 2)
                    for (int64_t k = first; k < first + cnt; k++) {
                        const RvPairRec &r = a->recs[(size_t)k];
                        if (best < 0 || r.l > bl || (r.l == bl && (int64_t)r.a < bmin)) { best = k; bl = r.l; bmin = (int64_t)r.a; }
                    }
            } else {
                for (int64_t k = first; k < first + cnt; k++) {
                    if (a->mn[(size_t)k] != want) continue;
                    int64_t mnp = a->mpos[(size_t)a->moff[(size_t)k]];
                    for (int64_t qq = a->moff[(size_t)k] + 1; qq < a->moff[(size_t)k + 1]; qq++) mnp = std::min(mnp, a->mpos[(size_t)qq]);
                    if (best < 0 || a->ml[(size_t)k] > bl || (a->ml[(size_t)k] == bl && mnp < bmin)) { best = k; bl = a->ml[(size_t)k]; bmin = mnp; }
                }
            }
            if (best >= 0) {
                // graphalign, linear interval model
                lead.clear(); trail.clear(); match.clear(); rest.clear();
                if (chain_pick) {}      // (sp holds the choice's members)
                else if (!a->multi) { sp.clear(); sp.push_back((int64_t)a->recs[(size_t)best].a); sp.push_back((int64_t)a->recs[(size_t)best].b); }
                else sp.assign(a->mpos.begin() + a->moff[(size_t)best], a->mpos.begin() + a->moff[(size_t)best + 1]);
                std::sort(sp.begin(), sp.end());
                if (a->picker == 2) {
                    // graphalign on the graph (rem.py:318-382): break the nodes that hold the members, merge the matched pieces, find the children's intervals
                    // by walking the graph around the merged node
                    RvGraphAlignOut &GO = a->g_out;
                    static_assert(sizeof(RvGraphIv) == sizeof(RvIntv), "interval layouts");
                    const double tg0 = now_s();
                    int grc;
                    try { grc = rv_graph_do_align(a->ggraph, (const RvGraphIv *)nodes, nn, a->g_left[(size_t)s], a->g_right[(size_t)s], bl, pk_pos.data(), (int)sp.size(), GO); }
                    catch (const std::exception &e) { rv_set_error("graphalign: %s", e.what()); grc = -1; }
                    catch (...) { rv_set_error("graphalign failed"); grc = -1; }
                    RV_TRY(grc);
                    a->galign_ns += (int64_t)((now_s() - tg0) * 1e9);
                    a->g_newleft[(size_t)s] = GO.newleft; a->g_newright[(size_t)s] = GO.newright;
                    auto cp = [](std::vector<RvIntv> &d, const std::vector<RvGraphIv> &v) { for (const RvGraphIv &x : v) d.push_back({x.b, x.e}); };
                    cp(lead, GO.lead); cp(trail, GO.trail); cp(match, GO.match); cp(rest, GO.rest);
                    sp.erase(std::unique(sp.begin(), sp.end()), sp.end());
                } else {
                touched.assign(nn, 0);
                for (int64_t p : sp) {
                    size_t lo = 0, hi = nn;
                    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (nodes[mid].begin <= p) lo = mid + 1; else hi = mid; }
                    if (lo == 0 || p >= nodes[lo - 1].end || p + (int64_t)bl > nodes[lo - 1].end) { rv_set_error("match at %lld is not inside an interval of its sub-index", (long long)p); return -1; }
                    const RvIntv iv = nodes[lo - 1];
                    touched[lo - 1] = 1;
                    if (p > iv.begin) lead.push_back({iv.begin, p});
                    if (p + (int64_t)bl < iv.end) trail.push_back({p + (int64_t)bl, iv.end});
                    match.push_back({p, p + (int64_t)bl});
                }
                for (size_t k = 0; k < nn; k++) if (!touched[k]) rest.push_back(nodes[k]);
                }
                RV_TRY(add_decision(h, s, bl, sp.data(), (int)sp.size(), lead.data(), (int)lead.size(), trail.data(), (int)trail.size(),
                                    match.data(), (int)match.size(), rest.data(), (int)rest.size()));
                a->an_l.push_back(bl);
                if (chain_pick) a->an_pos.insert(a->an_pos.end(), pk_pos.begin(), pk_pos.begin() + (ptrdiff_t)pk_members);      // (the picker's member order: graphalign merges the nodes in it)
                else
                a->an_pos.insert(a->an_pos.end(), sp.begin(), sp.end());
                a->an_off.push_back((int64_t)a->an_pos.size());
                a->st.splits++; a->st.anchored_bp += bl;
                if (a->trace_on) { tr.picked = 1; tr.l = bl; tr.mn = (int32_t)sp.size(); tr.sp_min = sp[0]; }
            }
            if (a->trace_on) a->trace.push_back(tr);
        }
        a->st.t_host += now_s() - t0;
        if (a->feed && a->an_l.size() > a->fed) {      // this level's anchors to the surgery's thread
            rv_replay_feed_push(a->feed, a->an_l.data() + a->fed, a->an_off.data() + a->fed, a->an_pos.data() + a->an_off[a->fed], a->an_l.size() - a->fed);
            a->fed = a->an_l.size();
        }
        const double tl1 = level_log ? now_s() : 0.0;
        RV_TRY(rv_frontier_commit(h, nullptr));
        if (a->picker == 2) {
            // the children's left / right graph nodes (reveal.c:884-950: leading (parent's left, newright), trailing (newleft, parent's right), the others the parent's)
            const int nn2 = a->lv.size();
            std::vector<RvGraphIv> gl((size_t)nn2, RvGraphIv{-1, -1}), gr((size_t)nn2, RvGraphIv{-1, -1});
            for (int s2 = 0; s2 < nn2; s2++) {
                const int p = a->lv.parent[(size_t)s2], kd = a->lv.kind[(size_t)s2];
                if (p < 0 || p >= (int)a->g_left.size()) continue;
                gl[(size_t)s2] = kd == 2 ? a->g_newleft[(size_t)p] : a->g_left[(size_t)p];
                gr[(size_t)s2] = kd == 1 ? a->g_newright[(size_t)p] : a->g_right[(size_t)p];
            }
            a->g_left.swap(gl); a->g_right.swap(gr);
        }
        if (a->picker != 0) {
            // a child whose parent left it a list is not scanned (reveal.c:802, 830-837): its picker call takes the list's middle
            const int nn2 = a->lv.size();
            std::vector<Align::SeedList> nxt((size_t)nn2);
            for (int s2 = 0; s2 < nn2; s2++) {
                const int p = a->lv.parent[(size_t)s2], kd = a->lv.kind[(size_t)s2];
                if (p < 0 || p >= (int)a->seeds_lead.size()) continue;
                Align::SeedList &src = kd == 1 ? a->seeds_lead[(size_t)p] : a->seeds_trail[(size_t)p];
                if ((kd == 1 || kd == 2) && src.size() > 0) {
                    nxt[(size_t)s2] = std::move(src);
                    if ((int)a->skip_scan.size() != nn2) a->skip_scan.assign((size_t)nn2, 0);
                    a->skip_scan[(size_t)s2] = 1;
                }
            }
            a->seeds_cur.swap(nxt);
            a->seeds_lead.clear(); a->seeds_trail.clear();
        }
        if (level_log) {
            const double tl2 = now_s();
            (void)hipStreamSynchronize(q);
            const double tl3 = now_s();
            int64_t biggest = 0;
            for (int s2 = 0; s2 < a->lv.size(); s2++) biggest = std::max<int64_t>(biggest, a->lv.n[(size_t)s2]);
            unsigned long long dbg[8] = {0};
            if (a->dDbg.p) { (void)hipMemcpy(dbg, a->dDbg.p, 64, hipMemcpyDeviceToHost); (void)hipMemset(a->dDbg.p, 0, 64); }
            fprintf(stderr, "      bubble (sequential kernels): cuts %llu actives %llu whole-wg visits %llu chunks %llu concurrent %llu | slowest child: total %.1f us (%llu actives), finding actives %.1f, sort+visits %.1f\n", dbg[3], dbg[4], dbg[0], dbg[1], dbg[2], (dbg[5] >> 24) / 100.0, dbg[5] & 0xFFFFFFull, dbg[6] / 100.0, dbg[7] / 100.0);
            fprintf(stderr, "      upload: tile_sub %.1f pack %.1f next-level tables %.1f offsets %.1f copy %.1f us (%zu bytes)\n", (a->lgx[0] - a->lg[0]) * 1e6, (a->lgx[1] - a->lgx[0]) * 1e6,
                    (a->lgx[2] - a->lgx[1]) * 1e6, (a->lgx[3] - a->lgx[2]) * 1e6, (a->lg[1] - a->lgx[3]) * 1e6, a->pk.size());
            fprintf(stderr, "level %3d subs %7d (leaf %7zu) ranks %10lld | leafprep %6.1f scan %6.1f host %6.1f | commit: tables %6.1f upload %6.1f split-enq %6.1f bubble-enq %6.1f | drain %7.1f us | next biggest %lld\n",
                    log_level, log_ns, log_leaf, (long long)log_m, (tl_leaf - tl0) * 1e6, (t0 - tl_leaf) * 1e6, (tl1 - t0) * 1e6,
                    a->lg[0] * 1e6, (a->lg[1] - a->lg[0]) * 1e6, (a->lg[2] - a->lg[1]) * 1e6, (tl2 - tl1) * 1e6 - a->lg[2] * 1e6, (tl3 - tl2) * 1e6,
                    (long long)biggest);
        }
    }
    // drained, and the deferred error word of the last commit looked at: a hand-off must not export a broken level
    RV_HIP(hipStreamSynchronize(q));
    if (a->lv.size() > 0) {
        u32 err = 0;
        RV_HIP(hipMemcpy(&err, a->dErr.p, 4, hipMemcpyDeviceToHost));
        if (err & 1u) { rv_set_error("split: the intervals returned by graphalign do not partition the sub-index (child size mismatch)"); return -1; }
        if (err) { rv_set_error("device-side error %u in the level before the hand-off", err); return -1; }
    }
    return 0;
}

// off[k] = base + 2 (k + 1), k < na: the member offsets of na two-member anchors, on a few host threads
static void fill_offsets(int64_t *off, size_t na, int64_t base) {
    const int nt = na > ((size_t)1 << 18) ? 4 : 1;
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) {
        const size_t lo = na / nt * t, hi = t + 1 == nt ? na : na / nt * (t + 1);
        th.emplace_back([=]() { for (size_t k = lo; k < hi; k++) off[k] = base + 2 * (int64_t)(k + 1); });
    }
    for (size_t k = 0; k < na / nt; k++) off[k] = base + 2 * (int64_t)(k + 1);
    for (auto &x : th) x.join();
}
// collect what the leaf launches produced
static int builtin_finish(rv_index *h, rv_align_stats *out) {
    Align *a = h->al;
    hipStream_t q = h->ws.stream;
    h->rb.direct = false;
    if (a->use_leaf && a->running) {
        u32 *lf_counters = a->lf_counters; u32 *lf_l = a->lf_l; int64_t *lf_pos = a->lf_pos; rv_trace *lf_tr = a->lf_tr;
        // everything through pinned staging (pageable destinations are staged by the runtime, copy by copy: ~0.25 ms per run)
        RV_HIP(hipStreamSynchronize(a->leaf_stream));
        RV_HIP(hipStreamSynchronize(a->leaf_stream2));
        for (int k = 0; k <= RV_LEVEL_BUFS; k++) a->leaf_pending[k] = false;
        RV_TRY(a->hLeafOut.reserve(256));
        RV_HIP(hipMemcpyAsync(a->hLeafOut.p, lf_counters, 128, hipMemcpyDeviceToHost, q));      // counters at +0, statistics at +64
        RV_HIP(hipStreamSynchronize(q));
        u32 cnt[4]; unsigned long long stv[8];
        memcpy(cnt, a->hLeafOut.p, sizeof cnt); memcpy(stv, a->hLeafOut.as<uint8_t>() + 64, sizeof stv);
        if (h->ws.opt.leaf_prof)      // (a build with -DRV_LEAF_PROF: wave cycles waiting for a sub-index / scan + pick / split / bubble + children)
            fprintf(stderr, "leaf: steps %llu splits %llu | cycles idle %llu scan %llu split %llu bubble %llu\n", stv[0], stv[1], stv[4], stv[5], stv[6], stv[7]);
        if (cnt[2]) { rv_set_error(cnt[2] & 4u ? "leaf kernel: recursion stack overflow" : "leaf kernel: a sub-index does not match its intervals"); return -1; }
        if (cnt[0] > a->leaf_anchor_cap || cnt[1] > a->leaf_trace_cap) { rv_set_error("leaf kernel: output buffer too small"); return -1; }
        a->leaf_na = 0;
        if (cnt[0]) {
            // the anchors stay in the pinned buffer in the layout rv_fetch_anchors hands out (2 x 10^6 of them at 2 x 250 Mbp: appending
            // them to the host vectors one by one cost 5 ms per run, most of it page faults of the freshly grown vectors)
            const size_t na = cnt[0];
            const size_t nl = a->an_l.size(), np = a->an_pos.size();
            // The caller's own arrays when it has set them and they are large enough (rv_set_result_buffers: page-locked): the anchors go
            // straight to where rv_fetch_anchors would copy them -- through the staging buffer the 2 x 10^6 anchors of 2 x 250 Mbp were
            // 65 MB of host writes behind the run, 1.3-1.6 ms with the GPU idle (tools/rocpd_gaps.py)
            rv_index::ResultBufs &rb = h->rb;
            rb.direct = rb.l && (int64_t)(nl + na) <= rb.l_cap && (int64_t)(nl + na + 1) <= rb.off_cap && (int64_t)(np + 2 * na) <= rb.pos_cap;
            int64_t *pp; u32 *pl;
            if (rb.direct) { pp = rb.pos + np; pl = rb.l + nl; }
            else { RV_TRY(a->hLeafOut.reserve(na * 20 + 64)); pp = a->hLeafOut.as<int64_t>(); pl = (u32 *)(pp + 2 * na); }
            // (both streams are idle here: the copies to the host run on the side stream beside the lower-casing -- 0.9 and 0.7 ms at 2 x 250 Mbp)
            RV_HIP(hipMemcpyAsync(pp, lf_pos, na * 16, hipMemcpyDeviceToHost, a->leaf_stream));
            RV_HIP(hipMemcpyAsync(pl, lf_l, na * 4, hipMemcpyDeviceToHost, a->leaf_stream));
            RV_TRY(rv_leaf_lower_launch(h->ws, h->dT.as<uint8_t>(), lf_pos, lf_l, (u32)na));      // their matched text (nothing read it during the run)
            if (rb.direct) fill_offsets(rb.off + nl + 1, na, (int64_t)np);      // (two members per anchor: the host writes them while the GPU works)
            RV_HIP(hipStreamSynchronize(a->leaf_stream));
            RV_HIP(hipStreamSynchronize(q));
            a->leaf_na = na;
        }
        if (a->trace_on && cnt[1]) {
            const size_t at = a->trace.size();
            a->trace.resize(at + cnt[1]);
            RV_HIP(hipMemcpy(a->trace.data() + at, lf_tr, (size_t)cnt[1] * sizeof(rv_trace), hipMemcpyDeviceToHost));
        }
        a->st.steps += (int64_t)stv[0]; a->st.splits += (int64_t)stv[1]; a->st.anchored_bp += (int64_t)stv[2];
        if ((int64_t)stv[3] > a->st.maxdepth) a->st.maxdepth = (int32_t)stv[3];
    }
    a->running = false;
    if (out) *out = a->st;
    return 0;
}

// The anchor cascade (rv_cascade.hip) in front of the level pipeline: an untraced two-sample run with one sequence per sample
// is decided from the top-level match list wherever that is provably the reference's result; what is left undecided is rebuilt
// from its text and finished by the leaf kernel.  It either does the whole run or leaves no trace (RV_NO_CASCADE=1: never tried).
static int install_frontier(rv_index *h, int level, int nsubs, const int64_t *meta, const int64_t *node_first, const int64_t *nodes, int64_t m,
                            const void *sa, const void *lcp, const void *bwt, int on_device);
static int builtin_cascade(rv_index *h) {
    Align *a = h->al;
    memset(&a->cas_out, 0, sizeof a->cas_out);
    if (a->trace_on || h->rc != 0 || h->n <= RV_LEAF_N || a->minl < 4 || h->ws.opt.no_cascade || a->picker != 0) return 0;
    const bool second_try = !a->multi && a->use_leaf && h->ws.opt.cascade_second == 2;      // (test hook: straight to the second attempt)
    auto interval_cascade = [&]() -> int {
        // the decided part's anchors come back on the host, what is undecided becomes the frontier of the level pipeline
        RvCascadeMultiOut mo;
        RV_TRY(rv_cascade_multi_run(h, a->cas, a->minl, &mo));
        a->cas_out.done = mo.done; a->cas_out.levels = mo.levels; a->cas_out.cands = mo.cands; a->cas_out.witnesses = mo.witnesses; a->cas_out.children = mo.children;
        a->cas_out.undecided = mo.undecided; a->cas_out.rebuilt_ranks = mo.rebuilt_ranks; a->cas_out.why = mo.why;
        if (!mo.done) return 0;
        const int k = h->nsamples;
        a->st.splits += (int64_t)mo.an_l.size(); a->st.steps += mo.steps; a->st.levels += mo.levels; a->st.scanned_ranks += h->n;
        if (mo.maxdepth > a->st.maxdepth) a->st.maxdepth = mo.maxdepth;
        h->main_arrays_freed = true;
        if (mo.undecided > 0) {
            RV_TRY(install_frontier(h, 1, (int)mo.undecided, mo.meta.data(), mo.node_first.data(), mo.nodes.data(), mo.rebuilt_ranks, mo.d_sa, mo.d_lcp, mo.d_bwt, 1));
        } else {
            a->lv.clear();
            a->level = 1;
        }
        {      // the decided part's anchors behind the level pipeline's own (whole arrays at once, behind the frontier's launches: an anchor at a time the
               // host spent 0.3 ms here with the GPU idle at 10 x 5 Mbp)
            const size_t na = mo.an_l.size(), l0 = a->an_l.size(), p0 = a->an_pos.size(), o0 = a->an_off.size();
            a->an_l.insert(a->an_l.end(), mo.an_l.begin(), mo.an_l.end());
            a->an_pos.resize(p0 + na * (size_t)k);
            for (size_t x = 0; x < na * (size_t)k; x++) a->an_pos[p0 + x] = (int64_t)mo.an_pos[x];
            a->an_off.resize(o0 + na);
            for (size_t x = 0; x < na; x++) a->an_off[o0 + x] = (int64_t)(p0 + (x + 1) * (size_t)k);
            int64_t bp = 0;
            for (size_t x = 0; x < na; x++) bp += mo.an_l[x];
            a->st.anchored_bp += bp;
            (void)l0;
        }
        return 0;
    };
    if (a->multi || second_try) return interval_cascade();
    if (!a->use_leaf) return 0;
    RvCascadeIO io;
    io.anchor_count = a->lf_counters; io.anchor_cap = (u32)a->leaf_anchor_cap; io.anchor_l = a->lf_l; io.anchor_pos = a->lf_pos;
    io.stats = a->lf_stats; io.leaf_err = a->lf_counters + 2;
    io.stage_cap = (u32)h->ws.opt.leaf_acap;
    io.lvSA = &a->lvSA[0]; io.lvLCP = &a->lvLCP[0]; io.lvBWT = &a->lvBWT[0]; io.roots = &a->dLeafRoots[0];
    // RV_CASCADE_DANGER=0: no second attempt of this kind; =2: the first attempt already decides large undecided sub-indices from their witnesses (test hook)
    const int dmode = (int)h->ws.opt.cascade_danger;
    RV_TRY(rv_cascade_run(h, a->cas, io, a->minl, &a->cas_out, dmode == 2 ? 1 : 0, 0));
    if (!a->cas_out.done && a->cas_out.undecided > 0 && dmode == 1) {
        // an undecided sub-index above the leaf kernel's size: the same cascade again (its lists are still there), now with the large
        // undecided sub-indices decided from their witnesses (k_cas_dwalk)
        RV_HIP(hipMemsetAsync(a->dLeaf.p, 0, 256, h->ws.stream));      // anchors and counters of the attempt
        RV_TRY(rv_cascade_run(h, a->cas, io, a->minl, &a->cas_out, 1, 1));
    }
    if (!a->cas_out.done) {
        RV_HIP(hipMemsetAsync(a->dLeaf.p, 0, 256, h->ws.stream));      // anchors and counters of the attempt
        // still one left: another attempt with the bound of rv_cascade_multi.hip (repeats inside one
        // sample: tighter) whose undecided sub-indices -- up to 8192 suffixes -- go to the level pipeline instead of the leaf kernel
        if (a->cas_out.undecided > 0 && !h->ws.opt.cascade_second_off) return interval_cascade();
        return 0;
    }
    a->st.levels += a->cas_out.levels;
    a->st.scanned_ranks += h->n;
    if (!a->cas.lin_rest.empty() && h->nodes.size() > 2) {
        // several sequences per sample: the chain of rest sub-indices stopped at a member the match list does not decide (left-over sequences whose
        // only matches are as short as their repeats).  It becomes the level pipeline's frontier: ONE split of the root whose rest class is that
        // member's sequences -- the split's range minima over the dropped ranks are what the chain of splits in between would have left (split only
        // takes minima, reveal.c:582-664; bubble_sort touches leading children only) -- at the depth the chain had reached.
        // The sequences the chain has consumed are labelled like matched ranks: dropped WITH the update of the running minima.  Left unlabelled they
        // would take the reference's `continue` in front of that update (reveal.c:616-620, which in the reference only ever meets the '$' suffixes of
        // the root), and the member's LCP values would come out too large.
        std::vector<RvIntv> rest, gone;
        for (size_t k = 0; k + 1 < a->cas.lin_rest.size(); k += 2) rest.push_back({a->cas.lin_rest[k], a->cas.lin_rest[k + 1]});
        std::sort(rest.begin(), rest.end(), intv_less);
        for (const RvIntv &v : h->nodes) {
            if (v.end <= v.begin) continue;
            const auto it = std::lower_bound(rest.begin(), rest.end(), v, intv_less);
            if (it == rest.end() || it->begin != v.begin) gone.push_back(v);
        }
        RV_TRY(add_decision(h, 0, 0, nullptr, 0, gone.data(), (int)gone.size(), nullptr, 0, nullptr, 0, rest.data(), (int)rest.size()));
        a->dec.drop_lead = true;
        RV_TRY(rv_frontier_commit(h, nullptr));
        for (size_t s2 = 0; s2 < a->lv.depth.size(); s2++) a->lv.depth[s2] = a->cas.lin_rest_depth;
        return 0;
    }
    h->main_arrays_freed = true;                                         /* reveal.c:1279-1284: align() consumes the main index */
    a->level = 1;
    a->lv.clear();                                                       // nothing left for the level pipeline
    return 0;
}

int rv_cascade_info(const rv_index *h, int64_t *out) {
    for (int k = 0; k < 8; k++) out[k] = 0;
    if (!h->al) return 0;
    const RvCascadeOut &c = h->al->cas_out;
    out[0] = c.done ? 1 : 0; out[1] = c.levels; out[2] = c.cands; out[3] = c.witnesses; out[4] = c.children; out[5] = c.undecided; out[6] = c.rebuilt_ranks; out[7] = c.solved;
    return 0;
}

/* why the cascade left the last built-in run to the level pipeline ("" when it did the run, or was never tried) */
const char *rv_cascade_why(const rv_index *h) {
    if (!h->al || h->al->cas_out.done || !h->al->cas_out.why) return "";
    return h->al->cas_out.why;
}

int rv_align_builtin(rv_index *h, int minl, int minn, rv_align_stats *out) {
    RV_TRY(builtin_setup(h, minl, minn));
    Align *a = h->al;
    if (a->replay_graph && a->picker == 1) {
        a->feed = rv_replay_feed_start(a->replay_graph);
        a->fed = 0;
        a->replay_graph = nullptr;      // (one run)
        if (!a->feed) return -1;
    }
    int rc = builtin_cascade(h);
    if (rc == 0) rc = builtin_levels(h, 0);
    if (rc == 0) rc = builtin_finish(h, out);
    if (a->feed) {
        RvReplayFeed *f = a->feed; a->feed = nullptr;
        if (rc == 0) { if (rv_replay_feed_finish(f) != 0) return -1; }
        else { const std::string keep = rv_last_error(); (void)rv_replay_feed_finish(f); rv_set_error("%s", keep.c_str()); }
    }
    return rc;
}

/* ---- frontier hand-off (second granularity) ---------------------------
 * After a split the children of a sub-index cover disjoint text and disjoint ranges of the shared inverse, and never
 * read what a sibling writes (reveal.c:1230-1234 lower-cases before the push at :1296), so the sub-indices of a frontier
 * are independent units of work.  A built-in run can stop once the frontier is wide enough, hand any subset of its
 * sub-indices (metadata + their SA / LCP / BWT segments, 9 B per rank) to index handles on other devices that only hold the
 * text, and every handle finishes its share with the same level loop.  The union of the anchors is the anchor set of the
 * undivided run; the lower-cased text follows from the anchors.  No collective: segments travel point to point. */
int rv_align_builtin_until(rv_index *h, int minl, int minn, int stop_subs, rv_align_stats *out) {
    RV_TRY(builtin_setup(h, minl, minn));
    RV_TRY(builtin_cascade(h));      // (a two-sample run the cascade decides is finished before there is a frontier to hand out: returns 0)
    RV_TRY(builtin_levels(h, stop_subs));
    Align *a = h->al;
    if (a->lv.size() == 0) { RV_TRY(builtin_finish(h, out)); return 0; }
    if (out) *out = a->st;
    return a->lv.size();
}

/* a run stopped by rv_align_builtin_until goes on until its frontier holds at least stop_subs sub-indices: the owner of a divided
 * alignment widens the frontier while its largest sub-index is still too large a share for one device (reveal_amd/shard.py) */
int rv_align_builtin_continue(rv_index *h, int stop_subs, rv_align_stats *out) {
    RV_TRY(need_align(h));
    if (!h->al->running) { rv_set_error("no built-in run in progress (rv_align_builtin_until / rv_frontier_import)"); return -1; }
    if (stop_subs < 1) { rv_set_error("rv_align_builtin_continue: stop_subs must be positive"); return -1; }
    RV_HIP(hipSetDevice(h->device));
    Align *a = h->al;
    // (builtin_levels stops before a level when the frontier is already wide enough: ask for one sub-index more than there is)
    RV_TRY(builtin_levels(h, std::max(stop_subs, a->lv.size() + 1)));
    if (a->lv.size() == 0) { RV_TRY(builtin_finish(h, out)); return 0; }
    if (out) *out = a->st;
    return a->lv.size();
}

int rv_align_builtin_resume(rv_index *h, rv_align_stats *out) {
    RV_TRY(need_align(h));
    if (!h->al->running) { rv_set_error("no built-in run in progress (rv_align_builtin_until / rv_frontier_import)"); return -1; }
    RV_HIP(hipSetDevice(h->device));
    RV_TRY(builtin_levels(h, 0));
    return builtin_finish(h, out);
}

int rv_frontier_counts(rv_index *h, int64_t *out) {
    RV_TRY(need_align(h));
    const Level &lv = h->al->lv;
    out[0] = lv.size(); out[1] = lv.size() ? lv.m : 0; out[2] = (int64_t)lv.nodes.size(); out[3] = h->al->level;
    return 0;
}

/* meta: 6 numbers per sub-index (off, n, depth, nsamples, kind, parent); node_first: nsubs+1; nodes: (begin, end) pairs */
int rv_frontier_export(rv_index *h, int64_t *meta, int64_t *node_first, int64_t *nodes) {
    RV_TRY(need_align(h));
    if (h->al->picker == 2) { rv_set_error("rv_frontier_export: a run on a graph (rv_set_graph_picker) is not handed off -- its sub-indices share the graph"); return -1; }
    const Level &lv = h->al->lv;
    for (int s = 0; s < lv.size(); s++) {
        int64_t *m6 = meta + 6 * (size_t)s;
        m6[0] = lv.off[(size_t)s]; m6[1] = lv.n[(size_t)s]; m6[2] = lv.depth[(size_t)s]; m6[3] = lv.nsamples[(size_t)s]; m6[4] = lv.kind[(size_t)s]; m6[5] = lv.parent[(size_t)s];
    }
    for (size_t k = 0; k < lv.node_first.size(); k++) node_first[k] = lv.node_first[k];
    for (size_t k = 0; k < lv.nodes.size(); k++) { nodes[2 * k] = lv.nodes[k].begin; nodes[2 * k + 1] = lv.nodes[k].end; }
    return 0;
}

/* rank segments of the listed sub-indices, back to back in list order, into caller memory (device or host) */
int64_t rv_frontier_pack(rv_index *h, const int32_t *subs, int k, void *sa, void *lcp, void *bwt, int on_device) {
    if (need_align(h)) return -1;
    Align *a = h->al;
    if (hipSetDevice(h->device) != hipSuccess) { rv_set_error("hipSetDevice failed"); return -1; }
    hipStream_t q = h->ws.stream;
    const Level &lv = a->lv;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    int64_t at = 0;
    for (int i = 0; i < k;) {
        if (subs[i] < 0 || subs[i] >= lv.size()) { rv_set_error("sub-index %d out of range", (int)subs[i]); return -1; }
        int j = i;       // neighbours in the level arrays leave as one copy
        int64_t cnt = lv.n[(size_t)subs[i]];
        while (j + 1 < k && subs[j + 1] == subs[j] + 1) { j++; cnt += lv.n[(size_t)subs[j]]; }
        const int64_t off = lv.off[(size_t)subs[i]];
        hipError_t e = hipMemcpyAsync((sa_t *)sa + at, cur_sa(h) + off, (size_t)cnt * sizeof(sa_t), kind, q);
        if (e == hipSuccess) e = hipMemcpyAsync((lcp_t *)lcp + at, cur_lcp(h) + off, (size_t)cnt * sizeof(lcp_t), kind, q);
        if (e == hipSuccess) e = hipMemcpyAsync((uint8_t *)bwt + at, cur_bwt(h) + off, (size_t)cnt, kind, q);
        if (e != hipSuccess) { rv_set_error("rv_frontier_pack: copy failed: %s", hipGetErrorString(e)); return -1; }
        at += cnt;
        i = j + 1;
    }
    if (hipStreamSynchronize(q) != hipSuccess) { rv_set_error("rv_frontier_pack: stream failed"); return -1; }
    return at;
}

// the given sub-indices (segments back to back in sa / lcp / bwt) become the frontier of the handle; the per-level tables a
// commit would have shipped for them go with it
static int install_frontier(rv_index *h, int level, int nsubs, const int64_t *meta, const int64_t *node_first, const int64_t *nodes, int64_t m,
                            const void *sa, const void *lcp, const void *bwt, int on_device) {
    hipStream_t q = h->ws.stream;
    Align *a = h->al;
    RV_HIP(hipStreamSynchronize(q));
    if (a->leaf_stream) { RV_HIP(hipStreamSynchronize(a->leaf_stream)); RV_HIP(hipStreamSynchronize(a->leaf_stream2)); }      // (a leaf launch may read the level buffers replaced below)
    for (int k = 0; k <= RV_LEVEL_BUFS; k++) a->leaf_pending[k] = false;
    RV_HIP(hipMemsetAsync(a->dErr.p, 0, 64, q));
    Level &lv = a->lv;
    lv.clear();
    int64_t off = 0;
    for (int s = 0; s < nsubs; s++) {
        const int64_t *m6 = meta + 6 * (size_t)s;
        const int64_t nf = node_first[s], nl = node_first[s + 1];
        if (m6[1] <= 0 || nf > nl) { rv_set_error("rv_frontier_import: bad sub-index %d", s); return -1; }
        lv.off.push_back(off); lv.n.push_back(m6[1]); lv.depth.push_back((int32_t)m6[2]); lv.nsamples.push_back((int32_t)m6[3]);
        lv.kind.push_back((int32_t)m6[4]); lv.parent.push_back(-1);
        for (int64_t k = nf; k < nl; k++) {
            if (nodes[2 * k] < 0 || nodes[2 * k + 1] > h->nT || nodes[2 * k] > nodes[2 * k + 1]) { rv_set_error("rv_frontier_import: interval outside the text"); return -1; }
            lv.nodes.push_back({nodes[2 * k], nodes[2 * k + 1]});
        }
        lv.node_first.push_back((int64_t)lv.nodes.size());
        off += m6[1];
    }
    if (off != m) { rv_set_error("rv_frontier_import: the sub-index sizes add up to %lld, not %lld", (long long)off, (long long)m); return -1; }
    lv.m = m;
    a->nx.clear();
    a->level = std::max(level, 1);
    a->cur = 0;
    RV_TRY(a->lvSA[0].reserve((size_t)(m + 64) * sizeof(sa_t)));
    RV_TRY(a->lvLCP[0].reserve((size_t)(m + 64) * sizeof(lcp_t)));
    RV_TRY(a->lvBWT[0].reserve((size_t)m + 64));
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (m) {
        RV_HIP(hipMemcpyAsync(a->lvSA[0].p, sa, (size_t)m * sizeof(sa_t), kind, q));
        RV_HIP(hipMemcpyAsync(a->lvLCP[0].p, lcp, (size_t)m * sizeof(lcp_t), kind, q));
        RV_HIP(hipMemcpyAsync(a->lvBWT[0].p, bwt, (size_t)m, kind, q));
    }
    // what the commit in front of a level ships for it
    prep_level_tables(h, lv, m);
    Packer &pk = a->pk;
    pk.clear();
    const size_t o_ss = pk.addv(a->next_ss), o_want = pk.addv(lv.nsamples);
    size_t o_tsub = 0, o_nodes = 0, o_flags = 0, o_tsub2 = 0;
    if (a->multi) o_tsub = pk.addv(a->next_tsub);
    if (a->next_dev_ok) { o_nodes = pk.addv(a->next_nodes); o_flags = pk.addv(a->next_flags); o_tsub2 = a->multi ? o_tsub : pk.addv(a->next_tsub); }
    RV_TRY(a->dTab0.reserve(pk.size() + 64));
    if (pk.pageable) RV_HIP(hipMemcpyAsync(a->dTab0.p, pk.data(), pk.size(), hipMemcpyHostToDevice, q));
    else { pk.grow(pk.size() + 16); RV_TRY(rv_h2d_copy(h->ws, pk.data(), a->dTab0.p, pk.size())); }
    RV_HIP(hipStreamSynchronize(q));
    const uint8_t *tb = a->dTab0.as<uint8_t>();
    a->d_next_ss = (const int64_t *)(tb + o_ss); a->d_next_want = (const int *)(tb + o_want); a->d_next_tsub = (const int *)(tb + o_tsub);
    a->d_next_nodes = (const sa_t *)(tb + o_nodes); a->d_next_flags = tb + o_flags; a->d_next_tsub2 = (const int *)(tb + o_tsub2);
    a->cur_dev_ok = a->next_dev_ok; a->early_done = false; a->early_bubble = false;
    a->scanned = false; a->d_err = nullptr;
    a->dec.reset(lv.size());
    a->skip_scan.assign((size_t)lv.size(), 0);
    // (seed lists are indexed by the sub-indices of the frontier they were made for: rv_frontier_seeds_import brings the new frontier's)
    a->seeds_cur.clear(); a->seeds_lead.clear(); a->seeds_trail.clear();
    return 0;
}


/* Replace the frontier of the handle by the given sub-indices (segments back to back in sa / lcp / bwt, m ranks in all).
 * A handle with a run in progress keeps its anchors and goes on with the new frontier (the share it keeps for itself);
 * any other handle only needs its samples (no construct): it becomes a worker that starts at this frontier. */
int rv_frontier_import(rv_index *h, int minl, int minn, uint32_t maxlcp, int level, int nsubs, const int64_t *meta,
                       const int64_t *node_first, const int64_t *nodes, int64_t m, const void *sa, const void *lcp, const void *bwt, int on_device) {
    RV_HIP(hipSetDevice(h->device));
    if (h->nsamples < 2) { rv_set_error("align needs at least two samples"); return -1; }
    if (nsubs < 0 || m < 0 || m >= ((int64_t)1 << 32)) { rv_set_error("rv_frontier_import: bad sizes"); return -1; }
    const bool fresh = !(h->al && h->al->running);
    if (fresh) {
        // (a handle that already is a worker -- a share taken from the queue before -- keeps its text; only the bound on the LCP values travels)
        if (h->text_only) h->maxlcp = maxlcp;
        else if (!h->constructed || h->main_arrays_freed) RV_TRY(rv_text_only(h, maxlcp));
        const bool keep_trace = h->al && h->al->trace_on;
        if (!h->al) h->al = new Align();
        Align *a = h->al;
        a->trace_on = keep_trace;
        a->minl = minl; a->minn = minn;
        a->multi = h->nsamples > 2;
        a->scanned = false; a->d_err = nullptr; a->flag_clean = false;
        a->next_dev_ok = a->cur_dev_ok = a->early_done = a->early_bubble = false;
        a->par_min_cur = (h->ws.opt.bubble_par_min >= 0 ? h->ws.opt.bubble_par_min : bubble_par_default(a->multi));
        RV_TRY(a->dErr.reserve(64));
        memset(&a->st, 0, sizeof a->st);
        a->full_only = !a->trace_on && a->picker == 0;      // (as builtin_setup: the chain picker wants every match of a sub-index)
        a->pick_filter = a->picker != 0 && a->multi && !a->trace_on;
        if (a->pick_filter) a->presel_on = true;
        if (a->picker == 2) { rv_set_error("rv_frontier_import: a run on a graph (rv_set_graph_picker) is not handed off -- its sub-indices share the graph"); return -1; }
        if (a->picker == 1) {
            if ((int)h->nodes.size() != h->nsamples) { rv_set_error("the native picker (rv_set_picker) takes one sequence per sample: %d sequences in %d samples", (int)h->nodes.size(), h->nsamples); return -1; }
            if (h->rc) { rv_set_error("the native picker (rv_set_picker) with construct(rc=1) is not supported"); return -1; }
        }
        a->picker_calls = a->picker_seeded = 0; a->picker_ns = a->picker_list_ns = a->galign_ns = 0;
        a->an_l.clear(); a->an_off.assign(1, 0); a->an_pos.clear(); a->trace.clear(); a->leaf_na = 0;
        RV_TRY(builtin_leaf_setup(h));
    }
    RV_TRY(install_frontier(h, level, nsubs, meta, node_first, nodes, m, sa, lcp, bwt, on_device));
    h->al->running = true;
    return 0;
}

/* rv_set_picker(1): the seed lists the parents' picker calls left for the listed sub-indices of the frontier (schemes.py:321-332; reveal.c:1157, 1180
 * hands them to the children as skipmums) as int64 words -- per sub-index: count, then per seed l, n, score, members, (sample, position) x members.
 * Returns the number of words; they are written when out holds that many (cap).  A frontier whose sub-indices carry no seeds: a zero per sub-index. */
int64_t rv_frontier_seeds_export(rv_index *h, const int32_t *subs, int k, int64_t *out, int64_t cap) {
    if (need_align(h)) return -1;
    const Align *a = h->al;
    const int ns = a->lv.size();
    int64_t need = 0;
    for (int pass = 0; pass < 2; pass++) {
        int64_t at = 0;
        for (int i = 0; i < k; i++) {
            const int s = subs[i];
            if (s < 0 || s >= ns) { rv_set_error("sub-index %d out of range", s); return -1; }
            const Align::SeedList *sd = (s < (int)a->seeds_cur.size() && a->seeds_cur[(size_t)s].size() > 0) ? &a->seeds_cur[(size_t)s] : nullptr;
            if (pass) out[at] = sd ? (int64_t)sd->size() : 0;
            at++;
            if (!sd) continue;
            for (size_t r = 0; r < sd->size(); r++) {
                const int64_t mem = sd->off[r + 1] - sd->off[r];
                if (pass) {
                    out[at] = sd->l[r]; out[at + 1] = sd->n[r]; out[at + 2] = sd->score[r]; out[at + 3] = mem;
                    for (int64_t q2 = 0; q2 < mem; q2++) { out[at + 4 + 2 * q2] = sd->so[(size_t)(sd->off[r] + q2)]; out[at + 5 + 2 * q2] = sd->pos[(size_t)(sd->off[r] + q2)]; }
                }
                at += 4 + 2 * mem;
            }
        }
        need = at;
        if (!out || cap < need) break;
    }
    return need;
}

/* ... and their way into the handle that took those sub-indices over (rv_frontier_import first, same sub-indices in the same order): a seeded
 * sub-index is not scanned, its picker call takes the middle of its list (reveal.c:802, 830-837; schemes.py:346-351) */
int rv_frontier_seeds_import(rv_index *h, int nsubs, const int64_t *words, int64_t nwords) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    if (!a->running || nsubs != a->lv.size()) { rv_set_error("rv_frontier_seeds_import: %d seed lists for a frontier of %d sub-indices", nsubs, a->lv.size()); return -1; }
    if (a->picker != 1) { rv_set_error("rv_frontier_seeds_import: seeds belong to the native picker (rv_set_picker)"); return -1; }
    std::vector<Align::SeedList> nxt((size_t)nsubs);
    int64_t at = 0;
    for (int s = 0; s < nsubs; s++) {
        if (at >= nwords) { rv_set_error("rv_frontier_seeds_import: truncated seed lists"); return -1; }
        const int64_t cnt = words[at++];
        if (cnt < 0) { rv_set_error("rv_frontier_seeds_import: bad seed count"); return -1; }
        for (int64_t r = 0; r < cnt; r++) {
            if (at + 4 > nwords) { rv_set_error("rv_frontier_seeds_import: truncated seed lists"); return -1; }
            const int64_t l = words[at], n = words[at + 1], sc = words[at + 2], mem = words[at + 3];
            if (mem < 0 || mem > h->nsamples || at + 4 + 2 * mem > nwords) { rv_set_error("rv_frontier_seeds_import: bad seed record"); return -1; }
            std::vector<uint16_t> so((size_t)mem); std::vector<int64_t> pos((size_t)mem);
            for (int64_t q2 = 0; q2 < mem; q2++) {
                so[(size_t)q2] = (uint16_t)words[at + 4 + 2 * q2]; pos[(size_t)q2] = words[at + 5 + 2 * q2];
                if (pos[(size_t)q2] < 0 || pos[(size_t)q2] + l > h->nT) { rv_set_error("rv_frontier_seeds_import: seed outside the text"); return -1; }
            }
            nxt[(size_t)s].push((u32)l, (int32_t)n, so.data(), pos.data(), (int)mem, sc);
            at += 4 + 2 * mem;
        }
        if (cnt > 0) {
            if ((int)a->skip_scan.size() != nsubs) a->skip_scan.assign((size_t)nsubs, 0);
            a->skip_scan[(size_t)s] = 1;
        }
    }
    a->seeds_cur.swap(nxt);
    return 0;
}

/* ---- host-driven single steps: splitindex / extract / copy (reveal.c:1386-1748, interface.c:432-470) ----------------
 * A detached (sub)index owns its SA / LCP / BWT in HBM and borrows text, shared inverse and separators from its main
 * handle, like the reference's child objects (reveal.c:1679-1735).  Every step loads it as a one-sub-index frontier and
 * runs the same label / split / lower-casing / bubble kernels as a level of align(). */
}  // extern "C"

struct rv_subindex {
    rv_index *h = nullptr;
    int64_t n = 0;
    int32_t depth = 0, nsamples = 0;
    std::vector<RvIntv> nodes;     // the interval list this index was made from (reveal.h:36)
    std::vector<RvIntv> cover;     // text positions whose suffixes are ranks of this index: sorted, disjoint
    DBuf sa, lcp, bwt;
    ~rv_subindex() { sa.release(); lcp.release(); bwt.release(); }
};

static int sx_alloc(rv_subindex *x, int64_t n) {
    RV_TRY(x->sa.reserve((size_t)(n + 64) * sizeof(sa_t)));
    RV_TRY(x->lcp.reserve((size_t)(n + 64) * sizeof(lcp_t)));
    RV_TRY(x->bwt.reserve((size_t)n + 64));
    x->n = n;
    return 0;
}

// x becomes the frontier of its handle in host-driven mode (what rv_align_begin sets up, one level down)
static int sx_load(rv_subindex *x, int minl, int minn) {
    rv_index *h = x->h;
    RV_HIP(hipSetDevice(h->device));
    if (!h->constructed && !h->text_only) { rv_set_error("Index not yet constructed."); return -1; }
    if (h->al && h->al->running) { rv_set_error("a built-in alignment is in progress on this index"); return -1; }
    if (h->nsamples < 2) { rv_set_error("at least two samples are needed"); return -1; }
    if (x->n <= 0 || x->n >= ((int64_t)1 << 32)) { rv_set_error("sub-index of %lld ranks not supported", (long long)x->n); return -1; }
    const bool keep_trace = h->al && h->al->trace_on;
    if (!h->al) h->al = new Align();
    Align *a = h->al;
    a->trace_on = keep_trace;
    a->minl = minl; a->minn = minn;
    a->multi = h->nsamples > 2;
    a->scanned = false; a->d_err = nullptr; a->flag_clean = false; a->full_only = false; a->use_leaf = false; a->leaf_launch_due = false;
    a->next_dev_ok = a->cur_dev_ok = a->early_done = a->early_bubble = false;
    a->presel_on = false;        // getmums / getmultimums of a detached index hand out every match
    a->par_min_cur = (h->ws.opt.bubble_par_min >= 0 ? h->ws.opt.bubble_par_min : bubble_par_default(a->multi));
    RV_TRY(a->dErr.reserve(64));
    memset(&a->st, 0, sizeof a->st);
    const int64_t meta[6] = {0, x->n, x->depth, x->nsamples, 0, -1};
    const int64_t nf[2] = {0, (int64_t)x->nodes.size()};
    static_assert(sizeof(RvIntv) == 2 * sizeof(int64_t), "RvIntv layout");
    RV_TRY(install_frontier(h, std::max((int)x->depth, 1), 1, meta, nf, (const int64_t *)x->nodes.data(), x->n, x->sa.p, x->lcp.p, x->bwt.p, 1));
    return 0;
}

// the error word of the last commit (pair mode defers it to the next scan; a single step has none)
static int sx_commit_check(rv_index *h) {
    u32 err = 0;
    RV_HIP(hipMemcpyAsync(&err, h->al->dErr.p, 4, hipMemcpyDeviceToHost, h->ws.stream));
    RV_HIP(hipStreamSynchronize(h->ws.stream));
    h->al->d_err = nullptr;
    if (err & 1u) { rv_set_error("split: the intervals do not partition the index (child size mismatch)"); return -1; }
    if (err) { rv_set_error("split / bubble_sort reported error bits %u", err); return -1; }
    return 0;
}

// sorted copy; fails on intervals outside cover or overlapping each other
static int sx_check_inside(const rv_subindex *x, const RvIntv *iv, int n, const char *what, std::vector<RvIntv> &sorted) {
    sorted.assign(iv, iv + n);
    std::sort(sorted.begin(), sorted.end(), intv_less);
    size_t c = 0;
    for (size_t k = 0; k < sorted.size(); k++) {
        const RvIntv &v = sorted[k];
        if (v.begin >= v.end) { rv_set_error("%s: empty interval [%lld,%lld)", what, (long long)v.begin, (long long)v.end); return -1; }
        if (k && v.begin < sorted[k - 1].end) { rv_set_error("%s: overlapping intervals", what); return -1; }
        while (c < x->cover.size() && x->cover[c].end <= v.begin) c++;
        if (c >= x->cover.size() || v.begin < x->cover[c].begin || v.end > x->cover[c].end) {
            rv_set_error("%s: interval [%lld,%lld) is not part of this index", what, (long long)v.begin, (long long)v.end);
            return -1;
        }
    }
    return 0;
}

// bubble_sort visits the matching intervals in the order given (reveal.c:673-674).  The kernels look for crossing suffixes
// only in the window in front of each cut, which is the same thing unless a LATER interval of the list begins less than
// maxlcp in front of an earlier one with no separator in between (a suffix could then cross both cuts and the reference
// would move it twice).  A match never does that (its members lie in different sequences); refuse the rest.
static int sx_check_cut_order(const rv_index *h, const RvIntv *match, int nmatch) {
    for (int i = 0; i < nmatch; i++)
        for (int j = i + 1; j < nmatch; j++) {
            if (match[j].begin >= match[i].begin) continue;
            if (match[i].begin - match[j].begin >= (int64_t)h->maxlcp) continue;
            bool sep = false;
            for (int64_t p = match[j].begin; p < match[i].begin && !sep; p++) sep = h->T[(size_t)p] == '$' || h->T[(size_t)p] == 'N';
            if (!sep) { rv_set_error("matching intervals closer than the longest repeat must be listed in ascending order"); return -1; }
        }
    return 0;
}

static rv_subindex *sx_take_child(rv_index *h, int slot, int depth, const RvIntv *nodes, size_t nnodes) {
    Align *a = h->al;
    rv_subindex *c = new rv_subindex();
    c->h = h; c->depth = depth; c->nsamples = a->lv.nsamples[(size_t)slot];
    c->nodes.assign(nodes, nodes + nnodes);
    c->cover = c->nodes;
    std::sort(c->cover.begin(), c->cover.end(), intv_less);
    const int64_t off = a->lv.off[(size_t)slot], n = a->lv.n[(size_t)slot];
    hipStream_t q = h->ws.stream;
    bool ok = sx_alloc(c, n) == 0;
    ok = ok && hipMemcpyAsync(c->sa.p, cur_sa(h) + off, (size_t)n * sizeof(sa_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->lcp.p, cur_lcp(h) + off, (size_t)n * sizeof(lcp_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->bwt.p, cur_bwt(h) + off, (size_t)n, hipMemcpyDeviceToDevice, q) == hipSuccess;
    if (!ok) { rv_set_error("copy of a child index failed"); delete c; return nullptr; }
    return c;
}

extern "C" {

/* the main index as a detached copy (its SA / LCP stay untouched by the steps below: the reference's splitindex leaves
 * the parent's arrays alone too, reveal.c:1738) */
rv_subindex *rv_sx_main(rv_index *h) {
    if (!h->constructed || h->main_arrays_freed) { rv_set_error("Index not yet constructed."); return nullptr; }
    if (hipSetDevice(h->device) != hipSuccess) { rv_set_error("hipSetDevice failed"); return nullptr; }
    if (rv_need_sai(h)) return nullptr;
    rv_subindex *x = new rv_subindex();
    x->h = h; x->depth = 0; x->nsamples = h->nsamples;
    x->nodes = h->nodes;
    std::sort(x->nodes.begin(), x->nodes.end(), intv_less);
    x->cover.assign(1, RvIntv{0, h->nT});
    hipStream_t q = h->ws.stream;
    bool ok = sx_alloc(x, h->n) == 0;
    ok = ok && hipMemcpyAsync(x->sa.p, h->dSA.p, (size_t)h->n * sizeof(sa_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(x->lcp.p, h->dLCP.p, (size_t)h->n * sizeof(lcp_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(x->bwt.p, h->dBWT.p, (size_t)h->n, hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipStreamSynchronize(q) == hipSuccess;
    if (!ok) { rv_set_error("copy of the main index failed"); delete x; return nullptr; }
    return x;
}

rv_subindex *rv_sx_copy(rv_subindex *x) {
    rv_index *h = x->h;
    if (hipSetDevice(h->device) != hipSuccess) { rv_set_error("hipSetDevice failed"); return nullptr; }
    rv_subindex *c = new rv_subindex();
    c->h = h; c->depth = x->depth; c->nsamples = x->nsamples; c->nodes = x->nodes; c->cover = x->cover;
    hipStream_t q = h->ws.stream;
    bool ok = sx_alloc(c, x->n) == 0;
    ok = ok && hipMemcpyAsync(c->sa.p, x->sa.p, (size_t)x->n * sizeof(sa_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->lcp.p, x->lcp.p, (size_t)x->n * sizeof(lcp_t), hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->bwt.p, x->bwt.p, (size_t)x->n, hipMemcpyDeviceToDevice, q) == hipSuccess;
    ok = ok && hipStreamSynchronize(q) == hipSuccess;
    if (!ok) { rv_set_error("copy of a sub-index failed"); delete c; return nullptr; }
    return c;
}

void rv_sx_free(rv_subindex *x) {
    if (!x) return;
    (void)hipSetDevice(x->h->device);
    (void)hipStreamSynchronize(x->h->ws.stream);
    delete x;
}

int rv_sx_info(const rv_subindex *x, rv_sub *out) {
    memset(out, 0, sizeof *out);
    out->n = x->n; out->depth = x->depth; out->nsamples = x->nsamples; out->nnodes = (int32_t)x->nodes.size(); out->parent = -1;
    return 0;
}

int rv_sx_nodes(const rv_subindex *x, int64_t *be) {
    for (size_t k = 0; k < x->nodes.size(); k++) { be[2 * k] = x->nodes[k].begin; be[2 * k + 1] = x->nodes[k].end; }
    return 0;
}

int64_t rv_sx_array(rv_subindex *x, int which, void *out, int64_t cap) {
    (void)hipSetDevice(x->h->device);
    if (which != RV_SA && which != RV_LCP) { rv_set_error("rv_sx_array: bad array id"); return -1; }
    if (cap < x->n) { rv_set_error("buffer too small"); return -1; }
    (void)hipStreamSynchronize(x->h->ws.stream);
    const hipError_t e = which == RV_SA ? hipMemcpy(out, x->sa.p, (size_t)x->n * sizeof(sa_t), hipMemcpyDeviceToHost)
                                        : hipMemcpy(out, x->lcp.p, (size_t)x->n * sizeof(lcp_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { rv_set_error("D2H failed"); return -1; }
    return x->n;
}

/* getmums / getmultimums (reveal.c:55-116, :436-580) over this index: count (and member count); rv_sx_fetch hands them out
 * in the CSR form of rv_sub_mums */
int64_t rv_sx_scan(rv_subindex *x, int minl, int minn, int64_t *members) {
    rv_index *h = x->h;
    if (sx_load(x, minl, minn)) return -1;
    if (rv_frontier_scan(h)) return -1;
    rv_sub info;
    if (rv_sub_info(h, 0, &info)) return -1;
    if (members) *members = info.nmembers;
    return info.nmums;
}

int rv_sx_fetch(rv_subindex *x, uint32_t *l, int32_t *n, int64_t *off, uint16_t *so, int64_t *pos) {
    return rv_sub_mums(x->h, 0, l, n, off, so, pos);
}

/* splitindex (reveal.c:1515-1748): label by the four interval lists, split, lower-case the matching intervals, bubble_sort
 * the leading child over them in the order given.  out[0..2] = leading, trailing, parallel child or NULL (reveal.c:1744). */
int rv_sx_split(rv_subindex *x, const int64_t *lead, int nlead, const int64_t *trail, int ntrail,
                const int64_t *match, int nmatch, const int64_t *rest, int nrest, rv_subindex **out) {
    rv_index *h = x->h;
    out[0] = out[1] = out[2] = nullptr;
    std::vector<RvIntv> all, tmp;
    const int64_t *lists[4] = {lead, trail, match, rest};
    const int cnts[4] = {nlead, ntrail, nmatch, nrest};
    for (int q = 0; q < 4; q++) all.insert(all.end(), (const RvIntv *)lists[q], (const RvIntv *)lists[q] + cnts[q]);
    RV_TRY(sx_check_inside(x, all.data(), (int)all.size(), "splitindex", tmp));
    RV_TRY(sx_check_cut_order(h, (const RvIntv *)match, nmatch));
    RV_TRY(sx_load(x, 0, 0));
    RV_TRY(add_decision(h, 0, 0, nullptr, 0, (const RvIntv *)lead, nlead, (const RvIntv *)trail, ntrail, (const RvIntv *)match, nmatch,
                        (const RvIntv *)rest, nrest, true));
    int32_t kids[3] = {-1, -1, -1};
    RV_TRY(rv_frontier_commit(h, kids));
    RV_TRY(sx_commit_check(h));
    const int cls_list[3] = {0, 1, 3};
    for (int c = 0; c < 3; c++) {
        if (kids[c] < 0) continue;
        out[c] = sx_take_child(h, kids[c], x->depth + 1, (const RvIntv *)lists[cls_list[c]], (size_t)cnts[cls_list[c]]);
        if (!out[c]) { for (int k = 0; k < 3; k++) { delete out[k]; out[k] = nullptr; } return -1; }
    }
    RV_HIP(hipStreamSynchronize(h->ws.stream));
    (void)rv_align_end(h);
    return 0;
}

/* extract (reveal.c:1386-1505): the suffixes of the given intervals leave the index, the intervals are lower-cased, the rest
 * is bubble_sorted over them.  With rc=1 query-side intervals arrive in reverse-complement coordinates and are mapped back
 * (written to `intervals`, as the reference replaces the list items, reveal.c:1411-1427).  The reference never writes the
 * new SA[0] (it stays rank 0 here) and overruns its buffers when rank 0 itself is matched: that case is refused. */
int rv_sx_extract(rv_subindex *x, int64_t *intervals, int niv) {
    rv_index *h = x->h;
    RV_HIP(hipSetDevice(h->device));
    RvIntv *iv = (RvIntv *)intervals;
    if (h->rc == 1)
        for (int k = 0; k < niv; k++)
            if (iv[k].begin > h->nsep[0]) {
                const int64_t b = h->nsep[0] + (h->nT - iv[k].begin - (iv[k].end - iv[k].begin)), e = h->nsep[0] + (h->nT - iv[k].begin);
                iv[k].begin = b; iv[k].end = e;
            }
    std::vector<RvIntv> sorted, lead;
    RV_TRY(sx_check_inside(x, iv, niv, "extract", sorted));
    RV_TRY(sx_check_cut_order(h, iv, niv));
    {   // rank 0
        sa_t first = 0;
        RV_HIP(hipStreamSynchronize(h->ws.stream));
        RV_HIP(hipMemcpy(&first, x->sa.p, sizeof(sa_t), hipMemcpyDeviceToHost));
        for (const RvIntv &v : sorted)
            if ((int64_t)first >= v.begin && (int64_t)first < v.end) { rv_set_error("extract: the first suffix of the index is matched (the reference overruns its buffers there)"); return -1; }
    }
    size_t m = 0;
    int64_t left = 0;
    for (const RvIntv &c : x->cover) {       // what stays = cover minus the intervals
        int64_t at = c.begin;
        while (m < sorted.size() && sorted[m].begin < c.end) {
            if (sorted[m].begin > at) lead.push_back({at, sorted[m].begin});
            at = sorted[m].end;
            m++;
        }
        if (at < c.end) lead.push_back({at, c.end});
    }
    for (const RvIntv &v : lead) left += v.end - v.begin;
    if (niv == 0) return 0;
    RV_TRY(sx_load(x, 0, 0));
    RV_TRY(add_decision(h, 0, 0, nullptr, 0, lead.data(), (int)lead.size(), nullptr, 0, iv, niv, nullptr, 0, true));
    int32_t kids[3] = {-1, -1, -1};
    RV_TRY(rv_frontier_commit(h, kids));
    RV_TRY(sx_commit_check(h));
    if (kids[0] < 0 || h->al->lv.n[(size_t)kids[0]] != left) { rv_set_error("extract: internal size mismatch"); return -1; }
    const int64_t off = h->al->lv.off[(size_t)kids[0]];
    hipStream_t q = h->ws.stream;
    RV_HIP(hipMemcpyAsync(x->sa.p, cur_sa(h) + off, (size_t)left * sizeof(sa_t), hipMemcpyDeviceToDevice, q));
    RV_HIP(hipMemcpyAsync(x->lcp.p, cur_lcp(h) + off, (size_t)left * sizeof(lcp_t), hipMemcpyDeviceToDevice, q));
    RV_HIP(hipMemcpyAsync(x->bwt.p, cur_bwt(h) + off, (size_t)left, hipMemcpyDeviceToDevice, q));
    RV_HIP(hipStreamSynchronize(q));
    x->n = left;
    x->cover.swap(lead);
    (void)rv_align_end(h);
    return 0;
}

uint32_t rv_maxlcp(const rv_index *h) { return h->maxlcp; }

int64_t rv_anchor_count(rv_index *h, int64_t *members) {
    if (need_align(h)) return -1;
    if (members) *members = (int64_t)(h->al->an_pos.size() + 2 * h->al->leaf_na);
    return (int64_t)(h->al->an_l.size() + h->al->leaf_na);
}
// a large copy into caller memory (fresh pages on the caller's side: first touch is most of its cost) on a few host threads
static void copy_parallel(void *dst, const void *src, size_t bytes) {
    const size_t chunk = (size_t)8 << 20;
    if (bytes < 2 * chunk) { memcpy(dst, src, bytes); return; }
    const int nt = (int)std::min<size_t>(8, bytes / chunk);
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) {
        const size_t lo = bytes / nt * t, hi = t + 1 == nt ? bytes : bytes / nt * (t + 1);
        th.emplace_back([=]() { memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    memcpy(dst, src, bytes / nt);
    for (auto &x : th) x.join();
}

int rv_set_result_buffers(rv_index *h, uint32_t *l, int64_t l_cap, int64_t *off, int64_t off_cap, int64_t *pos, int64_t pos_cap) {
    if (!h) { rv_set_error("rv_set_result_buffers: null handle"); return -1; }
    rv_index::ResultBufs &rb = h->rb;
    if (l == rb.l && off == rb.off && pos == rb.pos && l_cap == rb.l_cap && off_cap == rb.off_cap && pos_cap == rb.pos_cap) return 0;
    RV_HIP(hipSetDevice(h->device));
    if (rb.l) {      // (a copy into them may still be in flight only inside rv_align_builtin: not here)
        if (rb.direct && h->al && h->al->leaf_na) {
            // the last run delivered its leaf / cascade anchors into these arrays and they have not been fetched yet: they move to the
            // staging buffer (the layout rv_fetch_anchors reads: pos[2 na], l[na]) before the arrays stop being the library's
            Align *a = h->al;
            const size_t na = a->leaf_na, nl = a->an_l.size(), np = a->an_pos.size();
            RV_TRY(a->hLeafOut.reserve(na * 20 + 64));
            int64_t *pp = a->hLeafOut.as<int64_t>();
            memcpy(pp, rb.pos + np, na * 16);
            memcpy(pp + 2 * na, rb.l + nl, na * 4);
        }
        (void)hipHostUnregister(rb.l); (void)hipHostUnregister(rb.off); (void)hipHostUnregister(rb.pos);
        rb = rv_index::ResultBufs();
    }
    if (!l || !off || !pos || l_cap <= 0 || off_cap <= 0 || pos_cap <= 0) return 0;
    // (whole pages of the caller's only: locking / unlocking act on pages, and a page shared with other heap objects loses its GPU mapping
    //  under them -- numpy arrays from the C heap, locked here, ended in GPU memory faults of later copies into their neighbours)
    if ((((uintptr_t)l | (uintptr_t)off | (uintptr_t)pos) & 4095u) != 0 && !h->ws.opt.lock_any) return 0;      // (RV_LOCK_ANY: diagnostics)
    if (hipHostRegister(l, (size_t)l_cap * 4, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return 0; }      // (not page-lockable: the staging buffer as before)
    if (hipHostRegister(off, (size_t)off_cap * 8, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); (void)hipHostUnregister(l); return 0; }
    if (hipHostRegister(pos, (size_t)pos_cap * 8, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); (void)hipHostUnregister(l); (void)hipHostUnregister(off); return 0; }
    rb.l = l; rb.off = off; rb.pos = pos; rb.l_cap = l_cap; rb.off_cap = off_cap; rb.pos_cap = pos_cap;
    return 0;
}

int rv_fetch_anchors(rv_index *h, uint32_t *l, int64_t *off, int64_t *pos) {
    RV_TRY(need_align(h));
    Align *a = h->al;
    const size_t nl = a->an_l.size(), np = a->an_pos.size(), na = a->leaf_na;
    memcpy(l, a->an_l.data(), nl * 4);
    memcpy(off, a->an_off.data(), a->an_off.size() * 8);
    memcpy(pos, a->an_pos.data(), np * 8);
    if (na && h->rb.direct && l == h->rb.l && off == h->rb.off && pos == h->rb.pos) return 0;      // delivered by the run itself (rv_set_result_buffers)
    if (na) {      // the leaf launches' (and the cascade's) anchors, still in the pinned staging buffer (pos[2 na], l[na]) -- or in the result buffers
        const int64_t *pp = h->rb.direct ? h->rb.pos + np : a->hLeafOut.as<int64_t>();
        const u32 *pl = h->rb.direct ? h->rb.l + nl : (const u32 *)(pp + 2 * na);
        std::thread t_off;
        if (na > (size_t)1 << 18) t_off = std::thread([=]() { for (size_t k = 0; k < na; k++) off[nl + 1 + k] = (int64_t)(np + 2 * (k + 1)); });
        else for (size_t k = 0; k < na; k++) off[nl + 1 + k] = (int64_t)(np + 2 * (k + 1));
        memcpy(l + nl, pl, na * 4);
        copy_parallel(pos + np, pp, na * 16);
        if (t_off.joinable()) t_off.join();
    }
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

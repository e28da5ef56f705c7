// rv_graph.hip -- the alignment graph of a finished `reveal rem` run, built from its anchors on the host side of the ABI (no device code).
//
// With the picker inside the library (rv_set_picker) the recursion returns the anchors in the order it chose them, and the graph the reference
// builds callback by callback (reveal/rem.py:318-345 graphalign: break the nodes that hold the members, merge the pieces; rem.py:14-131 breaknode,
// 133-200 mergenodes) depends on that order alone.  reveal_amd/rem.py replay_anchors_fast does the surgery in Python: 48 us per broken node, 25 s for
// five genomes of 5 Mbp -- most of what such a job takes.  This file does the same surgery on plain arrays and hands the result back in the
// order the Python structure would have it: the writer numbers the nodes in creation order and prints a node's links in the order they were (last)
// made, prune_nodes picks the survivor of a merge by the order of its neighbours, so "ordered dictionary" is part of the result.
//
// Scope: the native picker's -- FASTA inputs, one sequence per sample (path id = sample), every edge on the forward strand.
// Test infrastructure compares it with the Python surgery node for node and edge for edge (tests/test_cpu_graph_native.py).
#include "rv_graph.h"
#include <cstdlib>
#include "rv_graphrem.h"
#include <atomic>
#include <thread>

void rv_graph_align_out_free(void *p);      // rv_graphrem.hip
rv_graph::~rv_graph() { if (align_out_) rv_graph_align_out_free(align_out_); }

extern "C" {

// (no exception may leave through the C ABI: bad_alloc from the vectors' growth becomes NULL / -1 and rv_last_error's text)
static void replay_start(rv_graph *g, int nseq, const int64_t *begin, const int64_t *end) {
    g->nseq = nseq;
    g->made_on = true;      // (a member's node through the bitmap of begins and the hash, rv_graph::fast_node_at: new_node keeps both up)
    g->use_map = false;     // (... and the position map is not kept while the anchors are applied: rv_graph.h use_map)
    // the FASTA reader's graph (utils.py:304-375): start sentinel, the sequence, end sentinel -- per sequence, in this order
    for (int s = 0; s < nseq; s++) {
        const int st = g->new_node(s, 0, -1), iv = g->new_node(begin[s], end[s], 0), en = g->new_node(s, 1, -1);
        g->start_of.push_back(st);
        g->nodes[(size_t)st].sent = 1; g->nodes[(size_t)en].sent = 2;
        g->nodes[(size_t)st].off.push_back({s, 0}); g->nodes[(size_t)iv].off.push_back({s, 0}); g->nodes[(size_t)en].off.push_back({s, end[s] - begin[s]});
        PathSet only; only.add(s);
        g->add_edge(st, iv, only); g->add_edge(iv, en, only);
    }
}
// anchors [0, na): member k of anchor a at an_pos[an_off[a] - an_off[0] + k]
static bool replay_apply(rv_graph *g, int64_t na, const uint32_t *an_l, const int64_t *an_off, const int64_t *an_pos, std::vector<int> &mns) {
    const int64_t base = na ? an_off[0] : 0;
    for (int64_t a = 0; a < na; a++) {
        mns.clear();
        const int64_t l = (int64_t)an_l[a];
        for (int64_t k = an_off[a] - base; k < an_off[a + 1] - base; k++) {
            const int x = g->fast_node_at(an_pos[k]);
            if (x < 0 || an_pos[k] + l > g->nodes[(size_t)x].e) { g->err = "rv_graph_replay: an anchor's member lies in no node of the graph"; return false; }
            mns.push_back(g->breaknode(x, an_pos[k], l));
        }
        if (!mns.empty()) g->mergenodes(mns);
    }
    return true;
}
static rv_graph *graph_replay(int nseq, const int64_t *begin, const int64_t *end, int64_t na, const uint32_t *an_l, const int64_t *an_off, const int64_t *an_pos) {
    std::unique_ptr<rv_graph> own(new rv_graph());      // (freed when the surgery below throws)
    rv_graph *g = own.get();
    replay_start(g, nseq, begin, end);
    std::vector<int> mns;
    if (!replay_apply(g, na, an_l, an_off, an_pos, mns)) { g->finish(); return own.release(); }
    g->compact();
    g->finish();
    return own.release();
}
/* the graph of the inputs alone (rv_graph_replay with no anchors, not renumbered): what rv_set_replay_graph feeds while a run goes on */
rv_graph *rv_graph_replay_begin(int nseq, const int64_t *begin, const int64_t *end) {
    try { std::unique_ptr<rv_graph> own(new rv_graph()); replay_start(own.get(), nseq, begin, end); return own.release(); }
    catch (...) { rv_set_error("rv_graph_replay_begin: out of host memory"); return nullptr; }
}
rv_graph *rv_graph_replay(int nseq, const int64_t *begin, const int64_t *end, int64_t na, const uint32_t *an_l, const int64_t *an_off, const int64_t *an_pos) {
    try { return graph_replay(nseq, begin, end, na, an_l, an_off, an_pos); }
    catch (const std::exception &e) { rv_set_error("rv_graph_replay: %s", e.what()); return nullptr; }
    catch (...) { rv_set_error("rv_graph_replay: failed"); return nullptr; }
}

/* after changes to the graph (rv_graph_align) and before rv_graph_sizes / rv_graph_export: the numbering of live nodes and links in dictionary order */
int rv_graph_finish(rv_graph *g) {
    if (getenv("RV_GRAPH_TIMES") && g->t_phase[0] + g->t_phase[2] > 0)
        fprintf(stderr, "graphalign: look-ups %.3f s, breaks + merge %.3f, walks %.3f, lists %.3f, sorts %.3f\n", g->t_phase[0], g->t_phase[1], g->t_phase[2], g->t_phase[3], g->t_phase[4]);
    try { g->finish(); return 0; } catch (...) { rv_set_error("rv_graph_finish: out of host memory"); return -1; }
}
const char *rv_graph_error(const rv_graph *g) { return g->err.empty() ? nullptr : g->err.c_str(); }

/* out[0] nodes, out[1] offset entries, out[2] edges, out[3] path entries of the edges */
int rv_graph_sizes(const rv_graph *g, int64_t *out) {
    int64_t no = 0, ne = 0, np = 0;
    for (int x : g->order) {
        no += (int64_t)g->nodes[(size_t)x].off.size();
        for (int e : g->nodes[(size_t)x].succ) { ne++; np += (int64_t)g->edges[(size_t)e].paths.size(); }
    }
    out[0] = (int64_t)g->order.size(); out[1] = no; out[2] = ne; out[3] = np;
    return 0;
}

/* nodes in dictionary order: (b, e, aligned; -1 = sentinel: b = sample, e = 0 start / 1 end); offsets as CSR; per node its links forwards and backwards as CSR of
 * (neighbour's number, edge number) in dictionary order; per edge its path ids as CSR */
static int graph_export(const rv_graph *g, int64_t *node_b, int64_t *node_e, int8_t *node_aligned, int64_t *off_ptr, int32_t *off_sid, int64_t *off_val,
                    int64_t *succ_ptr, int32_t *succ_to, int32_t *succ_edge, int64_t *pred_ptr, int32_t *pred_from, int32_t *pred_edge, int64_t *edge_ptr, int32_t *edge_paths) {
    std::vector<int> number(g->nodes.size(), -1);
    for (size_t i = 0; i < g->order.size(); i++) number[(size_t)g->order[i]] = (int)i;
    int64_t no = 0, ns = 0, np = 0;
    int64_t nedges = 0;
    for (int e : g->edge_no) if (e >= 0) nedges++;
    std::vector<int> by_no((size_t)nedges, -1);
    for (size_t e = 0; e < g->edge_no.size(); e++) if (g->edge_no[e] >= 0) by_no[(size_t)g->edge_no[e]] = (int)e;
    for (size_t i = 0; i < g->order.size(); i++) {
        const GNode &n = g->nodes[(size_t)g->order[i]];
        node_b[i] = n.b; node_e[i] = n.e; node_aligned[i] = n.aligned;
        off_ptr[i] = no;
        for (auto &a : n.off) { off_sid[no] = a.first; off_val[no] = a.second; no++; }
        succ_ptr[i] = ns;
        for (int e : n.succ) { succ_to[ns] = number[(size_t)g->edges[(size_t)e].v]; succ_edge[ns] = g->edge_no[(size_t)e]; ns++; }
        pred_ptr[i] = np;
        for (int e : n.pred) { pred_from[np] = number[(size_t)g->edges[(size_t)e].u]; pred_edge[np] = g->edge_no[(size_t)e]; np++; }
    }
    off_ptr[g->order.size()] = no; succ_ptr[g->order.size()] = ns; pred_ptr[g->order.size()] = np;
    int64_t pp = 0;
    for (int64_t k = 0; k < nedges; k++) {
        edge_ptr[k] = pp;
        g->edges[(size_t)by_no[(size_t)k]].paths.each([&](int p) { edge_paths[pp++] = p; });
    }
    edge_ptr[nedges] = pp;
    return 0;
}

/* rem.py:384-447 prune_nodes, as reveal_amd/alngraph.py runs it (work list instead of whole passes, same fixpoint): siblings with the same sequence
 * that hang on one parent (child) and have no other are merged.  T = the index text after the run (matched text lower-cased: an aligned and an
 * unaligned sibling spell differently).  Which sibling survives, and where the survivor stands among its neighbours' links, follows the dictionary
 * order -- kept here. */
static int graph_prune(rv_graph *g, const char *T) {
    auto &nodes = g->nodes;
    for (;;) {
        bool merged_any = false;
        g->finish();
        std::deque<int> queue(g->order.begin(), g->order.end());
        std::vector<char> queued(nodes.size(), 0);
        for (int x : queue) queued[(size_t)x] = 1;
        std::vector<int> neis, again, rep, grp;
        while (!queue.empty()) {
            const int node = queue.front(); queue.pop_front(); queued[(size_t)node] = 0;
            if (!nodes[(size_t)node].alive) continue;
            for (int dir = 0; dir < 2; dir++) {
                const LinkVec &adj = dir == 0 ? nodes[(size_t)node].succ : nodes[(size_t)node].pred;
                if (adj.size() < 2) continue;
                neis.clear();
                for (size_t k = 0; k < adj.size(); k++) neis.push_back(adj.links()[k].to);
                // groups of neighbours that spell the same, in order of their first member; a group's members in the neighbours' order.  (rep[k] = the place of the
                // first neighbour that spells like neighbour k: no list of lists made and thrown away per node -- most nodes have two neighbours that differ)
                const size_t nk = neis.size();
                rep.assign(nk, -1);
                bool any_group = false;
                for (size_t k = 0; k < nk; k++) {
                    const GNode &q = nodes[(size_t)neis[k]];
                    if (q.aligned < 0) continue;
                    rep[k] = (int)k;
                    for (size_t j = 0; j < k; j++) {
                        if (rep[j] != (int)j) continue;
                        const GNode &r = nodes[(size_t)neis[j]];
                        if (r.e - r.b == q.e - q.b && (q.e == q.b || T[r.b] == T[q.b]) && memcmp(T + r.b, T + q.b, (size_t)(q.e - q.b)) == 0) { rep[k] = (int)j; any_group = true; break; }
                    }
                }
                if (!any_group) continue;
                for (size_t k0 = 0; k0 < nk; k0++) {
                    if (rep[k0] != (int)k0) continue;
                    grp.clear();
                    for (size_t k = k0; k < nk; k++) if (rep[k] == (int)k0) grp.push_back(neis[k]);
                    if (grp.size() < 2) continue;
                    bool single = true;
                    for (int v : grp) if ((dir == 0 ? nodes[(size_t)v].pred : nodes[(size_t)v].succ).size() > 1) { single = false; break; }
                    if (!single) continue;
                    const int ref = g->mergenodes(grp);
                    merged_any = true;
                    again.clear(); again.push_back(node); again.push_back(ref);
                    for (size_t k = 0; k < nodes[(size_t)ref].succ.size(); k++) again.push_back(nodes[(size_t)ref].succ.links()[k].to);
                    for (size_t k = 0; k < nodes[(size_t)ref].pred.size(); k++) again.push_back(nodes[(size_t)ref].pred.links()[k].to);
                    for (int x : again)
                        if (!queued[(size_t)x] && nodes[(size_t)x].alive) { queue.push_back(x); queued[(size_t)x] = 1; }
                }
            }
        }
        if (!merged_any) break;
    }
    g->finish();
    return 0;
}

static inline void put_int(std::string &o, int v) {
    char b[16]; int n = 0;
    if (v == 0) b[n++] = '0';
    while (v > 0) { b[n++] = (char)('0' + v % 10); v /= 10; }
    while (n > 0) o.push_back(b[--n]);
}

/* utils.py:710-839 write_gfa as reveal_amd/alngraph.py writes GFA1: H, then per sequence node (numbered from 1 in dictionary order) its S line -- the text of
 * its interval, upper-cased when aligned -- and an L line per link to a sequence node, then a P line per path (names[0 .. npaths), path id = position),
 * walked from its start sentinel.  The text stays with the graph until it is freed; *out points at it, the return value is its length. */
static int64_t graph_gfa(rv_graph *g, const char *T, int npaths, const char *const *names, const char *cmdline, const char **out) {
    g->finish();      // (the numbering follows the graph as it stands)
    auto &nodes = g->nodes; auto &edges = g->edges;
    std::string &o = g->gfa;
    o.clear();
    std::vector<int> ident(nodes.size(), 0);
    int nid = 0;
    for (int x : g->order) if (nodes[(size_t)x].aligned >= 0) ident[(size_t)x] = ++nid;
    o += "H\tVN:Z:1.0\tCL:Z:"; o += cmdline ? cmdline : ""; o += "\n";
    // S and L lines: the nodes in dictionary order, in stretches written side by side and put together afterwards
    const size_t norder = g->order.size();
    const char *env_min = getenv("RV_GFA_PARALLEL_MIN");      // (tests: the threaded form on small graphs)
    const size_t par_min = env_min ? (size_t)atoll(env_min) : 100000;
    const int nchunks = norder >= std::max<size_t>(par_min, 16) ? 16 : 1;
    std::vector<std::string> chunks((size_t)nchunks);
    auto write_chunk = [&](int c) {
        std::string &oc = chunks[(size_t)c];
        const size_t lo = norder * (size_t)c / (size_t)nchunks, hi = norder * (size_t)(c + 1) / (size_t)nchunks;
        size_t need = 64;
        for (size_t k = lo; k < hi; k++) { const GNode &n = nodes[(size_t)g->order[k]]; if (n.aligned >= 0) need += (size_t)(n.e - n.b) + 16 + 40 * n.succ.size(); }
        oc.reserve(need);
        for (size_t k = lo; k < hi; k++) {
            const int x = g->order[k];
            const GNode &n = nodes[(size_t)x];
            if (n.aligned < 0) continue;
            oc += "S\t"; put_int(oc, ident[(size_t)x]); oc += '\t';
            const size_t at = oc.size();
            oc.append(T + n.b, (size_t)(n.e - n.b));
            if (n.aligned > 0)
                for (size_t i = at; i < oc.size(); i++) { const char ch = oc[i]; if (ch >= 'a' && ch <= 'z') oc[i] = (char)(ch - 32); }
            oc += "\n";
            const Link *lk = n.succ.links();
            for (size_t q = 0; q < n.succ.size(); q++) {
                const int v = lk[q].to;
                if (nodes[(size_t)v].aligned < 0) continue;
                oc += "L\t"; put_int(oc, ident[(size_t)x]); oc += "\t+\t"; put_int(oc, ident[(size_t)v]); oc += "\t+\t0M\n";
            }
        }
    };
    // the start sentinels in the reader's order (start_of: sample s' is node 3 s of the replay, wherever compact() has moved it).  The walks are independent of each
    // other and read only: on a few threads (a hundred paths of 10^6 steps each, a cache miss per step, were half of the writer's time)
    std::vector<std::string> plines((size_t)std::max(npaths, 0));
    auto walk = [&](int sid) {
        std::string path, cigar;
        char pbuf[32];
        for (size_t sq = 0; sq < g->start_of.size(); sq++) {
            const size_t st = (size_t)g->start_of[sq];
            bool has = false;
            for (auto &a : nodes[st].off) has |= a.first == sid;
            if (!nodes[st].alive || !has) continue;
            int node = (int)st;
            for (;;) {
                int nout = 0, v = -1;
                for (int e : nodes[(size_t)node].succ)
                    if (edges[(size_t)e].paths.has(sid)) { nout++; v = edges[(size_t)e].v; }
                if (nout != 1) break;
                if (nodes[(size_t)v].sent == 2) break;      // an end sentinel
                if (nodes[(size_t)v].aligned >= 0) {
                    const int w = snprintf(pbuf, sizeof pbuf, "%s%d+", path.empty() ? "" : ",", ident[(size_t)v]);
                    path.append(pbuf, (size_t)w);
                    if (nodes[(size_t)node].aligned >= 0) { cigar += cigar.empty() ? "0M" : ",0M"; }
                }
                node = v;
            }
            break;
        }
        std::string &ln = plines[(size_t)sid];
        ln.reserve(path.size() + cigar.size() + 64);
        ln += "P\t"; ln += names[sid]; ln += "\t"; ln += path; ln += "\t"; ln += cigar; ln += "\n";
    };
    {
        // one pool for both kinds of work: the stretches of nodes first, then a path each
        const int ntasks = nchunks + std::max(npaths, 0);
        int nt = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
        nt = std::min(nt, ntasks);
        if (nodes.size() < par_min) nt = 1;
        std::atomic<int> next{0};
        std::atomic<bool> failed{false};
        auto worker = [&]() {
            try { for (;;) { const int k = next.fetch_add(1); if (k >= ntasks) return; if (k < nchunks) write_chunk(k); else walk(k - nchunks); } }
            catch (...) { failed = true; }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) { try { th.emplace_back(worker); } catch (...) { break; } }
        worker();
        for (auto &t : th) t.join();
        if (failed) throw std::bad_alloc();
    }
    {
        size_t need = o.size() + 16;
        for (const std::string &c : chunks) need += c.size();
        for (const std::string &c : plines) need += c.size();
        o.reserve(need);
        for (std::string &c : chunks) { o += c; std::string().swap(c); }
    }
    for (int sid = 0; sid < npaths; sid++) o += plines[(size_t)sid];
    *out = o.data();
    return (int64_t)o.size();
}

#define RV_GRAPH_GUARD(call)                                                                         \
    try { return call; }                                                                             \
    catch (const std::exception &e) { rv_set_error("%s: %s", __func__, e.what()); return -1; }       \
    catch (...) { rv_set_error("%s: failed", __func__); return -1; }
int rv_graph_export(const rv_graph *g, int64_t *node_b, int64_t *node_e, int8_t *node_aligned, int64_t *off_ptr, int32_t *off_sid, int64_t *off_val,
                    int64_t *succ_ptr, int32_t *succ_to, int32_t *succ_edge, int64_t *pred_ptr, int32_t *pred_from, int32_t *pred_edge, int64_t *edge_ptr, int32_t *edge_paths) {
    RV_GRAPH_GUARD(graph_export(g, node_b, node_e, node_aligned, off_ptr, off_sid, off_val, succ_ptr, succ_to, succ_edge, pred_ptr, pred_from, pred_edge, edge_ptr, edge_paths))
}
int rv_graph_prune(rv_graph *g, const char *T) { RV_GRAPH_GUARD(graph_prune(g, T)) }
int64_t rv_graph_gfa(rv_graph *g, const char *T, int npaths, const char *const *names, const char *cmdline, const char **out) { RV_GRAPH_GUARD(graph_gfa(g, T, npaths, names, cmdline, out)) }

void rv_graph_free(rv_graph *g) { delete g; }

}

// ---- the surgery as a follower of a running recursion (rv_graphrem.h) ------------------------------------------------------------------------------------------
#include <condition_variable>
#include <mutex>
struct RvReplayFeed {
    struct Chunk { std::vector<uint32_t> l; std::vector<int64_t> off, pos; };
    rv_graph *g = nullptr;
    std::thread th;
    std::mutex mu; std::condition_variable cv;
    std::deque<Chunk> q;
    bool closed = false, failed = false;
    std::string why;
    void run() {
        std::vector<int> mns;
        for (;;) {
            Chunk c;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return closed || !q.empty(); });
                if (q.empty()) return;
                c = std::move(q.front()); q.pop_front();
            }
            if (failed) continue;      // (drain)
            try {
                if (!replay_apply(g, (int64_t)c.l.size(), c.l.data(), c.off.data(), c.pos.data(), mns)) { failed = true; why = g->err; }
            } catch (const std::exception &e) { failed = true; why = std::string("the anchors' surgery: ") + e.what(); }
            catch (...) { failed = true; why = "the anchors' surgery failed"; }
        }
    }
};
RvReplayFeed *rv_replay_feed_start(rv_graph *g) {
    try {
        std::unique_ptr<RvReplayFeed> f(new RvReplayFeed());
        f->g = g;
        f->th = std::thread([p = f.get()] { p->run(); });
        return f.release();
    } catch (...) { rv_set_error("replay feed: no thread / memory to be had"); return nullptr; }
}
void rv_replay_feed_push(RvReplayFeed *f, const uint32_t *l, const int64_t *off, const int64_t *pos, size_t count) {
    if (!count) return;
    RvReplayFeed::Chunk c;
    try {
        c.l.assign(l, l + count); c.off.assign(off, off + count + 1); c.pos.assign(pos, pos + (off[count] - off[0]));
        std::lock_guard<std::mutex> lk(f->mu);
        f->q.push_back(std::move(c));
    } catch (...) { std::lock_guard<std::mutex> lk(f->mu); f->failed = true; f->why = "replay feed: out of host memory"; }
    f->cv.notify_one();
}
int rv_replay_feed_finish(RvReplayFeed *f) {
    { std::lock_guard<std::mutex> lk(f->mu); f->closed = true; }
    f->cv.notify_one();
    if (f->th.joinable()) f->th.join();
    const bool bad = f->failed;
    const std::string why = f->why;
    rv_graph *g = f->g;
    delete f;
    if (bad) { rv_set_error("%s", why.c_str()); return -1; }
    try { g->compact(); g->finish(); } catch (...) { rv_set_error("replay feed: out of host memory"); return -1; }
    return 0;
}

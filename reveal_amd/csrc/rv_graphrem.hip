// rv_graphrem.hip -- `reveal rem` for GRAPH inputs behind the ABI (host code, no device code): the two callbacks the reference runs per sub-index --
// schemes.graphmumpicker (reveal/schemes.py:197-361) and rem.graphalign (reveal/rem.py:318-382) -- on the structure of rv_graph.h.
//
// With FASTA inputs (one sequence per sample) a position's path offset is pos - begin of its sequence and graphalign's intervals follow from the match
// alone, so the picker could run by itself (rv_pick_chain) and the graph be replayed afterwards (rv_graph_replay).  With graphs as inputs (the levels
// 1 and 2 of `reveal align --order=sequential`: GFA files of earlier jobs, reveal/utils.py:377-677) neither holds: a node lies on many paths, the
// picker maps a match to per-PATH offsets through the nodes that hold its members (maptooffsets, schemes.py:128-158), and graphalign finds the
// leading / trailing intervals of a sub-index by walking the graph around the merged node (segmentgraph, rem.py:228-316) -- between two picks.  So
// both move here together, and the recursion (rv_align.hip, picker kind 2) calls them per sub-index in the reference's order.  reveal_amd/alngraph.py +
// schemes.py + rem.py are the Python forms of the same code; tests/test_cpu_graphrem_native.py runs the two side by side on every call of whole
// alignments (reference index, no GPU), tests/test_gpu_graphrem.py compares the GFA files.
//
// What is kept of the Python structure because results depend on it: dictionary order of nodes, of a node's offsets and of its links (rv_graph.h), the
// order in which a match's members are looked up and its offsets collected (an ordered mapping path -> offset; a second member on a path a first one
// already lies on overwrites the value and counts again), `mapping` keyed by the VALUES of that mapping (a later match with the same values replaces an
// earlier one), the stable sorts, the chain's rules (rv_chain), "largest of the chain" = the last of equal lengths.  Edges on the reverse strand are not
// supported (rv_graph_import refuses them, the Python callbacks take such inputs).
#include "rv_graph.h"
#include "rv_graphrem.h"
#include "rv_pick.h"
#include <chrono>
#include <cmath>
#include <unordered_map>

extern "C" int64_t rv_chain(int64_t m, int k, const uint32_t *len, const int32_t *nmem, const int64_t *crd, const int64_t *left,
                            const int64_t *right, int64_t wscore, int64_t wpen, int model, int64_t *out_idx, int64_t *out_score);

namespace {

inline uint32_t next_epoch(rv_graph *g) {
    if (g->stamp.size() < g->nodes.size()) { g->stamp.resize(g->nodes.size() + g->nodes.size() / 4 + 64, 0); g->stamp2.resize(g->stamp.size(), 0); }
    if (++g->epoch == 0) { std::fill(g->stamp.begin(), g->stamp.end(), 0u); std::fill(g->stamp2.begin(), g->stamp2.end(), 0u); g->epoch = 1; }
    return g->epoch;
}
inline void grow_stamps(rv_graph *g) {
    if (g->stamp.size() < g->nodes.size()) { g->stamp.resize(g->nodes.size() + g->nodes.size() / 4 + 64, 0); g->stamp2.resize(g->stamp.size(), 0); }
}
inline bool real_edge(const rv_graph *g, const PathSet &p, bool any_star) {      // alngraph._real: carried by at least one path whose name does not begin with '*'
    if (!any_star) return true;
    bool real = false;
    p.each([&](int sid) { real |= !(sid < (int)g->star.size() && g->star[(size_t)sid]); });
    return real;
}

// alngraph._bfs (rem.py:228-258): from `source` over edges carried by a real path; unaligned nodes are walked through (kind 0), aligned ones stop the walk (1)
// unless marked in `ignore`, sentinels stop it (2).  The caller supplies the visited epoch (a node is reported once).
struct BfsHit { int node; int kind; };
void bfs(rv_graph *g, int source, bool reverse, const std::vector<uint8_t> *ignore, std::vector<BfsHit> &out, std::vector<int> &queue) {
    bool any_star = false;
    for (uint8_t s : g->star) any_star |= s != 0;
    grow_stamps(g);
    const uint32_t ep = next_epoch(g);
    out.clear(); queue.clear();
    g->stamp[(size_t)source] = ep;
    queue.push_back(source);
    for (size_t qi = 0; qi < queue.size(); qi++) {
        const int parent = queue[qi];
        const LinkVec &adj = reverse ? g->nodes[(size_t)parent].pred : g->nodes[(size_t)parent].succ;
        for (int e : adj) {
            const int child = reverse ? g->edges[(size_t)e].u : g->edges[(size_t)e].v;
            if (g->stamp[(size_t)child] == ep || !real_edge(g, g->edges[(size_t)e].paths, any_star)) continue;
            g->stamp[(size_t)child] = ep;
            const GNode &c = g->nodes[(size_t)child];
            if (c.aligned < 0) out.push_back({child, 2});
            else if (c.aligned == 0 || (ignore && (*ignore)[(size_t)child])) { queue.push_back(child); out.push_back({child, 0}); }
            else out.push_back({child, 1});
        }
    }
}

}  // namespace

// ---- what rv_align.hip calls (rv_graphrem.h; the extern "C" entries below wrap them for the tests) ------------------------------------------------------
static int node_of(rv_graph *g, int64_t b, int64_t e, const char *what) {
    auto it = g->at.find(b);
    if (it == g->at.end() || !g->nodes[(size_t)it->second].alive || g->nodes[(size_t)it->second].e != e) { rv_set_error("graph: %s [%lld,%lld) is not a node of the graph", what, (long long)b, (long long)e); return -1; }
    return it->second;
}

// rem.graphalign (rem.py:318-382) for one sub-index: nodes = its intervals (graph nodes), left / right = its left / right graph node, the match (l, members in the
// picker's order).  The graph is changed (nodes broken and merged); the intervals of the children, the merged node and the children's left / right nodes come back.
static inline double gnow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int rv_graph_do_align(rv_graph *g, const RvGraphIv *nodes, size_t nn, RvGraphIv left, RvGraphIv right, uint32_t l, const int64_t *pos, int npos, RvGraphAlignOut &O) {
    double tp = gnow();
    auto phase = [&](int k) { const double t = gnow(); g->t_phase[k] += t - tp; tp = t; };
    O.lead.clear(); O.trail.clear(); O.match.clear(); O.rest.clear();
    std::vector<int> mine; mine.reserve(nn + 2 * (size_t)npos);
    {   // intervals -> nodes through the hash of begins (rv_graph.h BeginHash), three steps apart in a software pipeline: the slot is prefetched, then read and the node it
        // names prefetched, then the node is checked (alive, the same interval) and marked.  An interval the hash does not hold goes to the position map.
        if (!g->made_on && nn >= 64) {
            g->made.reserve(g->at.size() + g->at.size() / 2);
            for (auto &kv : g->at) { g->made.put(kv.first, kv.second); g->mark_begin(kv.first); }
            g->made_on = true;      // (what is made from here on registers itself: rv_graph::new_node)
        }
        std::vector<int> &cand = g->look_tmp;
        cand.assign(nn, -1);
        const size_t D = 12;
        for (size_t i = 0; i < nn + 2 * D; i++) {
            if (g->made_on && i < nn) __builtin_prefetch(g->made.home(nodes[i].b));
            if (g->made_on && i >= D && i - D < nn) { const int c = g->made.get(nodes[i - D].b); cand[i - D] = c; if (c >= 0) __builtin_prefetch(&g->nodes[(size_t)c]); }
            if (i >= 2 * D) {
                const size_t j = i - 2 * D;
                const int64_t b = nodes[j].b;
                int x = cand[j];
                if (x >= 0 && !(g->nodes[(size_t)x].alive && g->nodes[(size_t)x].b == b && g->nodes[(size_t)x].e == nodes[j].e)) x = -1;
                if (x < 0) {
                    auto it = g->at.find(b);
                    if (it == g->at.end() || !g->nodes[(size_t)it->second].alive || g->nodes[(size_t)it->second].e != nodes[j].e) {
                        rv_set_error("graph: interval of the sub-index [%lld,%lld) is not a node of the graph", (long long)b, (long long)nodes[j].e); return -1;
                    }
                    x = it->second;
                }
                mine.push_back(x);
            }
        }
    }
    phase(0);
    // marks live in the nodes (GNode::ep_sub / cls / ep_walk): a node belongs to this call's sub-index while its ep_sub is this call's number
    if (++g->sub_epoch == 0) { for (GNode &n : g->nodes) n.ep_sub = 0; g->sub_epoch = 1; }
    const uint32_t se = g->sub_epoch;
    for (int x : mine) { GNode &n = g->nodes[(size_t)x]; n.ep_sub = se; n.cls = 0; }
    std::vector<int> mns;
    for (int q = 0; q < npos; q++) {
        O.match.push_back({pos[q], pos[q] + (int64_t)l});
        const int old = g->fast_node_at(pos[q]);
        if (old < 0 || pos[q] + (int64_t)l > g->nodes[(size_t)old].e) { rv_set_error("graph: a member of the match lies in no node of the graph"); return -1; }
        int pn = -1, sn = -1;
        const int mn = g->breaknode(old, pos[q], (int64_t)l, &pn, &sn);
        mns.push_back(mn);
        if (pn >= 0) { mine.push_back(pn); g->nodes[(size_t)pn].ep_sub = se; g->nodes[(size_t)pn].cls = 0; }      // the pieces belong to the sub-index
        if (sn >= 0) { mine.push_back(sn); g->nodes[(size_t)sn].ep_sub = se; g->nodes[(size_t)sn].cls = 0; }
    }
    std::sort(O.match.begin(), O.match.end(), [](const RvGraphIv &a, const RvGraphIv &b) { return a.b < b.b; });
    O.match.erase(std::unique(O.match.begin(), O.match.end(), [](const RvGraphIv &a, const RvGraphIv &b) { return a.b == b.b && a.e == b.e; }), O.match.end());
    const int mn = g->mergenodes(mns);
    for (int m2 : mns) g->nodes[(size_t)m2].ep_sub = 0;      // never the match nodes
    phase(1);
    bool any_star = false;
    for (uint8_t st : g->star) any_star |= st != 0;
    // alngraph._bfs from the merged node: unaligned nodes are walked through -- those of the sub-index get the side's bit --, aligned ones and sentinels end the walk
    auto walk = [&](bool reverse, uint8_t bit) {
        if (++g->walk_epoch == 0) { for (GNode &n : g->nodes) n.ep_walk = 0; g->walk_epoch = 1; }
        const uint32_t we = g->walk_epoch;
        std::vector<int> &queue = g->walk_queue;
        queue.clear(); queue.push_back(mn);
        g->nodes[(size_t)mn].ep_walk = we;
        for (size_t qi = 0; qi < queue.size(); qi++) {
            const GNode &par = g->nodes[(size_t)queue[qi]];
            const LinkVec &adj = reverse ? par.pred : par.succ;
            const Link *lk = adj.links();
            for (size_t k = 0, nk = adj.size(); k < nk; k++) {
                GNode &c = g->nodes[(size_t)lk[k].to];
                if (c.ep_walk == we || (any_star && !real_edge(g, g->edges[(size_t)lk[k].e].paths, true))) continue;
                c.ep_walk = we;
                if (c.aligned == 0) {
                    queue.push_back(lk[k].to);
                    if (reverse) __builtin_prefetch((const char *)&c + 64);      // (its links backwards are in the node's second line)
                    if (c.ep_sub == se) c.cls |= bit;
                }
            }
        }
    };
    if (!g->literal_segments) { walk(false, 2); walk(true, 1); }
    else {
        // rem.py:282-287 / 303-308: a walk that ended at several places keeps only what a walk back from every one of them reaches as well
        auto side = [&](bool reverse, uint8_t bit) {
            std::vector<BfsHit> hits; std::vector<int> queue;
            bfs(g, mn, reverse, nullptr, hits, queue);
            std::vector<int> walked, ends;
            for (const BfsHit &h : hits) (h.kind == 0 ? walked : ends).push_back(h.node);
            if (ends.size() > 1) {
                std::vector<uint8_t> ign(g->nodes.size(), 0), back(g->nodes.size(), 0);
                for (int e : ends) ign[(size_t)e] = 1;
                std::vector<BfsHit> h2;
                for (int e : ends) { bfs(g, e, !reverse, &ign, h2, queue); for (const BfsHit &h : h2) if (h.kind == 0) back[(size_t)h.node] = 1; }
                for (int x : walked) if (back[(size_t)x] && g->nodes[(size_t)x].ep_sub == se) g->nodes[(size_t)x].cls |= bit;
            } else for (int x : walked) if (g->nodes[(size_t)x].ep_sub == se) g->nodes[(size_t)x].cls |= bit;
        };
        side(false, 2);
        side(true, 1);
    }
    phase(2);
    // leading / trailing: the walked nodes that belong to the sub-index; rest: what is left of it.  (A node both walks reach counts as leading AND trailing in
    // the reference's sets; it cannot happen in a graph whose paths run forwards only.)
    // the merged node's paths (every id, '*' paths included: set(G.offsets[mn]))
    std::vector<uint8_t> &msam = g->pmark;
    if (msam.size() < g->id2end.size() + 1) msam.resize(g->id2end.size() + 64, 0);
    for (auto &a : g->nodes[(size_t)mn].off) { if ((size_t)a.first >= msam.size()) msam.resize((size_t)a.first + 64, 0); msam[(size_t)a.first] = 1; }
    // "clean": every path of every leading (trailing) node crosses the merged node.  A merged node on every path of the graph settles it; one offending node does,
    // too (the walk over a node's offsets was the larger half of this loop: 10^7 nodes per level, tens of paths through each)
    const bool covers_all = g->nodes[(size_t)mn].off.size() >= g->id2end.size() && !g->id2end.empty();
    bool lead_clean = true, trail_clean = true;
    size_t cut[3] = {0, 0, 0};      // where the pieces begin in the three lists (what was handed in came sorted)
    for (size_t mi = 0; mi < mine.size(); mi++) {      // what was handed in and is still alive (a broken node is gone), the pieces; each once (its mark goes as it is listed)
        if (mi == nn) { cut[0] = O.lead.size(); cut[1] = O.trail.size(); cut[2] = O.rest.size(); }
        const int x = mine[mi];
        GNode &n = g->nodes[(size_t)x];
        if (!n.alive || n.ep_sub != se) continue;
        n.ep_sub = 0;
        const uint8_t c = n.cls;
        if (c & 1) { O.lead.push_back({n.b, n.e}); if (lead_clean && !covers_all) for (auto &a : n.off) if ((size_t)a.first >= msam.size() || !msam[(size_t)a.first]) { lead_clean = false; break; } }
        if (c & 2) { O.trail.push_back({n.b, n.e}); if (trail_clean && !covers_all) for (auto &a : n.off) if ((size_t)a.first >= msam.size() || !msam[(size_t)a.first]) { trail_clean = false; break; } }
        if (!c) O.rest.push_back({n.b, n.e});
    }
    for (auto &a : g->nodes[(size_t)mn].off) msam[(size_t)a.first] = 0;
    phase(3);
    auto by_b = [](const RvGraphIv &a, const RvGraphIv &b) { return a.b < b.b; };
    if (mine.size() <= nn) { cut[0] = O.lead.size(); cut[1] = O.trail.size(); cut[2] = O.rest.size(); }
    {   // the few pieces sorted and merged into the sorted rest (a sort of the whole list per call was a seventh of graphalign)
        std::vector<RvGraphIv> *lists[3] = {&O.lead, &O.trail, &O.rest};
        for (int k = 0; k < 3; k++) {
            std::vector<RvGraphIv> &v = *lists[k];
            if (cut[k] < v.size()) { std::sort(v.begin() + (ptrdiff_t)cut[k], v.end(), by_b); std::inplace_merge(v.begin(), v.begin() + (ptrdiff_t)cut[k], v.end(), by_b); }
            if (!std::is_sorted(v.begin(), v.end(), by_b)) std::sort(v.begin(), v.end(), by_b);      // (the caller's intervals were not sorted)
        }
    }
    O.merged = {g->nodes[(size_t)mn].b, g->nodes[(size_t)mn].e};
    O.newleft = O.newright = O.merged;
    if (!lead_clean) O.newright = right;        // no clean dissection of all paths on the left (rem.py:367-370)
    if (!trail_clean) O.newleft = left;
    phase(4);
    return 0;
}

namespace {
// a match after maptooffsets: its entry in the trimmed list, the number of (real) path crossings, and its offsets as an ordered mapping path -> offset
struct RelMum { uint32_t i; int32_t n; uint32_t first, cnt; };      // its (sid, value) pairs: pt[first .. first + cnt)
}

// schemes.graphmumpicker, not-precomputed branch (schemes.py:197-361), for one sub-index of a graph alignment.  left / right: the sub-index' left / right graph
// node (b < 0: None).  Returns like rv_pick_chain: 1 picked, 0 the reference's `()`, -2 where the reference's own code raises, -1 bad arguments.
int rv_graph_do_pick(rv_graph *g, const rv_picker_args *A, int nsub, int64_t m, const uint32_t *l, const int32_t *n, const int64_t *off, const uint16_t *so, const int64_t *pos,
                     RvGraphIv left, RvGraphIv right, int minlength, rv_picker_out *O) {
    if (!g || !A || !O || m < 0) { rv_set_error("rv_graph_pick: bad arguments"); return -1; }
    O->picked = 0; O->nleft = O->nright = 0; O->nseed_members = 0;
    if (m == 0) return 0;
    const PkCtx X{so, pos};
    auto item = [&](int64_t i) { PkItem x; x.l = l[i]; x.shift = 0; x.off = off[i]; x.n = n[i]; x.nm = (int32_t)(off[i + 1] - off[i]); return x; };
    auto ids_of = [&](const PkItem &x) { std::vector<uint16_t> k(so + x.off, so + x.off + x.nm); std::sort(k.begin(), k.end()); return k; };
    std::vector<PkItem> mm;
    for (int64_t i = 0; i < m; i++) if (n[i] == nsub) mm.push_back(item(i));
    if (mm.empty() && nsub > 2) {      // schemes.segment (:107-126): the sample subset whose matches cover the most, the first of equal ones in order of appearance
        std::vector<std::vector<uint16_t>> reps; std::vector<int64_t> zsum; std::vector<int32_t> grp((size_t)m);
        for (int64_t i = 0; i < m; i++) {
            const std::vector<uint16_t> ids = ids_of(item(i));
            size_t q = 0;
            for (; q < reps.size(); q++) if (reps[q] == ids) break;
            if (q == reps.size()) { reps.push_back(ids); zsum.push_back(0); }
            zsum[q] += l[i]; grp[(size_t)i] = (int32_t)q;
        }
        int64_t best = 0; size_t part = (size_t)-1;
        for (size_t q = 0; q < reps.size(); q++) { const int64_t z = zsum[q] * (int64_t)reps[q].size(); if (z > best) { best = z; part = q; } }
        if (part == (size_t)-1) { rv_set_error("rv_graph_pick: no sample subset (the reference raises KeyError here)"); return -2; }
        for (int64_t i = 0; i < m; i++) if (grp[(size_t)i] == (int32_t)part) mm.push_back(item(i));
    }
    if (A->trim) {
        if (!mm.empty() && !pk_trim_overlap(mm, X)) { rv_set_error("rv_graph_pick: trim_overlap ran out of matches (the reference raises IndexError here)"); return -2; }
        if (mm.empty()) return 0;
    }
    if (mm.empty()) return 0;
    std::stable_sort(mm.begin(), mm.end(), [](const PkItem &a, const PkItem &b) { return a.l > b.l; });      // :240
    // maptooffsets (:128-158): every member's node, every path through it
    const size_t cnt = mm.size();
    std::vector<std::pair<int32_t, int64_t>> pt;
    std::vector<RelMum> rel(cnt);
    std::vector<int32_t> &where = g->pwhere;      // path id -> place in the current match's mapping (valid while pmark is set)
    std::vector<uint8_t> &pm = g->pmark;
    const size_t np = std::max(g->id2end.size(), g->star.size()) + 1;
    if (where.size() < np) where.resize(np, 0);
    if (pm.size() < np) pm.resize(np, 0);
    for (size_t i = 0; i < cnt; i++) {
        RelMum &r = rel[i];
        r.i = (uint32_t)i; r.n = 0; r.first = (uint32_t)pt.size();
        for (int q = 0; q < mm[i].nm; q++) {
            const int64_t p = X.at(mm[i], (size_t)q);
            const int x = g->fast_node_at(p);
            if (x < 0) { for (size_t z = r.first; z < pt.size(); z++) pm[(size_t)pt[z].first] = 0; rv_set_error("rv_graph_pick: a match's member lies in no node of the graph"); return -1; }
            const GNode &nd = g->nodes[(size_t)x];
            for (auto &a : nd.off) {
                if ((size_t)a.first < g->star.size() && g->star[(size_t)a.first]) continue;
                if ((size_t)a.first >= pm.size()) { pm.resize((size_t)a.first + 64, 0); where.resize(pm.size(), 0); }
                r.n++;
                if (pm[(size_t)a.first]) pt[(size_t)where[(size_t)a.first]].second = a.second + (p - nd.b);      // (a second member on the same path: the value is replaced, the place kept)
                else { pm[(size_t)a.first] = 1; where[(size_t)a.first] = (int32_t)pt.size(); pt.push_back({a.first, a.second + (p - nd.b)}); }
            }
        }
        r.cnt = (uint32_t)(pt.size() - r.first);
        for (size_t z = r.first; z < pt.size(); z++) pm[(size_t)pt[z].first] = 0;
    }
    // mapping[tuple(values)] = the LAST match with these values (:150-158)
    auto same_vals = [&](size_t a, size_t b) {
        if (rel[a].cnt != rel[b].cnt) return false;
        for (uint32_t z = 0; z < rel[a].cnt; z++) if (pt[rel[a].first + z].second != pt[rel[b].first + z].second) return false;
        return true;
    };
    auto val_hash = [&](size_t i) { uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)rel[i].cnt; for (uint32_t z = 0; z < rel[i].cnt; z++) h ^= (uint64_t)pt[rel[i].first + z].second + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h; };
    std::unordered_map<uint64_t, uint32_t> last_by_hash;      // (made when somebody asks: most calls hold one or two matches)
    bool hashed = false;
    auto mapped = [&](size_t i) -> size_t {
        if (cnt <= 8) { size_t r = i; for (size_t j = i + 1; j < cnt; j++) if (same_vals(j, i)) r = j; return r; }
        if (!hashed) { last_by_hash.reserve(cnt * 2); for (size_t j = 0; j < cnt; j++) last_by_hash[val_hash(j)] = (uint32_t)j; hashed = true; }
        const auto it = last_by_hash.find(val_hash(i));
        if (it != last_by_hash.end() && same_vals(it->second, i)) return it->second;
        size_t r = i;      // (two different value tuples under one hash: the walk)
        for (size_t j = i + 1; j < cnt; j++) if (same_vals(j, i)) r = j;
        return r;
    };
    std::vector<uint32_t> ord(cnt);
    for (size_t i = 0; i < cnt; i++) ord[i] = (uint32_t)i;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return rel[a].n != rel[b].n ? rel[a].n < rel[b].n : mm[a].l < mm[b].l; });      // :247
    auto keys_of = [&](size_t i) { std::vector<int32_t> k; k.reserve(rel[i].cnt); for (uint32_t z = 0; z < rel[i].cnt; z++) k.push_back(pt[rel[i].first + z].first); std::sort(k.begin(), k.end()); return k; };
    const std::vector<int32_t> last = keys_of(ord.back());      // ascending path ids: the chain's dimensions (schemes.chain sorts the keys)
    std::vector<uint32_t> relm;
    for (uint32_t i : ord) if (rel[i].cnt == last.size() && keys_of(i) == last) relm.push_back(i);
    if (relm.empty()) return 0;
    const int k = (int)last.size();
    if (k < 1) return 0;
    auto off_of = [&](const GNode &nd, int sid, int64_t *out) { for (auto &a : nd.off) if (a.first == sid) { *out = a.second; return true; } return false; };
    std::vector<int64_t> lf((size_t)k), rt((size_t)k);
    if (left.b >= 0) {
        const int x = node_of(g, left.b, left.e, "the sub-index' left node");
        if (x < 0) return -1;
        for (int j = 0; j < k; j++) { int64_t o; if (!off_of(g->nodes[(size_t)x], last[(size_t)j], &o)) { rv_set_error("rv_graph_pick: the left node is not on path %d (the reference raises KeyError here)", last[(size_t)j]); return -2; } lf[(size_t)j] = o + (left.e - left.b) - 1; }
    } else for (int j = 0; j < k; j++) lf[(size_t)j] = -1;
    if (right.b >= 0) {
        const int x = node_of(g, right.b, right.e, "the sub-index' right node");
        if (x < 0) return -1;
        for (int j = 0; j < k; j++) { int64_t o; if (!off_of(g->nodes[(size_t)x], last[(size_t)j], &o)) { rv_set_error("rv_graph_pick: the right node is not on path %d (the reference raises KeyError here)", last[(size_t)j]); return -2; } rt[(size_t)j] = o; }
    } else for (int j = 0; j < k; j++) { if ((size_t)last[(size_t)j] >= g->id2end.size()) { rv_set_error("rv_graph_pick: path without a length"); return -1; } rt[(size_t)j] = g->id2end[(size_t)last[(size_t)j]]; }
    size_t split;
    std::vector<std::pair<size_t, int64_t>> chained;
    if (relm.size() == 1) split = relm[0];
    else {
        if (A->maxmums > 0 && (int64_t)relm.size() > A->maxmums) relm.erase(relm.begin(), relm.end() - (ptrdiff_t)A->maxmums);      // :287-289
        const int64_t mc = (int64_t)relm.size();
        std::vector<uint32_t> cl((size_t)mc); std::vector<int32_t> cn((size_t)mc); std::vector<int64_t> crd((size_t)mc * k), oi((size_t)mc), osc((size_t)mc);
        std::vector<std::pair<int32_t, int64_t>> row((size_t)k);
        for (int64_t i = 0; i < mc; i++) {
            cl[(size_t)i] = (uint32_t)mm[relm[(size_t)i]].l; cn[(size_t)i] = rel[relm[(size_t)i]].n;
            // (the match's paths ARE `last`, as a set: its pairs sorted by path id are the row -- looked up one by one they cost k^2 per match, 10^7 steps per call of
            //  a hundred paths and a thousand matches)
            const RelMum &r = rel[relm[(size_t)i]];
            std::copy(pt.begin() + r.first, pt.begin() + r.first + r.cnt, row.begin());
            std::sort(row.begin(), row.end(), [](const std::pair<int32_t, int64_t> &a, const std::pair<int32_t, int64_t> &b) { return a.first < b.first; });
            for (int j = 0; j < k; j++) crd[(size_t)i * k + j] = row[(size_t)j].second;
        }
        const int64_t r = rv_chain(mc, k, cl.data(), cn.data(), crd.data(), lf.data(), rt.data(), A->wscore, A->wpen, A->gcmodel, oi.data(), osc.data());
        if (r < 0) return -1;
        if (r == 0) return 0;
        for (int64_t q = 0; q < r; q++) chained.push_back({relm[(size_t)oi[(size_t)q]], osc[(size_t)q]});
        split = chained[0].first;
        for (auto &c : chained) if (mm[c.first].l >= mm[split].l) split = c.first;      // "largest": sorted by length (stable), the last one (:313-315)
    }
    struct Seed { size_t i; int64_t sc; bool right; };
    std::vector<Seed> seeds;
    if (!chained.empty() && A->seedsize > 0) {
        int64_t at = 0; bool rgt = false;
        for (auto &c : chained) {
            if (c.first == split) { at = c.second; rgt = true; continue; }
            seeds.push_back({mapped(c.first), c.second - at, rgt});
        }
    }
    const size_t sm = mapped(split);
    if (minlength == 0) {      // :336-348
        long double o = 1;
        for (int j = 0; j < k; j++) o *= (long double)(rt[(size_t)j] - lf[(size_t)j]);
        const double nn2 = (double)mm[sm].n, ll = (double)mm[sm].l;
        double p = std::pow(std::pow(0.25, nn2 - 1.0), ll);
        if (p > 0) p = p < 1 ? 1.0 - std::exp(std::log(1.0 - p) * (double)o) : 1.0;
        if (p > A->pcutoff) return 0;
    }
    auto put = [&](size_t i, uint32_t *ol, int32_t *on, uint16_t *oso, int64_t *opos) -> int {
        *ol = (uint32_t)mm[i].l; *on = mm[i].n;
        for (int q = 0; q < mm[i].nm; q++) { oso[q] = so[mm[i].off + q]; opos[q] = X.at(mm[i], (size_t)q); }
        return mm[i].nm;
    };
    if ((int64_t)mm[sm].nm > O->member_cap) { rv_set_error("rv_graph_pick: output too small"); return -1; }
    O->picked = 1;
    O->pick_members = put(sm, &O->pick_l, &O->pick_n, O->pick_so, O->pick_pos);
    int64_t w = 0, wm = 0;
    for (const Seed &s2 : seeds) {
        if (mm[s2.i].l < A->seedsize) continue;
        if (w >= O->seed_cap || wm + (int64_t)mm[s2.i].nm > O->seed_member_cap) { rv_set_error("rv_graph_pick: seed output too small"); return -1; }
        O->seed_off[w] = wm;
        wm += put(s2.i, &O->seed_l[w], &O->seed_n[w], O->seed_so + wm, O->seed_pos + wm);
        O->seed_score[w] = s2.sc;
        O->seed_right[w] = s2.right ? 1 : 0;
        if (s2.right) O->nright++; else O->nleft++;
        w++;
    }
    O->seed_off[w] = wm;
    O->nseed_members = wm;
    return 1;
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------------------------------
extern "C" {

/* The Python layer's graph after its readers (alngraph.read_gfa / read_fasta), in dictionary order: nodes (b, e, aligned; sentinels: aligned = -1 and sent = 1
 * start / 2 end), their offsets as CSR, the links in the order "node by node, each node's links forwards in dictionary order" (their path ids as CSR), and
 * every node's links BACKWARDS as edge numbers in dictionary order; per path: is its name a '*' name, its length.  Links on the reverse strand have no place
 * here (the caller keeps the Python callbacks for such inputs). */
rv_graph *rv_graph_import(int64_t nnodes, const int64_t *node_b, const int64_t *node_e, const int8_t *node_aligned, const int8_t *node_sent,
                          const int64_t *off_ptr, const int32_t *off_sid, const int64_t *off_val,
                          int64_t nedges, const int32_t *edge_u, const int32_t *edge_v, const int64_t *edge_ptr, const int32_t *edge_paths,
                          const int64_t *pred_ptr, const int32_t *pred_edge, int npaths, const uint8_t *star, const int64_t *id2end,
                          int nstart, const int32_t *start_nodes, int literal_segments) {
    try {
        std::unique_ptr<rv_graph> own(new rv_graph());
        rv_graph *g = own.get();
        g->nseq = npaths;
        g->nodes.reserve((size_t)nnodes + 16); g->edges.reserve((size_t)nedges + 16);
        for (int64_t i = 0; i < nnodes; i++) {
            const int x = g->new_node(node_b[i], node_e[i], node_aligned[i]);
            GNode &n = g->nodes[(size_t)x];
            n.sent = node_sent[i];
            n.off.reserve((size_t)(off_ptr[i + 1] - off_ptr[i]));
            for (int64_t q = off_ptr[i]; q < off_ptr[i + 1]; q++) n.off.push_back({off_sid[q], off_val[q]});
        }
        for (int64_t e = 0; e < nedges; e++) {
            if (edge_u[e] < 0 || edge_u[e] >= nnodes || edge_v[e] < 0 || edge_v[e] >= nnodes) { rv_set_error("rv_graph_import: a link names a node that is not there"); return nullptr; }
            GEdge ed; ed.u = edge_u[e]; ed.v = edge_v[e];
            for (int64_t q = edge_ptr[e]; q < edge_ptr[e + 1]; q++) ed.paths.add(edge_paths[q]);
            g->edges.push_back(std::move(ed));
            g->nodes[(size_t)edge_u[e]].succ.push_back((int)e, edge_v[e]);
        }
        for (int64_t i = 0; i < nnodes; i++)
            for (int64_t q = pred_ptr[i]; q < pred_ptr[i + 1]; q++) {
                if (pred_edge[q] < 0 || pred_edge[q] >= nedges || g->edges[(size_t)pred_edge[q]].v != (int)i) { rv_set_error("rv_graph_import: a node's backward links do not match the links"); return nullptr; }
                g->nodes[(size_t)i].pred.push_back(pred_edge[q], g->edges[(size_t)pred_edge[q]].u);
            }
        g->star.assign(star, star + npaths); g->id2end.assign(id2end, id2end + npaths);
        g->start_of.assign(start_nodes, start_nodes + nstart);
        g->literal_segments = literal_segments != 0;
        g->finish();
        return own.release();
    } catch (const std::exception &e) { rv_set_error("rv_graph_import: %s", e.what()); return nullptr; }
    catch (...) { rv_set_error("rv_graph_import: failed"); return nullptr; }
}

/* graphalign for one sub-index (tests: beside rem.GraphAligner.graphalign).  nodes: nn (begin, end) pairs; left / right: (begin, end) or begin < 0 for None.
 * counts[0..3] = leading, trailing, matching, rest intervals; out6 = merged, newleft, newright as (begin, end) pairs (begin < 0: None).  The intervals
 * themselves: rv_graph_align_fetch (lead, trail, match, rest back to back, (begin, end) pairs). */
int rv_graph_align(rv_graph *g, const int64_t *nodes, int64_t nn, const int64_t *left, const int64_t *right, uint32_t l, const int64_t *pos, int npos,
                   int64_t *counts, int64_t *out6) {
    try {
        static_assert(sizeof(RvGraphIv) == 16, "RvGraphIv layout");
        RvGraphAlignOut &O = *reinterpret_cast<RvGraphAlignOut *>(g->align_out());
        if (rv_graph_do_align(g, reinterpret_cast<const RvGraphIv *>(nodes), (size_t)nn, RvGraphIv{left[0], left[1]}, RvGraphIv{right[0], right[1]}, l, pos, npos, O) != 0) return -1;
        counts[0] = (int64_t)O.lead.size(); counts[1] = (int64_t)O.trail.size(); counts[2] = (int64_t)O.match.size(); counts[3] = (int64_t)O.rest.size();
        out6[0] = O.merged.b; out6[1] = O.merged.e; out6[2] = O.newleft.b; out6[3] = O.newleft.e; out6[4] = O.newright.b; out6[5] = O.newright.e;
        return 0;
    } catch (const std::exception &e) { rv_set_error("rv_graph_align: %s", e.what()); return -1; }
    catch (...) { rv_set_error("rv_graph_align: failed"); return -1; }
}
int rv_graph_align_fetch(rv_graph *g, int64_t *out) {
    RvGraphAlignOut &O = *reinterpret_cast<RvGraphAlignOut *>(g->align_out());
    size_t at = 0;
    for (const std::vector<RvGraphIv> *v : {&O.lead, &O.trail, &O.match, &O.rest}) for (const RvGraphIv &x : *v) { out[at++] = x.b; out[at++] = x.e; }
    return 0;
}
int rv_graph_pick(rv_graph *g, const rv_picker_args *args, int nsub, int64_t m, const uint32_t *l, const int32_t *n, const int64_t *off, const uint16_t *so, const int64_t *pos,
                  const int64_t *left, const int64_t *right, int minlength, rv_picker_out *out) {
    try { return rv_graph_do_pick(g, args, nsub, m, l, n, off, so, pos, RvGraphIv{left[0], left[1]}, RvGraphIv{right[0], right[1]}, minlength, out); }
    catch (const std::exception &e) { rv_set_error("rv_graph_pick: %s", e.what()); return -1; }
    catch (...) { rv_set_error("rv_graph_pick: failed"); return -1; }
}

}  // extern "C"

// the one RvGraphAlignOut of a graph (rv_graph.h keeps it as an opaque pointer: the type lives here)
void *rv_graph::align_out() {
    if (!align_out_) align_out_ = new RvGraphAlignOut();
    return align_out_;
}
void rv_graph_align_out_free(void *p) { delete reinterpret_cast<RvGraphAlignOut *>(p); }

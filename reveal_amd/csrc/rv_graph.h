// rv_graph.h -- the alignment graph behind the ABI (host code): the structure rv_graph.hip builds from a finished run's anchors (rv_graph_replay) and
// rv_graphrem.hip works on while a run goes on (graph inputs: rv_graph_import, the picker and graphalign of `reveal rem` for graphs).
#pragma once
#include "../../include/reveal_amd.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <map>
#include <memory>
#include <string>
#include <vector>

void rv_set_error(const char *fmt, ...);      // rv_api.hip

struct GNode {
    int64_t b, e;                                   // text interval; sentinels: b = sample, e = 0 (start) / 1 (end)
    int8_t aligned;                                 // -1: sentinel
    int8_t sent = 0;                                // sentinels: 1 a start node, 2 an end node
    bool alive;
    // graphalign's marks, here and not in arrays of their own: the walk looks at a node's `aligned` anyway, so the marks cost no cache line of their own (a level of
    // the last job of config 5 touches 10^7 nodes a dozen times).  Valid while the epoch matches (rv_graph::sub_epoch / walk_epoch): nothing is reset between calls.
    uint8_t cls = 0;                                // bit 0 leading, bit 1 trailing (while ep_sub == sub_epoch)
    uint32_t ep_sub = 0, ep_walk = 0;               // belongs to the sub-index of the current graphalign call / reached by the current walk
    uint64_t order;                                 // position in the graph's node dictionary (creation order)
    std::vector<std::pair<int, int64_t>> off;       // path id -> offset, in dictionary order
    std::vector<int> succ, pred;                    // edge ids, in dictionary order
};
// The path ids an edge carries.  A graph of up to 256 paths keeps them as four words (uniting two sets, and asking for a member, cost a few instructions
// instead of an allocation: the anchors' surgery spent most of its time in malloc -- and again when the last job of config 5, a hundred paths, outgrew the
// one word this began with: 64 of graphalign's 73 us per call); more paths: a sorted vector as before.
struct PathSet {
    static constexpr int W = 4;
    uint64_t m[W] = {0, 0, 0, 0};
    std::vector<int> v;        // used when `big`
    bool big = false;
    PathSet() = default;
    explicit PathSet(const std::vector<int> &ids) { for (int p : ids) add(p); }
    void add(int p) {
        if (!big && p >= 0 && p < 64 * W) { m[p >> 6] |= 1ull << (p & 63); return; }
        grow();
        auto it = std::lower_bound(v.begin(), v.end(), p);
        if (it == v.end() || *it != p) v.insert(it, p);
    }
    void grow() {
        if (big) return;
        big = true;
        for (int q = 0; q < 64 * W; q++) if ((m[q >> 6] >> (q & 63)) & 1ull) v.push_back(q);
        for (int w = 0; w < W; w++) m[w] = 0;
    }
    void unite(const PathSet &o) {
        if (!big && !o.big) { for (int w = 0; w < W; w++) m[w] |= o.m[w]; return; }
        grow();
        if (o.big) { std::vector<int> r; r.reserve(v.size() + o.v.size()); std::set_union(v.begin(), v.end(), o.v.begin(), o.v.end(), std::back_inserter(r)); v.swap(r); }
        else for (int q = 0; q < 64 * W; q++) if ((o.m[q >> 6] >> (q & 63)) & 1ull) add(q);
    }
    bool has(int p) const { return big ? std::binary_search(v.begin(), v.end(), p) : (p >= 0 && p < 64 * W && ((m[p >> 6] >> (p & 63)) & 1ull)); }
    size_t size() const {
        if (big) return v.size();
        size_t c = 0;
        for (int w = 0; w < W; w++) c += (size_t)__builtin_popcountll(m[w]);
        return c;
    }
    template <class F> void each(F f) const {      // ascending
        if (big) { for (int p : v) f(p); return; }
        for (int w = 0; w < W; w++) for (uint64_t x = m[w]; x; x &= x - 1) f(64 * w + __builtin_ctzll(x));
    }
};
struct GEdge { int u, v; PathSet paths; };

struct rv_graph {
    std::vector<GNode> nodes;
    std::vector<GEdge> edges;
    std::map<int64_t, int> at;                      // begin -> node, sequence nodes that are alive
    uint64_t counter = 0;
    std::vector<int> order;                         // export: alive nodes in dictionary order
    std::vector<int> edge_no;                       // export: edge id -> dense number (-1: dead)
    std::string err, gfa, names_buf;                // (names_buf: the path names the last rv_graph_read_gfa added, one per line)
    int nseq = 0;
    std::vector<std::pair<int, PathSet>> in_tmp, out_tmp;      // scratch of breaknode / mergenodes: a node's links while it is taken apart
    std::vector<int> start_of;                      // the start sentinels (FASTA reader: one per sequence, in the reader's order; graph inputs: one per component)
    // graph inputs (rv_graph_import): what the picker and graphalign look at besides the nodes
    std::vector<uint8_t> star;                      // path id -> its name begins with '*' (not a real path: walks and look-ups pass it over)
    std::vector<int64_t> id2end;                    // path id -> its length
    bool literal_segments = false;                  // segmentgraph in the reference's form (alngraph.check_segment_shortcut said no)
    std::vector<uint32_t> stamp, stamp2; uint32_t epoch = 0;      // scratch of the walks: visited marks per node
    std::vector<uint8_t> mark, mark2, pmark; std::vector<int32_t> pwhere;      // scratch of graphalign / the picker (all zero between calls)
    double t_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // graphalign: seconds in look-ups / breaks + merge / walks / lists / sorts (RV_GRAPH_TIMES=1 prints them when the graph is renumbered)
    uint32_t sub_epoch = 0, walk_epoch = 0; std::vector<int> walk_queue;      // graphalign: see GNode::ep_sub / ep_walk
    std::vector<int64_t> orig_b; std::vector<int> orig_id;      // graphalign: begin -> node of the nodes that were there when the run began, sorted (see rv_graph_do_align)
    void *align_out_ = nullptr;                     // rv_graphrem.hip: the result of the last rv_graph_do_align through the C ABI
    void *align_out();
    ~rv_graph();

    int new_node(int64_t b, int64_t e, int8_t aligned) {
        GNode n;
        n.b = b; n.e = e; n.aligned = aligned; n.alive = true; n.order = counter++;
        nodes.push_back(std::move(n));
        const int id = (int)nodes.size() - 1;
        if (aligned >= 0) at[b] = id;
        return id;
    }
    // alngraph.py add_edge: one edge per (u, v); adding it again unites the path sets
    void add_edge(int u, int v, const PathSet &paths) {
        for (int e : nodes[(size_t)u].succ)
            if (edges[(size_t)e].v == v) { edges[(size_t)e].paths.unite(paths); return; }
        edges.push_back({u, v, paths});
        const int e = (int)edges.size() - 1;
        nodes[(size_t)u].succ.push_back(e);
        nodes[(size_t)v].pred.push_back(e);
    }
    void remove_node(int x) {
        GNode &n = nodes[(size_t)x];
        for (int e : n.succ) { auto &p = nodes[(size_t)edges[(size_t)e].v].pred; p.erase(std::find(p.begin(), p.end(), e)); edges[(size_t)e].u = -1; }
        for (int e : n.pred) { auto &s = nodes[(size_t)edges[(size_t)e].u].succ; s.erase(std::find(s.begin(), s.end(), e)); edges[(size_t)e].u = -1; }
        n.succ.clear(); n.pred.clear(); n.off.clear(); n.alive = false;
        auto it = at.find(n.b);
        if (n.aligned >= 0 && it != at.end() && it->second == x) at.erase(it);
    }
    int node_at(int64_t pos) {
        auto it = at.upper_bound(pos);
        if (it == at.begin()) return -1;
        --it;
        const GNode &n = nodes[(size_t)it->second];
        return (n.alive && pos < n.e) ? it->second : -1;
    }
    // rem.py:14-131 for a node every path crosses forwards
    int breaknode(int x, int64_t pos, int64_t l, int *prefix = nullptr, int *suffix = nullptr) {      // (prefix / suffix: the nodes made in front of / behind the match, -1 none)
        if (prefix) *prefix = -1;
        if (suffix) *suffix = -1;
        const int64_t nb = nodes[(size_t)x].b, ne = nodes[(size_t)x].e;
        if (nb == pos && ne == pos + l) return x;
        const std::vector<std::pair<int, int64_t>> att = std::move(nodes[(size_t)x].off);      // (the node is about to go)
        in_tmp.clear(); out_tmp.clear();
        for (int e : nodes[(size_t)x].pred) in_tmp.push_back({edges[(size_t)e].u, edges[(size_t)e].paths});
        for (int e : nodes[(size_t)x].succ) out_tmp.push_back({edges[(size_t)e].v, edges[(size_t)e].paths});
        const size_t n_in = in_tmp.size(), n_out = out_tmp.size();
        PathSet pospaths;
        if (n_in == 0 && n_out == 0) {
            for (auto &a : att) pospaths.add(a.first);
        } else {
            for (size_t k = 0; k < n_in; k++) pospaths.unite(in_tmp[k].second);
            for (size_t k = 0; k < n_out; k++) pospaths.unite(out_tmp[k].second);
        }
        // (the old node leaves the position map first: the match or prefix node shares its begin)
        { auto it = at.find(nb); if (it != at.end() && it->second == x) at.erase(it); }
        const int mn = new_node(pos, pos + l, 0);
        nodes[(size_t)mn].off.reserve(att.size());
        for (auto &a : att) nodes[(size_t)mn].off.push_back({a.first, a.second + (pos - nb)});
        int pn = mn, sn = mn;
        if (nb != pos) {
            pn = new_node(nb, pos, 0);
            nodes[(size_t)pn].off = att;
            add_edge(pn, mn, pospaths);
            if (prefix) *prefix = pn;
        }
        if (ne != pos + l) {
            sn = new_node(pos + l, ne, 0);
            nodes[(size_t)sn].off.reserve(att.size());
            for (auto &a : att) nodes[(size_t)sn].off.push_back({a.first, a.second + (pos + l - nb)});
            add_edge(mn, sn, pospaths);
            if (suffix) *suffix = sn;
        }
        remove_node(x);
        for (size_t k = 0; k < n_in; k++) add_edge(in_tmp[k].first, pn, in_tmp[k].second);
        for (size_t k = 0; k < n_out; k++) add_edge(sn, out_tmp[k].first, out_tmp[k].second);
        return mn;
    }
    // rem.py:133-200: the first node absorbs the others
    int mergenodes(const std::vector<int> &mns) {
        const int ref = mns[0];
        std::vector<std::pair<int, int64_t>> merged;      // an ordered mapping path -> offset: a later node's value for a path that is there replaces it in place
        for (int x : mns)
            for (auto &a : nodes[(size_t)x].off) {
                if ((size_t)a.first >= pmark.size()) { pmark.resize((size_t)a.first + 64, 0); pwhere.resize(pmark.size(), 0); }
                if (pwhere.size() < pmark.size()) pwhere.resize(pmark.size(), 0);
                if (pmark[(size_t)a.first]) merged[(size_t)pwhere[(size_t)a.first]].second = a.second;
                else { pmark[(size_t)a.first] = 1; pwhere[(size_t)a.first] = (int32_t)merged.size(); merged.push_back(a); }
            }
        for (auto &a : merged) pmark[(size_t)a.first] = 0;
        nodes[(size_t)ref].off.swap(merged);
        nodes[(size_t)ref].aligned = 1;
        for (size_t k = 1; k < mns.size(); k++) {
            const int x = mns[k];
            if (x == ref) continue;
            in_tmp.clear(); out_tmp.clear();
            for (int e : nodes[(size_t)x].pred) in_tmp.push_back({edges[(size_t)e].u, edges[(size_t)e].paths});
            for (int e : nodes[(size_t)x].succ) out_tmp.push_back({edges[(size_t)e].v, edges[(size_t)e].paths});
            for (size_t k2 = 0; k2 < in_tmp.size(); k2++) add_edge(in_tmp[k2].first, ref, in_tmp[k2].second);
            for (size_t k2 = 0; k2 < out_tmp.size(); k2++) add_edge(ref, out_tmp[k2].first, out_tmp[k2].second);
            remove_node(x);
        }
        return ref;
    }
    // The surgery leaves three dead nodes for every live one (a broken node stays in the array): prune_nodes and the writer then walk a structure four
    // times the size it needs to be, a cache miss per step.  Live nodes and edges move together, in their old order (a node's number IS its place in the
    // dictionary), ids are renamed.
    void compact() {
        std::vector<int> nmap(nodes.size(), -1), emap(edges.size(), -1);
        size_t nn = 0;
        for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].alive) nmap[i] = (int)nn++;
        size_t ne = 0;
        for (size_t e = 0; e < edges.size(); e++) if (edges[e].u >= 0 && nodes[(size_t)edges[e].u].alive && nodes[(size_t)edges[e].v].alive) emap[e] = (int)ne++;
        std::vector<GNode> n2; n2.reserve(nn);
        for (size_t i = 0; i < nodes.size(); i++) {
            if (!nodes[i].alive) continue;
            GNode &n = nodes[i];
            for (int &e : n.succ) e = emap[(size_t)e];
            for (int &e : n.pred) e = emap[(size_t)e];
            n2.push_back(std::move(n));
        }
        std::vector<GEdge> e2; e2.reserve(ne);
        for (size_t e = 0; e < edges.size(); e++) if (emap[e] >= 0) { GEdge &x = edges[e]; x.u = nmap[(size_t)x.u]; x.v = nmap[(size_t)x.v]; e2.push_back(std::move(x)); }
        nodes.swap(n2); edges.swap(e2);
        for (auto &kv : at) kv.second = nmap[(size_t)kv.second];
        for (int &x : start_of) x = nmap[(size_t)x];
        orig_b.clear(); orig_id.clear();
    }
    void finish() {
        order.clear();
        for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].alive) order.push_back((int)i);      // (a node's number IS its creation order: new_node hands both out together)
        edge_no.assign(edges.size(), -1);
        int ne = 0;
        for (int x : order) for (int e : nodes[(size_t)x].succ) edge_no[(size_t)e] = ne++;
    }
};


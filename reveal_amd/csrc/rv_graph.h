// rv_graph.h -- the alignment graph behind the ABI (host code): the structure rv_graph.hip builds from a finished run's anchors (rv_graph_replay) and
// rv_graphrem.hip works on while a run goes on (graph inputs: rv_graph_import, the picker and graphalign of `reveal rem` for graphs).
#pragma once
#include "../../include/reveal_amd.h"
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <vector>

void rv_set_error(const char *fmt, ...);      // rv_api.hip

// The graph's large arrays (nodes, links, the hash of begins) are read all over -- a walk's next node is anywhere in a gigabyte -- so with 4 KB pages nearly every access
// also misses the TLB.  With RV_HUGEPAGES=1 in the environment allocations of 4 MB and more are made on 2 MB boundaries and the kernel is asked for huge pages (transparent
// huge pages in `madvise` mode).  Off by default: where memory is fragmented the kernel compacts it at the first touch, and a graph that grows by doubling pays that again and
// again (the anchors' surgery of five 5 Mbp genomes: 2.7 -> 11 s in the build container).
#include <sys/mman.h>
inline bool rv_hugepages() { static const bool on = [] { const char *e = getenv("RV_HUGEPAGES"); return e && *e && *e != '0'; }(); return on; }
template <class T> struct HugeAlloc {
    typedef T value_type;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        void *p = nullptr;
        if (bytes >= ((size_t)4 << 20) && rv_hugepages()) {
            const size_t huge = (size_t)2 << 20, sz = (bytes + huge - 1) / huge * huge;
            p = aligned_alloc(huge, sz);
            if (p) (void)madvise(p, sz, MADV_HUGEPAGE);
        } else {
            const size_t al = alignof(T) > 16 ? alignof(T) : 16, sz = (bytes + al - 1) / al * al;
            p = aligned_alloc(al, sz ? sz : al);
        }
        if (!p) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t) { free(p); }
    template <class U> bool operator==(const HugeAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};

// A node's links in one direction: (edge id, neighbour) pairs in dictionary order, the first two inside the node.  graphalign's walks go from node to node along
// chains of small bubbles, every step waiting for the one before: with a vector of edge ids a step was three dependent loads (the vector's heap block, the edge,
// the neighbour); now it is the neighbour alone.  Iterating yields the edge ids, as the vector did.
struct Link { int e, to; };
class LinkVec {
    uint32_t n_ = 0, cap_ = 2;
    union { Link inl_[2]; Link *heap_; };
    Link *p() { return cap_ > 2 ? heap_ : inl_; }
    const Link *p() const { return cap_ > 2 ? heap_ : inl_; }
    void take(LinkVec &o) { n_ = o.n_; cap_ = o.cap_; if (cap_ > 2) heap_ = o.heap_; else { inl_[0] = o.inl_[0]; inl_[1] = o.inl_[1]; } o.n_ = 0; o.cap_ = 2; }
    void copy(const LinkVec &o) {
        n_ = 0; cap_ = 2;
        for (uint32_t i = 0; i < o.n_; i++) push_back(o.p()[i].e, o.p()[i].to);
    }
public:
    LinkVec() { inl_[0] = inl_[1] = Link{-1, -1}; }
    ~LinkVec() { if (cap_ > 2) free(heap_); }
    LinkVec(const LinkVec &o) { copy(o); }
    LinkVec(LinkVec &&o) noexcept { take(o); }
    LinkVec &operator=(const LinkVec &o) { if (this != &o) { if (cap_ > 2) free(heap_); copy(o); } return *this; }
    LinkVec &operator=(LinkVec &&o) noexcept { if (this != &o) { if (cap_ > 2) free(heap_); take(o); } return *this; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    void clear() { n_ = 0; }
    void push_back(int e, int to) {
        if (n_ == cap_) {
            const uint32_t c2 = cap_ * 2;
            Link *q = (Link *)malloc(sizeof(Link) * c2);
            if (!q) throw std::bad_alloc();
            memcpy(q, p(), sizeof(Link) * n_);
            if (cap_ > 2) free(heap_);
            heap_ = q; cap_ = c2;
        }
        p()[n_++] = Link{e, to};
    }
    void remove(int e) {      // the first entry of edge e goes, the others keep their order
        Link *q = p();
        for (uint32_t i = 0; i < n_; i++) if (q[i].e == e) { for (uint32_t j = i + 1; j < n_; j++) q[j - 1] = q[j]; n_--; return; }
    }
    const Link *links() const { return p(); }
    Link *links() { return p(); }
    struct It { Link *q; int &operator*() const { return q->e; } It &operator++() { ++q; return *this; } bool operator!=(const It &o) const { return q != o.q; } };
    struct CIt { const Link *q; int operator*() const { return q->e; } CIt &operator++() { ++q; return *this; } bool operator!=(const CIt &o) const { return q != o.q; } };
    It begin() { return It{p()}; }
    It end() { return It{p() + n_}; }
    CIt begin() const { return CIt{p()}; }
    CIt end() const { return CIt{p() + n_}; }
};

// A node's offsets: path id -> offset in dictionary order, the first pair inside the node (a sequence of a FASTA input lies on one path: the anchors' surgery made and
// freed a heap block per node it made).  The interface of the vector this replaces, as far as it was used.
struct OffEnt { int first; int64_t second; };
class OffVec {
    uint32_t n_ = 0, cap_ = 1;
    union { OffEnt inl_[1]; OffEnt *heap_; };
    OffEnt *p() { return cap_ > 1 ? heap_ : inl_; }
    const OffEnt *p() const { return cap_ > 1 ? heap_ : inl_; }
    void take(OffVec &o) { n_ = o.n_; cap_ = o.cap_; if (cap_ > 1) heap_ = o.heap_; else inl_[0] = o.inl_[0]; o.n_ = 0; o.cap_ = 1; }
    void copy(const OffVec &o) { n_ = 0; cap_ = 1; reserve(o.n_); for (uint32_t i = 0; i < o.n_; i++) p()[n_++] = o.p()[i]; }
public:
    OffVec() { inl_[0] = OffEnt{0, 0}; }
    ~OffVec() { if (cap_ > 1) free(heap_); }
    OffVec(const OffVec &o) { copy(o); }
    OffVec(OffVec &&o) noexcept { take(o); }
    OffVec &operator=(const OffVec &o) { if (this != &o) { if (cap_ > 1) free(heap_); copy(o); } return *this; }
    OffVec &operator=(OffVec &&o) noexcept { if (this != &o) { if (cap_ > 1) free(heap_); take(o); } return *this; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    void clear() { n_ = 0; }
    void reserve(size_t want) {
        if (want <= cap_) return;
        OffEnt *q = (OffEnt *)malloc(sizeof(OffEnt) * want);
        if (!q) throw std::bad_alloc();
        memcpy(q, p(), sizeof(OffEnt) * n_);
        if (cap_ > 1) free(heap_);
        heap_ = q; cap_ = (uint32_t)want;
    }
    void push_back(const OffEnt &x) { if (n_ == cap_) reserve(cap_ < 4 ? 4 : (size_t)cap_ * 2); p()[n_++] = x; }
    OffEnt &back() { return p()[n_ - 1]; }
    const OffEnt &back() const { return p()[n_ - 1]; }
    OffEnt *begin() { return p(); }
    OffEnt *end() { return p() + n_; }
    const OffEnt *begin() const { return p(); }
    const OffEnt *end() const { return p() + n_; }
};

// A node is two cache lines (128 B, aligned): what a walk reads -- interval, flags, marks and the links it follows forwards -- in the first, the links backwards, the offsets and
// the dictionary position in the second.
struct alignas(64) GNode {
    int64_t b, e;                                   // text interval; sentinels: b = sample, e = 0 (start) / 1 (end)
    int8_t aligned;                                 // -1: sentinel
    int8_t sent = 0;                                // sentinels: 1 a start node, 2 an end node
    bool alive;
    // graphalign's marks, here and not in arrays of their own: the walk looks at a node's `aligned` anyway, so the marks cost no cache line of their own (a level of
    // the last job of config 5 touches 10^7 nodes a dozen times).  Valid while the epoch matches (rv_graph::sub_epoch / walk_epoch): nothing is reset between calls.
    uint8_t cls = 0;                                // bit 0 leading, bit 1 trailing (while ep_sub == sub_epoch)
    uint32_t ep_sub = 0, ep_walk = 0;               // belongs to the sub-index of the current graphalign call / reached by the current walk
    LinkVec succ;                                   // links forwards, in dictionary order (iterating yields edge ids)
    LinkVec pred;                                   // links backwards
    OffVec off;                                     // path id -> offset, in dictionary order
    uint64_t order;                                 // position in the graph's node dictionary (creation order)
};
static_assert(sizeof(GNode) == 128 && offsetof(GNode, succ) + sizeof(LinkVec) <= 64 && offsetof(GNode, pred) + sizeof(LinkVec) <= 128, "GNode layout");
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

// begin -> node for graphalign's look-ups (rv_graph_do_align is handed intervals and needs nodes).  The position map (std::map) answers them with a search from its
// root -- 23 levels of a tree of 10^7 entries, a cache miss each -- and a graph file lists its segments in the order an earlier run MADE them, so the nodes of one
// sub-index lie all over the table: a third of graphalign in the last job of config 5.  Open addressing, one 16-byte slot per entry (one cache line per look-up, and the
// caller prefetches it); nothing is ever taken out: an entry whose node is gone fails the caller's check, a new node with the same begin takes the entry over.
struct BeginHash {
    struct Slot { int64_t key; int val; int pad; };
    std::vector<Slot, HugeAlloc<Slot>> slot;
    size_t mask = 0, used = 0;
    static size_t mix(int64_t b) { uint64_t x = (uint64_t)b * 0x9E3779B97F4A7C15ull; return (size_t)(x ^ (x >> 29)); }
    void clear() { slot.clear(); slot.shrink_to_fit(); mask = used = 0; }
    void reserve(size_t entries) { size_t cap = 1024; while (cap * 3 < entries * 5 + 16) cap *= 2; if (cap > slot.size()) rehash(cap); }
    void rehash(size_t cap) {
        std::vector<Slot, HugeAlloc<Slot>> s2(cap, Slot{-1, -1, 0});
        const size_t m2 = cap - 1;
        for (const Slot &x : slot) if (x.key >= 0) { size_t h = mix(x.key) & m2; while (s2[h].key >= 0) h = (h + 1) & m2; s2[h] = x; }
        slot.swap(s2); mask = m2;
    }
    void put(int64_t b, int id) {
        if ((used + 1) * 5 > slot.size() * 3) rehash(slot.empty() ? 1024 : slot.size() * 2);
        size_t h = mix(b) & mask;
        while (slot[h].key >= 0 && slot[h].key != b) h = (h + 1) & mask;
        if (slot[h].key < 0) { slot[h].key = b; used++; }
        slot[h].val = id;
    }
    const Slot *home(int64_t b) const { return slot.empty() ? nullptr : &slot[mix(b) & mask]; }
    int get(int64_t b) const {
        if (slot.empty()) return -1;
        size_t h = mix(b) & mask;
        while (slot[h].key >= 0) { if (slot[h].key == b) return slot[h].val; h = (h + 1) & mask; }
        return -1;
    }
};

struct rv_graph {
    std::vector<GNode, HugeAlloc<GNode>> nodes;
    std::vector<GEdge, HugeAlloc<GEdge>> edges;
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
    std::vector<uint64_t, HugeAlloc<uint64_t>> begbits;                  // one bit per text position: a node begins here (kept with `made`; the predecessor search of fast_node_at)
    BeginHash made; bool made_on = false;           // graphalign: begin -> node, every sequence node from its first large call on (new_node keeps it up)
    std::vector<int> look_tmp;      // graphalign: begin -> node of the nodes that were there when the run began, sorted (see rv_graph_do_align)
    void *align_out_ = nullptr;                     // rv_graphrem.hip: the result of the last rv_graph_do_align through the C ABI
    void *align_out();
    ~rv_graph();

    int new_node(int64_t b, int64_t e, int8_t aligned) {
        GNode n;
        n.b = b; n.e = e; n.aligned = aligned; n.alive = true; n.order = counter++;
        nodes.push_back(std::move(n));
        const int id = (int)nodes.size() - 1;
        if (aligned >= 0) { if (use_map) at[b] = id; if (made_on) { made.put(b, id); mark_begin(b); } }
        return id;
    }
    // alngraph.py add_edge: one edge per (u, v); adding it again unites the path sets
    void add_edge(int u, int v, const PathSet &paths) {
        { const LinkVec &sv = nodes[(size_t)u].succ; const Link *lk = sv.links();
          for (size_t k = 0; k < sv.size(); k++) if (lk[k].to == v) { edges[(size_t)lk[k].e].paths.unite(paths); return; } }
        edges.push_back({u, v, paths});
        const int e = (int)edges.size() - 1;
        nodes[(size_t)u].succ.push_back(e, v);
        nodes[(size_t)v].pred.push_back(e, u);
    }
    // The position map costs a tree insertion per node made and an erasure per node broken.  The surgery of a finished run's anchors (rv_graph_replay, the follower of
    // rv_set_replay_graph) needs none of it: an anchor's member always lies in a live node, whose begin is the nearest set bit of the bitmap in front of it (a dead node's begin
    // never lies inside a live node's interval), and the hash names the node.  use_map = false while that surgery runs; compact() makes the map again.
    bool use_map = true;
    bool has_dead = false;                          // a node or a link has gone since the last renumbering (compact() has work to do)
    void remove_node(int x) {
        has_dead = true;
        GNode &n = nodes[(size_t)x];
        for (int e : n.succ) { nodes[(size_t)edges[(size_t)e].v].pred.remove(e); edges[(size_t)e].u = -1; }
        for (int e : n.pred) { nodes[(size_t)edges[(size_t)e].u].succ.remove(e); edges[(size_t)e].u = -1; }
        n.succ.clear(); n.pred.clear(); n.off.clear(); n.alive = false;
        if (use_map) {
            auto it = at.find(n.b);
            if (n.aligned >= 0 && it != at.end() && it->second == x) at.erase(it);
        }
    }
    void mark_begin(int64_t b) {
        const size_t w = (size_t)(b >> 6);
        if (w >= begbits.size()) begbits.resize(w + w / 4 + 1024, 0);
        begbits[w] |= 1ull << (b & 63);
    }
    // node_at without the search from the root of the position map: the nearest begin at or in front of pos from the bitmap (nodes are tens of positions long: the same
    // word, or the one before), its node from the hash; anything that does not check out -- a begin whose node is gone -- goes to node_at
    int fast_node_at(int64_t pos) {
        if (!made_on || pos < 0) return node_at(pos);
        size_t w = (size_t)(pos >> 6);
        uint64_t bits;
        if (w >= begbits.size()) { if (begbits.empty()) return node_at(pos); w = begbits.size() - 1; bits = begbits[w]; }
        else bits = begbits[w] & (~0ull >> (63 - (pos & 63)));
        for (int guard = 0; !bits; guard++) { if (w == 0 || (use_map && guard > 64)) return node_at(pos); bits = begbits[--w]; }      // (without the map: as far back as the node is long)
        const int64_t b = (int64_t)(w << 6) + 63 - __builtin_clzll(bits);
        const int id = made.get(b);
        if (id >= 0) { const GNode &n = nodes[(size_t)id]; if (n.alive && n.aligned >= 0 && n.b == b && pos < n.e) return id; }
        return node_at(pos);
    }
    int node_at(int64_t pos) {
        if (!use_map) return -1;      // (the surgery of a finished run: the bitmap and the hash answer every look-up of a valid anchor, see use_map)
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
        const OffVec att = std::move(nodes[(size_t)x].off);      // (the node is about to go)
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
        if (use_map) { auto it = at.find(nb); if (it != at.end() && it->second == x) at.erase(it); }
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
        OffVec merged;      // an ordered mapping path -> offset: a later node's value for a path that is there replaces it in place
        for (int x : mns)
            for (auto &a : nodes[(size_t)x].off) {
                if ((size_t)a.first >= pmark.size()) { pmark.resize((size_t)a.first + 64, 0); pwhere.resize(pmark.size(), 0); }
                if (pwhere.size() < pmark.size()) pwhere.resize(pmark.size(), 0);
                if (pmark[(size_t)a.first]) merged.begin()[(size_t)pwhere[(size_t)a.first]].second = a.second;
                else { pmark[(size_t)a.first] = 1; pwhere[(size_t)a.first] = (int32_t)merged.size(); merged.push_back(a); }
            }
        for (auto &a : merged) pmark[(size_t)a.first] = 0;
        nodes[(size_t)ref].off = std::move(merged);
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
        made.clear(); made_on = false; begbits.clear();
        if (!use_map) {      // the map of the live sequence nodes, made in one go
            use_map = true;
            std::vector<std::pair<int64_t, int>> by_b;
            for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].alive && nodes[i].aligned >= 0) by_b.push_back({nodes[i].b, (int)i});
            std::sort(by_b.begin(), by_b.end());
            at.clear();
            for (auto &kv : by_b) at.emplace_hint(at.end(), kv.first, kv.second);
        }
        if (!has_dead) return;
        has_dead = false;
        std::vector<int> nmap(nodes.size(), -1), emap(edges.size(), -1);
        size_t nn = 0;
        for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].alive) nmap[i] = (int)nn++;
        size_t ne = 0;
        for (size_t e = 0; e < edges.size(); e++) if (edges[e].u >= 0 && nodes[(size_t)edges[e].u].alive && nodes[(size_t)edges[e].v].alive) emap[e] = (int)ne++;
        std::vector<GNode, HugeAlloc<GNode>> n2; n2.reserve(nn);
        for (size_t i = 0; i < nodes.size(); i++) {
            if (!nodes[i].alive) continue;
            GNode &n = nodes[i];
            for (int &e : n.succ) e = emap[(size_t)e];
            for (int &e : n.pred) e = emap[(size_t)e];
            for (size_t k = 0; k < n.succ.size(); k++) n.succ.links()[k].to = nmap[(size_t)n.succ.links()[k].to];
            for (size_t k = 0; k < n.pred.size(); k++) n.pred.links()[k].to = nmap[(size_t)n.pred.links()[k].to];
            n2.push_back(std::move(n));
        }
        std::vector<GEdge, HugeAlloc<GEdge>> e2; e2.reserve(ne);
        for (size_t e = 0; e < edges.size(); e++) if (emap[e] >= 0) { GEdge &x = edges[e]; x.u = nmap[(size_t)x.u]; x.v = nmap[(size_t)x.v]; e2.push_back(std::move(x)); }
        nodes.swap(n2); edges.swap(e2);
        for (auto &kv : at) kv.second = nmap[(size_t)kv.second];
        for (int &x : start_of) x = nmap[(size_t)x];
    }
    void finish() {
        order.clear();
        for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].alive) order.push_back((int)i);      // (a node's number IS its creation order: new_node hands both out together)
        edge_no.assign(edges.size(), -1);
        int ne = 0;
        for (int x : order) for (int e : nodes[(size_t)x].succ) edge_no[(size_t)e] = ne++;
    }
};


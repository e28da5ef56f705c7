// rv_gfaread.hip -- the input side of `reveal rem` for graphs behind the ABI (host code): the graph the reference's readers leave (reveal/utils.py:304-375
// read_fasta, :377-677 read_gfa with the defaults `reveal rem` uses) built straight into the structure of rv_graph.h, the segments' text appended to the index
// on the way.  reveal_amd/alngraph.py read_fasta / read_gfa are the Python forms (tests/test_cpu_graphrem_native.py reads every fixture both ways and compares
// node for node in dictionary order); a level-1 job of config 5 at 25 x 1 Mbp spent 9.3 of its 14.8 s there and another 1.8 s moving the result behind the ABI.
//
// What the reference's reader does, in its order: every S line becomes a '$'-terminated sequence of the current sample and an unaligned node; L lines become
// links with empty path sets; every P line numbers a new path, walks its steps -- a node's offset on the path, the link from the step before gets the path -- and
// hangs the walk between a start and an end sentinel of its own; links and nodes no path uses go; per weakly connected component of what the file added, the end
// sentinels of its paths are merged into one, then the start sentinels.  The reference picks components and merges sentinels in the iteration order of Python
// sets of random names; here (and in alngraph.read_gfa) both go by creation order, one of the orders the reference can take.
// Links on the reverse strand are not supported behind the ABI: -2 comes back and the caller takes the Python route.
#include "rv_graph.h"
#include <chrono>
#include <cstdlib>
#include <string_view>
#include <unordered_map>

// what a file parsed on its own (rv_gfa_parse: any thread, no shared state) leaves for rv_graph_adopt: its graph with intervals counted from 0 and path ids from 0,
// the text its segments add to the index ('$' behind every sequence), the names of its paths
struct GfaParsed {
    rv_graph frag;
    std::string text; std::vector<int64_t> seq_len;
    std::string names;
    int64_t npaths = 0;      // -1 error (err), -2 links on the reverse strand
    std::string err;
};

namespace {

inline int new_sentinel(rv_graph *g, int kind) {
    const int x = g->new_node((int64_t)g->counter, 0, -1);
    g->nodes[(size_t)x].sent = (int8_t)kind;
    return x;
}
inline void drop_edge(rv_graph *g, int e) {
    GEdge &ed = g->edges[(size_t)e];
    g->nodes[(size_t)ed.u].succ.remove(e);
    g->nodes[(size_t)ed.v].pred.remove(e);
    ed.u = -1;
    g->has_dead = true;
}
// alngraph.check_segment_shortcut: every sequence node goes on over a link carried by a real path, in both directions -- or segmentgraph takes the reference's form
void check_shortcut(rv_graph *g) {
    bool any_star = false;
    for (uint8_t s : g->star) any_star |= s != 0;
    auto real = [&](const PathSet &p) {
        if (!any_star) return p.size() > 0;
        bool r = false;
        p.each([&](int sid) { r |= !g->star[(size_t)sid]; });
        return r;
    };
    for (const GNode &n : g->nodes) {
        if (!n.alive || n.aligned < 0) continue;
        bool f = false, b = false;
        for (int e : n.succ) if (real(g->edges[(size_t)e].paths)) { f = true; break; }
        for (int e : n.pred) if (real(g->edges[(size_t)e].paths)) { b = true; break; }
        if (!f || !b) { g->literal_segments = true; return; }
    }
}

struct Fields {      // the first columns of a tab-separated line
    std::string_view f[6]; int n = 0;
    explicit Fields(std::string_view line, int want = 6) {
        size_t at = 0;
        while (n < want && at <= line.size()) {
            size_t t = line.find('\t', at);
            if (t == std::string_view::npos) t = line.size();
            f[n++] = line.substr(at, t - at);
            at = t + 1;
        }
    }
};

inline double rnow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int64_t read_gfa(rv_graph *g, rv_index *h, int64_t *text_n, const char *data, int64_t len, GfaParsed *sink = nullptr) {
    const bool times = getenv("RV_GRAPH_TIMES") != nullptr;
    double tq = rnow(), tph[6] = {0, 0, 0, 0, 0, 0};
    auto phase = [&](int k) { const double t = rnow(); tph[k] += t - tq; tq = t; };
    // segment names: the numbers 1 .. N in files reveal writes -- a table then, not a hash look-up per step of every path (10^8 steps in the last job of config 5);
    // any other name moves all of them into the map
    std::unordered_map<std::string_view, int> nmap;
    std::vector<int> by_number;
    std::vector<std::string_view> snames;      // (kept while the table serves, for the move)
    bool numbered = true;
    auto number_of = [](std::string_view id, int64_t limit) -> int64_t {
        if (id.empty() || id.size() > 10 || (id.size() > 1 && id[0] == '0')) return -1;
        int64_t v = 0;
        for (char ch : id) { if (ch < '0' || ch > '9') return -1; v = v * 10 + (ch - '0'); }
        return v < limit ? v : -1;
    };
    std::vector<std::string_view> llines, plines;
    const size_t first_node = g->nodes.size();
    std::string up;
    g->names_buf.clear();
    // S lines: text and nodes
    {
        size_t lines = 0;
        for (int64_t i = 0; i < len; i++) lines += data[i] == '\n';
        by_number.assign(lines + 2, -1);
        g->nodes.reserve(g->nodes.size() + lines + 64);
    }
    const int64_t number_limit = (int64_t)by_number.size();
    for (int64_t at = 0; at < len;) {
        const char *nl = (const char *)memchr(data + at, '\n', (size_t)(len - at));
        const int64_t end = nl ? nl - data : len;
        std::string_view line(data + at, (size_t)(end - at));
        at = end + 1;
        if (line.empty()) continue;
        if (line[0] == 'S') {
            Fields c(line, 4);
            if (c.n < 2) { rv_set_error("read_gfa: an S line without a name"); return -1; }
            std::string_view seq = c.n > 2 ? c.f[2] : std::string_view();
            up.assign(seq.data(), seq.size());
            for (char &ch : up) if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
            int64_t b = 0, e = 0;
            if (h) { if (rv_add_sequence(h, up.data(), (int64_t)up.size(), &b, &e) != 0) return -1; }
            else {
                b = *text_n; e = b + (int64_t)up.size(); *text_n = e + 1;
                if (sink) { sink->text.append(up); sink->text.push_back('$'); sink->seq_len.push_back((int64_t)up.size()); }
            }
            const int x = g->new_node(b, e, 0);
            if (numbered) {
                const int64_t num = number_of(c.f[1], number_limit);
                if (num >= 0) { by_number[(size_t)num] = x; snames.push_back(c.f[1]); }
                else {
                    numbered = false;
                    nmap.reserve(by_number.size());
                    for (std::string_view nm : snames) nmap[nm] = by_number[(size_t)number_of(nm, number_limit)];
                    snames.clear(); snames.shrink_to_fit();
                }
            }
            if (!numbered) nmap[c.f[1]] = x;
        } else if (line[0] == 'L') llines.push_back(line);
        else if (line[0] == 'P') plines.push_back(line);
    }
    phase(0);
    auto node_named = [&](std::string_view id) -> int {
        if (numbered) {
            const int64_t num = number_of(id, number_limit);
            if (num >= 0 && by_number[(size_t)num] >= 0) return by_number[(size_t)num];
            rv_set_error("read_gfa: no segment named '%.*s'", (int)std::min<size_t>(id.size(), 60), id.data()); return -1;
        }
        auto it = nmap.find(id);
        if (it == nmap.end()) { rv_set_error("read_gfa: no segment named '%.*s'", (int)std::min<size_t>(id.size(), 60), id.data()); return -1; }
        return it->second;
    };
    for (std::string_view line : llines) {
        Fields c(line, 6);
        if (c.n < 5) { rv_set_error("read_gfa: an L line with fewer than five columns"); return -1; }
        if (c.f[2] != "+" || c.f[4] != "+") return -2;
        const int u = node_named(c.f[1]), v = node_named(c.f[3]);
        if (u < 0 || v < 0) return -1;
        g->add_edge(u, v, PathSet());
    }
    phase(1);
    if (plines.empty()) { rv_set_error("no paths defined in the GFA input"); return -1; }
    std::vector<int> starts, ends;
    int64_t added = 0;
    for (std::string_view line : plines) {
        Fields c(line, 4);
        if (c.n < 2) { rv_set_error("read_gfa: a P line without a name"); return -1; }
        const int sid = (int)g->id2end.size();
        g->names_buf.append(c.f[1].data(), c.f[1].size()); g->names_buf += '\n';
        g->star.push_back(!c.f[1].empty() && c.f[1][0] == '*');
        int64_t o = 0;
        int prev = -1, first = -1;
        std::string_view steps = c.n > 2 ? c.f[2] : std::string_view();
        for (size_t at = 0; !steps.empty() && at <= steps.size();) {
            size_t cm = steps.find(',', at);
            if (cm == std::string_view::npos) cm = steps.size();
            std::string_view st = steps.substr(at, cm - at);
            at = cm + 1;
            if (st.empty()) { rv_set_error("read_gfa: an empty step in path %.*s", (int)std::min<size_t>(c.f[1].size(), 60), c.f[1].data()); return -1; }
            if (st.back() != '+') return -2;
            const int node = node_named(st.substr(0, st.size() - 1));
            if (node < 0) return -1;
            GNode &n = g->nodes[(size_t)node];
            if (!n.off.empty() && n.off.back().first == sid) n.off.back().second = o;      // (a path through a node twice: the later offset stands, in the first one's place)
            else n.off.push_back({sid, o});
            o += n.e - n.b;
            if (prev >= 0) {
                int hit = -1;
                { const LinkVec &sv = g->nodes[(size_t)prev].succ; const Link *lk = sv.links(); for (size_t k = 0; k < sv.size(); k++) if (lk[k].to == node) { hit = lk[k].e; break; } }
                if (hit < 0) { rv_set_error("path %.*s steps over a link the graph does not have", (int)std::min<size_t>(c.f[1].size(), 60), c.f[1].data()); return -1; }
                g->edges[(size_t)hit].paths.add(sid);
            } else first = node;
            prev = node;
        }
        const int start = new_sentinel(g, 1), end = new_sentinel(g, 2);
        g->nodes[(size_t)start].off.push_back({sid, 0});
        g->nodes[(size_t)end].off.push_back({sid, o});
        if (first >= 0) {
            PathSet one; one.add(sid);
            g->add_edge(start, first, one);
            g->add_edge(prev, end, one);
        }
        starts.push_back(start); ends.push_back(end);
        g->id2end.push_back(o);
        added++;
    }
    phase(2);
    // links, then nodes, no path uses
    const size_t file_end = g->nodes.size();
    std::vector<int> dead;
    for (size_t x = first_node; x < file_end; x++) {
        dead.clear();
        for (int e : g->nodes[x].succ) if (g->edges[(size_t)e].paths.size() == 0) dead.push_back(e);
        for (int e : dead) drop_edge(g, e);
    }
    for (size_t x = first_node; x < file_end; x++) if (g->nodes[x].alive && g->nodes[x].aligned >= 0 && g->nodes[x].off.empty()) g->remove_node((int)x);
    // weakly connected components of what the file added, each by its first member in creation order; one end and one start sentinel per component
    std::vector<int> comp_of(file_end - first_node, -1);
    std::vector<std::vector<int>> comp_ends, comp_starts;
    std::vector<int> stack;
    for (size_t x0 = first_node; x0 < file_end; x0++) {
        if (!g->nodes[x0].alive || comp_of[x0 - first_node] >= 0) continue;
        const int cid = (int)comp_ends.size();
        comp_ends.emplace_back(); comp_starts.emplace_back();
        stack.assign(1, (int)x0);
        comp_of[x0 - first_node] = cid;
        while (!stack.empty()) {
            const int x = stack.back(); stack.pop_back();
            const GNode &n = g->nodes[(size_t)x];
            for (int e : n.succ) { const int v = g->edges[(size_t)e].v; if (comp_of[(size_t)v - first_node] < 0) { comp_of[(size_t)v - first_node] = cid; stack.push_back(v); } }
            for (int e : n.pred) { const int u = g->edges[(size_t)e].u; if (comp_of[(size_t)u - first_node] < 0) { comp_of[(size_t)u - first_node] = cid; stack.push_back(u); } }
        }
    }
    for (int x : ends) comp_ends[(size_t)comp_of[(size_t)x - first_node]].push_back(x);           // (creation order)
    for (int x : starts) comp_starts[(size_t)comp_of[(size_t)x - first_node]].push_back(x);
    std::vector<std::pair<int, PathSet>> links;
    for (size_t cidx = 0; cidx < comp_ends.size(); cidx++) {
        for (int forward = 0; forward < 2; forward++) {
            const std::vector<int> &group = forward ? comp_starts[cidx] : comp_ends[cidx];
            if (group.empty()) continue;
            const int sentinel = new_sentinel(g, forward ? 1 : 2);
            if (forward) g->start_of.push_back(sentinel);
            for (int old : group) {
                for (auto &a : g->nodes[(size_t)old].off) {
                    bool found = false;
                    for (auto &m : g->nodes[(size_t)sentinel].off) if (m.first == a.first) { m.second = a.second; found = true; break; }
                    if (!found) g->nodes[(size_t)sentinel].off.push_back(a);
                }
                links.clear();
                if (forward) { for (int e : g->nodes[(size_t)old].succ) links.push_back({g->edges[(size_t)e].v, g->edges[(size_t)e].paths}); }
                else for (int e : g->nodes[(size_t)old].pred) links.push_back({g->edges[(size_t)e].u, g->edges[(size_t)e].paths});
                for (auto &lk : links) { if (forward) g->add_edge(sentinel, lk.first, lk.second); else g->add_edge(lk.first, sentinel, lk.second); }
                g->remove_node(old);
            }
        }
    }
    phase(3);
    // (the per-path sentinels and unused segments stay behind as dead entries until rv_graph_seal renumbers the graph once, after the last file)
    phase(4);
    if (times) fprintf(stderr, "read_gfa: segments %.3f s, links %.3f, paths %.3f, unused + components %.3f\n", tph[0], tph[1], tph[2], tph[3]);
    return added;
}

}  // namespace

extern "C" {

rv_graph *rv_graph_new(void) {
    try { return new rv_graph(); } catch (...) { rv_set_error("rv_graph_new: out of host memory"); return nullptr; }
}

/* utils.py:304-375 read_fasta for one sequence the caller has added to the index as [b, e): a new path, its start sentinel, the node, its end sentinel -> path id */
int rv_graph_add_linear(rv_graph *g, int64_t b, int64_t e, int star) {
    try {
        const int sid = (int)g->id2end.size();
        g->star.push_back(star ? 1 : 0); g->id2end.push_back(e - b);
        const int st = new_sentinel(g, 1), x = g->new_node(b, e, 0), en = new_sentinel(g, 2);
        g->nodes[(size_t)st].off.push_back({sid, 0}); g->nodes[(size_t)x].off.push_back({sid, 0}); g->nodes[(size_t)en].off.push_back({sid, e - b});
        PathSet one; one.add(sid);
        g->add_edge(st, x, one); g->add_edge(x, en, one);
        g->start_of.push_back(st);
        return sid;
    } catch (...) { rv_set_error("rv_graph_add_linear: out of host memory"); return -1; }
}

/* utils.py:377-677 read_gfa on the text of a GFA1 file (data, len): segments appended to the index h as sequences of its current sample (h == NULL: intervals
 * counted from *text_n on, which moves -- tests without a device), the graph added to g.  -> number of paths added (their names, one per line: *names, valid
 * until the next call), -1 error, -2 the file holds links on the reverse strand (g and h are then in an undefined state: start over on the Python route) */
int64_t rv_graph_read_gfa(rv_graph *g, rv_index *h, int64_t *text_n, const char *data, int64_t len, const char **names) {
    try {
        if (!g || !data || len < 0 || (!h && !text_n)) { rv_set_error("rv_graph_read_gfa: bad arguments"); return -1; }
        const int64_t r = read_gfa(g, h, text_n, data, len);
        if (names) *names = g->names_buf.c_str();
        return r;
    } catch (const std::exception &e) { rv_set_error("rv_graph_read_gfa: %s", e.what()); return -1; }
    catch (...) { rv_set_error("rv_graph_read_gfa: failed"); return -1; }
}

/* A file parsed by itself -- rv_gfa_parse touches no graph and no index, so the files of a job are parsed side by side on the caller's threads (the last job of config 5 reads
 * four files of 2.4 x 10^6 segments and 25 paths of 10^6 steps: 3.3 s each, one after the other) -- and rv_graph_adopt, in the order of the inputs, appends its text to the index
 * (one copy) and its graph to g: node and edge numbers, intervals and path ids moved behind what is there.  The result is the graph rv_graph_read_gfa makes. */
GfaParsed *rv_gfa_parse(const char *data, int64_t len) {
    GfaParsed *P = nullptr;
    try {
        P = new GfaParsed();
        if (!data || len < 0) { P->npaths = -1; P->err = "rv_gfa_parse: bad arguments"; return P; }
        {   // one allocation of the text: the file is never shorter
            P->text.reserve((size_t)len / 2 + 64);
        }
        int64_t tn = 0;
        const int64_t r = read_gfa(&P->frag, nullptr, &tn, data, len, P);
        P->npaths = r;
        if (r == -1) P->err = rv_last_error();
        if (r >= 0) P->names = P->frag.names_buf;      // (its dead entries -- the paths' own sentinels, unused segments -- are left out when it is adopted)
        return P;
    } catch (const std::exception &e) { if (P) { P->npaths = -1; P->err = std::string("rv_gfa_parse: ") + e.what(); } else rv_set_error("rv_gfa_parse: out of host memory"); return P; }
    catch (...) { if (P) { P->npaths = -1; P->err = "rv_gfa_parse failed"; } else rv_set_error("rv_gfa_parse: out of host memory"); return P; }
}
void rv_gfa_parsed_free(GfaParsed *P) { delete P; }

int64_t rv_graph_adopt(rv_graph *g, rv_index *h, int64_t *text_n, GfaParsed *P, const char **names) {
    try {
        if (!g || !P || (!h && !text_n)) { rv_set_error("rv_graph_adopt: bad arguments"); return -1; }
        if (P->npaths == -2) return -2;
        if (P->npaths < 0) { rv_set_error("%s", P->err.c_str()); return -1; }
        int64_t text_base = 0;
        if (h) {
            text_base = rv_n(h);
            if (rv_add_sequences(h, P->text.data(), (int64_t)P->text.size(), P->seq_len.data(), (int64_t)P->seq_len.size()) != 0) return -1;
        } else { text_base = *text_n; *text_n += (int64_t)P->text.size(); }
        rv_graph &f = P->frag;
        const int sid_base = (int)g->id2end.size();
        // live nodes and links get their numbers behind what is there, in their order; the rest stays behind
        std::vector<int> nmap(f.nodes.size(), -1), emap(f.edges.size(), -1);
        int nn = (int)g->nodes.size(), ne = (int)g->edges.size();
        for (size_t i = 0; i < f.nodes.size(); i++) if (f.nodes[i].alive) nmap[i] = nn++;
        for (size_t e = 0; e < f.edges.size(); e++) { const GEdge &ed = f.edges[e]; if (ed.u >= 0 && f.nodes[(size_t)ed.u].alive && f.nodes[(size_t)ed.v].alive) emap[e] = ne++; }
        g->nodes.reserve((size_t)nn); g->edges.reserve((size_t)ne);
        for (size_t e = 0; e < f.edges.size(); e++) {
            if (emap[e] < 0) continue;
            GEdge &ed = f.edges[e];
            GEdge x; x.u = nmap[(size_t)ed.u]; x.v = nmap[(size_t)ed.v];
            if (sid_base == 0) x.paths = std::move(ed.paths); else ed.paths.each([&](int q) { x.paths.add(q + sid_base); });
            g->edges.push_back(std::move(x));
        }
        for (size_t i = 0; i < f.nodes.size(); i++) {
            if (nmap[i] < 0) continue;
            GNode &n = f.nodes[i];
            if (n.aligned >= 0) { n.b += text_base; n.e += text_base; } else n.b = (int64_t)g->counter;
            n.order = g->counter++;
            n.ep_sub = n.ep_walk = 0; n.cls = 0;
            for (auto &a : n.off) a.first += sid_base;
            for (size_t k = 0; k < n.succ.size(); k++) { Link &lk = n.succ.links()[k]; lk.e = emap[(size_t)lk.e]; lk.to = nmap[(size_t)lk.to]; }
            for (size_t k = 0; k < n.pred.size(); k++) { Link &lk = n.pred.links()[k]; lk.e = emap[(size_t)lk.e]; lk.to = nmap[(size_t)lk.to]; }
            g->nodes.push_back(std::move(n));
            const int id = (int)g->nodes.size() - 1;
            const GNode &m = g->nodes[(size_t)id];
            if (m.aligned >= 0) { g->at.emplace_hint(g->at.end(), m.b, id); if (g->made_on) { g->made.put(m.b, id); g->mark_begin(m.b); } }      // (begins grow with the text)
        }
        const int node_base = 0;      // (start_of below goes through nmap)
        (void)node_base;
        for (int x : f.start_of) if (nmap[(size_t)x] >= 0) g->start_of.push_back(nmap[(size_t)x]);
        g->star.insert(g->star.end(), f.star.begin(), f.star.end());
        g->id2end.insert(g->id2end.end(), f.id2end.begin(), f.id2end.end());
        g->literal_segments = g->literal_segments || f.literal_segments;
        g->names_buf = P->names;
        if (names) *names = g->names_buf.c_str();
        const int64_t r = P->npaths;
        f.nodes.clear(); f.edges.clear(); f.at.clear();
        return r;
    } catch (const std::exception &e) { rv_set_error("rv_graph_adopt: %s", e.what()); return -1; }
    catch (...) { rv_set_error("rv_graph_adopt: failed"); return -1; }
}

/* after the last input: dead entries go, live nodes and links are renumbered in their order, and alngraph.check_segment_shortcut's question is asked once for the
 * whole graph (the reference asks it after every file, of every node read so far: the same answer) */
int rv_graph_seal(rv_graph *g) {
    try { g->compact(); check_shortcut(g); g->finish(); return 0; }
    catch (...) { rv_set_error("rv_graph_seal: out of host memory"); return -1; }
}

/* the paths of a graph made by the two readers: -> their number; id2end (may be NULL): their lengths */
int rv_graph_paths(const rv_graph *g, int64_t *id2end) {
    if (id2end) for (size_t k = 0; k < g->id2end.size(); k++) id2end[k] = g->id2end[k];
    return (int)g->id2end.size();
}

int rv_graph_literal(const rv_graph *g) { return g->literal_segments ? 1 : 0; }      /* segmentgraph takes the reference's literal form (alngraph.check_segment_shortcut said no) */

/* per node in the order of rv_graph_export (after rv_graph_finish): 0 a sequence node, 1 a start sentinel, 2 an end sentinel */
int rv_graph_node_kinds(const rv_graph *g, int8_t *out) {
    size_t k = 0;
    for (int x : g->order) out[k++] = g->nodes[(size_t)x].aligned < 0 ? g->nodes[(size_t)x].sent : 0;
    return 0;
}

}

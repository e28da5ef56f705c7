// rv_chain.hip -- host-side C++ of the anchor picker behind the C ABI ("move chain() to C++ since it
// dominates wall time").  No device code: the chaining DP of the reference's Python picker runs on a few hundred to a
// thousand pre-selected matches per call (schemes.py:287-289 caps them at --maxmums) between two GPU levels.
//
// rv_chain restates reveal/schemes.py:20-105 `chain()` + reveal/utils.py:162-183 `gapcost()` decision for decision --
// including what looks incidental there but decides ties: the stable sort of the input by the first path's coordinate, the
// `active` list kept in arrival order and stably re-sorted by score before every match, the early exit once the best
// possible score of the remaining predecessors is below the best found, strict `>` when two predecessors tie, and the
// dictionaries keyed by the first path's coordinate.  Pinned against the reference's own function on random and
// adversarial inputs: tests/golden/chain_vectors.json (oracle/gen_chain_golden.py), tests/test_cpu_chain.py.
#include "../../include/reveal_amd.h"
#include "rv_common.h"
#include <algorithm>
#include <unordered_map>
#include <vector>

namespace {

// utils.gapcost (utils.py:162-183); a = end of the predecessor, b = start of the match, per path
int64_t gapcost(const int64_t *a, const int64_t *b, int k, int model, std::vector<int64_t> &D) {
    D.resize((size_t)k);
    if (model == 1) {            // star-avg: abs(sum(a_i - b_i)) / k   (integer division)
        int64_t s = 0;
        for (int i = 0; i < k; i++) s += a[i] - b[i];
        return (s < 0 ? -s : s) / k;
    }
    for (int i = 0; i < k; i++) { const int64_t d = a[i] - b[i]; D[(size_t)i] = d < 0 ? -d : d; }
    if (model == 2) {            // star-med: sorted(|a_i - b_i|)[k / 2]
        std::sort(D.begin(), D.end());
        return D[(size_t)(k / 2)];
    }
    // sumofpairs: all pairwise differences of the per-path gaps.  Sorted ascending, D[i] is added i times and subtracted
    // k-1-i times: sum_{i<j} |D_i - D_j| = sum_i (2i - k + 1) D_(i) -- the same integer in O(k log k) instead of O(k^2)
    // (a merge of four graphs of 25 genomes chains over 100 paths: the quadratic form was the whole job)
    if (k <= 8) {
        int64_t p = 0;
        for (int i = 0; i < k; i++)
            for (int j = i + 1; j < k; j++) { const int64_t d = D[(size_t)i] - D[(size_t)j]; p += d < 0 ? -d : d; }
        return p;
    }
    std::sort(D.begin(), D.end());
    int64_t p = 0;
    for (int i = 0; i < k; i++) p += (int64_t)(2 * i - k + 1) * D[(size_t)i];
    return p;
}

}  // namespace

extern "C" int64_t rv_chain(int64_t m, int k, const uint32_t *len, const int32_t *nmem, const int64_t *crd, const int64_t *left,
                            const int64_t *right, int64_t wscore, int64_t wpen, int model, int64_t *out_idx, int64_t *out_score) {
    if (m < 0 || k < 1 || model < 0 || model > 2) { rv_set_error("rv_chain: bad arguments"); return -1; }
    if (m == 0) return 0;                                       // schemes.py:21-22
    // elements 0..m-1 = the matches, m = `right` (appended before the sort, schemes.py:29), m+1 = `left`
    const int64_t R = m, L = m + 1;
    auto C = [&](int64_t e, int j) -> int64_t { return e == R ? right[j] : e == L ? left[j] : crd[(size_t)e * k + j]; };
    auto LEN = [&](int64_t e) -> int64_t { return e >= m ? 0 : (int64_t)len[e]; };
    std::vector<int64_t> order((size_t)m + 1);
    for (int64_t i = 0; i <= m; i++) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return C(a, 0) < C(b, 0); });      // schemes.py:30
    std::unordered_map<int64_t, int64_t> sp2mum, link, score;   // all keyed by the coordinate on the first path, as in the reference
    for (int64_t e : order) sp2mum[C(e, 0)] = e;                // schemes.py:32-34 (a later equal coordinate replaces the earlier)
    score[left[0]] = 0;
    std::vector<int64_t> active{L}, processed, keep, D;
    std::vector<int64_t> ea((size_t)k), sb((size_t)k);
    int64_t best = -1;
    auto ends_before = [&](int64_t a, int64_t b) -> bool {      // a ends at or in front of b's start on every path (schemes.py:49-51, 63-65)
        for (int j = 0; j < k; j++) if (C(a, j) + LEN(a) > C(b, j)) return false;
        return true;
    };
    for (int64_t e : order) {
        keep.clear();
        for (int64_t p : processed) { if (ends_before(p, e)) active.push_back(p); else keep.push_back(p); }      // schemes.py:47-57
        processed.swap(keep);
        std::stable_sort(active.begin(), active.end(), [&](int64_t a, int64_t b) { return score[C(a, 0)] > score[C(b, 0)]; });      // :59
        bool have = false;
        int64_t w = 0;
        const int64_t n = e >= m ? 0 : (int64_t)nmem[e];
        const int64_t gain = wscore * (LEN(e) * ((n * (n - 1)) / 2));
        for (int j = 0; j < k; j++) sb[(size_t)j] = C(e, j);
        for (int64_t a : active) {
            if (!ends_before(a, e)) continue;
            const int64_t s = score[C(a, 0)] + gain;
            if (have && w > s) break;                           // sorted by score: nothing better can follow (schemes.py:70-72)
            for (int j = 0; j < k; j++) ea[(size_t)j] = C(a, j) + LEN(a);
            const int64_t tmpw = s - wpen * gapcost(ea.data(), sb.data(), k, model, D);
            if (!have || tmpw > w) { w = tmpw; best = a; have = true; }
        }
        if (best < 0 || !have) { rv_set_error("rv_chain: a match has no predecessor (it does not lie behind `left` on every path)"); return -1; }
        link[C(e, 0)] = C(best, 0);
        score[C(e, 0)] = w;
        processed.push_back(e);
    }
    // backtrack from `right` (schemes.py:97-103); the chain is handed out left to right, `right` itself left out
    std::vector<std::pair<int64_t, int64_t>> path;
    int64_t end = right[0];
    const int64_t start = left[0];
    int64_t guard = 0;
    while (end != start) {
        auto it = sp2mum.find(end);
        if (it == sp2mum.end() || ++guard > m + 2) { rv_set_error("rv_chain: broken back-pointer chain"); return -1; }
        path.push_back({it->second, score[end]});
        end = link[end];
    }
    int64_t cnt = 0;
    for (size_t q = path.size(); q-- > 1;) {                    // path[0] is `right`
        if (path[q].first >= m) { rv_set_error("rv_chain: sentinel inside the chain"); return -1; }
        out_idx[cnt] = path[q].first; out_score[cnt] = path[q].second; cnt++;
    }
    return cnt;
}

// ---- the reference's default picker for one sub-index (reveal/schemes.py:197-361 `graphmumpicker`, not-precomputed branch) --------------
// FASTA inputs with one sequence per sample: a position's graph node is its own sequence, its path offset pos - seq_begin[sample], and the
// sentinels `left` / `right` of the chain are the ends of the sub-index' interval of every path (the left / right graph nodes of the reference
// stand right in front of / behind them: schemes.py:252-274).  What looks incidental there but decides results is kept: the stable sorts, the
// list filter of trim_overlap that looks at the LAST element for the first (Python's index -1), the dictionary keyed by the offsets that lets a
// later match replace an earlier one, the chain's tie rules (rv_chain), "the largest of the chain" = the last of equal lengths.
namespace {
struct PkMum { int64_t l; int32_t n; std::vector<uint16_t> so; std::vector<int64_t> pos; };

// schemes.py:160-193; -> false: the reference's own code would raise here (trimmed[-1] of an empty list)
bool pk_trim_overlap(std::vector<PkMum> &mums) {
    if (mums.empty()) return true;
    const size_t ncoord = mums[0].pos.size();
    for (size_t c = 0; c < ncoord; c++) {
        if (mums.size() <= 1) break;
        for (const PkMum &m : mums) if (m.pos.size() <= c) return false;      // (a match with fewer members than the first: IndexError there)
        std::stable_sort(mums.begin(), mums.end(), [c](const PkMum &a, const PkMum &b) { return a.pos[c] != b.pos[c] ? a.pos[c] < b.pos[c] : a.l > b.l; });
        auto end = [c](const PkMum &m) { return m.pos[c] + m.l; };
        std::vector<PkMum> kept;
        const size_t cnt = mums.size();
        for (size_t i = 0; i < cnt; i++) {
            const PkMum &mm = mums[i];
            const PkMum &prev = mums[i == 0 ? cnt - 1 : i - 1];      // (i - 1 == -1: the last one)
            if ((i == 0 && end(mums[1]) > end(mm)) || end(prev) < end(mm)) kept.push_back(mm);
        }
        mums.swap(kept);
        if (mums.size() <= 1) break;
        std::vector<PkMum> trimmed;
        trimmed.push_back(mums[0]);
        for (size_t i = 1; i < mums.size(); i++) {
            if (trimmed.empty()) return false;
            const PkMum &mum = mums[i];
            PkMum &pm = trimmed.back();
            const int64_t overlap = end(pm) - mum.pos[c];
            if (overlap > 0) {
                if (pm.l - overlap > 0) pm.l -= overlap; else trimmed.pop_back();
                if (mum.l - overlap > 0) {
                    PkMum t = mum;
                    t.l -= overlap;
                    for (int64_t &p : t.pos) p += overlap;
                    trimmed.push_back(std::move(t));
                }
            } else trimmed.push_back(mum);
        }
        mums.swap(trimmed);
    }
    return true;
}
}  // namespace

extern "C" int rv_pick_chain(const rv_picker_args *A, int nsub, int64_t m, const uint32_t *l, const int32_t *n, const int64_t *off, const uint16_t *so,
                             const int64_t *pos, int nsamples, const int64_t *seq_begin, const int64_t *iv_begin, const int64_t *iv_end, int minlength,
                             rv_picker_out *O) {
    if (!A || !O || m < 0 || nsamples < 1) { rv_set_error("rv_pick_chain: bad arguments"); return -1; }
    O->picked = 0; O->nleft = O->nright = 0; O->nseed_members = 0;
    if (m == 0) return 0;
    std::vector<PkMum> all((size_t)m);
    for (int64_t i = 0; i < m; i++) {
        PkMum &x = all[(size_t)i];
        x.l = l[i]; x.n = n[i];
        x.so.assign(so + off[i], so + off[i + 1]); x.pos.assign(pos + off[i], pos + off[i + 1]);
    }
    // schemes.py:227-233: the matches in every sample of the sub-index; none and more than two samples: the best sample subset (`segment`, :107-126)
    std::vector<PkMum> mm;
    for (const PkMum &x : all) if (x.n == nsub) mm.push_back(x);
    if (mm.empty() && nsub > 2) {
        std::vector<std::vector<uint16_t>> keys; std::vector<std::vector<size_t>> members;
        for (size_t i = 0; i < all.size(); i++) {
            std::vector<uint16_t> k = all[i].so;
            std::sort(k.begin(), k.end());
            size_t g = 0;
            for (; g < keys.size(); g++) if (keys[g] == k) break;
            if (g == keys.size()) { keys.push_back(k); members.emplace_back(); }
            members[g].push_back(i);
        }
        int64_t best = 0; size_t part = (size_t)-1;
        for (size_t g = 0; g < keys.size(); g++) {
            int64_t z = 0;
            for (size_t i : members[g]) z += all[i].l;
            z *= (int64_t)keys[g].size();
            if (z > best) { best = z; part = g; }
        }
        if (part == (size_t)-1) { rv_set_error("rv_pick_chain: no sample subset (the reference raises KeyError here)"); return -2; }
        for (size_t i : members[part]) mm.push_back(all[i]);
    }
    if (A->trim) {
        if (!mm.empty() && !pk_trim_overlap(mm)) { rv_set_error("rv_pick_chain: trim_overlap ran out of matches (the reference raises IndexError here)"); return -2; }
        if (mm.empty()) return 0;
    }
    if (mm.empty()) return 0;
    std::stable_sort(mm.begin(), mm.end(), [](const PkMum &a, const PkMum &b) { return a.l > b.l; });      // :240 (reverse=True keeps equal lengths in order)
    // maptooffsets (:150-158): rel[i] = offsets per path in member order; `mapping` keyed by the offsets, a later equal key replaces the earlier
    const size_t cnt = mm.size();
    std::vector<std::vector<int64_t>> rel(cnt);
    for (size_t i = 0; i < cnt; i++) {
        rel[i].resize(mm[i].pos.size());
        for (size_t j = 0; j < mm[i].pos.size(); j++) {
            const int s = mm[i].so[j];
            if (s >= nsamples) { rv_set_error("rv_pick_chain: sample id out of range"); return -1; }
            rel[i][j] = mm[i].pos[j] - seq_begin[s];
        }
    }
    auto mapped = [&](size_t i) -> size_t {      // mapping[tuple(rel.values())]: the LAST match with these offsets
        size_t r = i;
        for (size_t j = i + 1; j < cnt; j++) if (rel[j] == rel[i]) r = j;
        return r;
    };
    std::vector<size_t> ord(cnt);
    for (size_t i = 0; i < cnt; i++) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {      // :247 key (n, l); n = paths of the members' nodes = members
        const size_t na = mm[a].pos.size(), nb = mm[b].pos.size();
        return na != nb ? na < nb : mm[a].l < mm[b].l;
    });
    auto keyset = [&](size_t i) { std::vector<uint16_t> k = mm[i].so; std::sort(k.begin(), k.end()); return k; };
    const std::vector<uint16_t> last = keyset(ord.back());
    std::vector<size_t> relm;
    for (size_t i : ord) if (keyset(i) == last) relm.push_back(i);
    if (relm.empty()) return 0;
    // the chain's sentinels over the paths of the last match, in ascending path id (schemes.chain sorts the keys)
    const int k = (int)last.size();
    std::vector<int64_t> lf((size_t)k), rt((size_t)k);
    for (int j = 0; j < k; j++) {
        const int s = last[(size_t)j];
        if (iv_begin[s] < 0) { rv_set_error("rv_pick_chain: a match in a sample the sub-index does not hold"); return -1; }
        lf[(size_t)j] = iv_begin[s] - 1 - seq_begin[s];
        rt[(size_t)j] = iv_end[s] - seq_begin[s];
    }
    auto coord = [&](size_t i, int j) -> int64_t {      // offset of match i on path last[j]
        for (size_t q = 0; q < mm[i].so.size(); q++) if (mm[i].so[q] == last[(size_t)j]) return rel[i][q];
        return 0;
    };
    size_t split;
    std::vector<std::pair<size_t, int64_t>> chained;      // (match, score) left to right
    if (relm.size() == 1) split = relm[0];
    else {
        if (A->maxmums > 0 && (int64_t)relm.size() > A->maxmums) relm.erase(relm.begin(), relm.end() - (ptrdiff_t)A->maxmums);      // :287-289 relmums[-maxmums:]
        const int64_t mc = (int64_t)relm.size();
        std::vector<uint32_t> cl((size_t)mc); std::vector<int32_t> cn((size_t)mc); std::vector<int64_t> crd((size_t)mc * k), oi((size_t)mc), osc((size_t)mc);
        for (int64_t i = 0; i < mc; i++) {
            cl[(size_t)i] = (uint32_t)mm[relm[(size_t)i]].l; cn[(size_t)i] = (int32_t)mm[relm[(size_t)i]].pos.size();
            for (int j = 0; j < k; j++) crd[(size_t)i * k + j] = coord(relm[(size_t)i], j);
        }
        const int64_t r = rv_chain(mc, k, cl.data(), cn.data(), crd.data(), lf.data(), rt.data(), A->wscore, A->wpen, A->gcmodel, oi.data(), osc.data());
        if (r < 0) return -1;
        if (r == 0) return 0;
        for (int64_t q = 0; q < r; q++) chained.push_back({relm[(size_t)oi[(size_t)q]], osc[(size_t)q]});
        // "largest" of the chain: sorted by length (stable), the last one (:313-315)
        split = chained[0].first;
        for (auto &c : chained) if (mm[c.first].l >= mm[split].l) split = c.first;
    }
    // seeds for the children (:321-332): the rest of the chain, scores relative to the split's, above --seedsize
    struct Seed { size_t i; int64_t sc; bool right; };
    std::vector<Seed> seeds;
    if (!chained.empty() && A->seedsize > 0) {
        int64_t at = 0; bool right = false;
        for (auto &c : chained) {
            if (c.first == split) { at = c.second; right = true; continue; }
            seeds.push_back({mapped(c.first), c.second - at, right});
        }
    }
    const size_t sm = mapped(split);
    if (minlength == 0) {      // :336-348
        long double o = 1;
        for (int j = 0; j < k; j++) o *= (long double)(rt[(size_t)j] - lf[(size_t)j]);
        const double nn = (double)mm[sm].n, ll = (double)mm[sm].l;
        double p = std::pow(std::pow(0.25, nn - 1.0), ll);
        if (p > 0) p = p < 1 ? 1.0 - std::exp(std::log(1.0 - p) * (double)o) : 1.0;
        if (p > A->pcutoff) return 0;
    }
    auto put = [&](size_t i, uint32_t *ol, int32_t *on, uint16_t *oso, int64_t *opos) -> int {
        *ol = (uint32_t)mm[i].l; *on = mm[i].n;
        for (size_t q = 0; q < mm[i].pos.size(); q++) { oso[q] = mm[i].so[q]; opos[q] = mm[i].pos[q]; }
        return (int)mm[i].pos.size();
    };
    if ((int64_t)mm[sm].pos.size() > O->member_cap) { rv_set_error("rv_pick_chain: output too small"); return -1; }
    O->picked = 1;
    O->pick_members = put(sm, &O->pick_l, &O->pick_n, O->pick_so, O->pick_pos);
    int64_t w = 0, wm = 0;
    for (const Seed &s : seeds) {
        if (mm[s.i].l < A->seedsize) continue;
        if (w >= O->seed_cap || wm + (int64_t)mm[s.i].pos.size() > O->seed_member_cap) { rv_set_error("rv_pick_chain: seed output too small"); return -1; }
        O->seed_off[w] = wm;
        wm += put(s.i, &O->seed_l[w], &O->seed_n[w], O->seed_so + wm, O->seed_pos + wm);
        O->seed_score[w] = s.sc;
        O->seed_right[w] = s.right ? 1 : 0;
        if (s.right) O->nright++; else O->nleft++;
        w++;
    }
    O->seed_off[w] = wm;
    O->nseed_members = wm;
    return 1;
}

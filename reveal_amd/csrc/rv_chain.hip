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
#include "rv_pick.h"
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
    // The reference keys three dictionaries by the coordinate on the first path (sp2mum, link, score: two matches that start at the same place on
    // that path share an entry, the later one replaces the earlier).  Same semantics on arrays: every distinct coordinate gets a slot once.  (As hash
    // maps looked at inside the sort's comparator they were most of a call: 1000 matches = 100 ms, 10.5 of the 11 s five 5 Mbp genomes spent in the
    // library with the native picker.)
    std::vector<int64_t> cs((size_t)m + 2);
    for (int64_t e = 0; e <= m + 1; e++) cs[(size_t)e] = C(e, 0);
    std::vector<int64_t> uniq(cs);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    std::vector<int32_t> slot((size_t)m + 2);
    for (int64_t e = 0; e <= m + 1; e++) slot[(size_t)e] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), cs[(size_t)e]) - uniq.begin());
    const size_t ns = uniq.size();
    std::vector<int64_t> sp2mum(ns, -1), score(ns, 0);
    std::vector<int32_t> link(ns, -1);
    for (int64_t e : order) sp2mum[(size_t)slot[(size_t)e]] = e;                // schemes.py:32-34 (a later equal coordinate replaces the earlier)
    score[(size_t)slot[(size_t)L]] = 0;
    std::vector<int64_t> active{L}, processed, keep, D;
    std::vector<int64_t> ea((size_t)k), sb((size_t)k);
    int64_t best = -1;
    auto ends_before = [&](int64_t a, int64_t b) -> bool {      // a ends at or in front of b's start on every path (schemes.py:49-51, 63-65)
        const int64_t la = LEN(a);
        for (int j = 0; j < k; j++) if (C(a, j) + la > C(b, j)) return false;
        return true;
    };
    auto by_score = [&](int64_t a, int64_t b) { return score[(size_t)slot[(size_t)a]] > score[(size_t)slot[(size_t)b]]; };
    for (int64_t e : order) {
        keep.clear();
        const size_t old = active.size();
        for (int64_t p : processed) { if (ends_before(p, e)) active.push_back(p); else keep.push_back(p); }      // schemes.py:47-57
        processed.swap(keep);
        // schemes.py:59: the list stably sorted by score.  What was there is in that order already unless a match that shares its first coordinate with an
        // earlier one has replaced that one's score: then (checked) the whole list is sorted, otherwise the newcomers are, and merged in -- the same list
        if (std::is_sorted(active.begin(), active.begin() + (ptrdiff_t)old, by_score)) {
            std::stable_sort(active.begin() + (ptrdiff_t)old, active.end(), by_score);
            std::inplace_merge(active.begin(), active.begin() + (ptrdiff_t)old, active.end(), by_score);
        } else std::stable_sort(active.begin(), active.end(), by_score);
        bool have = false;
        int64_t w = 0;
        const int64_t n = e >= m ? 0 : (int64_t)nmem[e];
        const int64_t gain = wscore * (LEN(e) * ((n * (n - 1)) / 2));
        for (int j = 0; j < k; j++) sb[(size_t)j] = C(e, j);
        for (int64_t a : active) {
            if (!ends_before(a, e)) continue;
            const int64_t sc = score[(size_t)slot[(size_t)a]] + gain;
            if (have && w > sc) break;                           // sorted by score: nothing better can follow (schemes.py:70-72)
            const int64_t la = LEN(a);
            for (int j = 0; j < k; j++) ea[(size_t)j] = C(a, j) + la;
            const int64_t tmpw = sc - wpen * gapcost(ea.data(), sb.data(), k, model, D);
            if (!have || tmpw > w) { w = tmpw; best = a; have = true; }
        }
        if (best < 0 || !have) { rv_set_error("rv_chain: a match has no predecessor (it does not lie behind `left` on every path)"); return -1; }
        link[(size_t)slot[(size_t)e]] = slot[(size_t)best];
        score[(size_t)slot[(size_t)e]] = w;
        processed.push_back(e);
    }
    // backtrack from `right` (schemes.py:97-103); the chain is handed out left to right, `right` itself left out
    std::vector<std::pair<int64_t, int64_t>> path;
    int32_t end = slot[(size_t)R];
    const int32_t start = slot[(size_t)L];
    int64_t guard = 0;
    while (end != start) {
        if (end < 0 || sp2mum[(size_t)end] < 0 || ++guard > m + 2) { rv_set_error("rv_chain: broken back-pointer chain"); return -1; }
        path.push_back({sp2mum[(size_t)end], score[(size_t)end]});
        end = link[(size_t)end];
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
extern "C" int rv_pick_chain(const rv_picker_args *A, int nsub, int64_t m, const uint32_t *l, const int32_t *n, const int64_t *off, const uint16_t *so,
                             const int64_t *pos, int nsamples, const int64_t *seq_begin, const int64_t *iv_begin, const int64_t *iv_end, int minlength,
                             rv_picker_out *O) {
    if (!A || !O || m < 0 || nsamples < 1) { rv_set_error("rv_pick_chain: bad arguments"); return -1; }
    O->picked = 0; O->nleft = O->nright = 0; O->nseed_members = 0;
    if (m == 0) return 0;
    const PkCtx X{so, pos};
    auto item = [&](int64_t i) { PkItem x; x.l = l[i]; x.shift = 0; x.off = off[i]; x.n = n[i]; x.nm = (int32_t)(off[i + 1] - off[i]); return x; };
    // the sample set of a match: a bit mask (more than 64 samples: the sorted ids)
    const bool wide = nsamples > 64;
    auto mask_of = [&](const PkItem &x) { uint64_t k = 0; for (int q = 0; q < x.nm; q++) k |= 1ull << (so[x.off + q] & 63); return k; };
    auto ids_of = [&](const PkItem &x) { std::vector<uint16_t> k(so + x.off, so + x.off + x.nm); std::sort(k.begin(), k.end()); return k; };
    auto same_set = [&](const PkItem &a, const PkItem &b) { return a.nm == b.nm && (wide ? ids_of(a) == ids_of(b) : mask_of(a) == mask_of(b)); };
    for (int64_t i = 0; i < m; i++)
        for (int64_t q = off[i]; q < off[i + 1]; q++) if (so[q] >= nsamples) { rv_set_error("rv_pick_chain: sample id out of range"); return -1; }
    // schemes.py:227-233: the matches in every sample of the sub-index; none and more than two samples: the best sample subset (`segment`, :107-126)
    std::vector<PkItem> mm;
    for (int64_t i = 0; i < m; i++) if (n[i] == nsub) mm.push_back(item(i));
    if (mm.empty() && nsub > 2) {
        std::vector<PkItem> reps; std::vector<int64_t> zsum; std::vector<int32_t> grp((size_t)m);
        for (int64_t i = 0; i < m; i++) {
            const PkItem x = item(i);
            size_t g = 0;
            for (; g < reps.size(); g++) if (same_set(reps[g], x)) break;
            if (g == reps.size()) { reps.push_back(x); zsum.push_back(0); }
            zsum[g] += x.l; grp[(size_t)i] = (int32_t)g;
        }
        int64_t best = 0; size_t part = (size_t)-1;
        for (size_t g = 0; g < reps.size(); g++) { const int64_t z = zsum[g] * reps[g].nm; if (z > best) { best = z; part = g; } }
        if (part == (size_t)-1) { rv_set_error("rv_pick_chain: no sample subset (the reference raises KeyError here)"); return -2; }
        for (int64_t i = 0; i < m; i++) if (grp[(size_t)i] == (int32_t)part) mm.push_back(item(i));
    }
    if (A->trim) {
        if (!mm.empty() && !pk_trim_overlap(mm, X)) { rv_set_error("rv_pick_chain: trim_overlap ran out of matches (the reference raises IndexError here)"); return -2; }
        if (mm.empty()) return 0;
    }
    if (mm.empty()) return 0;
    std::stable_sort(mm.begin(), mm.end(), [](const PkItem &a, const PkItem &b) { return a.l > b.l; });      // :240 (reverse=True keeps equal lengths in order)
    // maptooffsets (:150-158): a match's offsets per path in member order; `mapping` is keyed by them, a later equal key replaces the earlier
    const size_t cnt = mm.size();
    auto rel = [&](size_t i, int q) { return X.at(mm[i], (size_t)q) - seq_begin[so[mm[i].off + q]]; };
    auto same_rel = [&](size_t a, size_t b) {
        if (mm[a].nm != mm[b].nm) return false;
        for (int q = 0; q < mm[a].nm; q++) if (rel(a, q) != rel(b, q)) return false;
        return true;
    };
    // mapping[tuple(rel.values())]: the LAST match with these offsets.  One look-up walks the list; the seeds of a long chain over a long list
    // (1000 x 1.5 x 10^6 at the root of five 5 Mbp genomes) go through a table of the offsets' hashes made once
    std::unordered_map<uint64_t, uint32_t> last_by_hash;
    auto rel_hash = [&](size_t i) { uint64_t hsh = 0x9E3779B97F4A7C15ull ^ (uint64_t)mm[i].nm; for (int q = 0; q < mm[i].nm; q++) { hsh ^= (uint64_t)rel(i, q) + 0x9E3779B97F4A7C15ull + (hsh << 6) + (hsh >> 2); } return hsh; };
    auto mapped = [&](size_t i) -> size_t {
        if (!last_by_hash.empty()) {
            const auto it = last_by_hash.find(rel_hash(i));
            if (it != last_by_hash.end() && same_rel(it->second, i)) return it->second;      // (two different offset tuples under one hash: the walk)
        }
        size_t r = i;
        for (size_t j = i + 1; j < cnt; j++) if (same_rel(j, i)) r = j;
        return r;
    };
    std::vector<uint32_t> ord(cnt);
    for (size_t i = 0; i < cnt; i++) ord[i] = (uint32_t)i;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {      // :247 key (n, l); n = paths of the members' nodes = members
        return mm[a].nm != mm[b].nm ? mm[a].nm < mm[b].nm : mm[a].l < mm[b].l;
    });
    const PkItem lastm = mm[ord.back()];
    std::vector<uint32_t> relm;
    for (uint32_t i : ord) if (same_set(mm[i], lastm)) relm.push_back(i);
    if (relm.empty()) return 0;
    // the chain's sentinels over the paths of the last match, in ascending path id (schemes.chain sorts the keys)
    const std::vector<uint16_t> last = ids_of(lastm);
    const int k = (int)last.size();
    std::vector<int64_t> lf((size_t)k), rt((size_t)k);
    for (int j = 0; j < k; j++) {
        const int s2 = last[(size_t)j];
        if (iv_begin[s2] < 0) { rv_set_error("rv_pick_chain: a match in a sample the sub-index does not hold"); return -1; }
        lf[(size_t)j] = iv_begin[s2] - 1 - seq_begin[s2];
        rt[(size_t)j] = iv_end[s2] - seq_begin[s2];
    }
    auto coord = [&](size_t i, int j) -> int64_t {      // offset of match i on path last[j]
        for (int q = 0; q < mm[i].nm; q++) if (so[mm[i].off + q] == last[(size_t)j]) return rel(i, q);
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
            cl[(size_t)i] = (uint32_t)mm[relm[(size_t)i]].l; cn[(size_t)i] = mm[relm[(size_t)i]].nm;
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
        if (chained.size() > 4 && cnt > 64) { last_by_hash.reserve(cnt * 2); for (size_t i = 0; i < cnt; i++) last_by_hash[rel_hash(i)] = (uint32_t)i; }
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
        for (int q = 0; q < mm[i].nm; q++) { oso[q] = so[mm[i].off + q]; opos[q] = X.at(mm[i], (size_t)q); }
        return mm[i].nm;
    };
    if ((int64_t)mm[sm].nm > O->member_cap) { rv_set_error("rv_pick_chain: output too small"); return -1; }
    O->picked = 1;
    O->pick_members = put(sm, &O->pick_l, &O->pick_n, O->pick_so, O->pick_pos);
    int64_t w = 0, wm = 0;
    for (const Seed &s2 : seeds) {
        if (mm[s2.i].l < A->seedsize) continue;
        if (w >= O->seed_cap || wm + (int64_t)mm[s2.i].nm > O->seed_member_cap) { rv_set_error("rv_pick_chain: seed output too small"); return -1; }
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

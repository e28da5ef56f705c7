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

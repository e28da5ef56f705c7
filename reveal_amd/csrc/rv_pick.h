// rv_pick.h -- what the two pickers behind the ABI share (rv_chain.hip: FASTA inputs, one sequence per sample; rv_graphrem.hip: graph inputs):
// a list entry that points into the caller's arrays, and schemes.trim_overlap on such entries.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

// a match of the list: its length after trimming, how far trimming moved its members (the same distance on every path), and where its
// members stand in the caller's arrays -- the members themselves are never copied (a list of 1.5 x 10^6 five-way matches as vectors of vectors
// made the root's call 6 s of a 5 x 5 Mbp job's 13.8 s in the picker)
struct PkItem { int64_t l, shift; int64_t off; int32_t n, nm; };

struct PkCtx {
    const uint16_t *so; const int64_t *pos;
    int64_t at(const PkItem &m, size_t c) const { return pos[m.off + (int64_t)c] + m.shift; }
};

// schemes.py:160-193; -> false: the reference's own code would raise here (trimmed[-1] of an empty list, or a match with fewer members than the first)
inline bool pk_trim_overlap(std::vector<PkItem> &mums, const PkCtx &X) {
    if (mums.empty()) return true;
    const size_t ncoord = (size_t)mums[0].nm;
    std::vector<PkItem> kept, trimmed;
    for (size_t c = 0; c < ncoord; c++) {
        if (mums.size() <= 1) break;
        for (const PkItem &m : mums) if ((size_t)m.nm <= c) return false;
        std::stable_sort(mums.begin(), mums.end(), [&](const PkItem &a, const PkItem &b) { const int64_t pa = X.at(a, c), pb = X.at(b, c); return pa != pb ? pa < pb : a.l > b.l; });
        auto end = [&](const PkItem &m) { return X.at(m, c) + m.l; };
        kept.clear();
        const size_t cnt = mums.size();
        for (size_t i = 0; i < cnt; i++) {
            const PkItem &mm = mums[i];
            const PkItem &prev = mums[i == 0 ? cnt - 1 : i - 1];      // (i - 1 == -1: the last one)
            if ((i == 0 && end(mums[1]) > end(mm)) || end(prev) < end(mm)) kept.push_back(mm);
        }
        mums.swap(kept);
        if (mums.size() <= 1) break;
        trimmed.clear();
        trimmed.push_back(mums[0]);
        for (size_t i = 1; i < mums.size(); i++) {
            if (trimmed.empty()) return false;
            const PkItem &mum = mums[i];
            PkItem &pm = trimmed.back();
            const int64_t overlap = end(pm) - X.at(mum, c);
            if (overlap > 0) {
                if (pm.l - overlap > 0) pm.l -= overlap; else trimmed.pop_back();
                if (mum.l - overlap > 0) { PkItem t = mum; t.l -= overlap; t.shift += overlap; trimmed.push_back(t); }
            } else trimmed.push_back(mum);
        }
        mums.swap(trimmed);
    }
    return true;
}

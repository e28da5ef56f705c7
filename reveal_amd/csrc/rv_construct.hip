// rv_construct.hip -- suffix array, inverse and LCP on gfx950.
//
// Replaces the reference's construct() pipeline (reveallib/interface.c:160-291):
//   divsufsort(T,SA,n)            interface.c:215-222  -> rv_build_sa   (prefix doubling + radix sort)
//   SAi[SA[i]]=i                  interface.c:236-238  -> rv_build_inverse
//   compute_lcp (Kasai, stops at '$'/'N')  interface.c:97-114 -> rv_build_lcp (text order with the h-1 carry; or closed form, one thread per rank)
//
// A suffix array is unique for a text (unsigned byte order, shorter suffix
// first), so prefix doubling yields divsufsort's SA bit for bit.
#include "rv_common.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <limits>

#ifdef RV_SA64
typedef u64 sav_t;   // suffix ids as radix-sort payload
#else
typedef u32 sav_t;
#endif

namespace {

constexpr int TB = 256;

// ---- eight text bytes at a time ----------------------------------------------------------------------------
// 0x80 in every byte of v that equals c (exact: no carries between bytes)
__device__ inline u64 swar_eq(u64 v, uint8_t c) {
    const u64 t = v ^ (0x0101010101010101ull * c);
    return ~(((t & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | t) & 0x8080808080808080ull;
}
// 0x80 in every byte that is one of A, C, G, T
__device__ inline u64 swar_acgt(u64 v) { return swar_eq(v, 'A') | swar_eq(v, 'C') | swar_eq(v, 'G') | swar_eq(v, 'T'); }
// the 2-bit codes (A, C, G, T = 0..3: ((c >> 1) ^ (c >> 2)) & 3 on their ASCII codes) of eight bytes, byte j in bits 2j, 2j + 1
__device__ inline u32 swar_code2(u64 v) {
    u64 y = ((v >> 1) ^ (v >> 2)) & 0x0303030303030303ull;
    y = (y | (y >> 6)) & 0x000f000f000f000full;
    y = (y | (y >> 12)) & 0x000000ff000000ffull;
    y = (y | (y >> 24)) & 0xffffull;
    return (u32)y;
}

// ---- alphabet ---------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_hist256(const uint8_t *__restrict__ T, int64_t n, u32 *__restrict__ hist) {
    // A DNA text is A, C, G, T and the separators: those five are counted in registers, eight bytes per step (equality masks and a population
    // count); only other bytes go to the LDS table.  (One LDS atomic per byte sent every lane of a wave to the same four words: 0.32 ms for 0.5 GB.)
    __shared__ u32 hh[256];
    hh[threadIdx.x] = 0;
    __syncthreads();
    u32 cA = 0, cC = 0, cG = 0, cT = 0, cS = 0;
    const int64_t stride = (int64_t)gridDim.x * TB * 16;
    for (int64_t base = ((int64_t)blockIdx.x * TB + threadIdx.x) * 16; base < n; base += stride) {
        if (base + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(T + base);
            const u64 w[2] = {(u64)v.x | ((u64)v.y << 32), (u64)v.z | ((u64)v.w << 32)};
            u64 sep[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const u64 eA = swar_eq(w[k], 'A'), eC = swar_eq(w[k], 'C'), eG = swar_eq(w[k], 'G'), eT = swar_eq(w[k], 'T'), eS = swar_eq(w[k], '$');
                cA += (u32)__popcll(eA); cC += (u32)__popcll(eC); cG += (u32)__popcll(eG); cT += (u32)__popcll(eT); cS += (u32)__popcll(eS);
                sep[k] = eS;
                u64 other = ~(eA | eC | eG | eT | eS) & 0x8080808080808080ull;
                while (other) {
                    const int b = __builtin_ctzll(other) >> 3;
                    atomicAdd(&hh[(w[k] >> (8 * b)) & 255u], 1u);
                    other &= other - 1;
                }
            }
            // two separators next to each other (an empty sequence): hist[256] -- rv_build_sa's shorter alphabet needs to know
            if (sep[0] | sep[1]) for (int64_t i = base; i < base + 16; i++) if (T[i] == '$' && i + 1 < n && T[i + 1] == '$') atomicAdd(&hist[256], 1u);
        } else {
            for (int64_t i = base; i < n; i++) { atomicAdd(&hh[T[i]], 1u); if (T[i] == '$' && i + 1 < n && T[i + 1] == '$') atomicAdd(&hist[256], 1u); }
        }
    }
    if (cA) atomicAdd(&hh[(uint8_t)'A'], cA);
    if (cC) atomicAdd(&hh[(uint8_t)'C'], cC);
    if (cG) atomicAdd(&hh[(uint8_t)'G'], cG);
    if (cT) atomicAdd(&hh[(uint8_t)'T'], cT);
    if (cS) atomicAdd(&hh[(uint8_t)'$'], cS);
    __syncthreads();
    const u32 tot = hh[threadIdx.x];
    if (tot) atomicAdd(&hist[threadIdx.x], tot);
}

// ---- first key: K symbols as digits of a base-(sigma+1) number -----------------
// key(i) = sum_j code[T[i+j]] * radix^(K-1-j), code 0 = past the end of the text
// (so the shorter suffix sorts first).  Packing by radix instead of by bits puts
// twelve DNA symbols (codes 0..5) into 32 bits: four radix-sort passes for n = 1e7
// where 3 bits per symbol needed six.  A block stages 1024+K symbols as codes in
// LDS; each thread then packs 4 keys from LDS.
constexpr int KEY_TILE = 1024;

// ---- the diagonal hint ------------------------------------------------------------------------
// Two related genomes: nearly every suffix p of the first sample shares its first K symbols with exactly one other suffix, its
// twin p + D on the diagonal D = nsep[0] + 1 (same coordinate in the second sample), and what orders the two is the next
// position where the samples differ -- the same position for every suffix of a substitution-free stretch.  The text round
// used to find it pair by pair: two random 40-byte reads of the packed text per pair and step, 33 of the build's 70 ms at
// 2 x 250 Mbp.  Along one diagonal it is a streaming computation instead: k_diag_bits marks, for every position y, whether
// T[y] and T[y + D] differ (or either is not A / C / G / T: an exception) and which is smaller; k_init_keys then knows for
// every suffix p how far its twin agrees with it (nd = next marked position - p) and which of the two is smaller, and puts
// that into the spare bits of p's first key, above the bits the sort looks at.  The text round reads a pair's order and LCP
// from the key it has in registers anyway; anything the hint does not cover (other diagonals after an indel, stretches longer
// than the field, exceptions, unrelated suffixes that collide in their K symbols) is compared on the text as before.
struct DiagBits { const u64 *stop, *exc, *lt; int64_t D; int tab; };      // one bit per text position each; stop = differs or exception; tab: the bits follow the diagonal table (k_diag_bits_tab)
struct DiagSamples { int ns; int64_t sep[15], Ds[16]; };            // (HINT_K - 1 separators, HINT_K diagonals: declared below)
// the partner of position y: its homologue in the first sample for a position of a later sample, its homologue in the second for one of the first
__device__ inline int64_t diag_partner(const DiagSamples &ds, int64_t y) {
    int s = 0;
    for (int q = 0; q < ds.ns - 1; q++) s += ds.sep[q] < y ? 1 : 0;
    return s == 0 ? y + ds.Ds[1] : (s < 16 ? y - ds.Ds[s] : -1);
}
__global__ __launch_bounds__(TB) void k_diag_bits(const uint8_t *__restrict__ T, int64_t n, DiagSamples ds, u64 *__restrict__ stop, u64 *__restrict__ exc,
                                                  u64 *__restrict__ lt, int64_t nwords) {
    const int64_t w = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (w >= nwords) return;
    const int64_t y0 = w * 64;
    u64 ws = 0, we = 0, wl = 0;
    const int64_t p0 = diag_partner(ds, y0), p63 = y0 + 63 < n ? diag_partner(ds, y0 + 63) : -1;
    if (y0 + 64 <= n && p0 >= 0 && p63 == p0 + 63 && p0 + 64 <= n) {      // the whole word on one diagonal
        u64 a[8], b[8];
        __builtin_memcpy(a, T + y0, 64);
        __builtin_memcpy(b, T + p0, 64);
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const u32 ca = (u32)(a[k] >> (8 * j)) & 0xffu, cb = (u32)(b[k] >> (8 * j)) & 0xffu;
                const bool ea = !((ca == 'A') | (ca == 'C') | (ca == 'G') | (ca == 'T')), eb = !((cb == 'A') | (cb == 'C') | (cb == 'G') | (cb == 'T'));
                const int bit = 8 * k + j;
                we |= (u64)(ea | eb) << bit; ws |= (u64)((ca != cb) | ea | eb) << bit; wl |= (u64)(ca < cb) << bit;
            }
    } else {
        for (int bit = 0; bit < 64; bit++) {
            const int64_t y = y0 + bit;
            const int64_t p = y < n ? diag_partner(ds, y) : -1;
            if (p < 0 || p >= n) { we |= 1ull << bit; ws |= 1ull << bit; continue; }      // no partner on the diagonal
            const u32 ca = T[y], cb = T[p];
            const bool ea = !((ca == 'A') | (ca == 'C') | (ca == 'G') | (ca == 'T')), eb = !((cb == 'A') | (cb == 'C') | (cb == 'G') | (cb == 'T'));
            we |= (u64)(ea | eb) << bit; ws |= (u64)((ca != cb) | ea | eb) << bit; wl |= (u64)(ca < cb) << bit;
        }
    }
    stop[w] = ws; exc[w] = we; lt[w] = wl;
}
// ---- piecewise diagonals (two samples) ----------------------------------------------------------------------------------
// One fixed diagonal D = nsep[0] + 1 only describes a pair of genomes without insertions or deletions: behind the first indel every
// position of the second sample has moved off it, and neither the hint nor the twins' leaving covers anything.  The table gives every
// tile of 16 text positions its own diagonal: partner(y) = y + D + dtab[y >> 4] for a position of the first sample,
// y - D - dtab[y >> 4] for one of the second -- the same deviation dd on both sides of a pair.  A position is LINKED when its
// partner lies in the other sample and the partner's tile carries the same dd: then the partner's partner is the position itself, so
// both sides see the same pairs whatever the table holds (every entry may be wrong: a link is only ever used through the characters
// k_diag_bits_tab compares along it).  A position that is not linked, or whose predecessor is linked differently (the start of a run,
// a change of diagonal), is marked as an exception: the marks of a pair are equal on both sides (link(y - 1) != link(y) holds for y
// iff it holds for its partner), which is what the twins' predicate needs -- a window without marks lies on ONE diagonal.
// The table comes from seeds: 32-mers sampled by content (hash & 15 == 0: the same k-mers in both samples), sorted by hash; a hash
// that occurs exactly twice, once per sample, votes dd = q - p - D for the tiles the two copies cover (the smallest vote wins); a tile
// without a vote looks for the nearest votes on either side and, when they differ (an indel in between), takes the one along which
// more of its own sixteen characters agree.  An indel costs the tile it lies in, not the rest of the genome.
constexpr int32_t DT_NONE = 0x7fffffff;
constexpr int DT_SHIFT = 4, DT_TILE = 1 << DT_SHIFT;
constexpr int SEED_K = 32, SEED_SLOTS = 4;          // seed length; seeds a word of 64 positions may emit
constexpr u64 SEED_EMPTY = (1ull << 40) - 1;        // key of an unused slot: sorts behind every hash (a hash that spells it is dropped)
struct DiagTab { const int32_t *dtab; int64_t S2, D; };      // S2 = first position of the second sample (= D: the fixed diagonal it deviates from)
__device__ inline u64 seed_mix(u64 x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }
// one thread per word: the 32-mers ENDING inside it, rolled as 2-bit codes; the first SEED_SLOTS that pass the filter go to the word's slots
__global__ __launch_bounds__(TB) void k_seed_sample(const uint8_t *__restrict__ T, int64_t n, u64 *__restrict__ keys, sav_t *__restrict__ vals, int64_t nwords) {
    const int64_t w = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (w >= nwords) return;
    const int64_t y0 = w * 64;
    u64 code = 0; int run = 0, cnt = 0;
    u64 ok[SEED_SLOTS]; sav_t ov[SEED_SLOTS];
#pragma unroll
    for (int k = 0; k < SEED_SLOTS; k++) { ok[k] = SEED_EMPTY; ov[k] = 0; }
    for (int64_t i = y0 - (SEED_K - 1); i < y0 + 64 && i < n; i++) {
        const u32 c = i >= 0 ? T[i] : 0u;
        const u32 c2 = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
        run = c2 < 4u ? run + 1 : 0;
        code = (code << 2) | (c2 & 3u);
        if (i >= y0 && run >= SEED_K) {
            const u64 h = seed_mix(code);
            if ((h & 15u) == 0u && cnt < SEED_SLOTS) {
                const u64 k40 = h >> 24;
                if (k40 != SEED_EMPTY) {
#pragma unroll
                    for (int k = 0; k < SEED_SLOTS; k++) if (k == cnt) { ok[k] = k40; ov[k] = (sav_t)(i - (SEED_K - 1)); }
                    cnt++;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < SEED_SLOTS; k++) { keys[w * SEED_SLOTS + k] = ok[k]; vals[w * SEED_SLOTS + k] = ov[k]; }
}
// sorted seeds: a hash held by exactly two seeds, one in each sample, votes for the tiles its two copies cover
__global__ __launch_bounds__(TB) void k_seed_pairs(const u64 *__restrict__ keys, const sav_t *__restrict__ vals, int64_t m, int64_t S2, int64_t D, int32_t *__restrict__ raw) {
    const int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (j + 1 >= m) return;
    const u64 mk = (1ull << 40) - 1;
    const u64 k0 = keys[j] & mk, k1 = keys[j + 1] & mk;
    if (k0 == SEED_EMPTY || k0 != k1) return;
    if (j > 0 && (keys[j - 1] & mk) == k0) return;
    if (j + 2 < m && (keys[j + 2] & mk) == k0) return;
    int64_t a = (int64_t)vals[j], b = (int64_t)vals[j + 1];
    if (a > b) { const int64_t t = a; a = b; b = t; }
    if (!(a + SEED_K < S2 && b >= S2)) return;
    const int64_t dd = b - a - D;
    if (dd <= -(int64_t)0x3fffffff || dd >= (int64_t)0x3fffffff) return;
    for (int64_t t = a >> DT_SHIFT; t <= (a + SEED_K - 1) >> DT_SHIFT; t++) atomicMin(&raw[t], (int32_t)dd);
    for (int64_t t = b >> DT_SHIFT; t <= (b + SEED_K - 1) >> DT_SHIFT; t++) atomicMin(&raw[t], (int32_t)dd);
}
// characters of tile t that agree with their partners along deviation dd (every character counts: it only ranks two candidates)
__device__ inline int dtab_agree(const uint8_t *__restrict__ T, int64_t n, int64_t S2, int64_t D, int64_t t, int32_t dd) {
    int c = 0;
    for (int k = 0; k < DT_TILE; k++) {
        const int64_t y = (t << DT_SHIFT) + k;
        if (y >= n) break;
        const int64_t z = y < S2 ? y + D + (int64_t)dd : y - D - (int64_t)dd;
        c += (z >= 0 && z < n && T[y] == T[z]) ? 1 : 0;
    }
    return c;
}
// a tile without a vote: the nearest votes to its left and right (64 tiles each way); two different ones -> the one its own text follows
__global__ __launch_bounds__(TB) void k_dtab_fill(const uint8_t *__restrict__ T, int64_t n, int64_t S2, int64_t D, const int32_t *__restrict__ raw, int32_t *__restrict__ dtab, int64_t ntiles) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= ntiles) return;
    int32_t v = raw[t];
    if (v >= 0x7f000000) {
        int32_t L = DT_NONE, R = DT_NONE;
        for (int d = 1; d <= 64 && L == DT_NONE; d++) if (t - d >= 0) { const int32_t x = raw[t - d]; if (x < 0x7f000000) L = x; }
        for (int d = 1; d <= 64 && R == DT_NONE; d++) if (t + d < ntiles) { const int32_t x = raw[t + d]; if (x < 0x7f000000) R = x; }
        if (L == DT_NONE) v = R;
        else if (R == DT_NONE || R == L) v = L;
        else v = dtab_agree(T, n, S2, D, t, R) > dtab_agree(T, n, S2, D, t, L) ? R : L;
    }
    dtab[t] = v;
}
__device__ inline int32_t dtab_link(const DiagTab &dt, int64_t y, int64_t n, int64_t *partner) {
    if (y < 0 || y >= n || y == dt.S2 - 1) return DT_NONE;      // (the separator between the samples belongs to neither: a run of links never crosses it)
    const int32_t dd = dt.dtab[y >> DT_SHIFT];
    if (dd == DT_NONE) return DT_NONE;
    const bool s1 = y < dt.S2;
    const int64_t z = s1 ? y + dt.D + (int64_t)dd : y - dt.D - (int64_t)dd;
    if (z < 0 || z >= n || (s1 ? z < dt.S2 : z >= dt.S2 - 1)) return DT_NONE;
    if (dt.dtab[z >> DT_SHIFT] != dd) return DT_NONE;
    *partner = z;
    return dd;
}
// k_diag_bits along the table's diagonals: stop = the characters differ, or an exception; exception = not A / C / G / T on either side,
// not linked, or the first position of a run of equal links.  A thread takes a word, tile by tile.
__global__ __launch_bounds__(TB) void k_diag_bits_tab(const uint8_t *__restrict__ T, int64_t n, DiagTab dt, u64 *__restrict__ stop, u64 *__restrict__ exc,
                                                      u64 *__restrict__ lt, int64_t nwords) {
    const int64_t w = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (w >= nwords) return;
    const int64_t y0 = w * 64;
    u64 ws = 0, we = 0, wl = 0;
    int64_t zp = 0;
    int32_t prev = dtab_link(dt, y0 - 1, n, &zp);      // link of the position in front of the one looked at
    for (int sub = 0; sub < 64 / DT_TILE; sub++) {
        const int64_t ys = y0 + sub * DT_TILE;
        u32 ms = 0, me = 0, ml = 0;      // this tile's sixteen bits
        const int32_t dd = ys < n ? dt.dtab[ys >> DT_SHIFT] : DT_NONE;
        const bool one_side = ys >= dt.S2 || ys + DT_TILE - 1 < dt.S2 - 1;
        const int64_t z0 = ys < dt.S2 ? ys + dt.D + (int64_t)dd : ys - dt.D - (int64_t)dd;
        const bool other_side = ys < dt.S2 ? z0 >= dt.S2 : z0 + DT_TILE - 1 < dt.S2 - 1;
        if (dd == DT_NONE) { ms = me = 0xffffu; prev = DT_NONE; }
        else if (one_side && ys + DT_TILE <= n && z0 >= 0 && z0 + DT_TILE <= n && other_side) {      // the tile and its partners as two 16-byte loads
            const int off = (int)(z0 & (DT_TILE - 1));
            const bool v0 = dt.dtab[z0 >> DT_SHIFT] == dd, v1 = off ? dt.dtab[(z0 >> DT_SHIFT) + 1] == dd : v0;
            const u32 lowm = off ? ((1u << (DT_TILE - off)) - 1u) : 0xffffu;
            const u32 valid = (v0 ? lowm : 0u) | (v1 ? (~lowm & 0xffffu) : 0u);
            u64 a[2], b[2];
            __builtin_memcpy(a, T + ys, 16);
            __builtin_memcpy(b, T + z0, 16);
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const u32 ca = (u32)(a[k] >> (8 * j)) & 0xffu, cb = (u32)(b[k] >> (8 * j)) & 0xffu;
                    const bool ea = !((ca == 'A') | (ca == 'C') | (ca == 'G') | (ca == 'T')), eb = !((cb == 'A') | (cb == 'C') | (cb == 'G') | (cb == 'T'));
                    const int bit = 8 * k + j;
                    me |= (u32)(ea | eb) << bit; ms |= (u32)((ca != cb) | ea | eb) << bit; ml |= (u32)(ca < cb) << bit;
                }
            u32 forced = 0;
            if ((valid & 1u) && prev != dd) forced |= 1u;
            if (off && !v0 && v1) forced |= 1u << (DT_TILE - off);
            ms |= (~valid & 0xffffu) | forced; me |= (~valid & 0xffffu) | forced; ml &= valid & ~forced;
            prev = (valid >> (DT_TILE - 1)) & 1u ? dd : DT_NONE;
        } else {
            for (int bit = 0; bit < DT_TILE; bit++) {
                const int64_t y = ys + bit;
                int64_t z = 0;
                const int32_t l = dtab_link(dt, y, n, &z);
                if (l == DT_NONE || l != prev) { me |= 1u << bit; ms |= 1u << bit; prev = l; continue; }
                const u32 ca = T[y], cb = T[z];
                const bool ea = !((ca == 'A') | (ca == 'C') | (ca == 'G') | (ca == 'T')), eb = !((cb == 'A') | (cb == 'C') | (cb == 'G') | (cb == 'T'));
                me |= (u32)(ea | eb) << bit; ms |= (u32)((ca != cb) | ea | eb) << bit; ml |= (u32)((ca < cb) & !(ea | eb)) << bit;
            }
        }
        ws |= (u64)ms << (sub * DT_TILE); we |= (u64)me << (sub * DT_TILE); wl |= (u64)ml << (sub * DT_TILE);
    }
    stop[w] = ws; exc[w] = we; lt[w] = wl;
}
// ---- twins leave before the sort --------------------------------------------------------------------------------------
// Two samples: a suffix q of the second sample whose K symbols -- and the byte in front of it -- are those of its homologue p = q - D in
// the first (no marked position in [q - 1, q + K - 1]: k_diag_bits) would sort into p's group, carry p's key (same symbols, same byte in
// front, same agreement nd with its twin, the opposite "smaller" bit) and is told from p by the hint alone.  Such a q is not sorted: p
// is flagged instead (top bit of its payload), and k_heads_publish_tc writes q behind p when the sorted list is laid out in rank order.
// At 1 % divergence six in seven suffixes of the second sample leave: the radix passes move 0.57 n pairs instead of n.
// The predicate is a function of the stop bits alone, and those are the same for p and q (the same two characters are compared):
// both sides evaluate it on their own words.
constexpr sav_t TW_FLAG = (sav_t)1 << (sizeof(sav_t) * 8 - 1);
// bit b: no marked position in [64 w + b - 1, 64 w + b + K - 1]    (prev / lo / hi = words w - 1, w, w + 1 of the stop bits)
__device__ inline u64 tw_window_clear(u64 prev, u64 lo, u64 hi, int K) {
    u64 a = lo, b = hi;      // bit q of (b : a): a mark in [q, q + m) -- doubled up to K, then one step for what is left
    int m = 1;
    while (2 * m <= K) { a |= (a >> m) | (b << (64 - m)); b |= b >> m; m *= 2; }
    if (m < K) { const int d = K - m; a |= (a >> d) | (b << (64 - d)); }
    return ~(a | (lo << 1) | (prev >> 63));
}
// bits of word w whose positions lie in [a, b)
__device__ inline u64 tw_range(int64_t w, int64_t a, int64_t b) {
    const int64_t lo = a - 64 * w, hi = b - 64 * w;
    if (hi <= 0 || lo >= 64) return 0ull;
    const u64 from = lo <= 0 ? ~0ull : (~0ull << lo), to = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    return from & to;
}
// positions of word w that are twins: flagged (first sample, [1, D - 1)) or left out of the sort (second sample, [D + 1, min(2 D - 1, n)))
// (tab: piecewise diagonals -- a position without a partner is marked, so the samples' ranges are all that is left to say)
__device__ inline u64 tw_mask(u64 prev, u64 lo, u64 hi, int64_t w, int K, int64_t D, int64_t n, int tab) {
    const int64_t e = 2 * D - 1 < n ? 2 * D - 1 : n;
    return tw_window_clear(prev, lo, hi, K) & (tab ? (tw_range(w, 0, D - 1) | tw_range(w, D, n)) : (tw_range(w, 1, D - 1) | tw_range(w, D + 1, e)));
}
// per tile of KEY_TILE positions: the suffixes of the second sample that stay in the sort
__global__ __launch_bounds__(TB) void k_tw_count(const u64 *__restrict__ stop, int64_t nwords, int K, int64_t D, int64_t n, u32 *__restrict__ tilecnt, int64_t ntiles, int tab) {
    const int64_t w = (int64_t)blockIdx.x * TB + threadIdx.x;
    u32 c = 0;
    if (w < nwords) {
        const u64 prev = w > 0 ? stop[w - 1] : ~0ull, lo = stop[w], hi = w + 1 < nwords ? stop[w + 1] : ~0ull;
        c = (u32)__popcll(tw_range(w, D, n) & ~tw_mask(prev, lo, hi, w, K, D, n, tab));
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) c += __shfl_down(c, d, 16);      // KEY_TILE / 64 = 16 words per tile
    const int64_t tile = w >> 4;
    if ((threadIdx.x & 15) == 0 && tile < ntiles) tilecnt[tile] = c;
}
// layout of the key's upper bits on the fused path: [0, bits) the sort key; [nd_shift, nd_shift + nd_bits) nd, all ones = not
// known; bit nd_shift + nd_bits: the suffix is smaller than its twin; [at_shift, at_shift + at_bits) first stop among the K
// symbols, all ones = none; [56, 64) the byte in front of the suffix
struct KeyLayout { int at_shift, at_bits, nd_shift, nd_bits; u64 sortmask; };
// The sort key itself: the K symbols in FIELDS of g digits -- a field is its g codes read as a base-radix number, in fb = the bits of
// radix^g - 1; the last field holds the gl <= g symbols that are left, in fbl bits.  Five-letter DNA: three digits in seven bits
// (125 of 128 values), seventeen symbols in 5 x 7 + 5 = 40 bits -- what one base-5 number of seventeen digits needs, too.  One number
// was what the first key used to be; taking it apart again costs divisions: the common prefix of two neighbouring keys (a head's LCP,
// k_heads_publish_tc) took ten multiply-high steps per head, most of that kernel's 470 vector instructions per entry, and rolling a
// 40-bit number along the text most of k_init_keys'.  With fields the first differing field is a count of leading zeros of a ^ b, the
// digits inside it two multiplies, and a key is its fields' values side by side: k_init_keys makes every position's field value once
// (in LDS) and a key is nf reads of them.  The order of the keys is the order of the symbol strings either way.
struct KeyPack { int g, fb, nf, gl, fbl, bits; u32 rad, fmask, lmask, lscale /* radix^(g - gl) */, fdiv /* t / fb = (t * fdiv) >> 16, t < 64 */;
                 u32 lastmul /* the first gl digits of a field value: (f * lastmul) >> 20 */, dmul[4]; };      // f / radix^(g - i) = (f * dmul[i - 1]) >> 20 for f < radix^g (i = 1 .. g - 1: a field's first i digits)
constexpr int KP_MAXG = 5;
constexpr u32 KP_MAXFIELD = 4096;      // radix^g <= this: the reciprocals are exact (checked when the layout is chosen) and a field fits 16 bits
// number of leading symbols (of K) two different keys share
__device__ inline u32 key_common_digits(u64 x, u64 y, const KeyPack &kp) {
    const u64 d = x ^ y;
    if (d == 0ull) return (u32)((kp.nf - 1) * kp.g + kp.gl);
    const int top = 63 - __builtin_clzll(d);
    const bool last = top < kp.fbl;
    const u32 f = last ? (u32)(kp.nf - 1) : ((u32)(kp.bits - 1 - top) * kp.fdiv) >> 16;
    const int sh = last ? 0 : kp.bits - (int)(f + 1u) * kp.fb;
    const u32 fm = last ? kp.lmask : kp.fmask, sc = last ? kp.lscale : 1u;
    const u32 a = __umul24((u32)(x >> sh) & fm, sc), b = __umul24((u32)(y >> sh) & fm, sc);
    u32 c = f * (u32)kp.g;
#pragma unroll
    for (int i = 1; i < KP_MAXG; i++)
        if (i < kp.g) c += ((a * kp.dmul[i - 1]) >> 20) == ((b * kp.dmul[i - 1]) >> 20) ? 1u : 0u;
    return c;
}
constexpr int ND_WORDS = 40;      // the words of the diagonal bit arrays a block of k_init_keys stages: KEY_TILE / 64 + up to 1536 positions ahead
// ---- k_init_keys ----
// A workgroup takes a tile of KEY_TILE positions:
//   * the text as words, translated to codes in LDS; which codes are stops ('$', 'N', past the end) as one bit per position (eight threads'
//     four bits joined by three DPP moves): the first stop among a key's K symbols is a funnel shift and a count of trailing zeros;
//   * every position's field value (its g codes) and the value of its first gl codes, once, as 16-bit words in LDS;
//   * a key = the fields at k, k + g, k + 2 g, ... shifted together; four positions in a row per thread.  The hint's marked position,
//     exception and order bit stay in registers until the window has moved past the position;
//   * a tile of the second sample whose twins leave (tw_off) only makes the keys that stay: their positions as a list in LDS, one key
//     per thread -- at 1 % divergence 164 of 1024 positions, written next to each other.
// (One base-radix number per key, rolled along the text with 64-bit multiplies, K stop tests and the hint's bit reads per position: 551
// vector instructions per wave of 256 positions, SQ_INSTS_VALU -- 1.75 of the kernel's 2.8 ms at 2 x 250 Mbp were its instructions.)
constexpr int IK_AHEAD = 96;                             // codes staged behind the tile: K - 1 + g - 1 <= 68, rounded to whole groups of eight words
constexpr int IK_WORDS = (KEY_TILE + IK_AHEAD) / 4;      // code words
constexpr int IK_FPOS = KEY_TILE + 64;                   // positions that get a field value
__device__ inline u32 ik_first_stop(const u32 *s_sc, int k, int K, u32 at_none) {
    const int w = k >> 5, sh = k & 31;
    if (K <= 32) {
        const u32 win = __funnelshift_r(s_sc[w], s_sc[w + 1], (u32)sh) & (K >= 32 ? ~0u : (1u << K) - 1u);
        return win ? (u32)__builtin_ctz(win) : at_none;
    }
    u64 win = (((u64)s_sc[w + 1] << 32) | s_sc[w]) >> sh;
    if (sh) win |= (u64)s_sc[w + 2] << (64 - sh);
    win &= K >= 64 ? ~0ull : (1ull << K) - 1ull;
    return win ? (u32)__builtin_ctzll(win) : at_none;
}
// a key's fields side by side as two 32-bit halves: the whole fields from bit kp.bits down, the last one at bit 0.  The shifts are the same for
// every thread -- which half a field goes to is decided on the scalar side, a field costs a shift-or per key
template <int G, int N> __device__ inline void ik_fields(const volatile uint16_t *s_tf, int p, const KeyPack &kp, u32 (&lo)[N], u32 (&hi)[N]) {
#pragma unroll
    for (int r = 0; r < N; r++) lo[r] = hi[r] = 0u;
    int sh = kp.bits;
    for (int f = 0; f + 1 < kp.nf; f++, p += G) {
        sh -= kp.fb;
        u32 v[N];
#pragma unroll
        for (int r = 0; r < N; r++) v[r] = s_tf[p + r];
        if (sh >= 32) {
#pragma unroll
            for (int r = 0; r < N; r++) hi[r] |= v[r] << (sh - 32);
        } else {
#pragma unroll
            for (int r = 0; r < N; r++) lo[r] |= v[r] << sh;
            if (sh + kp.fb > 32) {
#pragma unroll
                for (int r = 0; r < N; r++) hi[r] |= v[r] >> (32 - sh);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < N; r++) lo[r] |= ((u32)s_tf[p + r] * kp.lastmul) >> 20;      // the first gl of the g codes
}
// what a thread brings in for a tile ahead of time: its text word(s) and, for the first threads, the words of the diagonal bit arrays
struct IkFetch { u32 raw0, raw1, pb; u64 ds, de, dl; };
__device__ inline u32 ik_text_word(const uint8_t *__restrict__ T, int64_t n, int64_t i, bool whole) {
    if (whole) return *reinterpret_cast<const u32 *>(T + i);
    u32 raw = 0;
    for (int r = 0; r < 4; r++) if (i + r < n) raw |= (u32)T[i + r] << (8 * r);
    return raw;
}
__device__ inline void ik_fetch(IkFetch &f, const uint8_t *__restrict__ T, int64_t n, const DiagBits &dg, bool hint, int64_t tile, bool aligned) {
    const int64_t base = tile * KEY_TILE;
    const bool whole = aligned && base + KEY_TILE + IK_AHEAD <= n;      // every staged word lies inside the text
    f.raw0 = ik_text_word(T, n, base + 4 * (int64_t)threadIdx.x, whole);
    f.raw1 = (int)threadIdx.x < IK_WORDS - TB ? ik_text_word(T, n, base + 4 * (int64_t)(TB + threadIdx.x), whole) : 0u;
    f.pb = (threadIdx.x == 0 && base > 0) ? (u32)T[base - 1] : (u32)'$';      // the byte in front of the tile ('$' in front of the text)
    f.ds = ~0ull; f.de = ~0ull; f.dl = 0ull;
    if (hint && (int)threadIdx.x <= ND_WORDS) {      // words base / 64 - 1 ... : thread 0 brings the word in front of the tile
        const int64_t w = base / 64 - 1 + threadIdx.x, nwords = (n + 63) / 64;
        if (w >= 0 && w < nwords) { f.ds = dg.stop[w]; if (threadIdx.x) { f.de = dg.exc[w]; f.dl = dg.lt[w]; } }
    }
}
// The kernel is persistent: a workgroup takes tiles blockIdx.x, blockIdx.x + gridDim.x, ... and loads the next one's words before it works on
// the current one.  One workgroup per tile spent its life waiting -- the table, then the text, then the stores, each a trip to memory with
// eight workgroups of four waves per CU to hide it: 1.4 ms at 2 x 250 Mbp with the computation switched off, 1.75 with the stores switched off.
// (Taking the tiles alternately from the two halves of the text -- second-sample tiles, few keys and few stores, between first-sample ones -- cost 0.15 ms.)
template <int G>
__global__ __launch_bounds__(TB) void k_init_keys(const uint8_t *__restrict__ T, int64_t n, const uint8_t *__restrict__ lut, int K, KeyPack kp,
                                                  u64 *__restrict__ keys, sav_t *__restrict__ vals, int pay, u32 stop0, u32 stop1,
                                                  KeyLayout ly, DiagBits dg, const u32 *__restrict__ tw_off /* != NULL: twins leave (k_tw_count's scan) */, int64_t ntiles) {
    __shared__ u32 code32[IK_WORDS + 2];
    __shared__ u32 s_sc[IK_WORDS / 8 + 3];      // bit k: code[k] is a stop
    __shared__ uint8_t slut[256], sstop[256];
    __shared__ __align__(8) uint16_t s_tf[IK_FPOS];      // the g codes from k on as a number
    __shared__ u64 s_stop0[ND_WORDS + 1], s_exc[ND_WORDS], s_lt[ND_WORDS];      // (s_stop0[0] = the word in front of the tile)
    __shared__ u64 s_tw[KEY_TILE / 64], s_keep[KEY_TILE / 64];
    __shared__ u32 s_kpre[KEY_TILE / 64 + 1];
    __shared__ uint16_t s_list[KEY_TILE];
    __shared__ u32 s_nm[64];      // the first marked position at or behind the start of word w: position << 2 | order bit << 1 | exception (all ones: none staged)
    __shared__ u32 s_raw32[KEY_TILE / 4 + 1];      // the tile's bytes as words from word 1 on, the byte in front of the tile the last byte of word 0: T[base + k - 1] = s_raw[k + 3]
    u64 *const s_stop = s_stop0 + 1;
    uint8_t *const s_raw = reinterpret_cast<uint8_t *>(s_raw32);
    const bool hint = ly.nd_bits > 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(T) & 3u) == 0;
    auto tile_of = [&](int64_t x) { return x; };
    IkFetch fx;
    int64_t x = blockIdx.x;
    if (x < ntiles) ik_fetch(fx, T, n, dg, hint, tile_of(x), aligned);
    {
        const u32 c = lut[threadIdx.x];
        slut[threadIdx.x] = (uint8_t)c;
        sstop[threadIdx.x] = ((c == stop0) | (c == stop1) | (c == 0u)) ? 1 : 0;      // (byte 0 -- what is read past the end -- has code 0)
        if (threadIdx.x < 2) code32[IK_WORDS + threadIdx.x] = 0u;
        if (threadIdx.x < 3) s_sc[IK_WORDS / 8 + threadIdx.x] = ~0u;
    }
    __syncthreads();
    const u32 at_none = (1u << ly.at_bits) - 1u;
    const int nd_none = hint ? (int)((1u << ly.nd_bits) - 1u) : 0;
    for (; x < ntiles; x += gridDim.x) {
    const int64_t tile = tile_of(x);
    const int64_t base = tile * KEY_TILE;
    // the words brought in ahead of time -> codes and stop bits in LDS
#pragma unroll
    for (int turn = 0; turn < 2; turn++) {      // (the second turn: the first 24 threads, whole groups of eight)
        const int wq = (int)threadIdx.x + turn * TB;
        if (wq >= IK_WORDS) break;
        const u32 raw = turn ? fx.raw1 : fx.raw0;
        u32 c4 = 0, sb = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const u32 b = (raw >> (8 * r)) & 0xffu;
            c4 |= (u32)slut[b] << (8 * r);
            sb |= (u32)sstop[b] << r;
        }
        code32[wq] = c4;
        if (turn == 0) s_raw32[1 + wq] = raw;
        int v = (int)(sb << (4 * (wq & 7)));
        v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);       // quad_perm [1,0,3,2]
        v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
        v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);      // row_half_mirror: the other quad of the eight
        if ((wq & 7) == 0) s_sc[wq >> 3] = (u32)v;
    }
    if (threadIdx.x == 0) s_raw[3] = (uint8_t)fx.pb;
    if (hint && (int)threadIdx.x <= ND_WORDS) {
        s_stop0[threadIdx.x] = fx.ds;
        if (threadIdx.x) { s_exc[threadIdx.x - 1] = fx.de; s_lt[threadIdx.x - 1] = fx.dl; }
    }
    if (x + gridDim.x < ntiles) ik_fetch(fx, T, n, dg, hint, tile_of(x + gridDim.x), aligned);      // in flight while this tile is worked on
    __syncthreads();
    // field values: a thread takes four positions in a row -- their codes are two words
    for (int wq = threadIdx.x; wq < IK_FPOS / 4; wq += TB) {
        const u64 cw = ((u64)code32[wq + 1] << 32) | code32[wq];
        u32 f4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32 h = 0;
#pragma unroll
            for (int e = 0; e < G; e++) h = __umul24(h, kp.rad) + ((u32)(cw >> (8 * (j + e))) & 0xffu);
            f4[j] = h;
        }
        *reinterpret_cast<uint2 *>(s_tf + 4 * wq) = make_uint2(f4[0] | (f4[1] << 16), f4[2] | (f4[3] << 16));
    }
    if (hint && (threadIdx.x >> 6) == 1) {      // the second wave: every word's first mark, then the nearest one at or behind each word (a minimum towards the front)
        const int w = (int)threadIdx.x - 64;
        u32 v = ~0u;
        if (w < ND_WORDS && s_stop[w] != 0ull) {
            const int b = __builtin_ctzll(s_stop[w]);
            v = ((u32)(64 * w + b) << 2) | ((u32)((s_lt[w] >> b) & 1ull) << 1) | (u32)((s_exc[w] >> b) & 1ull);
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_down(v, d, 64); if (w + d < 64 && o < v) v = o; }
        s_nm[w] = v;
    }
    if (tw_off && threadIdx.x < KEY_TILE / 64) {      // sixteen lanes: a word each, the kept positions in front of it by a scan among them
        const int xw = threadIdx.x;
        const int64_t w = base / 64 + xw;
        const u64 t = tw_mask(s_stop0[xw], s_stop[xw], s_stop[xw + 1], w, K, dg.D, n, dg.tab);
        const u64 keep = tw_range(w, dg.D, n) & ~t;
        s_tw[xw] = t; s_keep[xw] = keep;
        const u32 c = (u32)__popcll(keep);
        u32 inc = c;
#pragma unroll
        for (int d = 1; d < KEY_TILE / 64; d <<= 1) { const u32 up = __shfl_up(inc, d, KEY_TILE / 64); if (xw >= d) inc += up; }
        s_kpre[xw] = inc - c;
        if (xw == KEY_TILE / 64 - 1) s_kpre[KEY_TILE / 64] = inc;
    }
    __syncthreads();
    if (tw_off && base >= dg.D) {
        // ---- a tile of the second sample: the keys of the suffixes that stay ----
        {
            const int t = (int)threadIdx.x, xw = t >> 4, b = (t & 15) * 4;
            const u64 kw = s_keep[xw];
            u32 nib = (u32)(kw >> b) & 15u;
            if (nib) {
                u32 at = s_kpre[xw] + (u32)__popcll(kw & ((1ull << b) - 1ull));
                while (nib) { s_list[at++] = (uint16_t)(4 * t + __builtin_ctz(nib)); nib &= nib - 1u; }
            }
        }
        __syncthreads();
        const u32 total = s_kpre[KEY_TILE / 64];
        const int64_t obase = dg.D + (int64_t)tw_off[tile];
        for (u32 j = threadIdx.x; j < total; j += TB) {
            const int k = (int)s_list[j];
            const int64_t i = base + k;
            u32 lo[1], hi[1];
            ik_fields<G, 1>(s_tf, k, kp, lo, hi);
            u64 key = ((u64)hi[0] << 32) | lo[0];
            if (pay) key |= ((u64)s_raw[k + 3] << 56) | ((u64)ik_first_stop(s_sc, k, K, at_none) << ly.at_shift);
            if (hint) {
                const int wi = k >> 6, sft = k & 63;
                const u64 A = s_stop[wi] >> sft;
                const u32 nx = s_nm[wi + 1];
                const bool has = A != 0ull;
                const int d = has ? __builtin_ctzll(A) : (int)(nx >> 2) - k;      // to the next marked position: in this word / the first one of the words behind it
                const int at = (sft + d) & 63;
                const u32 f = has ? ((u32)((s_lt[wi] >> at) & 1ull) << 1) | (u32)((s_exc[wi] >> at) & 1ull) : nx & 3u;
                const bool known = (d < nd_none) & !(f & 1u);
                key |= (u64)(known ? (u32)d | ((f >> 1) << ly.nd_bits) : (u32)nd_none) << ly.nd_shift;
            }
            keys[obase + j] = key; vals[obase + j] = (sav_t)i;
        }
    } else {
    // ---- every position of the tile: four in a row per thread ----
    // (a position per thread and four turns -- 16-bit field reads in different banks, a wave's stores one run -- took 30 % more instructions:
    // what the four share -- the word's marks, the stop window, the byte in front -- is most of the work)
    constexpr int PER = KEY_TILE / TB;
    static_assert(PER == 4, "four bytes in front of four positions");
    const int k0 = (int)threadIdx.x * PER;
    const int64_t i0 = base + k0;
    u64 okey[PER];
    {
        u32 lo[PER], hi[PER];
        ik_fields<G, PER>(s_tf, k0, kp, lo, hi);
#pragma unroll
        for (int r = 0; r < PER; r++) okey[r] = ((u64)hi[r] << 32) | lo[r];
    }
    if (pay) {
        // bits 56..63 (above everything the sort looks at): the byte in front of the suffix -- the BWT byte of its rank travels with the key
        // instead of being gathered from the text at the end ('$' for position 0).  Below it: where the common prefix of this suffix
        // with anything ends at the latest (the first stop among its K symbols) -- k_heads and the text round read it off the key
        const u32 prevb = (u32)s_raw[k0 + 3] | (s_raw32[threadIdx.x + 1] << 8);      // T[i0 - 1 .. i0 + 2]: the last byte of the word in front, three of the thread's own
        // (no stop among the K + 3 codes from k0 on -- nearly every thread of a genome: none for all four)
        const bool clear = K <= 29 && (__funnelshift_r(s_sc[k0 >> 5], s_sc[(k0 >> 5) + 1], (u32)(k0 & 31)) & ((1u << (K + 3)) - 1u)) == 0u;
#pragma unroll
        for (int r = 0; r < PER; r++)
            okey[r] |= ((u64)((prevb >> (8 * r)) & 0xffu) << 56) | ((u64)(clear ? at_none : ik_first_stop(s_sc, k0 + r, K, at_none)) << ly.at_shift);
    }
    if (hint) {
        // The next marked position of each of the four, without a branch (a wave covers 256 positions: at 1 % divergence some lane of it always
        // has a mark among its own four, and a branch per case made every wave run every case -- 179 of 443 instructions): the word's marks from
        // k0 on, the first one from the fourth position on (or the first one of the words behind), and the three bits in front of it
        const int wi = k0 >> 6, sft = k0 & 63;
        const u64 A = s_stop[wi] >> sft, E = s_exc[wi] >> sft, Lt = s_lt[wi] >> sft;
        const u32 nx = s_nm[wi + 1];
        const u64 Ah = A >> 3;
        const bool hh = Ah != 0ull;
        const int dh = hh ? 3 + __builtin_ctzll(Ah) : (int)(nx >> 2) - k0;      // relative to k0
        const u32 fh = hh ? ((u32)((Lt >> (dh & 63)) & 1ull) << 1) | (u32)((E >> (dh & 63)) & 1ull) : nx & 3u;
        const u32 a3 = (u32)A & 7u, e3 = (u32)E & 7u, l3 = (u32)Lt & 7u;
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const u32 t = a3 >> r;
            const int m = t ? r + __builtin_ctz(t) : dh;
            const u32 f = t ? (((l3 >> m) & 1u) << 1) | ((e3 >> m) & 1u) : fh;
            const int nd = m - r;
            const bool known = (nd < nd_none) & !(f & 1u);
            okey[r] |= (u64)(known ? (u32)nd | ((f >> 1) << ly.nd_bits) : (u32)nd_none) << ly.nd_shift;
        }
    }
    if (!tw_off || i0 + PER <= dg.D) {
        // where they are (the first sample flagged when its twin leaves)
        const u32 twn = tw_off ? (u32)(s_tw[k0 >> 6] >> (k0 & 63)) & 15u : 0u;
        sav_t oval[PER];
#pragma unroll
        for (int r = 0; r < PER; r++) oval[r] = (sav_t)(i0 + r) | (((twn >> r) & 1u) ? TW_FLAG : (sav_t)0);
        if (i0 + PER <= n) {
            ulonglong2 *kd2 = reinterpret_cast<ulonglong2 *>(keys + i0);
            kd2[0] = make_ulonglong2(okey[0], okey[1]); kd2[1] = make_ulonglong2(okey[2], okey[3]);
            if constexpr (sizeof(sav_t) == 4) *reinterpret_cast<uint4 *>(vals + i0) = make_uint4((u32)oval[0], (u32)oval[1], (u32)oval[2], (u32)oval[3]);
            else {
                ulonglong2 *vd2 = reinterpret_cast<ulonglong2 *>(vals + i0);
                vd2[0] = make_ulonglong2((u64)oval[0], (u64)oval[1]); vd2[1] = make_ulonglong2((u64)oval[2], (u64)oval[3]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < PER; r++) if (i0 + r < n) { keys[i0 + r] = okey[r]; vals[i0 + r] = oval[r]; }
        }
    } else {
        // the tile the second sample starts in
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int k = k0 + r;
            const int64_t i = base + k;
            if (i >= n) continue;
            const int xw = k >> 6, b = k & 63;
            const bool twin = (s_tw[xw] >> b) & 1ull;
            if (i < dg.D) { keys[i] = okey[r]; vals[i] = (sav_t)i | (twin ? TW_FLAG : (sav_t)0); }
            else if (!twin) {
                // what stays of the second sample behind the first, in text order
                const int64_t o = dg.D + (int64_t)tw_off[tile] + s_kpre[xw] + (u32)__popcll(s_keep[xw] & ((b == 0) ? 0ull : (~0ull >> (64 - b))));
                keys[o] = okey[r]; vals[o] = (sav_t)i;
            }
        }
    }
    }
    __syncthreads();      // (the next tile's codes overwrite what this one still reads)
    }
}

// ---- group heads / ranks ------------------------------------------------------
// head[j] = 1 if sorted key j starts a new group; seed[j] = head ? j : 0
// LCP != NULL (the fused path of rv_build_sa): a head's LCP with its predecessor in the suffix array -- whichever member of the
// group in front ends up last, it shares that group's K symbols -- is the common prefix of the two keys, cut at the first
// '$' / 'N' (interface.c:97-114): the digits of both keys, least significant first, by multiply-high division.
constexpr int HINT_K = 16;      // samples the diagonal hint knows (the first ones)
// D: diagonal of the second sample against the first (nsep[0] + 1).  ns / sep / Ds: samples, their separators and every sample's
// diagonal against the FIRST sample (Ds[s] = nsep[s-1] + 1, where sample s starts when every sample is one sequence)
struct KeyDigits { int K; u32 stop0, stop1; KeyPack kp; KeyLayout ly; int64_t D; int ns; int64_t sep[HINT_K - 1], Ds[HINT_K];
                   const int32_t *dtab; };      // dtab != NULL: two samples on piecewise diagonals (k_diag_bits_tab): partner(y) = y +- (D + dtab[y >> 4])
// first position among a key's K symbols that holds a stop ('$', 'N', past the end), or 0xFFFFFFFF: k_init_keys left it in bits 48..55
__device__ inline u32 key_first_stop(u64 key, const KeyDigits &kd) {
    const u32 none = (1u << kd.ly.at_bits) - 1u;
    const u32 at = (u32)(key >> kd.ly.at_shift) & none;
    return at == none ? 0xFFFFFFFFu : at;
}
// the diagonal hint of a key (k_init_keys): *nd = symbols its suffix shares with its twin on the diagonal, *lt = it is the smaller one
__device__ inline bool key_hint(u64 key, const KeyDigits &kd, u32 *nd, bool *lt) {
    const u32 none = (1u << kd.ly.nd_bits) - 1u;
    const u32 v = (u32)(key >> kd.ly.nd_shift) & none;
    *nd = v; *lt = (key >> (kd.ly.nd_shift + kd.ly.nd_bits)) & 1ull;
    return kd.ly.nd_bits > 0 && v != none;
}
// The diagonal hint for any number of samples: a suffix of sample s >= 1 carries how far it agrees with its homologue in the FIRST
// sample (position - Ds[s]) and whether it is the smaller of the two; a suffix of the first sample carries the same against its
// homologue in the second.  Two suffixes with the same homologue in the first sample ("base") are ordered from that alone:
//   against the base itself: the variant's own hint;
//   two variants on different sides of the base: the smaller side first, common prefix = the shorter agreement;
//   two variants on the same side: the one that leaves the base EARLIER is the smaller one below the base and the larger one
//   above it (the other still spells the base there); equal agreement: not known (both left the base at the same place).
// -> true when known: *c < 0: u is the smaller suffix; *l = their common prefix (no stop inside: an exception ends every hint)
__device__ inline int hint_sample(const KeyDigits &kd, int64_t p) {
    int s = 0;
    for (int q = 0; q < kd.ns - 1; q++) s += kd.sep[q] < p ? 1 : 0;
    return s;
}
// (linked: the caller knows u and v to be partners -- a flagged suffix and the twin made from it: the table is not read again)
__device__ inline bool hint_cmp(const KeyDigits &kd, int64_t u, u64 key_u, int64_t v, u64 key_v, int *c, u32 *l, bool linked = false) {
    if (kd.ly.nd_bits <= 0) return false;
    const int su = hint_sample(kd, u), sv = hint_sample(kd, v);
    if (su >= HINT_K || sv >= HINT_K) return false;
    if (kd.dtab) {      // piecewise diagonals: the key of the second sample's suffix carries the hint only when it is linked (k_diag_bits_tab)
        if (su == sv) return false;
        if (!linked) {
            const int64_t a = su == 0 ? u : v, b = su == 0 ? v : u;
            const int32_t dd = kd.dtab[b >> DT_SHIFT];
            if (dd == DT_NONE || b - kd.D - (int64_t)dd != a) return false;
        }
    } else {
    const int64_t bu = u - (su ? kd.Ds[su] : 0), bv = v - (sv ? kd.Ds[sv] : 0);
    if (bu != bv || su == sv) return false;
    }
    u32 au = 0, av = 0; bool ltu = false, ltv = false;
    const bool ku = su ? key_hint(key_u, kd, &au, &ltu) : false, kv = sv ? key_hint(key_v, kd, &av, &ltv) : false;
    if (su == 0) { if (!kv) return false; *c = ltv ? 1 : -1; *l = av; return true; }
    if (sv == 0) { if (!ku) return false; *c = ltu ? -1 : 1; *l = au; return true; }
    if (!ku || !kv) return false;
    if (ltu != ltv) { *c = ltu ? -1 : 1; *l = au < av ? au : av; return true; }
    if (au == av) return false;
    *l = au < av ? au : av;
    *c = ((au < av) == ltu) ? -1 : 1;      // below the base the earlier one is smaller, above it the later one
    return true;
}

// hint_cmp for exactly two samples (kd.ns == 2: what k_heads_publish_tc runs on) -- no walk over the separators
__device__ inline bool hint_cmp2(const KeyDigits &kd, int64_t u, u64 key_u, int64_t v, u64 key_v, int *c, u32 *l) {
    const bool su = u > kd.sep[0], sv = v > kd.sep[0];
    if (su == sv) return false;
    const int64_t a = su ? v : u, b = su ? u : v;      // a: the suffix of the first sample
    if (kd.dtab) {
        const int32_t dd = kd.dtab[b >> DT_SHIFT];
        if (dd == DT_NONE || b - kd.D - (int64_t)dd != a) return false;
    } else if (b - kd.D != a) return false;
    u32 nd = 0; bool lt = false;
    if (!key_hint(su ? key_u : key_v, kd, &nd, &lt)) return false;      // the second sample's suffix carries the pair's hint
    *c = (lt == su) ? -1 : 1; *l = nd;
    return true;
}

__global__ __launch_bounds__(TB) void k_heads(const u64 *__restrict__ keys, int64_t n, uint8_t *__restrict__ head, u32 *__restrict__ seed,
                                              lcp_t *__restrict__ LCP, KeyDigits kd, u32 *__restrict__ d_maxlcp) {
    const int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x;
    // the payload bits (56..63 the byte in front, 48..55 the first stop) only exist on the fused path; without it the key may
    // use all 64 bits (sigma = 3 and n > 2^28: 29 symbols in 58 bits) and is compared whole
    const u64 mask = LCP ? kd.ly.sortmask : ~0ull;
    const bool in = j < n;
    const u64 kb = in ? keys[j] & mask : 0ull, ka = (in && j > 0) ? keys[j - 1] & mask : 0ull;
    const bool hd = in && ((j == 0) || ka != kb);
    if (in) { head[j] = hd; seed[j] = hd ? (u32)j : 0u; }
    u32 l = 0;
    if (LCP && hd) {
        if (j > 0) {
            const u64 dm = kd.ly.sortmask;
            l = key_common_digits(ka & dm, kb & dm, kd.kp);
            const u32 st = key_first_stop(keys[j], kd);
            l = l < st ? l : st;
        }
        LCP[j] = (lcp_t)l;
    }
    // the largest LCP of the index also counts the heads' values (fused_put only sees the other group members): one guarded
    // atomic per wave -- the values are below K, so after the first few waves nothing passes the guard any more
    if (LCP && d_maxlcp) {
        const u32 wm = (u32)rv_wave_max_u64((u64)l);
        if ((threadIdx.x & 63) == 0 && wm > *reinterpret_cast<volatile u32 *>(d_maxlcp)) atomicMax(d_maxlcp, wm);
    }
}

// SA[j] = vals[j].  ISA (rank of every suffix's group) is only an intermediate of the doubling rounds and of the radix
// path for groups above MEDIUM_GROUP: related genomes finish in the text round without either, so the scatter
// ISA[SA[j]] = grp[j] (a random 4-byte write per position, 21 ms at n = 5e8) waits until something asks for it.
__global__ __launch_bounds__(TB) void k_publish0(const sav_t *__restrict__ vals, int64_t n, sa_t *__restrict__ SA,
                                                 const u64 *__restrict__ keys, uint8_t *__restrict__ BWT, sa_t side_sep,
                                                 KeyDigits kd, lcp_t *__restrict__ LCP, uint8_t *__restrict__ head, u32 *__restrict__ d_maxlcp, u32 *__restrict__ grp) {
    const int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x;
    u32 lmax = 0;
    // the workgroup's keys and suffixes with their neighbours (two in front, two behind) through LDS: every thread looks at five keys
    __shared__ u64 s_key[TB + 4];
    __shared__ sav_t s_val[TB + 2];
    const bool hint = BWT && kd.ly.nd_bits > 0;
    if (hint) {
        const int64_t j0 = (int64_t)blockIdx.x * TB;
        for (int k = threadIdx.x; k < TB + 4; k += TB) { const int64_t i = j0 - 2 + k; s_key[k] = (i >= 0 && i < n) ? keys[i] : 0ull; }
        for (int k = threadIdx.x; k < TB + 2; k += TB) { const int64_t i = j0 - 1 + k; s_val[k] = (i >= 0 && i < n) ? vals[i] : (sav_t)0; }
        __syncthreads();
    }
    if (j < n) {
        const int t = threadIdx.x;
        const sav_t s = hint ? s_val[t + 1] : vals[j];
        int64_t rank = j;
        const u64 key = hint ? s_key[t + 2] : (BWT ? keys[j] : 0ull);
        // (fused path, diagonal hint) a group of exactly two suffixes that are twins on the diagonal is finished right here, from the
        // keys: both take their final ranks, the second one gets its LCP and becomes a head, and neither enters the list of
        // not yet unique suffixes -- four in five groups at 2 x 250 Mbp, where the text round used to stream the whole array again
        if (hint) {
            const u64 mk = kd.ly.sortmask;
            const u64 k0 = key & mk;
            const u64 km2 = j >= 2 ? s_key[t] & mk : ~k0, km1 = j >= 1 ? s_key[t + 1] & mk : ~k0;
            const u64 kp1r = j + 1 < n ? s_key[t + 3] : ~key, kp2 = j + 2 < n ? s_key[t + 4] & mk : ~k0;
            const u64 km1r = j >= 1 ? s_key[t + 1] : ~key;
            const u64 kp1 = j + 1 < n ? kp1r & mk : ~k0;
            const bool first = (km1 != k0) & (kp1 == k0) & (kp2 != k0);
            const bool second = (km1 == k0) & (kp1 != k0) & (km2 != k0);
            if (first | second) {
                const sav_t ps = first ? s_val[t + 2] : s_val[t];
                const u64 pkey = first ? kp1r : km1r;
                int c; u32 nd;
                if (hint_cmp(kd, (int64_t)s, key, (int64_t)ps, pkey, &c, &nd)) {      // I against my partner
                    const bool i_smaller = c < 0;
                    const int64_t base = first ? j : j - 1;
                    rank = base + (i_smaller ? 0 : 1);
                    if (rank != base) {
                        const u32 st = key_first_stop(key, kd);
                        const u32 l = nd < st ? nd : st;
                        LCP[rank] = (lcp_t)l;
                        lmax = l;
                        grp[rank] = (u32)rank;      // a group of its own from here on (k_isa_from_groups: its rank, should a doubling round ask)
                    }
                    if (second) head[j] = 1;
                }
            }
        }
        SA[rank] = (sa_t)s;
        // (fused path) final for every suffix that is alone in its group or finished above; members of other groups are rewritten where they are ordered
        if (BWT) BWT[rank] = (uint8_t)((u32)(key >> 56) | ((sa_t)s > side_sep ? RV_BWT_SIDE : 0u));
    }
    if (BWT && kd.ly.nd_bits > 0) {
        const u32 wm = (u32)rv_wave_max_u64((u64)lmax);
        if ((threadIdx.x & 63) == 0 && wm > __atomic_load_n(d_maxlcp, __ATOMIC_RELAXED)) atomicMax(d_maxlcp, wm);
    }
}
// k_heads and k_publish0 in one pass over the sorted (key, suffix) pairs, for the fused path with the diagonal hint (both read every
// key and its neighbours: 4 GB at n = 5e8 that need not be streamed twice).  head / seed / a head's LCP as in k_heads; SA / BWT and the
// twin pairs as in k_publish0 -- the second member of a finished pair is a head with its own seed, so the max-scan over the seeds
// gives it its own group rank.
constexpr int HP_ITEMS = 4;
__global__ __launch_bounds__(TB) void k_heads_publish(const u64 *__restrict__ keys, const sav_t *__restrict__ vals, int64_t n, uint8_t *__restrict__ head,
                                                      u32 *__restrict__ seed, lcp_t *__restrict__ LCP, sa_t *__restrict__ SA, uint8_t *__restrict__ BWT, sa_t side_sep,
                                                      KeyDigits kd, u32 *__restrict__ d_maxlcp, int twins) {
    // (HP_ITEMS x TB entries per workgroup, staged once: an entry per thread was 7.8 x 10^5 waves at 10 x 5 Mbp, 0.6 ms for 1.3 GB)
    __shared__ u64 s_key[HP_ITEMS * TB + 4];
    __shared__ sav_t s_val[HP_ITEMS * TB + 2];
    {
        const int64_t j0 = (int64_t)blockIdx.x * (HP_ITEMS * TB);
        for (int k = threadIdx.x; k < HP_ITEMS * TB + 4; k += TB) { const int64_t i = j0 - 2 + k; s_key[k] = (i >= 0 && i < n) ? keys[i] : 0ull; }
        for (int k = threadIdx.x; k < HP_ITEMS * TB + 2; k += TB) { const int64_t i = j0 - 1 + k; s_val[k] = (i >= 0 && i < n) ? vals[i] : (sav_t)0; }
        __syncthreads();
    }
    u32 lmax = 0;
#pragma unroll 1
    for (int it = 0; it < HP_ITEMS; it++) {
    const int64_t j = (int64_t)blockIdx.x * (HP_ITEMS * TB) + it * TB + threadIdx.x;
    if (j < n) {
        const int t = it * TB + (int)threadIdx.x;
        const sav_t s = s_val[t + 1];
        const u64 key = s_key[t + 2];
        const u64 mk = kd.ly.sortmask;
        const u64 k0 = key & mk;
        const u64 km2 = j >= 2 ? s_key[t] & mk : ~k0, km1 = j >= 1 ? s_key[t + 1] & mk : ~k0;
        const u64 kp1r = j + 1 < n ? s_key[t + 3] : ~key, kp2 = j + 2 < n ? s_key[t + 4] & mk : ~k0;
        const u64 km1r = j >= 1 ? s_key[t + 1] : ~key;
        const u64 kp1 = j + 1 < n ? kp1r & mk : ~k0;
        bool hd = (j == 0) | (km1 != k0);
        if (hd) {      // a head's LCP with its predecessor: the common prefix of the two keys (k_heads)
            u32 l = 0;
            if (j > 0) {
                l = key_common_digits(km1, k0, kd.kp);
                const u32 st = key_first_stop(key, kd);
                l = l < st ? l : st;
            }
            LCP[j] = (lcp_t)l;
            lmax = l > lmax ? l : lmax;
        }
        int64_t rank = j;
        const bool first = twins & (km1 != k0) & (kp1 == k0) & (kp2 != k0);
        const bool second = twins & (km1 == k0) & (kp1 != k0) & (km2 != k0);
        if (first | second) {
            const sav_t ps = first ? s_val[t + 2] : s_val[t];
            const u64 pkey = first ? kp1r : km1r;
            int c; u32 nd;
            if (hint_cmp(kd, (int64_t)s, key, (int64_t)ps, pkey, &c, &nd)) {      // I against my partner
                const int64_t base = first ? j : j - 1;
                rank = base + (c < 0 ? 0 : 1);
                if (rank != base) {
                    const u32 st = key_first_stop(key, kd);
                    const u32 l = nd < st ? nd : st;
                    LCP[rank] = (lcp_t)l;
                    lmax = l > lmax ? l : lmax;
                }
                if (second) hd = true;      // finished: a group of its own from here on
            }
        }
        head[j] = hd; seed[j] = hd ? (u32)j : 0u;
        SA[rank] = (sa_t)s;
        BWT[rank] = (uint8_t)((u32)(key >> 56) | ((sa_t)s > side_sep ? RV_BWT_SIDE : 0u));
    }
    }
    const u32 wm = (u32)rv_wave_max_u64((u64)lmax);
    if ((threadIdx.x & 63) == 0 && wm > __atomic_load_n(d_maxlcp, __ATOMIC_RELAXED)) atomicMax(d_maxlcp, wm);
}
// ---- the sorted list without the twins -> rank order with them (k_init_keys with tw_off) ----
constexpr int TC_PER = 4, TC_TILE = TB * TC_PER;      // k_heads_publish_tc: entries per thread / per workgroup
__global__ __launch_bounds__(TB) void k_tw_flags(const sav_t *__restrict__ vals, int64_t m, u32 *__restrict__ blockcnt) {
    __shared__ u32 wsum[TB / 64];
    const int64_t j0 = ((int64_t)blockIdx.x * TB + threadIdx.x) * TC_PER;
    u32 c = 0;
    if (j0 + TC_PER <= m) {
        sav_t v[TC_PER];
        __builtin_memcpy(v, vals + j0, sizeof v);
#pragma unroll
        for (int r = 0; r < TC_PER; r++) c += (v[r] & TW_FLAG) ? 1u : 0u;
    } else {
        for (int r = 0; r < TC_PER; r++) if (j0 + r < m && (vals[j0 + r] & TW_FLAG)) c++;
    }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// the key k_init_keys would have given the twin of a flagged suffix: same symbols, same byte in front, same agreement, the other side
__device__ inline u64 tw_twin_key(u64 key, const KeyDigits &kd) {
    const u32 none = (1u << kd.ly.nd_bits) - 1u;
    const bool known = ((u32)(key >> kd.ly.nd_shift) & none) != none;
    return known ? key ^ (1ull << (kd.ly.nd_shift + kd.ly.nd_bits)) : key;
}
// k_heads_publish over the list of m sorted pairs that lacks the twins: entry j goes to rank j + (flagged entries in front of it), a
// flagged entry's twin to the rank behind it.  A flagged suffix alone in its group and its twin are the pair k_heads_publish finishes
// from the keys; every other member of a group waits for the text round, which finds its key in kexp and its suffix in vexp (both in
// rank order, written for the unfinished only).  No group ranks: the compaction of the unfinished reads them off the head flags (k_cp_emit_g).
// (Laying a round's ranks out in LDS first and writing whole runs made no difference -- switching the writes off altogether takes 0.1 of
// 12 ms at 2 x 100 Mbp: the kernel waits for its keys, not for its stores.)
#ifdef RV_SA64
#define RV_TC_BOUNDS __launch_bounds__(TB)          // (eight-byte suffixes: 24.6 KB of LDS per workgroup, six workgroups per CU whatever the registers)
#else
#define RV_TC_BOUNDS __launch_bounds__(TB, 8)
#endif
__global__ RV_TC_BOUNDS void k_heads_publish_tc(const u64 *__restrict__ keys, const sav_t *__restrict__ vals, int64_t m, const u32 *__restrict__ blockoff,
                                                         uint8_t *__restrict__ head, lcp_t *__restrict__ LCP, sa_t *__restrict__ SA,
                                                         uint8_t *__restrict__ BWT, sa_t side_sep, KeyDigits kd, u32 *__restrict__ d_maxlcp, int twins,
                                                         u64 *__restrict__ kexp, sav_t *__restrict__ vexp) {
    // A thread takes four entries in a row, with the two keys in front of them and the two behind straight from memory (its neighbours'
    // loads hit the same lines): one entry per thread through an LDS tile was bound by its instruction count -- 5 ms for 9.5 GB.
    __shared__ u32 wsum[TB / 64];
    // The ranks of a workgroup's entries are one stretch [R0, R0 + entries + flagged entries): SA / LCP / BWT / head are laid out in LDS and leave as
    // whole lines.  Written from the threads' registers (four entries in a row each: every store instruction of a wave touched 64 lanes x 16-32 B
    // apart) every 64-byte line went to memory once per instruction -- 3.4 x 10^8 write requests, 20.9 GB for 5.4 GB of results at 2 x 250 Mbp
    // (TCC_EA0_WRREQ / _64B, profiles/r04_wrreq_c4.txt), the kernel at the copy ceiling of the node.  A slot that is not this workgroup's to write (a
    // pair across the tile border is finished by both sides, each writing its own final rank) keeps its sentinel and is left alone.
    constexpr int TC_SPAN = 2 * TC_TILE + 2;
    constexpr sa_t SA_NONE = (sa_t)~(sa_t)0;
    __shared__ sa_t o_sa[TC_SPAN];
    __shared__ uint16_t o_lcp[TC_SPAN];      // (a head's common digits, a twin's agreement: below 2^11.  Four bytes each the arrays were 20 560 bytes -- seven workgroups per CU, eighty bytes short of eight)
    __shared__ uint8_t o_bw[TC_SPAN], o_hd[TC_SPAN];
    for (int x = threadIdx.x; x < TC_SPAN; x += TB) { o_sa[x] = SA_NONE; o_lcp[x] = (uint16_t)0xFFFFu; }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t j0 = ((int64_t)blockIdx.x * TB + threadIdx.x) * TC_PER;
    u64 kk[TC_PER + 4];          // keys j0 - 2 .. j0 + 5
    sav_t vv[TC_PER + 2];        // payloads j0 - 1 .. j0 + 4
    if (j0 >= 2 && j0 + TC_PER + 2 <= m) {
        __builtin_memcpy(kk, keys + j0 - 2, sizeof kk);
        __builtin_memcpy(vv, vals + j0 - 1, sizeof vv);
    } else {
#pragma unroll
        for (int x = 0; x < TC_PER + 4; x++) { const int64_t i = j0 - 2 + x; kk[x] = (i >= 0 && i < m) ? keys[i] : 0ull; }
#pragma unroll
        for (int x = 0; x < TC_PER + 2; x++) { const int64_t i = j0 - 1 + x; vv[x] = (i >= 0 && i < m) ? vals[i] : (sav_t)0; }
    }
    u32 myflags = 0;
#pragma unroll
    for (int e = 0; e < TC_PER; e++) myflags += (j0 + e < m && (vv[e + 1] & TW_FLAG)) ? 1u : 0u;
    const u32 inc = rv_wave_incl_sum_u32(myflags);      // (DPP: six shuffles through the LDS crossbar were a sixth of a microsecond of every wave)
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 before = blockoff[blockIdx.x] + inc - myflags;
    for (int k = 0; k < wv; k++) before += wsum[k];
    const int64_t R0 = (int64_t)blockIdx.x * TC_TILE + (int64_t)blockoff[blockIdx.x] - 1;      // slot 0 = the rank in front of the stretch
    u32 span_flags = 0;
    for (int k = 0; k < TB / 64; k++) span_flags += wsum[k];
    // (ranks as slots of the workgroup's stretch -- 32-bit -- wherever they are only compared or used as LDS indices: rank = R0 + slot)
    auto put_sa = [&](int x, sa_t v, uint8_t bw) {
        if (x >= 0 && x < TC_SPAN) { o_sa[x] = v; o_bw[x] = bw; } else { SA[R0 + x] = v; BWT[R0 + x] = bw; }
    };
    auto put_lcp = [&](int x, u32 l) {
        if (x >= 0 && x < TC_SPAN && l < 0xFFFFu) o_lcp[x] = (uint16_t)l; else LCP[R0 + x] = (lcp_t)l;
    };
    auto put_head = [&](int x, bool v) { o_hd[x] = v ? 1 : 0; };      // (always a slot of this stretch)
    u64 *const kexpR = kexp + R0; sav_t *const vexpR = vexp + R0;
    u32 lmax = 0;
    const u64 mk = kd.ly.sortmask;
#pragma unroll
    for (int e = 0; e < TC_PER; e++) {
        const int64_t j = j0 + e;
        if (j >= m) break;
        const int r = (int)(j - (int64_t)blockIdx.x * TC_TILE) + (int)(before - blockoff[blockIdx.x]) + 1;      // the entry's slot
        const sav_t sv = vv[e + 1];
        const bool flagged = (sv & TW_FLAG) != 0;
        const sav_t s = sv & ~TW_FLAG;
        const u64 key = kk[e + 2];
        const u64 k0 = key & mk;
        const u64 km2 = j >= 2 ? kk[e] & mk : ~k0, km1 = j >= 1 ? kk[e + 1] & mk : ~k0;
        const u64 kp1r = j + 1 < m ? kk[e + 3] : ~key, kp2 = j + 2 < m ? kk[e + 4] & mk : ~k0;
        const u64 km1r = j >= 1 ? kk[e + 1] : ~key;
        const u64 kp1 = j + 1 < m ? kp1r & mk : ~k0;
        const bool f_m1 = j >= 1 && (vv[e] & TW_FLAG), f_p1 = j + 1 < m && (vv[e + 2] & TW_FLAG);
        bool hd = (j == 0) | (km1 != k0);
        if (hd) {      // a head's LCP with its predecessor: the common prefix of the two keys (k_heads)
            u32 l = 0;
            if (j > 0) {
                l = key_common_digits(km1, k0, kd.kp);
                const u32 st = key_first_stop(key, kd);
                l = l < st ? l : st;
            }
            put_lcp(r, l);
            lmax = l > lmax ? l : lmax;
        }
        const u32 pay = (u32)(key >> 56);
        if (flagged) {
            const sav_t q = s + (sav_t)kd.D + (kd.dtab ? (sav_t)(int64_t)kd.dtab[s >> DT_SHIFT] : (sav_t)0);      // (a flagged suffix is linked: its tile's diagonal)
            const u64 qkey = tw_twin_key(key, kd);
            const bool alone = hd & (kp1 != k0);
            // s against the twin made from it (hint_cmp of a first-sample suffix and its partner: the partner's hint is s' own with the
            // order bit turned round)
            u32 nd = 0; bool lt = false;
            const bool fin = twins && alone && key_hint(key, kd, &nd, &lt);
            const int c = lt ? -1 : 1;
            const int rs = r + ((fin && c >= 0) ? 1 : 0), rq = r + ((fin && c >= 0) ? 0 : 1);
            if (fin) {
                const u32 st = key_first_stop(key, kd);
                const u32 l = nd < st ? nd : st;
                put_lcp(r + 1, l);
                lmax = l > lmax ? l : lmax;
            } else { kexpR[r] = key; kexpR[r + 1] = qkey; vexpR[r] = s; vexpR[r + 1] = q; }
            put_head(r, hd); put_head(r + 1, fin);
            put_sa(rs, (sa_t)s, (uint8_t)(pay | ((sa_t)s > side_sep ? RV_BWT_SIDE : 0u)));
            put_sa(rq, (sa_t)q, (uint8_t)(pay | ((sa_t)q > side_sep ? RV_BWT_SIDE : 0u)));
            before++;
        } else {
            int rank = r;
            // a group of exactly two entries, neither flagged (a twin whose byte in front differs, two unrelated suffixes): k_heads_publish's pair
            const bool first = twins & (km1 != k0) & (kp1 == k0) & (kp2 != k0) & !f_p1;
            const bool second = twins & (km1 == k0) & (kp1 != k0) & (km2 != k0) & !f_m1;
            bool fin = hd & (kp1 != k0);      // alone in its group
            if (first | second) {
                const sav_t ps = (first ? vv[e + 2] : vv[e]) & ~TW_FLAG;
                const u64 pkey = first ? kp1r : km1r;
                int c; u32 nd;
                if (hint_cmp2(kd, (int64_t)s, key, (int64_t)ps, pkey, &c, &nd)) {      // I against my partner
                    const int base = first ? r : r - 1;
                    rank = base + (c < 0 ? 0 : 1);
                    if (rank != base) {
                        const u32 st = key_first_stop(key, kd);
                        const u32 l = nd < st ? nd : st;
                        put_lcp(rank, l);
                        lmax = l > lmax ? l : lmax;
                    }
                    if (second) hd = true;      // finished: a group of its own from here on
                    fin = true;
                }
            }
            if (!fin) { kexpR[r] = key; vexpR[r] = s; }
            put_head(r, hd);
            put_sa(rank, (sa_t)s, (uint8_t)(pay | ((sa_t)s > side_sep ? RV_BWT_SIDE : 0u)));
        }
    }
    __syncthreads();
    {
        const int64_t tile_j0 = (int64_t)blockIdx.x * TC_TILE;
        const int64_t ents = m - tile_j0 < (int64_t)TC_TILE ? m - tile_j0 : (int64_t)TC_TILE;
        const int own = (int)(ents + (int64_t)span_flags);      // slots 1 .. own are this workgroup's ranks; 0 and own + 1 its neighbours'
        for (int x = threadIdx.x; x < own + 2; x += TB) {
            const int64_t rank = R0 + x;
            if (rank < 0) continue;
            const sa_t v = o_sa[x];
            if (v != SA_NONE) { SA[rank] = v; BWT[rank] = o_bw[x]; }
            const u32 l = o_lcp[x];
            if (l != 0xFFFFu) LCP[rank] = (lcp_t)l;
            if (x >= 1 && x <= own) head[rank] = o_hd[x];
        }
    }
    const u32 wm = (u32)rv_wave_max_u64((u64)lmax);
    if ((threadIdx.x & 63) == 0 && wm > __atomic_load_n(d_maxlcp, __ATOMIC_RELAXED)) atomicMax(d_maxlcp, wm);
}
// the start of every rank's group read off the head flags: what the max-scan over the seeds gave (a 4-byte word per rank written,
// scanned and read again) for the compaction of the unfinished ranks of round 0 -- per tile the last head, a scan over the tiles, and
// inside the tile the nearest head in front of a rank from the waves' ballots
__global__ __launch_bounds__(TB) void k_cp_count_g(const uint8_t *__restrict__ head, int64_t n, u32 *__restrict__ tilecnt, u32 *__restrict__ tilelast);
__global__ __launch_bounds__(TB) void k_cp_emit_g(const uint8_t *__restrict__ head, int64_t n, const u32 *__restrict__ tileoff, const u32 *__restrict__ tilelast,
                                                  const sav_t *__restrict__ suf_in, u32 *__restrict__ P, sav_t *__restrict__ S, u32 *__restrict__ G);
__global__ __launch_bounds__(TB) void k_isa_identity(const sa_t *__restrict__ SA, int64_t n, u32 *__restrict__ ISA) {
    const int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (j < n) ISA[SA[j]] = (u32)j;
}
__global__ __launch_bounds__(TB) void k_isa_list(const sa_t *__restrict__ SA, const u32 *__restrict__ P, const u32 *__restrict__ G, int64_t m, u32 *__restrict__ ISA) {
    const int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (q < m) ISA[SA[P[q]]] = G[q];
}
// Group ranks are rank ranges and every round only permutes suffixes inside their group, so (SA, grp of round 0) still
// describe round 0's ISA after the text round has reordered SA.
__global__ __launch_bounds__(TB) void k_isa_from_groups(const sa_t *__restrict__ SA, const u32 *__restrict__ grp, int64_t n, u32 *__restrict__ ISA) {
    const int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (j < n) ISA[SA[j]] = grp[j];
}

// ---- ordered compaction of the not-yet-unique suffixes -------------------------
// An element is finished when it is a group of its own: head[j] && head[j+1].
// Round 0 compacts from the full arrays (pos = j); later rounds compact the
// previous (already compacted) list.  Two passes: per-tile counts -> scan ->
// emit, so the order (and with it the group layout) is preserved.
constexpr int CP_ITEMS = 8;
constexpr int CP_TILE = TB * CP_ITEMS;

__device__ inline bool unsorted_at(const uint8_t *head, int64_t j, int64_t n) {
    return !(head[j] && (j + 1 == n || head[j + 1]));
}

__global__ __launch_bounds__(TB) void k_cp_count(const uint8_t *__restrict__ head, int64_t n, u32 *__restrict__ tilecnt) {
    __shared__ u32 wsum[TB / 64];
    // a thread counts CP_ITEMS consecutive entries from one 8-byte load (+ the byte behind them): counts only, so the order inside the
    // tile does not matter (k_cp_emit keeps its strided, order-preserving layout)
    static_assert(CP_ITEMS == 8, "eight head bytes per load");
    const int64_t j0 = (int64_t)blockIdx.x * CP_TILE + (int64_t)threadIdx.x * CP_ITEMS;
    u32 c = 0;
    if (j0 + CP_ITEMS < n) {
        const u64 hb = *reinterpret_cast<const u64 *>(head + j0);
        const u64 nx = (hb >> 8) | ((u64)head[j0 + CP_ITEMS] << 56);        // byte k = head[j + 1]
#pragma unroll
        for (int k = 0; k < CP_ITEMS; k++) c += !(((hb >> (8 * k)) & 0xFFu) && ((nx >> (8 * k)) & 0xFFu));
    } else {
        for (int64_t j = j0; j < j0 + CP_ITEMS && j < n; j++) c += unsorted_at(head, j, n);
    }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// emit: P[q] = position in SA, S[q] = suffix, G[q] = group rank.
// pos_in == nullptr means "position is j itself" (round 0).
__global__ __launch_bounds__(TB) void k_cp_emit(const uint8_t *__restrict__ head, int64_t n, const u32 *__restrict__ tileoff,
                                                const u32 *__restrict__ pos_in, const sav_t *__restrict__ suf_in, const u32 *__restrict__ grp_in,
                                                u32 *__restrict__ P, sav_t *__restrict__ S, u32 *__restrict__ G,
                                                const uint8_t *__restrict__ cls_in = nullptr, uint8_t *__restrict__ cls_out = nullptr, int cls_is = 0) {
    __shared__ u32 wbase[TB / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * CP_TILE;
    u32 run = tileoff[blockIdx.x];
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll 1
    for (int r = 0; r < CP_ITEMS; r++) {
        const int64_t j = base + (int64_t)r * TB + threadIdx.x;
        const bool f = (j < n) && unsorted_at(head, j, n);
        const u64 bal = __ballot(f);
        if (lane == 0) wbase[w] = (u32)__popcll(bal);
        __syncthreads();
        u32 before = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < TB / 64; k++) { const u32 c = wbase[k]; if (k < w) before += c; tot += c; }
        if (f) {
            const u32 q = run + before + (u32)__popcll(bal & lt);
            P[q] = pos_in ? pos_in[j] : (u32)j;
            S[q] = suf_in[j];
            G[q] = grp_in[j];
            if (cls_out) cls_out[q] = cls_in ? (cls_is ? (cls_in[j] == (uint8_t)cls_is) : cls_in[j]) : (uint8_t)0;      // (cls_is: the flag value that means "of the slow class")
        }
        run += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(TB) void k_cp_count_g(const uint8_t *__restrict__ head, int64_t n, u32 *__restrict__ tilecnt, u32 *__restrict__ tilelast) {
    __shared__ u32 wsum[TB / 64], wlast[TB / 64];
    static_assert(CP_ITEMS == 8, "eight head bytes per load");
    const int64_t j0 = (int64_t)blockIdx.x * CP_TILE + (int64_t)threadIdx.x * CP_ITEMS;
    u32 c = 0, last = 0;      // last = position of the last head among my entries + 1
    if (j0 + CP_ITEMS < n) {      // (k_cp_count's load: eight flags and the one behind them)
        const u64 hb = *reinterpret_cast<const u64 *>(head + j0);
        const u64 nx = (hb >> 8) | ((u64)head[j0 + CP_ITEMS] << 56);
#pragma unroll
        for (int k = 0; k < CP_ITEMS; k++) c += !(((hb >> (8 * k)) & 0xFFu) && ((nx >> (8 * k)) & 0xFFu));
        if (hb) last = (u32)(j0 + ((63 - __builtin_clzll(hb)) >> 3) + 1);
    } else {
        for (int64_t j = j0; j < j0 + CP_ITEMS && j < n; j++) { c += unsorted_at(head, j, n); last = head[j] ? (u32)(j + 1) : last; }
    }
    for (int d = 32; d >= 1; d >>= 1) { c += __shfl_down(c, d, 64); const u32 o = __shfl_down(last, d, 64); last = o > last ? o : last; }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = c; wlast[threadIdx.x >> 6] = last; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tilecnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        u32 l = wlast[0];
        for (int k = 1; k < TB / 64; k++) l = wlast[k] > l ? wlast[k] : l;
        tilelast[blockIdx.x] = l;
    }
}
// tilelast: inclusive maximum over the tiles (so tile t starts behind tilelast[t - 1]).  A thread takes eight flags in a row from one load
// (entry by entry, 256 at a time with two ballots and two barriers each, the kernel moved 0.6 TB/s).
__global__ __launch_bounds__(TB) void k_cp_emit_g(const uint8_t *__restrict__ head, int64_t n, const u32 *__restrict__ tileoff, const u32 *__restrict__ tilelast,
                                                  const sav_t *__restrict__ suf_in, u32 *__restrict__ P, sav_t *__restrict__ S, u32 *__restrict__ G) {
    __shared__ u32 wcnt[TB / 64], wlast[TB / 64];
    static_assert(CP_ITEMS == 8, "eight head bytes per load");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * CP_TILE + (int64_t)threadIdx.x * CP_ITEMS;
    u32 hm = 0, um = 0;      // bit k: entry j0 + k is a head / is not finished
    if (j0 + CP_ITEMS < n) {
        const u64 hb = *reinterpret_cast<const u64 *>(head + j0);
        const u64 nx = (hb >> 8) | ((u64)head[j0 + CP_ITEMS] << 56);
#pragma unroll
        for (int k = 0; k < CP_ITEMS; k++) {
            const bool h = (hb >> (8 * k)) & 0xFFu, hn = (nx >> (8 * k)) & 0xFFu;
            hm |= (u32)h << k; um |= (u32)!(h && hn) << k;
        }
    } else {
        for (int k = 0; k < CP_ITEMS; k++) { const int64_t j = j0 + k; if (j < n) { hm |= (u32)(head[j] != 0) << k; um |= (u32)unsorted_at(head, j, n) << k; } }
    }
    const u32 c = (u32)__popc(um);
    const u32 mylast = hm ? (u32)(j0 + (31 - __clz(hm)) + 1) : 0u;      // position of my last head + 1
    // exclusive prefix of the counts and running maximum of the last heads, over the wave
    u32 inc = c, lastinc = mylast;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d, 64), l = __shfl_up(lastinc, d, 64);
        if (lane >= d) { inc += t; lastinc = l > lastinc ? l : lastinc; }
    }
    if (lane == 63) { wcnt[w] = inc; wlast[w] = lastinc; }
    __syncthreads();
    u32 before = inc - c;
    u32 prevlast = __shfl_up(lastinc, 1, 64);
    if (lane == 0) prevlast = 0;
    u32 seen = blockIdx.x ? tilelast[blockIdx.x - 1] : 0u;
    for (int k = 0; k < w; k++) { before += wcnt[k]; seen = wlast[k] > seen ? wlast[k] : seen; }
    prevlast = prevlast > seen ? prevlast : seen;               // last head in front of my eight entries (position + 1)
    u32 q = tileoff[blockIdx.x] + before;
    u32 cur = prevlast;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; k++) {
        const int64_t j = j0 + k;
        if ((hm >> k) & 1u) cur = (u32)(j + 1);
        if ((um >> k) & 1u) {
            P[q] = (u32)j;
            S[q] = suf_in[j];
            G[q] = cur - 1u;
            q++;
        }
    }
}

// ---- refinement rounds --------------------------------------------------------
// ISA of the list's suffixes after a round: the start of their (new) group
// (oldG: the round's own group ranks -- an entry whose group did not change keeps its word: the doubling rounds below TEXT_LIM pass over
//  everything the text round left tied, a random write per entry and round for nothing)
__global__ __launch_bounds__(TB) void k_round_isa(const sav_t *__restrict__ S, const u32 *__restrict__ newG, int64_t m, u32 *__restrict__ ISA, const u32 *__restrict__ oldG = nullptr) {
    const int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (q >= m) return;
    const u32 g = newG[q];
    if (oldG && oldG[q] == g) return;
    ISA[S[q]] = g;
}

// ---- doubling round, small groups ------------------------------------------------
// In closely related genomes almost every group that is still unsorted after the
// first key has two to four members (a suffix and its twins in the other
// samples).  A group's members sit next to each other in the list (P[q] - G[q] is
// the offset inside the group), so the group's first thread can sort up to
// SMALL_GROUP members by the rank of suffix+h directly; only larger groups go
// through the radix sort.
constexpr int SMALL_GROUP = 8;

__device__ inline u32 key2_of(sav_t s, int64_t h, int64_t n, const u32 *__restrict__ ISA) {
    const int64_t s2 = (int64_t)s + h;
    return (s2 < n) ? ISA[s2] + 1u : 0u;
}

__global__ __launch_bounds__(TB) void k_round_small(sav_t *__restrict__ S, const u32 *__restrict__ G, const u32 *__restrict__ P, int64_t m, int64_t n, int64_t h,
                                                    const u32 *__restrict__ ISA, uint8_t *__restrict__ headq, uint8_t *__restrict__ bigflag,
                                                    sa_t *__restrict__ SA, const uint8_t *__restrict__ slow = nullptr) {
    const int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (q >= m) return;
    const u32 g = G[q];
    const u32 off = P[q] - g;
    // slow != NULL: only the entries it marks take part in this round (h is below what the others are known to share: their groups stand)
    if (slow && !slow[q]) { bigflag[q] = 0; headq[q] = off == 0; return; }
    const int64_t look = q + (SMALL_GROUP - (int64_t)off);
    const bool big = off >= (u32)SMALL_GROUP || (look < m && G[look] == g);
    bigflag[q] = big;
    if (big || off != 0) return;
    sav_t s[SMALL_GROUP]; u32 k[SMALL_GROUP];
    int size = 1;
    s[0] = S[q]; k[0] = key2_of(s[0], h, n, ISA);
#pragma unroll
    for (int j = 1; j < SMALL_GROUP; j++) {
        if (size == j && q + j < m && G[q + j] == g) { s[j] = S[q + j]; k[j] = key2_of(s[j], h, n, ISA); size = j + 1; }
    }
    // insertion sort by k (at most 8 keys; fully unrolled compare-exchange network would also do)
#pragma unroll
    for (int a = 1; a < SMALL_GROUP; a++) {
        if (a < size) {
            const sav_t cs = s[a]; const u32 ck = k[a];
            int b = a;
#pragma unroll
            for (int t = SMALL_GROUP - 1; t >= 1; t--) {
                if (t <= a && t == b && k[t - 1] > ck) { s[t] = s[t - 1]; k[t] = k[t - 1]; b = t - 1; }
            }
#pragma unroll
            for (int t = 0; t < SMALL_GROUP; t++) if (t == b) { s[t] = cs; k[t] = ck; }
        }
    }
#pragma unroll
    for (int j = 0; j < SMALL_GROUP; j++) {
        if (j < size) {
            S[q + j] = s[j];
            SA[(size_t)g + j] = (sa_t)s[j];
            headq[q + j] = (j == 0) || (k[j] != k[j - (j > 0)]);
        }
    }
}

__device__ inline u64 zero_bytes(u64 y) {   // 0x80 in every zero byte of y; exact at and below the lowest hit
    return (y - 0x0101010101010101ull) & ~y & 0x8080808080808080ull;
}
__device__ inline u64 load8(const uint8_t *p) {
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// First round: the members of a small group are told apart by comparing their texts directly (from symbol h on,
// big-endian words, at most TEXT_LIM bytes) instead of by the rank of suffix+h.  In closely related genomes a
// suffix and its twin agree for about 1/divergence symbols: doubling needs log2 of that many rounds over nearly
// the whole list, the direct comparison finishes them in one.  Members still equal after TEXT_LIM bytes (long
// repeats, identical inputs) stay one group and go on with the doubling rounds.
constexpr int TEXT_LIM = 4096;
// 32 bytes per side and step: the loop is a chain of dependent memory round trips (the next step starts when this
// one's compare is known), and a wave takes as many steps as its longest pair -- fewer, fatter steps.
// ---- 2-bit text for the text round -----------------------------------------------------------
// The round is bound by the sectors its comparisons fetch (wider byte steps made it slower).  DNA needs two bits per base:
// packed 32 bases to a 64-bit word (base i of a word in bits 2i, 2i+1; A, C, G, T = 0..3 keeps the byte order), one 40-byte
// read per side covers 128 bases where a 32-byte read of the text covers 32.  Everything that is not exactly A, C, G or T
// ('$', 'N', IUPAC codes, lower case, the zero padding) is an exception: one flag byte per 128-base block (4 MB for 5e8
// bases: cache resident); a comparison window that touches a flagged block goes on byte by byte on the text itself.
constexpr int PK_BLOCK = 128;
__global__ __launch_bounds__(TB) void k_pack2(const uint8_t *__restrict__ T, int64_t n, u64 *__restrict__ Tp, uint8_t *__restrict__ blk, int64_t nwords) {
    const int64_t w = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (w >= nwords) return;
    const int64_t p0 = w * 32;
    u64 word = 0; bool exc = false;
    if (p0 + 32 <= n) {
        // (eight bytes at a time: written byte by byte with ?: chains this compiled to 214 exec-mask branches, 0.49 ms for 0.5 GB)
        const uint4 *q4 = reinterpret_cast<const uint4 *>(T + p0);
        const uint4 lo = q4[0], hi = q4[1];
        const u64 v[4] = {(u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32), (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32)};
        u64 ok = 0x8080808080808080ull;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u64 m = swar_acgt(v[k]);
            ok &= m;
            // (a byte that is no base contributes whatever its bits give: the block is flagged, its window is compared on the text)
            word |= (u64)swar_code2(v[k] & (m | (m >> 1) | (m >> 2) | (m >> 3) | (m >> 4) | (m >> 5) | (m >> 6) | (m >> 7))) << (16 * k);
        }
        exc = ok != 0x8080808080808080ull;
    } else {
        exc = true;                                  // the tail (and everything behind the text)
        for (int j = 0; j < 32 && p0 + j < n; j++) {
            const u32 c = T[p0 + j];
            const u32 code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
            word |= (u64)code << (2 * j);
        }
    }
    Tp[w] = word;
    if (exc) blk[p0 / PK_BLOCK] = 1;
}
struct Packed { const u64 *Tp; const uint8_t *blk; };

// -> order of suffixes a, b (-1 / +1; 0 = equal for TEXT_LIM bytes); *lcp = their common prefix as compute_lcp counts it
// (interface.c:97-114: equal characters up to the first '$' / 'N' / end of text), valid when the result is not 0.  The
// compare starts at the suffixes' first byte although their first h symbols are known to be equal: a stop among those
// symbols ends the LCP, and the first 32-byte step covers them anyway.
template <int W>
__device__ inline int cmp_text(const uint8_t *__restrict__ T, sav_t a, sav_t b, u32 *lcp, int h0 = 0, u32 stop0 = 0xFFFFFFFFu) {
    // 8 * W bytes per side and step.  The loop is a chain of dependent memory round trips and a wave takes as many steps as its
    // slowest lane: with 1 % divergence a 32-byte step ends a comparison with probability 0.28 (ten steps until 32 of them are
    // through), a 64-byte step with 0.47 (five).
    // h0 / stop0: the first h0 symbols are known to be equal (same key) and their first stop, if any, is known from the key
    const uint8_t *pa = T + (int64_t)a, *pb = T + (int64_t)b;
    u32 stop_at = stop0;
    for (int off = h0; off < TEXT_LIM; off += 8 * W) {
        u64 wa[W], wb[W];
        __builtin_memcpy(wa, pa + off, 8 * W);
        __builtin_memcpy(wb, pb + off, 8 * W);
        if (stop_at == 0xFFFFFFFFu) {
            u64 any = 0;
#pragma unroll
            for (int k = 0; k < W; k++) any |= zero_bytes(wb[k] ^ 0x2424242424242424ull) | zero_bytes(wb[k] ^ 0x4E4E4E4E4E4E4E4Eull) | zero_bytes(wb[k]);
            if (any) {
#pragma unroll
                for (int k = W - 1; k >= 0; k--) {       // (downwards: the last assignment is the first word with a stop)
                    const u64 st = zero_bytes(wb[k] ^ 0x2424242424242424ull) | zero_bytes(wb[k] ^ 0x4E4E4E4E4E4E4E4Eull) | zero_bytes(wb[k]);
                    stop_at = st ? (u32)off + 8u * (u32)k + (u32)(__builtin_ctzll(st) >> 3) : stop_at;
                }
            }
        }
        u64 x = 0, y = 0; u32 at = 0; bool diff = false;
#pragma unroll
        for (int k = W - 1; k >= 0; k--) {               // (downwards: ends with the first differing word)
            const bool d = wa[k] != wb[k];
            x = d ? wa[k] : x; y = d ? wb[k] : y; at = d ? 8u * (u32)k : at; diff |= d;
        }
        if (diff) {
            const u32 dpos = (u32)off + at + (u32)(__builtin_ctzll(x ^ y) >> 3);
            *lcp = dpos < stop_at ? dpos : stop_at;
            // big-endian compare of the first differing word; the shorter suffix runs into the zero padding first and sorts first
            return __builtin_bswap64(x) < __builtin_bswap64(y) ? -1 : 1;
        }
    }
    *lcp = 0;
    return 0;
}

// the same comparison on the 2-bit text, 128 bases per step; leaves to cmp_text where a window touches an exception block
template <int W>
__device__ inline int cmp_suffix(const uint8_t *__restrict__ T, const Packed &pk, sav_t a, sav_t b, u32 *lcp, int h0, u32 stop0) {
#ifdef RV_RT_NOCMP      // (measurement only: the round without its text accesses)
    *lcp = 20; return a < b ? -1 : 1;
#endif
    if (pk.Tp == nullptr) return cmp_text<W>(T, a, b, lcp, h0, stop0);
    for (int off = h0; off < TEXT_LIM; off += PK_BLOCK) {
        const int64_t pa = (int64_t)a + off, pb = (int64_t)b + off;
        const int64_t ba = pa / PK_BLOCK, bb = pb / PK_BLOCK;
        if (pk.blk[ba] | pk.blk[ba + 1] | pk.blk[bb] | pk.blk[bb + 1]) return cmp_text<W>(T, a, b, lcp, off, stop0);
        u64 ra[5], rb[5];
        __builtin_memcpy(ra, pk.Tp + (pa >> 5), 40);
        __builtin_memcpy(rb, pk.Tp + (pb >> 5), 40);
        const int sa = (int)(pa & 31) * 2, sb = (int)(pb & 31) * 2;
        u64 x = 0, y = 0; u32 at = 0; bool diff = false;
#pragma unroll
        for (int k = 3; k >= 0; k--) {                       // (downwards: ends with the first differing word)
            const u64 xa = sa ? (ra[k] >> sa) | (ra[k + 1] << (64 - sa)) : ra[k];
            const u64 xb = sb ? (rb[k] >> sb) | (rb[k + 1] << (64 - sb)) : rb[k];
            const bool d = xa != xb;
            x = d ? xa : x; y = d ? xb : y; at = d ? 32u * (u32)k : at; diff |= d;
        }
        if (diff) {
            const int bit = __builtin_ctzll(x ^ y) & ~1;
            const u32 dpos = (u32)off + at + (u32)(bit >> 1);
            *lcp = dpos < stop0 ? dpos : stop0;
            return ((x >> bit) & 3u) < ((y >> bit) & 3u) ? -1 : 1;
        }
    }
    *lcp = 0;
    return 0;
}

// what the fused path writes besides SA: BWT byte of every member at its final rank, LCP of every member but the group's first
// (its LCP with the member in front of it = the largest common prefix it has with any smaller member), the running maximum
struct FusedOut { lcp_t *LCP; uint8_t *BWT; const u64 *keys; u32 *maxlcp; sa_t side_sep; KeyDigits kd; int h; Packed pk; int far_defer; };
// a (first sample) and b (second) are partners on the diagonal the hint follows -- the predicate of k_far_twins (far_linked)
__device__ inline bool twins_linked(const KeyDigits &kd, int64_t n, int64_t a, int64_t b) {
    if (a < 0 || a >= kd.D - 1 || b >= n || b < kd.D) return false;
    if (!kd.dtab) return b - a == kd.D;
    const int32_t dd = kd.dtab[a >> DT_SHIFT];
    return dd != DT_NONE && kd.D + (int64_t)dd == b - a && kd.dtab[b >> DT_SHIFT] == dd;
}
__device__ inline void fused_put(const FusedOut &f, size_t rank, sav_t suf, u32 pay, bool first, u32 lcp, u32 &lmax) {
    f.BWT[rank] = (uint8_t)(pay | ((sa_t)suf > f.side_sep ? RV_BWT_SIDE : 0u));
    if (!first) {
        f.LCP[rank] = (lcp_t)lcp;
        lmax = lcp > lmax ? lcp : lmax;
    }
}
// the running maximum of the index' LCP values: one look at the word per wave.  (Every put looked at it: 5 x 10^8 loads of one
// address per round at 2 x 250 Mbp -- one L2 channel per XCD serving them was what the round waited for, 24 of its 29 ms.)
__device__ inline void fused_max_flush(const FusedOut &f, u32 lmax) {
    const u32 wm = (u32)rv_wave_max_u64((u64)lmax);
    if ((threadIdx.x & 63) == 0 && f.maxlcp && wm > __atomic_load_n(f.maxlcp, __ATOMIC_RELAXED)) atomicMax(f.maxlcp, wm);
}

constexpr int MEDIUM_GROUP = 64;      // groups of up to 64 members: every member ranks itself by text comparison (into Sout; k_medium_back copies back)
__global__ __launch_bounds__(TB) void k_medium_back(const uint8_t *__restrict__ flag, const sav_t *__restrict__ Sout, sav_t *__restrict__ S, int64_t m) {
    // eight list entries per thread, their flags in one load (one byte load per thread made this 1.3 ms at m = 5e8: 7.8e6 waves for 0.5 GB)
    const int64_t q0 = ((int64_t)blockIdx.x * TB + threadIdx.x) * 8;
    if (q0 >= m) return;
    if (q0 + 8 <= m) {
        const u64 f = *reinterpret_cast<const u64 *>(flag + q0);
        if (((f ^ 0x0202020202020202ull) - 0x0101010101010101ull) & ~(f ^ 0x0202020202020202ull) & 0x8080808080808080ull) {      // some byte == 2
#pragma unroll
            for (int k = 0; k < 8; k++) if (((f >> (8 * k)) & 0xFFu) == 2u) S[q0 + k] = Sout[q0 + k];
        }
    } else {
        for (int64_t q = q0; q < m; q++) if (flag[q] == 2) S[q] = Sout[q];
    }
}
// MODE 2: every member of a group finds its own rank -- one text comparison with each other member, all lanes busy.
// MODE 1: the same for groups of three and more; a pair is ordered by its first thread alone.
// MODE 0: groups of up to SMALL_GROUP members are ordered by their first thread (all pairs in registers), larger ones rank themselves.
template <int W, int MODE>
__device__ __forceinline__ void round_text_body(const uint8_t *__restrict__ T, sav_t *__restrict__ S, const u32 *__restrict__ G, const u32 *__restrict__ P,
                                                int64_t m, uint8_t *__restrict__ headq, uint8_t *__restrict__ bigflag, sa_t *__restrict__ SA,
                                                sav_t *__restrict__ Sout, const FusedOut &fo, int64_t q, u32 &lmax) {
    const u32 g = G[q];
    const u32 off = P[q] - g;
    const int64_t look = q + (MEDIUM_GROUP - (int64_t)off);
    const bool big = off >= (u32)MEDIUM_GROUP || (look < m && G[look] == g);
    const int64_t qs = q - (int64_t)off;                       // the group's first list entry (a group is contiguous in the list)
    const bool fused = fo.LCP != nullptr;
    // the comparisons start behind the h symbols the group's key stands for; a stop among those symbols comes from the key's digits
    const int h0 = fo.h;
    const u32 stop0 = (fused && !big) ? key_first_stop(fo.keys[g], fo.kd) : 0xFFFFFFFFu;
    constexpr int DIRECT = MODE == 0 ? SMALL_GROUP : MODE == 1 ? 2 : 1;      // groups up to this size are ordered by their first thread
    const bool self = !big && qs + DIRECT < m && G[qs + DIRECT] == g;
    bigflag[q] = big ? 1 : self ? 2 : 0;
    if (big) return;
    if (self) {
        int size = (int)off + 1;
        while (qs + size < m && G[qs + size] == g) size++;
        const sav_t mine = S[q];
        int rank = 0; bool tie_before = false; u32 best = 0;
        for (int j = 0; j < size; j++) {
            if (j == (int)off) continue;
            u32 l;
            const int c = cmp_suffix<W>(T, fo.pk, S[qs + j], mine, &l, h0, stop0);
            rank += (c < 0) | ((c == 0) & (j < (int)off));
            tie_before |= (c == 0) & (j < (int)off);
            best = (c < 0 && l > best) ? l : best;               // LCP with the member in front = the longest common prefix with any smaller one
        }
        Sout[qs + rank] = mine;                                  // (S itself is still being read by the other members)
        SA[(size_t)g + rank] = (sa_t)mine;
        headq[qs + rank] = !tie_before;
        if (fused) fused_put(fo, (size_t)g + rank, mine, (u32)(fo.keys[(size_t)g + off] >> 56), rank == 0, best, lmax);
        return;
    }
    if (off != 0) return;
    if (MODE >= 1) {                                             // a pair
        const sav_t s0 = S[q], s1 = S[q + 1];
        u32 l;
        const int c = cmp_suffix<W>(T, fo.pk, s0, s1, &l, h0, stop0);
        const sav_t lo = c <= 0 ? s0 : s1, hi = c <= 0 ? s1 : s0;
        S[q] = lo; S[q + 1] = hi;
        SA[(size_t)g] = (sa_t)lo; SA[(size_t)g + 1] = (sa_t)hi;
        headq[q] = 1; headq[q + 1] = c != 0;
        if (fused) {
            const u32 p0 = (u32)(fo.keys[(size_t)g] >> 56), p1 = (u32)(fo.keys[(size_t)g + 1] >> 56);
            fused_put(fo, (size_t)g, lo, c <= 0 ? p0 : p1, true, 0, lmax);
            fused_put(fo, (size_t)g + 1, hi, c <= 0 ? p1 : p0, false, l, lmax);
        }
        return;
    }
    sav_t s[SMALL_GROUP];
    int size = 1;
    s[0] = S[q];
#pragma unroll
    for (int j = 1; j < SMALL_GROUP; j++) {
        if (size == j && q + j < m && G[q + j] == g) { s[j] = S[q + j]; size = j + 1; }
    }
    if (size == 2) {
        u32 l;
        const int c = cmp_suffix<W>(T, fo.pk, s[0], s[1], &l, h0, stop0);
        const sav_t lo = c <= 0 ? s[0] : s[1], hi = c <= 0 ? s[1] : s[0];
        S[q] = lo; S[q + 1] = hi;
        SA[(size_t)g] = (sa_t)lo; SA[(size_t)g + 1] = (sa_t)hi;
        headq[q] = 1; headq[q + 1] = c != 0;
        if (fused) {
            const u32 p0 = (u32)(fo.keys[(size_t)g] >> 56), p1 = (u32)(fo.keys[(size_t)g + 1] >> 56);
            fused_put(fo, (size_t)g, lo, c <= 0 ? p0 : p1, true, 0, lmax);
            fused_put(fo, (size_t)g + 1, hi, c <= 0 ? p1 : p0, false, l, lmax);
        }
        return;
    }
    // all pairs once: less[i] bit j = (s[j] < s[i]); eq likewise; rank = #smaller + #equal with smaller index.
    // best[i] = the largest common prefix of member i with a smaller member = its LCP with the one that ends up in front of it
    u32 less[SMALL_GROUP], eq[SMALL_GROUP], best[SMALL_GROUP];
#pragma unroll
    for (int i = 0; i < SMALL_GROUP; i++) { less[i] = 0; eq[i] = 0; best[i] = 0; }
#pragma unroll
    for (int i = 0; i < SMALL_GROUP; i++) {
#pragma unroll
        for (int j = i + 1; j < SMALL_GROUP; j++) {
            if (j < size) {
                u32 l;
                const int c = cmp_suffix<W>(T, fo.pk, s[i], s[j], &l, h0, stop0);
                if (c < 0) { less[j] |= 1u << i; best[j] = l > best[j] ? l : best[j]; }
                else if (c > 0) { less[i] |= 1u << j; best[i] = l > best[i] ? l : best[i]; }
                else { eq[j] |= 1u << i; eq[i] |= 1u << j; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < SMALL_GROUP; i++) {
        if (i < size) {
            const u32 before = eq[i] & ((1u << i) - 1u);
            const int r = __popc(less[i]) + __popc(before);
            S[q + r] = s[i];
            SA[(size_t)g + r] = (sa_t)s[i];
            headq[q + r] = before == 0;
            if (fused) fused_put(fo, (size_t)g + r, s[i], (u32)(fo.keys[(size_t)g + i] >> 56), r == 0, best[i], lmax);
        }
    }
}


template <int W, int MODE>
__global__ __launch_bounds__(TB) void k_round_text(const uint8_t *__restrict__ T, sav_t *__restrict__ S, const u32 *__restrict__ G, const u32 *__restrict__ P,
                                                   int64_t m, uint8_t *__restrict__ headq, uint8_t *__restrict__ bigflag, sa_t *__restrict__ SA,
                                                   sav_t *__restrict__ Sout, FusedOut fo) {
    const int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x;
    u32 lmax = 0;
    if (q < m) round_text_body<W, MODE>(T, S, G, P, m, headq, bigflag, SA, Sout, fo, q, lmax);
    if (fo.LCP) fused_max_flush(fo, lmax);
}

// The first round with the diagonal hint (k_init_keys): groups of two to four by one thread, larger ones rank themselves
// as in k_round_text.  Twins on the diagonal are ordered from their keys; an unrelated pair (two twin pairs that collide in
// their K symbols) is compared on the text once and the other members' order against each other follows -- x and its twin x'
// agree for nd symbols, so against a third suffix z they behave alike up to there: if z leaves x' before nd, it leaves x at the
// same place the same way; if later, x against z is x against x'.  Whatever that does not settle is compared on the text.
// Two passes: k_round_text3 streams the list and finishes every group its keys decide (four in five at 2 x 250 Mbp); a group
// that needs the text goes to a work list, and k_round_text3b takes one group per thread from it.  (In one pass the round was
// bound by the latency of the few lanes per wave that went to the text: 29 ms for 36 GB of traffic.)
constexpr int RT_REGIONS = 1024;
template <int W, bool ALLOW_TEXT>
__device__ __forceinline__ bool order_small(const uint8_t *__restrict__ T, const FusedOut &fo, int64_t q, u32 g, const sav_t *s, const u64 *kk, int size,
                                            int h0, u32 stop0, sav_t *__restrict__ S, sa_t *__restrict__ SA, uint8_t *__restrict__ headq, u32 &lmax) {
    // pair (i, j), i < j, lives at index i * 4 + j; c < 0: s[i] is the smaller suffix
    int c[16]; u32 l[16]; bool kn[16], tw[16];
#pragma unroll
    for (int x = 0; x < 16; x++) { c[x] = 0; l[x] = 0; kn[x] = false; tw[x] = false; }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i + 1; j < 4; j++) {
            if (j < size) {
                int cc; u32 nd;
                if (hint_cmp(fo.kd, (int64_t)s[i], kk[i], (int64_t)s[j], kk[j], &cc, &nd)) {
                    c[i * 4 + j] = cc; l[i * 4 + j] = nd < stop0 ? nd : stop0; kn[i * 4 + j] = true; tw[i * 4 + j] = true;
                }
            }
        }
    bool need_text = false;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i + 1; j < 4; j++) {
            if (j < size && !kn[i * 4 + j]) {
                bool done = false;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k != i && k != j && k < size && !done) {
                        const int ik = i < k ? i * 4 + k : k * 4 + i, kj = k < j ? k * 4 + j : j * 4 + k;
                        const int c_ik = i < k ? c[ik] : -c[ik], c_kj = k < j ? c[kj] : -c[kj];      // oriented: i against k, k against j
                        if (tw[ik] && kn[kj] && c_kj != 0) {              // i and k are twins
                            if (l[kj] < l[ik]) { c[i * 4 + j] = c_kj; l[i * 4 + j] = l[kj]; done = true; }
                            else if (l[kj] > l[ik]) { c[i * 4 + j] = c_ik; l[i * 4 + j] = l[ik]; done = true; }
                        } else if (tw[kj] && kn[ik] && c_ik != 0) {       // j and k are twins
                            if (l[ik] < l[kj]) { c[i * 4 + j] = c_ik; l[i * 4 + j] = l[ik]; done = true; }
                            else if (l[ik] > l[kj]) { c[i * 4 + j] = c_kj; l[i * 4 + j] = l[kj]; done = true; }
                        }
                    }
                }
                if (!done) {
                    if (ALLOW_TEXT) { u32 ll = 0; c[i * 4 + j] = cmp_suffix<W>(T, fo.pk, s[i], s[j], &ll, h0, stop0); l[i * 4 + j] = ll; }
                    else need_text = true;
                }
                kn[i * 4 + j] = true;
            }
        }
    if (!ALLOW_TEXT && need_text) return false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < size) {
            int r = 0; bool tie_before = false; u32 best = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j != i && j < size) {
                    const int ji = j < i ? j * 4 + i : i * 4 + j;
                    const int cji = j < i ? c[ji] : -c[ji];               // j against i
                    r += (cji < 0) | ((cji == 0) & (j < i));
                    tie_before |= (cji == 0) & (j < i);
                    best = (cji < 0 && l[ji] > best) ? l[ji] : best;
                }
            }
            S[q + r] = s[i];
            SA[(size_t)g + r] = (sa_t)s[i];
            headq[q + r] = !tie_before;
            fused_put(fo, (size_t)g + r, s[i], (u32)(kk[i] >> 56), r == 0, best, lmax);
        }
    }
    return true;
}
// the members of the group that starts at list entry q (at most four), everything loaded at once: clamped to the arrays instead of guarded
__device__ __forceinline__ int load_small(const sav_t *__restrict__ S, const u32 *__restrict__ G, const u64 *__restrict__ keys, int64_t q, u32 g, u64 key_g,
                                          int64_t m, int64_t n, sav_t *s, u64 *kk) {
    u32 gg[4];
    s[0] = S[q]; kk[0] = key_g; gg[0] = g;
#pragma unroll
    for (int j = 1; j < 4; j++) {
        const int64_t qj = q + j < m ? q + j : m - 1;
        const int64_t rj = (int64_t)g + j < n ? (int64_t)g + j : n - 1;
        gg[j] = G[qj]; s[j] = S[qj]; kk[j] = keys[rj];
    }
    int size = 1;
#pragma unroll
    for (int j = 1; j < 4; j++) size += (size == j && q + j < m && gg[j] == g) ? 1 : 0;
    return size;
}
// (eight waves per SIMD: 64 registers, three spilled -- the kernel waits for memory, 1.49 -> 1.30 ms at 10 x 5 Mbp; k_round_text3b at six: 0.45 -> 0.70 ms at 2 x 250 Mbp, not taken)
template <int W>
__global__ __launch_bounds__(TB, 8) void k_round_text3(const uint8_t *__restrict__ T, sav_t *__restrict__ S, const u32 *__restrict__ G, const u32 *__restrict__ P,
                                                    int64_t m, int64_t n, uint8_t *__restrict__ headq, uint8_t *__restrict__ bigflag, sa_t *__restrict__ SA,
                                                    sav_t *__restrict__ Sout, FusedOut fo, u32 *__restrict__ work, u32 *__restrict__ work_count, u32 reg_cap) {
    const int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool defer = false;
    u32 lmax = 0;
    // Every entry's suffix, key, sample, position of its homologue in the first sample and hint, once, in LDS: the members of a self-ranking
    // group (ten homologues per group with ten samples) compare from there.  Taken apart per comparison -- two loads and two walks over the
    // separators each -- the kernel was 3.3 ms of a 14.8 ms step at 10 x 5 Mbp.  (Fixed diagonals only: fo.kd.dtab is for two samples.)
    // p_w, an entry's place among the homologues of its base as ONE word: bit 31 the hint says something (the base itself, or a variant whose
    // agreement with the base is known), bits 12..24 the order -- a variant below the base 2 nd (it leaves the base after nd symbols: the earlier,
    // the smaller), the base 4095, a variant above it 2 (4095 - nd) (the later it leaves, the nearer to the base) --, bits 0..11 the agreement
    // (the base: all ones).  Two entries with the same base and different orders are ordered by them, their common prefix is the smaller
    // agreement: hint_cmp case by case.  (Decoded per pair from sample / known / side / agreement the loop below was ~130 vector instructions
    // per pair, 1 320 per entry with ten samples.)
    // RT_HALO entries on either side of the workgroup's are staged as well: a group that crosses the workgroup's border -- one in 25 with ten
    // samples -- sent its members to the path that reads global memory and walks the separators per comparison, and took the whole wave along
    // for ten rounds: half the waves of the kernel, 0.85 of its 2.35 ms at 10 x 5 Mbp
    constexpr int RT_HALO = 16;
    __shared__ int64_t p_base[TB + 2 * RT_HALO];
    __shared__ sav_t p_suf[TB + 2 * RT_HALO];
    __shared__ u32 p_w[TB + 2 * RT_HALO];
    __shared__ u64 p_key[TB + 2 * RT_HALO];      // (the key itself: the second member of a pair of twins)
    const bool pre = fo.kd.ly.nd_bits > 0 && fo.kd.dtab == nullptr;
    const int64_t q0 = (int64_t)blockIdx.x * TB;
    __shared__ u32 s_G[TB + MEDIUM_GROUP + 1];      // the group ranks of the workgroup's entries and of the 64 behind them (where a group ends is asked entry by entry)
    // Two trips to memory, each with everything it can bring: suffix, rank and group of the thread's own entry, of a halo entry, of the entries behind
    // the workgroup -- then the keys at those ranks.  (Staging, group ranks, the entry's own rank / suffix / key and its group's key one after the
    // other were five dependent trips: 61 % of the waves' cycles waiting for memory, 11 us per wave for 836 vector instructions.)
    const bool have = q < m;
    sav_t sf_own = 0; u32 p_own = 0;
    if (have) { sf_own = S[q]; p_own = P[q]; }
    const int64_t gi0 = q0 + threadIdx.x, gi1 = q0 + TB + threadIdx.x;
    const u32 gv0 = G[gi0 < m ? gi0 : m - 1];
    const u32 gv1 = (int)threadIdx.x <= MEDIUM_GROUP ? G[gi1 < m ? gi1 : m - 1] : 0u;
    int64_t e2 = -1; sav_t sf2 = 0; u32 p2 = 0;      // the thread's halo entry (the first 2 RT_HALO threads)
    const int idx2 = (int)threadIdx.x < RT_HALO ? (int)threadIdx.x : TB + (int)threadIdx.x;
    if (pre && (int)threadIdx.x < 2 * RT_HALO) {
        e2 = (int)threadIdx.x < RT_HALO ? q0 - RT_HALO + threadIdx.x : q0 + TB + threadIdx.x - RT_HALO;
        if (e2 >= 0 && e2 < m) { sf2 = S[e2]; p2 = P[e2]; } else e2 = -1;
    }
    const u64 key_own = have ? fo.keys[p_own] : 0ull;
    const u64 key2 = e2 >= 0 ? fo.keys[p2] : 0ull;
    s_G[threadIdx.x] = gv0;
    if ((int)threadIdx.x <= MEDIUM_GROUP) s_G[TB + threadIdx.x] = gv1;
    if (pre) {
        auto stage = [&](int idx, bool there, sav_t sf, u64 key) {
            u32 w = 0; int64_t base = -1;
            if (there) {
                const int sm = hint_sample(fo.kd, (int64_t)sf);
                if (sm < HINT_K && sm < 15) {
                    base = (int64_t)sf - (sm ? fo.kd.Ds[sm] : 0);
                    u32 nd = 0; bool ltb = false;
                    const bool known = key_hint(key, fo.kd, &nd, &ltb);      // (nd_bits <= 11: nd <= 2046)
                    if (sm == 0) w = 0x80000000u | (4095u << 12) | 0xFFFu;
                    else if (known) w = 0x80000000u | ((ltb ? 2u * nd : 2u * (4095u - nd)) << 12) | nd;
                }
            }
            p_base[idx] = base; p_suf[idx] = sf; p_w[idx] = w; p_key[idx] = key;
        };
        stage((int)threadIdx.x + RT_HALO, have, sf_own, key_own);
        if ((int)threadIdx.x < 2 * RT_HALO) stage(idx2, e2 >= 0, sf2, key2);
    }
    __syncthreads();
    auto Gat = [&](int64_t i) -> u32 { const int64_t x = i - q0; return (x >= 0 && x <= TB + MEDIUM_GROUP) ? s_G[x] : G[i]; };
    if (q < m) {
        const u32 g = s_G[threadIdx.x];
        const u32 off = p_own - g;
        const int64_t qs = q - (int64_t)off;                       // the group's first list entry (a group is contiguous in the list)
        const int64_t look = qs + MEDIUM_GROUP, i4 = qs + 4 > q ? qs + 4 : q;      // (the fifth entry of the group, or the entry itself when it is a later one: staged)
        const u32 g_look = Gat(look < m ? look : m - 1), g_4 = Gat(i4 < m ? i4 : m - 1);
        // (the group's first key is only asked for its first stop among the K symbols -- every member's key holds the same -- and as the first member's own)
        const u64 key_g = key_own;
        const bool big = off >= (u32)MEDIUM_GROUP || (look < m && g_look == g);
        const bool self = !big && i4 < m && g_4 == g;
        if (!self) bigflag[q] = big ? 1 : 0;
        if (big) fo.BWT[p_own] = (uint8_t)((u32)(key_own >> 56) | ((sa_t)sf_own > fo.side_sep ? RV_BWT_SIDE : 0u));      // (rank p_own holds this suffix until the radix path moves it; k_lcp_list has the last word)
        const int h0 = fo.h;
        const u32 stop0 = key_first_stop(key_g, fo.kd);
        if (self) {
            // the members behind this one, four staged group ranks at a time (one at a time the walk was ten LDS round trips in a row)
            int size;
            {
                const int t = (int)threadIdx.x;
                const int64_t left = m - q;                              // entries from this one to the end of the list
                int d = 1;
                for (;;) {
                    const int i0 = t + d;                                // (a group has fewer than MEDIUM_GROUP members: the indices stay inside s_G but for the last step's)
                    const u32 a = s_G[i0 < TB + MEDIUM_GROUP ? i0 : TB + MEDIUM_GROUP], b = s_G[i0 + 1 < TB + MEDIUM_GROUP ? i0 + 1 : TB + MEDIUM_GROUP];
                    const u32 c = s_G[i0 + 2 < TB + MEDIUM_GROUP ? i0 + 2 : TB + MEDIUM_GROUP], e = s_G[i0 + 3 < TB + MEDIUM_GROUP ? i0 + 3 : TB + MEDIUM_GROUP];
                    int k = a != g ? 0 : b != g ? 1 : c != g ? 2 : e != g ? 3 : 4;
                    if ((int64_t)(d + k) > left) k = (int)(left - d);      // (behind the list's end the staged ranks repeat the last entry's)
                    d += k;
                    if (k < 4 || d >= MEDIUM_GROUP) break;
                }
                size = (int)off + d;
            }
            const sav_t mine = sf_own;
            int rank = 0; bool tie_before = false; u32 best = 0;
            const u64 key_mine = key_own;
            const bool in_lds = pre && qs >= q0 - RT_HALO && qs + size <= q0 + TB + RT_HALO;
            const u32 my_w = in_lds ? p_w[threadIdx.x + RT_HALO] : 0u;
            const int64_t my_base = in_lds ? p_base[threadIdx.x + RT_HALO] : -1;
            const u32 km = (my_w >> 12) & 0x1FFFu, am = my_w & 0xFFFu;
            if (in_lds) {
                const int xs = (int)(qs - q0) + RT_HALO;
                // homologues of one position of the first sample are ordered from their keys (hint_cmp on the words the entries left in LDS): ten
                // samples put ten of them into every group.  No branch inside: the rounds' LDS reads overlap; a pair the words do not order
                // (two variants that leave the base at the same place, a chance member of the group: rare) is counted and compared on the text
                // in a second loop that most waves never enter
                int unordered = 0; u32 best_h = 0;
                for (int j0 = 0; j0 < size; j0 += 4) {                   // (four members a round: eight LDS reads in flight)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const bool in = j0 + u < size;
                        const int x = xs + (in ? j0 + u : (int)off);     // (past the group's end: the entry itself, which nothing orders)
                        const u32 wo = p_w[x];
                        const u32 ko = (wo >> 12) & 0x1FFFu, ao = wo & 0xFFFu;
                        const bool hinted = (((wo & my_w) >> 31) != 0u) & (p_base[x] == my_base) & (ko != km);
                        const bool lt = hinted & (ko < km);              // the other one is the smaller suffix
                        const u32 l = ao < am ? ao : am;
                        unordered += (in & !hinted) ? 1 : 0;
                        rank += lt ? 1 : 0;
                        best_h = (lt && l > best_h) ? l : best_h;
                    }
                }
                best = best_h < stop0 ? best_h : stop0;
                if (unordered > 1) {                                     // (one: the entry itself)
                    for (int j = 0; j < size; j++) {
                        const u32 wo = p_w[xs + j];
                        const bool hinted = (((wo & my_w) >> 31) != 0u) & (p_base[xs + j] == my_base) & (((wo >> 12) & 0x1FFFu) != km);
                        if (!hinted && j != (int)off) {
                            u32 l2;
                            const int c = cmp_suffix<W>(T, fo.pk, p_suf[xs + j], mine, &l2, h0, stop0);
                            rank += (c < 0) | ((c == 0) & (j < (int)off));
                            tie_before |= (c == 0) & (j < (int)off);
                            best = (c < 0 && l2 > best) ? l2 : best;
                        }
                    }
                }
            } else {
                for (int j = 0; j < size; j++) {
                    if (j == (int)off) continue;
                    u32 l; int c;
                    const sav_t other = S[qs + j];
                    const bool hinted = hint_cmp(fo.kd, (int64_t)other, fo.keys[(size_t)g + j], (int64_t)mine, key_mine, &c, &l);
                    if (hinted) l = l < stop0 ? l : stop0;
                    else c = cmp_suffix<W>(T, fo.pk, other, mine, &l, h0, stop0);
                    rank += (c < 0) | ((c == 0) & (j < (int)off));
                    tie_before |= (c == 0) & (j < (int)off);
                    best = (c < 0 && l > best) ? l : best;
                }
            }
            // a group inside the workgroup's own entries has been read (into LDS) before the barrier, by this workgroup and as halo of its neighbours,
            // which rank none of its members: its list entries are rewritten in place.  A group across the border is ranked by two workgroups
            // that both read its entries: into the spare list, k_medium_back copies those back (flag 2)
            const bool own = in_lds && qs >= q0 && qs + size <= q0 + TB;
            bigflag[q] = own ? 0 : 2;
            (own ? S : Sout)[qs + rank] = mine;
            SA[(size_t)g + rank] = (sa_t)mine;
            headq[qs + rank] = !tie_before;
            fused_put(fo, (size_t)g + rank, mine, (u32)(key_mine >> 56), rank == 0, best, lmax);
        } else if (!big && off == 0) {
            // a pair of twins is finished here, from its keys; everything else needs the text at least once: the work list
            const int64_t q2 = q + 2 < m ? q + 2 : m - 1, r1 = (int64_t)g + 1 < n ? (int64_t)g + 1 : n - 1;
            const u32 g_2 = Gat(q2);
            const bool staged = pre && q + 1 < m;                // (the next entry holds rank g + 1 when it belongs to the group -- and nothing is a pair otherwise)
            const sav_t s0 = sf_own, s1 = staged ? p_suf[threadIdx.x + RT_HALO + 1] : S[q + 1 < m ? q + 1 : m - 1];
            const u64 key_1 = staged ? p_key[threadIdx.x + RT_HALO + 1] : fo.keys[r1];
            const bool pair = !(q + 2 < m && g_2 == g);
            int cc; u32 nd;
            const bool hinted = pair && hint_cmp(fo.kd, (int64_t)s0, key_g, (int64_t)s1, key_1, &cc, &nd);
            if (hinted) {
                const bool first0 = cc < 0;                          // s0 is the smaller suffix
                const sav_t lo = first0 ? s0 : s1, hi = first0 ? s1 : s0;
                S[q] = lo; S[q + 1] = hi;
                SA[(size_t)g] = (sa_t)lo; SA[(size_t)g + 1] = (sa_t)hi;
                headq[q] = 1; headq[q + 1] = 1;
                const u32 p0 = (u32)(key_g >> 56), p1 = (u32)(key_1 >> 56);
                fused_put(fo, (size_t)g, lo, first0 ? p0 : p1, true, 0, lmax);
                fused_put(fo, (size_t)g + 1, hi, first0 ? p1 : p0, false, nd < stop0 ? nd : stop0, lmax);
            } else if (pair && fo.far_defer && twins_linked(fo.kd, n, (int64_t)(s0 < s1 ? s0 : s1), (int64_t)(s0 < s1 ? s1 : s0))) {
                // partners whose agreement does not fit the key's field (near-identical inputs): left tied without a look at the text --
                // k_far_twins reads their order and LCP off the diagonal's marks (2 x 50 Mbp of identical text: 5 x 10^7 comparisons of 4 KB)
                SA[(size_t)g] = (sa_t)s0; SA[(size_t)g + 1] = (sa_t)s1;
                headq[q] = 1; headq[q + 1] = 0;
                // (the bytes in front of the two, in this order: k_far_twins swaps them with the suffixes instead of reading the text)
                fused_put(fo, (size_t)g, s0, (u32)(key_g >> 56), true, 0, lmax);
                fused_put(fo, (size_t)g + 1, s1, (u32)(key_1 >> 56), false, 0, lmax);
            } else defer = true;
        }
    }
    fused_max_flush(fo, lmax);
    // the work list in RT_REGIONS regions with a counter each (one list, one counter: 7.8 x 10^6 returning atomics on one address,
    // 79 ms at 2 x 250 Mbp); a region holds the groups of every RT_REGIONS-th workgroup
    const u64 bal = __ballot(defer);
    if (bal) {
        const u32 reg = blockIdx.x & (RT_REGIONS - 1);
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&work_count[reg], (u32)__popcll(bal));
        base = (u32)__shfl((int)base, 0, 64);
        if (defer) work[(size_t)reg * reg_cap + base + (u32)__popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = (u32)q;
    }
}
template <int W>
__global__ __launch_bounds__(TB) void k_round_text3b(const uint8_t *__restrict__ T, sav_t *__restrict__ S, const u32 *__restrict__ G, int64_t m, int64_t n,
                                                     uint8_t *__restrict__ headq, sa_t *__restrict__ SA, FusedOut fo, const u32 *__restrict__ work,
                                                     const u32 *__restrict__ work_count, u32 reg_cap, u32 blocks_per_region) {
    const u32 reg = blockIdx.x / blocks_per_region;
    const u32 w = (blockIdx.x % blocks_per_region) * TB + threadIdx.x;
    u32 lmax = 0;
    if (w < work_count[reg]) {
        const int64_t q = (int64_t)work[(size_t)reg * reg_cap + w];
        const u32 g = G[q];
        const u64 key_g = fo.keys[g];
        sav_t s[4]; u64 kk[4];
        const int size = load_small(S, G, fo.keys, q, g, key_g, m, n, s, kk);
        (void)order_small<W, true>(T, fo, q, g, s, kk, size, fo.h, key_first_stop(key_g, fo.kd), S, SA, headq, lmax);
    }
    fused_max_flush(fo, lmax);
}

// ---- far twins: a pair on one diagonal that agrees beyond the text round's window --------------------------------------
// Two samples with the diagonal hint.  A suffix p of the first sample and its twin p + delta (delta = D, or D + dtab[..] on piecewise
// diagonals) that the text round left tied agree for TEXT_LIM characters and more: nearly identical genomes (0.1 % divergence: one pair
// in fifty; identical inputs: every pair).  Prefix doubling orders them in log2(agreement / K) rounds over everything that is left --
// 9 rounds at 0.1 %, 22 at 2 x 50 Mbp of identical text (9.8 s) -- and loses the fused LCP (another pass over the whole index).  Along
// the diagonal their order is a look-up instead: the next marked position y of k_diag_bits (the reversed max-scan M over the words of the
// stop bits gives the next word that holds a mark) is where the two texts differ next -- LCP = y - p, the `lt` bit says which one is
// smaller.  A mark that is an exception ('$', 'N', IUPAC, lower case, a change of diagonal) is looked at on the text, TEXT_LIM bytes at
// a time, then the diagonal is taken up again; FAR_MAXIT such windows and the pair is left to the doubling rounds.
struct FarTwins { const u64 *stop, *exc, *lt; const u32 *M; int64_t nw, n, S2, D; const int32_t *dtab; };
constexpr int FAR_MAXIT = 64;
__global__ __launch_bounds__(TB) void k_far_rev(const u64 *__restrict__ stop, int64_t nw, u32 *__restrict__ R) {
    const int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (j < nw) R[j] = stop[nw - 1 - j] ? (u32)(j + 1) : 0u;
}
__device__ inline int64_t far_next_mark(const FarTwins &ft, int64_t pos) {      // first marked position >= pos (64 nw: none)
    int64_t w = pos >> 6;
    if (w >= ft.nw) return ft.nw * 64;
    u64 bits = ft.stop[w] & (~0ull << (pos & 63));
    if (!bits) {
        if (w + 1 >= ft.nw) return ft.nw * 64;
        const u32 mm = ft.M[ft.nw - 2 - w];
        if (!mm) return ft.nw * 64;
        w = ft.nw - (int64_t)mm;
        bits = ft.stop[w];
    }
    return w * 64 + __builtin_ctzll(bits);
}
__device__ inline bool far_linked(const FarTwins &ft, int64_t y, int64_t delta) {      // y (first sample) and y + delta are partners
    if (y < 0 || y >= ft.S2 - 1 || y + delta >= ft.n || y + delta < ft.S2) return false;
    if (!ft.dtab) return delta == ft.D;
    const int32_t dd = ft.dtab[y >> DT_SHIFT];
    return dd != DT_NONE && ft.D + (int64_t)dd == delta && ft.dtab[(y + delta) >> DT_SHIFT] == dd;
}
// -> -1 / +1: a / b is the smaller suffix (0: not decided); *lcp as compute_lcp counts it (interface.c:97-114)
// (pos0: nothing is marked in [a, pos0) -- the two texts agree that far, base for base)
__device__ inline int far_cmp(const uint8_t *__restrict__ T, const FarTwins &ft, int64_t a, int64_t b, u32 *lcp, int64_t pos0) {
    const int64_t delta = b - a;
    int64_t pos = pos0, stop_at = INT64_MAX;
    bool on_diag = true;
    for (int it = 0; it < FAR_MAXIT; it++) {
        int64_t y = pos;
        if (on_diag) {
            y = far_next_mark(ft, pos);
            if (y >= ft.S2 - 1) y = ft.S2 - 1;                   // (the separator between the samples is an exception at the latest)
            const bool is_exc = y >= ft.nw * 64 || ((ft.exc[y >> 6] >> (y & 63)) & 1ull) != 0ull;
            if (!is_exc) {      // both A / C / G / T, linked, different: the texts part here
                const int64_t l = y - a < stop_at ? y - a : stop_at;
                *lcp = (u32)l;
                return ((ft.lt[y >> 6] >> (y & 63)) & 1ull) ? -1 : 1;
            }
        }
        // a window of the text from y on: 32 bytes a step (cmp_text's loop, offsets beyond its int range)
        const uint8_t *pa = T + y, *pb = T + y + delta;
        for (int off = 0; off < TEXT_LIM; off += 32) {
            u64 wa[4], wb[4];
            __builtin_memcpy(wa, pa + off, 32);
            __builtin_memcpy(wb, pb + off, 32);
            if (stop_at == INT64_MAX) {
#pragma unroll
                for (int k = 3; k >= 0; k--) {
                    const u64 st = zero_bytes(wb[k] ^ 0x2424242424242424ull) | zero_bytes(wb[k] ^ 0x4E4E4E4E4E4E4E4Eull) | zero_bytes(wb[k]);
                    if (st) stop_at = (y - a) + off + 8 * k + (__builtin_ctzll(st) >> 3);
                }
            }
            u64 x = 0, z = 0; int at = 0; bool diff = false;
#pragma unroll
            for (int k = 3; k >= 0; k--) { const bool d = wa[k] != wb[k]; x = d ? wa[k] : x; z = d ? wb[k] : z; at = d ? 8 * k : at; diff |= d; }
            if (diff) {
                const int64_t dpos = (y - a) + off + at + (__builtin_ctzll(x ^ z) >> 3);
                *lcp = (u32)(dpos < stop_at ? dpos : stop_at);
                return __builtin_bswap64(x) < __builtin_bswap64(z) ? -1 : 1;
            }
        }
        pos = y + TEXT_LIM;
        on_diag = far_linked(ft, pos, delta);
    }
    return 0;
}
// The next mark of every position of the first sample, in text order: nd[p] = distance to the first marked position at or behind p (bits 0..29),
// bit 30 that mark is an exception, bit 31 the first sample's base is the smaller one there; FAR_NONE: no mark, or too far for the field.  Made once
// (a stream over the words of the bit arrays, 4 bytes written per position) so that a pair of k_far_twins -- the pairs come in rank order, their
// positions are anywhere in the text -- reads ONE word instead of walking stop word, reversed scan, stop word, exception word, order word one
// after the other (33 ms for 8.9 x 10^7 pairs at 2 x 250 Mbp with 0.1 % divergence).
constexpr u32 FAR_NONE = 0xFFFFFFFFu, FAR_EXC = 1u << 30, FAR_LT = 1u << 31, FAR_DIST = (1u << 30) - 1u;
__global__ __launch_bounds__(TB) void k_far_nd(FarTwins ft, u32 *__restrict__ nd, int64_t nwords1) {
    const int64_t w = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (w >= nwords1) return;
    const u64 sw = ft.stop[w], ew = ft.exc[w], lw = ft.lt[w];
    int64_t cy = -1; u32 ce = 0, cl = 0;
    if (w + 1 < ft.nw) {
        const u32 mm = ft.M[ft.nw - 2 - w];
        if (mm) {
            const int64_t w2 = ft.nw - (int64_t)mm;
            const int b = __builtin_ctzll(ft.stop[w2]);
            cy = w2 * 64 + b; ce = (u32)(ft.exc[w2] >> b) & 1u; cl = (u32)(ft.lt[w2] >> b) & 1u;
        }
    }
    const int64_t last = ft.S2 - 1;      // (positions of the first sample in front of its separator)
    for (int c4 = 15; c4 >= 0; c4--) {
        u32 v[4];
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const int b = 4 * c4 + k;
            const int64_t p = w * 64 + b;
            if ((sw >> b) & 1ull) { cy = p; ce = (u32)(ew >> b) & 1u; cl = (u32)(lw >> b) & 1u; }
            const int64_t d = cy - p;
            v[k] = (cy < 0 || d > (int64_t)FAR_DIST) ? FAR_NONE : ((u32)d | (ce ? FAR_EXC : 0u) | (cl ? FAR_LT : 0u));
        }
        const int64_t p0 = w * 64 + 4 * c4;
        if (p0 + 4 <= last) *reinterpret_cast<uint4 *>(nd + p0) = make_uint4(v[0], v[1], v[2], v[3]);
        else for (int k = 0; k < 4; k++) if (p0 + k < last) nd[p0 + k] = v[k];
    }
}
// One "round" over the list the text round left: a group of exactly two partners is finished (SA, LCP, BWT, both heads); every other
// entry keeps its group.  Every entry's head flag and big-group flag are written: the round's bookkeeping reads them for the whole list.
__global__ __launch_bounds__(TB) void k_far_twins(const uint8_t *__restrict__ T, sav_t *__restrict__ S, const u32 *__restrict__ G, const u32 *__restrict__ P, int64_t m,
                                                  FarTwins ft, const u32 *__restrict__ nd, uint8_t *__restrict__ headq, uint8_t *__restrict__ bigflag, sa_t *__restrict__ SA,
                                                  lcp_t *__restrict__ LCP, uint8_t *__restrict__ BWT, sa_t side_sep, u32 *__restrict__ maxlcp) {
    const int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x;
    u32 lmax = 0;
    if (q < m) {
        const u32 g = G[q];
        const u32 off = P[q] - g;
        bigflag[q] = 0;
        const bool first_of_pair = off == 0 && q + 1 < m && G[q + 1] == g && !(q + 2 < m && G[q + 2] == g);
        const bool second_of_pair = off == 1 && !(q + 1 < m && G[q + 1] == g);
        if (first_of_pair) {
            const int64_t s0 = (int64_t)S[q], s1 = (int64_t)S[q + 1];
            const int64_t a = s0 < s1 ? s0 : s1, b = s0 < s1 ? s1 : s0;
            int c = 0; u32 l = 0;
            if (far_linked(ft, a, b - a)) {
                const u32 v = nd ? nd[a] : FAR_NONE;
                if (v != FAR_NONE && !(v & FAR_EXC)) { l = v & FAR_DIST; c = (v & FAR_LT) ? -1 : 1; }      // both A / C / G / T, linked, different: the texts part there
                else c = far_cmp(T, ft, a, b, &l, v != FAR_NONE ? a + (int64_t)(v & FAR_DIST) : a);
            }
            if (c != 0) {
                const int64_t lo = c < 0 ? a : b, hi = c < 0 ? b : a;
                S[q] = (sav_t)lo; S[q + 1] = (sav_t)hi;
                SA[(size_t)g] = (sa_t)lo; SA[(size_t)g + 1] = (sa_t)hi;
                headq[q] = 1; headq[q + 1] = 1;
                LCP[(size_t)g + 1] = (lcp_t)l;
                // the bytes in front of the two suffixes stand at the two ranks, in the list's order (the text round wrote them with the
                // suffixes): they change places with them.  (Read from the text they were two more random sectors per pair.)
                if (lo != s0) { const uint8_t b0 = BWT[(size_t)g], b1 = BWT[(size_t)g + 1]; BWT[(size_t)g] = b1; BWT[(size_t)g + 1] = b0; }
                lmax = l;
            } else { headq[q] = 1; headq[q + 1] = 0; }
        } else if (!second_of_pair) headq[q] = off == 0;
    }
    const u32 wm = (u32)rv_wave_max_u64((u64)lmax);
    // (no counter of the finished pairs: a wave's atomic on one address was 2.8 x 10^6 of them at 2 x 250 Mbp with 0.1 % divergence, ~11 ns each:
    //  the whole kernel's 31.6 ms; the host reads the number off the list's length before and after)
    if ((threadIdx.x & 63) == 0 && wm > __atomic_load_n(maxlcp, __ATOMIC_RELAXED)) atomicMax(maxlcp, wm);
}

// ---- LCP / BWT of a list of ranks (what the doubling rounds ordered), straight from the text --------------------------
// The fused path leaves LCP and BWT at every rank the first key and the text round finish.  The ranks they do not finish -- ties beyond
// TEXT_LIM, groups above MEDIUM_GROUP: repeats -- used to send the WHOLE index through rv_build_lcp (three random passes over n: 19 of
// the 29 ms a 2 x 50 Mbp pair with 2 % repeats cost more than one without).  They are a short list: one wave per rank compares the two
// suffixes 512 bytes a step.  A rank whose suffixes agree for more than LL_MAX_STEPS steps raises the flag, and the caller falls back.
constexpr int LL_MAX_STEPS = 4096;      // 2 MB of common prefix
__device__ inline u64 stop_mask(u64 wa, u64 wb);
__global__ __launch_bounds__(TB) void k_lcp_list(const uint8_t *__restrict__ T, int64_t n, const sa_t *__restrict__ SA, const u32 *__restrict__ ranks, int64_t cnt,
                                                 lcp_t *__restrict__ LCP, uint8_t *__restrict__ BWT, sa_t side_sep, u32 *__restrict__ maxlcp, u32 *__restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int64_t i = ((int64_t)blockIdx.x * TB + threadIdx.x) >> 6;
    if (i >= cnt) return;
    const int64_t k = (int64_t)ranks[i];
    const int64_t pb = (int64_t)SA[k];
    if (lane == 0) BWT[k] = (uint8_t)((pb > 0 ? T[pb - 1] : (uint8_t)'$') | ((sa_t)pb > side_sep ? RV_BWT_SIDE : 0));
    if (k == 0) return;
    const int64_t pa = (int64_t)SA[k - 1];
    u32 h = 0; bool done = false;
    for (int step = 0; step < LL_MAX_STEPS; step++) {
        const int64_t off = (int64_t)h + 8 * lane;
        // (the text is zero padded for 64 bytes only: words that would start beyond count as the end of the text)
        const u64 wa = (pa + off < n + 48) ? load8(T + pa + off) : 0ull;
        const u64 wb = (pb + off < n + 48) ? load8(T + pb + off) : 0ull;
        const u64 stop = stop_mask(wa, wb);
        const u64 bal = __ballot(stop != 0);
        if (bal) {
            const int f = (int)__builtin_ctzll(bal);
            const u32 lo = (u32)__shfl((int)(u32)stop, f, 64), hi = (u32)__shfl((int)(u32)(stop >> 32), f, 64);
            const u64 st = ((u64)hi << 32) | lo;
            h += 8u * (u32)f + (u32)(__builtin_ctzll(st) >> 3);
            done = true;
            break;
        }
        h += 512;
    }
    if (lane == 0) {
        if (!done) { atomicOr(overflow, 1u); return; }
        LCP[k] = (lcp_t)h;
        if (h > __atomic_load_n(maxlcp, __ATOMIC_RELAXED)) atomicMax(maxlcp, h);
    }
}

// members of big groups -> sublist (ordered)
__global__ __launch_bounds__(TB) void k_flag_count(const uint8_t *__restrict__ flag, int64_t n, u32 *__restrict__ tilecnt) {
    __shared__ u32 wsum[TB / 64];
    const int64_t j0 = (int64_t)blockIdx.x * CP_TILE + (int64_t)threadIdx.x * CP_ITEMS;      // (consecutive entries per thread: counts only)
    u32 c = 0;
    if (j0 + CP_ITEMS <= n) {
        const u64 f = *reinterpret_cast<const u64 *>(flag + j0);
#pragma unroll
        for (int k = 0; k < CP_ITEMS; k++) c += ((f >> (8 * k)) & 0xFFu) == 1u;
    } else {
        for (int64_t j = j0; j < n && j < j0 + CP_ITEMS; j++) c += flag[j] == 1;
    }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(TB) void k_flag_emit(const uint8_t *__restrict__ flag, int64_t n, const u32 *__restrict__ tileoff,
                                                  const u32 *__restrict__ P, const sav_t *__restrict__ S, const u32 *__restrict__ G, int64_t nn, int64_t h,
                                                  const u32 *__restrict__ ISA, u32 *__restrict__ Pb, sav_t *__restrict__ Sb, u32 *__restrict__ Qb, u64 *__restrict__ kk) {
    __shared__ u32 wbase[TB / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * CP_TILE;
    u32 run = tileoff[blockIdx.x];
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll 1
    for (int r = 0; r < CP_ITEMS; r++) {
        const int64_t j = base + (int64_t)r * TB + threadIdx.x;
        const bool f = (j < n) && flag[j] == 1;
        const u64 bal = __ballot(f);
        if (lane == 0) wbase[w] = (u32)__popcll(bal);
        __syncthreads();
        u32 before = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < TB / 64; k++) { const u32 c = wbase[k]; if (k < w) before += c; tot += c; }
        if (f) {
            const u32 qb = run + before + (u32)__popcll(bal & lt);
            const sav_t s = S[j];
            Pb[qb] = P[j]; Sb[qb] = s; Qb[qb] = (u32)j;
            kk[qb] = ((u64)G[j] << 32) | key2_of(s, h, nn, ISA);
        }
        run += tot;
        __syncthreads();
    }
}
// sorted big sublist back into the list
__global__ __launch_bounds__(TB) void k_big_writeback(const u64 *__restrict__ kk, const u32 *__restrict__ Pb, const u32 *__restrict__ Qb, const sav_t *__restrict__ Sb,
                                                      int64_t mb, sav_t *__restrict__ S, uint8_t *__restrict__ headq, sa_t *__restrict__ SA) {
    const int64_t qb = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (qb >= mb) return;
    const sav_t s = Sb[qb];
    const u32 q = Qb[qb];
    S[q] = s;
    SA[Pb[qb]] = (sa_t)s;
    headq[q] = (qb == 0) || kk[qb] != kk[qb - 1];
}
__global__ __launch_bounds__(TB) void k_seed(const uint8_t *__restrict__ headq, const u32 *__restrict__ P, int64_t m, u32 *__restrict__ seed) {
    const int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (q < m) seed[q] = headq[q] ? P[q] : 0u;
}

__global__ __launch_bounds__(TB) void k_inverse(const sa_t *__restrict__ SA, sa_t *__restrict__ SAi, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) SAi[SA[i]] = (sa_t)i;
}

// SA read from a `sa=` file (interface.c:224-232) is untrusted: scatter only in-range entries, then every text position must
// have got its own rank back -- anything else (stale cache, wrong width) is an error, not an out-of-bounds write
__global__ __launch_bounds__(TB) void k_inverse_checked(const sa_t *__restrict__ SA, sa_t *__restrict__ SAi, int64_t n, u32 *__restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const sa_t v = SA[i];
    if ((u64)v < (u64)n) SAi[v] = (sa_t)i; else atomicOr(err, 1u);
}
__global__ __launch_bounds__(TB) void k_inverse_verify(const sa_t *__restrict__ SA, const sa_t *__restrict__ SAi, int64_t n, u32 *__restrict__ err) {
    const int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (j >= n) return;
    const sa_t r = SAi[j];
    if ((u64)r >= (u64)n || SA[r] != (sa_t)j) atomicOr(err, 2u);
}

// ---- LCP ---------------------------------------------------------------------

// LCP[k] = min( lcp(T[SA[k-1]..], T[SA[k]..]), distance from SA[k] to the first
// '$' or 'N' )  -- the closed form of compute_lcp (interface.c:97-114).
// Also emits BWT[k] = T[SA[k]-1] ('$' for SA[k]==0: "nothing to the left" counts
// as left-maximal exactly like a '$', reveal.c:81-85), the byte the scans use
// for the left-maximality test.
__global__ __launch_bounds__(TB) void k_lcp(const uint8_t *__restrict__ T, const sa_t *__restrict__ SA, lcp_t *__restrict__ LCP, int64_t n,
                                            u32 *__restrict__ maxlcp, uint8_t *__restrict__ BWT, sa_t side_sep) {
    const int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    u32 h = 0;
    if (k < n && BWT) { const sa_t p = SA[k]; BWT[k] = (uint8_t)((p > 0 ? T[p - 1] : (uint8_t)'$') | (p > side_sep ? RV_BWT_SIDE : 0)); }
    if (k < n && k > 0) {
        const uint8_t *pa = T + SA[k - 1], *pb = T + SA[k];
        for (;;) {
            const u64 wa = load8(pa + h), wb = load8(pb + h);
            u64 stop = (wa ^ wb);
            stop |= zero_bytes(wb ^ 0x2424242424242424ull);   // '$'
            stop |= zero_bytes(wb ^ 0x4E4E4E4E4E4E4E4Eull);   // 'N'
            stop |= zero_bytes(wb);                            // end of text (zero padding)
            if (stop) { h += (u32)(__builtin_ctzll(stop) >> 3); break; }
            h += 8;
        }
    }
    if (k < n) LCP[k] = (lcp_t)h;
    u32 m = h;
    for (int d = 32; d >= 1; d >>= 1) { u32 o = __shfl_down(m, d, 64); m = o > m ? o : m; }
    // one address for the whole grid: only waves that would raise the maximum go to the atomic unit
    // (an unconditional atomic per wave serialised the kernel: 1.8 ms of 1.9 ms at n = 1e7)
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(maxlcp, __ATOMIC_RELAXED)) atomicMax(maxlcp, m);
}

// The same LCP in text order (Kasai's invariant): if suffix p shares h symbols with its predecessor in
// the suffix array, suffix p+1 shares at least h-1 with its own predecessor -- also with the '$'/'N' stops,
// because the h symbols that matched contained no stop.  One thread walks PHI_K consecutive text positions
// and only ever compares the symbols beyond h-1, so closely related genomes (long matches) cost about
// two word compares per position instead of lcp/8.  PLCP[p] is written in text order; k_lcp_gather turns
// it into LCP[k] = PLCP[SA[k]] and emits BWT and the maximum.
// A lane whose match goes on after the first word (a new long match right behind a difference) does not loop on
// its own -- that would stall the other 63 lanes for lcp/8 steps almost every iteration -- the wave compares 64
// consecutive words of that lane's two suffixes at once.
constexpr int PHI_K = 32, PHI_B = 8;
__device__ inline u64 stop_mask(u64 wa, u64 wb) {
    u64 stop = (wa ^ wb);
    stop |= zero_bytes(wb ^ 0x2424242424242424ull);   // '$'
    stop |= zero_bytes(wb ^ 0x4E4E4E4E4E4E4E4Eull);   // 'N'
    stop |= zero_bytes(wb);                            // end of text (zero padding)
    return stop;
}
// PHI[p] = the suffix in front of suffix p in the suffix array (-1 for rank 0): one scatter in rank order, after which the
// text-order pass reads its partner positions as a stream.  (With the inverse instead -- SAi[p], then SA[SAi[p] - 1] -- the
// pass paid a random gather per position on top of the scatter that built SAi; the inverse itself is not needed by
// construct or by the recursion any more and is made on demand, rv_build_inverse.)
__global__ __launch_bounds__(TB) void k_phi(const sa_t *__restrict__ SA, int64_t n, sa_t *__restrict__ PHI) {
    const int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= n) return;
    PHI[SA[k]] = k > 0 ? SA[k - 1] : (sa_t)-1;
}
// PL[p] = PLCP[p] | (BWT byte of suffix p) << 32: the rank-order pass then fetches both with ONE random 8-byte gather
// per rank instead of one for PLCP[SA[k]] and one for T[SA[k] - 1].
__global__ __launch_bounds__(TB) void k_plcp(const uint8_t *__restrict__ T, const sa_t *__restrict__ PHI, u64 *__restrict__ PL, int64_t n, sa_t side_sep) {
    const int lane = threadIdx.x & 63;
    const int64_t p0 = ((int64_t)blockIdx.x * TB + threadIdx.x) * PHI_K;      // (whole waves run past n together: no early return, ballots below)
    u32 h = 0;
#pragma unroll 1
    for (int b = 0; b < PHI_K; b += PHI_B) {
        int64_t q[PHI_B];
#pragma unroll
        for (int j = 0; j < PHI_B; j++) { const int64_t p = p0 + b + j; q[j] = p < n ? (int64_t)PHI[p] : -1; }
#pragma unroll
        for (int j = 0; j < PHI_B; j++) {
            const int64_t p = p0 + b + j;
            const bool act = p < n && q[j] >= 0;
            bool more = false;
            if (p < n && q[j] < 0) h = 0;                                       // rank 0 has no predecessor
            if (act) {
                const u64 stop = stop_mask(load8(T + q[j] + h), load8(T + p + h));
                if (stop) h += (u32)(__builtin_ctzll(stop) >> 3); else { h += 8; more = true; }
            }
            u64 todo = __ballot(more);
            while (todo) {
                const int l = (int)__builtin_ctzll(todo);
                const int64_t qa = __shfl((long long)q[j], l, 64), pb = __shfl((long long)p, l, 64);
                u32 hl = (u32)__shfl((int)h, l, 64);
                for (;;) {
                    const int64_t off = (int64_t)hl + 8 * lane;
                    // the text is zero padded for 64 bytes only: words that would start beyond count as "end of text"
                    const u64 wa = (qa + off < n + 48) ? load8(T + qa + off) : 0ull;
                    const u64 wb = (pb + off < n + 48) ? load8(T + pb + off) : 0ull;
                    const u64 stop = stop_mask(wa, wb);
                    const u64 bal = __ballot(stop != 0);
                    if (bal) {
                        const int f = (int)__builtin_ctzll(bal);
                        const u32 lo = (u32)__shfl((int)(u32)stop, f, 64), hi = (u32)__shfl((int)(u32)(stop >> 32), f, 64);
                        const u64 st = ((u64)hi << 32) | lo;
                        hl += 8u * (u32)f + (u32)(__builtin_ctzll(st) >> 3);
                        break;
                    }
                    hl += 512;
                }
                if (lane == l) h = hl;
                todo &= todo - 1;
            }
            if (p < n) {
                const u32 bw = (u32)(p > 0 ? T[p - 1] : (uint8_t)'$') | (p > (int64_t)side_sep ? RV_BWT_SIDE : 0u);
                PL[p] = (u64)h | ((u64)bw << 32);
            }
            h = h > 0 ? h - 1 : 0;
        }
    }
}
__global__ __launch_bounds__(TB) void k_lcp_gather(const sa_t *__restrict__ SA, const u64 *__restrict__ PL,
                                                   lcp_t *__restrict__ LCP, int64_t n, u32 *__restrict__ maxlcp, uint8_t *__restrict__ BWT) {
    const int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    u32 h = 0;
    if (k < n) {
        const u64 v = PL[SA[k]];
        h = (u32)v;
        LCP[k] = (lcp_t)h;
        if (BWT) BWT[k] = (uint8_t)(v >> 32);
    }
    u32 m = h;
    for (int d = 32; d >= 1; d >>= 1) { u32 o = __shfl_down(m, d, 64); m = o > m ? o : m; }
    // one address for the whole grid: only waves that would raise the maximum go to the atomic unit
    // (an unconditional atomic per wave serialised the kernel: 1.8 ms of 1.9 ms at n = 1e7)
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(maxlcp, __ATOMIC_RELAXED)) atomicMax(maxlcp, m);
}

__global__ __launch_bounds__(TB) void k_bwt(const uint8_t *__restrict__ T, const sa_t *__restrict__ SA, int64_t n, uint8_t *__restrict__ BWT, sa_t side_sep) {
    const int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (k < n) { const sa_t p = SA[k]; BWT[k] = (uint8_t)((p > 0 ? T[p - 1] : (uint8_t)'$') | (p > side_sep ? RV_BWT_SIDE : 0)); }
}

// ---- a text of at most TINY_N characters (the bubbles `reveal refine` realigns are a few hundred bases) ----
// The general build is two dozen launches and half a dozen host round trips whatever the size: 0.33 ms.  Here (up to TINY_N / 2 characters) every suffix counts the
// suffixes in front of it (byte order, the shorter one first when one is a prefix of the other: what divsufsort gives, interface.c:215-222),
// 256 suffixes per workgroup with the text in LDS, the first eight bytes of a comparison from registers; a second launch writes SA, the
// LCP with its stops (interface.c:97-114), the BWT byte and the largest LCP.  No host round trip.
constexpr int TINY_N = 2048;
__global__ __launch_bounds__(TB) void k_sa_tiny_rank(const uint8_t *__restrict__ T, int n, uint16_t *__restrict__ ord) {
    __shared__ __attribute__((aligned(8))) uint8_t txt[TINY_N + 32];
    for (int k = threadIdx.x; k < n + 32; k += TB) txt[k] = k < n ? T[k] : (uint8_t)0;
    __syncthreads();
    const int i = (int)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    u64 ki = 0, wj = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) { ki |= (u64)txt[i + b] << (8 * b); wj |= (u64)txt[b] << (8 * b); }
    int cnt = 0;
    // the other suffix' first eight bytes are a window that moves a byte per step; the bytes that enter it come eight at a time from one
    // aligned load, so no step waits for LDS (a byte per step: 0.68 ms at n = 2002)
    for (int j0 = 0; j0 < n; j0 += 8) {
        u64 nxt = *reinterpret_cast<const u64 *>(txt + j0 + 8);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int j = j0 + e;
            if (j < n && j != i) {      // (a suffix against itself: a wave would wait for the whole text once per lane)
                const int lim = n - (i > j ? i : j);
                const u64 d = ki ^ wj;
                int x = d ? (__builtin_ctzll(d) >> 3) : 8;
                bool j_less;
                if (x < 8 && x < lim) j_less = (u32)((wj >> (8 * x)) & 0xffu) < (u32)((ki >> (8 * x)) & 0xffu);
                else {
                    if (x >= 8 && lim > 8) {
                        x = 8;
                        while (x + 8 <= lim) {      // eight bytes per step (the two samples of a bubble agree for most of their length)
                            u64 a, b;
                            __builtin_memcpy(&a, txt + i + x, 8);
                            __builtin_memcpy(&b, txt + j + x, 8);
                            if (a != b) { x += __builtin_ctzll(a ^ b) >> 3; break; }
                            x += 8;
                        }
                        while (x < lim && txt[i + x] == txt[j + x]) x++;
                    }
                    j_less = (x < lim) ? (txt[j + x] < txt[i + x]) : (j > i);      // equal as far as the shorter one goes: the shorter one (the later start) is smaller
                }
                cnt += j_less ? 1 : 0;
            }
            wj = (wj >> 8) | (nxt << 56);
            nxt >>= 8;
        }
    }
    ord[cnt] = (uint16_t)i;
}
__global__ __launch_bounds__(TB) void k_sa_tiny_emit(const uint8_t *__restrict__ T, int n, const uint16_t *__restrict__ ord, sa_t *__restrict__ SA, lcp_t *__restrict__ LCP,
                                                     uint8_t *__restrict__ BWT, sa_t side_sep, u32 *__restrict__ d_maxlcp) {
    __shared__ uint8_t txt[TINY_N + 16];
    for (int k = threadIdx.x; k < n + 16; k += TB) txt[k] = k < n ? T[k] : (uint8_t)0;
    __syncthreads();
    const int r = (int)blockIdx.x * TB + threadIdx.x;
    u32 l = 0;
    if (r < n) {
        const int i = ord[r];
        if (r > 0) {
            const int j = ord[r - 1];
            const int lim = n - (i > j ? i : j);
            int x = 0;
            while (x < lim) { const uint8_t c = txt[i + x]; if (c != txt[j + x] || c == '$' || c == 'N') break; x++; }
            l = (u32)x;
        }
        SA[r] = (sa_t)i; LCP[r] = (lcp_t)l;
        BWT[r] = (uint8_t)((i > 0 ? txt[i - 1] : (uint8_t)'$') | ((sa_t)i > side_sep ? RV_BWT_SIDE : 0u));
    }
    const u32 wm = (u32)rv_wave_max_u64((u64)l);
    if ((threadIdx.x & 63) == 0 && wm) atomicMax(d_maxlcp, wm);
}

inline int bitlen(u64 v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

}  // namespace

int rv_build_inverse(Workspace &ws, const sa_t *SA, sa_t *SAi, int64_t n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_inverse, dim3((unsigned)ceil_div(n, TB)), dim3(TB), 0, ws.stream, SA, SAi, n);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_build_inverse_checked(Workspace &ws, const sa_t *SA, sa_t *SAi, int64_t n) {
    if (n <= 0) return 0;
    RV_TRY(ws.misc[1].reserve(64));
    u32 *d_err = ws.misc[1].as<u32>();
    RV_HIP(hipMemsetAsync(d_err, 0, 4, ws.stream));
    RV_HIP(hipMemsetAsync(SAi, 0xFF, (size_t)n * sizeof(sa_t), ws.stream));
    hipLaunchKernelGGL(k_inverse_checked, dim3((unsigned)ceil_div(n, TB)), dim3(TB), 0, ws.stream, SA, SAi, n, d_err);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_inverse_verify, dim3((unsigned)ceil_div(n, TB)), dim3(TB), 0, ws.stream, SA, (const sa_t *)SAi, n, d_err);
    RV_LAUNCH_CHECK();
    u32 err = 0;
    RV_TRY(rv_read_back(ws, &err, d_err, 4));
    if (err) { rv_set_error("the suffix array file does not hold a permutation of 0..n-1 (%s): stale or foreign cache file?", (err & 1u) ? "entry out of range" : "duplicate entries"); return -1; }
    return 0;
}

int rv_build_bwt(Workspace &ws, const uint8_t *T, const sa_t *SA, int64_t n, uint8_t *BWT, sa_t side_sep) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_bwt, dim3((unsigned)ceil_div(n, TB)), dim3(TB), 0, ws.stream, T, SA, n, BWT, side_sep);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_build_lcp(Workspace &ws, const uint8_t *T, const sa_t *SA, bool by_rank, lcp_t *LCP, int64_t n, u32 *d_maxlcp, uint8_t *BWT, sa_t side_sep) {
    SaScratchInUse in_use(ws);
    if (n <= 0) return 0;
    RV_HIP(hipMemsetAsync(d_maxlcp, 0, sizeof(u32), ws.stream));
    if (by_rank || ws.opt.lcp_by_rank) {        // one thread per rank, every pair compared from scratch
        hipLaunchKernelGGL(k_lcp, dim3((unsigned)ceil_div(n, TB)), dim3(TB), 0, ws.stream, T, SA, LCP, n, d_maxlcp, BWT, side_sep);
        RV_LAUNCH_CHECK();
        return 0;
    }
    // text order (Kasai's carry): PHI scatter -> PLCP (+ BWT byte) per position -> one gather per rank.  The SA build is done
    // with its scratch: the two key buffers hold PHI and the packed PLCP.
    DBuf &bphi = ws.sa[1], &bpl = ws.sa[0];
    RV_TRY(bphi.reserve((size_t)n * sizeof(sa_t)));
    RV_TRY(bpl.reserve((size_t)n * 8));
    hipLaunchKernelGGL(k_phi, dim3((unsigned)ceil_div(n, TB)), dim3(TB), 0, ws.stream, SA, n, bphi.as<sa_t>());
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_plcp, dim3((unsigned)ceil_div(ceil_div(n, PHI_K), TB)), dim3(TB), 0, ws.stream, T, (const sa_t *)bphi.as<sa_t>(), bpl.as<u64>(), n, side_sep);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lcp_gather, dim3((unsigned)ceil_div(n, TB)), dim3(TB), 0, ws.stream, SA, (const u64 *)bpl.as<u64>(), LCP, n, d_maxlcp, BWT);
    RV_LAUNCH_CHECK();
    return 0;
}

int rv_build_sa(Workspace &ws, const uint8_t *T, int64_t n, sa_t *SA, RvSaStats *st,
                lcp_t *LCP, uint8_t *BWT, sa_t side_sep, u32 *d_maxlcp, bool *fused_done, const int64_t *seps, int nseps) {
    SaScratchInUse in_use(ws);
    RvSaStats s;
    memset(&s, 0, sizeof s);
    if (fused_done) *fused_done = false;
    if (n <= 0) { if (st) *st = s; return 0; }
    if (n >= ((int64_t)1 << 32) - 2) { rv_set_error("SA build: n >= 2^32-2 not supported yet"); return -1; }
    hipStream_t q = ws.stream;
    if (n <= TINY_N / 2 && LCP && BWT && d_maxlcp && !ws.opt.no_tiny_sa) {      // (measured: 0.17 / 0.23 ms at n = 202 / 602 against 0.33; at n = 2002 the general build's 0.36 wins against 0.62)
        RV_TRY(ws.sa[0].reserve((size_t)TINY_N * 2 + 64));
        RV_HIP(hipMemsetAsync(d_maxlcp, 0, sizeof(u32), q));
        const unsigned nb = (unsigned)ceil_div(n, TB);
        hipLaunchKernelGGL(k_sa_tiny_rank, dim3(nb), dim3(TB), 0, q, T, (int)n, ws.sa[0].as<uint16_t>());
        RV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_sa_tiny_emit, dim3(nb), dim3(TB), 0, q, T, (int)n, (const uint16_t *)ws.sa[0].as<uint16_t>(), SA, LCP, BWT, side_sep, d_maxlcp);
        RV_LAUNCH_CHECK();
        s.rounds = 0; s.sorted_elems = 0;
        if (st) *st = s;
        if (fused_done) *fused_done = true;
        return 0;
    }

    // -- alphabet -> order-preserving dense codes (0 is reserved for "past the end")
    DBuf &d_hist = ws.sa[16], &d_lut = ws.sa[17];
    RV_TRY(d_hist.reserve(264 * sizeof(u32)));
    RV_TRY(d_lut.reserve(256));
    RV_HIP(hipMemsetAsync(d_hist.p, 0, 264 * sizeof(u32), q));
    {
        int64_t blocks = ceil_div(n, (int64_t)TB * 16);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_hist256, dim3((unsigned)blocks), dim3(TB), 0, q, T, n, d_hist.as<u32>());
        RV_LAUNCH_CHECK();
    }
    u32 hist[257];
    RV_TRY(rv_read_back(ws, hist, d_hist.p, sizeof hist));
    uint8_t lut[256];
    int sigma = 0;
    for (int c = 0; c < 256; c++) lut[c] = hist[c] ? (uint8_t)(++sigma) : (uint8_t)0;
    // first key: K symbols as a base-(sigma+1) number.  K = what the text size needs ((sigma-1)^K >= n: about one suffix per
    // key value, so what stays unresolved are twins and repeats), raised as far as the same number of radix passes allows.
    u32 radix = (u32)sigma + 1;
    if (sigma >= 255) radix = 256;
    if (sigma >= 255) for (int c = 0; c < 256; c++) lut[c] = (uint8_t)c;      /* 0 byte never occurs in a C-string text */
    // A digit value only for "past the end" is a sixth value for a five-letter text: two symbols fewer in the same 40 bits (15 instead of
    // 17 at 5 x 10^8 -- sixteen times as many unrelated suffixes that collide in their key and wait for the text round).  When the text
    // ends with the separator, the separator is its smallest character and never stands twice in a row, "past the end" can spell the
    // separator instead: a suffix reaches the text's last '$' before it reaches the padding, and the only other suffixes that agree with
    // it that far stand in front of another '$' -- they go on with a larger character, so the order is the same and no two keys tie on padding.
    const bool short_alphabet = sigma >= 2 && sigma < 255 && T != nullptr && hist[(uint8_t)'$'] > 0 && lut[(uint8_t)'$'] == 1 && hist[256] == 0 && !ws.opt.no_short_alphabet;
    bool ends_with_sep = false;
    if (short_alphabet) { uint8_t last = 0; RV_TRY(rv_read_back(ws, &last, T + (n - 1), 1)); ends_with_sep = last == (uint8_t)'$'; }
    if (short_alphabet && ends_with_sep) {
        for (int c = 0; c < 256; c++) lut[c] = hist[c] ? (uint8_t)(lut[c] - 1) : (uint8_t)0;      // (absent bytes are never looked up)
        radix = (u32)sigma;       // code 0 = '$' = past the end
    }
    // The key's layout (KeyPack): for every field size g the largest K -- what the text size needs ((sigma-1)^K >= n), raised as far as
    // the same number of radix passes allows -- then the g with the most symbols, the fewest bits, the smallest fields.
    int K = 1;
    int bits = 0;
    KeyPack kp;
    {
        const double base = sigma > 2 ? (double)(sigma - 1) : 2.0;
        // (more than two samples: every group holds a homologue per sample, and every unrelated suffix that collides with it is compared
        // with all of them -- sixteen key values per suffix: at 10 x 5 Mbp a fifth radix pass costs 0.4 ms and takes 2.3 off the text round)
        const double want = (double)n * ((seps && nseps > 1) ? 16.0 : 1.0);
        int best_g = 0, best_K = 0, best_bits = 0;
        u64 rg = 1;
        for (int g = 1; g <= KP_MAXG; g++) {
            rg *= radix;
            if (rg > KP_MAXFIELD && g > 1) break;
            const int fb = bitlen(rg - 1);
            auto bits_of = [&](int k) { u64 rl = 1; for (int e = 0; e < k % g; e++) rl *= radix; return (k / g) * fb + (k % g ? bitlen(rl - 1) : 0); };
            bool exact = true;      // the reciprocals of radix^(g - i): checked for every field value
            for (int i = 1; i < g && exact; i++) {
                u64 d = 1; for (int e = 0; e < g - i; e++) d *= radix;
                const u32 m = (u32)(((1ull << 20) + d - 1) / d);
                for (u64 f = 0; f < rg && exact; f++) exact = ((f * m) >> 20) == f / d && f * m < (1ull << 32);
            }
            if (!exact) continue;
            int k = 1; double cap = base;
            while (cap < want && bits_of(k + 1) <= 62) { cap *= base; k++; }
            const int passes = rv_radix_passes(ws, bits_of(k));
            while (bits_of(k + 1) <= 62 && rv_radix_passes(ws, bits_of(k + 1)) == passes) k++;
            const int b = bits_of(k);
            if (k > best_K || (k == best_K && b < best_bits)) { best_g = g; best_K = k; best_bits = b; }
        }
        K = best_K; bits = best_bits;
        u64 rgb = 1; for (int e = 0; e < best_g; e++) rgb *= radix;
        kp.g = best_g; kp.fb = bitlen(rgb - 1); kp.nf = (K + best_g - 1) / best_g; kp.gl = K - (kp.nf - 1) * best_g;
        u64 rl = 1; for (int e = 0; e < kp.gl; e++) rl *= radix;
        kp.fbl = bitlen(rl - 1); kp.bits = bits; kp.rad = radix; kp.fmask = (1u << kp.fb) - 1u; kp.lmask = (1u << kp.fbl) - 1u;
        kp.lscale = (u32)(rgb / rl); kp.fdiv = (1u << 16) / (u32)kp.fb + 1u;
        kp.lastmul = kp.gl == best_g ? (1u << 20) : (u32)(((1ull << 20) + kp.lscale - 1) / kp.lscale);      // (= dmul[gl - 1]: checked above)
        for (int i = 1; i < KP_MAXG; i++) {
            u64 d = 1; for (int e = 0; e < best_g - i; e++) d *= radix;
            kp.dmul[i - 1] = i < best_g ? (u32)(((1ull << 20) + d - 1) / d) : 0u;
        }
    }
    s.sigma = sigma; s.bits = bits; s.k0 = K;        // (bits: of the whole first key)
    RV_HIP(hipMemcpyAsync(d_lut.p, lut, 256, hipMemcpyHostToDevice, q));
    // Fused LCP / BWT (interface.c:97-114 folded into the build).  Related genomes are finished by the first key plus the text
    // round, and both already hold what LCP needs: two suffixes of different groups share less than K symbols -- their LCP is the
    // common prefix of the two keys -- and the members of a group are told apart by a text comparison that finds their common
    // prefix on the way; the byte in front of every suffix rides in the spare top bits of its key.  So LCP and BWT leave in
    // rank order with the suffix array itself, and the three gather / scatter passes of rv_build_lcp (PHI, PLCP, rank-order
    // gather: 95 ms of a 160 ms construct at n = 5e8) are not run.  Whatever the text round cannot finish (ties beyond 4 KB,
    // groups above 64 members: repeats, identical inputs) falls back to rv_build_lcp for the whole index.
    bool fused = LCP && BWT && d_maxlcp && bits <= 48 && !ws.opt.no_fused_lcp;
    KeyDigits kd;
    // upper bits of the keys (k_init_keys): first stop among the K symbols in 5 bits when K allows (8 otherwise), the diagonal hint
    // in what is left between the sort key's last whole digit and that field -- 11 bits at n = 5e8 (nd up to 1022), none at n = 2.2e9 (47-bit keys)
    kd.ly.sortmask = bits >= 64 ? ~0ull : (1ull << bits) - 1;
    kd.ly.at_bits = K <= 30 ? 5 : 8; kd.ly.at_shift = 56 - kd.ly.at_bits;
    kd.ly.nd_shift = (bits + 7) / 8 * 8; kd.ly.nd_bits = 0;      // (the last radix pass looks at a whole 8-bit digit)
    kd.D = (side_sep > 0 && (int64_t)side_sep < n - 1) ? (int64_t)side_sep + 1 : 0;
    kd.ns = 0;
    for (int q2 = 0; q2 < HINT_K - 1; q2++) kd.sep[q2] = std::numeric_limits<int64_t>::max();      // (later samples count as the last one the hint knows: no hint for them)
    for (int q2 = 0; q2 < HINT_K; q2++) kd.Ds[q2] = 0;
    if (seps && nseps > 0 && kd.D > 0) {
        kd.ns = (nseps + 1 < HINT_K + 1) ? nseps + 1 : HINT_K + 1;                                   // (ns - 1 separators are looked at)
        if (kd.ns > HINT_K) kd.ns = HINT_K;
        for (int q2 = 0; q2 < kd.ns - 1; q2++) kd.sep[q2] = seps[q2];
        for (int q2 = 1; q2 < kd.ns; q2++) kd.Ds[q2] = seps[q2 - 1] + 1;
        // every separator the hint does not know makes the samples behind it look like the last known one: their positions would be
        // taken for that sample's and get a wrong diagonal -- the hint is only used for inputs it knows completely
        if (nseps + 1 > HINT_K) kd.ns = 0;
    }
    const bool want_hint = fused && kd.D > 0 && kd.ns >= 2 && kd.ly.at_shift - kd.ly.nd_shift >= 8 && !ws.opt.no_diag && !ws.opt.no_packed_text;
    if (want_hint) kd.ly.nd_bits = std::min(kd.ly.at_shift - kd.ly.nd_shift - 1, 11);
    kd.K = K; kd.kp = kp; kd.stop0 = lut[(uint8_t)'$']; kd.stop1 = lut[(uint8_t)'N'];
    if (fused) RV_HIP(hipMemsetAsync(d_maxlcp, 0, sizeof(u32), q));

    // -- buffers: kept in the workspace (grow-only), a construct() per benchmark step must not pay for hipMalloc
    DBuf &bk0 = ws.sa[0], &bk1 = ws.sa[1], &bv0 = ws.sa[2], &bv1 = ws.sa[3], &bhead = ws.sa[4], &bseed = ws.sa[5], &bgrp = ws.sa[6], &bisa = ws.sa[7],
         &bP0 = ws.sa[8], &bP1 = ws.sa[9], &bG0 = ws.sa[10], &bG1 = ws.sa[11], &btile = ws.sa[12], &bbig = ws.sa[13], &bQb = ws.sa[14], &bPb = ws.sa[15];
    auto freeall = [&]() {};
#define SA_TRY(x) do { int r__ = (x); if (r__) { freeall(); return r__; } } while (0)
#define SA_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { rv_set_error("%s:%d %s", __FILE__, __LINE__, hipGetErrorString(e__)); freeall(); return -1; } } while (0)
    SA_TRY(bk0.reserve((size_t)n * 8)); SA_TRY(bk1.reserve((size_t)n * 8));
    SA_TRY(bv0.reserve((size_t)n * sizeof(sav_t))); SA_TRY(bv1.reserve((size_t)n * sizeof(sav_t)));
    SA_TRY(bhead.reserve((size_t)n + 16));
    // (seed / group ranks: n entries where round 0 writes them per rank; with the twins' leaving only the rounds' lists use them -- m entries, reserved then:
    //  8 of the build's ~54 bytes per position)
    auto need_seed_grp = [&](int64_t cnt) -> int {
        RV_TRY(bseed.reserve((size_t)std::max<int64_t>(cnt, 1) * (sizeof(sav_t) > 4 ? sizeof(sav_t) : 4)));
        return bgrp.reserve((size_t)std::max<int64_t>(cnt, 1) * 4);
    };
    SA_TRY(bisa.reserve((size_t)n * 4));
    const unsigned nblk = (unsigned)ceil_div(n, TB);

    // -- first key, sorted on its K*bits significant bits
    DiagBits dg; dg.stop = dg.exc = dg.lt = nullptr; dg.D = kd.D; dg.tab = 0;
    kd.dtab = nullptr;
    const int64_t nw = (n + 63) / 64;
    DiagSamples dsm; dsm.ns = kd.ns;
    for (int q2 = 0; q2 < HINT_K - 1; q2++) dsm.sep[q2] = kd.sep[q2];
    for (int q2 = 0; q2 < HINT_K; q2++) dsm.Ds[q2] = kd.Ds[q2];
    if (kd.ly.nd_bits > 0) {
        DBuf &bds = ws.sa[20], &bde = ws.sa[21], &bdl = ws.sa[22];
        SA_TRY(bds.reserve((size_t)nw * 8)); SA_TRY(bde.reserve((size_t)nw * 8)); SA_TRY(bdl.reserve((size_t)nw * 8));
        hipLaunchKernelGGL(k_diag_bits, dim3((unsigned)ceil_div(nw, TB)), dim3(TB), 0, q, T, n, dsm, bds.as<u64>(), bde.as<u64>(), bdl.as<u64>(), nw);
        SA_HIP(hipGetLastError());
        dg.stop = bds.as<u64>(); dg.exc = bde.as<u64>(); dg.lt = bdl.as<u64>();
    }
    // two samples with the hint: the twins of the second sample leave before the sort (k_tw_count / k_init_keys / k_heads_publish_tc)
    const bool collapse = fused && kd.ly.nd_bits > 0 && kd.ns == 2 && K < 64 && kd.D > 1 && n > kd.D + 1 && !ws.opt.no_heads_fusion && !ws.opt.no_twin_collapse
                          && (sizeof(sav_t) > 4 || n < ((int64_t)1 << 31));
    int64_t nsort = n;
    const u32 *tw_off = nullptr;
    if (collapse) {
        const int64_t ntiles = ceil_div(n, KEY_TILE);
        DBuf &btw = ws.sa[23];
        SA_TRY(btw.reserve((size_t)(std::max<int64_t>(ntiles, ceil_div(n, TB)) + 1) * 4 + 64));      // (also the flag counts per workgroup of k_heads_publish_tc)
        u32 *tc = btw.as<u32>();
        // what stays of the second sample in the sort, along the diagonals the bit arrays describe right now
        auto count_kept = [&](int tab, u32 *kept_out) -> int {
            hipLaunchKernelGGL(k_tw_count, dim3((unsigned)ceil_div(nw, TB)), dim3(TB), 0, q, (const u64 *)dg.stop, nw, K, kd.D, n, tc, ntiles, tab);
            RV_LAUNCH_CHECK();
            RV_HIP(hipMemsetAsync(tc + ntiles, 0, 4, q));
            RV_TRY(rv_exclusive_sum_u32(ws, tc, tc, ntiles + 1));
            return rv_read_back(ws, kept_out, tc + ntiles, 4);
        };
        u32 kept = 0;
        SA_TRY(count_kept(0, &kept));
        // Most of the second sample stays although the samples are related enough for the hint to have been asked for: it has left the
        // fixed diagonal (indels).  Piecewise diagonals from seeds (k_seed_sample ... k_diag_bits_tab); kept when they let more twins leave.
        const int dt_mode = (int)ws.opt.diag_table;      // -1: when it pays, 0: never, 1: always (test hook)
        const int64_t n2 = n - kd.D;
        if (dt_mode != 0 && n >= 4096 && (dt_mode == 1 || (int64_t)kept * 10 > n2 * 3)) {
            const int64_t mseeds = nw * SEED_SLOTS;
            DBuf &bdt = ws.sa[25];
            const int64_t ndt = (n >> DT_SHIFT) + 1;
            SA_TRY(bdt.reserve((size_t)ndt * 8 + 64));
            int32_t *raw = bdt.as<int32_t>(), *tab = raw + ndt;
            const int did = ws.prof_begin(11 /* RV_K_DIAG_TABLE */, (double)n + (double)mseeds * 12.0);
            hipLaunchKernelGGL(k_seed_sample, dim3((unsigned)ceil_div(nw, TB)), dim3(TB), 0, q, T, n, bk0.as<u64>(), bv0.as<sav_t>(), nw);
            SA_HIP(hipGetLastError());
            int sin1 = 0;
            SA_TRY(rv_radix_sort_pairs<sav_t>(ws, bk0.as<u64>(), bv0.as<sav_t>(), bk1.as<u64>(), bv1.as<sav_t>(), mseeds, 0, 40, &sin1));
            SA_HIP(hipMemsetAsync(raw, 0x7f, (size_t)ndt * 4, q));      // (0x7f7f7f7f: above every vote; the fill writes DT_NONE where nothing is found)
            hipLaunchKernelGGL(k_seed_pairs, dim3((unsigned)ceil_div(mseeds, TB)), dim3(TB), 0, q, (const u64 *)(sin1 ? bk1.as<u64>() : bk0.as<u64>()),
                               (const sav_t *)(sin1 ? bv1.as<sav_t>() : bv0.as<sav_t>()), mseeds, kd.D, kd.D, raw);
            SA_HIP(hipGetLastError());
            hipLaunchKernelGGL(k_dtab_fill, dim3((unsigned)ceil_div(ndt, TB)), dim3(TB), 0, q, T, n, kd.D, kd.D, (const int32_t *)raw, tab, ndt);
            SA_HIP(hipGetLastError());
            DiagTab dt; dt.dtab = tab; dt.S2 = kd.D; dt.D = kd.D;
            hipLaunchKernelGGL(k_diag_bits_tab, dim3((unsigned)ceil_div(nw, TB)), dim3(TB), 0, q, T, n, dt, ws.sa[20].as<u64>(), ws.sa[21].as<u64>(), ws.sa[22].as<u64>(), nw);
            SA_HIP(hipGetLastError());
            ws.prof_end(did);
            u32 kept_tab = 0;
            SA_TRY(count_kept(1, &kept_tab));
            if (dt_mode == 1 || (int64_t)kept_tab * 5 < (int64_t)kept * 4) { kept = kept_tab; kd.dtab = tab; dg.tab = 1; }
            else {      // the table lets no more twins leave than the fixed diagonal (unrelated or rearranged samples): back to the fixed one
                hipLaunchKernelGGL(k_diag_bits, dim3((unsigned)ceil_div(nw, TB)), dim3(TB), 0, q, T, n, dsm, ws.sa[20].as<u64>(), ws.sa[21].as<u64>(), ws.sa[22].as<u64>(), nw);
                SA_HIP(hipGetLastError());
                SA_TRY(count_kept(0, &kept));
            }
        }
        nsort = kd.D + (int64_t)kept;
        tw_off = tc;
        s.diag_table = dg.tab;
    }
    {
        const int64_t iktiles = ceil_div(n, KEY_TILE);
        const dim3 ig((unsigned)std::min<int64_t>(iktiles, 256 * 8));      // persistent: eight workgroups of four waves per CU
#define RV_IK_(G) hipLaunchKernelGGL(k_init_keys<G>, ig, dim3(TB), 0, q, T, n, d_lut.as<uint8_t>(), K, kp, bk0.as<u64>(), bv0.as<sav_t>(), fused ? 1 : 0, \
                                     kd.stop0, kd.stop1, kd.ly, dg, tw_off, iktiles)
        const int ikid = ws.prof_begin(12 /* RV_K_INIT_KEYS */, (double)n + (double)nsort * (8.0 + sizeof(sav_t) + 1.0));
        switch (kp.g) { case 1: RV_IK_(1); break; case 2: RV_IK_(2); break; case 3: RV_IK_(3); break; case 4: RV_IK_(4); break; default: RV_IK_(5); break; }
        ws.prof_end(ikid);
#undef RV_IK_
    }
    SA_HIP(hipGetLastError());
    int in1 = 0;
    SA_TRY(rv_radix_sort_pairs<sav_t>(ws, bk0.as<u64>(), bv0.as<sav_t>(), bk1.as<u64>(), bv1.as<sav_t>(), nsort, 0, bits, &in1));
    s.radix_passes += rv_radix_passes(ws, bits); s.sorted_elems += nsort;
    u64 *ks = in1 ? bk1.as<u64>() : bk0.as<u64>();
    sav_t *vs = in1 ? bv1.as<sav_t>() : bv0.as<sav_t>();
    u64 *kt = in1 ? bk0.as<u64>() : bk1.as<u64>();      // the free pair
    sav_t *vt = in1 ? bv0.as<sav_t>() : bv1.as<sav_t>();

    uint8_t *head = bhead.as<uint8_t>();
    if (!collapse) SA_TRY(need_seed_grp(n));
    u32 *seed = bseed.as<u32>(), *grp = bgrp.as<u32>(), *ISA = bisa.as<u32>();      // (collapse: set again once the list's length is known)
    const u64 *keys_by_rank = ks;          // what the text round reads for the members of a group
    const sav_t *vals_by_rank = vs;
    if (collapse) {
        // rank of a list entry = its index + the flagged entries in front of it
        const int64_t nb = ceil_div(nsort, TC_TILE);
        u32 *bc = ws.sa[23].as<u32>();      // (the tile offsets of k_init_keys are used up)
        hipLaunchKernelGGL(k_tw_flags, dim3((unsigned)nb), dim3(TB), 0, q, (const sav_t *)vs, nsort, bc);
        SA_HIP(hipGetLastError());
        SA_TRY(rv_exclusive_sum_u32(ws, bc, bc, nb));
        sav_t *vexp = reinterpret_cast<sav_t *>(bisa.p);      // (the inverse is only built on demand, after the round-0 list has been made)
        if (sizeof(sav_t) > 4) { SA_TRY(ws.sa[24].reserve((size_t)n * sizeof(sav_t))); vexp = ws.sa[24].as<sav_t>(); }
        const int pbid = ws.prof_begin(13 /* RV_K_PUBLISH */, (double)nsort * 2.0 * (8.0 + sizeof(sav_t)) + (double)n * (1.0 + sizeof(sa_t) + sizeof(lcp_t) + 1.0));
        hipLaunchKernelGGL(k_heads_publish_tc, dim3((unsigned)nb), dim3(TB), 0, q, (const u64 *)ks, (const sav_t *)vs, nsort, (const u32 *)bc, head, LCP, SA, BWT,
                           side_sep, kd, d_maxlcp, ws.opt.no_pub_twins ? 0 : 1, kt, vexp);
        ws.prof_end(pbid);
        SA_HIP(hipGetLastError());
        keys_by_rank = kt; vals_by_rank = vexp;
    } else if (fused && kd.ly.nd_bits > 0 && !ws.opt.no_heads_fusion) {
        hipLaunchKernelGGL(k_heads_publish, dim3((unsigned)ceil_div(n, (int64_t)HP_ITEMS * TB)), dim3(TB), 0, q, (const u64 *)ks, (const sav_t *)vs, n, head, seed, LCP, SA, BWT, side_sep, kd, d_maxlcp,
                           ws.opt.no_pub_twins ? 0 : 1);
        SA_HIP(hipGetLastError());
        SA_TRY(rv_inclusive_max_u32(ws, seed, grp, n));
    } else {
        hipLaunchKernelGGL(k_heads, dim3(nblk), dim3(TB), 0, q, (const u64 *)ks, n, head, seed, fused ? LCP : (lcp_t *)nullptr, kd, fused ? d_maxlcp : (u32 *)nullptr);
        SA_HIP(hipGetLastError());
        SA_TRY(rv_inclusive_max_u32(ws, seed, grp, n));
        {
            KeyDigits kp = kd;
            if (ws.opt.no_pub_twins) kp.ly.nd_bits = 0;      // (test hook: twin pairs go through the text round's first pass instead)
            hipLaunchKernelGGL(k_publish0, dim3(nblk), dim3(TB), 0, q, (const sav_t *)vs, n, SA, (const u64 *)ks, fused ? BWT : (uint8_t *)nullptr, side_sep,
                               kp, fused ? LCP : (lcp_t *)nullptr, head, d_maxlcp, grp);
        }
        SA_HIP(hipGetLastError());
    }
    bool isa_built = false;
    // ISA, when something first asks for it (the radix path of groups above MEDIUM_GROUP, a doubling round): a finished rank is a group of
    // its own -- ISA[SA[r]] = r --, the entries of the CURRENT list carry their group's rank.  (It used to be made from round 0's groups
    // and kept up to date at the end of every round: a random write per position, 21 ms at n = 5 x 10^8, also for inputs whose ties the
    // far round finishes without ever reading it.)  SA holds a group's members at the group's ranks from round 0 on, so every suffix of
    // the text is written once by the first kernel; the second overwrites the list's.
    const sav_t *cur_S = nullptr; const u32 *cur_G = nullptr; int64_t cur_m = 0;
    auto need_isa = [&]() -> int {
        if (isa_built) return 0;
        hipLaunchKernelGGL(k_isa_identity, dim3(nblk), dim3(TB), 0, q, (const sa_t *)SA, n, ISA);
        RV_LAUNCH_CHECK();
        if (cur_m > 0) {
            hipLaunchKernelGGL(k_round_isa, dim3((unsigned)ceil_div(cur_m, TB)), dim3(TB), 0, q, cur_S, cur_G, cur_m, ISA);
            RV_LAUNCH_CHECK();
        }
        isa_built = true;
        return 0;
    };

    // -- compaction of the non-unique suffixes (round 0: from the full arrays)
    int64_t ntile = ceil_div(n, CP_TILE);
    SA_TRY(btile.reserve(std::max((size_t)(ntile + 1) * 8 + 64, (size_t)RT_REGIONS * 4)));      // (also the work-list counters of the text round; twice: the tiles' last heads)
    u32 *tile = btile.as<u32>();
    // compaction of the not yet unique suffixes in two steps, so that a round that leaves nothing behind can stop
    // before it updates ISA
    auto count_unsorted = [&](const uint8_t *hd, int64_t len, int64_t *m_out) -> int {
        const int64_t nt = ceil_div(len, CP_TILE);
        hipLaunchKernelGGL(k_cp_count, dim3((unsigned)nt), dim3(TB), 0, q, hd, len, tile);
        RV_LAUNCH_CHECK();
        // one extra slot so the scan also yields the total
        RV_HIP(hipMemsetAsync(tile + nt, 0, 4, q));
        RV_TRY(rv_exclusive_sum_u32(ws, tile, tile, nt + 1));
        u32 tot = 0;
        RV_TRY(rv_read_back(ws, &tot, tile + nt, 4));
        *m_out = tot;
        return 0;
    };
    auto emit_unsorted = [&](const uint8_t *hd, int64_t len, const u32 *pos_in, const sav_t *suf_in, const u32 *grp_in, u32 *P, sav_t *S, u32 *G,
                             const uint8_t *cls_in = nullptr, uint8_t *cls_out = nullptr, int cls_is = 0) -> int {
        const int64_t nt = ceil_div(len, CP_TILE);
        hipLaunchKernelGGL(k_cp_emit, dim3((unsigned)nt), dim3(TB), 0, q, hd, len, (const u32 *)tile, pos_in, suf_in, grp_in, P, S, G, cls_in, cls_out, cls_is);
        RV_LAUNCH_CHECK();
        return 0;
    };
    int64_t m = 0;
    // (the lists of the not yet unique hold m entries, not n -- 2.4 x 10^7 of 5 x 10^8 at 2 x 250 Mbp: reserved when m is known, the second pair when a
    //  round leaves something behind; they were 16 bytes per position)
    // round-0 suffix list goes to the free value buffer `vt`
    if (collapse) {
        // group ranks from the head flags (k_cp_emit_g)
        u32 *tlast = tile + (ntile + 1);
        hipLaunchKernelGGL(k_cp_count_g, dim3((unsigned)ntile), dim3(TB), 0, q, (const uint8_t *)head, n, tile, tlast);
        SA_HIP(hipGetLastError());
        SA_HIP(hipMemsetAsync(tile + ntile, 0, 4, q));
        SA_TRY(rv_exclusive_sum_u32(ws, tile, tile, ntile + 1));
        SA_TRY(rv_inclusive_max_u32(ws, tlast, tlast, ntile));
        u32 tot = 0;
        SA_TRY(rv_read_back(ws, &tot, tile + ntile, 4));
        m = tot;
        SA_TRY(bP0.reserve((size_t)std::max<int64_t>(m, 1) * 4)); SA_TRY(bG0.reserve((size_t)std::max<int64_t>(m, 1) * 4));
        SA_TRY(need_seed_grp(m));
        seed = bseed.as<u32>(); grp = bgrp.as<u32>();
        if (m > 0) {
            hipLaunchKernelGGL(k_cp_emit_g, dim3((unsigned)ntile), dim3(TB), 0, q, (const uint8_t *)head, n, (const u32 *)tile, (const u32 *)tlast, vals_by_rank,
                               bP0.as<u32>(), vt, bG0.as<u32>());
            SA_HIP(hipGetLastError());
        }
    } else {
        SA_TRY(count_unsorted(head, n, &m));
        SA_TRY(bP0.reserve((size_t)std::max<int64_t>(m, 1) * 4)); SA_TRY(bG0.reserve((size_t)std::max<int64_t>(m, 1) * 4));
        if (m > 0) SA_TRY(emit_unsorted(head, n, nullptr, vals_by_rank, grp, bP0.as<u32>(), vt, bG0.as<u32>()));
    }
    const int64_t m_list0 = m;
    u32 *P = bP0.as<u32>(), *G = bG0.as<u32>(), *Pn = nullptr, *Gn = nullptr;      // (the second pair: reserved by the first round that needs it)
    sav_t *S = vt;          // current list of suffixes (length m)
    sav_t *Sfree = vs;      // the other value buffer
    u64 *kA = ks, *kB = kt; // both key buffers are free from here on
    if (m > 0) { SA_TRY(bbig.reserve((size_t)m + 16)); SA_TRY(bQb.reserve((size_t)m * 4)); SA_TRY(bPb.reserve((size_t)m * 4)); }

    const int lowbits = bitlen((u64)n), highbits = bitlen((u64)(n > 1 ? n - 1 : 1));
    int64_t h = K;
    // What the first key and the text round do not finish.  (1) two samples with the hint: a tied pair of partners is read off the
    // diagonal's marks (k_far_twins: one more "round" over the list, no doubling, LCP and BWT written with it).  (2) the doubling rounds
    // order the rest by ranks; the ranks they touch -- `todo`: the members of groups above MEDIUM_GROUP in the text round, and the list
    // when the first doubling round starts -- get LCP and BWT from the text when SA is complete (k_lcp_list) instead of the whole
    // index going through rv_build_lcp.  RV_NO_FAR_TWINS / RV_NO_LCP_LIST: the old paths.
    const bool can_far = fused && kd.ly.nd_bits > 0 && kd.ns == 2 && dg.stop != nullptr && !ws.opt.no_far_twins;
    const bool can_list = fused && !ws.opt.no_lcp_list;
    bool text_round_ran = false, far_ran = false, partial = false;
    DBuf &btodo = ws.sa[26];
    int64_t ntodo = 0;
    auto todo_add = [&](const u32 *ranks, int64_t cnt) -> int {
        if (cnt <= 0) return 0;
        if ((size_t)(ntodo + cnt) * 4 > btodo.cap) {      // (grow, keeping what is there)
            DBuf nb;
            RV_TRY(nb.reserve((size_t)(ntodo + cnt) * 4 * 2 + 1024));
            if (ntodo) RV_HIP(hipMemcpyAsync(nb.p, btodo.p, (size_t)ntodo * 4, hipMemcpyDeviceToDevice, q));
            RV_HIP(hipStreamSynchronize(q));
            btodo.release();
            btodo = nb;
        }
        RV_HIP(hipMemcpyAsync(btodo.as<u32>() + ntodo, ranks, (size_t)cnt * 4, hipMemcpyDeviceToDevice, q));
        ntodo += cnt;
        return 0;
    };
    // the ordering is by ranks from here on: the fused LCP / BWT survive as "partial" when the list kernel may finish them
    auto leave_fused = [&](const u32 *ranks, int64_t cnt) -> int {
        if (!fused) return 0;
        if (!can_list) { fused = false; return 0; }
        partial = true;
        return todo_add(ranks, cnt);
    };
    bool list_in_todo = false;
    // Groups above MEDIUM_GROUP went through the radix path on the rank of suffix + K in the text round: their members (class "slow") are only
    // 2 K-ordered, everything else the text round left tied agrees for TEXT_LIM symbols.  The doubling rounds below TEXT_LIM / 2 are the slow
    // class' alone: the others keep their groups without a look at ISA (k_round_small) and without a write to it (k_round_isa) -- eight rounds
    // over 1.6 x 10^7 tied entries for a few hundred thousand members of tandem arrays at 2 x 250 Mbp with 2 % repeats.
    DBuf &bcls0 = ws.sa[30], &bcls1 = ws.sa[31];
    uint8_t *cls = nullptr, *cls_next = nullptr;      // class of the current list's entries (1 = slow); NULL: no classes (every entry takes part)
    while (m > 0) {
        if (h >= 2 * n + 2 + 2 * TEXT_LIM) { rv_set_error("SA build: did not converge"); freeall(); return -1; }
        s.rounds++;
        const unsigned mb = (unsigned)ceil_div(m, TB);
        uint8_t *bigflag = bbig.as<uint8_t>();
        cur_S = S; cur_G = G; cur_m = m;
        // (the far round always follows the text round when it may run at all: the text round leaves partners tied on the strength of it,
        //  without having looked at their text -- fo.far_defer)
        const bool far_round = text_round_ran && !far_ran && can_far;
        const bool text_round = !far_round && s.rounds == 1 && h <= 64 && !ws.opt.sa_no_text;
        bool text_big = false;
        if (far_round) {
            far_ran = true;
            DBuf &bR = ws.sa[27], &bM = ws.sa[28];
            const int64_t nw1 = ceil_div(kd.D - 1, (int64_t)64);      // words of the first sample
            SA_TRY(bR.reserve(std::max((size_t)nw * 4 + 64, (size_t)(nw1 * 64 + 64) * 4))); SA_TRY(bM.reserve((size_t)nw * 4 + 64));      // (bR: the reversed words, then the table of k_far_nd)
            hipLaunchKernelGGL(k_far_rev, dim3((unsigned)ceil_div(nw, TB)), dim3(TB), 0, q, (const u64 *)dg.stop, nw, bR.as<u32>());
            SA_HIP(hipGetLastError());
            SA_TRY(rv_inclusive_max_u32(ws, bR.as<u32>(), bM.as<u32>(), nw));
            FarTwins ft; ft.stop = dg.stop; ft.exc = dg.exc; ft.lt = dg.lt; ft.M = bM.as<u32>(); ft.nw = nw; ft.n = n; ft.S2 = kd.D; ft.D = kd.D; ft.dtab = kd.dtab;
            // (the table pays for itself from about a million pairs on: 0.6 ms at 2 x 250 Mbp; the few pairs of a 1 % pair walk the marks themselves)
            const bool nd_table = m >= ((int64_t)1 << 21) || ws.opt.far_table == 1;
            if (nd_table) {
                hipLaunchKernelGGL(k_far_nd, dim3((unsigned)ceil_div(nw1, TB)), dim3(TB), 0, q, ft, bR.as<u32>(), nw1);
                SA_HIP(hipGetLastError());
            }
            hipLaunchKernelGGL(k_far_twins, dim3(mb), dim3(TB), 0, q, T, S, (const u32 *)G, (const u32 *)P, m, ft, nd_table ? (const u32 *)bR.as<u32>() : (const u32 *)nullptr, head, bigflag, SA, LCP, BWT, side_sep, d_maxlcp);
            SA_HIP(hipGetLastError());
        }
        // groups of up to SMALL_GROUP members: sorted by their first thread, in place
        else if (text_round)
        {
            text_round_ran = true;
            FusedOut fo;
            fo.pk.Tp = nullptr; fo.pk.blk = nullptr;
            if (!ws.opt.no_packed_text) {       // 2-bit copy of the text for the comparisons (n/4 bytes + a flag per 128 bases)
                DBuf &bTp = ws.sa[18], &bBlk = ws.sa[19];
                const int64_t nwords = n / 32 + 4, nblk = n / PK_BLOCK + 8;      // (windows read one word / test one block beyond)
                SA_TRY(bTp.reserve((size_t)(nwords + 8) * 8)); SA_TRY(bBlk.reserve((size_t)nblk + 16));
                SA_HIP(hipMemsetAsync(bBlk.p, 0, (size_t)nblk + 16, q));
                SA_HIP(hipMemsetAsync(bTp.as<u64>() + nwords, 0, 64, q));
                hipLaunchKernelGGL(k_pack2, dim3((unsigned)ceil_div(nwords, TB)), dim3(TB), 0, q, T, n, bTp.as<u64>(), bBlk.as<uint8_t>(), nwords);
                SA_HIP(hipGetLastError());
                fo.pk.Tp = bTp.as<u64>(); fo.pk.blk = bBlk.as<uint8_t>();
            }
            fo.LCP = fused ? LCP : (lcp_t *)nullptr; fo.BWT = BWT; fo.keys = keys_by_rank; fo.maxlcp = d_maxlcp; fo.side_sep = side_sep; fo.kd = kd;
            fo.h = (int)h;
            // (fixed diagonal only: along piecewise diagonals most keys without a known agreement stand behind a change of diagonal, and their
            //  texts part within a few symbols -- the text round is the cheaper place for those; 2 x 250 Mbp with indels: 49 against 39 ms)
            fo.far_defer = (can_far && kd.dtab == nullptr) ? 1 : 0;
            // measured (2 x 250 Mbp / 10 x 5 Mbp / 2 x 5 Mbp, ms of the whole build): first-thread pairs + self-ranking larger groups
            // 116-120 / 22.6 / 2.27; everything by the first thread (up to 8 members) 122 / 32.0 / 2.37; everything self-ranking
            // 125 / 22.6 / 2.33; 64-byte steps instead of 32: +14 / +3 / +0.3 (the round is bound by sector traffic, not by the
            // length of its dependent-load chains); 16-byte steps: +4 / -1 / +0.03.  RV_TEXT_MODE: test hook for the other two.
            // (with the diagonal hint: groups of up to four by their first thread, twins from their keys -- mode 3)
            const int tmode = ws.opt.text_mode >= 0 ? (int)ws.opt.text_mode : (kd.ly.nd_bits > 0 ? 3 : 1);
#define RT_LAUNCH(M_) hipLaunchKernelGGL((k_round_text<4, M_>), dim3(mb), dim3(TB), 0, q, T, S, (const u32 *)G, (const u32 *)P, m, head, bigflag, SA, Sfree, fo)
            // (bytes: the list entries in -- group rank, position, suffix, key -- and SA / LCP / BWT / heads / suffix out)
            const int tid = ws.prof_begin(9 /* RV_K_TEXT_ROUND */, (double)m * (4 + 4 + sizeof(sav_t) + 8) + (double)m * (sizeof(sav_t) + sizeof(sa_t) + sizeof(lcp_t) + 2));
            if (tmode == 3 && fused) {
                // (both free here: the list of big-group members comes later, the tile counts of the compaction are used up)
                u32 *work = bQb.as<u32>(), *work_count = tile;
                const u32 reg_cap = (u32)(ceil_div((int64_t)mb, RT_REGIONS) * (TB / 2));      // a group has two entries at least
                SA_HIP(hipMemsetAsync(work_count, 0, RT_REGIONS * 4, q));
                hipLaunchKernelGGL((k_round_text3<4>), dim3(mb), dim3(TB), 0, q, T, S, (const u32 *)G, (const u32 *)P, m, n, head, bigflag, SA, Sfree, fo, work, work_count, reg_cap);
                SA_HIP(hipGetLastError());
                u32 cnt[RT_REGIONS];
                SA_TRY(rv_read_back(ws, cnt, work_count, sizeof cnt));
                u32 mx = 0;
                for (int r = 0; r < RT_REGIONS; r++) mx = std::max(mx, cnt[r]);
                if (mx) {
                    const u32 bpr = (u32)ceil_div((int64_t)mx, TB);
                    hipLaunchKernelGGL((k_round_text3b<4>), dim3(bpr * RT_REGIONS), dim3(TB), 0, q, T, S, (const u32 *)G, m, n, head, SA, fo, (const u32 *)work,
                                       (const u32 *)work_count, reg_cap, bpr);
                }
            }
            else if (tmode == 0) RT_LAUNCH(0); else if (tmode == 2) RT_LAUNCH(2); else RT_LAUNCH(1);
#undef RT_LAUNCH
            SA_HIP(hipGetLastError());
            ws.prof_end(tid);
            hipLaunchKernelGGL(k_medium_back, dim3((unsigned)ceil_div(m, (int64_t)TB * 8)), dim3(TB), 0, q, (const uint8_t *)bigflag, (const sav_t *)Sfree, S, m);
        }
        else {
            // (a doubling round orders by ranks, not by text: no common prefixes come out of it -- every rank of the list is looked at again)
            if (!list_in_todo) { SA_TRY(leave_fused(P, m)); list_in_todo = true; }
            SA_TRY(need_isa());
            const uint8_t *slow = (cls && 2 * h <= TEXT_LIM) ? cls : nullptr;
            hipLaunchKernelGGL(k_round_small, dim3(mb), dim3(TB), 0, q, S, (const u32 *)G, (const u32 *)P, m, n, h, (const u32 *)ISA, head, bigflag, SA, slow);
        }
        SA_HIP(hipGetLastError());
        // members of larger groups: ordered sublist -> radix sort on (group rank, rank of suffix+h) -> back into the list
        if (!far_round) {
            const int64_t nt = ceil_div(m, CP_TILE);
            hipLaunchKernelGGL(k_flag_count, dim3((unsigned)nt), dim3(TB), 0, q, (const uint8_t *)bigflag, m, tile);
            SA_HIP(hipGetLastError());
            SA_HIP(hipMemsetAsync(tile + nt, 0, 4, q));
            SA_TRY(rv_exclusive_sum_u32(ws, tile, tile, nt + 1));
            u32 mbig = 0;
            SA_TRY(rv_read_back(ws, &mbig, tile + nt, 4));
            if (mbig > 0) {
                text_big = text_round;
                SA_TRY(need_isa());
                u32 *Pb = bPb.as<u32>(), *Qb = bQb.as<u32>();
                hipLaunchKernelGGL(k_flag_emit, dim3((unsigned)nt), dim3(TB), 0, q, (const uint8_t *)bigflag, m, (const u32 *)tile, (const u32 *)P, (const sav_t *)S,
                                   (const u32 *)G, n, h, (const u32 *)ISA, Pb, Sfree, Qb, kA);
                SA_HIP(hipGetLastError());
                if (!list_in_todo) SA_TRY(leave_fused(Pb, (int64_t)mbig));      // (the text round wrote nothing for them; a later round's are in the list already)
                // Sfree holds the sublist's suffixes; its partner buffer for the ping-pong is the seed array (free until k_seed)
                sav_t *sb0 = Sfree, *sb1 = reinterpret_cast<sav_t *>(bseed.p);
                int f1 = 0, f2 = 0;
                SA_TRY(rv_radix_sort_pairs<sav_t>(ws, kA, sb0, kB, sb1, mbig, 0, lowbits, &f1));
                u64 *k_in = f1 ? kB : kA, *k_out = f1 ? kA : kB;
                sav_t *s_in = f1 ? sb1 : sb0, *s_out = f1 ? sb0 : sb1;
                SA_TRY(rv_radix_sort_pairs<sav_t>(ws, k_in, s_in, k_out, s_out, mbig, 32, 32 + highbits, &f2));
                const u64 *kS = f2 ? k_out : k_in;
                const sav_t *sS = f2 ? s_out : s_in;
                s.radix_passes += rv_radix_passes(ws, lowbits) + rv_radix_passes(ws, highbits); s.sorted_elems += mbig;
                hipLaunchKernelGGL(k_big_writeback, dim3((unsigned)ceil_div(mbig, TB)), dim3(TB), 0, q, kS, (const u32 *)Pb, (const u32 *)Qb, sS, (int64_t)mbig, S, head, SA);
                SA_HIP(hipGetLastError());
            }
        }
        // what is still not unique?  Nothing: SA is complete (ISA is only an intermediate of this build)
        int64_t m2 = 0;
        SA_TRY(count_unsorted(head, m, &m2));
        if (far_round) s.far_pairs = (m - m2) / 2;
        if (m2 == 0) break;
        if (!can_far && !can_list) fused = false;      // (nothing can finish LCP / BWT for what is left)
        // new group ranks from the heads; ISA, if it exists, follows (made later, it starts from the list of that moment)
        hipLaunchKernelGGL(k_seed, dim3(mb), dim3(TB), 0, q, (const uint8_t *)head, (const u32 *)P, m, seed);
        SA_HIP(hipGetLastError());
        SA_TRY(rv_inclusive_max_u32(ws, seed, grp, m));
        if (isa_built) {
            hipLaunchKernelGGL(k_round_isa, dim3(mb), dim3(TB), 0, q, (const sav_t *)S, (const u32 *)grp, m, ISA, (const u32 *)G);
            SA_HIP(hipGetLastError());
        }
        // next list
        if (!Pn) {
            SA_TRY(bP1.reserve((size_t)std::max<int64_t>(m_list0, 1) * 4)); SA_TRY(bG1.reserve((size_t)std::max<int64_t>(m_list0, 1) * 4));
            Pn = bP1.as<u32>(); Gn = bG1.as<u32>();
        }
        // (classes: made when the text round had groups above MEDIUM_GROUP -- its flag 1 --, carried along afterwards)
        const bool want_cls = (text_round && text_big && !ws.opt.no_slow_class && n > 2 * TEXT_LIM) || cls != nullptr;
        if (want_cls) {
            SA_TRY(bcls0.reserve((size_t)m2 + 64)); SA_TRY(bcls1.reserve((size_t)m2 + 64));
            cls_next = (cls == bcls0.as<uint8_t>()) ? bcls1.as<uint8_t>() : bcls0.as<uint8_t>();
            SA_TRY(emit_unsorted(head, m, P, S, grp, Pn, Sfree, Gn, text_round ? (const uint8_t *)bigflag : (const uint8_t *)cls, cls_next, text_round ? 1 : 0));
            cls = cls_next;
        } else
        SA_TRY(emit_unsorted(head, m, P, S, grp, Pn, Sfree, Gn));
        { u32 *t = P; P = Pn; Pn = t; t = G; G = Gn; Gn = t; }
        { sav_t *t = S; S = Sfree; Sfree = t; }
        m = m2;
        // (what the text round leaves tied agrees for TEXT_LIM symbols -- unless groups above MEDIUM_GROUP went through the radix path on
        //  the rank of suffix + K: then the list is only 2 K-ordered)
        if (text_round && !text_big && !ws.opt.no_text_jump && h < TEXT_LIM && n > 2 * TEXT_LIM) h = TEXT_LIM;
        else if (!far_round) h *= 2;
    }
    if (fused && partial) {
        // LCP and BWT of the ranks the doubling rounds ordered, from the text; too long a common prefix: the whole index the old way
        DBuf &bov = ws.sa[29];
        SA_TRY(bov.reserve(64));
        SA_HIP(hipMemsetAsync(bov.p, 0, 4, q));
        hipLaunchKernelGGL(k_lcp_list, dim3((unsigned)ceil_div(ntodo * 64, TB)), dim3(TB), 0, q, T, n, (const sa_t *)SA, (const u32 *)btodo.as<u32>(), ntodo, LCP, BWT, side_sep,
                           d_maxlcp, bov.as<u32>());
        SA_HIP(hipGetLastError());
        u32 ov = 0;
        SA_TRY(rv_read_back(ws, &ov, bov.p, 4));
        if (ov) fused = false;
        s.lcp_list = ntodo;
    }
#undef SA_TRY
#undef SA_HIP
    freeall();
    if (st) *st = s;
    if (fused_done) *fused_done = fused;
    return 0;
}

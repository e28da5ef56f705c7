// rv_prims.hip -- device-wide primitives for gfx950 (wave64): multi-level
// scans and a stable LSD radix sort with ballot-based wave ranking and
// LDS-staged digit buckets.  Used by the SA build (the reference's divsufsort
// slot, interface.c:215-222) and by the split/compaction steps.
#include "rv_common.h"
#include <string.h>

// ---------------------------------------------------------------------------
// scans
// ---------------------------------------------------------------------------
namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS   = 8;
constexpr int SCAN_TILE    = SCAN_THREADS * SCAN_ITEMS;

template <class T> struct OpSum {
    __device__ static T id() { return (T)0; }
    __device__ static T f(T a, T b) { return a + b; }
};
template <class T> struct OpMax {
    __device__ static T id() { return (T)0; }
    __device__ static T f(T a, T b) { return a > b ? a : b; }
};

template <class T> __device__ inline T shfl_up_t(T v, int d) { return __shfl_up(v, d, 64); }
template <> __device__ inline u64 shfl_up_t<u64>(u64 v, int d) {
    u32 lo = __shfl_up((u32)v, d, 64), hi = __shfl_up((u32)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}

// block-wide exclusive scan of one value per thread; returns the exclusive
// prefix, *total receives the block total (valid in every thread).
template <class T, class Op>
__device__ inline T block_exclusive(T x, T *lds /* >= 4 */, T *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T t = shfl_up_t<T>(inc, d);
        if (lane >= d) inc = Op::f(t, inc);
    }
    T exc = shfl_up_t<T>(inc, 1);
    if (lane == 0) exc = Op::id();
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    T base = Op::id(), tot = Op::id();
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / 64; k++) {
        T v = lds[k];
        if (k < w) base = Op::f(base, v);
        tot = Op::f(tot, v);
    }
    __syncthreads();
    *total = tot;
    return Op::f(base, exc);
}

template <class T, class Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_reduce(const T *__restrict__ in, T *__restrict__ totals, int64_t n) {
    __shared__ T lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    T acc = Op::id();
    if (base + SCAN_ITEMS <= n) {      // a thread's items as vector loads (element by element every load instruction touched SCAN_ITEMS times the lines)
        T v[SCAN_ITEMS];
        __builtin_memcpy(v, in + base, sizeof v);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) acc = Op::f(acc, v[i]);
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++)
            if (base + i < n) acc = Op::f(acc, in[base + i]);
    }
    T tot;
    (void)block_exclusive<T, Op>(acc, lds, &tot);
    if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

template <class T, class Op, bool INCL>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_scan(const T *in, T *out, const T *__restrict__ tile_prefix, int64_t n) {
    __shared__ T lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    const bool whole = base + SCAN_ITEMS <= n;
    if (whole) __builtin_memcpy(v, in + base, sizeof v);
    else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) v[i] = (base + i < n) ? in[base + i] : Op::id();
    }
    T acc = Op::id();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) acc = Op::f(acc, v[i]);
    T tot;
    T pre = block_exclusive<T, Op>(acc, lds, &tot);
    if (tile_prefix) pre = Op::f(tile_prefix[blockIdx.x], pre);
    if (whole) {
        T o[SCAN_ITEMS];
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) { const T nx = Op::f(pre, v[i]); o[i] = INCL ? nx : pre; pre = nx; }
        __builtin_memcpy(out + base, o, sizeof o);
        return;
    }
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        T nx = Op::f(pre, v[i]);
        if (base + i < n) out[base + i] = INCL ? nx : pre;
        pre = nx;
    }
}

template <class T, class Op, bool INCL>
int scan_rec(Workspace &ws, const T *in, T *out, int64_t n, int level) {
    if (n <= 0) return 0;
    const int64_t nt = ceil_div(n, SCAN_TILE);
    if (nt == 1) {
        hipLaunchKernelGGL((k_tile_scan<T, Op, INCL>), dim3(1), dim3(SCAN_THREADS), 0, ws.stream, in, out, (const T *)nullptr, n);
        RV_LAUNCH_CHECK();
        return 0;
    }
    if (level >= 4) { rv_set_error("scan: too many levels"); return -1; }
    RV_TRY(ws.scan_tmp[level].reserve((size_t)nt * sizeof(T)));
    T *tot = ws.scan_tmp[level].as<T>();
    hipLaunchKernelGGL((k_tile_reduce<T, Op>), dim3((unsigned)nt), dim3(SCAN_THREADS), 0, ws.stream, in, tot, n);
    RV_LAUNCH_CHECK();
    RV_TRY((scan_rec<T, Op, false>(ws, tot, tot, nt, level + 1)));
    hipLaunchKernelGGL((k_tile_scan<T, Op, INCL>), dim3((unsigned)nt), dim3(SCAN_THREADS), 0, ws.stream, in, out, (const T *)tot, n);
    RV_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int rv_exclusive_sum_u32(Workspace &ws, const u32 *in, u32 *out, int64_t n) { return scan_rec<u32, OpSum<u32>, false>(ws, in, out, n, 0); }
int rv_exclusive_sum_u64(Workspace &ws, const u64 *in, u64 *out, int64_t n) { return scan_rec<u64, OpSum<u64>, false>(ws, in, out, n, 0); }
int rv_inclusive_max_u32(Workspace &ws, const u32 *in, u32 *out, int64_t n) { return scan_rec<u32, OpMax<u32>, true>(ws, in, out, n, 0); }
int rv_inclusive_max_u64(Workspace &ws, const u64 *in, u64 *out, int64_t n) { return scan_rec<u64, OpMax<u64>, true>(ws, in, out, n, 0); }

// ---------------------------------------------------------------------------
// radix sort: 8-bit digits, 256 threads x 16 keys per block.
//   pass = histogram kernel -> device scan of (digit-major) block histograms
//          -> scatter kernel.
// Stability inside a block comes from the element order (wave, item, lane):
// wave w owns 1024 consecutive keys, item r covers 64 consecutive keys (one
// coalesced load per item).  Ranking inside a wave is done with 8 ballots per
// item (one per digit bit): the lanes holding the same digit form `peers`,
// the lowest of them bumps the wave's LDS bucket counter, every peer takes
// old + popcount(peers below me).
// ---------------------------------------------------------------------------
namespace {

#ifndef RV_RS_THREADS
#define RV_RS_THREADS 256      // (512 x 16 = 8192 keys per block -- a digit's run in the output twice as long -- measures the same: 89 ms of SA build at 5e8 either way)
#endif
constexpr int RS_THREADS = RV_RS_THREADS;
#ifndef RV_RS_ITEMS
#define RV_RS_ITEMS 16
#endif
constexpr int RS_ITEMS   = RV_RS_ITEMS;
constexpr int RS_TILE    = RS_THREADS * RS_ITEMS;
constexpr int RS_WAVES   = RS_THREADS / 64;

// XCD-aware tile order.  The dispatcher hands consecutive workgroups to the eight XCDs in turn, each with its own L2.  A digit's run of
// one tile (16 keys on average: 128 B of keys, 64 B of suffixes) is followed in the output by the same digit's run of the NEXT tile;
// with tile = blockIdx the two runs are written by different XCDs, no L2 ever holds a whole line, and HBM sees partial-line writes.
// XCD: workgroup i takes tile (i % 8) * ceil(nb / 8) + i / 8 -- an XCD works through one contiguous eighth of the tiles, neighbouring
// runs meet in its L2.  (The histogram kernel only reads; it keeps the plain order.)
constexpr unsigned RS_XCDS = 8;
template <bool XCD> __device__ inline u32 rs_tile_of(u32 nb) {
    if (!XCD) return blockIdx.x;
    const u32 chunk = (nb + RS_XCDS - 1) / RS_XCDS;
    return (blockIdx.x % RS_XCDS) * chunk + blockIdx.x / RS_XCDS;
}

// (XCD: the tile order of k_rs_scatter.  The histograms are stored digit-major -- word d * nblocks + tile -- so the 4-byte words of
// consecutive tiles share a line; written by eight XCDs in turn no L2 ever holds the whole line: 375 MB of partial-line writes per pass for 72 MB
// of counts at 2 x 250 Mbp.)
template <int BITS, bool XCD>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const u64 *__restrict__ keys, int64_t n, int shift, u32 dmask, u32 *__restrict__ blockhist, u32 nblocks) {
    constexpr int NB = 1 << BITS;
    __shared__ u32 h[NB];
    const u32 tile = rs_tile_of<XCD>(nblocks);
    if (tile >= nblocks) return;
    for (int k = threadIdx.x; k < NB; k += RS_THREADS) h[k] = 0;
    __syncthreads();
    const int64_t base = (int64_t)tile * RS_TILE;
    u64 key[RS_ITEMS];          // (all loads first: see k_rs_scatter)
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const int64_t i = base + (int64_t)r * RS_THREADS + threadIdx.x;
        key[r] = i < n ? keys[i] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const int64_t i = base + (int64_t)r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(u32)(key[r] >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NB; k += RS_THREADS) blockhist[(size_t)k * nblocks + tile] = h[k];
}

// The same histogram from the digits themselves: the scatter of the pass before left every key's next digit as a byte at the key's new place
// (k_rs_scatter dnext), so the pass reads 1 byte per key instead of the 8-byte key -- sixteen digits per thread in one load.
template <bool XCD>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist_bytes(const uint8_t *__restrict__ digits, int64_t n, u32 *__restrict__ blockhist, u32 nblocks) {
    static_assert(RS_ITEMS == 16, "sixteen digit bytes per thread");
    __shared__ u32 h[256];
    const u32 tile = rs_tile_of<XCD>(nblocks);
    if (tile >= nblocks) return;
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)tile * RS_TILE + (int64_t)threadIdx.x * 16;
    if (i0 + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(digits + i0);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int b = 0; b < 4; b++) atomicAdd(&h[(w[k] >> (8 * b)) & 255u], 1u);
    } else {
        for (int64_t i = i0; i < n && i < i0 + 16; i++) atomicAdd(&h[digits[i]], 1u);
    }
    __syncthreads();
    blockhist[(size_t)threadIdx.x * nblocks + tile] = h[threadIdx.x];
}

// BITS: digit width (8: 256 bins; 10: 1024 bins -- a 40-bit key in four passes instead of five).  CNT: type of the waves' bucket
// counters (a tile holds 4096 keys: 16 bits are enough, and with them three workgroups fit a CU's LDS instead of two).
template <class V, int BITS, bool XCD, class CNT>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const u64 *__restrict__ kin, const V *__restrict__ vin,
                                                            u64 *__restrict__ kout, V *__restrict__ vout, int64_t n, int shift, u32 dmask,
                                                            const u32 *__restrict__ blockoff, u32 nblocks, uint8_t *__restrict__ dnext, int nshift, u32 nmask) {
    constexpr int NB = 1 << BITS;
    constexpr int BPT = NB / RS_THREADS;          // bins per thread in the per-digit steps
    static_assert(NB % RS_THREADS == 0 && RS_TILE < 65536, "bins per thread, 16-bit counters");
    __shared__ CNT cnt[RS_WAVES][NB];
    __shared__ u32 gbase[NB];
    __shared__ CNT dstart[NB];
    __shared__ u32 wtot[RS_WAVES];
    __shared__ u64 skey[RS_TILE];
    __shared__ V sval[RS_TILE];
    const u32 tile = rs_tile_of<XCD>(nblocks);
    if (tile >= nblocks) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < RS_WAVES * NB; k += RS_THREADS) (&cnt[0][0])[k] = 0;
    for (int k = threadIdx.x; k < NB; k += RS_THREADS) gbase[k] = blockoff[(size_t)k * nblocks + tile];
    __syncthreads();

    const int64_t wbase = (int64_t)tile * RS_TILE + (int64_t)w * (64 * RS_ITEMS);
    u64 key[RS_ITEMS];
    V val[RS_ITEMS];
    u32 rnk[RS_ITEMS];
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // Every load of the tile is issued before the first key is ranked.  (Loaded inside the ranking loop, each key waited for its
    // own round trip: the wave barriers around the bucket counters keep the loads of later items behind them -- 16 x ~0.8 us per
    // workgroup, and the same again for the values: 80 % of a workgroup's 28 us.)
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const int64_t i = wbase + (int64_t)r * 64 + lane;
        key[r] = i < n ? kin[i] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const int64_t i = wbase + (int64_t)r * 64 + lane;
        val[r] = i < n ? vin[i] : V(0);
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const int64_t i = wbase + (int64_t)r * 64 + lane;
        const bool valid = i < n;
        const u32 d = (u32)(key[r] >> shift) & dmask;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const u32 below = (u32)__popcll(peers & lt);
        const u32 old = cnt[w][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) cnt[w][d] = (CNT)(old + (u32)__popcll(peers));
        __builtin_amdgcn_wave_barrier();
        rnk[r] = old + below;
    }
    __syncthreads();
    // exclusive prefix over the waves of this block, per digit (a thread takes BPT digits in a row); tot = the block's count of my digits
    u32 tot[BPT], mytot = 0;
#pragma unroll
    for (int b = 0; b < BPT; b++) {
        const int d = threadIdx.x * BPT + b;
        u32 run = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) { const u32 c = cnt[k][d]; cnt[k][d] = (CNT)run; run += c; }
        tot[b] = run; mytot += run;
    }
    // The tile is first ordered by digit in LDS and then written out: a digit's keys of this block go to one contiguous
    // run in global memory, so consecutive threads store consecutive addresses.  (Scattering straight from registers made
    // every store of a wave hit up to 64 different sectors: 200 us per pass for 1e7 keys where the well-clustered top
    // digit took 55.)
    {
        u32 inc = mytot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const u32 t = __shfl_up(inc, dd, 64); if (lane >= dd) inc += t; }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        u32 before = inc - mytot;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) if (k < w) before += wtot[k];
#pragma unroll
        for (int b = 0; b < BPT; b++) { dstart[threadIdx.x * BPT + b] = (CNT)before; before += tot[b]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const int64_t i = wbase + (int64_t)r * 64 + lane;
        if (i < n) {
            const u32 d = (u32)(key[r] >> shift) & dmask;
            const u32 li = (u32)dstart[d] + (u32)cnt[w][d] + rnk[r];
            skey[li] = key[r];
            sval[li] = val[r];
        }
    }
    __syncthreads();
    const int64_t tbase = (int64_t)tile * RS_TILE;
    const u32 nvalid = (u32)((n - tbase) < (int64_t)RS_TILE ? (n - tbase) : (int64_t)RS_TILE);
    for (u32 li = threadIdx.x; li < nvalid; li += RS_THREADS) {
        const u64 k = skey[li];
        const u32 d = (u32)(k >> shift) & dmask;
        const size_t dst = (size_t)gbase[d] + (li - (u32)dstart[d]);
        kout[dst] = k;
        vout[dst] = sval[li];
        if (dnext) dnext[dst] = (uint8_t)((u32)(k >> nshift) & nmask);      // the next pass' digit, for its histogram (k_rs_hist_bytes)
    }
}

template <class V, int BITS, bool XCD, class CNT>
static void rs_scatter_launch(hipStream_t q, u32 nb, const u64 *ki, const V *vi, u64 *ko, V *vo, int64_t n, int shift, u32 dmask, const u32 *bh,
                              uint8_t *dnext, int nshift, u32 nmask) {
    const u32 grid = XCD ? RS_XCDS * ((nb + RS_XCDS - 1) / RS_XCDS) : nb;
    hipLaunchKernelGGL((k_rs_scatter<V, BITS, XCD, CNT>), dim3(grid), dim3(RS_THREADS), 0, q, ki, vi, ko, vo, n, shift, dmask, bh, nb, dnext, nshift, nmask);
}

}  // namespace

int rv_radix_passes(const Workspace &ws, int nbits) {
    const int w = ws.opt.rs_bits == 10 ? 10 : 8;
    return nbits <= 0 ? 0 : (nbits + w - 1) / w;
}

template <class V>
int rv_radix_sort_pairs(Workspace &ws, u64 *k0, V *v0, u64 *k1, V *v1, int64_t n, int bit_lo, int bit_hi, int *result_in_1) {
    *result_in_1 = 0;
    if (n <= 1 || bit_hi <= bit_lo) return 0;
    if (n >= ((int64_t)1 << 32)) { rv_set_error("radix sort: n >= 2^32 not supported"); return -1; }
    const int width = ws.opt.rs_bits == 10 ? 10 : 8;
    const bool xcd = ws.opt.rs_xcd != 0, c16 = ws.opt.rs_cnt16 != 0;
    const u32 nb = (u32)ceil_div(n, RS_TILE);
    RV_TRY(ws.rs_hist.reserve(((size_t)1 << width) * nb * sizeof(u32)));
    u32 *bh = ws.rs_hist.as<u32>();
    // the digits of the next pass as a byte per key (8-bit digits, inputs large enough for the saved reads to matter)
    const bool bytes = width == 8 && !ws.opt.rs_no_digit_bytes && n >= ((int64_t)1 << 20) && bit_hi - bit_lo > width;
    if (bytes) RV_TRY(ws.rs_digits.reserve((size_t)n + 64));
    uint8_t *dig = bytes ? ws.rs_digits.as<uint8_t>() : nullptr;
    bool have_digits = false;
    u64 *ki = k0, *ko = k1;
    V *vi = v0, *vo = v1;
    int flip = 0;
    for (int shift = bit_lo; shift < bit_hi; shift += width) {
        // (the last pass may cover fewer bits: whatever lies above bit_hi never takes part)
        const int wbits = bit_hi - shift < width ? bit_hi - shift : width;
        const u32 dmask = (1u << wbits) - 1u;
        int pid = ws.prof_begin(8 /* RV_K_RADIX_HIST */, (have_digits ? 1.0 : 8.0) * (double)n);
        const u32 hgrid = xcd ? RS_XCDS * ((nb + RS_XCDS - 1) / RS_XCDS) : nb;
        if (have_digits) {
            if (xcd) hipLaunchKernelGGL((k_rs_hist_bytes<true>), dim3(hgrid), dim3(RS_THREADS), 0, ws.stream, (const uint8_t *)dig, n, bh, nb);
            else hipLaunchKernelGGL((k_rs_hist_bytes<false>), dim3(hgrid), dim3(RS_THREADS), 0, ws.stream, (const uint8_t *)dig, n, bh, nb);
        } else if (width == 10) { if (xcd) hipLaunchKernelGGL((k_rs_hist<10, true>), dim3(hgrid), dim3(RS_THREADS), 0, ws.stream, (const u64 *)ki, n, shift, dmask, bh, nb);
                           else hipLaunchKernelGGL((k_rs_hist<10, false>), dim3(hgrid), dim3(RS_THREADS), 0, ws.stream, (const u64 *)ki, n, shift, dmask, bh, nb); }
        else             { if (xcd) hipLaunchKernelGGL((k_rs_hist<8, true>), dim3(hgrid), dim3(RS_THREADS), 0, ws.stream, (const u64 *)ki, n, shift, dmask, bh, nb);
                           else hipLaunchKernelGGL((k_rs_hist<8, false>), dim3(hgrid), dim3(RS_THREADS), 0, ws.stream, (const u64 *)ki, n, shift, dmask, bh, nb); }
        RV_LAUNCH_CHECK();
        ws.prof_end(pid);
        RV_TRY(rv_exclusive_sum_u32(ws, bh, bh, ((int64_t)1 << width) * nb));
        // (the next pass' digit, masked like the pass itself will mask it)
        const int nshift = shift + width;
        const bool more = nshift < bit_hi;
        const u32 nmask = more ? ((1u << (bit_hi - nshift < width ? bit_hi - nshift : width)) - 1u) : 0u;
        uint8_t *dn = (bytes && more) ? dig : nullptr;
        pid = ws.prof_begin(7 /* RV_K_RADIX_SCATTER */, (2.0 * (8.0 + sizeof(V)) + (dn ? 1.0 : 0.0)) * (double)n);
#define RS_GO(B_, X_, C_) rs_scatter_launch<V, B_, X_, C_>(ws.stream, nb, (const u64 *)ki, (const V *)vi, ko, vo, n, shift, dmask, (const u32 *)bh, dn, nshift, nmask)
        if (width == 10) { if (xcd) { if (c16) RS_GO(10, true, uint16_t); else RS_GO(10, true, u32); } else { if (c16) RS_GO(10, false, uint16_t); else RS_GO(10, false, u32); } }
        else             { if (xcd) { if (c16) RS_GO(8, true, uint16_t); else RS_GO(8, true, u32); } else { if (c16) RS_GO(8, false, uint16_t); else RS_GO(8, false, u32); } }
#undef RS_GO
        RV_LAUNCH_CHECK();
        ws.prof_end(pid);
        have_digits = dn != nullptr;
        u64 *tk = ki; ki = ko; ko = tk;
        V *tv = vi; vi = vo; vo = tv;
        flip ^= 1;
    }
    *result_in_1 = flip;
    return 0;
}

// Small host -> device table copy as an ordinary kernel (src = pinned host memory).  A hipMemcpyAsync goes through
// the copy path of the runtime; the kernel that follows it on the stream then started ~28 us late at every level of
// the recursion (cross-queue dependency), a kernel-to-kernel dependency costs ~6 us.
__global__ __launch_bounds__(256) void k_h2d_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int rv_read_back(Workspace &ws, void *dst, const void *dsrc, size_t bytes) {
    RV_TRY(ws.hpin.reserve(bytes < 4096 ? 4096 : bytes));
    if (!ws.ev_rb) RV_HIP(hipEventCreateWithFlags(&ws.ev_rb, hipEventDisableTiming));
    RV_HIP(hipMemcpyAsync(ws.hpin.p, dsrc, bytes, hipMemcpyDeviceToHost, ws.stream));
    RV_HIP(hipEventRecord(ws.ev_rb, ws.stream));
    const hipError_t e = rv_event_wait(ws.ev_rb);
    if (e != hipSuccess) { rv_set_error("rv_read_back: %s", hipGetErrorString(e)); return -1; }
    memcpy(dst, ws.hpin.p, bytes);
    return 0;
}

int rv_h2d_copy(Workspace &ws, const void *pinned_src, void *dst, size_t bytes) {
    const size_t n16 = (bytes + 15) / 16;
    if (n16 == 0) return 0;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_h2d_copy, dim3((unsigned)blocks), dim3(256), 0, ws.stream, (const uint4 *)pinned_src, (uint4 *)dst, n16);
    RV_LAUNCH_CHECK();
    return 0;
}

template int rv_radix_sort_pairs<u32>(Workspace &, u64 *, u32 *, u64 *, u32 *, int64_t, int, int, int *);
template int rv_radix_sort_pairs<u64>(Workspace &, u64 *, u64 *, u64 *, u64 *, int64_t, int, int, int *);

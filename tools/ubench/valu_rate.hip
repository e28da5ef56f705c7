// How many cycles does a wave's instruction take on gfx950?  Eight independent chains of one operation per thread, ITER x UNROLL times, every SIMD
// full (256 CUs x 4 SIMDs x 8 waves): cycles per wave instruction = SIMD cycles / instructions issued on it.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64;
constexpr int CH = 8, ITER = 4096;
#define KERNEL(name, T, INIT, OP)                                                             \
    __global__ __launch_bounds__(256) void name(T *out, u32 s, u32 s2) {                      \
        T v[CH];                                                                              \
        for (int c = 0; c < CH; c++) v[c] = INIT;                                             \
        for (int it = 0; it < ITER; it++) {                                                   \
            _Pragma("unroll") for (int c = 0; c < CH; c++) { OP; }                            \
        }                                                                                     \
        T acc = 0;                                                                            \
        for (int c = 0; c < CH; c++) acc += v[c];                                             \
        if (acc == (T)0x12345) out[threadIdx.x] = acc;                                        \
    }
KERNEL(k_add_u32, u32, threadIdx.x + c, v[c] = v[c] + s)
KERNEL(k_mul_lo, u32, threadIdx.x + c, v[c] = v[c] * s)
KERNEL(k_mul_hi, u32, threadIdx.x + c, v[c] = __umulhi(v[c], s))
KERNEL(k_mad24, u32, threadIdx.x + c, v[c] = __umul24(v[c], s) + s2)
KERNEL(k_lshl_or, u32, threadIdx.x + c, v[c] = (v[c] << 7) | s)
KERNEL(k_shl64, u64, threadIdx.x + c, v[c] = (v[c] << (s & 63)) | 1ull)
KERNEL(k_shr64, u64, (u64)(threadIdx.x + c) << 40, v[c] = (v[c] >> (s & 63)) ^ 0x8000000000000001ull)
KERNEL(k_add64, u64, threadIdx.x + c, v[c] = v[c] + ((u64)s << 20))
KERNEL(k_mad64, u64, threadIdx.x + c, v[c] = (u64)(u32)v[c] * s + v[c])
KERNEL(k_mulhi64, u64, threadIdx.x + c, v[c] = __umul64hi(v[c], ((u64)s << 32) | s2) + 77)
KERNEL(k_clz64, u64, ((u64)threadIdx.x << 33) + c, v[c] = (u64)__builtin_clzll(v[c] | 1) + (v[c] >> 1))
KERNEL(k_popc64, u64, ((u64)threadIdx.x << 33) + c, v[c] = (u64)__popcll(v[c]) + (v[c] << 1))
KERNEL(k_bfe, u32, threadIdx.x * 77 + c, v[c] = ((v[c] >> (s & 31)) & 0xffu) + 3)
KERNEL(k_cndmask, u32, threadIdx.x + c, v[c] = v[c] > s ? s2 : v[c] + 1)
KERNEL(k_fma64, double, (double)(threadIdx.x + c), v[c] = v[c] * 1.0000001 + 0.5)
KERNEL(k_fma32, float, (float)(threadIdx.x + c), v[c] = v[c] * 1.0000001f + 0.5f)
__global__ __launch_bounds__(256) void k_lds_u16(u32 *out, u32 s) {
    __shared__ uint16_t tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = (uint16_t)(i * 7);
    __syncthreads();
    u32 v[CH];
    for (int c = 0; c < CH; c++) v[c] = threadIdx.x * 4 + c;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = tab[(v[c] + s) & 4095];
    }
    u32 acc = 0;
    for (int c = 0; c < CH; c++) acc ^= v[c];
    if (acc == 0x12345) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_lds_b64(u32 *out, u32 s) {
    __shared__ u64 tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = (u64)(i * 7);
    __syncthreads();
    u64 v[CH];
    for (int c = 0; c < CH; c++) v[c] = threadIdx.x + c;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = tab[(v[c] + s) & 2047];
    }
    u64 acc = 0;
    for (int c = 0; c < CH; c++) acc ^= v[c];
    if (acc == 0x12345) out[threadIdx.x] = (u32)acc;
}
int main() {
    void *out; hipMalloc(&out, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;      // eight workgroups of four waves per CU: eight waves per SIMD
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk);
#define RUN(name, T, opsper)                                                                                   \
    {                                                                                                          \
        float best = 1e9f;                                                                                     \
        for (int it = 0; it < 4; it++) {                                                                       \
            hipEventRecord(a, 0);                                                                              \
            hipLaunchKernelGGL(name, dim3(blocks), dim3(256), 0, 0, (T *)out, 3u, 5u);                         \
            hipEventRecord(b, 0); hipEventSynchronize(b);                                                      \
            float ms; hipEventElapsedTime(&ms, a, b); if (it && ms < best) best = ms;                          \
        }                                                                                                      \
        const double waves_per_simd = 8.0, inst = waves_per_simd * CH * ITER * opsper;                         \
        printf("%-12s %.3f ms  -> %.2f cycles per wave instruction (at %d MHz, %d per loop body)\n", #name, best, best * 1e-3 * clk * 1e3 / inst, clk / 1000, opsper); \
    }
    RUN(k_add_u32, u32, 1) RUN(k_mul_lo, u32, 1) RUN(k_mul_hi, u32, 1) RUN(k_mad24, u32, 1) RUN(k_lshl_or, u32, 1)
    RUN(k_shl64, u64, 2) RUN(k_shr64, u64, 3) RUN(k_add64, u64, 2) RUN(k_mad64, u64, 1) RUN(k_mulhi64, u64, 1) RUN(k_clz64, u64, 1) RUN(k_popc64, u64, 1)
    RUN(k_bfe, u32, 2) RUN(k_cndmask, u32, 3) RUN(k_fma64, double, 1) RUN(k_fma32, float, 1)
    {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) { hipEventRecord(a, 0); hipLaunchKernelGGL(k_lds_u16, dim3(blocks), dim3(256), 0, 0, (u32 *)out, 3u); hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (it && ms < best) best = ms; }
        printf("k_lds_u16    %.3f ms  -> %.2f cycles per wave load per CU (4 SIMDs share the LDS)\n", best, best * 1e-3 * clk * 1e3 / (32.0 * CH * ITER));
        best = 1e9f;
        for (int it = 0; it < 4; it++) { hipEventRecord(a, 0); hipLaunchKernelGGL(k_lds_b64, dim3(blocks), dim3(256), 0, 0, (u32 *)out, 3u); hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (it && ms < best) best = ms; }
        printf("k_lds_b64    %.3f ms  -> %.2f cycles per wave load per CU\n", best, best * 1e-3 * clk * 1e3 / (32.0 * CH * ITER));
    }
    return 0;
}

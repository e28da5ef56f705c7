// How fast are non-returning global atomics on a small table?  n increments of pseudo-random bins (uniform over `bins`), per variant:
//   plain: one atomicAdd per element;  agg: a wave merges equal bins first (match by ballot on the low 6 bits is not done: just the plain form and
//   an LDS-privatised form for tables that fit).       hipcc --offload-arch=gfx950 -O3 atomic_hist.hip -o atomic_hist.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ __launch_bounds__(256) void k_plain(uint32_t *tab, int64_t n, uint32_t bins) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&tab[mix((uint64_t)i) % bins], 1u);
}
__global__ __launch_bounds__(256) void k_read(const uint64_t *keys, uint32_t *tab, int64_t n, uint32_t bins) {      // the same with a key read per element (8 B)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&tab[keys[i] % bins], 1u);
}
__global__ __launch_bounds__(256) void k_fill(uint64_t *keys, int64_t n) { const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) keys[i] = mix((uint64_t)i); }
int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 290000000;
    uint32_t *tab; uint64_t *keys;
    hipMalloc(&tab, (size_t)4 << 20 << 2); hipMalloc(&keys, (size_t)n * 8);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, keys, n);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint32_t bins : {256u, 4096u, 65536u, 390625u, 4194304u}) {
        for (int variant = 0; variant < 2; variant++) {
            float best = 1e9f;
            for (int it = 0; it < 4; it++) {
                hipMemsetAsync(tab, 0, (size_t)bins * 4, 0);
                hipEventRecord(a, 0);
                if (variant == 0) hipLaunchKernelGGL(k_plain, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tab, n, bins);
                else hipLaunchKernelGGL(k_read, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint64_t *)keys, tab, n, bins);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (it && ms < best) best = ms;
            }
            printf("bins %8u %s: %.3f ms for %lld atomics = %.1f G/s\n", bins, variant ? "keys read" : "computed ", best, (long long)n, n / best / 1e6);
        }
    }
    return 0;
}

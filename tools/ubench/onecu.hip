// micro-benchmark: what can ONE workgroup stream? (design input for the bubble pass)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT, int EL, bool BARRIER>
__global__ __launch_bounds__(NT) void k_shift(int *a, int64_t n) {   // a[i] = a[i-1] for i in (0, n), top-down
    for (int64_t hi = n - 1; hi > 0;) {
        const int64_t lo = hi - (int64_t)NT * EL + 1 > 1 ? hi - (int64_t)NT * EL + 1 : 1;
        int v[EL];
#pragma unroll
        for (int k = 0; k < EL; k++) { const int64_t idx = hi - (int64_t)k * NT - threadIdx.x; if (idx >= lo) v[k] = a[idx - 1]; }
        if (BARRIER) __syncthreads();
#pragma unroll
        for (int k = 0; k < EL; k++) { const int64_t idx = hi - (int64_t)k * NT - threadIdx.x; if (idx >= lo) a[idx] = v[k]; }
        if (BARRIER) { __threadfence_block(); __syncthreads(); }
        hi = lo - 1;
    }
}
template <int NT>
__global__ __launch_bounds__(NT) void k_copy(const int4 *a, int4 *b, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) b[i] = a[i];
}
__global__ void k_chase(const int *a, int steps, int *out) { int p = 0; for (int i = 0; i < steps; i++) p = a[p]; *out = p; }

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < reps; r++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int64_t n = 4 << 20;   // 4M ints = 16 MB
    int *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 1, n * 4));
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL((k_shift<256, 4, true>), dim3(1), dim3(256), 0, 0, a, n); }, 3);  printf("1 WG shift 256x4  barrier : %.3f ms  %.2f Gelem/s\n", ms, n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_shift<1024, 8, true>), dim3(1), dim3(1024), 0, 0, a, n); }, 3); printf("1 WG shift 1024x8 barrier : %.3f ms  %.2f Gelem/s\n", ms, n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_shift<512, 24, true>), dim3(1), dim3(512), 0, 0, a, n); }, 3);  printf("1 WG shift 512x24 barrier : %.3f ms  %.2f Gelem/s\n", ms, n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_shift<1024, 8, false>), dim3(1), dim3(1024), 0, 0, a, n); }, 3); printf("1 WG shift 1024x8 nobarr  : %.3f ms  %.2f Gelem/s\n", ms, n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_copy<1024>), dim3(1), dim3(1024), 0, 0, (const int4 *)a, (int4 *)b, n / 4); }, 3); printf("1 WG copy int4 1024 thr   : %.3f ms  %.1f GB/s (r+w)\n", ms, 2 * n * 4 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_copy<256>), dim3(1), dim3(256), 0, 0, (const int4 *)a, (int4 *)b, n / 4); }, 3);  printf("1 WG copy int4 256 thr    : %.3f ms  %.1f GB/s (r+w)\n", ms, 2 * n * 4 / ms / 1e6);
    for (int g : {2, 4, 8, 16, 64, 256, 2048}) {
        ms = timeit([&] { hipLaunchKernelGGL((k_copy<256>), dim3(g), dim3(256), 0, 0, (const int4 *)a, (int4 *)b, n / 4); }, 3); printf("%4d WG copy int4 256 thr   : %.3f ms  %.1f GB/s (r+w)\n", g, ms, 2 * n * 4 / ms / 1e6);
    }
    // dependent-load latency
    CK(hipMemset(a, 0, n * 4));
    ms = timeit([&] { hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, a, 1000, b); }, 3); printf("dependent load latency (L2 hit): %.1f ns\n", ms * 1e6 / 1000);
    ms = timeit([&] { hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, a, 1, b); }, 20); printf("empty-ish kernel launch+exec: %.1f us\n", ms * 1e3);
    return 0;
}

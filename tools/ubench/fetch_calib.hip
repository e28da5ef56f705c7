// FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on a known byte count in your
// own access pattern").  Four streaming reads of the SAME 1 GiB, differing only in bytes per lane and load: 16 (dwordx4, the pair scan's LCP
// stream), 8 (dwordx2, its BWT stream), 4 (dword) and 1 (byte); non-temporal like the scan's.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -o f -- tools/ubench/fetch_calib.bin
// and divide each kernel's counter (KB) by 2^20 KB: the factor a byte count has to be multiplied with.  tools/fetch_calib.sh does both.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_read16(const v4u *__restrict__ p, int64_t n, unsigned *out) {
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const v4u v = __builtin_nontemporal_load(p + i); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read8(const v2u *__restrict__ p, int64_t n, unsigned *out) {
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const v2u v = __builtin_nontemporal_load(p + i); acc += v.x ^ v.y; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read4(const unsigned *__restrict__ p, int64_t n, unsigned *out) {
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += __builtin_nontemporal_load(p + i);
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read1(const uint8_t *__restrict__ p, int64_t n, unsigned *out) {
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += __builtin_nontemporal_load(p + i);
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const int64_t bytes = (int64_t)1 << 30;
    uint8_t *buf; unsigned *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes)); CK(hipMemset(out, 0, 64));
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read16, dim3(8192), dim3(256), 0, 0, (const v4u *)buf, bytes / 16, out);
        hipLaunchKernelGGL(k_read8, dim3(8192), dim3(256), 0, 0, (const v2u *)buf, bytes / 8, out);
        hipLaunchKernelGGL(k_read4, dim3(8192), dim3(256), 0, 0, (const unsigned *)buf, bytes / 4, out);
        hipLaunchKernelGGL(k_read1, dim3(8192), dim3(256), 0, 0, (const uint8_t *)buf, bytes, out);
        CK(hipDeviceSynchronize());
    }
    printf("fetch_calib: 4 kernels x 3 launches over %lld bytes\n", (long long)bytes);
    return 0;
}

// Measured issue rate of v_mfma_f64_16x16x4_f64 (MI355X_MICROARCH.md has no fp64 row): every CU,
// 1 or 2 waves per SIMD, 8 independent accumulators per wave, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 profiles/diag/mfma_f64_peak.hip -o profiles/diag/bin/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(double* out, int iters) {
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c7, 0, 0, 0);
    }
    const f64x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    double* out;
    hipMalloc(&out, sizeof(double) * 512 * 4096);
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps, blocks = p.multiProcessorCount, iters = 20000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<<<blocks, threads>>>(out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<blocks, threads>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 16 * 16 * 4 * 8.0 * iters * (threads / 64) * blocks;
        printf("%d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s fp64  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wps, ms,
               flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (8.0 * iters * wps));
    }
    return 0;
}

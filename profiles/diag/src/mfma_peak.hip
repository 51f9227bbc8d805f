// MFMA-only ceiling of the device: every wave issues dependent-free v_mfma_f32_32x32x16_bf16 (or 32x32x2 f32)
// from registers, no memory traffic.  What the chip sustains (clock under matrix load included) next to the
// datasheet 2.5 PFLOP/s.   build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k_bf16(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 7); b[e] = (__bf16)1.0f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0];
    if (s == 123.456f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = (float)(threadIdx.x & 7), b = 1.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0];
    if (s == 123.456f) out[0] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wpc = 4; wpc <= 8; wpc += 4) {          // waves per CU: 4 (one per SIMD) or 8
        const int blocks = 256 * wpc / 4;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k_bf16<4><<<blocks, 256>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 4 * 32.0 * 32 * 16 * 2;
            printf("bf16 32x32x16: %d waves/CU, %d iters x 4 acc: %.3f ms  %.1f TFLOP/s\n", wpc, iters, ms, flop / ms * 1e-9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k_f32<4><<<blocks, 256>>>(out, iters / 2);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * (iters / 2) * 4 * 32.0 * 32 * 2 * 2;
            printf("f32  32x32x2 : %d waves/CU, %d iters x 4 acc: %.3f ms  %.1f TFLOP/s\n", wpc, iters / 2, ms, flop / ms * 1e-9);
        }
    }
    // short bursts, the length of one layer launch (~100 us): does the clock hold over a burst?
    for (int it : {100, 400, 1600}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k_bf16<4><<<512, 256>>>(out, it);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = 512.0 * 4 * it * 4 * 32.0 * 32 * 16 * 2;
            printf("bf16 burst %5d iters: %.4f ms  %.1f TFLOP/s\n", it, ms, flop / ms * 1e-9);
        }
    }
    return 0;
}

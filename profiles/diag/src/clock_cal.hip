// What do s_memtime ticks measure?  One wave spins for N ticks; compare with s_memrealtime (100 MHz) and hipEvents.
// Also: the same under full-chip MFMA load on a second stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1;
    do { t1 = __builtin_amdgcn_s_memtime(); } while (t1 - t0 < ticks);
    out[0] = t1 - t0;
    out[1] = __builtin_amdgcn_s_memrealtime() - r0;
}
__global__ __launch_bounds__(256) void burn(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
    unsigned s = seed + threadIdx.x * 2654435761u;
    for (int e = 0; e < 8; ++e) { s = s * 1664525u + 1013904223u; a[e] = (__bf16)((float)(s >> 8) * 1e-7f - 0.8f); s = s * 1664525u + 1013904223u; b[e] = (__bf16)((float)(s >> 8) * 1e-7f - 0.8f); }
    for (int i = 0; i < iters; ++i)
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    float t = 0.f;
    for (int j = 0; j < 4; ++j) t += acc[j][0];
    if (t == 123.456f) out[0] = t;
}
int main() {
    unsigned long long* out; hipMalloc(&out, 16);
    float* fo; hipMalloc(&fo, 4);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long h[2];
    for (int load = 0; load < 2; ++load) {
        if (load) burn<<<512, 256, 0, s2>>>(fo, 400000, 7);     // ~50 ms of random-data MFMA on every CU
        hipEventRecord(e0, s1);
        spin<<<1, 64, 0, s1>>>(20000000ull, out);
        hipEventRecord(e1, s1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("%s: %llu s_memtime ticks = %llu realtime ticks (100 MHz -> %.3f ms), events %.3f ms -> s_memtime %.1f MHz\n",
               load ? "under MFMA load" : "idle chip", h[0], h[1], h[1] / 1e5, ms, h[0] / (h[1] / 100.0));
        hipDeviceSynchronize();
    }
    // random-data MFMA throughput (vs the constant-data figure of mfma_peak)
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, s1);
        burn<<<512, 256, 0, s1>>>(fo, 20000, 11);
        hipEventRecord(e1, s1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("random-data bf16 MFMA: %.3f ms  %.1f TFLOP/s\n", ms, 512.0 * 4 * 20000 * 4 * 32 * 32 * 16 * 2 / ms * 1e-9);
    }
    return 0;
}

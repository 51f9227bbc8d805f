// How fast can 77.6 MB (layer 1's bf16 output) be written at all?  Streaming 16-byte stores, 512 blocks,
// buffer rotated so that the lines are not resident in L2 / Infinity Cache from the previous pass.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void fill(float4* p, size_t n, float v) {
    const float4 x = {v, v, v, v};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = x;
}
int main() {
    const size_t bytes = 75776ull * 512 * 2, n = bytes / 16;
    const int nbuf = 8;
    char* base; hipMalloc(&base, bytes * nbuf);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 1024, 4096}) {
        for (int rep = 0; rep < 10; ++rep) fill<<<grid, 256>>>((float4*)(base + (rep % nbuf) * bytes), n, 1.f);
        hipDeviceSynchronize();
        float best = 1e9f, sum = 0.f;
        for (int rep = 0; rep < 16; ++rep) {
            hipEventRecord(e0);
            fill<<<grid, 256>>>((float4*)(base + (rep % nbuf) * bytes), n, 2.f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best; sum += ms;
        }
        printf("grid %4d: %.1f MB  mean %.1f us  best %.1f us  -> %.2f TB/s (best)\n", grid, bytes / 1e6, sum / 16 * 1e3, best * 1e3, bytes / best * 1e-9);
    }
    // same buffer every pass (stays in the 256 MiB Infinity Cache)
    for (int rep = 0; rep < 4; ++rep) fill<<<1024, 256>>>((float4*)base, n, 1.f);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 16; ++rep) {
        hipEventRecord(e0);
        fill<<<1024, 256>>>((float4*)base, n, 3.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("same buffer every pass: best %.1f us -> %.2f TB/s\n", best * 1e3, bytes / best * 1e-9);
    return 0;
}

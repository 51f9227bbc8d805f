// Instruction-cache probe (diagnostic, not product): straight-line vector code of S KiB executed R times by every wave of a
// 512-thread block on every CU; cycles per pass as a function of S show the capacity the code of a persistent kernel's tile
// loop must fit (and the price per instruction once it does not).   hipcc --offload-arch=gfx950 -O3 icache_probe.hip -o icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KB>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int reps) {
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
    unsigned long long first = 0, sum = 0;
#pragma nounroll
    for (int r = 0; r < reps; ++r) {
        unsigned long long s0, s1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0)::"memory");
        // 64 bytes per .rept body (16 instructions of 4 bytes): KB * 16 bodies per KiB
        asm volatile(".rept %4\n\t"
                     "v_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %3, %3, %0\n\t"
                     "v_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %3, %3, %0\n\t"
                     "v_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %3, %3, %0\n\t"
                     "v_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %3, %3, %0\n\t"
                     ".endr"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(KB * 16));
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1)::"memory");
        const unsigned long long dt = s1 - s0;
        first = r == 0 ? dt : first;
        sum += r == 0 ? 0 : dt;
    }
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = first;
        out[2 * blockIdx.x + 1] = sum / (reps - 1);
    }
    if (a + b + c + d == 12345.f) out[0] = 0;
}

template <int KB>
void run(unsigned long long* dev, int blocks, int threads) {
    const int reps = 20;
    hipLaunchKernelGGL(probe<KB>, dim3(blocks), dim3(threads), 0, 0, dev, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
    double f = 0, s = 0;
    for (int i = 0; i < blocks; ++i) { f += h[2 * i]; s += h[2 * i + 1]; }
    const double n = KB * 256.0;
    printf("%4d KiB x %3d threads/block: first pass %7.2f cycles/instr, later passes %6.2f cycles/instr (%.0f cycles per pass)\n", KB,
           threads, f / blocks / n, s / blocks / n, s / blocks);
}

int main() {
    unsigned long long* dev;
    hipMalloc(&dev, 2 * 1024 * 8);
    for (int threads : {512, 64}) {
        run<4>(dev, 256, threads); run<8>(dev, 256, threads); run<16>(dev, 256, threads); run<24>(dev, 256, threads);
        run<32>(dev, 256, threads); run<48>(dev, 256, threads); run<56>(dev, 256, threads); run<64>(dev, 256, threads);
        run<80>(dev, 256, threads); run<96>(dev, 256, threads); run<112>(dev, 256, threads);
    }
    return 0;
}

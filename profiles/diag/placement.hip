// Diagnostic: where do the blocks of a 512-block / 64 KiB-LDS launch land?  Prints, for block b,
// the XCC, SE and CU ids, and checks whether blocks b and b+256 share a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k(unsigned* out) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID bits 0..3
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    smem[threadIdx.x] = threadIdx.x;
    // stay resident long enough for the whole grid to be placed
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 2000000ull) { }
    if (smem[threadIdx.x] < 0) out[0] = 0;
}
int main() {
    const int nb = 512;
    unsigned* d; hipMalloc(&d, nb * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<<<nb, 256, 65536>>>(d);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<int>> cu;
    for (int b = 0; b < nb; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        unsigned cu_id = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[((unsigned long long)xcc << 32) | (se << 8) | (sh << 4) | cu_id].push_back(b);
        if (b < 20) printf("block %3d: xcc %u se %u sh %u cu %u (hw_id %08x)\n", b, xcc, se, sh, cu_id, hw);
    }
    printf("distinct CUs used: %zu\n", cu.size());
    int pairs_256 = 0, shown = 0;
    std::map<int, int> hist;
    for (auto& kv : cu) {
        hist[(int)kv.second.size()]++;
        if (kv.second.size() == 2 && kv.second[1] - kv.second[0] == 256) ++pairs_256;
        if (shown++ < 12) { printf("cu %llx:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    }
    for (auto& kv : hist) printf("CUs with %d blocks: %d\n", kv.first, kv.second);
    printf("CUs whose two blocks are (b, b+256): %d\n", pairs_256);
    return 0;
}

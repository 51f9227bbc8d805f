// Layout probe of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4x1): A one-hot in lane la, B[l] = l + 1, C = 0.
// Prints for every la the (register, lane) pairs of D that come out non-zero and which B lane they saw.
//   hipcc --offload-arch=gfx950 -O2 mfma4x4_layout.hip -o mfma4x4_layout && ./mfma4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {   // out[la][reg][lane]
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        const float a = lane == la ? 1.f : 0.f, b = (float)(lane + 1);
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(la * 4 + r) * 64 + lane] = c[r];
    }
}
int main() {
    float* d;
    hipMalloc(&d, 64 * 4 * 64 * 4);
    probe<<<1, 64>>>(d);
    static float h[64 * 4 * 64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; la += 1) {
        if (la > 9 && la % 17 != 0) continue;
        printf("A lane %2d:", la);
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l)
                if (h[(la * 4 + r) * 64 + l] != 0.f) printf(" D[v%d][lane %2d]=B[lane %2d]", r, l, (int)h[(la * 4 + r) * 64 + l] - 1);
        printf("\n");
    }
    return 0;
}

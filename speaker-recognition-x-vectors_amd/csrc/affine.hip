// Segment-level affine layers (reference main.py:45-47,72-75,87-90):
//   y[M,N] = act( x[M,K] . W[N,K]^T + b ),  W in PyTorch nn.Linear layout, exact fp32.
// M is the number of utterances (<= a few hundred), so this is a skinny GEMM: one 32x32
// output tile per 256-thread block, the block's four waves split K (interleaved 8-wide k
// groups) on v_mfma_f32_32x32x2_f32 and combine through LDS.  Operands are read once per
// block straight into registers (16 B per lane); they are small enough to stay in L2.
#include "xvec_internal.h"

namespace xvec {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* row, int k, int K) {
    if (VEC) {
        return (k < K) ? *reinterpret_cast<const float4*>(row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float4 v;
        v.x = (k + 0 < K) ? row[k + 0] : 0.f;
        v.y = (k + 1 < K) ? row[k + 1] : 0.f;
        v.z = (k + 2 < K) ? row[k + 2] : 0.f;
        v.w = (k + 3 < K) ? row[k + 3] : 0.f;
        return v;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void affine_f32_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ W,
                                                         const float* __restrict__ b,
                                                         float* __restrict__ y, int M, int N, int K,
                                                         int relu) {
    __shared__ float red[4][32 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int m = min(m0 + r, M - 1), n = min(n0 + r, N - 1);   // clamped rows: results discarded
    const float* xa = x + (int64_t)m * K;
    const float* wb = W + (int64_t)n * K;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    // lane half h owns k = 8g+4h .. 8g+4h+3 of k-group g (for A and B alike)
    const int groups = (K + 7) / 8;
    int g = wave;
    for (; g + 4 < groups; g += 8) {
        const int k0 = 8 * g + 4 * h, k1 = 8 * (g + 4) + 4 * h;
        const float4 a0 = ld4<VEC>(xa, k0, K), b0 = ld4<VEC>(wb, k0, K);
        const float4 a1 = ld4<VEC>(xa, k1, K), b1 = ld4<VEC>(wb, k1, K);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
    }
    for (; g < groups; g += 4) {
        const int k0 = 8 * g + 4 * h;
        const float4 a0 = ld4<VEC>(xa, k0, K), b0 = ld4<VEC>(wb, k0, K);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
    }

    // accumulator element e of lane (r,h): row = (e&3) + 8*(e>>2) + 4*h, col = r
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = acc[e];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int idx = threadIdx.x + 256 * t;
        const int row = idx >> 5, col = idx & 31;
        const int o = row * 33 + col;
        if (m0 + row < M && n0 + col < N) {
            float v = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]) + b[n0 + col];
            if (relu) v = fmaxf(v, 0.f);
            y[(int64_t)(m0 + row) * N + n0 + col] = v;
        }
    }
}

hipError_t launch_affine_f32(const float* x, const float* W, const float* b, float* y, int M, int N,
                             int K, int relu, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    dim3 grid((N + 31) / 32, (M + 31) / 32);
    const bool vec = (K % 4 == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15) == 0);
    if (vec)
        affine_f32_kernel<true><<<grid, 256, 0, s>>>(x, W, b, y, M, N, K, relu);
    else
        affine_f32_kernel<false><<<grid, 256, 0, s>>>(x, W, b, y, M, N, K, relu);
    return hipGetLastError();
}

}  // namespace xvec

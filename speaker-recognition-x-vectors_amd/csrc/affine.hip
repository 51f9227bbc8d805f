// Segment-level affine layers (reference main.py:45-47,72-75,87-90):
//   y[M,N] = act( x[M,K] . W[N,K]^T + b ),  W in PyTorch nn.Linear layout, exact fp32.
// M is the number of utterances (<= a few hundred), so this is a skinny GEMM: one 32x32
// output tile per 512-thread block, the block's eight waves split K (interleaved 8-wide k
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

constexpr int kAffWaves = 8;

// four k-pairs of one 16-byte operand pair; two accumulators alternate so consecutive MFMAs do
// not wait on each other's result
#define AFF_MFMA4(a_, b_)                                                        \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.x, b_.x, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.y, b_.y, acc1, 0, 0, 0);      \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.z, b_.z, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.w, b_.w, acc1, 0, 0, 0);

template <bool VEC>
__global__ __launch_bounds__(64 * kAffWaves) void affine_f32_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ W,
                                                                    const float* __restrict__ b,
                                                                    float* __restrict__ y, int M, int N, int K,
                                                                    int relu) {
    __shared__ float red[kAffWaves][32 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int m = min(m0 + r, M - 1), n = min(n0 + r, N - 1);   // clamped rows: results discarded
    const float* xa = x + (int64_t)m * K;
    const float* wb = W + (int64_t)n * K;

    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }

    // lane half h owns k = 8g+4h .. 8g+4h+3 of k-group g (for A and B alike); the block's waves
    // take interleaved k-groups, four per trip (ld4 returns zeros past K).  The loads of trip t+1
    // are issued before the MFMAs of trip t: with one wave per SIMD nothing else hides the
    // ~1 us a load takes, and the launch is nothing but such trips (K = 3000: 12 of them).
    const int groups = (K + 7) / 8;
    constexpr int S = kAffWaves;
    float4 a0, a1, a2, a3, b0, b1, b2, b3;
#define AFF_LOAD(g_)                                                                            \
    {                                                                                           \
        const int k0 = 8 * (g_) + 4 * h;                                                        \
        a0 = ld4<VEC>(xa, k0, K); b0 = ld4<VEC>(wb, k0, K);                                     \
        a1 = ld4<VEC>(xa, k0 + 8 * S, K); b1 = ld4<VEC>(wb, k0 + 8 * S, K);                     \
        a2 = ld4<VEC>(xa, k0 + 16 * S, K); b2 = ld4<VEC>(wb, k0 + 16 * S, K);                   \
        a3 = ld4<VEC>(xa, k0 + 24 * S, K); b3 = ld4<VEC>(wb, k0 + 24 * S, K);                   \
    }
    AFF_LOAD(wave)
    for (int g = wave; g < groups; g += 4 * S) {
        const float4 c0 = a0, c1 = a1, c2 = a2, c3 = a3, d0 = b0, d1 = b1, d2 = b2, d3 = b3;
        if (g + 4 * S < groups) AFF_LOAD(g + 4 * S)
        AFF_MFMA4(c0, d0) AFF_MFMA4(c1, d1) AFF_MFMA4(c2, d2) AFF_MFMA4(c3, d3)
    }
#undef AFF_LOAD
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = acc0[e] + acc1[e];

    // accumulator element e of lane (r,h): row = (e&3) + 8*(e>>2) + 4*h, col = r
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = acc[e];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 1024 / (64 * kAffWaves); ++t) {
        const int idx = threadIdx.x + 64 * kAffWaves * t;
        const int row = idx >> 5, col = idx & 31;
        const int o = row * 33 + col;
        if (m0 + row < M && n0 + col < N) {
            float v = b[n0 + col];
#pragma unroll
            for (int w = 0; w < kAffWaves; ++w) v += red[w][o];
            if (relu) v = fmaxf(v, 0.f);
            y[(int64_t)(m0 + row) * N + n0 + col] = v;
        }
    }
}
#undef AFF_MFMA4

hipError_t launch_affine_f32(const float* x, const float* W, const float* b, float* y, int M, int N,
                             int K, int relu, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    dim3 grid((N + 31) / 32, (M + 31) / 32);
    const bool vec = (K % 4 == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15) == 0);
    if (vec)
        affine_f32_kernel<true><<<grid, 64 * kAffWaves, 0, s>>>(x, W, b, y, M, N, K, relu);
    else
        affine_f32_kernel<false><<<grid, 64 * kAffWaves, 0, s>>>(x, W, b, y, M, N, K, relu);
    return hipGetLastError();
}

}  // namespace xvec

// Segment-level affine layers (reference main.py:45-47,72-75,87-90):
//   y[M,N] = act( x[M,K] . W[N,K]^T + b ),  W in PyTorch nn.Linear layout, exact fp32.
// M is the number of utterances (<= a few hundred), so this is a skinny GEMM whose problem is
// filling the chip: one 16x16 output tile per 512-thread block (M=256, N=512: 512 blocks, two per
// CU), the block's eight waves split K on v_mfma_f32_16x16x4_f32 and combine through LDS.  (With
// 32x32 tiles only 128 blocks existed: half the CUs idle, two waves' MFMA chains per SIMD, 28 us
// for segment_layer6; now the chains are a quarter as long and every CU has two blocks.)
// Operands are read once per block straight into registers (16 B per lane, 64 contiguous bytes per
// row and instruction); they are small enough to stay in L2.
#include "xvec_internal.h"

namespace xvec {

typedef float f32x4v __attribute__((ext_vector_type(4)));

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

// the four k-quads of one 16-byte operand pair (lane quarter kq owns k = 16g+4kq .. +3 of k-group
// g, for A and B alike, so element t of every lane belongs to MFMA t of the group); two
// accumulators alternate: the dependent latency of this MFMA (40 cycles) exceeds its issue interval
#define AFF_MFMA4(a_, b_)                                                        \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.x, b_.x, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.y, b_.y, acc1, 0, 0, 0);      \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.z, b_.z, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.w, b_.w, acc1, 0, 0, 0);

template <bool VEC>
__global__ __launch_bounds__(64 * kAffWaves) void affine_f32_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ W,
                                                                    const float* __restrict__ b,
                                                                    float* __restrict__ y, int M, int N, int K,
                                                                    int relu) {
    __shared__ float red[kAffWaves][16 * 17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int m = min(m0 + i, M - 1), n = min(n0 + i, N - 1);   // clamped rows: results discarded
    const float* xa = x + (int64_t)m * K;
    const float* wb = W + (int64_t)n * K;

    f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

    // A trip of a wave is four consecutive k-groups (64 k = 256 contiguous bytes of every row); the
    // block's waves take interleaved trips (ld4 returns zeros past K).  The loads of trip t+1 are
    // issued before the MFMAs of trip t.
    const int trips = (K + 63) / 64;
    constexpr int S = kAffWaves;
    float4 a0, a1, a2, a3, b0, b1, b2, b3;
#define AFF_LOAD(t_)                                                                            \
    {                                                                                           \
        const int k0 = 64 * (t_) + 4 * kq;                                                      \
        a0 = ld4<VEC>(xa, k0, K); b0 = ld4<VEC>(wb, k0, K);                                     \
        a1 = ld4<VEC>(xa, k0 + 16, K); b1 = ld4<VEC>(wb, k0 + 16, K);                           \
        a2 = ld4<VEC>(xa, k0 + 32, K); b2 = ld4<VEC>(wb, k0 + 32, K);                           \
        a3 = ld4<VEC>(xa, k0 + 48, K); b3 = ld4<VEC>(wb, k0 + 48, K);                           \
    }
    AFF_LOAD(wave)
    for (int t = wave; t < trips; t += S) {
        const float4 c0 = a0, c1 = a1, c2 = a2, c3 = a3, d0 = b0, d1 = b1, d2 = b2, d3 = b3;
        if (t + S < trips) AFF_LOAD(t + S)
        AFF_MFMA4(c0, d0) AFF_MFMA4(c1, d1) AFF_MFMA4(c2, d2) AFF_MFMA4(c3, d3)
    }
#undef AFF_LOAD

    // accumulator element e of lane (i, kq): row = 4*kq + e, col = i
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][(4 * kq + e) * 17 + i] = acc0[e] + acc1[e];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
        if (m0 + row < M && n0 + col < N) {
            float v = b[n0 + col];
#pragma unroll
            for (int w = 0; w < kAffWaves; ++w) v += red[w][row * 17 + col];
            if (relu) v = fmaxf(v, 0.f);
            y[(int64_t)(m0 + row) * N + n0 + col] = v;
        }
    }
}
#undef AFF_MFMA4

hipError_t launch_affine_f32(const float* x, const float* W, const float* b, float* y, int M, int N,
                             int K, int relu, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    dim3 grid((N + 15) / 16, (M + 15) / 16);
    const bool vec = (K % 4 == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15) == 0);
    if (vec)
        affine_f32_kernel<true><<<grid, 64 * kAffWaves, 0, s>>>(x, W, b, y, M, N, K, relu);
    else
        affine_f32_kernel<false><<<grid, 64 * kAffWaves, 0, s>>>(x, W, b, y, M, N, K, relu);
    return hipGetLastError();
}

}  // namespace xvec

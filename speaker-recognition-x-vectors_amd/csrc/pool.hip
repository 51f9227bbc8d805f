// Statistics pooling (reference main.py:59-63): per utterance and channel, the mean and the
// UNBIASED standard deviation over the valid frames; out[u] = mean[0..C) ‖ std[0..C).
//
//  * stat_pool_kernel     -- stand-alone, HBM-bound: one pass over x with the first frame of
//                            the utterance as a shift (sums of (x-K), (x-K)^2 are well
//                            conditioned since K is a sample of the same distribution).
//  * pool_finalize_kernel -- merges the per-sub-tile (mean, M2) partials that the layer-5
//                            epilogue (tdnn_layer.hip) writes, with Chan's pairwise update.
//
// n == 1 gives NaN std exactly like torch.std (0/0); the caller rejects n < 1.
#include "xvec_internal.h"

namespace xvec {

template <int VEC>
struct VecT;
template <>
struct VecT<4> { using type = float4; };
template <>
struct VecT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ void ld(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void stat_pool_kernel(const PoolArgs a) {
    __shared__ float red[2][4][64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.y;
    const int ch = (blockIdx.x * 64 + lane) * VEC;
    const bool active = ch < a.C;

    const int64_t off = (int64_t)u * a.T;
    const int n = a.lengths ? a.lengths[u] : a.T;

    float K[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { K[k] = 0.f; s1[k] = 0.f; s2[k] = 0.f; }
    if (active && n > 0) {
        const float* base = a.X + off * (int64_t)a.C + ch;
        ld<VEC>(base, K);
        int f = wave;
        for (; f + 12 < n; f += 16) {   // 4 independent rows in flight per wave
            float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
            ld<VEC>(base + (int64_t)f * a.C, v0);
            ld<VEC>(base + (int64_t)(f + 4) * a.C, v1);
            ld<VEC>(base + (int64_t)(f + 8) * a.C, v2);
            ld<VEC>(base + (int64_t)(f + 12) * a.C, v3);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float d0 = v0[k] - K[k], d1 = v1[k] - K[k], d2 = v2[k] - K[k], d3 = v3[k] - K[k];
                s1[k] += (d0 + d1) + (d2 + d3);
                s2[k] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
        for (; f < n; f += 4) {
            float v0[VEC];
            ld<VEC>(base + (int64_t)f * a.C, v0);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float d0 = v0[k] - K[k];
                s1[k] += d0;
                s2[k] += d0 * d0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        red[0][wave][lane * VEC + k] = s1[k];
        red[1][wave][lane * VEC + k] = s2[k];
    }
    __syncthreads();
    if (wave == 0 && active) {
        const float fn = (float)n;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int i = lane * VEC + k;
            const float t1 = (red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i]);
            const float t2 = (red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i]);
            const float mean = K[k] + t1 / fn;
            const float num = fmaxf(t2 - t1 * t1 / fn, 0.f);
            const float sd = (n > 1) ? sqrtf(num / (fn - 1.f)) : __builtin_nanf("");
            float* o = a.out + (int64_t)u * 2 * a.C;
            o[ch + k] = mean;
            o[a.C + ch + k] = sd;
        }
    }
}

hipError_t launch_stat_pool(const PoolArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.C <= 0) return hipSuccess;
    const bool vec4 = (a.C % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
    if (vec4) {
        dim3 grid((a.C / 4 + 63) / 64, a.B);
        stat_pool_kernel<4><<<grid, 256, 0, s>>>(a);
    } else {
        dim3 grid((a.C + 63) / 64, a.B);
        stat_pool_kernel<1><<<grid, 256, 0, s>>>(a);
    }
    return hipGetLastError();
}

// One thread per (utterance, channel): walk the sub-tiles the utterance's pooled rows touch,
// recompute each one's row count from the geometry (the producer does not store it) and merge.
__global__ __launch_bounds__(256) void pool_finalize_kernel(const PoolFinalizeArgs a) {
    const int u = blockIdx.y;
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= a.C) return;
    const int64_t off = row_off(a.map, u), end = row_off(a.map, u + 1);   // pooled rows are [off, end)
    if (a.scale != nullptr) {
        // raw-sum partials of tdnn_pp.hip: (S1, S2) = sums of r and r^2, r = relu(z + bias), per (sub-tile,
        // utterance); y = scale*r + shift.  Totals and the difference S2 - S1^2/n in fp64.
        double s1 = 0.0, s2 = 0.0;
        for (int64_t sub = off / a.sub_rows; sub * a.sub_rows < end; ++sub) {
            const float* p = a.part + (sub + u) * (int64_t)(2 * a.n_pad);
            s1 += (double)p[ch];
            s2 += (double)p[a.n_pad + ch];
        }
        const double n = (double)(end - off);
        const double sc = (double)a.scale[ch], sh = (double)a.shift[ch];
        const double mean_r = s1 / n;
        double var_r = (s2 - s1 * mean_r) / (n - 1.0);
        var_r = var_r > 0.0 ? var_r : 0.0;
        float* o = a.out + (int64_t)u * 2 * a.C;
        o[ch] = (float)(sh + sc * mean_r);
        o[a.C + ch] = (n > 1.0) ? (float)((sc < 0.0 ? -sc : sc) * sqrt(var_r)) : __builtin_nanf("");
        return;
    }
    float n = 0.f, mean = 0.f, m2 = 0.f;
    // four sub-tiles per trip, their eight loads issued before the first merge: the kernel is a chain
    // of ~10 dependent round trips per thread otherwise (12 us for 31 MB of partials)
    for (int64_t sub0 = off / a.sub_rows; sub0 * a.sub_rows < end; sub0 += 4) {
        float nb[4], mb[4], m2b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t sub = sub0 + j;
            const int64_t lo = sub * a.sub_rows > off ? sub * a.sub_rows : off;
            const int64_t hi = (sub + 1) * a.sub_rows < end ? (sub + 1) * a.sub_rows : end;
            nb[j] = hi > lo ? (float)(hi - lo) : 0.f;
            mb[j] = 0.f;
            m2b[j] = 0.f;
            if (hi > lo) {
                const float* p = a.part + (sub + u) * (int64_t)(2 * a.n_pad);
                mb[j] = p[ch];
                m2b[j] = p[a.n_pad + ch];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (nb[j] > 0.f) {
                const float nn = n + nb[j];
                const float delta = mb[j] - mean;
                mean += delta * (nb[j] / nn);
                m2 += m2b[j] + delta * delta * (n * nb[j] / nn);
                n = nn;
            }
        }
    }
    float* o = a.out + (int64_t)u * 2 * a.C;
    o[ch] = mean;
    o[a.C + ch] = (n > 1.f) ? sqrtf(fmaxf(m2, 0.f) / (n - 1.f)) : __builtin_nanf("");
}

hipError_t launch_pool_finalize(const PoolFinalizeArgs& a, hipStream_t s) {
    if (a.map.n_utts <= 0) return hipSuccess;
    dim3 grid((a.C + 255) / 256, a.map.n_utts);
    pool_finalize_kernel<<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace xvec

// Statistics pooling (reference main.py:59-63): per utterance and channel, the mean and the
// UNBIASED standard deviation over the valid frames; out[u] = mean[0..C) ‖ std[0..C).
//
//  * stat_pool_kernel     -- stand-alone, HBM-bound: one pass over x with the first frame of
//                            the utterance as a shift (sums of (x-K), (x-K)^2 are well
//                            conditioned since K is a sample of the same distribution).
//  * pool_finalize_kernel -- merges the per-sub-tile pivoted partials that the layer-5
//                            epilogues (tdnn_layer.hip) write, in fp64; pool_finalize_seg_kernel the segment partials of tdnn_pp16.hip.
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

// One thread per (utterance, channel): merge the partials (K, S1, S2) -- pivot, sum (r - K), sum (r - K)^2 over the
// utterance's n_g frames in the sub-tile, r = relu(z + bias) -- that the layer-5 epilogues (tdnn_layer.hip,
// bf16x3 and small-batch bf16 included) wrote per (32-row sub-tile, utterance), then apply the folded BatchNorm y = scale*r + shift.
// Every sub-tile's sums are re-based to the pivot K0 of the utterance's first sub-tile in fp64
// (d = K - K0:  S1' = S1 + n_g*d,  S2' = S2 + 2*d*S1 + n_g*d^2: multiply-adds only), so
//   mean = shift + scale*(K0 + S1'/n),   std = |scale| * sqrt((S2' - S1'^2/n) / (n-1))
// cancel in fp64 about a sample of the channel; the fp32 sums inside a partial are sums of deviations
// (tdnn_common.h, pool_group_impl).  torch.std (main.py:61) is two-pass: this matches it to fp32 rounding of
// the inputs also for |mean| >> std.  Four sub-tiles per trip, their loads issued before the first add: the kernel
// is a chain of ~10 dependent round trips per thread otherwise.
__global__ __launch_bounds__(256) void pool_finalize_kernel(const PoolFinalizeArgs a) {
    const int u = blockIdx.y;
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= a.C) return;
    const int64_t off = row_off(a.map, u), end = row_off(a.map, u + 1);   // pooled rows are [off, end)
    double s1 = 0.0, s2 = 0.0, k0 = 0.0;
    const int64_t sub_first = off / a.sub_rows;
    const int64_t sub_end = (end + a.sub_rows - 1) / a.sub_rows;          // sub-tiles [sub_first, sub_end)
    for (int64_t sub0 = sub_first; sub0 < sub_end; sub0 += 4) {
        float pk[4], p1[4], p2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pk[j] = 0.f;
            p1[j] = 0.f;
            p2[j] = 0.f;
            if (sub0 + j < sub_end) {
                const float* p = a.part + (sub0 + j + u) * (int64_t)(3 * a.n_pad);
                pk[j] = p[ch];
                p1[j] = p[a.n_pad + ch];
                p2[j] = p[2 * a.n_pad + ch];
            }
        }
        if (sub0 == sub_first) k0 = (double)pk[0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (sub0 + j < sub_end) {
                const int64_t lo = (sub0 + j) * a.sub_rows > off ? (sub0 + j) * a.sub_rows : off;
                const int64_t hi = (sub0 + j + 1) * a.sub_rows < end ? (sub0 + j + 1) * a.sub_rows : end;
                const double ng = (double)(hi - lo), d = (double)pk[j] - k0, t1 = (double)p1[j];
                s1 += t1 + ng * d;
                s2 += (double)p2[j] + d * (2.0 * t1 + ng * d);
            }
        }
    }
    const double n = (double)(end - off);
    const double sc = (double)a.scale[ch], sh = (double)a.shift[ch];
    const double mean_d = s1 / n;                       // mean of (r - K0)
    double var_r = (s2 - s1 * mean_d) / (n - 1.0);
    var_r = var_r > 0.0 ? var_r : 0.0;
    float* o = a.out + (int64_t)u * 2 * a.C;
    o[ch] = (float)(sh + sc * (k0 + mean_d));
    // n == 1 gives NaN like torch.std (0/0)
    o[a.C + ch] = (n > 1.0) ? (float)((sc < 0.0 ? -sc : sc) * sqrt(var_r)) : __builtin_nanf("");
}

// The same merge for the segment partials of tdnn_pp16.hip: one partial per (block b of the column, utterance u, frames
// half g) at slot 2 (b + u) + g, n frames behind it in cnt[slot] (possibly none), for every block whose row range
// overlaps the utterance.
__global__ __launch_bounds__(256) void pool_finalize_seg_kernel(const PoolFinalizeArgs a) {
    const int u = blockIdx.y;
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= a.C) return;
    const int64_t off = row_off(a.map, u), end = row_off(a.map, u + 1);   // pooled rows are [off, end)
    // block holding unit x: the largest b with floor(units_total * b / blocks_per_col) <= x
    const int64_t U = a.units_total, P = a.blocks_per_col;
    const int b_lo = (int)((((off >> 6) + 1) * P - 1) / U), b_hi = (int)(((((end - 1) >> 6) + 1) * P - 1) / U);
    double s1 = 0.0, s2 = 0.0, k0 = 0.0;
    bool have = false;
    for (int b = b_lo; b <= b_hi; ++b) {
        int n[2];
        float pk[2], p1[2], p2[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int64_t slot = 2 * ((int64_t)b + u) + g;
            const float* p = a.part + slot * (int64_t)(3 * a.n_pad);
            n[g] = a.cnt[slot];
            pk[g] = p[ch];
            p1[g] = p[a.n_pad + ch];
            p2[g] = p[2 * a.n_pad + ch];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (n[g] > 0) {
                if (!have) { k0 = (double)pk[g]; have = true; }
                const double ng = (double)n[g], d = (double)pk[g] - k0, t1 = (double)p1[g];
                s1 += t1 + ng * d;
                s2 += (double)p2[g] + d * (2.0 * t1 + ng * d);
            }
        }
    }
    const double n = (double)(end - off);
    const double sc = (double)a.scale[ch], sh = (double)a.shift[ch];
    const double mean_d = s1 / n;
    double var_r = (s2 - s1 * mean_d) / (n - 1.0);
    var_r = var_r > 0.0 ? var_r : 0.0;
    float* o = a.out + (int64_t)u * 2 * a.C;
    o[ch] = (float)(sh + sc * (k0 + mean_d));
    o[a.C + ch] = (n > 1.0) ? (float)((sc < 0.0 ? -sc : sc) * sqrt(var_r)) : __builtin_nanf("");
}

hipError_t launch_pool_finalize(const PoolFinalizeArgs& a, hipStream_t s) {
    if (a.map.n_utts <= 0) return hipSuccess;
    dim3 grid((a.C + 255) / 256, a.map.n_utts);
    if (a.cnt) pool_finalize_seg_kernel<<<grid, 256, 0, s>>>(a);
    else pool_finalize_kernel<<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace xvec

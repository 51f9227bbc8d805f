// Device helpers shared by the frame-level kernels (tdnn_layer.hip, tdnn_bf16.hip).
#pragma once
#include "xvec_internal.h"

namespace xvec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks b, b+8, b+16.. share an XCD (round-robin dispatch); give each XCD a
    // contiguous run of logical ids (bijective for any nwg).
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// Fused statistics-pooling partial (main.py:59-63) of one 32-row group held in one accumulator:
// for every utterance overlapping compact rows [row_g, row_g+32), the mean and M2 (sum of squared
// deviations about that mean) of this lane's column over the utterance's frames in the group.
template <bool RAGGED>
__device__ __forceinline__ void pool_group_impl(const TdnnArgs& a, const f32x16& v, int64_t row_g, int h, int col) {
    const int64_t grp = row_g >> 5;
    const RowMap& m = a.out_map;
    const int64_t t_out = m.fixed_T - m.cum;
    auto first_row = [&](int u) -> int64_t {
        if (RAGGED) return m.offsets[u] - (int64_t)u * m.cum;
        return (int64_t)u * t_out;
    };
    int u0;
    if (RAGGED) {
        u0 = utt_of_row(m, row_g);
    } else {
        const int64_t q = row_g / t_out;
        u0 = (int)(q < m.n_utts - 1 ? q : m.n_utts - 1);
    }
    for (int u = u0; u < m.n_utts; ++u) {
        const int64_t off = first_row(u);
        if (off >= row_g + 32) break;
        const int64_t end = first_row(u + 1);
        const int64_t lo_r = off > row_g ? off : row_g;
        const int64_t hi_r = end < row_g + 32 ? end : row_g + 32;
        if (hi_r <= lo_r) continue;
        const int lo_l = (int)(lo_r - row_g), hi_l = (int)(hi_r - row_g);   // local rows [lo_l, hi_l)
        float s = 0.f, m2 = 0.f, mean;
        if (lo_l == 0 && hi_l == 32) {        // whole group inside one utterance: no masks
#pragma unroll
            for (int e = 0; e < 16; ++e) s += v[e];
            s += __shfl_xor(s, 32);
            mean = s * (1.f / 32.f);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = v[e] - mean;
                m2 = fmaf(d, d, m2);
            }
        } else {
            const float inv_cnt = 1.f / (float)(hi_l - lo_l);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int lr = (e & 3) + 8 * (e >> 2) + 4 * h;
                s += (lr >= lo_l && lr < hi_l) ? v[e] : 0.f;
            }
            s += __shfl_xor(s, 32);
            mean = s * inv_cnt;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int lr = (e & 3) + 8 * (e >> 2) + 4 * h;
                const float d = v[e] - mean;
                m2 += (lr >= lo_l && lr < hi_l) ? d * d : 0.f;
            }
        }
        m2 += __shfl_xor(m2, 32);
        if (h == 0) {
            float* part = a.pool_part + (grp + u) * (int64_t)(2 * a.ldy);
            part[col] = mean;
            part[a.ldy + col] = m2;
        }
    }
}

// (two code paths: see set_tile_rows in tdnn_layer.hip)
__device__ __forceinline__ void pool_group(const TdnnArgs& a, const f32x16& v, int64_t row_g, int h, int col) {
    if (a.out_map.offsets == nullptr) pool_group_impl<false>(a, v, row_g, h, col);
    else pool_group_impl<true>(a, v, row_g, h, col);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    // uniform by construction (kernel argument + blockIdx-derived offset); readfirstlane makes
    // that provable so hipcc emits no waterfall loop around the buffer loads
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, (short)0, 0x7fffffff, 0x00020000);
}


}  // namespace xvec

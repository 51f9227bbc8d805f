// EXPERIMENT RECORD (round 3), not compiled into the library: see profiles/experiments/README.md.
// This block sat in csrc/tdnn_first.hip between namespace first3 and tdnn_first3_applicable(); it uses that file's helpers
// (first::Cur, first::first_row_of, make_rsrc_bounded, utt_of_row).
// ---- exact fp32 (the headline path): layer 1 on v_mfma_f32_32x32x2_f32 with the weights in registers.  In the 128x128
// kernel this layer runs at 104 TFLOP/s (K = 120 padded to 128, a K loop of two chunks per tile, 89 us per 256 x 300
// frames); at the matrix rate its 9.3 GFLOP are 59 us.  Here a wave owns 64 channels and keeps their 64 x 120 weights in 120
// VGPRs (B operand of k-step ks, lane (channel r, half h): W[r][60 h + ks] -- the two k of an MFMA are 60 apart, so a lane's
// values are CONTIGUOUS in k and both operands are read 16 bytes at a time); eight waves per block, one block per CU; the
// 512 threads stage a 32-frame group of windows (fp32, no conversion) in a [32][132] LDS tile (row stride = 4 banks: the
// 16-byte fragment reads of a 16-lane group cover all 64 banks); per group and wave 120 MFMAs, no K padding.
namespace first32 {

using first::Cur;
using first::first_row_of;
using first::kConstFloats;

constexpr int kKH = 60;                    // k-steps: k = ks and 60 + ks
constexpr int kLd = 132;                   // LDS floats per frame
constexpr int kTileF = 32 * kLd;

struct Staged {
    float4 q0, q1;                         // 8 floats of the caller's rows
};

template <bool RAGGED>
__device__ __forceinline__ void fetch(const TdnnArgs& a, int64_t g, Cur& cu, int rr, int sk, Staged& st) {
    const RowMap& m = a.out_map;
    const int64_t m0 = g * 32;
    const int n_last = m.n_utts - 1;
    while (cu.end <= m0 && cu.u < n_last) {
        cu.u = __builtin_amdgcn_readfirstlane(cu.u + 1);
        cu.end = first_row_of<RAGGED>(m, cu.u + 1);
    }
    int c = 0;                             // boundaries inside the group: frame rr lies c utterances past cu.u
    {
        int u = cu.u;
        int64_t nxt = cu.end;
        while (nxt < m0 + 32 && u < n_last) {
            c += (m0 + rr >= nxt) ? 1 : 0;
            u = __builtin_amdgcn_readfirstlane(u + 1);
            nxt = first_row_of<RAGGED>(m, u + 1);
        }
    }
    const int64_t row0 = m0 + (int64_t)cu.u * a.span;
    const int64_t total = a.x_bytes ? a.x_bytes : a.x_rows * (int64_t)a.ldx * 4;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc_bounded(a.X, row0 * a.ldx * 4, total);
    const int voff = ((rr + c * a.span) * a.ldx + 8 * sk) * 4;
    st.q0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 0, 0));
    st.q1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 16, 0));
}

// blank the K tail (values past kpt belong to the next frame and meet zero weights, but 0 x Inf is not 0) and park
__device__ __forceinline__ void park(const TdnnArgs& a, float* tile, int rr, int sk, Staged st) {
    const int k0 = 8 * sk;
    if (k0 + 8 > a.kpt) {
        float* f = reinterpret_cast<float*>(&st);
#pragma unroll
        for (int d = 0; d < 8; ++d)
            if (k0 + d >= a.kpt) f[d] = 0.f;
    }
    *reinterpret_cast<float4*>(tile + rr * kLd + k0) = st.q0;
    *reinterpret_cast<float4*>(tile + rr * kLd + k0 + 4) = st.q1;
}

template <bool RAGGED>
__global__ __launch_bounds__(512) void tdnn_first32_kernel(const TdnnArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[2 * kTileF + kConstFloats];
    float* cst = smem + 2 * kTileF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    cst[tid] = a.bias[tid];
    cst[512 + tid] = a.scale[tid];
    cst[1024 + tid] = a.shift[tid];
    // this wave's weights: accumulator cg = channels 64*wave + 32*cg + r; packed fp32 rows of k_pad floats (pack.hip)
    float4 w0[kKH / 4], w1[kKH / 4];
    {
        const float* W = static_cast<const float*>(a.W);
        const float4* p0 = reinterpret_cast<const float4*>(W + (int64_t)(64 * wave + r) * a.k_pad + kKH * h);
        const float4* p1 = reinterpret_cast<const float4*>(W + (int64_t)(64 * wave + 32 + r) * a.k_pad + kKH * h);
#pragma unroll
        for (int j = 0; j < kKH / 4; ++j) {
            w0[j] = p0[j];
            w1[j] = p1[j];
        }
    }
    const int64_t g_begin = a.groups_total * (int64_t)blockIdx.x / gridDim.x;
    const int64_t g_end = a.groups_total * (int64_t)(blockIdx.x + 1) / gridDim.x;
    if (g_begin >= g_end) return;
    Cur cu;
    cu.u = __builtin_amdgcn_readfirstlane(utt_of_row(a.out_map, g_begin * 32));
    cu.end = first_row_of<RAGGED>(a.out_map, cu.u + 1);
    const int rr = tid >> 4, sk = tid & 15;        // staging: frame rr, floats 8 sk .. 8 sk + 7 of its window
    Staged sa, sb;                                 // two groups of look-ahead (see tdnn_first_kernel)
    fetch<RAGGED>(a, g_begin, cu, rr, sk, sa);
    park(a, smem, rr, sk, sa);
    if (g_begin + 1 < g_end) fetch<RAGGED>(a, g_begin + 1, cu, rr, sk, sb);
    __syncthreads();

    const float* frag = smem + r * kLd + kKH * h;  // A operand of lane (frame r, half h): k = 60 h + ks
    const float bi0 = cst[64 * wave + r], bi1 = cst[64 * wave + 32 + r];
    const float sc0 = cst[512 + 64 * wave + r], sc1 = cst[512 + 64 * wave + 32 + r];
    const float sh0 = cst[1024 + 64 * wave + r], sh1 = cst[1024 + 64 * wave + 32 + r];
    const int y_voff = (4 * h * a.ldy + 64 * wave + r) * 4;      // accumulator element e: frame (e&3) + 8*(e>>2) + 4*h
#define XF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#define XF32_GROUP(g_, buf_, ST_PARK, ST_FETCH)                                                                   \
    {                                                                                                             \
        if ((g_) + 2 < g_end) fetch<RAGGED>(a, (g_) + 2, cu, rr, sk, ST_FETCH);                                   \
        f32x16 acc0, acc1;                                                                                        \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;                                   \
        const float* tile = frag + (buf_) * kTileF;                                                               \
        _Pragma("unroll") for (int j = 0; j < kKH / 4; ++j) {                                                     \
            const float4 x = *reinterpret_cast<const float4*>(tile + 4 * j);                                      \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, w0[j].x, acc0, 0, 0, 0);                             \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, w1[j].x, acc1, 0, 0, 0);                             \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, w0[j].y, acc0, 0, 0, 0);                             \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, w1[j].y, acc1, 0, 0, 0);                             \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, w0[j].z, acc0, 0, 0, 0);                             \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, w1[j].z, acc1, 0, 0, 0);                             \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, w0[j].w, acc0, 0, 0, 0);                             \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, w1[j].w, acc1, 0, 0, 0);                             \
        }                                                                                                         \
        if ((g_) + 1 < g_end) park(a, smem + ((buf_) ^ 1) * kTileF, rr, sk, ST_PARK);                             \
        /* bias + ReLU + folded BatchNorm (tdnn_layer.py:30-39): 128 contiguous bytes of two rows per store */   \
        const __amdgpu_buffer_rsrc_t yr = make_rsrc(static_cast<char*>(a.Y) + (g_) * 32 * (int64_t)a.ldy * 4);    \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                          \
            const float v0 = fmaf(fmaxf(acc0[e] + bi0, 0.f), sc0, sh0);                                           \
            const float v1 = fmaf(fmaxf(acc1[e] + bi1, 0.f), sc1, sh1);                                           \
            const int so = ((e & 3) + 8 * (e >> 2)) * a.ldy * 4;                                                  \
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), yr, y_voff, so, 0);                        \
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), yr, y_voff, so + 128, 0);                  \
        }                                                                                                         \
        XF_LDS_BARRIER()                                                                                          \
    }
    for (int64_t g = g_begin; g < g_end; g += 2) {
        XF32_GROUP(g, 0, sb, sa)
        if (g + 1 < g_end) XF32_GROUP(g + 1, 1, sa, sb)
    }
#undef XF32_GROUP
#undef XF_LDS_BARRIER
}

}  // namespace first32

bool tdnn_first32_applicable(const TdnnArgs& a) {
    return a.ldy == 512 && a.k_pad == 128 && a.n_taps == 1 && a.kpt <= 2 * first32::kKH && a.terms == 1 &&
           (a.ldx * 4) % 16 == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(a.W) & 15) == 0 && a.groups_total > 0;
}

hipError_t launch_tdnn_first32(const TdnnArgs& a, int num_cu, hipStream_t s) {
    const int grid = (int)(a.groups_total < num_cu ? a.groups_total : num_cu);
    if (a.out_map.offsets != nullptr) first32::tdnn_first32_kernel<true><<<grid, 512, 0, s>>>(a);
    else first32::tdnn_first32_kernel<false><<<grid, 512, 0, s>>>(a);
    return hipGetLastError();
}


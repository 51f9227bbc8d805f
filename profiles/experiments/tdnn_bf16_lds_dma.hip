// Frame-level TDNN layer, bf16 operands / fp32 accumulation, large-tile variant for gfx950.
//
// Same math and data layout as tdnn_layer.hip (implicit GEMM over the compact frame axis, folded
// BatchNorm epilogue, optional fused statistics pooling); different machine mapping, because at
// bf16 rates (v_mfma_f32_32x32x16_bf16: 32 cycles for 32768 FLOP, 16x the fp32 MFMA) the 128x128
// tile of tdnn_layer.hip is bound by LDS traffic (5 fragment reads per 4 MFMAs, 8 ds_write_b128
// per chunk and thread):
//   * block = 256 threads = 4 wave64, tile 128 frames x 256 channels, TWO blocks per CU (72 KiB LDS
//     each): wave w owns all 128 frames x channels [64w, 64w+64): 4x2 accumulators of 32x32
//     (128 VGPRs), 6 fragment reads per 8 MFMAs.  (A 512-thread 256x256 block, one per CU, ran its
//     two waves per SIMD in lockstep -- both in the barrier / load phase, then both competing for
//     the matrix pipe -- and was no faster than the 128x128 kernel; two independent blocks per CU
//     de-correlate.)
//   * K chunks of 32 bf16 (64-byte rows, two MFMA k-steps) go global -> LDS directly (buffer_load ...
//     lds, 16 B per lane, 16 rows per wave instruction): no staging registers, no ds_write.  The
//     XOR swizzle of the 16-byte chunks is applied on the SOURCE address (the LDS image of a wave
//     instruction is lane-linear), the fragment reads apply the same XOR;
//   * a ring of three 24 KiB LDS stages: the loads of chunk it+2 are issued at the top of chunk it
//     and a COUNTED s_waitcnt vmcnt leaves one chunk in flight across the single barrier of the
//     chunk; the barrier sits before the last k-step so that 8 MFMAs cover the first fragment
//     reads of the next chunk;
//   * bf16 output tiles are transposed through LDS (free after the K loop) and stored as whole
//     128-byte rows with dwordx4 stores;
//   * persistent grid over equal row ranges per 256-channel column at 32-row granularity; a tile
//     has up to 4 row groups.
#include "tdnn_common.h"

namespace xvec {

namespace big {

constexpr int kBM = 128, kBN = 256;
constexpr int kRowBytes = 64;                         // one K chunk of one row: 32 bf16
constexpr int kStageBytes = (kBM + kBN) * kRowBytes;  // A tile then B tile: 24 KiB
constexpr int kStages = 3;
constexpr int kLdsBytes = kStages * kStageBytes;      // 72 KiB
constexpr int kThreads = 256;
constexpr int kEpiRowBytes = 128;                     // transposed output image: 64 bf16 per wave row

typedef __attribute__((address_space(3))) void* lds_ptr;

struct Lane {
    int h, r, wave;
    int a_rd, b_rd;        // fragment read byte offsets (row part) within a stage
    int sw;                // (r >> 2) & 3
    float bias0, scale0, shift0, bias1, scale1, shift1;   // epilogue constants of cols r and 32+r of the wave
    int col0;              // first of this lane's two output channels
};

typedef int i32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor as four SGPR words (inline-asm operand); built from readfirstlane'd halves
// so hipcc can prove it wave-uniform
__device__ __forceinline__ i32x4 make_srd(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    i32x4 d;
    d.x = (int)__builtin_amdgcn_readfirstlane((unsigned)v);
    d.y = (int)(__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) & 0xffffu);   // stride 0
    d.z = 0x7fffffff;                                                            // num_records (bytes)
    d.w = 0x00020000;
    return d;
}

struct Stream {
    i32x4 xrsrc, wrsrc;
    int xo0, xo1;             // per-lane source byte offsets of the wave's two A load instructions
    int wo0, wo1, wo2, wo3;   // and of its four W load instructions
    int lds_a, lds_b;         // wave-uniform LDS byte offsets (within a stage) of the first A / W instruction
};

// Direct global -> LDS copy of 16 rows x 64 B by one wave instruction (LDS-DMA).  Written as inline
// asm on purpose: with the builtin, hipcc 7.2 puts s_waitcnt vmcnt(0) in front of the next ds_read
// (it cannot tell the DMA's LDS destination from the fragment reads), which serialises the
// prefetch.  The asm is invisible to its wait bookkeeping; completion is awaited explicitly
// (counted vmcnt) before the block barrier that publishes the stage.  M0 (LDS destination base) is
// saved and restored inside the statement.
__device__ __forceinline__ void glds16(const i32x4& rsrc, const char* lds_dst, int voff, int soff) {
    const unsigned dst = (unsigned)(unsigned long long)(lds_ptr)lds_dst;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff))
        : "memory");
}

#define BG_ACC_DECL f32x16 acc00, acc01, acc10, acc11, acc20, acc21, acc30, acc31;
#define BG_FRAG_DECL float4 fa0_0, fa1_0, fa2_0, fa3_0, fb0_0, fb1_0, fa0_1, fa1_1, fa2_1, fa3_1, fb0_1, fb1_1;

#define BG_KO(s_) ((((2 * (s_)) + ln.h) ^ ln.sw) << 4)
#define BG_RD(dst_, base_, off_) dst_ = *reinterpret_cast<const float4*>((base_) + (off_));
// fragment reads of k-step s_ into fragment set f_ from stage base S_
#define BG_FRAGS(s_, f_, S_)                                              \
    {                                                                     \
        const char* fp_ = (S_) + BG_KO(s_);                               \
        BG_RD(fa0_##f_, fp_, a_rd0) BG_RD(fa1_##f_, fp_, a_rd0 + 32 * kRowBytes)                      \
        BG_RD(fa2_##f_, fp_, a_rd0 + 64 * kRowBytes) BG_RD(fa3_##f_, fp_, a_rd0 + 96 * kRowBytes)     \
        BG_RD(fb0_##f_, fp_, ln.b_rd) BG_RD(fb1_##f_, fp_, ln.b_rd + 32 * kRowBytes)                  \
    }
#define BG_MF(i_, n_, f_)                                                                              \
    acc##i_##n_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa##i_##_##f_),   \
                                                          __builtin_bit_cast(bf16x8, fb##n_##_##f_), acc##i_##n_, 0, 0, 0);
// the 8 MFMAs of one k-step; FULL tiles run all of them, partial tiles only the wave's active groups
#define BG_MFMAS(f_)                                                      \
    if (FULL || cnt > 0) { BG_MF(0, 0, f_) BG_MF(0, 1, f_) }              \
    if (FULL || cnt > 1) { BG_MF(1, 0, f_) BG_MF(1, 1, f_) }              \
    if (FULL || cnt > 2) { BG_MF(2, 0, f_) BG_MF(2, 1, f_) }              \
    if (FULL || cnt > 3) { BG_MF(3, 0, f_) BG_MF(3, 1, f_) }

// issue the 6 wave instructions that bring 32-wide K-chunk (tap, kc, itl) into ring stage `stage`:
// this wave's 32 A rows (its row group) and 64 W rows (its channel slice)
constexpr int kLoadsPerChunk = 6;
__device__ __forceinline__ void issue_chunk(const TdnnArgs& a, const Stream& st, char* smem, int stage, int tap,
                                            int kc, int itl) {
    char* sa = smem + stage * kStageBytes + st.lds_a;
    char* sb = smem + stage * kStageBytes + st.lds_b;
    const int xs = (tap * a.tap_rows * a.ldx + kc * 32) * 2;
    const int ws = itl * kRowBytes;
    glds16(st.xrsrc, sa, st.xo0, xs);
    glds16(st.xrsrc, sa + 16 * kRowBytes, st.xo1, xs);
    glds16(st.wrsrc, sb, st.wo0, ws);
    glds16(st.wrsrc, sb + 16 * kRowBytes, st.wo1, ws);
    glds16(st.wrsrc, sb + 32 * kRowBytes, st.wo2, ws);
    glds16(st.wrsrc, sb + 48 * kRowBytes, st.wo3, ws);
}

// wait until at most `chunks_in_flight` of this wave's chunk loads (6 instructions each) are pending
__device__ __forceinline__ void wait_chunks(int chunks_in_flight) {
    if (chunks_in_flight >= 1) __builtin_amdgcn_s_waitcnt(0x0F76);        // vmcnt(6)
    else __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
}

// One tile: g (1..4) row groups of 32 frames starting at group g0, 256 channels from n0.
template <bool FULL, bool POOL, bool STORE>
__device__ __forceinline__ void process_tile(const TdnnArgs& a, char* smem, Stream& st, const Lane& ln,
                                             int64_t g0, int g, int n0, int u_hint) {
    const int64_t m0 = g0 * 32;
    const int grp0 = 0;                 // every wave covers all row groups of the tile
    const int cnt = FULL ? 4 : g;       // active groups
    const int a_rd0 = ln.a_rd;

    // ---- source offsets of this wave's A rows: group `wave` of the tile (rows 32*wave + 16k + lane/4),
    // re-based into the input layout by the utterance of each row.  Every wave loads its group even
    // in a partial tile (rows past the tile are allocated; the data is not used), so each wave has
    // exactly four loads per chunk in flight and the counted waits below hold for all waves.
    st.xrsrc = make_srd(static_cast<const char*>(a.X) + m0 * (int64_t)a.ldx * 2);
    {
        const int l = threadIdx.x & 63;
        const int rb = a.ldx * 2;
        int u = u_hint;
        int64_t nxt = row_off(a.out_map, u + 1);
#define BG_XOFF(k_, dst_)                                                                           \
        {                                                                                               \
            const int lr = 32 * ln.wave + 16 * k_ + (l >> 2);                                           \
            const int64_t p = m0 + lr;                                                                  \
            while (p >= nxt && u < a.out_map.n_utts - 1) { ++u; nxt = row_off(a.out_map, u + 1); }      \
            const int lc = (l & 3) ^ ((lr >> 2) & 3);                                                   \
            dst_ = (lr + u * a.span) * rb + lc * 16;                                                    \
        }
        BG_XOFF(0, st.xo0) BG_XOFF(1, st.xo1)
#undef BG_XOFF
    }

    BG_ACC_DECL
    BG_FRAG_DECL
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f;
        acc20[e] = 0.f; acc21[e] = 0.f; acc30[e] = 0.f; acc31[e] = 0.f;
    }

    // ---- prologue: chunks 0 and 1 into ring stages 0 and 1
    int tap = 0, kc = 0, issued = 0;     // (tap, kc) of the next chunk to issue; chunks issued so far
    const int cpt32 = 2 * a.cpt;         // 32-wide chunks per tap (a.cpt counts 64-element chunks)
    const int n_chunks = a.n_taps * cpt32;
    for (; issued < 2 && issued < n_chunks; ++issued) {
        issue_chunk(a, st, smem, issued, tap, kc, issued);
        if (++kc == cpt32) { kc = 0; ++tap; }
    }
    wait_chunks(issued - 1);
    __syncthreads();
    BG_FRAGS(0, 0, smem)

    for (int it = 0; it < n_chunks; ++it) {
        const char* S = smem + (it % kStages) * kStageBytes;
        if (issued < n_chunks) {             // chunk it+2 -> the stage chunk it-1 has left
            issue_chunk(a, st, smem, issued % kStages, tap, kc, issued);
            if (++kc == cpt32) { kc = 0; ++tap; }
            ++issued;
        }
        SB();
        BG_FRAGS(1, 1, S) SB();
        BG_MFMAS(0) SB();
        // chunk it+1 has landed (chunk it+2 may still be in flight) and, past the barrier, nobody
        // reads chunk it's stage any more
        wait_chunks(issued - (it + 2));
        __syncthreads();
        if (it + 1 < n_chunks) { BG_FRAGS(0, 0, smem + ((it + 1) % kStages) * kStageBytes) }
        SB();
        BG_MFMAS(1) SB();
    }

    // ---- epilogue: bias + ReLU + folded BatchNorm (tdnn_layer.py:30-39)
    // accumulator element e of lane (r, h): row = (e&3) + 8*(e>>2) + 4*h, col = r
#define BG_EPI(i_)                                                                                     \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                   \
        acc##i_##0[e] = fmaf(fmaxf(acc##i_##0[e] + ln.bias0, 0.f), ln.scale0, ln.shift0);              \
        acc##i_##1[e] = fmaf(fmaxf(acc##i_##1[e] + ln.bias1, 0.f), ln.scale1, ln.shift1);              \
    }
    BG_EPI(0) BG_EPI(1) BG_EPI(2) BG_EPI(3)
#undef BG_EPI

    if (POOL) {
#define BG_POOL(i_)                                                                                    \
    if (i_ < cnt) {                                                                                    \
        const int64_t row_g = m0 + (int64_t)(grp0 + i_) * 32;                                          \
        pool_group(a, acc##i_##0, row_g, ln.h, ln.col0);                                               \
        pool_group(a, acc##i_##1, row_g, ln.h, ln.col0 + 32);                                          \
    }
        BG_POOL(0) BG_POOL(1) BG_POOL(2) BG_POOL(3)
#undef BG_POOL
    }

    if (STORE) {
        // transpose through LDS (free now): wave-private [128 rows][64 bf16] image, then whole
        // 128-byte rows with 16-byte stores
        char* wbuf = smem + ln.wave * (128 * kEpiRowBytes);
        const int l = threadIdx.x & 63;
        const bool odd = l & 1;
        // registers (e, e+1), e even, are rows R, R+1 of this lane's column: swap one of them with
        // the neighbour lane so each lane holds two ADJACENT columns of one row -> one packed dword
#define BG_PACK(i_, n_)                                                                                \
    _Pragma("unroll") for (int e = 0; e < 16; e += 2) {                                                \
        const float mine = odd ? acc##i_##n_[e + 1] : acc##i_##n_[e];                                  \
        const float give = odd ? acc##i_##n_[e] : acc##i_##n_[e + 1];                                  \
        const float got = __shfl_xor(give, 1);                                                         \
        const float lo = odd ? got : mine, hi = odd ? mine : got;                                      \
        const __bf16 blo = (__bf16)lo, bhi = (__bf16)hi;                                               \
        const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, blo) |                        \
                            ((unsigned)__builtin_bit_cast(unsigned short, bhi) << 16);                 \
        const int row = 32 * i_ + (e & 3) + 8 * (e >> 2) + 4 * ln.h + (odd ? 1 : 0);                   \
        const int colp = 32 * n_ + (ln.r & ~1);                                                        \
        *reinterpret_cast<unsigned*>(wbuf + row * kEpiRowBytes + colp * 2) = pk;                          \
    }
        BG_PACK(0, 0) BG_PACK(0, 1) BG_PACK(1, 0) BG_PACK(1, 1)
        BG_PACK(2, 0) BG_PACK(2, 1) BG_PACK(3, 0) BG_PACK(3, 1)
#undef BG_PACK
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): own LDS writes done (wave-private region)
        __bf16* Y = static_cast<__bf16*>(a.Y);
        const int64_t row_base = m0 + (int64_t)grp0 * 32;
        const int ccol = n0 + ln.wave * 64 + (l & 7) * 8;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int row = 8 * t + (l >> 3);
            if ((row >> 5) < cnt) {
                const float4 v = *reinterpret_cast<const float4*>(wbuf + row * kEpiRowBytes + (l & 7) * 16);
                *reinterpret_cast<float4*>(Y + (row_base + row) * a.ldy + ccol) = v;
            }
        }
    }
    // LDS is reused by the next tile's prologue (and by other waves' transposes): drain and sync
    __syncthreads();
}

template <bool POOL, bool STORE>
__global__ __launch_bounds__(kThreads, 2) void tdnn_bf16_big_kernel(const TdnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int j = lid % a.n_tiles;
    const int p = lid / a.n_tiles;
    const int64_t g_begin = a.groups_total * (int64_t)p / a.blocks_per_col;
    const int64_t g_end = a.groups_total * (int64_t)(p + 1) / a.blocks_per_col;
    const int n0 = j * kBN;

    const int tid = threadIdx.x;
    const int l = tid & 63;
    Lane ln;
    ln.wave = tid >> 6;
    ln.h = l >> 5;
    ln.r = l & 31;
    ln.sw = (ln.r >> 2) & 3;
    ln.a_rd = ln.r * kRowBytes;                                    // + first group of the wave (per tile)
    ln.b_rd = kBM * kRowBytes + (ln.wave * 64 + ln.r) * kRowBytes;
    ln.col0 = n0 + ln.wave * 64 + ln.r;
    ln.bias0 = a.bias[ln.col0];      ln.scale0 = a.scale[ln.col0];      ln.shift0 = a.shift[ln.col0];
    ln.bias1 = a.bias[ln.col0 + 32]; ln.scale1 = a.scale[ln.col0 + 32]; ln.shift1 = a.shift[ln.col0 + 32];

    Stream st;
    st.wrsrc = make_srd(static_cast<const char*>(a.W) + (int64_t)n0 * a.k_pad * 2);
    st.lds_a = (32 * ln.wave) * kRowBytes;
    st.lds_b = kBM * kRowBytes + (64 * ln.wave) * kRowBytes;
    {
        const int rbw = a.k_pad * 2;
#define BG_WOFF(k_, dst_)                                                    \
        {                                                                        \
            const int lr = 64 * ln.wave + 16 * k_ + (l >> 2);                    \
            const int lc = (l & 3) ^ ((lr >> 2) & 3);                            \
            dst_ = lr * rbw + lc * 16;                                           \
        }
        BG_WOFF(0, st.wo0) BG_WOFF(1, st.wo1) BG_WOFF(2, st.wo2) BG_WOFF(3, st.wo3)
#undef BG_WOFF
    }

    int u = utt_of_row(a.out_map, g_begin * 32);
    for (int64_t g = g_begin; g < g_end; g += 4) {
        const int cntg = (int)((g_end - g) < 4 ? (g_end - g) : 4);
        const int64_t m0 = g * 32;
        while (m0 >= row_off(a.out_map, u + 1) && u < a.out_map.n_utts - 1) ++u;
        if (cntg == 4) process_tile<true, POOL, STORE>(a, smem_b, st, ln, g, 4, n0, u);
        else process_tile<false, POOL, STORE>(a, smem_b, st, ln, g, cntg, n0, u);
    }
}

}  // namespace big

template <bool POOL, bool STORE>
static hipError_t launch_big(const TdnnArgs& a, hipStream_t s) {
    auto kern = big::tdnn_bf16_big_kernel<POOL, STORE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, big::kLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    kern<<<dim3(a.blocks_per_col * a.n_tiles), dim3(big::kThreads), big::kLdsBytes, s>>>(a);
    return hipGetLastError();
}

// a.n_tiles / a.blocks_per_col are in units of 256-channel columns here (two blocks per CU)
hipError_t launch_tdnn_bf16_big(const TdnnArgs& a, bool pool, hipStream_t s) {
    if (a.groups_total <= 0 || a.blocks_per_col <= 0 || a.n_tiles <= 0) return hipErrorInvalidValue;
    return pool ? launch_big<true, false>(a, s) : launch_big<false, true>(a, s);
}

}  // namespace xvec

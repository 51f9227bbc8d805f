// Frame-level TDNN layer for gfx950 (MI355X): exact fp32 (v_mfma_f32_32x32x2_f32) or bf16 inputs with
// fp32 accumulation (v_mfma_f32_32x32x16_bf16):
//   Y[p, n] = relu( sum_{tap, c} X[p + tap*dil, c] * W[n, tap, c] + bias[n] ) * scale[n] + shift[n]
// i.e. tdnn_layer.py:26-41 of the reference (context concat -> Linear -> ReLU -> eval
// BatchNorm1d) as ONE implicit-GEMM kernel over the flat frame axis.  No context copy
// (the reference's torch.cat, tdnn_layer.py:29) is ever materialised: a K-chunk that
// belongs to tap j is staged from rows p + j*dil of the same activation buffer.
//
// Machine mapping (CDNA4)
//   * persistent grid: 2 blocks per CU (64 KiB LDS each).  The output is cut into columns of
//     128 channels; the frames of one column are split, at 32-row granularity, into equal
//     contiguous ranges, one per block, so the last wave of work is ~1 % of the launch
//     instead of a partial round of fixed 128x128 tiles.  A block walks its range in tiles of
//     up to 4 row groups (128 frames); the blocks that cover the same frames for the different
//     channel columns get consecutive ids on one XCD, so the activation rows are shared in
//     that XCD's L2.
//   * 256 threads = 4 wave64; wave w owns channels [32w, 32w+32) of the tile for all its
//     frames: G accumulators of v_mfma_f32_32x32x2_f32 (<= 64 VGPRs).
//   * K is consumed in 32-wide chunks: global -> registers -> LDS (two LDS buffers, two
//     register sets, so a chunk's global loads are issued ~1.5 chunks before they are
//     stored), one block barrier per chunk.  LDS rows are 128 B with a 16-B-chunk XOR
//     swizzle: the ds_read_b128 fragment reads are bank-conflict free.
//   * the chunks of a tile are walked with the taps innermost (chunk kc of tap 0, 1, 2, then kc+1;
//     pack.hip stores the weights' K axis in that order): the dilated taps re-read a 128-byte slab
//     of activation rows while it is still in L2.  bf16: the weights are packed fragment-major and
//     go from global memory straight into the MFMA B operand, only activations pass through LDS.
//   * bf16x3 (template flag X3 of the bf16 instantiations): X and Y are two bf16 planes (hi, lo) of
//     fp32 values; an LDS buffer holds the hi and the lo tile of a chunk, the weight stream its W_hi and
//     W_lo fragments, and every k-step issues x_hi*W_hi + x_hi*W_lo + x_lo*W_hi (XV_CHUNK3; fp32-level
//     results: 1.5e-6 from the fp64 oracle).
//   * each ds_read_b128 feeds four MFMAs: lane half h owns k = 8q+4h..8q+4h+3 of every
//     8-wide k group, for A and B alike, so the products pair up (the k order inside a chunk
//     is permuted, which fp32 addition tolerates to rounding).
//   * every memory instruction of a chunk is slotted behind one MFMA (64 pipe cycles each):
//     in-kernel stamps showed bursts of 8 ds_write_b128 / 8 global loads idling the matrix
//     pipe for ~450 cycles each; spread out they are free.
//
// Epilogue: bias + ReLU + folded BatchNorm in registers; optional fused statistics pooling
// (main.py:59-63): per 32-row group and per utterance overlapping it, a pivot K (the group's frame 0)
// and the sums of (r - K), (r - K)^2 over the valid frames, r = relu(z + bias), go to a small partials
// buffer that pool_finalize merges in fp64 (tdnn_common.h, pool_group_impl), so the [frames,1500]
// activation of layer 5 never goes to HBM.
#include <cstdlib>

#include "tdnn_common.h"

namespace xvec {

constexpr int kBM = 128, kBN = 128;
constexpr int kStageFloats = (kBM + kBN) * kBK;   // one LDS buffer: A tile then B tile
constexpr int kConstFloats = 3 * kBN;             // bias | scale | shift of the block's 128 channels (pooling: bias | bias - K | -K)
constexpr int kLdsBytes = (2 * kStageFloats + kConstFloats) * 4;

#ifdef XVEC_DIAG
// Diagnostic build only (make DIAG=1): s_memtime stamps of wave 0 of every block.
__device__ unsigned long long g_diag[8 * 8192];
#endif

// Per-thread state of one tile walk.  Global reads go through raw buffer loads: a block-uniform
// descriptor per operand (rebased at the tile origin), a uniform scalar byte offset (tap row
// shift, row group, chunk column) and ONE 32-bit per-thread byte offset per operand, so no
// per-load 64-bit address is ever computed or kept in VGPRs.
struct Ctx {
    __amdgpu_buffer_rsrc_t xrsrc;   // X + m0*ldx  (tile the load stream is in)
    __amdgpu_buffer_rsrc_t wrsrc;   // W + n0*k_pad
    int x_base;          // r0*ldx*ES + c*16 bytes    (per thread)
    int ur0, ur1, ur2, ur3;   // u(row r0+32j of the tile)*span: rows to add to re-base into the input layout
    int xo0, xo1, xo2, xo3;   // x_base + ur_j*ldx*ES
    int w_toff;          // r0*k_pad*ES + c*16 bytes
    __amdgpu_buffer_rsrc_t wfrsrc;   // bf16: fragment-major weights of this wave's 32-channel column tile
    int wf_voff;         // lane*16
    int r0;
    int u_tile;          // utterance holding row m0 (block-uniform), and the first row of the next one
    int64_t off_next;
    int64_t m0;          // first flat row of the tile the load stream is in
    int64_t g_s, g_end;  // that tile's first row group; end of this block's row range
    PoolCur pool;        // compute side: pooling cursor (POOL variants)
    bool pivot_set;      // POOL: the block's pooling pivots are in LDS (false until its first tile's epilogue)
    int tap, kc, itl;    // next chunk to fetch: (tap, kc) and its linear index within the tile
    int es;              // bytes per input element (4: fp32, 2: bf16)
};

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    // (whole-vector bit cast: __builtin_bit_cast on single elements of the result made hipcc
    // 7.2 narrow the load to one dword and splat it)
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    const f32x4 f = __builtin_bit_cast(f32x4, v);
    return make_float4(f.x, f.y, f.z, f.w);
}

// eight fp32 values -> eight bf16 in the 16 bytes of a staging register (round to nearest even, as pack_rows)
__device__ __forceinline__ float4 cvt8_bf16(const float4& lo, const float4& hi) {
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    typedef float f32x8v __attribute__((ext_vector_type(8)));
    const f32x8v f8 = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_convertvector(f8, bf16x8v));
    return make_float4(b4.x, b4.y, b4.z, b4.w);
}

// per-thread input offsets of the four staging rows (r0 + 32j) of the tile the stream is in: the
// compact output row p of utterance u reads input rows p + u*span (+ tap shift).  The utterance of
// the tile's first row is tracked incrementally (cx.u_tile / cx.off_next: tiles only move
// forward); the few utterance boundaries inside the tile are walked with block-uniform values
// (scalar loads of the offsets for ragged batches) and each lane just counts how many of them its
// rows have passed -- no division, and no vector-memory load whose wait would drain the
// staging loads in flight.
// The fixed-length and the ragged case are two separate code paths on purpose: sharing one loop made
// hipcc put the ragged path's s_waitcnt vmcnt(0) (for the offsets load) on the fixed path too,
// draining the 16 staging loads in flight at every tile change.
template <bool RAGGED>
__device__ __forceinline__ void set_tile_rows_impl(const TdnnArgs& a, Ctx& cx) {
    const int n_last = a.out_map.n_utts - 1;
    const int64_t t_out = a.out_map.fixed_T - a.out_map.cum;          // fixed-length: rows per utterance
    auto next_off = [&](int u) -> int64_t {                            // first row of utterance u+1
        if (RAGGED) return a.out_map.offsets[u + 1] - (int64_t)(u + 1) * a.out_map.cum;
        return (int64_t)(u + 1) * t_out;
    };
    while (cx.m0 >= cx.off_next && cx.u_tile < n_last) {
        cx.u_tile = __builtin_amdgcn_readfirstlane(cx.u_tile + 1);
        cx.off_next = next_off(cx.u_tile);
    }
    const int64_t p = cx.m0 + cx.r0;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int u = cx.u_tile;
    int64_t nxt = cx.off_next;
    while (nxt < cx.m0 + 128 && u < n_last) {       // block-uniform walk over the boundaries in the tile
        c0 += (p >= nxt) ? 1 : 0;
        c1 += (p + 32 >= nxt) ? 1 : 0;
        c2 += (p + 64 >= nxt) ? 1 : 0;
        c3 += (p + 96 >= nxt) ? 1 : 0;
        u = __builtin_amdgcn_readfirstlane(u + 1);
        nxt = next_off(u);
    }
    const int rb = a.ldx * cx.es;
    cx.ur0 = (cx.u_tile + c0) * a.span;
    cx.ur1 = (cx.u_tile + c1) * a.span;
    cx.ur2 = (cx.u_tile + c2) * a.span;
    cx.ur3 = (cx.u_tile + c3) * a.span;
    cx.xo0 = cx.x_base + cx.ur0 * rb;
    cx.xo1 = cx.x_base + cx.ur1 * rb;
    cx.xo2 = cx.x_base + cx.ur2 * rb;
    cx.xo3 = cx.x_base + cx.ur3 * rb;
}

__device__ __forceinline__ void set_tile_rows(const TdnnArgs& a, Ctx& cx) {
    if (a.span == 0) {
        cx.ur0 = cx.ur1 = cx.ur2 = cx.ur3 = 0;
        cx.xo0 = cx.xo1 = cx.xo2 = cx.xo3 = cx.x_base;
        return;
    }
    if (a.out_map.offsets == nullptr) set_tile_rows_impl<false>(a, cx);
    else set_tile_rows_impl<true>(a, cx);
}

// Step the load stream to the next K-chunk.  The stream is continuous over the block's tiles:
// after the last chunk of a tile it moves to chunk 0 of the next tile (same channel column, next
// <=4 row groups), so a tile's first chunks are already in flight / in LDS when its MFMAs start
// and only the first tile of a block pays a prologue.  Past the block's last chunk it stays put
// (the look-ahead of the final chunks re-reads that chunk; the data is never used).
// Activation descriptor of the tile at row cx.m0.  GUARD (first layer): X is the caller's tensor,
// not a padded workspace buffer, so the descriptor ends with it and rows past the end read as 0.
template <int GUARD>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x_rsrc(const TdnnArgs& a, const Ctx& cx) {
    // GUARD == 2: the caller's rows are fp32 and are rounded to bf16 on their way into the staging registers
    // (the separate pack_rows pass of the bf16 path: 7.4 MB read + 3.7 MB written + a launch per batch)
    const int src_es = GUARD == 2 ? 4 : cx.es;
    const int64_t off = cx.m0 * (int64_t)a.ldx * src_es;
    if (GUARD) return make_rsrc_bounded(a.X, off, a.x_bytes ? a.x_bytes : a.x_rows * (int64_t)a.ldx * src_es);
    return make_rsrc(static_cast<const char*>(a.X) + off);
}

template <int GUARD, bool X3>
__device__ __forceinline__ void advance(const TdnnArgs& a, Ctx& cx, int n_chunks) {
    if (cx.itl + 1 < n_chunks) {
        // taps innermost: consecutive chunks re-read the same 128-byte slab of activation rows,
        // shifted by the dilation, while it is still in L2 (the packed weights follow this order)
        ++cx.itl;
        if (++cx.tap == a.n_taps) {
            cx.tap = 0;
            ++cx.kc;
        }
    } else {
        const int64_t g_rem = cx.g_end - cx.g_s;
        const int64_t g_next = cx.g_s + (g_rem < 4 ? g_rem : 4);
        if (g_next < cx.g_end) {
            cx.g_s = g_next;
            cx.m0 = g_next * 32;
            cx.xrsrc = x_rsrc<GUARD>(a, cx);
            set_tile_rows(a, cx);
            cx.itl = 0;
            cx.kc = 0;
            cx.tap = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The main loop is written with token-pasting macros over NAMED registers (sA<i>_<set>,
// fa<i>_<fset>, acc<i>): register arrays, even with compile-time indices through inlined
// lambdas, were left in scratch memory by hipcc (ROCm 7.2) once scheduling barriers were present.
// G (row groups of the tile, 1..4) is a template parameter; ops of absent groups vanish.
// ---------------------------------------------------------------------------------------------
#define XV_KO(q_) ((((2 * (q_)) + h) ^ sw) << 2)
// fragment reads (LDS -> VGPR) of k-group q_ into fragment set f_ from buffer base S_
#define XV_FRG_A(i_, q_, f_, S_) \
    if constexpr (G > i_) { rg.fa##i_##_##f_ = *reinterpret_cast<const float4*>((S_) + a_rd + i_ * 32 * kBK + XV_KO(q_)); }
// fp32: B fragment from the LDS image.  bf16: B fragments never touch LDS (see XV_GLB below).
#define XV_FRG_B(q_, f_, S_) \
    if constexpr (!INBF) { rg.fb_##f_ = *reinterpret_cast<const float4*>((S_) + b_rd + XV_KO(q_)); }
// LDS stores of staging set n_ into LDS buffer n_
#define XV_LST_A(i_, n_) \
    if constexpr (G > i_) { *reinterpret_cast<float4*>(smem + n_ * kStageFloats + st_off + i_ * 32 * kBK) = rg.sA##i_##_##n_; }
// the second half of an LDS buffer: fp32 -> the weight tile; bf16x3 -> the lo plane of the
// activation tile (same rows, x_plane_bytes further on in global memory); plain bf16 -> unused
#define XV_LST_B(j_, n_)                                                                                  \
    if constexpr (X3) {                                                                                   \
        if constexpr (G > j_) *reinterpret_cast<float4*>(smem + n_ * kStageFloats + st_off + kBM * kBK + j_ * 32 * kBK) = rg.sB##j_##_##n_; \
    } else if constexpr (!INBF) {                                                                         \
        *reinterpret_cast<float4*>(smem + n_ * kStageFloats + st_off + kBM * kBK + j_ * 32 * kBK) = rg.sB##j_##_##n_; \
    }
// bf16x3 keeps ONE staging set (its chunks are three times as long, so one chunk of lead hides the
// loads, and the registers are needed for the lo-plane fragments): set 0 -> LDS buffer b_
#define XV_LST3(i_, b_)                                                                                   \
    if constexpr (G > i_) {                                                                               \
        *reinterpret_cast<float4*>(smem + b_ * kStageFloats + st_off + i_ * 32 * kBK) = rg.sA##i_##_0;      \
        *reinterpret_cast<float4*>(smem + b_ * kStageFloats + st_off + kBM * kBK + i_ * 32 * kBK) = rg.sB##i_##_0; \
    }
// global loads of the chunk cx points at into staging set n_ (P_OFF: byte offset of the plane)
#define XV_GLD_X(dst_, i_, P_OFF)                                                                         \
    {                                                                                                     \
        const int row_shift = cx.tap * a.tap_rows;                                                        \
        const int soff = ((row_shift + 32 * i_) * a.ldx + cx.kc * BKE) * ES + (P_OFF);                    \
        if (GUARD) {                                                                                      \
            /* K past the layer's width (the folded taps of the next frame): an offset the descriptor's   \
               range check rejects, so the piece reads as zeros; rows past the tensor: same check */      \
            const bool ok_ = cx.kc * BKE + c * (16 / ES) < a.kpt;                                         \
            if constexpr (GUARD == 2) { /* fp32 source: 8 floats -> 8 bf16 */                             \
                const int voff = ok_ ? 2 * cx.xo##i_ : 0x7ffffff0;                                        \
                const float4 lo_ = buf_load16(cx.xrsrc, voff, 2 * soff);                                  \
                const float4 hi_ = buf_load16(cx.xrsrc, voff, 2 * soff + 16);                             \
                dst_ = cvt8_bf16(lo_, hi_);                                                               \
            } else {                                                                                      \
                const int voff = ok_ ? cx.xo##i_ : 0x7ffffff0;                                            \
                dst_ = buf_load16(cx.xrsrc, voff, soff);                                                  \
            }                                                                                             \
        } else {                                                                                          \
            dst_ = buf_load16(cx.xrsrc, cx.xo##i_, soff);                                                 \
        }                                                                                                 \
    }
#define XV_GLD_A(i_, n_) \
    if constexpr (G > i_) XV_GLD_X(rg.sA##i_##_##n_, i_, 0)
#define XV_GLD_B(j_, n_)                                                                                  \
    if constexpr (X3) {                                                                                   \
        if constexpr (G > j_) XV_GLD_X(rg.sB##j_##_##n_, j_, a.x_plane_bytes)                             \
    } else if constexpr (!INBF) {                                                                         \
        rg.sB##j_##_##n_ = buf_load16(cx.wrsrc, cx.w_toff, (32 * j_ * a.k_pad + cx.itl * BKE) * ES);         \
    }
// bf16: the weights are packed fragment-major at load time (pack.hip): for a 32-channel column
// tile and a 16-wide k-step, the 64 lanes' 16-byte MFMA B operands are one contiguous KiB.  Each
// wave reads its own B fragments straight into registers, one coalesced buffer load per k-step;
// LDS carries only the activations.  Fragment q (k-step q of a 64-wide chunk) of chunk c_ -> set s_.
#define XV_GLB(q_, s_, c_)                                                                                \
    if constexpr (INBF && !X3) {                                                                          \
        int cw_ = (c_);                                                                                   \
        if (cw_ >= n_chunks) cw_ -= n_chunks;                                                             \
        rg.gb##q_##_##s_ = buf_load16(cx.wfrsrc, cx.wf_voff, (4 * cw_ + q_) * 1024);                      \
    }
// bf16x3: per chunk the stream holds the four W_hi k-step blocks, then the four W_lo blocks; both
// fragments of k-step q_ of chunk c_ go to ONE register pair (gb<q>_0 = hi, gb<q>_1 = lo), reloaded
// as soon as the k-step's last MFMA has issued (three quarters of a chunk ahead of their use)
#define XV_GLB3(q_, c_)                                                                                   \
    if constexpr (X3) {                                                                                   \
        int cw_ = (c_);                                                                                   \
        if (cw_ >= n_chunks) cw_ -= n_chunks;                                                             \
        rg.gb##q_##_0 = buf_load16(cx.wfrsrc, cx.wf_voff, (8 * cw_ + q_) * 1024);                         \
        rg.gb##q_##_1 = buf_load16(cx.wfrsrc, cx.wf_voff, (8 * cw_ + 4 + q_) * 1024);                     \
    }
// bf16x3: lo-plane fragment of k-step q_ (second half of the LDS buffer) -> fragment set 1
#define XV_FRG_L(i_, q_, S_) \
    if constexpr (G > i_) { rg.fa##i_##_1 = *reinterpret_cast<const float4*>((S_) + kBM * kBK + a_rd + i_ * 32 * kBK + XV_KO(q_)); }
#define XV_GLD_ALL(n_) XV_GLD_A(0, n_) XV_GLD_A(1, n_) XV_GLD_A(2, n_) XV_GLD_A(3, n_) \
                       XV_GLD_B(0, n_) XV_GLD_B(1, n_) XV_GLD_B(2, n_) XV_GLD_B(3, n_)
#define XV_LST_ALL(n_) XV_LST_A(0, n_) XV_LST_A(1, n_) XV_LST_A(2, n_) XV_LST_A(3, n_) \
                       XV_LST_B(0, n_) XV_LST_B(1, n_) XV_LST_B(2, n_) XV_LST_B(3, n_)
// fp32: one MFMA (row group i_, k component c_, fragment set f_) and the statement slotted behind it
#define XV_MF(i_, c_, f_, slot_)                                                                          \
    if constexpr (G > i_) {                                                                               \
        acc##i_ = __builtin_amdgcn_mfma_f32_32x32x2f32(rg.fa##i_##_##f_.c_, rg.fb_##f_.c_, acc##i_, 0, 0, 0);   \
    }                                                                                                     \
    SB();                                                                                                 \
    slot_                                                                                                 \
    SB();
// bf16: one MFMA per row group consumes the whole 16-byte fragment (k-step of 16)
// (SWAP: weights as the first operand -- the accumulator's registers are then channels and its lanes frames,
// the layout store_acc turns into 16-byte stores)
#define XV_MFB(i_, f_, q_, P_)                                                                            \
    if constexpr (G > i_) {                                                                               \
        if constexpr (SWAP)                                                                               \
            acc##i_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rg.gb##q_##_##P_), \
                                                              __builtin_bit_cast(bf16x8, rg.fa##i_##_##f_), acc##i_, 0, 0, 0); \
        else                                                                                              \
            acc##i_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rg.fa##i_##_##f_), \
                                                              __builtin_bit_cast(bf16x8, rg.gb##q_##_##P_), acc##i_, 0, 0, 0); \
    }                                                                                                     \
    SB();
// one k-group with 16 slots.  fp32: 4 k components x 4 row groups = 16 MFMAs, one slot behind each;
// bf16: 4 MFMAs (k-step 16), four slots behind each
#define XV_KG(q_, P_, f_, s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)           \
    if constexpr (INBF) {                                                                                 \
        XV_MFB(0, f_, q_, P_) s0 s1 s2 s3 SB(); XV_MFB(1, f_, q_, P_) s4 s5 s6 s7 SB();                   \
        XV_MFB(2, f_, q_, P_) s8 s9 s10 s11 SB(); XV_MFB(3, f_, q_, P_) s12 s13 s14 s15 SB();             \
    } else {                                                                                              \
        XV_MF(0, x, f_, s0) XV_MF(1, x, f_, s1) XV_MF(2, x, f_, s2) XV_MF(3, x, f_, s3)                   \
        XV_MF(0, y, f_, s4) XV_MF(1, y, f_, s5) XV_MF(2, y, f_, s6) XV_MF(3, y, f_, s7)                   \
        XV_MF(0, z, f_, s8) XV_MF(1, z, f_, s9) XV_MF(2, z, f_, s10) XV_MF(3, z, f_, s11)                 \
        XV_MF(0, w, f_, s12) XV_MF(1, w, f_, s13) XV_MF(2, w, f_, s14) XV_MF(3, w, f_, s15)               \
    }
// bf16x3: four MFMAs (row groups) of fragment set f_ against B register b_, one slot behind each
#define XV_M3(i_, f_, b_)                                                                                 \
    if constexpr (G > i_) {                                                                               \
        acc##i_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rg.fa##i_##_##f_),   \
                                                          __builtin_bit_cast(bf16x8, rg.b_), acc##i_, 0, 0, 0); \
    }                                                                                                     \
    SB();
#define XV_G3(f_, b_, s0, s1, s2, s3) \
    XV_M3(0, f_, b_) s0 SB(); XV_M3(1, f_, b_) s1 SB(); XV_M3(2, f_, b_) s2 SB(); XV_M3(3, f_, b_) s3 SB();
#define XV_NOP ;
// One K-chunk held in LDS buffer P_; N_ = the other buffer = the staging set holding chunk it+1.
// Branch-free: the last chunk of a tile also stores/loads/reads ahead (clamped to the last
// chunk, results unused) -- n_chunks is even, so the two-chunk loop body needs no tail variants.
// (parenthesised: the template argument list must not be split by the slot macros' commas)
#define XV_ADVANCE (advance<GUARD, X3>)(a, cx, n_chunks);
#define XV_CHUNK(P_, N_, IT_)                                                                             \
    {                                                                                                     \
        const float* S = smem + P_ * kStageFloats;                                                        \
        const float* Sn = smem + N_ * kStageFloats;                                                       \
        XV_KG(0, P_, 0, XV_FRG_A(0, 1, 1, S), XV_FRG_A(1, 1, 1, S), XV_FRG_A(2, 1, 1, S),                 \
              XV_FRG_A(3, 1, 1, S), XV_FRG_B(1, 1, S),                                                    \
              XV_LST_A(0, N_), XV_LST_A(1, N_), XV_LST_A(2, N_), XV_LST_A(3, N_),                         \
              XV_LST_B(0, N_), XV_LST_B(1, N_), XV_LST_B(2, N_), XV_LST_B(3, N_),                         \
              XV_GLB(3, N_, (IT_) + 1), XV_NOP, XV_ADVANCE)                                \
        XV_KG(1, P_, 1, XV_FRG_A(0, 2, 0, S), XV_FRG_A(1, 2, 0, S), XV_FRG_A(2, 2, 0, S),                 \
              XV_FRG_A(3, 2, 0, S), XV_FRG_B(2, 0, S),                                                    \
              XV_GLD_A(0, N_), XV_GLD_A(1, N_), XV_GLD_A(2, N_), XV_GLD_A(3, N_),                         \
              XV_GLD_B(0, N_), XV_GLD_B(1, N_), XV_GLD_B(2, N_), XV_GLD_B(3, N_),                         \
              XV_GLB(0, P_, (IT_) + 2), XV_NOP, XV_NOP)                                                   \
        XV_KG(2, P_, 0, XV_FRG_A(0, 3, 1, S), XV_FRG_A(1, 3, 1, S), XV_FRG_A(2, 3, 1, S),                 \
              XV_FRG_A(3, 3, 1, S), XV_FRG_B(3, 1, S), XV_GLB(1, P_, (IT_) + 2), XV_NOP, XV_NOP, XV_NOP,  \
              XV_NOP, XV_NOP, XV_NOP, XV_NOP, XV_NOP, XV_NOP, XV_NOP)                                     \
        __syncthreads(); /* chunk it+1 complete in LDS; chunk it's buffer is free */                     \
        XV_KG(3, P_, 1, XV_FRG_A(0, 0, 0, Sn), XV_FRG_A(1, 0, 0, Sn), XV_FRG_A(2, 0, 0, Sn),              \
              XV_FRG_A(3, 0, 0, Sn), XV_FRG_B(0, 0, Sn), XV_GLB(2, P_, (IT_) + 2), XV_NOP, XV_NOP,        \
              XV_NOP, XV_NOP, XV_NOP, XV_NOP, XV_NOP, XV_NOP, XV_NOP, XV_NOP)                             \
    }

// bf16x3: one K-chunk = [hi tile | lo tile] in LDS buffer P_.  Per k-step q: x_hi*W_hi (hi frags in
// set 0, read during the previous k-step), x_hi*W_lo, x_lo*W_hi (lo frags into set 1 during the first
// group) -- 12 MFMAs per k-step, 48 per chunk and barrier; every fragment read and every weight
// fragment load feeds its MFMAs once, the staging of chunk it+1 / it+2 rides in the middle groups.
#define XV_CHUNK3(P_, N_, IT_)                                                                            \
    {                                                                                                     \
        const float* S = smem + P_ * kStageFloats;                                                        \
        const float* Sn = smem + N_ * kStageFloats;                                                       \
        XV_G3(0, gb0_0, XV_FRG_L(0, 0, S), XV_FRG_L(1, 0, S), XV_FRG_L(2, 0, S), XV_FRG_L(3, 0, S))       \
        XV_G3(0, gb0_1, XV_LST3(0, N_), XV_LST3(1, N_), XV_LST3(2, N_), XV_LST3(3, N_))                   \
        XV_G3(1, gb0_0, XV_FRG_A(0, 1, 0, S), XV_FRG_A(1, 1, 0, S), XV_FRG_A(2, 1, 0, S),                 \
              XV_FRG_A(3, 1, 0, S))                                                                       \
        XV_ADVANCE                                                                                        \
        XV_G3(0, gb1_0, XV_FRG_L(0, 1, S), XV_FRG_L(1, 1, S), XV_FRG_L(2, 1, S), XV_FRG_L(3, 1, S))       \
        XV_G3(0, gb1_1, XV_GLB3(0, (IT_) + 1), XV_GLD_A(0, 0) XV_GLD_A(1, 0),                             \
              XV_GLD_A(2, 0) XV_GLD_A(3, 0), XV_GLD_B(0, 0) XV_GLD_B(1, 0))                               \
        XV_G3(1, gb1_0, XV_FRG_A(0, 2, 0, S), XV_FRG_A(1, 2, 0, S), XV_FRG_A(2, 2, 0, S),                 \
              XV_FRG_A(3, 2, 0, S))                                                                       \
        XV_G3(0, gb2_0, XV_FRG_L(0, 2, S), XV_FRG_L(1, 2, S), XV_FRG_L(2, 2, S), XV_FRG_L(3, 2, S))       \
        XV_G3(0, gb2_1, XV_GLB3(1, (IT_) + 1), XV_GLD_B(2, 0) XV_GLD_B(3, 0), XV_NOP, XV_NOP)             \
        XV_G3(1, gb2_0, XV_FRG_A(0, 3, 0, S), XV_FRG_A(1, 3, 0, S), XV_FRG_A(2, 3, 0, S),                 \
              XV_FRG_A(3, 3, 0, S))                                                                       \
        XV_G3(0, gb3_0, XV_FRG_L(0, 3, S), XV_FRG_L(1, 3, S), XV_FRG_L(2, 3, S), XV_FRG_L(3, 3, S))       \
        XV_G3(0, gb3_1, XV_GLB3(2, (IT_) + 1), XV_NOP, XV_NOP, XV_NOP)                                    \
        __syncthreads(); /* chunk it+1 complete in LDS; chunk it's buffer is free */                     \
        XV_G3(1, gb3_0, XV_FRG_A(0, 0, 0, Sn), XV_FRG_A(1, 0, 0, Sn), XV_FRG_A(2, 0, 0, Sn),              \
              XV_FRG_A(3, 0, 0, Sn))                                                                      \
        XV_GLB3(3, (IT_) + 1)                                                                             \
    }

// Pipeline registers that live across tiles: two staging sets (_0/_1: A row groups 0..3 and W
// row blocks 0..3 of a chunk in flight) and two fragment sets.  A struct of named members, not
// arrays (see above).
struct Regs {
    float4 sA0_0, sA1_0, sA2_0, sA3_0, sB0_0, sB1_0, sB2_0, sB3_0;
    float4 sA0_1, sA1_1, sA2_1, sA3_1, sB0_1, sB1_1, sB2_1, sB3_1;
    float4 fa0_0, fa1_0, fa2_0, fa3_0, fb_0, fa0_1, fa1_1, fa2_1, fa3_1, fb_1;
    float4 gb0_0, gb1_0, gb2_0, gb3_0, gb0_1, gb1_1, gb2_1, gb3_1;   // bf16: B fragments of two chunks, from global
};

struct Lane {
    int h, sw, a_rd, b_rd, st_off, r0, c, col;
};

// Once per block: chunk 0 of the first tile -> LDS buffer 0, its first fragments -> set 0,
// chunks 1 and 2 in flight in the two staging sets (bf16x3: chunk 1 in its single set).
template <int GUARD, bool INBF, bool X3>
__device__ __forceinline__ void block_prologue(const TdnnArgs& a, float* smem, Ctx& cx, Regs& rg, const Lane& ln,
                                               int n_chunks) {
    constexpr int G = 4;   // fetch all four row groups: rows past a short first tile are allocated
    constexpr int ES = INBF ? 2 : 4, BKE = 128 / ES;
    const int h = ln.h, sw = ln.sw, a_rd = ln.a_rd, b_rd = ln.b_rd, st_off = ln.st_off, c = ln.c;
    if constexpr (X3) {      // one staging set: chunk 0 -> LDS buffer 0, chunk 1 in flight in the set
        XV_GLD_ALL(0)
        SB();
        XV_LST3(0, 0) XV_LST3(1, 0) XV_LST3(2, 0) XV_LST3(3, 0)
        SB();
        advance<GUARD, X3>(a, cx, n_chunks);
        XV_GLD_ALL(0)
    } else {
        XV_GLD_ALL(0)
        advance<GUARD, X3>(a, cx, n_chunks);
        XV_GLD_ALL(1)
        SB();
        XV_LST_ALL(0)
        SB();
        advance<GUARD, X3>(a, cx, n_chunks);
        XV_GLD_ALL(0)
    }
    __syncthreads();
    XV_FRG_A(0, 0, 0, smem) XV_FRG_A(1, 0, 0, smem) XV_FRG_A(2, 0, 0, smem) XV_FRG_A(3, 0, 0, smem)
    XV_FRG_B(0, 0, smem)
    XV_GLB(0, 0, 0) XV_GLB(1, 0, 0) XV_GLB(2, 0, 0) XV_GLB(3, 0, 0)
    XV_GLB(0, 1, 1) XV_GLB(1, 1, 1) XV_GLB(2, 1, 1) XV_GLB(3, 1, 1)
    XV_GLB3(0, 0) XV_GLB3(1, 0) XV_GLB3(2, 0) XV_GLB3(3, 0)
    SB();
}

// One tile of G row groups (32 frames each) x 128 channels, starting at row group g0.  On entry
// the pipeline is primed for this tile (block_prologue or the previous tile's last chunks).
template <int G, int GUARD, bool POOL, bool STORE, bool INBF, bool OUTBF, bool X3>
__device__ __forceinline__ void process_tile(const TdnnArgs& a, float* smem, Ctx& cx, Regs& rg, const Lane& ln,
                                             int64_t g0, int n0, int n_chunks) {
    constexpr int ES = INBF ? 2 : 4, BKE = 128 / ES;
    const int h = ln.h, sw = ln.sw, a_rd = ln.a_rd, b_rd = ln.b_rd, st_off = ln.st_off, c = ln.c;
    // bf16 in, bf16 out, one plane, stored: the transposed product (channels in the accumulator's registers), so
    // that a lane ends up with 8 consecutive channels of one frame = one 16-byte store.  With frames in the
    // registers a lane holds ONE channel and every value is its own 2-byte store: 64 store instructions per
    // wave and tile -- layer 1, whose K loop is two chunks long, spent most of its 30 us issuing them.
    constexpr bool SWAP = INBF && OUTBF && !X3 && STORE && !POOL;
    f32x16 acc0, acc1, acc2, acc3;
    // this wave's 32 channels, the lane half's 4 of every 8 (SWAP): bias | scale | shift tables in LDS
    const float* cstw = smem + 2 * kStageFloats + (ln.col - n0 - (int)(threadIdx.x & 31)) + 4 * ln.h;
    if constexpr (SWAP) {      // accumulators start at the bias of their register's channel
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 b4 = *reinterpret_cast<const float4*>(cstw + 8 * gq);
            acc0[4 * gq] = b4.x; acc0[4 * gq + 1] = b4.y; acc0[4 * gq + 2] = b4.z; acc0[4 * gq + 3] = b4.w;
        }
        acc1 = acc0; acc2 = acc0; acc3 = acc0;
    } else if constexpr (POOL) {
        // accumulators start at bias - K, K = the block's pooling pivot of this lane's channel (at 0 in the block's
        // first tile, whose epilogue picks K and adds bias - K: a large bias must not sit in the accumulator while
        // the K loop adds small terms to it -- every MFMA would round at ulp(bias)): tdnn_common.h, pool_group_impl
        const float b0 = smem[2 * kStageFloats + kBN + (ln.col - n0)];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc0[e] = b0;
        acc1 = acc0; acc2 = acc0; acc3 = acc0;
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; acc2[e] = 0.f; acc3[e] = 0.f; }
    }

#ifdef XVEC_DIAG
    SB();
    const unsigned long long dt0 = __builtin_amdgcn_s_memtime();
    SB();
#endif
    // ---- K chunks, two per trip (LDS buffer 0 then 1); n_chunks is even
    for (int it = 0; it < n_chunks; it += 2) {
        if constexpr (X3) {
            XV_CHUNK3(0, 1, it)
            XV_CHUNK3(1, 0, it + 1)
        } else {
            XV_CHUNK(0, 1, it)
            XV_CHUNK(1, 0, it + 1)
        }
    }
#ifdef XVEC_DIAG
    SB();
    const unsigned long long dt1 = __builtin_amdgcn_s_memtime();
    SB();
#endif

    const int64_t m0 = g0 * 32;
    // ---- epilogue: bias + ReLU + folded BatchNorm (tdnn_layer.py:30-39)
    // accumulator element e of lane (r, h): row = (e&3) + 8*(e>>2) + 4*h, col = r
    const int col = ln.col;
    // epilogue constants of this lane's channel: three LDS reads per tile instead of three registers held
    // across the K loop (the pooling and first-layer variants were 3-6 registers over the 256 budget)
    float* cst = smem + 2 * kStageFloats + (col - n0);
    const float bi = cst[0], sc = cst[kBN], sh = cst[2 * kBN];
    // pooling variants: the three slots are bias | bias - K | -K (this wave's own channels: no barrier)
    float negk = sh;
    if constexpr (POOL) {
        if (!cx.pivot_set) {       // block-uniform: the block's first tile, whose accumulators started at 0
            const float piv = lower_half(fmaxf(acc0[0] + bi, 0.f));     // r of the block's first frame
            const float bmk = bi - piv;
            negk = bmk - bi;                                      // minus the pivot the tiles really carry
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc0[e] += bmk;
                if constexpr (G > 1) acc1[e] += bmk;
                if constexpr (G > 2) acc2[e] += bmk;
                if constexpr (G > 3) acc3[e] += bmk;
            }
            cst[kBN] = bmk;
            cst[2 * kBN] = negk;
            cx.pivot_set = true;
        }
    }
    // Two phases per row group: (1) all 16 values finished IN PLACE in the accumulator registers,
    // (2) 16 stores issued back to back from those 16 distinct registers, addressed by a per-tile
    // buffer descriptor + one per-lane offset + a scalar row offset.  (Computing each value into a
    // shared temporary right before its store made every store wait for the previous one to have
    // read that register, and chained 64-bit address adds: ~23k cycles per tile in layer 2.)
    const int esz = OUTBF ? 2 : 4;
    __amdgpu_buffer_rsrc_t yrsrc;
    int y_voff = 0;
    __amdgpu_buffer_rsrc_t yrsrc_lo;     // bf16x3: descriptor of the lo plane (same offsets as the hi plane)
    if (STORE) {
        yrsrc = make_rsrc(static_cast<char*>(a.Y) + (m0 * (int64_t)a.ldy + n0) * esz);
        if constexpr (X3 && OUTBF)
            yrsrc_lo = make_rsrc(static_cast<char*>(a.Y) + a.y_plane_bytes + (m0 * (int64_t)a.ldy + n0) * esz);
        y_voff = (4 * h * a.ldy + (col - n0)) * esz;
    }
#define XV_EPI(i_)                                                                                        \
    if constexpr (G > i_) {                                                                               \
        /* pooling variant: the accumulator holds z + bias - K; ReLU and sums in pool_group, BatchNorm in  \
           pool_finalize */                                                                               \
        if constexpr (!POOL) {                                                                            \
            _Pragma("unroll") for (int e = 0; e < 16; ++e)                                                \
                acc##i_[e] = fmaf(fmaxf(acc##i_[e] + bi, 0.f), sc, sh);                                   \
            asm volatile("" : "+v"(acc##i_));                                                             \
        }                                                                                                 \
        if (STORE) {                                                                                      \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                              \
                const int soff = (i_ * 32 + (e & 3) + 8 * (e >> 2)) * a.ldy * esz;                        \
                if constexpr (OUTBF) {                                                                    \
                    const __bf16 hv = (__bf16)acc##i_[e];                                                 \
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), yrsrc,  \
                                                          y_voff, soff, 0);                               \
                    if constexpr (X3) { /* bf16x3: the remainder goes to the lo plane */                  \
                        const __bf16 lv = (__bf16)(acc##i_[e] - (float)hv);                               \
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, lv),     \
                                                              yrsrc_lo, y_voff, soff, 0);                 \
                    }                                                                                     \
                } else {                                                                                  \
                    const float fv = acc##i_[e]; /* scalar copy: bit_cast of a vector ELEMENT is miscompiled */ \
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fv), yrsrc, y_voff, soff, 0);   \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        if (POOL) pool_group(a, acc##i_, negk, m0 + i_ * 32, h, col, cx.pool);                            \
    }
    if constexpr (SWAP) {
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            sc4[gq] = *reinterpret_cast<const float4*>(cstw + kBN + 8 * gq);
            sh4[gq] = *reinterpret_cast<const float4*>(cstw + 2 * kBN + 8 * gq);
        }
        // lane = frame r of the group, 8 channels per store: (r, h) -> channels 8h.. (+16 for the second store)
        const int voff = ((int)(threadIdx.x & 31) * a.ldy + (ln.col - n0 - (int)(threadIdx.x & 31)) + 8 * h) * 2;
        if constexpr (G > 0) store_acc(acc0, sc4, sh4, yrsrc, voff, 0 * 32 * a.ldy * 2);
        if constexpr (G > 1) store_acc(acc1, sc4, sh4, yrsrc, voff, 1 * 32 * a.ldy * 2);
        if constexpr (G > 2) store_acc(acc2, sc4, sh4, yrsrc, voff, 2 * 32 * a.ldy * 2);
        if constexpr (G > 3) store_acc(acc3, sc4, sh4, yrsrc, voff, 3 * 32 * a.ldy * 2);
    } else {
        XV_EPI(0) XV_EPI(1) XV_EPI(2) XV_EPI(3)
    }
#undef XV_EPI
#ifdef XVEC_DIAG
    SB();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long dt2 = __builtin_amdgcn_s_memtime();
    SB();
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned long long* d = g_diag + blockIdx.x * 8;
        d[3] += dt1 - dt0;     // K loop
        d[4] += dt2 - dt1;     // epilogue (issue side; stores still in flight)
        d[5] += 1;             // tiles
    }
#endif
}

template <int GUARD, bool POOL, bool STORE, bool INBF, bool OUTBF, bool X3>
__global__ __launch_bounds__(256, 2) void tdnn_kernel(const TdnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef XVEC_DIAG
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), r_entry = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < 8192) { g_diag[blockIdx.x * 8 + 3] = 0; g_diag[blockIdx.x * 8 + 4] = 0; g_diag[blockIdx.x * 8 + 5] = 0; }
#endif
    // logical id -> (row range p, channel column j); the n_tiles columns of one range are
    // consecutive ids on one XCD
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int j = lid % a.n_tiles;
    const int p = lid / a.n_tiles;
    // Row range of this block.  G groups over P ranges: sizes differ by at most one.  With two
    // blocks per CU the dispatcher places blocks b and b + grid/2 on the same CU (observed,
    // profiles/diag/placement.hip; speed only): the ranges that get the extra group are chosen
    // among the first-slot blocks first, so a CU's two blocks sum to the same work everywhere.
    int64_t g_begin, g_end;
    if (a.pair_period > 0) {
        const int P = a.blocks_per_col, PQ = a.pair_period, hq = PQ >> 1;
        const int64_t base = a.groups_total / P;
        const int rem = (int)(a.groups_total % P);
        const int rem1 = rem < (P >> 1) ? rem : (P >> 1), rem2 = rem - rem1;
        const int xq = p / PQ, w = p % PQ;
        const int nf = xq * hq + (w < hq ? w : hq);          // first-slot ranges before p
        const int ns = xq * hq + (w > hq ? w - hq : 0);      // second-slot ranges before p
        g_begin = base * p + (nf < rem1 ? nf : rem1) + (ns < rem2 ? ns : rem2);
        const bool extra = (w < hq) ? (nf < rem1) : (ns < rem2);
        g_end = g_begin + base + (extra ? 1 : 0);
    } else {
        g_begin = a.groups_total * (int64_t)p / a.blocks_per_col;
        g_end = a.groups_total * (int64_t)(p + 1) / a.blocks_per_col;
    }
    const int n0 = j * kBN;
    const int n_chunks = a.n_taps * a.cpt;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    Lane ln;
    ln.h = lane >> 5;
    const int r = lane & 31;
    // staging map: thread -> (row r0 + 32*j, 16-byte chunk c) of a 32-wide K chunk
    ln.c = tid & 7;
    ln.r0 = tid >> 3;
    ln.st_off = ln.r0 * kBK + ((ln.c ^ ((ln.r0 >> 1) & 7)) << 2);
    // fragment read map: row (base + r), logical 16-B chunk 2q+h, swizzled by row
    ln.sw = (r >> 1) & 7;
    ln.a_rd = r * kBK;
    ln.b_rd = kBM * kBK + (wave * 32 + r) * kBK;
    ln.col = n0 + wave * 32 + r;
    if (tid < kBN) {
        smem[2 * kStageFloats + tid] = a.bias[n0 + tid];
        smem[2 * kStageFloats + kBN + tid] = POOL ? 0.f : a.scale[n0 + tid];
        smem[2 * kStageFloats + 2 * kBN + tid] = POOL ? 0.f : a.shift[n0 + tid];
    }

    Ctx cx;
    cx.g_s = g_begin;
    cx.g_end = g_end;
    cx.m0 = g_begin * 32;
    constexpr int ES = INBF ? 2 : 4;
    cx.es = ES;
    cx.xrsrc = x_rsrc<GUARD>(a, cx);
    cx.wrsrc = make_rsrc(static_cast<const char*>(a.W) + (int64_t)n0 * a.k_pad * ES);
    cx.x_base = ln.r0 * a.ldx * ES + ln.c * 16;
    cx.w_toff = ln.r0 * a.k_pad * ES + ln.c * 16;
    cx.r0 = ln.r0;
    if (INBF) {   // 1 KiB per (column tile of 32 channels, k-step of 16)
        const int64_t ct = n0 / 32 + wave;
        cx.wfrsrc = make_rsrc(static_cast<const char*>(a.Wf) + ct * (int64_t)(a.k_pad / 16) * 1024);
        cx.wf_voff = lane * 16;
    }
    cx.u_tile = __builtin_amdgcn_readfirstlane(utt_of_row(a.out_map, cx.m0));
    cx.off_next = row_off(a.out_map, cx.u_tile + 1);
    if (POOL) {
        cx.pool.u = cx.u_tile;
        cx.pool.end = cx.off_next;
    }
    cx.pivot_set = false;
    set_tile_rows(a, cx);
    cx.tap = 0;
    cx.kc = 0;
    cx.itl = 0;

    Regs rg;
    block_prologue<GUARD, INBF, X3>(a, smem, cx, rg, ln, n_chunks);
    int64_t g = g_begin;
    for (; g + 4 <= g_end; g += 4) process_tile<4, GUARD, POOL, STORE, INBF, OUTBF, X3>(a, smem, cx, rg, ln, g, n0, n_chunks);
    const int rem = (int)(g_end - g);
    if (rem == 3) process_tile<3, GUARD, POOL, STORE, INBF, OUTBF, X3>(a, smem, cx, rg, ln, g, n0, n_chunks);
    else if (rem == 2) process_tile<2, GUARD, POOL, STORE, INBF, OUTBF, X3>(a, smem, cx, rg, ln, g, n0, n_chunks);
    else if (rem == 1) process_tile<1, GUARD, POOL, STORE, INBF, OUTBF, X3>(a, smem, cx, rg, ln, g, n0, n_chunks);
#ifdef XVEC_DIAG
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* d = g_diag + blockIdx.x * 8;
        d[0] = t_entry;
        d[1] = __builtin_amdgcn_s_memtime();
        d[2] = (unsigned long long)(g_end - g_begin);
        d[6] = r_entry;                              // 100 MHz reference clock at entry / exit: core clock of the launch
        d[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int GUARD, bool POOL, bool STORE, bool INBF, bool OUTBF, bool X3 = false>
static hipError_t launch_variant(const TdnnArgs& a, hipStream_t s) {
    auto kern = tdnn_kernel<GUARD, POOL, STORE, INBF, OUTBF, X3>;
    static LdsOptIn opt;            // per variant and device
    if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(kern), kLdsBytes); e != hipSuccess) return e;
    const int grid = a.blocks_per_col * a.n_tiles;
    kern<<<dim3(grid), dim3(256), kLdsBytes, s>>>(a);
    return hipGetLastError();
}

#ifdef XVEC_DIAG
extern "C" int xvec_diag_read(unsigned long long* host, int n_words) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_diag), (size_t)n_words * 8);
}
#endif

hipError_t launch_tdnn(const TdnnArgs& a, TdnnVariant v, hipStream_t s) {
    if (a.groups_total <= 0 || a.blocks_per_col <= 0 || a.blocks_per_col > a.groups_total || (a.cpt & 1))
        return hipErrorInvalidValue;
    const bool x3 = a.terms == 2;     // bf16x3: compile-time mode of the bf16 instantiations
    switch (v) {
        case TdnnVariant::kF32First: return launch_variant<true, false, true, false, false>(a, s);
        case TdnnVariant::kF32: return launch_variant<false, false, true, false, false>(a, s);
        case TdnnVariant::kF32Pool: return launch_variant<false, true, false, false, false>(a, s);
        case TdnnVariant::kBf16FirstSrc32: return launch_variant<2, false, true, true, true>(a, s);
        case TdnnVariant::kBf16First:
            return x3 ? launch_variant<true, false, true, true, true, true>(a, s) : launch_variant<true, false, true, true, true>(a, s);
        case TdnnVariant::kBf16:
            return x3 ? launch_variant<false, false, true, true, true, true>(a, s) : launch_variant<false, false, true, true, true>(a, s);
        case TdnnVariant::kBf16Pool:
            return x3 ? launch_variant<false, true, false, true, false, true>(a, s) : launch_variant<false, true, false, true, false>(a, s);
        case TdnnVariant::kBf16ToF32:
            return x3 ? launch_variant<false, false, true, true, false, true>(a, s) : launch_variant<false, false, true, true, false>(a, s);
        case TdnnVariant::kBf16FirstToF32:
            return x3 ? launch_variant<true, false, true, true, false, true>(a, s) : launch_variant<true, false, true, true, false>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace xvec

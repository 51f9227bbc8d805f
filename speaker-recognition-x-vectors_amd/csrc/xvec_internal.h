// Internal declarations shared by the gfx950 kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace xvec {

constexpr int kBK = 32;          // K-chunk (fp32 elements) staged per main-loop step
constexpr int kRowPadTail = 264; // readable rows past M_pad: a masked tile of tdnn_pp16.hip may reach two 64-row units (+63 of
                                 // rounding) past the last valid row, + max tap reach (6); tdnn_layer.hip's look-ahead needs 134

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int64_t round_up64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Row layout of one layer's activations: the batch is ONE matrix of frames, utterance u in rows
//   [row_off(u), row_off(u+1)),  row_off(u) = (offsets ? offsets[u] : u*fixed_T) - u*cum
// where offsets / fixed_T describe the rows of the layer-1 INPUT and cum is the number of frames
// the valid convolutions have consumed so far (0, 4, 8, 14, 14, 14).  Every layer's output is
// compact: only valid frames exist, nothing is computed for frames whose context is missing.
struct RowMap {
    const int64_t* offsets;   // device [n_utts+1], or nullptr for fixed length
    int n_utts;
    int fixed_T;
    int cum;
};
#ifdef __HIPCC__
__device__ __forceinline__ int64_t row_off(const RowMap& m, int u) {
    return (m.offsets ? m.offsets[u] : (int64_t)u * m.fixed_T) - (int64_t)u * m.cum;
}
// utterance holding compact row p (clamped to the last utterance for rows past the end)
__device__ __forceinline__ int utt_of_row(const RowMap& m, int64_t p) {
    if (m.offsets == nullptr) {
        const int64_t u = p / (m.fixed_T - m.cum);
        return (int)(u < m.n_utts - 1 ? u : m.n_utts - 1);
    }
    int lo = 0, hi = m.n_utts;                // largest u with row_off(u) <= p
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (row_off(m, mid) <= p) lo = mid; else hi = mid;
    }
    return lo;
}
#endif

// Opt-in to more than 64 KiB of dynamic LDS, once per (kernel, device): the attribute belongs to the
// device that is current when it is set, so one flag per process would leave a second device without it.
struct LdsOptIn {
    bool done[64] = {};
    hipError_t ensure(const void* fn, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64 || !done[dev]) {
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) done[dev] = true;   // benign if raced (idempotent)
        }
        return hipSuccess;
    }
};

// One frame-level layer as an implicit GEMM over the flat frame axis:
//   Y[p, n] = bn( relu( sum_{tap,c} X[p + u(p)*span + tap*tap_rows, c] * W[n, tap, c] + bias[n] ) )
// for every compact output row p; u(p)*span re-bases the row into the (longer) input layout.
struct TdnnGeom {
    int cin;        // valid input channels per tap
    int cout;       // valid output channels
    int n_taps;     // taps after folding (contiguous dilation-1 contexts fold into one tap)
    int tap_rows;   // row step between taps (dilation); 0 when n_taps == 1
    int kpt;        // valid K per tap after folding
    int kpt_pad;    // kpt rounded up to kBK
    int k_pad;      // n_taps * kpt_pad  (row length of the packed weight)
    int n_pad;      // cout rounded up to 128 (row count of the packed weight, ld of Y)
    int ctx_span;   // c[-1]-c[0]: frames lost by this layer
    int tap_stride_src; // for packing: dest k of (tap, c) = tap*tap_stride + c
    int src_taps;       // taps in the PyTorch weight
    int src_cin;        // channels per tap in the PyTorch weight
    int chunk_k;        // K elements per 128-byte chunk (32 fp32, 64 bf16): unit of the packed K order
    int terms;          // bf16x3 weight stream: 2 (per chunk: W_hi blocks, W_lo blocks), else 1
};

struct TdnnArgs {
    const void* X;      // [rows][ldx]  fp32 or bf16
    const void* W;      // fp32: packed row-major [n_pad][k_pad]
    const void* Wf;     // bf16: fragment-major packing of the same matrix (see pack.hip)
    const float* bias;  // [n_pad]
    const float* scale; // [n_pad]  folded BatchNorm: y = relu(v)*scale + shift
    const float* shift; // [n_pad]
    void* Y;            // [m_pad][ldy]  fp32 or bf16
    int64_t x_rows;     // rows of X that may be read (guarded variant)
    int ldx, ldy;
    int n_taps, tap_rows, kpt, cpt;   // cpt = chunks per tap = kpt_pad / kBK
    int k_pad;
    int n_tiles;              // 128-channel columns
    int blocks_per_col;       // persistent blocks per column; grid = n_tiles * blocks_per_col
    int64_t groups_total;     // 32-row groups of the flat frame axis (ceil(rows / 32))
    int pair_period;          // >0: row ranges per XCD run; enables CU-pair-aware range sizes (see kernel)
    RowMap out_map;           // row layout of THIS layer's output
    int span;                 // frames this layer consumes (c[-1]-c[0]): input row = p + u(p)*span
    // fused statistics-pooling epilogue (layer 5)
    float* pool_part;         // [slots][3][n_pad]: pivot K | sum (r-K) | sum (r-K)^2, r = relu(z + bias), per (32-row group, utterance)
                              // (tdnn_pp16.hip: per (block of the column, utterance, frames half) -- see there)
    int* pool_cnt;            // tdnn_pp16.hip only: frames behind each of its partials, [slots]
    // bf16x3 (fp32 values carried as two bf16 planes hi + lo, three bf16 products per k-step:
    // x_hi*W_hi + x_hi*W_lo + x_lo*W_hi).  terms == 2: X has a lo plane x_plane_bytes after the hi
    // plane and Wf holds, per chunk, the W_hi fragments followed by the W_lo fragments.
    int terms;                // 1 (plain) or 2
    int x_plane_bytes;        // byte distance from the hi plane of X to its lo plane
    int y_plane_bytes;        // > 0: bf16 output as two planes, lo plane this many bytes after Y
    int64_t x_bytes;          // guarded variant: readable bytes from X (0: x_rows * ldx * element size)
};

// kernel instantiations: input/weight arithmetic x epilogue
enum class TdnnVariant {
    kF32First,        // fp32, guarded reads of the caller's rows (layer 1)
    kF32,             // fp32 -> fp32
    kF32Pool,         // fp32 -> pooling partials only (layer 5)
    kBf16First,       // layer 1 of the bf16 path: guarded reads of the bf16-converted MFCC rows
    kBf16FirstSrc32,  // the same reading the caller's fp32 rows, rounded to bf16 on the way in (no pack pass)
    kBf16,            // bf16 -> bf16
    kBf16Pool,        // bf16 -> pooling partials only
    kBf16ToF32,       // bf16 -> fp32 (per-layer test entry: layer 5, and every layer in bf16x3)
    kBf16FirstToF32   // layer 1, guarded, bf16 -> fp32 (per-layer test entry in bf16x3)
};
hipError_t launch_tdnn(const TdnnArgs& a, TdnnVariant v, hipStream_t s);
// Large-batch bf16 mapping (tdnn_pp16.hip, v_mfma_f32_16x16x32_bf16): 256-channel columns, 64-frame units.  Reads TdnnArgs with
//   W = K-tile major bf16 [n_pad/256][k_pad/64][256][64] (K order as the fp32 packing, 64-element chunks), n_tiles = n_pad / 256,
//   groups_total = ceil(rows / 64) units, blocks_per_col ranges per column (>= 1.8 units each: the measured crossover with the 128x128 kernel).
//   terms == 2 (bf16x3): X / Y are hi and lo planes x_plane_bytes / y_plane_bytes apart, W the tripled packing below.
hipError_t launch_tdnn_pp16(const TdnnArgs& a, bool pool, hipStream_t s);
// Layer 1 of the bf16 path as a streaming kernel (tdnn_first.hip): weights resident in registers, 16-byte stores.
// Reads TdnnArgs as the 128x128 kernel does (Wf = fragment-major bf16 weights); X holds the caller's fp32 rows.
// (Wf = the layer's fragment-major copy WITH the bias in k slots kpt, kpt + 1: launch_patch_bias_kslots)
bool tdnn_first_applicable(const TdnnArgs& a);
hipError_t launch_patch_bias_kslots(const float* bias, const TdnnGeom& geo, void* Wf16, hipStream_t s);
hipError_t launch_tdnn_first(const TdnnArgs& a, int num_cu, hipStream_t s);
// ... and of the bf16x3 path: terms == 2, Wf = the bf16x3 fragment stream, Y = two bf16 planes y_plane_bytes apart; X the caller's fp32 rows
bool tdnn_first3_applicable(const TdnnArgs& a);
hipError_t launch_tdnn_first3(const TdnnArgs& a, int num_cu, hipStream_t s);
// K-tile major bf16 copy of the packed weights for it
// (in_scale: nullptr, or the producing layer's folded BatchNorm scale per input channel -- plain bf16 defers BatchNorm, pack.hip)
hipError_t launch_pack_tdnn_rows_bf16(const float* W, const float* in_scale, const TdnnGeom& geo, void* Wr16, hipStream_t s);
// plain bf16: bias' = bias + W . shift_prev | scale' = 1 | shift' = 0 (3 x n_pad floats), see pack.hip
hipError_t launch_fold_bias(const float* W, const float* bias, const float* in_shift, const TdnnGeom& geo, float* vec16,
                            hipStream_t s);
// ... and for its bf16x3 form (a.terms == 2): [n_pad/256][3 * k_pad/64][256][64], per K-tile W_hi | W_lo | W_hi
hipError_t launch_pack_tdnn_rows_bf16x3(const float* W, const TdnnGeom& geo, void* Wr48, hipStream_t s);

struct PoolArgs {
    const float* X;          // [B][T][C]
    float* out;              // [B][2C]
    const int32_t* lengths;  // device [B] valid frames per utterance, or nullptr (all T)
    int B, T, C;
};
hipError_t launch_stat_pool(const PoolArgs& a, hipStream_t s);

struct PoolFinalizeArgs {
    const float* part;       // [slots][3][n_pad]: pivot K | sum (r-K) | sum (r-K)^2 of r = relu(z + bias) per (sub-tile, utterance)
    // segment form (partials of tdnn_pp16.hip: one per (block of the column, utterance, frames half), slot 2 (b + u) + g):
    const int* cnt;          // frames behind each partial, or nullptr for the sub-tile form
    int64_t units_total;     // 64-frame units of the layer's output and blocks per column of the launch that wrote the
    int blocks_per_col;      // partials: block b owns units [units_total b / blocks_per_col, units_total (b+1) / blocks_per_col)
    float* out;              // [B][2C]
    RowMap map;              // row layout of the pooled activation (layer 5 output)
    int C, n_pad, sub_rows;
    const float* scale;      // folded BatchNorm of layer 5, applied here: y = scale*r + shift
    const float* shift;
};
hipError_t launch_pool_finalize(const PoolFinalizeArgs& a, hipStream_t s);

// y[M,N] = act(x[M,K] . W[N,K]^T + b)   (PyTorch nn.Linear layout, fp32 MFMA).  With `scratch` (device memory the
// call may overwrite) a small problem is split over K into scratch and reduced by a second kernel.  With W3
// (launch_split_pairs of W) the split form's products are bf16x3 instead of fp32 MFMAs: the segment layers of
// XVEC_BF16.
hipError_t launch_affine_f32(const float* x, const float* W, const float* b, float* y, int M, int N,
                             int K, int relu, hipStream_t s, float* scratch = nullptr, size_t scratch_bytes = 0,
                             const void* W3 = nullptr);
// n floats (n % 4 == 0) -> n dwords: four consecutive values as bf16 pairs hi01 | hi23 | lo01 | lo23
hipError_t launch_split_pairs(const float* W, void* out, int64_t n, hipStream_t s);

// weight packing
hipError_t launch_pack_tdnn(const float* W, const float* bias, const float* g, const float* be,
                            const float* mu, const float* var, float eps, const TdnnGeom& geo,
                            float* Wp, float* bias_p, float* scale_p, float* shift_p, hipStream_t s);
// bf16 fragment-major packing: block (column tile ct of 32 channels, k-step ks of 16) = 64 lanes x 8 bf16,
// lane (r,h) element j = W[32*ct + r][16*ks + 8*h + j]
hipError_t launch_pack_tdnn_bf16(const float* W, const float* in_scale, const TdnnGeom& geo, void* Wf16, hipStream_t s);
// x[B,T,C] (+lengths) -> packed rows [sum len, c_pad] (fp32 or bf16); offsets on device
// un_scale / un_shift: nullptr, or a folded BatchNorm to invert on the way in (per-layer entries in plain bf16, pack.hip)
hipError_t launch_pack_rows(const float* x, const int64_t* offsets, int B, int T, int C, int c_pad,
                            void* out, bool out_bf16, hipStream_t s, const float* un_scale = nullptr,
                            const float* un_shift = nullptr);
// x[rows][C] fp32 -> bf16 planes hi | lo, each [.][c_pad], lo plane plane_elems elements after hi (bf16x3)
hipError_t launch_pack_rows_split(const float* x, int64_t rows, int C, int c_pad, int64_t plane_elems, void* out,
                                  hipStream_t s);
// flat [rows, ld] (fp32 or bf16) -> compact fp32 y[B, T_out, C]
hipError_t launch_unpack_rows_split(const void* flat, int64_t plane_elems, int ld, int B, int T_in, int T_out, int C,
                                    float* y, hipStream_t s);
// scale / shift: nullptr, or a folded BatchNorm applied on the way out
hipError_t launch_unpack_rows(const void* flat, bool in_bf16, int ld, int B, int T_in, int T_out, int C,
                              float* y, hipStream_t s, const float* scale = nullptr, const float* shift = nullptr);

}  // namespace xvec

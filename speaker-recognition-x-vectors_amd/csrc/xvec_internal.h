// Internal declarations shared by the gfx950 kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace xvec {

constexpr int kBK = 32;          // K-chunk (fp32 elements) staged per main-loop step
constexpr int kRowPadTail = 136; // readable rows past M_pad: a 4-group look-ahead fetch (128) + max tap reach (6)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int64_t round_up64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// One frame-level layer as an implicit GEMM over the FLAT frame axis:
//   Y[p, n] = bn( relu( sum_{tap,c} X[p + tap*tap_rows, c] * W[n, tap, c] + bias[n] ) )
// for every flat row p of the packed batch.  Rows whose receptive field leaves their
// utterance are computed too (garbage in, garbage out) and never read by a valid row
// of the next layer -- see DESIGN.md "flat frame axis".
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
};

struct TdnnArgs {
    const void* X;      // [rows][ldx]  fp32 or bf16
    const void* W;      // packed [n_pad][k_pad]  fp32 or bf16
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
    // fused statistics-pooling epilogue (layer 5)
    float* pool_part;         // [slots][2][n_pad] (mean, M2) per (32-row group, utterance)
    const int64_t* offsets;   // device [B+1] row offsets, or nullptr for fixed length
    int n_utts;
    int fixed_T;              // frames per utterance when offsets == nullptr
    int shrink;               // pooled frames of utterance u = len_u - shrink
};

// kernel instantiations: input/weight arithmetic x epilogue
enum class TdnnVariant {
    kF32First,        // fp32, guarded reads of the caller's rows (layer 1)
    kF32,             // fp32 -> fp32
    kF32Pool,         // fp32 -> pooling partials only (layer 5)
    kF32PoolStore,    // fp32 -> fp32 + pooling partials
    kF32FirstToBf16,  // layer 1 of the bf16 path: fp32 MFMA on the fp32 MFCCs, bf16 activations out
    kBf16,            // bf16 -> bf16
    kBf16Pool,        // bf16 -> pooling partials only
    kBf16ToF32        // bf16 -> fp32 (per-layer test entry of layer 5)
};
hipError_t launch_tdnn(const TdnnArgs& a, TdnnVariant v, hipStream_t s);

struct PoolArgs {
    const float* X;          // [rows][ld]
    float* out;              // [B][2C]
    const int64_t* offsets;  // device [B+1] or nullptr
    const int32_t* lengths;  // device [B] or nullptr (stand-alone masked pooling on [B,T,C])
    int B, C, ld;
    int fixed_T;             // row stride between utterances when offsets == nullptr
    int fixed_n;             // frames pooled per utterance when neither offsets nor lengths
    int shrink;              // with offsets: n_u = len_u - shrink; with lengths: n_u = lengths[u]
};
hipError_t launch_stat_pool(const PoolArgs& a, hipStream_t s);

struct PoolFinalizeArgs {
    const float* part;       // [slots][2][n_pad]
    float* out;              // [B][2C]
    const int64_t* offsets;
    int B, C, n_pad, fixed_T, shrink, sub_rows;
};
hipError_t launch_pool_finalize(const PoolFinalizeArgs& a, hipStream_t s);

// y[M,N] = act(x[M,K] . W[N,K]^T + b)   (PyTorch nn.Linear layout, fp32 MFMA)
hipError_t launch_affine_f32(const float* x, const float* W, const float* b, float* y, int M, int N,
                             int K, int relu, hipStream_t s);

// weight packing
hipError_t launch_pack_tdnn(const float* W, const float* bias, const float* g, const float* be,
                            const float* mu, const float* var, float eps, const TdnnGeom& geo,
                            float* Wp, float* bias_p, float* scale_p, float* shift_p, hipStream_t s);
hipError_t launch_pack_tdnn_bf16(const float* W, const TdnnGeom& geo, void* Wp16, hipStream_t s);
// x[B,T,C] (+lengths) -> packed rows [sum len, c_pad] (fp32 or bf16); offsets on device
hipError_t launch_pack_rows(const float* x, const int64_t* offsets, int B, int T, int C, int c_pad,
                            void* out, bool out_bf16, hipStream_t s);
// flat [rows, ld] (fp32 or bf16) -> compact fp32 y[B, T_out, C]
hipError_t launch_unpack_rows(const void* flat, bool in_bf16, int ld, int B, int T_in, int T_out, int C,
                              float* y, hipStream_t s);

}  // namespace xvec

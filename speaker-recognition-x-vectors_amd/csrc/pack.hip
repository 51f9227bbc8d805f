// Layout kernels: weight re-packing at load time and row packing/unpacking at the boundary.
// All are small, HBM-bound element-wise copies.
#include "xvec_internal.h"

namespace xvec {

// The kernel walks K in 128-byte chunks with the taps innermost (chunk kc of tap 0, of tap 1, ...,
// then chunk kc+1), so the packed K axis is stored in that order:
//   packed k = (kc*n_taps + tap)*chunk_k + w   <->   tap-major k = tap*tap_stride + kc*chunk_k + w.
__device__ __forceinline__ int tap_major_k(const TdnnGeom& g, int kd) {
    if (g.n_taps == 1) return kd;
    const int it = kd / g.chunk_k, w = kd % g.chunk_k;
    return (it % g.n_taps) * g.tap_stride_src + (it / g.n_taps) * g.chunk_k + w;
}

// PyTorch TdnnLayer.linear.weight[out, taps*cin] (column = tap*cin + c, tdnn_layer.py:19,29)
//   -> Wp[n_pad][k_pad], zero padded, K in the order above.
template <typename TO>
__global__ void pack_tdnn_weight_kernel(const float* __restrict__ W, TdnnGeom g, TO* __restrict__ Wp) {
    const int64_t total = (int64_t)g.n_pad * g.k_pad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / g.k_pad), kd = tap_major_k(g, (int)(i % g.k_pad));
        const int tap = kd / g.tap_stride_src, c = kd % g.tap_stride_src;
        float v = 0.f;
        if (n < g.cout && tap < g.src_taps && c < g.src_cin)
            v = W[(int64_t)n * (g.src_taps * g.src_cin) + tap * g.src_cin + c];
        Wp[i] = (TO)v;
    }
}

// bias and eval-mode BatchNorm1d folded to y = relu(v)*scale + shift (tdnn_layer.py:36-39):
//   scale = gamma / sqrt(var + eps), shift = beta - mean*scale;   no BN: scale 1, shift 0.
// Padded channels get bias = scale = shift = 0 so they stay exactly zero downstream.
__global__ void pack_tdnn_vec_kernel(const float* __restrict__ bias, const float* __restrict__ gam,
                                     const float* __restrict__ bet, const float* __restrict__ mu,
                                     const float* __restrict__ var, float eps, int cout, int n_pad,
                                     float* __restrict__ bias_p, float* __restrict__ scale_p,
                                     float* __restrict__ shift_p) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_pad) return;
    float b = 0.f, sc = 0.f, sh = 0.f;
    if (n < cout) {
        b = bias[n];
        if (gam) {
            sc = gam[n] / sqrtf(var[n] + eps);
            sh = bet[n] - mu[n] * sc;
        } else {
            sc = 1.f;
        }
    }
    bias_p[n] = b;
    scale_p[n] = sc;
    shift_p[n] = sh;
}

hipError_t launch_pack_tdnn(const float* W, const float* bias, const float* g, const float* be,
                            const float* mu, const float* var, float eps, const TdnnGeom& geo,
                            float* Wp, float* bias_p, float* scale_p, float* shift_p, hipStream_t s) {
    pack_tdnn_weight_kernel<float><<<1024, 256, 0, s>>>(W, geo, Wp);
    pack_tdnn_vec_kernel<<<(geo.n_pad + 255) / 256, 256, 0, s>>>(bias, g, be, mu, var, eps, geo.cout,
                                                                 geo.n_pad, bias_p, scale_p, shift_p);
    return hipGetLastError();
}

// Same matrix in bf16, fragment-major for v_mfma_f32_32x32x16_bf16: the 64 lanes' B operands of one
// (32-channel column tile, 16-wide k-step) are one contiguous KiB, so a wave fetches them with a
// single coalesced 16-byte-per-lane load and the weights never pass through LDS.
// in_scale (or nullptr): the folded BatchNorm scale of the PRODUCING layer, per input channel -- plain bf16 defers every
// layer's BatchNorm into its consumer's weights (xvec_api.hip, refold): W'[n, tap, c] = W[n, tap, c] * scale_prev[c]
__global__ void pack_tdnn_weight_frag_kernel(const float* __restrict__ W, const float* __restrict__ in_scale, TdnnGeom g,
                                             __bf16* __restrict__ Wf) {
    // g.terms == 2 (bf16x3): the stream holds every 64-wide chunk twice -- its four W_hi k-step blocks,
    // then its four W_lo blocks, W = W_hi + W_lo + O(2^-17) -- as the kernel's XV_GLB3 reads them
    const int terms = g.terms > 1 ? g.terms : 1;
    const int64_t total = (int64_t)g.n_pad * g.k_pad * terms;
    const int ksteps = g.k_pad * terms / 16;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t blk = i >> 9;
        const int within = (int)(i & 511), lane = within >> 3, j = within & 7;
        const int ct = (int)(blk / ksteps), ks = (int)(blk % ksteps);
        const int ks_stream = ks * 16 + 8 * (lane >> 5) + j;                 // position in the chunk stream
        const int chunk_s = ks_stream / g.chunk_k, w = ks_stream % g.chunk_k;
        const int term = chunk_s % terms, chunk = chunk_s / terms;
        const int n = ct * 32 + (lane & 31), kd = tap_major_k(g, chunk * g.chunk_k + w);
        const int tap = kd / g.tap_stride_src, c = kd % g.tap_stride_src;
        float v = 0.f;
        if (n < g.cout && tap < g.src_taps && c < g.src_cin) {
            v = W[(int64_t)n * (g.src_taps * g.src_cin) + tap * g.src_cin + c];
            if (in_scale) v *= in_scale[c];
        }
        const __bf16 hi = (__bf16)v;
        Wf[i] = (terms > 1 && term == terms - 1) ? (__bf16)(v - (float)hi) : hi;
    }
}

hipError_t launch_pack_tdnn_bf16(const float* W, const float* in_scale, const TdnnGeom& geo, void* Wf16, hipStream_t s) {
    pack_tdnn_weight_frag_kernel<<<1024, 256, 0, s>>>(W, in_scale, geo, static_cast<__bf16*>(Wf16));
    return hipGetLastError();
}

// bf16 weights for tdnn_pp16.hip, K-TILE major: [256-channel column block][K-tile of 64][256 rows][64 k]
// (K order as everywhere: 64-element chunks, taps innermost).  Both operands of that kernel reach LDS by
// DMA in 128-byte row slabs; a K-tile of a column block is then one contiguous 32 KiB, so the 256 CUs
// that fetch the same tile at the same time spread over all L2 channels (128-byte slabs of plain rows,
// 1-3 KiB apart, fall on two of them).
__global__ void pack_tdnn_weight_ktile_kernel(const float* __restrict__ W, const float* __restrict__ in_scale, TdnnGeom g,
                                              __bf16* __restrict__ Wt) {
    const int64_t total = (int64_t)g.n_pad * g.k_pad;
    const int nk = g.k_pad / 64;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i & 63), r = (int)((i >> 6) & 255);
        const int64_t t = i >> 14;                       // (column block, K-tile)
        const int cb = (int)(t / nk), q = (int)(t % nk);
        // row 16*b + c of a wave's 64 rows (wave wc = r >> 6) holds channel 64*wc + 4*c + b: lane c's four 16x16 accumulators
        // of a frame then own FOUR adjacent channels, which the epilogues of tdnn_pp16.hip write as one 8-byte piece
        const int n = cb * 256 + (r & ~63) + 4 * (r & 15) + ((r >> 4) & 3);
        const int kd = tap_major_k(g, q * 64 + w);
        const int tap = kd / g.tap_stride_src, c = kd % g.tap_stride_src;
        float v = 0.f;
        if (n < g.cout && tap < g.src_taps && c < g.src_cin) {
            v = W[(int64_t)n * (g.src_taps * g.src_cin) + tap * g.src_cin + c];
            if (in_scale) v *= in_scale[c];        // (see pack_tdnn_weight_frag_kernel)
        }
        Wt[i] = (__bf16)v;
    }
}
// The same for the bf16x3 form of tdnn_pp16.hip: every K-tile three times -- W_hi, W_lo = bf16(W - W_hi), W_hi again --
// in the order its K loop meets them (hi slab x W_hi, hi slab x W_lo, lo slab x W_hi): [column block][3 * K-tile + j][256][64]
__global__ void pack_tdnn_weight_ktile3_kernel(const float* __restrict__ W, TdnnGeom g, __bf16* __restrict__ Wt) {
    const int64_t total = (int64_t)g.n_pad * g.k_pad * 3;
    const int nk3 = 3 * (g.k_pad / 64);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i & 63), r = (int)((i >> 6) & 255);
        const int64_t t = i >> 14;                       // (column block, K-tile of the tripled sequence)
        const int cb = (int)(t / nk3), q3 = (int)(t % nk3);
        const int q = q3 / 3, j = q3 % 3;
        const int n = cb * 256 + (r & ~63) + 4 * (r & 15) + ((r >> 4) & 3);      // row order: pack_tdnn_weight_ktile_kernel
        const int kd = tap_major_k(g, q * 64 + w);
        const int tap = kd / g.tap_stride_src, c = kd % g.tap_stride_src;
        float v = 0.f;
        if (n < g.cout && tap < g.src_taps && c < g.src_cin)
            v = W[(int64_t)n * (g.src_taps * g.src_cin) + tap * g.src_cin + c];
        const __bf16 hi = (__bf16)v;
        Wt[i] = j == 1 ? (__bf16)(v - (float)hi) : hi;
    }
}
hipError_t launch_pack_tdnn_rows_bf16x3(const float* W, const TdnnGeom& geo, void* Wr48, hipStream_t s) {
    if (geo.n_pad % 256 != 0) return hipSuccess;
    pack_tdnn_weight_ktile3_kernel<<<1024, 256, 0, s>>>(W, geo, static_cast<__bf16*>(Wr48));
    return hipGetLastError();
}
hipError_t launch_pack_tdnn_rows_bf16(const float* W, const float* in_scale, const TdnnGeom& geo, void* Wr16, hipStream_t s) {
    if (geo.n_pad % 256 != 0) return hipSuccess;        // tdnn_pp16.hip is not used for such a layer
    pack_tdnn_weight_ktile_kernel<<<1024, 256, 0, s>>>(W, in_scale, geo, static_cast<__bf16*>(Wr16));
    return hipGetLastError();
}

// Epilogue constants of a layer in plain bf16, whose BatchNorm is deferred into the NEXT layer (xvec_api.hip, refold):
// this layer's bias absorbs the producing layer's folded shift,
//   bias'[n] = bias[n] + sum_{tap, c} W[n, tap, c] * shift_prev[c]        (exact fp32 weights, fp64 sum)
// and its own scale / shift leave the frame-level kernels (scale' = 1, shift' = 0: they store relu(z + bias')).
// One wave per output channel; padded channels get bias' = scale' = shift' = 0 and stay exactly zero downstream.
__global__ __launch_bounds__(64) void fold_bias_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                       const float* __restrict__ in_shift, int cout, int n_pad, int taps, int cin,
                                                       float* __restrict__ out) {
    const int n = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    if (n < cout && in_shift) {
        const float* w = W + (int64_t)n * taps * cin;
        for (int k = lane; k < taps * cin; k += 64) acc += (double)w[k] * (double)in_shift[k % cin];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        out[n] = n < cout ? (float)((double)bias[n] + acc) : 0.f;
        out[n_pad + n] = n < cout ? 1.f : 0.f;
        out[2 * n_pad + n] = 0.f;
    }
}
hipError_t launch_fold_bias(const float* W, const float* bias, const float* in_shift, const TdnnGeom& geo, float* vec16,
                            hipStream_t s) {
    fold_bias_kernel<<<geo.n_pad, 64, 0, s>>>(W, bias, in_shift, geo.cout, geo.n_pad, geo.src_taps, geo.src_cin, vec16);
    return hipGetLastError();
}

// Layer 1 of plain bf16 on the streaming kernel (tdnn_first.hip): its copy of the fragment-major weights carries the bias in
// the two spare k slots k = kpt, kpt + 1 of the padded K -- bf16(b) and bf16(b - bf16(b)); the staged input holds 1.0 there.
// Element (n, k) of the fragment-major packing: ((n / 32 * ksteps + k / 16) * 64 + n % 32 + 32 * (k / 8 % 2)) * 8 + k % 8.
__global__ void patch_bias_kslots_kernel(const float* __restrict__ bias, int cout, int kpt, int ksteps, __bf16* __restrict__ Wf) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= cout) return;
    const float b = bias[n];
    const __bf16 hi = (__bf16)b, lo = (__bf16)(b - (float)hi);
    for (int j = 0; j < 2; ++j) {
        const int k = kpt + j;
        Wf[((int64_t)((n >> 5) * ksteps + (k >> 4)) * 64 + (n & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7)] = j ? lo : hi;
    }
}
hipError_t launch_patch_bias_kslots(const float* bias, const TdnnGeom& geo, void* Wf16, hipStream_t s) {
    patch_bias_kslots_kernel<<<(geo.cout + 255) / 256, 256, 0, s>>>(bias, geo.cout, geo.kpt, geo.k_pad / 16, static_cast<__bf16*>(Wf16));
    return hipGetLastError();
}

// x[B,T,C] -> packed rows out[offsets[u] + t][c_pad] for t < len_u (zero padded channels).
// un_scale / un_shift (or nullptr): the folded BatchNorm of the layer that PRODUCED x, inverted on the way in --
// r = (x - shift) / scale, 0 where |scale| < 1e-18 (gamma = 0 or denormal-small: the reference's BatchNorm output is then
// x = shift whatever r was, the channel meets folded weights W * scale that are zero or about to flush, and the quotient
// would overflow to Inf -- Inf * 0 = NaN) -- for the per-layer entries in
// plain bf16, whose kernels read the producing layer's relu output and carry its BatchNorm in their weights.
template <typename TO>
__global__ void pack_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ offsets, int T,
                                 int C, int c_pad, TO* __restrict__ out, const float* __restrict__ un_scale,
                                 const float* __restrict__ un_shift) {
    const int u = blockIdx.y;
    const int64_t off = offsets ? offsets[u] : (int64_t)u * T;
    const int64_t len = offsets ? offsets[u + 1] - off : T;
    const int64_t total = len * c_pad;
    const float* src = x + (int64_t)u * T * C;
    TO* dst = out + off * c_pad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / c_pad;
        const int c = (int)(i % c_pad);
        float v = (c < C) ? src[t * C + c] : 0.f;
        if (un_scale && c < C) {
            const float sc = un_scale[c];
            v = fabsf(sc) >= 1e-18f ? (v - un_shift[c]) / sc : 0.f;
        }
        dst[i] = (TO)v;
    }
}

// x[rows][C] fp32 -> two bf16 planes out[plane][rows_alloc][c_pad]: hi = bf16(x), lo = bf16(x - hi)
__global__ void pack_rows_split_kernel(const float* __restrict__ x, int64_t rows, int C, int c_pad,
                                       int64_t plane_elems, __bf16* __restrict__ out) {
    const int64_t total = rows * c_pad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c_pad;
        const int c = (int)(i % c_pad);
        const float v = (c < C) ? x[r * C + c] : 0.f;
        const __bf16 hi = (__bf16)v;
        out[i] = hi;
        out[plane_elems + i] = (__bf16)(v - (float)hi);
    }
}

hipError_t launch_pack_rows_split(const float* x, int64_t rows, int C, int c_pad, int64_t plane_elems, void* out,
                                  hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    pack_rows_split_kernel<<<2048, 256, 0, s>>>(x, rows, C, c_pad, plane_elems, static_cast<__bf16*>(out));
    return hipGetLastError();
}

hipError_t launch_pack_rows(const float* x, const int64_t* offsets, int B, int T, int C, int c_pad,
                            void* out, bool out_bf16, hipStream_t s, const float* un_scale, const float* un_shift) {
    if (B <= 0) return hipSuccess;
    const int64_t per = (int64_t)T * c_pad;
    int64_t gx = (per + 255) / 256;
    const int64_t cap = B >= 32 ? 64 : 2048 / B;   // enough blocks to fill the chip for small B
    if (gx > cap) gx = cap;
    if (out_bf16)
        pack_rows_kernel<__bf16><<<dim3((unsigned)gx, B), 256, 0, s>>>(x, offsets, T, C, c_pad, static_cast<__bf16*>(out), un_scale, un_shift);
    else
        pack_rows_kernel<float><<<dim3((unsigned)gx, B), 256, 0, s>>>(x, offsets, T, C, c_pad, static_cast<float*>(out), un_scale, un_shift);
    return hipGetLastError();
}

// flat[u*T_in + t][ld] -> y[u][t][0..C) for t < T_out
// scale / shift (or nullptr): this layer's folded BatchNorm, applied here in fp32 (plain bf16 stores relu outputs)
template <typename TI>
__global__ void unpack_rows_kernel(const TI* __restrict__ flat, int ld, int T_in, int T_out, int C,
                                   float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift) {
    const int u = blockIdx.y;
    const int64_t total = (int64_t)T_out * C;
    const TI* src = flat + (int64_t)u * T_in * ld;
    float* dst = y + (int64_t)u * total;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / C;
        const int c = (int)(i % C);
        const float v = (float)src[t * ld + c];
        dst[i] = scale ? fmaf(v, scale[c], shift[c]) : v;
    }
}

// the same from the two bf16 planes of a bf16x3 activation buffer: y = hi + lo
__global__ void unpack_rows_split_kernel(const __bf16* __restrict__ flat, int64_t plane_elems, int ld, int T_in, int T_out,
                                         int C, float* __restrict__ y) {
    const int u = blockIdx.y;
    const int64_t total = (int64_t)T_out * C;
    const __bf16* src = flat + (int64_t)u * T_in * ld;
    float* dst = y + (int64_t)u * total;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / C;
        const int c = (int)(i % C);
        dst[i] = (float)src[t * ld + c] + (float)src[plane_elems + t * ld + c];
    }
}
hipError_t launch_unpack_rows_split(const void* flat, int64_t plane_elems, int ld, int B, int T_in, int T_out, int C,
                                    float* y, hipStream_t s) {
    if (B <= 0 || T_out <= 0) return hipSuccess;
    const int64_t per = (int64_t)T_out * C;
    int gx = (int)((per + 255) / 256);
    if (gx > 64) gx = 64;
    unpack_rows_split_kernel<<<dim3((unsigned)gx, B), 256, 0, s>>>(static_cast<const __bf16*>(flat), plane_elems, ld, T_in,
                                                                  T_out, C, y);
    return hipGetLastError();
}

hipError_t launch_unpack_rows(const void* flat, bool in_bf16, int ld, int B, int T_in, int T_out, int C, float* y,
                              hipStream_t s, const float* scale, const float* shift) {
    if (B <= 0 || T_out <= 0) return hipSuccess;
    const int64_t per = (int64_t)T_out * C;
    int gx = (int)((per + 255) / 256);
    if (gx > 64) gx = 64;
    if (in_bf16)
        unpack_rows_kernel<__bf16><<<dim3((unsigned)gx, B), 256, 0, s>>>(static_cast<const __bf16*>(flat), ld, T_in, T_out, C, y, scale, shift);
    else
        unpack_rows_kernel<float><<<dim3((unsigned)gx, B), 256, 0, s>>>(static_cast<const float*>(flat), ld, T_in, T_out, C, y, scale, shift);
    return hipGetLastError();
}

}  // namespace xvec

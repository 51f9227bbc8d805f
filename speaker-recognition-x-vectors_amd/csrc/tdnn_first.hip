// Layer 1 of the bf16 path (reference tdnn_layer.py:26-60 with context [-2..2] on 24 MFCCs): K = 5 x 24 = 120
// inputs per frame, 512 outputs -- 0.3 % of the path's flops and 77 MB of bf16 output per 256 x 300-frame batch.
// It is a streaming kernel, not a GEMM tile problem: in the 128x128 kernel (tdnn_layer.hip) a tile's K loop is
// two chunks long and a tile costs 8.4k cycles for 1k cycles of MFMA -- 31-33 us, where a plain fill of the same
// 77.6 MB takes 15-17 us (profiles/diag/src/write_bw.hip).
//
// Here the weights never move: a wave keeps its block of the 512 x 128 bf16 weight matrix in registers for the whole
// launch, a block covers the 512 channels of one 32-frame group at a time.  The group's input -- 32 windows of 120
// consecutive floats, 96 B apart -- is fetched by the block's threads together, rounded to bf16 and parked in an LDS tile
// (two tiles: the next group is fetched into registers while this one is multiplied; one barrier per group).  No per-tile
// descriptors, no tile switch; the utterance bookkeeping (input row = output row + 4 x utterance index) is two scalar
// compares per group on the fixed-length path.  What cost time in the first version was every wave waiting for its stores
// to be acknowledged at each group's barrier (XF_LDS_BARRIER).  Plain bf16: tdnn_first_kernel (round 4's form is described
// above it); bf16x3: first3::tdnn_first3_kernel.
//
// Shapes: n_pad == 512, one folded tap with k_pad == 128 (input_size * 5 <= 126 for plain bf16, rows contiguous: ldx ==
// input_size) and 16-byte aligned rows; run_tdnn falls back to the 128x128 kernel otherwise.
#include "tdnn_common.h"

namespace xvec {
namespace first {

constexpr int kRowB = 272;                 // LDS bytes per frame: 128 bf16 + 16 (conflict-free 16-byte fragment reads)
constexpr int kTileB = 32 * kRowB;
constexpr int kConstFloats = 3 * 512;      // bias | scale | shift

__device__ __forceinline__ u32x4 cvt8(const u32x4& lo, const u32x4& hi) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    u32x4 o;
    o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{__uint_as_float(lo[0]), __uint_as_float(lo[1])}, bf16x2));
    o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{__uint_as_float(lo[2]), __uint_as_float(lo[3])}, bf16x2));
    o[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{__uint_as_float(hi[0]), __uint_as_float(hi[1])}, bf16x2));
    o[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{__uint_as_float(hi[2]), __uint_as_float(hi[3])}, bf16x2));
    return o;
}

template <bool RAGGED>
__device__ __forceinline__ int64_t first_row_of(const RowMap& m, int u) {
    u = __builtin_amdgcn_readfirstlane(u);
    if (RAGGED) return sload_i64(m.offsets + u) - (int64_t)u * m.cum;
    return (int64_t)u * (m.fixed_T - m.cum);
}

// utterance cursor of a block: u = utterance of the group's first row, end = first output row of utterance u+1
struct Cur {
    int u;
    int64_t end;
};

// ---- round 4: one block of EIGHT waves per CU, 64 channels per wave, v_mfma_f32_16x16x32_bf16 -------------------------------
// What changed against round 3's form (two blocks of four waves x 128 channels, 32x32x16) and why:
//   * start-up.  Every wave fetched its 32 KB of weight fragments: 512 blocks x 4 waves x 32 KB = 64 MB of L2 reads for 128 KB
//     of weights, a third of the kernel.  Eight waves of 64 channels hold 16 KB each and a CU has one block: 32 MB.
//   * the bias rides in the two spare k slots of the padded K = 128 (k = kpt, kpt + 1; the reference's 5 x 24 = 120): the staged
//     input carries 1.0 there and the weight copy of this kernel (xvec_api.hip, refold: Wp16b) bf16(bias) and
//     bf16(bias - bf16(bias)) -- the sum is the bias to 2^-17, the products are exact -- so the epilogue has no add;
//   * the accumulator layout of tdnn_pp16.hip: frames on the registers and lane quads, the channel on the lane, a lane's four
//     accumulators of a frame on four ADJACENT channels (which column of W a lane multiplies is only a matter of which fragment
//     bytes it loaded): two v_cvt_pk_bf16_f32, the ReLU on the packed pairs, one 8-byte store = whole 128-byte row segments of
//     four frames per instruction; 8 store instructions per group and wave (round 3: 32 of 4 bytes).
// The BatchNorm behind the ReLU (tdnn_layer.py:36-39) is deferred into layer 2's weights (xvec_api.hip, refold).
constexpr int kRowB16 = 288;               // LDS bytes per frame: 128 bf16 + 32 (72 dwords: the ds_read_b128 A fragments of the sixteen
constexpr int kTileB16 = 32 * kRowB16;     // lanes of a service group fall on sixteen distinct 4-bank windows)

struct Staged8 {
    u32x4 q0, q1;                          // 8 floats of the caller's fp32 rows
};

template <bool RAGGED>
__device__ __forceinline__ void fetch8(const TdnnArgs& a, int64_t g, Cur& cu, int rr, int sk, Staged8& st) {
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
    // descriptor at the group's first input row (64-bit), bounded by the end of the caller's tensor: reads past it return
    // zeros; the lane offset is small (a group's rows + the utterances it skips)
    const int64_t row0 = m0 + (int64_t)cu.u * a.span;
    const int64_t total = a.x_bytes ? a.x_bytes : a.x_rows * (int64_t)a.ldx * 4;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc_bounded(a.X, row0 * a.ldx * 4, total);
    const int voff = ((rr + c * a.span) * a.ldx + 8 * sk) * 4;
    st.q0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 0, 0));
    st.q1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 16, 0));
}

// round to bf16, blank the K tail (values past kpt belong to the next frame and meet zero weights, but 0 x Inf is not 0), put
// 1.0 | 1.0 into the bias slots k = kpt, kpt + 1 (one aligned dword: kpt is a multiple of 4) and park the 16 bytes
__device__ __forceinline__ void park8(const TdnnArgs& a, char* tile, int rr, int sk, const Staged8& st) {
    u32x4 o = cvt8(st.q0, st.q1);
    const int k0 = 8 * sk;
    if (k0 + 8 > a.kpt) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int k = k0 + 2 * d;
            if (k == a.kpt) o[d] = 0x3f803f80u;
            else if (k > a.kpt) o[d] = 0u;
        }
    }
    *reinterpret_cast<u32x4*>(tile + rr * kRowB16 + sk * 16) = o;
}

template <bool RAGGED>
__global__ __launch_bounds__(512) void tdnn_first_kernel(const TdnnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * kTileB16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably uniform: scalar fragment offsets
    const int c = lane & 15, kq = lane >> 4;
    // this wave's weights: channels [64*wave, +64).  Accumulator (frame block, cb), lane c <-> channel 64*wave + 4*c + cb.  B operand
    // of k-step s (32 wide): lane (c, kq) holds k = 32 s + 8 kq ..+7.  The fragment-major packing (pack.hip) keeps
    // W[32*ct + l][16*ks + 8*h ..+7] at ((ct*8 + ks)*64 + l + 32*h)*16 B
    u32x4 wf[4][4];
    {
        const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.Wf);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int n = 64 * wave + 4 * c + cb;
            const int voff = ((n >> 5) * 8 * 64 + (n & 31) + 32 * (kq & 1)) * 16 + (kq >> 1) * 1024;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                wf[cb][s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, 2 * s * 1024, 0));
        }
    }
    // contiguous range of 32-frame groups per block
    const int64_t g_begin = a.groups_total * (int64_t)blockIdx.x / gridDim.x;
    const int64_t g_end = a.groups_total * (int64_t)(blockIdx.x + 1) / gridDim.x;
    if (g_begin >= g_end) return;
    Cur cu;
    cu.u = __builtin_amdgcn_readfirstlane(utt_of_row(a.out_map, g_begin * 32));
    cu.end = first_row_of<RAGGED>(a.out_map, cu.u + 1);
    const int rr = tid >> 4, sk = tid & 15;        // staging: frame rr, half k-step sk (8 floats)
    // Two groups of look-ahead: loads and stores share the wave's in-order vmcnt counter, so waiting for loads
    // issued AFTER a group's stores would wait for those stores to be acknowledged by memory (several us at
    // full write rate).  The loads of group g+2 are issued before the stores of group g, and the wait for group
    // g+1's loads leaves the newer operations in flight.
    Staged8 sa, sb;
    fetch8<RAGGED>(a, g_begin, cu, rr, sk, sa);
    park8(a, smem, rr, sk, sa);
    if (g_begin + 1 < g_end) fetch8<RAGGED>(a, g_begin + 1, cu, rr, sk, sb);
    __syncthreads();                               // tile 0 visible

    const char* frag = smem + c * kRowB16 + 16 * kq;   // A operand of lane (c, kq): frame 16 fb + c, k = 32 s + 8 kq ..+7
    // accumulator register e of lane (c, q = kq): frame 16 fb + 4 q + e; the lane's 8 bytes sit at column 64*wave + 4c
    const int y_voff = (4 * kq * a.ldy + 64 * wave + 4 * c) * 2;
    // __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: it would hold every wave until its stores
    // of the group are acknowledged by memory.  Only the LDS traffic has to be ordered here.
#define XF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#define XF_MF(x_, w_, acc_) acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x_), __builtin_bit_cast(bf16x8, w_), acc_, 0, 0, 0);
    // one group: MFMAs on LDS tile `buf_`, park the staged group g+1 (ST_PARK) into the other tile, fetch group
    // g+2 into the set that was parked last time (ST_FETCH), epilogue + stores
#define XF_GROUP(g_, buf_, ST_PARK, ST_FETCH)                                                                     \
    {                                                                                                             \
        if ((g_) + 2 < g_end) fetch8<RAGGED>(a, (g_) + 2, cu, rr, sk, ST_FETCH);                                  \
        f32x4 acc[2][4];                                                                                          \
        _Pragma("unroll") for (int fb = 0; fb < 2; ++fb)                                                          \
            _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) acc[fb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};             \
        const char* tile = frag + (buf_) * kTileB16;                                                              \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                           \
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(tile + s * 64);                                      \
            const u32x4 x1 = *reinterpret_cast<const u32x4*>(tile + 16 * kRowB16 + s * 64);                       \
            _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                    \
                XF_MF(x0, wf[cb][s], acc[0][cb]) XF_MF(x1, wf[cb][s], acc[1][cb])                                 \
            }                                                                                                     \
        }                                                                                                         \
        if ((g_) + 1 < g_end) park8(a, smem + ((buf_) ^ 1) * kTileB16, rr, sk, ST_PARK);                          \
        /* ReLU (tdnn_layer.py:31; bias inside the products, BatchNorm deferred); rows of the group at g*32 (the */ \
        /* row buffer is padded past the last valid frame) */                                                     \
        const __amdgpu_buffer_rsrc_t yr = make_rsrc(static_cast<char*>(a.Y) + (g_) * 32 * (int64_t)a.ldy * 2);    \
        _Pragma("unroll") for (int fb = 0; fb < 2; ++fb)                                                          \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                       \
                u32x2v pk = {__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc[fb][0][e], acc[fb][1][e]}, bf16x2)), \
                             __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc[fb][2][e], acc[fb][3][e]}, bf16x2))}; \
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk[0]) : "v"(pk[0]));   /* ReLU of the packed pair (tdnn_pp16.hip, relu_pk_bf16) */ \
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk[1]) : "v"(pk[1]));                                         \
                __builtin_amdgcn_raw_buffer_store_b64(pk, yr, y_voff, (16 * fb + e) * a.ldy * 2, 0);              \
            }                                                                                                     \
        XF_LDS_BARRIER() /* the other tile is written, this one read by every wave */                            \
    }
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    for (int64_t g = g_begin; g < g_end; g += 2) {
        XF_GROUP(g, 0, sb, sa)
        if (g + 1 < g_end) XF_GROUP(g + 1, 1, sa, sb)
    }
#undef XF_GROUP
#undef XF_MF
#undef XF_LDS_BARRIER
}

}  // namespace first

// ---- bf16x3 (fp32 values as hi + lo bf16 planes): the same streaming kernel with both halves of the weights
// in registers.  W_hi and W_lo of 128 channels would be 256 VGPRs, so a block is EIGHT waves of 64 channels (2 x 8
// fragments of each half: 128 VGPRs), one block per CU; the 512 threads fetch half a k-step (8 floats) of one frame each,
// split it into bf16(x) and bf16(x - bf16(x)) and park the two halves in two LDS tiles; per k-step and accumulator three
// MFMAs (x_hi W_hi, x_hi W_lo, x_lo W_hi); the epilogue stores bf16(v) and bf16(v - bf16(v)) to the two planes of the
// activation buffer (y_plane_bytes apart).  Reads the caller's fp32 rows: the hi/lo split pass of the MFCCs
// (pack_rows_split, 8 us per 256 x 300 frames) is not needed either.  128x128 form of this layer: 52 us; this: see DESIGN.
namespace first3 {

using first::Cur;
using first::first_row_of;
using first::kConstFloats;
using first::kRowB;
using first::kTileB;

struct Staged {
    u32x4 q0, q1;                          // 8 floats of the caller's fp32 rows
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
    st.q0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 0, 0));
    st.q1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 16, 0));
}

// split into bf16 hi and lo, blank the K tail, park 16 bytes in each tile
__device__ __forceinline__ void park(const TdnnArgs& a, char* tile_hi, int rr, int sk, const Staged& st) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    float f[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        f[d] = __uint_as_float(st.q0[d]);
        f[4 + d] = __uint_as_float(st.q1[d]);
    }
    const int k0 = 8 * sk;
    if (k0 + 8 > a.kpt) {                  // values past kpt belong to the next frame and meet zero weights, but 0 x Inf is not 0
#pragma unroll
        for (int d = 0; d < 8; ++d)
            if (k0 + d >= a.kpt) f[d] = 0.f;
    }
    u32x4 hi, lo;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        hi[d] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{f[2 * d], f[2 * d + 1]}, bf16x2));
        const float l0 = f[2 * d] - __uint_as_float(hi[d] << 16), l1 = f[2 * d + 1] - __uint_as_float(hi[d] & 0xffff0000u);
        lo[d] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{l0, l1}, bf16x2));
    }
    *reinterpret_cast<u32x4*>(tile_hi + rr * kRowB + sk * 16) = hi;
    *reinterpret_cast<u32x4*>(tile_hi + kTileB + rr * kRowB + sk * 16) = lo;
}

template <bool RAGGED>
__global__ __launch_bounds__(512) void tdnn_first3_kernel(const TdnnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[4 * kTileB + kConstFloats * 4];     // [buffer][hi | lo] tiles, constants
    float* cst = reinterpret_cast<float*>(smem + 4 * kTileB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    cst[tid] = a.bias[tid];
    cst[512 + tid] = a.scale[tid];
    cst[1024 + tid] = a.shift[tid];
    // this wave's weights: channels [64*wave, +64); accumulator cg, lane r <-> channel 64*wave + 2r + cg.  The bf16x3
    // fragment stream (pack.hip, terms == 2) holds per 32-channel tile and 64-wide chunk four W_hi k-step blocks, then four W_lo
    u32x4 wh[2][8], wl[2][8];
    {
        const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.Wf);
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const int ch = 64 * wave + 2 * r + cg;
            const int voff = ((ch >> 5) * 16 * 64 + (ch & 31) + 32 * h) * 16;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int blk = (ks >> 2) * 8 + (ks & 3);
                wh[cg][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, blk * 1024, 0));
                wl[cg][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, (blk + 4) * 1024, 0));
            }
        }
    }
    const int64_t g_begin = a.groups_total * (int64_t)blockIdx.x / gridDim.x;
    const int64_t g_end = a.groups_total * (int64_t)(blockIdx.x + 1) / gridDim.x;
    if (g_begin >= g_end) return;
    Cur cu;
    cu.u = __builtin_amdgcn_readfirstlane(utt_of_row(a.out_map, g_begin * 32));
    cu.end = first_row_of<RAGGED>(a.out_map, cu.u + 1);
    const int rr = tid >> 4, sk = tid & 15;        // staging: frame rr, half k-step sk
    Staged sa, sb;                                 // two groups of look-ahead (see tdnn_first_kernel)
    fetch<RAGGED>(a, g_begin, cu, rr, sk, sa);
    park(a, smem, rr, sk, sa);
    if (g_begin + 1 < g_end) fetch<RAGGED>(a, g_begin + 1, cu, rr, sk, sb);
    __syncthreads();

    const char* frag = smem + r * kRowB + 16 * h;
    const float* c0 = cst + 64 * wave + 2 * r;
    const float2 bi = *reinterpret_cast<const float2*>(c0), sc = *reinterpret_cast<const float2*>(c0 + 512),
                 sh = *reinterpret_cast<const float2*>(c0 + 1024);
    const int y_voff = (4 * h * a.ldy + 64 * wave + 2 * r) * 2;
    typedef float f32x2v __attribute__((ext_vector_type(2)));
#define XF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#define XF3_MF(x_, w_, acc_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x_), __builtin_bit_cast(bf16x8, w_), acc_, 0, 0, 0);
#define XF3_GROUP(g_, buf_, ST_PARK, ST_FETCH)                                                                    \
    {                                                                                                             \
        if ((g_) + 2 < g_end) fetch<RAGGED>(a, (g_) + 2, cu, rr, sk, ST_FETCH);                                   \
        f32x16 acc0, acc1;                                                                                        \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;                                   \
        const char* tile = frag + (buf_) * 2 * kTileB;                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                        \
            const u32x4 xh = *reinterpret_cast<const u32x4*>(tile + ks * 32);                                     \
            const u32x4 xl = *reinterpret_cast<const u32x4*>(tile + kTileB + ks * 32);                            \
            XF3_MF(xh, wh[0][ks], acc0) XF3_MF(xh, wh[1][ks], acc1)                                               \
            XF3_MF(xh, wl[0][ks], acc0) XF3_MF(xh, wl[1][ks], acc1)                                               \
            XF3_MF(xl, wh[0][ks], acc0) XF3_MF(xl, wh[1][ks], acc1)                                               \
        }                                                                                                         \
        if ((g_) + 1 < g_end) park(a, smem + ((buf_) ^ 1) * 2 * kTileB, rr, sk, ST_PARK);                         \
        /* bias + ReLU + folded BatchNorm (tdnn_layer.py:30-39), then the two planes */                           \
        const __amdgpu_buffer_rsrc_t yr = make_rsrc(static_cast<char*>(a.Y) + (g_) * 32 * (int64_t)a.ldy * 2);    \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                          \
            const float v0 = fmaf(fmaxf(acc0[e] + bi.x, 0.f), sc.x, sh.x);                                        \
            const float v1 = fmaf(fmaxf(acc1[e] + bi.y, 0.f), sc.y, sh.y);                                        \
            const unsigned ph = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v0, v1}, bf16x2));    \
            const float l0 = v0 - __uint_as_float(ph << 16), l1 = v1 - __uint_as_float(ph & 0xffff0000u);        \
            const unsigned pl = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{l0, l1}, bf16x2));    \
            const int so = ((e & 3) + 8 * (e >> 2)) * a.ldy * 2;                                                  \
            __builtin_amdgcn_raw_buffer_store_b32(ph, yr, y_voff, so, 0);                                         \
            __builtin_amdgcn_raw_buffer_store_b32(pl, yr, y_voff, so + a.y_plane_bytes, 0);                       \
        }                                                                                                         \
        XF_LDS_BARRIER()                                                                                          \
    }
    for (int64_t g = g_begin; g < g_end; g += 2) {
        XF3_GROUP(g, 0, sb, sa)
        if (g + 1 < g_end) XF3_GROUP(g + 1, 1, sa, sb)
    }
#undef XF3_GROUP
#undef XF3_MF
#undef XF_LDS_BARRIER
}

}  // namespace first3

bool tdnn_first3_applicable(const TdnnArgs& a) {
    return a.ldy == 512 && a.k_pad == 256 && a.n_taps == 1 && a.kpt <= 128 && a.terms == 2 && a.y_plane_bytes > 0 &&
           (a.ldx * 4) % 16 == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0 && a.groups_total > 0;
}

hipError_t launch_tdnn_first3(const TdnnArgs& a, int num_cu, hipStream_t s) {
    const int grid = (int)(a.groups_total < num_cu ? a.groups_total : num_cu);
    if (a.out_map.offsets != nullptr) first3::tdnn_first3_kernel<true><<<grid, 512, 0, s>>>(a);
    else first3::tdnn_first3_kernel<false><<<grid, 512, 0, s>>>(a);
    return hipGetLastError();
}

bool tdnn_first_applicable(const TdnnArgs& a) {
    // (kpt + 2 <= 128: two spare k slots carry the bias; kpt a multiple of 4 follows from the aligned, contiguous rows)
    return a.ldy == 512 && a.k_pad == 128 && a.n_taps == 1 && a.kpt <= 126 && a.kpt % 4 == 0 && a.terms == 1 &&
           (a.ldx * 4) % 16 == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0 && a.groups_total > 0;
}

hipError_t launch_tdnn_first(const TdnnArgs& a, int num_cu, hipStream_t s) {
    const int grid = (int)(a.groups_total < num_cu ? a.groups_total : num_cu);       // one block of eight waves per CU
    if (a.out_map.offsets != nullptr) first::tdnn_first_kernel<true><<<grid, 512, 0, s>>>(a);
    else first::tdnn_first_kernel<false><<<grid, 512, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace xvec

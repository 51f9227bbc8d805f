// Segment-level affine layers (reference main.py:45-47,72-75,87-90):
//   y[M,N] = act( x[M,K] . W[N,K]^T + b ),  W in PyTorch nn.Linear layout, exact fp32.
// M is the number of utterances (<= a few hundred), so this is a skinny GEMM whose problem is
// filling the chip: one 16x16 output tile per 512-thread block (M=256, N=512: 512 blocks, two per
// CU), the block's eight waves split K on v_mfma_f32_16x16x4_f32 and combine through LDS.  (With
// 32x32 tiles only 128 blocks existed: half the CUs idle, two waves' MFMA chains per SIMD, 28 us
// for segment_layer6; now the chains are a quarter as long and every CU has two blocks.)
// Operands are read once per block straight into registers (16 B per lane, 64 contiguous bytes per
// row and instruction); they are small enough to stay in L2.
#include "xvec_internal.h"
#include "tdnn_common.h"
#include <algorithm>

namespace xvec {

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* row, int k, int K) {
    if (VEC) {
        return (k < K) ? *reinterpret_cast<const float4*>(row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float4 v;
        v.x = (k + 0 < K) ? row[k + 0] : 0.f;
        v.y = (k + 1 < K) ? row[k + 1] : 0.f;
        v.z = (k + 2 < K) ? row[k + 2] : 0.f;
        v.w = (k + 3 < K) ? row[k + 3] : 0.f;
        return v;
    }
}

constexpr int kAffWaves = 8;

// the four k-quads of one 16-byte operand pair (lane quarter kq owns k = 16g+4kq .. +3 of k-group
// g, for A and B alike, so element t of every lane belongs to MFMA t of the group); two
// accumulators alternate: the dependent latency of this MFMA (40 cycles) exceeds its issue interval
#define AFF_MFMA4(a_, b_)                                                        \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.x, b_.x, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.y, b_.y, acc1, 0, 0, 0);      \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.z, b_.z, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.w, b_.w, acc1, 0, 0, 0);

template <bool VEC>
__global__ __launch_bounds__(64 * kAffWaves) void affine_f32_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ W,
                                                                    const float* __restrict__ b,
                                                                    float* __restrict__ y, int M, int N, int K,
                                                                    int relu) {
    __shared__ float red[kAffWaves][16 * 17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int m = min(m0 + i, M - 1), n = min(n0 + i, N - 1);   // clamped rows: results discarded
    const float* xa = x + (int64_t)m * K;
    const float* wb = W + (int64_t)n * K;

    f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

    // A trip of a wave is four consecutive k-groups (64 k = 256 contiguous bytes of every row); the
    // block's waves take interleaved trips (ld4 returns zeros past K).  The loads of trip t+1 are
    // issued before the MFMAs of trip t.
    const int trips = (K + 63) / 64;
    constexpr int S = kAffWaves;
    float4 a0, a1, a2, a3, b0, b1, b2, b3;
#define AFF_LOAD(t_)                                                                            \
    {                                                                                           \
        const int k0 = 64 * (t_) + 4 * kq;                                                      \
        a0 = ld4<VEC>(xa, k0, K); b0 = ld4<VEC>(wb, k0, K);                                     \
        a1 = ld4<VEC>(xa, k0 + 16, K); b1 = ld4<VEC>(wb, k0 + 16, K);                           \
        a2 = ld4<VEC>(xa, k0 + 32, K); b2 = ld4<VEC>(wb, k0 + 32, K);                           \
        a3 = ld4<VEC>(xa, k0 + 48, K); b3 = ld4<VEC>(wb, k0 + 48, K);                           \
    }
    AFF_LOAD(wave)
    for (int t = wave; t < trips; t += S) {
        const float4 c0 = a0, c1 = a1, c2 = a2, c3 = a3, d0 = b0, d1 = b1, d2 = b2, d3 = b3;
        if (t + S < trips) AFF_LOAD(t + S)
        AFF_MFMA4(c0, d0) AFF_MFMA4(c1, d1) AFF_MFMA4(c2, d2) AFF_MFMA4(c3, d3)
    }
#undef AFF_LOAD

    // accumulator element e of lane (i, kq): row = 4*kq + e, col = i
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][(4 * kq + e) * 17 + i] = acc0[e] + acc1[e];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
        if (m0 + row < M && n0 + col < N) {
            float v = b[n0 + col];
#pragma unroll
            for (int w = 0; w < kAffWaves; ++w) v += red[w][row * 17 + col];
            if (relu) v = fmaxf(v, 0.f);
            y[(int64_t)(m0 + row) * N + n0 + col] = v;
        }
    }
}
#undef AFF_MFMA4

// ---- split-K form (the forward path, which has scratch memory to offer) -----------------------------------------
// The kernel above reads a 16-row strip of x and of W per 16x16 tile: 4 flop per operand byte, 196 MB of L2
// reads for segment_layer6 at M=256 (x is 3 MB, W 6 MB) -- 21 us, bound by those reads.  Here a block owns a
// 64x64 tile (16 flop per byte, 48 MB) and a K range: M=256, N=512, K=3000 -> 4 x 8 tiles x 16 ranges = 512
// blocks of four waves, each wave a 32x32 quadrant on v_mfma_f32_32x32x2_f32 (13 us + 4.6 us for the reduction;
// 8 to 16 ranges measure the same, 4 ranges 26 us: profiles/diag/aff_sweep.py).  The operand tiles go through LDS:
// 16 lanes fetch 256 contiguous bytes of a row (a lane-per-row fetch straight into the fragment layout touches
// 32 cache lines per instruction for 32 useful bytes each -- that version took 20 us), rows padded to 68 floats so
// that the 16-byte fragment reads are conflict-free.  Partial tiles go to scratch [S][M][N] and are summed in
// range order by affine_reduce_kernel (deterministic; bias and ReLU there).  S == 1 (many utterances: the tiles
// alone fill the chip) writes y directly.
typedef float f32x16v __attribute__((ext_vector_type(16)));
constexpr int kAfsLd = 68;                       // LDS row stride in floats (64 + 4)
constexpr int kAfsTile = 64 * kAfsLd;            // one operand tile

template <bool DIRECT>
__global__ __launch_bounds__(256, 2) void affine_splitk_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                               const float* __restrict__ b, float* __restrict__ out, int M,
                                                               int N, int K, int relu, int trips_per_split, int tn,
                                                               int tiles, int S) {
    __shared__ __attribute__((aligned(16))) float lds[2 * kAfsTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    // Workgroups go to the eight XCDs round-robin and every XCD has its own L2: with the K ranges dealt out the
    // same way (range s on XCD s % 8) an XCD fetches only ITS eighth of x and W from HBM.  With the tiles spread
    // over the XCDs instead every L2 pulled all 9 MB in -- 72 MB over the fabric, 14 us for a 5 us kernel.
    int split = 0, tile = blockIdx.x;
    if (!DIRECT) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        split = xcd + 8 * (idx / tiles);
        tile = idx % tiles;
        if (split >= S) return;
    }
    const int bm0 = (tile / tn) * 64, bn0 = (tile % tn) * 64;
    const int trips = (K + 63) / 64;
    const int t_lo = split * trips_per_split, t_hi = min(trips, t_lo + trips_per_split);

    // staging map: 16 lanes per row (16 x 16 B = the trip's 64 floats), 16 rows per pass, 4 passes per operand
    const int s_row = tid >> 4, s_c = (tid & 15) * 4;
    const float* xs[4];
    const float* ws[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {                // clamped rows: results discarded
        xs[p] = x + (int64_t)min(bm0 + s_row + 16 * p, M - 1) * K;
        ws[p] = W + (int64_t)min(bn0 + s_row + 16 * p, N - 1) * K;
    }
    // two register sets: the loads of trips t+1 and t+2 are in flight while trip t is multiplied (a block has only
    // a few trips, so the memory latency is paid about once instead of once per trip)
    float4 ra0[4], rb0[4], ra1[4], rb1[4];
#define AFS_LOAD(S_, t_)                                         \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {              \
        ra##S_[p] = ld4<true>(xs[p], 64 * (t_) + s_c, K);        \
        rb##S_[p] = ld4<true>(ws[p], 64 * (t_) + s_c, K);        \
    }
#define AFS_TRIP(S_, t_)                                                                           \
    {                                                                                              \
        __syncthreads(); /* the previous trip's fragment reads are done */                         \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                            \
            *reinterpret_cast<float4*>(lds + (s_row + 16 * p) * kAfsLd + s_c) = ra##S_[p];         \
            *reinterpret_cast<float4*>(lds + kAfsTile + (s_row + 16 * p) * kAfsLd + s_c) = rb##S_[p]; \
        }                                                                                          \
        __syncthreads();                                                                           \
        if ((t_) + 2 < t_hi) AFS_LOAD(S_, (t_) + 2)                                                \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) { /* lane half kh owns k = 8g + 4kh .. +3 */  \
            const float4 c = *reinterpret_cast<const float4*>(fa + 8 * g);                         \
            const float4 d = *reinterpret_cast<const float4*>(fb + 8 * g);                         \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, d.x, acc0, 0, 0, 0);                  \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, d.y, acc1, 0, 0, 0);                  \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, d.z, acc0, 0, 0, 0);                  \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, d.w, acc1, 0, 0, 0);                  \
        }                                                                                          \
    }
    f32x16v acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
    const float* fa = lds + (32 * (wave >> 1) + i) * kAfsLd + 4 * kh;
    const float* fb = lds + kAfsTile + (32 * (wave & 1) + i) * kAfsLd + 4 * kh;
    if (t_lo < t_hi) AFS_LOAD(0, t_lo)
    if (t_lo + 1 < t_hi) AFS_LOAD(1, t_lo + 1)
    for (int t = t_lo; t < t_hi; t += 2) {
        AFS_TRIP(0, t)
        if (t + 1 < t_hi) AFS_TRIP(1, t + 1)
    }
#undef AFS_TRIP
#undef AFS_LOAD
    // accumulator element e of lane (i, kh): row = (e&3) + 8*(e>>2) + 4*kh, col = i
    const int m0 = bm0 + 32 * (wave >> 1), col = bn0 + 32 * (wave & 1) + i;
    if (col < N) {
        float* dst = DIRECT ? out : out + (int64_t)split * M * N;
        const float bias = DIRECT ? b[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (row < M) {
                float v = acc0[e] + acc1[e] + bias;
                if (DIRECT && relu) v = fmaxf(v, 0.f);
                dst[(int64_t)row * N + col] = v;
            }
        }
    }
}

// ---- round 4: bf16x3 arithmetic for the segment layers of XVEC_BF16 ------------------------------------------------
// XVEC_BF16 runs the frame-level stack at bf16 matrix rates and then spent 13 us in the fp32 split-K GEMM above, 5 of
// them MFMA time (v_mfma_f32_32x32x2_f32: 256 flop per cycle and CU; measured by knocking 7/8 of the MFMAs out).
// Here every fp32 operand is two bf16 (hi = bf16(v), lo = bf16(v - hi)) and a product is x_lo*W_hi + x_hi*W_lo +
// x_hi*W_hi on v_mfma_f32_16x16x32_bf16, as the frame-level layers of XVEC_BF16X3 (every product exact in the fp32
// accumulator, what is dropped is 2^-16 relative: 6e-6 of the fp64 result, test_segment_layers_inside_the_path;
// XVEC_BF16X3, which promises the fp32 bar end to end, keeps the fp32 kernel).  W comes pre-split
// (split_pairs_kernel: the 16 bytes of four consecutive k hold hi01 hi23 lo01 lo23, so the row stride and the
// 256-byte coalesced staging loads are those of the fp32 matrix); x is split while it is staged.  LDS: four planes
// (x hi, x lo, W hi, W lo) of 64 rows x 64 k bf16, 16-byte chunk c of row r at chunk c ^ (r & 7): the fragment reads
// (ds_read_b128) and the staging writes (8 bytes per plane) are conflict-free by the guide's lane-group rules
// (simulated, every access).  13.2 -> 10.5 us; the second kernel stays.  (Tried and dropped: the tile's last block
// adding the K ranges up itself -- its ranges sit behind eight different L2s, and the agent-scope fences that make
// the partials visible, buffer_wbl2 sc1 / buffer_inv sc1, walk the whole L2 once per block: 75 us.)
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// W[N][K] fp32 -> the same [N][K] dwords, every four consecutive k as hi01 | hi23 | lo01 | lo23 (K % 4 == 0)
__global__ __launch_bounds__(256) void split_pairs_kernel(const float4* __restrict__ W, uint4* __restrict__ out, int64_t n4) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n4) return;
    const float4 v = W[q];
    uint4 o;
    o.x = pk_bf16(v.x, v.y);
    o.y = pk_bf16(v.z, v.w);
    o.z = pk_bf16(v.x - __uint_as_float(o.x << 16), v.y - __uint_as_float(o.x & 0xffff0000u));
    o.w = pk_bf16(v.z - __uint_as_float(o.y << 16), v.w - __uint_as_float(o.y & 0xffff0000u));
    out[q] = o;
}

hipError_t launch_split_pairs(const float* W, void* out, int64_t n, hipStream_t s) {
    if (n <= 0 || (n & 3)) return hipErrorInvalidValue;
    const int64_t n4 = n / 4;
    split_pairs_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(reinterpret_cast<const float4*>(W),
                                                                     static_cast<uint4*>(out), n4);
    return hipGetLastError();
}

constexpr int kAx3Plane = 64 * 128;                // bytes: 64 rows x 64 bf16

__global__ __launch_bounds__(256, 2) void affine_splitk_x3_kernel(const float* __restrict__ x, const uint4* __restrict__ W3,
                                                                  const float* __restrict__ b, float* __restrict__ out, int M,
                                                                  int N, int K, int relu, int trips_per_split, int tn,
                                                                  int tiles, int S) {
    __shared__ __attribute__((aligned(16))) char lds[4 * kAx3Plane];     // x hi | x lo | W hi | W lo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    int split = 0, tile = blockIdx.x;
    if (S > 1) {                                    // K ranges dealt to the XCDs: see affine_splitk_kernel
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        split = xcd + 8 * (idx / tiles);
        tile = idx % tiles;
        if (split >= S) return;
    }
    const int bm0 = (tile / tn) * 64, bn0 = (tile % tn) * 64;
    const int trips = (K + 63) / 64;
    const int t_lo = split * trips_per_split, t_hi = min(trips, t_lo + trips_per_split);
    const int K4 = K >> 2;

    // staging map: 16 lanes per row (the trip's 64 k), 16 rows per pass, 4 passes per operand
    const int s_row = tid >> 4, q = tid & 15;
    const float* xs[4];
    const uint4* ws[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {                // clamped rows: results discarded
        xs[p] = x + (int64_t)min(bm0 + s_row + 16 * p, M - 1) * K;
        ws[p] = W3 + (int64_t)min(bn0 + s_row + 16 * p, N - 1) * K4;
    }
    const int st_off = s_row * 128 + (((q >> 1) ^ (s_row & 7)) << 4) + ((q & 1) << 3);     // + 16 * p rows
    float4 ra0[4], ra1[4];        // two register sets, as affine_splitk_kernel (a third measured the same)
    uint4 rb0[4], rb1[4];
#define AX3_LOAD(S_, t_)                                                                              \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                   \
        const int k = 64 * (t_) + 4 * q;                                                              \
        ra##S_[p] = ld4<true>(xs[p], k, K);                                                           \
        rb##S_[p] = (k < K) ? ws[p][k >> 2] : make_uint4(0u, 0u, 0u, 0u);                             \
    }
    f32x4v acc[2][2];
#pragma unroll
    for (int r_ = 0; r_ < 2; ++r_)
#pragma unroll
        for (int c_ = 0; c_ < 2; ++c_) acc[r_][c_] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // fragment reads: lane (i, kq) holds k = 32s + 8kq .. +7 of row i of a 16-row block
    const int fa = (32 * (wave >> 1) + i) * 128, fb = 2 * kAx3Plane + (32 * (wave & 1) + i) * 128;
    const int sw = i & 7;
#define AX3_TRIP(S_, t_)                                                                              \
    {                                                                                                 \
        __syncthreads(); /* the previous trip's fragment reads are done */                            \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                               \
            const float4 v = ra##S_[p];                                                               \
            uint2 hi, lo;                                                                             \
            hi.x = pk_bf16(v.x, v.y);                                                                 \
            hi.y = pk_bf16(v.z, v.w);                                                                 \
            lo.x = pk_bf16(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u)); \
            lo.y = pk_bf16(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u)); \
            char* d = lds + st_off + 16 * p * 128;                                                    \
            *reinterpret_cast<uint2*>(d) = hi;                                                        \
            *reinterpret_cast<uint2*>(d + kAx3Plane) = lo;                                            \
            *reinterpret_cast<uint2*>(d + 2 * kAx3Plane) = make_uint2(rb##S_[p].x, rb##S_[p].y);      \
            *reinterpret_cast<uint2*>(d + 3 * kAx3Plane) = make_uint2(rb##S_[p].z, rb##S_[p].w);      \
        }                                                                                             \
        __syncthreads();                                                                              \
        if ((t_) + 2 < t_hi) AX3_LOAD(S_, (t_) + 2)                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                            \
            const int co = ((4 * ks + kq) ^ sw) << 4;                                                 \
            bf16x8 ah[2], al[2], bh[2], bl[2];                                                        \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                        \
                ah[r_] = *reinterpret_cast<const bf16x8*>(lds + fa + r_ * 16 * 128 + co);             \
                al[r_] = *reinterpret_cast<const bf16x8*>(lds + kAx3Plane + fa + r_ * 16 * 128 + co); \
                bh[r_] = *reinterpret_cast<const bf16x8*>(lds + fb + r_ * 16 * 128 + co);             \
                bl[r_] = *reinterpret_cast<const bf16x8*>(lds + kAx3Plane + fb + r_ * 16 * 128 + co); \
            }                                                                                         \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                          \
                _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                    \
                    acc[r_][c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[r_], bh[c_], acc[r_][c_], 0, 0, 0); \
                    acc[r_][c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r_], bl[c_], acc[r_][c_], 0, 0, 0); \
                }                                                                                     \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                          \
                _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                      \
                    acc[r_][c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r_], bh[c_], acc[r_][c_], 0, 0, 0); \
        }                                                                                             \
    }
    if (t_lo < t_hi) AX3_LOAD(0, t_lo)
    if (t_lo + 1 < t_hi) AX3_LOAD(1, t_lo + 1)
    for (int t = t_lo; t < t_hi; t += 2) {
        AX3_TRIP(0, t)
        if (t + 1 < t_hi) AX3_TRIP(1, t + 1)
    }
#undef AX3_TRIP
#undef AX3_LOAD
    // accumulator element e of lane (i, kq) of block (r_, c_): row = 16 r_ + 4 kq + e, col = 16 c_ + i
    const bool direct = S == 1;
    float* dst = direct ? out : out + (int64_t)split * M * N;
#pragma unroll
    for (int c_ = 0; c_ < 2; ++c_) {
        const int col = bn0 + 32 * (wave & 1) + 16 * c_ + i;
        if (col < N) {
            const float bias = direct ? b[col] : 0.f;
#pragma unroll
            for (int r_ = 0; r_ < 2; ++r_)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = bm0 + 32 * (wave >> 1) + 16 * r_ + 4 * kq + e;
                    if (row < M) {
                        float v = acc[r_][c_][e] + bias;
                        if (direct && relu) v = fmaxf(v, 0.f);
                        dst[(int64_t)row * N + col] = v;
                    }
                }
        }
    }
}

// y = act(b + sum over the S ranges, in range order); one float4 of y per thread (N % 4 == 0)
__global__ __launch_bounds__(256) void affine_reduce_kernel(const float* __restrict__ part, const float* __restrict__ b,
                                                            float* __restrict__ y, int64_t MN, int N, int S, int relu) {
    const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (q >= MN) return;
    float4 v = *reinterpret_cast<const float4*>(b + (int)(q % N));
    int s = 0;
    if (S == 16) {
        // the usual split (segment_layer6 at the bench batch): all sixteen partials requested before the first add -- the slabs
        // were written by other CUs a moment ago, every load is a round trip to another XCD's L2 or beyond, and four at a time
        // (below) is four such trips in a row.  Same order of additions.
        float4 p[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) p[j] = *reinterpret_cast<const float4*>(part + j * MN + q);
#pragma unroll
        for (int j = 0; j < 16; ++j) { v.x += p[j].x; v.y += p[j].y; v.z += p[j].z; v.w += p[j].w; }
        s = 16;
    }
    for (; s + 4 <= S; s += 4) {                 // four independent loads in flight, summed in order
        const float4 p0 = *reinterpret_cast<const float4*>(part + (s + 0) * MN + q);
        const float4 p1 = *reinterpret_cast<const float4*>(part + (s + 1) * MN + q);
        const float4 p2 = *reinterpret_cast<const float4*>(part + (s + 2) * MN + q);
        const float4 p3 = *reinterpret_cast<const float4*>(part + (s + 3) * MN + q);
        v.x = (((v.x + p0.x) + p1.x) + p2.x) + p3.x;
        v.y = (((v.y + p0.y) + p1.y) + p2.y) + p3.y;
        v.z = (((v.z + p0.z) + p1.z) + p2.z) + p3.z;
        v.w = (((v.w + p0.w) + p1.w) + p2.w) + p3.w;
    }
    for (; s < S; ++s) {
        const float4 p = *reinterpret_cast<const float4*>(part + s * MN + q);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(y + q) = v;
}

hipError_t launch_affine_f32(const float* x, const float* W, const float* b, float* y, int M, int N,
                             int K, int relu, hipStream_t s, float* scratch, size_t scratch_bytes, const void* W3) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const bool vec16 = (K % 4 == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15) == 0);
    const int64_t MN = (int64_t)M * N;
    if (scratch && vec16 && N % 4 == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(b)) & 15) == 0) {
        const int tm = (M + 63) / 64, tn = (N + 63) / 64, trips = (K + 63) / 64;
        int S = (512 + tm * tn - 1) / (tm * tn);                     // about two blocks per CU
        S = std::min(S, std::max(trips / 2, 1));                     // at least two trips per range
        S = (int)std::min<int64_t>(std::min(S, 16), (int64_t)(scratch_bytes / 4) / MN);
        const uint4* w3 = static_cast<const uint4*>(W3);      // bf16x3 form: see affine_splitk_x3_kernel
        if (S <= 1 && tm * tn >= 256) {
            if (w3)
                affine_splitk_x3_kernel<<<dim3(tn * tm), 256, 0, s>>>(x, w3, b, y, M, N, K, relu, trips, tn, tn * tm, 1);
            else
                affine_splitk_kernel<true><<<dim3(tn * tm), 256, 0, s>>>(x, W, b, y, M, N, K, relu, trips, tn, tn * tm, 1);
            return hipGetLastError();
        }
        if (S > 1) {
            const int tps = (trips + S - 1) / S;
            S = (trips + tps - 1) / tps;
            const int s_pad = (S + 7) & ~7;                       // ranges s_pad-S..: blocks that exit at once
            if (w3)
                affine_splitk_x3_kernel<<<dim3(s_pad * tn * tm), 256, 0, s>>>(x, w3, b, scratch, M, N, K, relu, tps, tn,
                                                                               tn * tm, S);
            else
                affine_splitk_kernel<false><<<dim3(s_pad * tn * tm), 256, 0, s>>>(x, W, b, scratch, M, N, K, relu, tps, tn,
                                                                                   tn * tm, S);
            affine_reduce_kernel<<<(unsigned)((MN / 4 + 255) / 256), 256, 0, s>>>(scratch, b, y, MN, N, S, relu);
            return hipGetLastError();
        }
    }
    dim3 grid((N + 15) / 16, (M + 15) / 16);
    if (vec16)
        affine_f32_kernel<true><<<grid, 64 * kAffWaves, 0, s>>>(x, W, b, y, M, N, K, relu);
    else
        affine_f32_kernel<false><<<grid, 64 * kAffWaves, 0, s>>>(x, W, b, y, M, N, K, relu);
    return hipGetLastError();
}

}  // namespace xvec

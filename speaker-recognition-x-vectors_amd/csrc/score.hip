// Scoring back end (next row N4): fp64 score matrices on v_mfma_f64_16x16x4_f64.
//   reference: plda_classifier.py:81-87 (fast_PLDA_scoring of speechbrain 0.5.12, numpy float64)
// C ABI: include/xvec_score.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>

#include "../../include/xvec_hip.h"
#include "../../include/xvec_score.h"
#include "tdnn_common.h"

namespace xvec {
namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// block tile = (32 WT) x (32 WT) of C: WT = 4 (128 x 128, the score matrix) or 2 (64 x 64: products whose 128 x 128 tiles would not
// give every block slot of the chip at least a tile and a half -- the two [n, 512] x [512, 512] products in front of a PLDA score
// matrix are 312 such tiles for 512 slots, and ran at 55 % of the score matrix's rate)
constexpr int kSK = 16;    // K chunk: 16 doubles = one 128-byte row = eight 16-byte pairs (k = 2c, 2c+1 in pair c)
constexpr int kSLD = 16;   // LDS row stride in doubles: 128 B, no padding.  Pair c of row r sits at position c ^ ((r >> 1) & 7)
                           // (the swizzle of tdnn_pp16.hip): a ds_read_b128 service group -- 16 lanes on 16 different rows,
                           // eight of them one pair further on -- then covers all 64 banks, and the eight lanes of a
                           // ds_write_b128 group fill one whole row.  (Round 4 padded rows to 144 B for ds_read_b64 fragments;
                           // hipcc merged those into ds_read2_b64 -- banks mod 32, 16-lane groups: rows r and r + 8 collided,
                           // SQ_LDS_BANK_CONFLICT 40 % of SQ_LDS_IDX_ACTIVE, and the LDS as busy as the matrix pipe.)

struct GemmArgs {
    const double* A;
    const double* B;
    const double* rowv;
    const double* colv;
    double* C;
    int64_t lda, ldb, ldc, M, N;
    int K, tiles_m, tiles_n, n_tiles;
    double cst, scale;
};

// Tile t of the walk -> (row tile, column tile).  The tiles are walked in 8 x 8 SUPERTILES (bands of eight row tiles, inside a
// band eight column tiles at a time, row-major inside the supertile): the 64 consecutive tiles an XCD's blocks work on at a time
// (xcd_remap) then share eight row operands and eight column operands -- 8 MiB at K = 512, the XCD's L2 twice over -- instead
// of two row operands and ALL column operands (row-major walk, round 4: 950 MB per 4874 x 4874 score matrix for 230 MB of
// operands and scores).  Bijective for any tiles_m x tiles_n (the last band / the last supertile of a band are narrower).
__device__ __forceinline__ void tile_rc(int t, int tiles_m, int tiles_n, int& r, int& c) {
    const int band = t / (8 * tiles_n);
    const int tb = t - band * 8 * tiles_n;
    const int h = min(8, tiles_m - 8 * band);
    const int sc = tb / (8 * h);
    const int w = min(8, tiles_n - 8 * sc);
    const int within = tb - sc * 8 * h;
    r = band * 8 + within / w;
    c = sc * 8 + within % w;
}

// two consecutive doubles of row `row` at column k (zero outside the matrix)
template <bool VEC>
__device__ __forceinline__ f64x2 load2(const double* __restrict__ P, int64_t ld, int64_t rows, int K, int64_t row,
                                       int k) {
    f64x2 v = {0.0, 0.0};
    if (row < rows) {
        const double* p = P + row * ld + k;
        if (VEC) {
            if (k < K) v = *reinterpret_cast<const f64x2*>(p);   // K even: k+1 < K as well
        } else {
            if (k < K) v.x = p[0];
            if (k + 1 < K) v.y = p[1];
        }
    }
    return v;
}

// 4 waves as 2x2, each (16 WT) x (16 WT) = WT x WT MFMA tiles of 16x16 (C/D: col = lane&15, row = (lane>>4) + 4*reg).
// K in 16-wide chunks through double-buffered LDS; the next chunk's global loads are in flight
// while the 64 MFMAs (64 cycles each) of the current one run, so the kernel is matrix-pipe bound.
// Persistent (round 3): two blocks per CU walk the tiles with a grid stride, and the first chunk of a block's NEXT tile is
// requested before the epilogue of the current one, so a tile no longer starts with an exposed memory round trip.
template <bool VEC, int WT>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(const GemmArgs g) {
    constexpr int kSM = 32 * WT, kSN = 32 * WT;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int tile = xcd_remap(blockIdx.x, gridDim.x);          // 64 consecutive tiles = one supertile share an XCD's L2
    if (tile >= g.n_tiles) return;
    int tr, tc;
    tile_rc(tile, g.tiles_m, g.tiles_n, tr, tc);
    int64_t m0 = (int64_t)tr * kSM;
    int64_t n0 = (int64_t)tc * kSN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, l4 = lane >> 4;
    const int piece = tid & 7, row0 = tid >> 3;

    double* sA = sm;                          // [2][kSM][kSLD]
    double* sB = sm + 2 * kSM * kSLD;         // [2][kSN][kSLD]

    f64x4 acc[WT][WT];
    f64x2 ra[WT], rb[WT];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < WT; ++p) {
            ra[p] = load2<VEC>(g.A, g.lda, g.M, g.K, m0 + row0 + 32 * p, k0 + 2 * piece);
            rb[p] = load2<VEC>(g.B, g.ldb, g.N, g.K, n0 + row0 + 32 * p, k0 + 2 * piece);
        }
    };
    const int wpos = 2 * (piece ^ ((row0 >> 1) & 7));       // (row0 + 32 p) >> 1 & 7 is the same for every p
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < WT; ++p) {
            *reinterpret_cast<f64x2*>(sA + (buf * kSM + row0 + 32 * p) * kSLD + wpos) = ra[p];
            *reinterpret_cast<f64x2*>(sB + (buf * kSN + row0 + 32 * p) * kSLD + wpos) = rb[p];
        }
    };
    // fragment reads: lane (l15, l4) takes pair l4 (k = 2 l4, 2 l4 + 1) and pair l4 + 4 of its row; an MFMA's four k (one per
    // lane quad) are then {0,2,4,6}, {1,3,5,7}, {8,..}, {9,..} -- the same permutation in both operands, so the products pair up
    const int rsw = (l15 >> 1) & 7;                         // rows wr*64 + 16 i + l15: the swizzle depends on l15 only
    const int rpos0 = 2 * (l4 ^ rsw), rpos1 = 2 * ((l4 + 4) ^ rsw);

    const int n_chunks = (g.K + kSK - 1) / kSK;
    gload(0);
    lstore(0);
    __syncthreads();
  for (;;) {
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) gload((c + 1) * kSK);
        const double* a_base = sA + (buf * kSM + wr * 16 * WT + l15) * kSLD;
        const double* b_base = sB + (buf * kSN + wc * 16 * WT + l15) * kSLD;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f64x2 a[WT], b[WT];
            const int rp = h ? rpos1 : rpos0;
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                a[i] = *reinterpret_cast<const f64x2*>(a_base + i * 16 * kSLD + rp);
                b[i] = *reinterpret_cast<const f64x2*>(b_base + i * 16 * kSLD + rp);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < WT; ++i)
#pragma unroll
                    for (int j = 0; j < WT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < n_chunks) lstore(buf ^ 1);
        __syncthreads();
    }

    // the block's next tile: its first chunk is requested now and flies under the epilogue
    const int64_t em0 = m0, en0 = n0;
    tile += gridDim.x;
    const bool more = tile < g.n_tiles;
    if (more) {
        tile_rc(tile, g.tiles_m, g.tiles_n, tr, tc);
        m0 = (int64_t)tr * kSM;
        n0 = (int64_t)tc * kSN;
        gload(0);
    }
    // ---- epilogue: + rowv[m] + colv[n] + cst, * scale; 16 lanes write 128 contiguous bytes
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        const int64_t n = en0 + wc * 16 * WT + j * 16 + l15;
        if (n >= g.N) continue;
        const double cv = (g.colv ? g.colv[n] : 0.0) + g.cst;
#pragma unroll
        for (int i = 0; i < WT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = em0 + wr * 16 * WT + i * 16 + l4 + 4 * r;
                if (m < g.M) {
                    const double rv = g.rowv ? g.rowv[m] : 0.0;
                    g.C[m * g.ldc + n] = g.scale * (acc[i][j][r] + rv + cv);
                }
            }
        }
    }
    if (!more) break;
    lstore(0);              // (every wave has passed the barrier behind the last chunk: both buffers are free)
    __syncthreads();
  }
}

// out[i,d] = x[i,d] - mean[d]
__global__ void center_rows_kernel(const double* __restrict__ x, const double* __restrict__ mean, int64_t n, int dim,
                                   double* __restrict__ out) {
    const int64_t total = n * dim;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] - mean[i % dim];
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// out[i] = 0.5 * <a_i, b_i>   (one wave per row)
__global__ void half_rowdot_kernel(const double* __restrict__ a, int64_t lda, const double* __restrict__ b, int64_t ldb,
                                   int64_t n, int dim, double* __restrict__ out) {
    const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int d = lane; d < dim; d += 64) s = fma(a[row * lda + d], b[row * ldb + d], s);
    s = wave_sum(s);
    if (lane == 0) out[row] = 0.5 * s;
}

// out[i,:] = x[i,:] / |x[i,:]|   (one wave per row; a zero row stays zero)
__global__ void normalize_rows_kernel(const double* __restrict__ x, int64_t n, int dim, double* __restrict__ out) {
    const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int d = lane; d < dim; d += 64) s = fma(x[row * dim + d], x[row * dim + d], s);
    s = wave_sum(s);
    const double inv = s > 0.0 ? 1.0 / sqrt(s) : 0.0;
    for (int d = lane; d < dim; d += 64) out[row * dim + d] = x[row * dim + d] * inv;
}

thread_local char g_serr[384] = "";

int sfail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_serr, sizeof(g_serr), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sfail(XVEC_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return XVEC_OK;
}

int gemm_nt(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t M, int64_t N, int K,
            const double* rowv, const double* colv, double cst, double scale, double* C, int64_t ldc, hipStream_t s) {
    if (M == 0 || N == 0) return XVEC_OK;
    static int num_cu_cache[64] = {};          // per device: two persistent blocks per CU
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return sfail(XVEC_ERR_HIP, "hipGetDevice failed");
    int num_cu = (dev >= 0 && dev < 64) ? num_cu_cache[dev] : 0;
    if (num_cu == 0) {
        hipDeviceProp_t prop;
        num_cu = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        if (dev >= 0 && dev < 64) num_cu_cache[dev] = num_cu;
    }
    // 128 x 128 tiles unless they would leave the chip's block slots under a tile and a half each: then 64 x 64
    const int64_t slots = 2 * (int64_t)num_cu;
    const bool small = ((M + 127) / 128) * ((N + 127) / 128) * 2 < 3 * slots;
    const int ts = small ? 64 : 128;
    const int64_t tm = (M + ts - 1) / ts, tn = (N + ts - 1) / ts;
    if (tm * tn > 0x7fffffff) return sfail(XVEC_ERR_ARG, "score matrix too large for one launch");
    GemmArgs g{A, B, rowv, colv, C, lda, ldb, ldc, M, N, K, (int)tm, (int)tn, (int)(tm * tn), cst, scale};
    const unsigned grid = (unsigned)std::min<int64_t>(tm * tn, slots);
    const size_t lds = (size_t)2 * (ts + ts) * kSLD * sizeof(double);
    const bool vec = (K % 2 == 0) && (lda % 2 == 0) && (ldb % 2 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(B) % 16 == 0);
#define SCORE_LAUNCH(VEC_, WT_)                                                                                           \
    {                                                                                                                     \
        static LdsOptIn opt;                                                                                              \
        const hipError_t ea = opt.ensure(reinterpret_cast<const void*>(&gemm_nt_f64_kernel<VEC_, WT_>), (int)lds);        \
        if (ea != hipSuccess) return sfail(XVEC_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(ea));        \
        gemm_nt_f64_kernel<VEC_, WT_><<<grid, 256, lds, s>>>(g);                                                          \
    }
    if (vec) { if (small) SCORE_LAUNCH(true, 2) else SCORE_LAUNCH(true, 4) }
    else { if (small) SCORE_LAUNCH(false, 2) else SCORE_LAUNCH(false, 4) }
#undef SCORE_LAUNCH
    return check_launch("gemm_nt_f64_kernel");
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct ScorePlan {
    double *ec, *tc, *uv, *w, *mp, *sp;
    size_t total;
};

ScorePlan make_score_plan(void* ws, int64_t ne, int64_t nt, int dim) {
    ScorePlan p{};
    char* base = static_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        double* ptr = reinterpret_cast<double*>(base + off);
        off += align256(bytes);
        return ptr;
    };
    p.ec = take((size_t)ne * dim * 8);        // centred / normalised enrol vectors
    p.tc = take((size_t)nt * dim * 8);        // ... test vectors
    p.uv = take((size_t)ne * 2 * dim * 8);    // [ e Psi | e Phi ]
    p.w = take((size_t)nt * dim * 8);         // t Phi
    p.mp = take((size_t)ne * 8);
    p.sp = take((size_t)nt * 8);
    p.total = off;
    return p;
}

}  // namespace
}  // namespace xvec

using namespace xvec;

extern "C" {

const char* xvec_score_last_error(void) { return g_serr; }

int xvec_gemm_nt_f64(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t M, int64_t N, int32_t K,
                     const double* rowv, const double* colv, double cst, double scale, double* C, int64_t ldc,
                     xvec_stream stream) {
    if (M < 0 || N < 0 || K < 1) return sfail(XVEC_ERR_ARG, "bad GEMM shape M=%lld N=%lld K=%d", (long long)M, (long long)N, K);
    if ((M && !A) || (N && !B) || (M && N && !C)) return sfail(XVEC_ERR_ARG, "null matrix pointer");
    if (lda < K || ldb < K || ldc < N) return sfail(XVEC_ERR_ARG, "row stride smaller than the row");
    return gemm_nt(A, lda, B, ldb, M, N, K, rowv, colv, cst, scale, C, ldc, static_cast<hipStream_t>(stream));
}

size_t xvec_score_workspace_bytes(int64_t n_enroll, int64_t n_test, int32_t dim) {
    if (n_enroll < 0 || n_test < 0 || dim < 1) return 0;
    return make_score_plan(nullptr, n_enroll, n_test, dim).total;
}

int xvec_plda_score(const double* enroll, int64_t n_enroll, const double* test, int64_t n_test, int32_t dim,
                    const double* mean, const double* psi_t, const double* phi_t, double plda_cst,
                    double scaling_factor, double* scores, void* workspace, size_t workspace_bytes,
                    xvec_stream stream) {
    const bool self = (test == nullptr);
    if (self) n_test = n_enroll;
    if (n_enroll < 1 || n_test < 1 || dim < 1) return sfail(XVEC_ERR_ARG, "empty enrol/test set or dim < 1");
    if (!enroll || !mean || !psi_t || !phi_t || !scores || !workspace) return sfail(XVEC_ERR_ARG, "null pointer");
    const ScorePlan p = make_score_plan(workspace, n_enroll, self ? 0 : n_test, dim);
    if (workspace_bytes < p.total)
        return sfail(XVEC_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, p.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    // centre (StatObject_SB.center_stat1)
    center_rows_kernel<<<1024, 256, 0, s>>>(enroll, mean, n_enroll, dim, p.ec);
    if ((rc = check_launch("center_rows_kernel"))) return rc;
    // [e Psi | e Phi]: ONE GEMM against [Psi^T ; Phi^T] when the caller keeps the two stacked in one buffer (the host module
    // does), otherwise two calls on the same A
    if (phi_t == psi_t + (size_t)dim * dim) {
        if ((rc = gemm_nt(p.ec, dim, psi_t, dim, n_enroll, 2 * (int64_t)dim, dim, nullptr, nullptr, 0.0, 1.0, p.uv, 2 * dim, s))) return rc;
    } else {
        if ((rc = gemm_nt(p.ec, dim, psi_t, dim, n_enroll, dim, dim, nullptr, nullptr, 0.0, 1.0, p.uv, 2 * dim, s))) return rc;
        if ((rc = gemm_nt(p.ec, dim, phi_t, dim, n_enroll, dim, dim, nullptr, nullptr, 0.0, 1.0, p.uv + dim, 2 * dim, s))) return rc;
    }
    const unsigned rb_e = (unsigned)((n_enroll + 3) / 4);
    half_rowdot_kernel<<<rb_e, 256, 0, s>>>(p.uv + dim, 2 * dim, p.ec, dim, n_enroll, dim, p.mp);   // model_part
    if ((rc = check_launch("half_rowdot_kernel"))) return rc;
    const double* tc = p.ec;
    const double* sp = p.mp;
    if (!self) {
        center_rows_kernel<<<1024, 256, 0, s>>>(test, mean, n_test, dim, p.tc);
        if ((rc = check_launch("center_rows_kernel"))) return rc;
        if ((rc = gemm_nt(p.tc, dim, phi_t, dim, n_test, dim, dim, nullptr, nullptr, 0.0, 1.0, p.w, dim, s))) return rc;
        half_rowdot_kernel<<<(unsigned)((n_test + 3) / 4), 256, 0, s>>>(p.w, dim, p.tc, dim, n_test, dim, p.sp);   // seg_part
        if ((rc = check_launch("half_rowdot_kernel"))) return rc;
        tc = p.tc;
        sp = p.sp;
    }
    // scores = scaling * (model_part[:,None] + seg_part[None,:] + plda_cst + (e Psi) t^T)
    return gemm_nt(p.uv, 2 * dim, tc, dim, n_enroll, n_test, dim, p.mp, sp, plda_cst, scaling_factor, scores, n_test, s);
}

int xvec_cosine_score(const double* enroll, int64_t n_enroll, const double* test, int64_t n_test, int32_t dim,
                      double* scores, void* workspace, size_t workspace_bytes, xvec_stream stream) {
    const bool self = (test == nullptr);
    if (self) n_test = n_enroll;
    if (n_enroll < 1 || n_test < 1 || dim < 1) return sfail(XVEC_ERR_ARG, "empty enrol/test set or dim < 1");
    if (!enroll || !scores || !workspace) return sfail(XVEC_ERR_ARG, "null pointer");
    const ScorePlan p = make_score_plan(workspace, n_enroll, self ? 0 : n_test, dim);
    if (workspace_bytes < p.total)
        return sfail(XVEC_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, p.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    normalize_rows_kernel<<<(unsigned)((n_enroll + 3) / 4), 256, 0, s>>>(enroll, n_enroll, dim, p.ec);
    if ((rc = check_launch("normalize_rows_kernel"))) return rc;
    const double* tc = p.ec;
    if (!self) {
        normalize_rows_kernel<<<(unsigned)((n_test + 3) / 4), 256, 0, s>>>(test, n_test, dim, p.tc);
        if ((rc = check_launch("normalize_rows_kernel"))) return rc;
        tc = p.tc;
    }
    return gemm_nt(p.ec, dim, tc, dim, n_enroll, n_test, dim, nullptr, nullptr, 0.0, 1.0, scores, n_test, s);
}

}  // extern "C"

// Scoring back end (next row N4): fp64 score matrices on v_mfma_f64_16x16x4_f64.
//   reference: plda_classifier.py:81-87 (fast_PLDA_scoring of speechbrain 0.5.12, numpy float64)
// C ABI: include/xvec_score.h.
#include <hip/hip_runtime.h>
#include <cstdlib>

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

// block tile = (32 WT) x (32 WT) of C: WT = 4 (128 x 128, two blocks per CU) or 2 (64 x 64, four blocks per CU); gemm_nt() picks by how
// full the chip's last round of tiles is (the two preludes of a PLDA score matrix always take 64 x 64)
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
    // --- the scorers' fused forms (all off in xvec_gemm_nt_f64) ---
    // score product (template PRE = false):
    int sym;                 // M == N and C symmetric by construction: walk tiles with tile row <= tile column only, write
                             // every element above the diagonal twice ([m,n] and [n,m]: the SAME value, so C == C^T bit for bit)
    int rowv_parts;          // > 0: rowv of row m is the sum of rowv_parts partials at rowv[m * rowv_ld ..] (in that order)
    int colv_parts;          // ... colv likewise
    int64_t rowv_ld, colv_ld;
    // prelude product (template PRE = true): A operand = A[m,k] - sub[k] (the centring of center_stat1, applied while a chunk
    // is staged); the blocks of tile column 0 write the centred rows to cen_out[M, K] on the way (they become the score
    // product's B operand); row dots in the epilogue:
    //   dot_out[m * dot_ld + slot] = 0.5 * sum over one wave's columns n >= dot_col0 of acc[m, n] * (A[m, n - dot_col0] - sub[n - dot_col0])
    // (0.5 e Phi e' of fast_PLDA_scoring: the product [e Phi] is never written -- columns >= dot_col0 of C are not stored)
    const double* sub;
    double* cen_out;
    double* dot_out;
    int64_t dot_ld;
    int dot_col0;
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

// The same walk over the tiles ON OR ABOVE the diagonal of a square tile grid (T x T, tile row <= tile column): band by band,
// the diagonal supertile first (its upper triangle row by row), then the band's other supertiles as above.  t < T (T + 1) / 2.
__device__ __forceinline__ void tile_rc_sym(int t, int T, int& r, int& c) {
    int band = 0;
    for (;;) {                                   // (T + 7) / 8 iterations at most, scalar
        const int h = min(8, T - 8 * band);
        const int cnt = h * (h + 1) / 2 + h * (T - 8 * band - h);
        if (t < cnt) break;
        t -= cnt;
        ++band;
    }
    const int h = min(8, T - 8 * band);
    int i = 0;
    for (; i < h; ++i) {                         // diagonal supertile: row i holds h - i tiles
        if (t < h - i) {
            r = band * 8 + i;
            c = band * 8 + i + t;
            return;
        }
        t -= h - i;
    }
    const int sc = t / (8 * h);                  // (h == 8 here: a shorter band is the last one and has no supertile to its right)
    const int w = min(8, T - 8 * (band + 1 + sc));
    const int within = t - sc * 8 * h;
    r = band * 8 + within / w;
    c = (band + 1 + sc) * 8 + within % w;
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
template <bool VEC, int WT, bool PRE>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(const GemmArgs g) {
    constexpr int kSM = 32 * WT, kSN = 32 * WT;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int tile = xcd_remap(blockIdx.x, gridDim.x);          // 64 consecutive tiles = one supertile share an XCD's L2
    if (tile >= g.n_tiles) return;
    int tr, tc;
    if (!PRE && g.sym) tile_rc_sym(tile, g.tiles_m, tr, tc);
    else tile_rc(tile, g.tiles_m, g.tiles_n, tr, tc);
    int64_t m0 = (int64_t)tr * kSM;
    int64_t n0 = (int64_t)tc * kSN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, l4 = lane >> 4;
    const int piece = tid & 7, row0 = tid >> 3;

    double* sA = sm;                          // [2][kSM][kSLD]
    double* sB = sm + 2 * kSM * kSLD;         // [2][kSN][kSLD]

    f64x4 acc[WT][WT];
    f64x2 ra[WT], rb[WT];
    f64x2 rsub = {0.0, 0.0};                                // prelude product: sub[k], sub[k + 1] of the chunk in flight
    int rk = 0;
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < WT; ++p) {
            ra[p] = load2<VEC>(g.A, g.lda, g.M, g.K, m0 + row0 + 32 * p, k0 + 2 * piece);
            rb[p] = load2<VEC>(g.B, g.ldb, g.N, g.K, n0 + row0 + 32 * p, k0 + 2 * piece);
        }
        if constexpr (PRE) {
            rk = k0 + 2 * piece;
            rsub.x = g.sub && rk < g.K ? g.sub[rk] : 0.0;
            rsub.y = g.sub && rk + 1 < g.K ? g.sub[rk + 1] : 0.0;
        }
    };
    const int wpos = 2 * (piece ^ ((row0 >> 1) & 7));       // (row0 + 32 p) >> 1 & 7 is the same for every p
    auto lstore = [&](int buf) {
        if constexpr (PRE) {
            // centre the A rows HERE, behind the MFMAs of the chunk before (in gload the subtraction would wait for the loads
            // it has just issued); k >= K stays zero (the padding of the last chunk), rows >= M are never stored; the blocks
            // of tile column 0 write the centred rows out for the score product
#pragma unroll
            for (int p = 0; p < WT; ++p) {
                ra[p] -= rsub;
                const int64_t m = m0 + row0 + 32 * p;
                if (g.cen_out && n0 == 0 && m < g.M) {
                    if (rk < g.K) g.cen_out[m * g.K + rk] = ra[p].x;
                    if (rk + 1 < g.K) g.cen_out[m * g.K + rk + 1] = ra[p].y;
                }
            }
        }
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

    // the tile's row / column terms (vectors, or partials summed ONCE per tile: thread t < 32 WT takes row t, the next 32 WT
    // threads the columns) go through the second operand buffer, which nobody touches before the next tile's second chunk
    // (before the next tile's first chunk is requested: its 8 WT registers are not live here)
    if constexpr (!PRE) {
        double* sums = sA + kSM * kSLD;
        const int64_t em0 = m0, en0 = n0;
        if (tid < kSM) {
            const int64_t m = em0 + tid;
            double v = 0.0;
            if (g.rowv && m < g.M) {
                if (g.rowv_parts > 0) {
                    const double* q = g.rowv + m * g.rowv_ld;
                    v = q[0];
                    for (int i = 1; i < g.rowv_parts; ++i) v += q[i];
                } else {
                    v = g.rowv[m];
                }
            }
            sums[tid] = v;
        } else if (tid < kSM + kSN) {
            const int64_t n = en0 + tid - kSM;
            double v = 0.0;
            if (g.colv && n < g.N) {
                if (g.colv_parts > 0) {
                    const double* q = g.colv + n * g.colv_ld;
                    v = q[0];
                    for (int i = 1; i < g.colv_parts; ++i) v += q[i];
                } else {
                    v = g.colv[n];
                }
            }
            sums[tid] = v;
        }
        __syncthreads();
    }
    // the block's next tile: its first chunk is requested now and flies under the epilogue
    const int64_t em0 = m0, en0 = n0;
    const bool diag = !PRE && g.sym && tr == tc;
    tile += gridDim.x;
    const bool more = tile < g.n_tiles;
    if (more) {
        if (!PRE && g.sym) tile_rc_sym(tile, g.tiles_m, tr, tc);
        else tile_rc(tile, g.tiles_m, g.tiles_n, tr, tc);
        m0 = (int64_t)tr * kSM;
        n0 = (int64_t)tc * kSN;
        gload(0);
    }
    // ---- epilogue: + rowv[m] + colv[n] + cst, * scale; 16 lanes write 128 contiguous bytes
    if constexpr (PRE) {
        // row dots of the columns n >= dot_col0 with the centred rows of A (every lane of a 16-lane group holds one column of
        // each of the wave's WT column blocks; the group's sum goes to the wave's slot), plain stores of the columns below
        const int64_t first = en0 + wc * 16 * WT;                     // the wave's first column
        const bool any = first + 16 * WT > g.dot_col0 && first < g.N;
        const int slot = any ? (int)((first - (g.dot_col0 - g.dot_col0 % (16 * WT))) / (16 * WT)) : 0;
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                const int64_t n = en0 + wc * 16 * WT + j * 16 + l15;
                const int64_t d = n - g.dot_col0;
                if (n >= g.N) continue;
                const double sub = d >= 0 && g.sub ? g.sub[d] : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = em0 + wr * 16 * WT + i * 16 + l4 + 4 * r;
                    if (m >= g.M) continue;
                    if (d >= 0) part[r] = fma(acc[i][j][r], g.A[m * g.lda + d] - sub, part[r]);
                    else g.C[m * g.ldc + n] = acc[i][j][r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = part[r];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                const int64_t m = em0 + wr * 16 * WT + i * 16 + l4 + 4 * r;
                if (any && g.dot_out && l15 == 0 && m < g.M) g.dot_out[m * g.dot_ld + slot] = 0.5 * v;
            }
        }
    } else {
        const double* sums = sA + kSM * kSLD;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int cn = wc * 16 * WT + j * 16 + l15;
            const int64_t n = en0 + cn;
            if (n >= g.N) continue;
            const double cv = sums[kSM + cn] + g.cst;
#pragma unroll
            for (int i = 0; i < WT; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rm = wr * 16 * WT + i * 16 + l4 + 4 * r;
                    const int64_t m = em0 + rm;
                    if (m < g.M && !(diag && m > n)) {
                        const double v = g.scale * (acc[i][j][r] + sums[rm] + cv);
                        g.C[m * g.ldc + n] = v;
                        if (g.sym && m != n) g.C[n * g.ldc + m] = v;      // the mirror image: the same bits
                    }
                }
            }
        }
    }
    if (!more) break;
    lstore(0);              // (every wave has passed the barrier behind the last chunk: both buffers are free)
    __syncthreads();
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
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

// the scorers' fused forms of one product (see GemmArgs)
struct GemmExtra {
    bool sym = false;
    int rowv_parts = 0, colv_parts = 0;
    int64_t rowv_ld = 0, colv_ld = 0;
    bool pre = false;
    const double* sub = nullptr;
    double* cen_out = nullptr;
    double* dot_out = nullptr;
    int64_t dot_ld = 0;
    int dot_col0 = 0;
    int* parts_out = nullptr;          // how many row-dot partials per row the launch writes
};

int gemm_nt(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t M, int64_t N, int K,
            const double* rowv, const double* colv, double cst, double scale, double* C, int64_t ldc, hipStream_t s,
            const GemmExtra* ex = nullptr) {
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
    const bool sym = ex && ex->sym, pre = ex && ex->pre;
    if (sym && M != N) return sfail(XVEC_ERR_ARG, "symmetric walk needs a square score matrix");
    // Two tilings (a symmetric walk visits only the tiles on or above the diagonal):
    //   128 x 128, two blocks per CU (255 registers);  64 x 64, FOUR blocks per CU (121 registers, 32 KiB of LDS each).
    // Per CU the two finish the same work in the same time within a few per cent (a round of four 64 x 64 tiles against a round
    // of two 128 x 128 ones: 0.49 for K <= 256, 0.53 for longer K, measured N = 1200 ... 16384, profiles/experiments/README.md), and
    // a block slot walks ceil(tiles / slots) tiles -- so what decides is how full the LAST round is.  The reference's self-score
    // of 4874 vectors is 780 tiles of 128 x 128 on 512 slots (two rounds, the second half empty) against 3003 of 64 x 64 on 1024
    // (three rounds): 0.261 -> 0.200 ms in the low-rank form, 0.487 -> 0.411 dense (round 6; until then the 64 x 64 kernel ran two
    // blocks per CU and served only products under a tile and a half per slot).
    const int64_t slots128 = 2 * (int64_t)num_cu, slots64 = 4 * (int64_t)num_cu;
    auto count = [&](int64_t ts) {
        const int64_t tm = (M + ts - 1) / ts, tn = (N + ts - 1) / ts;
        return sym ? tm * (tm + 1) / 2 : tm * tn;
    };
    // (the prelude product stays on 64 x 64 tiles at every size: with the centring and the row dots on top, the 128 x 128 form
    //  needs more registers than it has)
    const int64_t r128 = (count(128) + slots128 - 1) / slots128, r64 = (count(64) + slots64 - 1) / slots64;
    const bool small = pre || count(128) * 2 < 3 * slots128 || r64 * (K <= 256 ? 49 : 53) < r128 * 100;
    const int ts = small ? 64 : 128;
    const int64_t tm = (M + ts - 1) / ts, tn = (N + ts - 1) / ts;
    const int64_t n_tiles = count(ts);
    if (n_tiles > 0x7fffffff) return sfail(XVEC_ERR_ARG, "score matrix too large for one launch");
    GemmArgs g{};
    g.A = A; g.B = B; g.rowv = rowv; g.colv = colv; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N;
    g.K = K; g.tiles_m = (int)tm; g.tiles_n = (int)tn; g.n_tiles = (int)n_tiles;
    g.cst = cst; g.scale = scale;
    if (ex) {
        g.sym = sym ? 1 : 0;
        g.rowv_parts = ex->rowv_parts; g.colv_parts = ex->colv_parts; g.rowv_ld = ex->rowv_ld; g.colv_ld = ex->colv_ld;
        g.sub = ex->sub; g.cen_out = ex->cen_out; g.dot_out = ex->dot_out; g.dot_ld = ex->dot_ld; g.dot_col0 = ex->dot_col0;
        if (pre && !g.dot_out) g.dot_col0 = (int)std::min<int64_t>(N, 0x7fffffff);      // no row dots: every column is stored
        if (pre && g.dot_out) {
            const int wcols = ts / 2;                       // columns of one wave: 16 WT
            const int parts = (int)((N + wcols - 1) / wcols - g.dot_col0 / wcols);
            if (parts > g.dot_ld) return sfail(XVEC_ERR_ARG, "row-dot partials do not fit their rows");
            if (ex->parts_out) *ex->parts_out = parts;
        }
    }
    const unsigned grid = (unsigned)std::min<int64_t>(n_tiles, small ? slots64 : slots128);
    const size_t lds = (size_t)2 * (ts + ts) * kSLD * sizeof(double);
    const bool vec = (K % 2 == 0) && (lda % 2 == 0) && (ldb % 2 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(B) % 16 == 0);
#define SCORE_LAUNCH(VEC_, WT_, PRE_)                                                                                     \
    {                                                                                                                     \
        static LdsOptIn opt;                                                                                              \
        const hipError_t ea = opt.ensure(reinterpret_cast<const void*>(&gemm_nt_f64_kernel<VEC_, WT_, PRE_>), (int)lds);  \
        if (ea != hipSuccess) return sfail(XVEC_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(ea));        \
        gemm_nt_f64_kernel<VEC_, WT_, PRE_><<<grid, 256, lds, s>>>(g);                                                    \
    }
    if (vec) { if (pre) SCORE_LAUNCH(true, 2, true) else if (small) SCORE_LAUNCH(true, 2, false) else SCORE_LAUNCH(true, 4, false) }
    else { if (pre) SCORE_LAUNCH(false, 2, true) else if (small) SCORE_LAUNCH(false, 2, false) else SCORE_LAUNCH(false, 4, false) }
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
    p.ec = take((size_t)ne * dim * 8);        // centred (PLDA) / normalised (cosine) enrol vectors
    p.tc = take((size_t)nt * dim * 8);        // ... test vectors
    p.uv = take((size_t)ne * 2 * dim * 8);    // rows of [ e Psi | the row's model_part partials (e Phi itself is never written) ]
    p.w = take((size_t)nt * dim * 8);         // rows of the test vectors' seg_part partials (t Phi is never written)
    p.mp = take((size_t)ne * 8);              // (unused since round 6; kept so that the workspace size does not change)
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

// Launches (round 6).  Self case (the reference's use, plda_score_stat.py:19-20: en_stat = te_stat): TWO --
//   1. [e Psi | e Phi] = (enroll - mean) [Psi^T ; Phi^T]^T: the centring (center_stat1) happens while the A chunks are staged,
//      the blocks of tile column 0 write the centred rows out on the way; the Psi half is stored, the Phi half only feeds
//      the row dots 0.5 e Phi e' of its epilogue (one partial per row and wave column: deterministic, no atomics)
//   2. scores = scaling (model_part[:,None] + model_part[None,:] + cst + (e Psi) e^T) over the tiles on or above the diagonal
//      only (Psi is symmetric, so S = S^T: 780 of 1521 tiles at N = 4874), every off-diagonal element written twice with
//      the same bits; model_part = the sum of the row's partials, taken in the epilogue.
// With a separate test set: one more prelude product ((test - mean) Phi: nothing stored but the centred rows and the row
// dots) and the full tile walk.
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
    int parts_e = 0, parts_t = 0;
    GemmExtra ex;
    ex.pre = true;
    ex.sub = mean;
    ex.cen_out = p.ec;
    ex.dot_out = p.uv + dim;              // model_part partials in the (never written) Phi half of the row
    ex.dot_ld = 2 * (int64_t)dim;
    ex.parts_out = &parts_e;
    if (phi_t == psi_t + (size_t)dim * dim) {
        // the caller keeps [Psi^T ; Phi^T] stacked in one buffer (the host module does): ONE product
        ex.dot_col0 = dim;
        if ((rc = gemm_nt(enroll, dim, psi_t, dim, n_enroll, 2 * (int64_t)dim, dim, nullptr, nullptr, 0.0, 1.0, p.uv, 2 * dim, s, &ex))) return rc;
    } else {
        // two: e Psi (stored; no column takes part in a row dot: dot_col0 = its width), then e Phi (row dots only)
        GemmExtra e1 = ex;
        e1.dot_col0 = dim;
        e1.parts_out = nullptr;
        if ((rc = gemm_nt(enroll, dim, psi_t, dim, n_enroll, dim, dim, nullptr, nullptr, 0.0, 1.0, p.uv, 2 * dim, s, &e1))) return rc;
        ex.cen_out = nullptr;
        ex.dot_col0 = 0;
        if ((rc = gemm_nt(enroll, dim, phi_t, dim, n_enroll, dim, dim, nullptr, nullptr, 0.0, 1.0, p.uv + dim, 2 * dim, s, &ex))) return rc;
    }
    const double* tvec = p.ec;
    const double* sp = p.uv + dim;
    int64_t sp_ld = 2 * (int64_t)dim;
    parts_t = parts_e;
    if (!self) {
        GemmExtra et;
        et.pre = true;
        et.sub = mean;
        et.cen_out = p.tc;
        et.dot_out = p.w;                 // seg_part partials [n_test, dim]
        et.dot_ld = dim;
        et.dot_col0 = 0;
        et.parts_out = &parts_t;
        if ((rc = gemm_nt(test, dim, phi_t, dim, n_test, dim, dim, nullptr, nullptr, 0.0, 1.0, p.w, dim, s, &et))) return rc;
        tvec = p.tc;
        sp = p.w;
        sp_ld = dim;
    }
    // scores = scaling * (model_part[:,None] + seg_part[None,:] + plda_cst + (e Psi) t^T)
    GemmExtra eb;
    eb.sym = self;
    eb.rowv_parts = parts_e;
    eb.rowv_ld = 2 * (int64_t)dim;
    eb.colv_parts = parts_t;
    eb.colv_ld = sp_ld;
    return gemm_nt(p.uv, 2 * dim, tvec, dim, n_enroll, n_test, dim, p.uv + dim, sp, plda_cst, scaling_factor, scores, n_test, s, &eb);
}

// The same scores through the model's low-rank structure (round 6).  With tot = F F' + Sigma, L = tot^-1 F [dim, rank],
// G = F' L, W = (I - G^2)^-1 and Z = W G, the matrices of fast_PLDA_scoring are Phi = -L Z L' and Psi = L W L' EXACTLY
// (Woodbury; the host checks nothing, it derives L, W, Z instead of Phi, Psi: scoring.plda_lowrank), so with y = (x - mean) L
//   0.5 x' Phi x = -0.5 y Z y'       e' Psi t = (y_e W) y_t'
// and the [n, n] product runs over K = rank instead of dim (the reference trains rank_f = 50 .. 200 on 512-d x-vectors,
// main.py:293-308: 2.6 .. 10 times fewer multiply-adds, results equal to rounding).  Launches, self case: y (centring while
// staged), [y W | row dots of y (-Z) with y], the score matrix over the upper triangle of tiles.
int xvec_plda_score_lowrank(const double* enroll, int64_t n_enroll, const double* test, int64_t n_test, int32_t dim,
                            int32_t rank, const double* mean, const double* l_t, const double* wz_t, double plda_cst,
                            double scaling_factor, double* scores, void* workspace, size_t workspace_bytes,
                            xvec_stream stream) {
    const bool self = (test == nullptr);
    if (self) n_test = n_enroll;
    if (n_enroll < 1 || n_test < 1 || dim < 1 || rank < 1 || rank > dim)
        return sfail(XVEC_ERR_ARG, "empty enrol/test set, dim < 1 or rank outside [1, dim]");
    if (!enroll || !mean || !l_t || !wz_t || !scores || !workspace) return sfail(XVEC_ERR_ARG, "null pointer");
    const ScorePlan p = make_score_plan(workspace, n_enroll, self ? 0 : n_test, dim);
    if (workspace_bytes < p.total)
        return sfail(XVEC_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, p.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t R = rank;
    int rc;
    int parts_e = 0, parts_t = 0;
    // y_e = (enroll - mean) L -> p.ec [n_enroll, R]
    GemmExtra ey;
    ey.pre = true;
    ey.sub = mean;
    if ((rc = gemm_nt(enroll, dim, l_t, dim, n_enroll, R, dim, nullptr, nullptr, 0.0, 1.0, p.ec, R, s, &ey))) return rc;
    // [u | v] = y_e [W | -Z]: u stored in p.uv[:, :R], v only as the row dots 0.5 <v_i, y_i> (partials in p.uv[:, R:])
    GemmExtra eu;
    eu.pre = true;
    eu.dot_out = p.uv + R;
    eu.dot_ld = 2 * R;
    eu.dot_col0 = rank;
    eu.parts_out = &parts_e;
    if ((rc = gemm_nt(p.ec, R, wz_t, R, n_enroll, 2 * R, rank, nullptr, nullptr, 0.0, 1.0, p.uv, 2 * R, s, &eu))) return rc;
    const double* yt = p.ec;
    const double* sp = p.uv + R;
    int64_t sp_ld = 2 * R;
    parts_t = parts_e;
    if (!self) {
        GemmExtra et = ey;
        if ((rc = gemm_nt(test, dim, l_t, dim, n_test, R, dim, nullptr, nullptr, 0.0, 1.0, p.tc, R, s, &et))) return rc;
        GemmExtra ev;
        ev.pre = true;
        ev.dot_out = p.w;                 // seg_part partials [n_test, R]
        ev.dot_ld = R;
        ev.dot_col0 = 0;
        ev.parts_out = &parts_t;
        if ((rc = gemm_nt(p.tc, R, wz_t + R * R, R, n_test, R, rank, nullptr, nullptr, 0.0, 1.0, p.w, R, s, &ev))) return rc;
        yt = p.tc;
        sp = p.w;
        sp_ld = R;
    }
    GemmExtra eb;
    eb.sym = self;
    eb.rowv_parts = parts_e;
    eb.rowv_ld = 2 * R;
    eb.colv_parts = parts_t;
    eb.colv_ld = sp_ld;
    return gemm_nt(p.uv, 2 * R, yt, R, n_enroll, n_test, rank, p.uv + R, sp, plda_cst, scaling_factor, scores, n_test, s, &eb);
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
    GemmExtra ex;
    ex.sym = self;                        // <a, b> = <b, a>: walk the upper triangle, mirror the rest
    return gemm_nt(p.ec, dim, tc, dim, n_enroll, n_test, dim, nullptr, nullptr, 0.0, 1.0, scores, n_test, s, &ex);
}

}  // extern "C"

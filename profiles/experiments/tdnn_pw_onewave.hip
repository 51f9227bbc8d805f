// Frame-level TDNN layer, plain bf16, store epilogue (layers 2-4 of the bf16 path at large batches): ONE wave per SIMD.
// The same implicit GEMM as tdnn_pp16.hip (reference tdnn_layer.py:26-41: context gather -> Linear -> ReLU; the eval
// BatchNorm is deferred into the consumer's weights, xvec_api.hip refold) on a mapping with fewer operand reads and
// fewer DMA requests per MFMA (round 6; the bounds that led here: profiles/experiments/README.md, onewave_bound.hip):
//   * 256-thread block per CU, block tile = 64 MF frames x 256 channels (MF = 4, 3 or 2 units of 64 frames), 2 x 2 waves,
//     wave tile = 8 MF frames x 128 channels on v_mfma_f32_16x16x32_bf16: up to 8 x 8 accumulators of 16 x 16 = 256 registers,
//     all in AGPRs (one wave per SIMD owns the whole 512-entry file).  Per 32-deep stage a wave issues 8 MF MFMAs for
//     MF + 8 fragment reads: 0.25 ds_read_b128 per MFMA at MF = 8 (tdnn_pp16.hip's 128 x 64 wave tile: 0.375).
//   * K in STAGES of 32 (64-byte rows), taps innermost: stage (c, t) = 32-channel slab c of the input at tap t.  The
//     activation slab of a stage is the tile's frames PLUS the rows the other taps and the utterance boundaries inside the
//     tile add -- one contiguous window of input rows, because a tile's output rows p read input rows p + u(p) span + t dil
//     and span = 2 dil: the next utterance's first row follows the previous one's last tap-2 row.  It is requested ONCE per
//     slab and read by all three taps at row offsets (one DMA piece per 16 rows x 64 B; 5 per wave and slab instead of 12).
//   * weights stage-major [column block][stage][256 rows][32 k] (pack.hip, pack_tdnn_weight_stage_kernel): a stage is one
//     contiguous 16 KiB; row 16 j + c of a wave's 128 rows holds channel 8 c + j, so a lane's eight accumulators of a frame
//     are eight ADJACENT channels: the epilogue stores 16 bytes per lane.
//   * both operands by LDS-DMA into rings of four slots (W 16 KiB, A 20 KiB each), requested three stages / three slabs ahead
//     behind a counted vmcnt, ONE barrier per stage; the 16-byte-chunk swizzle q ^ 3 (row >> 3 & 1), applied on the source
//     address, makes a fragment read (16 consecutive rows x 64 B) conflict-free when its first row is a multiple of 8.
//   * everything in one in-order stream per wave: per row of eight MFMAs one A fragment two rows ahead, one or two W
//     fragments of the next stage, at most two DMA pieces.
//   * the accumulators start from srcC = 0 in a tile's first stage; the epilogue adds the folded bias, takes the ReLU on the
//     packed pair and writes whole 256-byte row segments.
// Persistent blocks, equal ranges of 64-frame units per 256-channel column, cut into tiles of 4 / 3 / 2 units, as tdnn_pp16.hip.
#include "tdnn_common.h"

namespace xvec {
namespace pw {

constexpr int kWSlot = 256 * 64;                  // one stage of a column block's weights: 16 KiB
constexpr int kWRing = 4 * kWSlot;
constexpr int kASlabRows = 320;                   // rows of an activation slab slot: 256 frames + 64 rows of taps / boundaries
constexpr int kASlot = kASlabRows * 64;           // 20 KiB
constexpr int kAOff = kWRing;
constexpr int kConstOff = kAOff + 4 * kASlot;     // bias of the block's 256 channels
constexpr int kLdsBytes = kConstOff + 256 * 4;
constexpr int kThreads = 256;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ i32x4 make_srd(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    i32x4 d;
    d.x = (int)__builtin_amdgcn_readfirstlane((unsigned)v);
    d.y = (int)(__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) & 0xffffu);   // stride 0
    d.z = 0x7fffffff;
    d.w = 0x00020000;
    return d;
}
// One DMA piece: 64 lanes x 16 B from per-lane source offsets to 1 KiB of LDS at `dst` (wave uniform).  Inline asm: hipcc
// would wait vmcnt(0) for the builtin form before the next ds_read (tdnn_pp16.hip, dma16).
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned dst, int voff, int soff) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff)
        : "memory", "m0");
}

#define PW_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")
#define PW_WAIT_LGKM()                          \
    {                                           \
        SB();                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);     \
        SB();                                   \
    }
#define PW_BARRIER()                  \
    {                                 \
        SB();                         \
        __builtin_amdgcn_s_barrier(); \
        SB();                         \
    }

struct Tile {
    int64_t m0;            // first output row
    int units;             // 64-frame units: 4, 3 or 2 (MF = 2 units)
};

// what a tile's requests need: the descriptor at its slab's first input row
struct TileSrc {
    i32x4 xrsrc;
};

struct Cursor {            // utterance holding the current tile's first row, and where the next one starts
    int u;
    int64_t off_next;
};

// first output row of utterance u + 1 (the offsets are read with a scalar load: sload_i64, tdnn_common.h)
__device__ __forceinline__ int64_t next_off(const RowMap& m, int u) {
    u = __builtin_amdgcn_readfirstlane(u);
    if (m.offsets != nullptr) return sload_i64(m.offsets + u + 1) - (int64_t)(u + 1) * m.cum;
    return (int64_t)(u + 1) * (m.fixed_T - m.cum);
}

// advance the cursor to the utterance that holds row m0
__device__ __forceinline__ void seek(const RowMap& m, Cursor& c, int64_t m0) {
    const int n_last = m.n_utts - 1;
    while (m0 >= c.off_next && c.u < n_last) {
        c.u = __builtin_amdgcn_readfirstlane(c.u + 1);
        c.off_next = next_off(m, c.u);
    }
}

// accumulator (frame block i, channel block j) += A fragment x W fragment; the FIRST stage of a tile writes it from srcC = 0
#define PW_MF(i_, j_, af_, wf_)                                                                                   \
    if constexpr (kFirst) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[i_][j_]) : "v"(af_), "v"(wf_)); \
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(af_), "v"(wf_));
#define PW_RD(dst_, off_) dst_ = *reinterpret_cast<const f32x4*>(smem + (off_));

// One tile: mf frame blocks per wave (8 / 6 / 4: a run-time value -- ONE body serves every tile height, rows i >= mf are skipped
// by a uniform branch each; as three instantiations per tap count the kernel ran out of scalar registers), NT taps (3: layers
// 2-3, one slab per three stages; 1: layer 4).
template <int NT>
__device__ __forceinline__ void process_tile(const TdnnArgs& a, char* smem, const Tile& t, const i32x4& xr_cur, const i32x4& xr_next,
                                             const i32x4& wrsrc, bool has_next, int64_t valid_end, Cursor cur, int n0, int wave_in,
                                             int n_slabs, unsigned lds0) {
    constexpr int NPA = NT == 3 ? 5 : 4;           // activation pieces per wave and slab
    // lane id and wave index from OPAQUE instructions, once per tile: derived from threadIdx hipcc computes every lane- and
    // wave-dependent address before the tile loop and carries them through it
    int lane, wave = wave_in;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    asm volatile("s_mov_b32 %0, %1" : "=s"(wave) : "s"(wave));
    const int mf = 2 * t.units;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int dil = a.tap_rows;
    const int nst = n_slabs * NT;                  // stages of a tile (a multiple of 4: the launcher checks the shapes)

    // ---- this lane's fragment rows: slab row of (frame block i, tap t) = 16 (wr mf + i) + r + (boundaries passed) span + t dil
    unsigned a_off[8][NT];
    {
        int cnt[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cnt[i] = 0;
        const int p0 = 16 * (wr * mf) + r;                 // rows relative to the tile's first: 32-bit
        const int t_rows = 64 * t.units;
        const int n_last = a.out_map.n_utts - 1;
        int u = cur.u;
        int64_t nxt = cur.off_next;
        while (nxt < t.m0 + t_rows && u < n_last) {
            const int nrel = (int)(nxt - t.m0);
#pragma unroll
            for (int i = 0; i < 8; ++i) cnt[i] += (p0 + 16 * i >= nrel) ? 1 : 0;
            u = __builtin_amdgcn_readfirstlane(u + 1);
            nxt = next_off(a.out_map, u);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
                const int row = p0 + 16 * i + cnt[i] * a.span + tp * dil;
                a_off[i][tp] = kAOff + row * 64 + ((q ^ (3 * ((row >> 3) & 1))) << 4);
            }
    }
    // weight fragments: rows 128 wc + 16 j + r of the stage's slot
    unsigned w_rd = wc * 8 * 1024 + r * 64 + ((q ^ (3 * (r >> 3))) << 4);
    asm volatile("" : "+v"(w_rd));
    // DMA lanes: piece = 16 rows x 64 B; lane -> row lane >> 2, LDS position lane & 3 holds source chunk (lane & 3) ^ 3 (row >> 3)
    const int prow = lane >> 2;
    const int pchunk = ((lane & 3) ^ (3 * (prow >> 3))) << 4;
    // this wave's pieces, one lane offset each (registers are plentiful here, scalar registers are not): activation piece p of
    // a slab = slab rows 16 (NPA wave + p) ..+15, + 64 slab bytes (scalar); weight piece p of a stage = rows 16 (4 wave + p) ..
    int av[NPA], wv[4];
#pragma unroll
    for (int p = 0; p < NPA; ++p) av[p] = (16 * (wave * NPA + p) + prow) * a.ldx * 2 + pchunk;
#pragma unroll
    for (int p = 0; p < 4; ++p) wv[p] = (16 * (wave * 4 + p) + prow) * 64 + pchunk;

    f32x4 acc[8][8];
    // fragments: two weight sets (this stage | the next); activations: rows 0, 1 of a stage in nx[], row i >= 2 in af[(i - 2) % 3]
    // (read two rows ahead: rows i, i + 1 live, i + 2 in flight), the next stage's rows 0, 1 back into nx[] behind rows mf - 2, mf - 1
    f32x4 wf[2][8], nx[2], af[3];

    // the tile's first fragments (its first slab and stage landed behind the previous tile's last barrier / the kernel prologue)
#pragma unroll
    for (int j = 0; j < 8; ++j) PW_RD(wf[0][j], w_rd + j * 1024)
    PW_RD(nx[0], a_off[0][0])
    PW_RD(nx[1], a_off[1][0])

    // ---- one stage.  c: slab, TAP, P: parity of the weight fragment registers; the stage's index in the tile is st
    //   behind MFMA 1 of row i: A fragment of row i + 2; behind MFMA 3 of rows 2, 3: rows 0, 1 of the NEXT stage
    //   behind MFMAs 4, 5 of rows 0-3: next-stage W fragments i, 4 + i
    //   behind MFMAs 2, 6 of rows 0-3: DMA pieces 2 i, 2 i + 1 of the stage's (at most eight)
    // (nothing here depends on mf but the two branches around rows 4-5 and 6-7: conditional register WRITES made hipcc
    //  merge register assignments with 150 moves per loop body, each behind a wait for its LDS read)
#define PW_ROW(i_, TAP_, P_)                                                                                       \
    {                                                                                                              \
        constexpr int i = i_;                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                            \
            if constexpr (i < 2) { PW_MF(i, j, nx[i], wf[P_][j]) } else { PW_MF(i, j, af[(i - 2) % 3], wf[P_][j]) } \
            if constexpr (i < 6) {          /* row i + 2 of this stage (a tile of fewer rows reads it for nothing) */ \
                if (j == 1) {                                                                                      \
                    SB();                                                                                          \
                    PW_RD(af[i % 3], a_off[i + 2][TAP_] + a_cur)                                                   \
                    SB();                                                                                          \
                }                                                                                                  \
            }                                                                                                      \
            if constexpr (i == 2 || i == 3) {   /* rows 0, 1 of the NEXT stage, behind the rows that used their registers */ \
                if (j == 3) {                                                                                      \
                    SB();                                                                                          \
                    PW_RD(nx[i - 2], a_off[i - 2][kTapN] + a_nxt)                                                  \
                    SB();                                                                                          \
                }                                                                                                  \
            }                                                                                                      \
            if constexpr (i < 4) {                                                                                 \
                if (j == 2 || j == 6) {                                                                            \
                    const int n_ = 2 * i + (j == 6 ? 1 : 0);                                                       \
                    if (n_ < kNP) {                                                                                \
                        SB();                                                                                      \
                        if (n_ < kNA) dma16(xr_req, a_dst + (kA0 + n_) * 1024, av[(kA0 + n_) % NPA], c_req * 64);  \
                        else dma16(wrsrc, w_dst + (n_ - kNA) * 1024, wv[(n_ - kNA) & 3], s_req * kWSlot);          \
                        SB();                                                                                      \
                    }                                                                                              \
                }                                                                                                  \
                if (j == 4) {                                                                                      \
                    SB();                                                                                          \
                    PW_RD(wf[(P_) ^ 1][i], w_rd + w_nxt + i * 1024)                                                \
                    SB();                                                                                          \
                }                                                                                                  \
                if (j == 5) {                                                                                      \
                    SB();                                                                                          \
                    PW_RD(wf[(P_) ^ 1][4 + i], w_rd + w_nxt + (4 + i) * 1024)                                      \
                    SB();                                                                                          \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }
#define PW_STAGE(TAP_, P_, FIRST_)                                                                                 \
    {                                                                                                              \
        constexpr bool kFirst = FIRST_;                                                                            \
        constexpr int kTapN = (TAP_) + 1 < NT ? (TAP_) + 1 : 0;             /* the next stage's tap */              \
        constexpr int kNA = NT == 3 ? ((TAP_) == 2 ? 1 : 2) : 4;            /* A pieces of this wave in this stage */ \
        constexpr int kA0 = NT == 3 ? 2 * (TAP_) : 0;                       /* ... starting at this piece of the slab */ \
        constexpr int kNP = kNA + 4;                                        /* + four W pieces */                   \
        const unsigned a_cur = (unsigned)(c & 3) * kASlot;                                                         \
        const int cn = (TAP_) + 1 < NT ? c : c + 1;                                                                \
        const unsigned a_nxt = (unsigned)(cn & 3) * kASlot;                                                        \
        const unsigned w_nxt = (unsigned)((st + 1) & 3) * kWSlot;                                                  \
        /* requests: weights of stage st + 3, activation slab c + 3 (the next tile's behind this one's end; a block's last    \
           tile requests its own first stages again -- nobody reads them -- so that every stage issues the same pieces   \
           and its wait is a constant) */                                                                          \
        const int s_req = st + 3 < nst ? st + 3 : st + 3 - nst;                                                    \
        const int c_req = c + 3 < n_slabs ? c + 3 : c + 3 - n_slabs;                                               \
        const unsigned w_dst = lds0 + (unsigned)((st + 3) & 3) * kWSlot + wave * 4 * 1024;                         \
        const unsigned a_dst = lds0 + kAOff + (unsigned)((c + 3) & 3) * kASlot + wave * NPA * 1024;                \
        i32x4 xr_req = xr_cur;                                                                                     \
        if (c + 3 >= n_slabs) xr_req = xr_next;                                                                    \
        PW_ROW(0, TAP_, P_) PW_ROW(1, TAP_, P_) PW_ROW(2, TAP_, P_) PW_ROW(3, TAP_, P_)                            \
        if (mf > 4) { PW_ROW(4, TAP_, P_) PW_ROW(5, TAP_, P_) }                                                    \
        if (mf > 6) { PW_ROW(6, TAP_, P_) PW_ROW(7, TAP_, P_) }                                                    \
        SB();                                                                                                      \
        /* everything older than this stage's own pieces has landed: the weights of stage st + 2 and slabs up to c + 2 */ \
        if (kNP == 8) { PW_WAIT_VM(8); } else if (kNP == 6) { PW_WAIT_VM(6); } else { PW_WAIT_VM(5); }             \
        PW_WAIT_LGKM();                                                                                            \
        PW_BARRIER()                                                                                               \
        ++st;                                                                                                      \
    }

    int st = 0;
    if constexpr (NT == 3) {
        {   // slabs 0, 1: the tile's first stage starts the accumulators
            int c = 0;
            PW_STAGE(0, 0, true) PW_STAGE(1, 1, false) PW_STAGE(2, 0, false)
            c = 1;
            PW_STAGE(0, 1, false) PW_STAGE(1, 0, false) PW_STAGE(2, 1, false)
        }
        for (int c2 = 2; c2 < n_slabs; c2 += 2) {
            int c = c2;
            PW_STAGE(0, 0, false) PW_STAGE(1, 1, false) PW_STAGE(2, 0, false)
            c = c2 + 1;
            PW_STAGE(0, 1, false) PW_STAGE(1, 0, false) PW_STAGE(2, 1, false)
        }
    } else {
        {
            int c = 0;
            PW_STAGE(0, 0, true)
            c = 1;
            PW_STAGE(0, 1, false)
        }
        for (int c2 = 2; c2 < n_slabs; c2 += 2) {
            int c = c2;
            PW_STAGE(0, 0, false)
            c = c2 + 1;
            PW_STAGE(0, 1, false)
        }
    }
#undef PW_STAGE
#undef PW_ROW
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");        // the last MFMAs' results are in the accumulators

    // ---- epilogue: + bias', ReLU on the packed pair, 16 bytes (eight adjacent channels) per lane and frame
    {
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        const float* cst = reinterpret_cast<const float*>(smem + kConstOff) + wc * 128 + 8 * r;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(cst), b1 = *reinterpret_cast<const f32x4*>(cst + 4);
        const __amdgpu_buffer_rsrc_t yrsrc = make_rsrc(static_cast<char*>(a.Y) + (t.m0 * (int64_t)a.ldy + n0) * 2);
        const int y_voff = (4 * q * a.ldy + wc * 128 + 8 * r) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row16 = 16 * (wr * mf + i);
            SB();               // one frame block at a time: left alone hipcc copies all 256 accumulators out of the AGPRs at once
            if (i < mf && t.m0 + row16 < valid_end) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u32x4 o;
                    o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc[i][0][e] + b0[0], acc[i][1][e] + b0[1]}, bf16x2));
                    o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc[i][2][e] + b0[2], acc[i][3][e] + b0[3]}, bf16x2));
                    o[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc[i][4][e] + b1[0], acc[i][5][e] + b1[1]}, bf16x2));
                    o[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc[i][6][e] + b1[2], acc[i][7][e] + b1[3]}, bf16x2));
#pragma unroll
                    for (int w = 0; w < 4; ++w) asm("v_pk_max_i16 %0, %1, 0" : "=v"(o[w]) : "v"(o[w]));
                    __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, y_voff + e * a.ldy * 2, row16 * a.ldy * 2, 0);
                    asm volatile("s_nop 1" ::"v"(o));          // the 128-bit store's data hazard (tdnn_common.h, store_acc)
                }
            }
        }
    }
}

__device__ __forceinline__ void run_block(const TdnnArgs& a, char* smem) {
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int jcol = lid % a.n_tiles;                   // 256-channel column
    const int prange = lid / a.n_tiles;
    const int64_t u_begin = a.groups_total * (int64_t)prange / a.blocks_per_col;     // 64-frame units
    const int64_t u_end = a.groups_total * (int64_t)(prange + 1) / a.blocks_per_col;
    const int n0 = jcol * 256;
    const int n_slabs = a.cpt * 2;                      // 32-channel slabs per tap (cpt counts 64-element chunks)
    const int nst = n_slabs * a.n_taps;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr)(smem);

    *reinterpret_cast<float*>(smem + kConstOff + tid * 4) = a.bias[n0 + tid];
    const int n = (int)(u_end - u_begin);
    if (n <= 0) return;
    int nt = (n + 3) / 4;
    int base = n / nt, extra = n % nt;
    if (base < 2) { base = 2; extra = 0; nt = (n + 1) / 2; }
    const int64_t range_end = u_end * 64;

    Cursor cur;
    cur.u = __builtin_amdgcn_readfirstlane(utt_of_row(a.out_map, u_begin * 64));
    cur.off_next = next_off(a.out_map, cur.u);
    auto tile_at = [&](int idx, int64_t m0) {
        Tile t;
        t.m0 = m0;
        t.units = idx < extra ? base + 1 : base;
        return t;
    };
    auto src_of = [&](const Tile& t, const Cursor& c) {   // descriptor at the slab's first input row: m0 + u span
        return make_srd(static_cast<const char*>(a.X) + (t.m0 + (int64_t)c.u * a.span) * a.ldx * 2);
    };
    const i32x4 wrsrc = make_srd(static_cast<const char*>(a.W) + (int64_t)jcol * nst * kWSlot);

    Tile t = tile_at(0, u_begin * 64);
    i32x4 xr = src_of(t, cur);
    // prologue: weights of stages 0..2 and slabs 0..2 of the first tile
    {
        const int prow = lane >> 2;
        const int pchunk = ((lane & 3) ^ (3 * (prow >> 3))) << 4;
        const int av = prow * a.ldx * 2 + pchunk, wv = prow * 64 + pchunk;
        const int npa = a.n_taps == 3 ? 5 : 4;
        for (int s = 0; s < 3; ++s) {
            for (int p = 0; p < 4; ++p) dma16(wrsrc, lds0 + s * kWSlot + (wave * 4 + p) * 1024, wv, s * kWSlot + (wave * 4 + p) * 1024);
            for (int p = 0; p < npa; ++p)
                dma16(xr, lds0 + kAOff + s * kASlot + (wave * npa + p) * 1024, av, (wave * npa + p) * 16 * a.ldx * 2 + s * 64);
        }
        PW_WAIT_VM(0);
    }
    __syncthreads();
    for (int idx = 0; idx < nt; ++idx) {
        const bool has_next = idx + 1 < nt;
        Tile nxt = t;
        Cursor cn = cur;
        i32x4 xr_next = xr;
        if (has_next) {
            nxt = tile_at(idx + 1, t.m0 + 64 * t.units);
            seek(a.out_map, cn, nxt.m0);
            xr_next = src_of(nxt, cn);
        }
        if (a.n_taps == 3) process_tile<3>(a, smem, t, xr, xr_next, wrsrc, has_next, range_end, cur, n0, wave, n_slabs, lds0);
        else process_tile<1>(a, smem, t, xr, xr_next, wrsrc, has_next, range_end, cur, n0, wave, n_slabs, lds0);
        t = nxt;
        cur = cn;
        xr = xr_next;
    }
    PW_WAIT_VM(0);              // the last tile's surplus requests have landed before the block's LDS is given away
}

__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void tdnn_pw_kernel(const TdnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    run_block(a, smem);
}

}  // namespace pw

// Shapes the kernel takes: 1 or 3 taps with span = 2 dil (the window argument above), an even number of 64-element chunks
// per tap, 256-channel columns; every utterance must keep at least `min_rows_out` >= 32 output rows (a 256-row tile then meets
// at most eight boundaries: 54 extra slab rows of the 64 a slot has) -- the caller checks that one.
bool tdnn_pw_applicable(const TdnnArgs& a) {
    const bool taps_ok = (a.n_taps == 3 && a.span == 2 * a.tap_rows && a.tap_rows >= 1 && a.tap_rows <= 3) || (a.n_taps == 1 && a.span == 0);
    return taps_ok && a.terms == 1 && a.cpt >= 2 && a.cpt % 2 == 0 && a.n_tiles > 0 && a.groups_total > 0 && a.blocks_per_col > 0 &&
           a.blocks_per_col <= a.groups_total && a.ldx % 8 == 0 && a.ldy % 8 == 0;
}

hipError_t launch_tdnn_pw(const TdnnArgs& a, hipStream_t s) {
    if (!tdnn_pw_applicable(a)) return hipErrorInvalidValue;
    static LdsOptIn opt;
    if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(pw::tdnn_pw_kernel), pw::kLdsBytes); e != hipSuccess) return e;
    pw::tdnn_pw_kernel<<<dim3(a.blocks_per_col * a.n_tiles), dim3(pw::kThreads), pw::kLdsBytes, s>>>(a);
    return hipGetLastError();
}

}  // namespace xvec

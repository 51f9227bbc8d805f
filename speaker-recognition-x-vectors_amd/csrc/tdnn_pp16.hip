// Frame-level TDNN layer, bf16 operands / fp32 accumulation, for the large batches of the bf16 path (BASELINE configs[4]) and,
// with the template flag X3, of bf16x3 (fp32 values as hi + lo bf16 planes).  The same implicit GEMM as
// tdnn_layer.hip (reference tdnn_layer.py:26-41: context gather -> Linear -> ReLU -> eval BatchNorm, optional fused statistics
// pooling, main.py:59-63) on a machine mapping built for the bf16 matrix rate, on v_mfma_f32_16x16x32_bf16.
//
// Why a second mapping.  The 128x128-tile kernel of tdnn_layer.hip moves 512 B from L2 per 32x32x16-equivalent MFMA; at the
// bf16 rate that is ~52 B/clk per CU against the ~64 B/clk the L2 -> CU path delivers, so loads and MFMAs add up instead of
// overlapping.  Here:
//   * ONE 512-thread block per CU, tile = up to 256 frames x 256 channels, K in 64-wide tiles (128-byte rows): half the L2
//     traffic per MFMA.
//   * both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write), 1-KiB pieces
//     of 8 rows x 128 B; the 16-byte-chunk XOR swizzle (row >> 1) & 7 that makes the ds_read_b128 fragment reads conflict-free
//     is applied on the SOURCE address (the LDS image of a DMA piece is lane-linear).  Two 64-KiB LDS buffers (K-tile parity),
//     refilled slot by slot two K-tiles ahead behind COUNTED s_waitcnt vmcnt -- the queue is never drained in the loop.
//   * 8 waves = 2 groups (frames halves) x 4 (64-channel columns); wave tile = MR x 32 frames x 64 channels (MR = 2, 3 or 4 per
//     tile: 128, 192 or 256 frames), per 32-frame acc row 2 frame blocks x 4 channel blocks of 16x16 (4 registers each: frame
//     4 q + e of the block on lane quad q = l >> 4, channel c = l & 15 of the channel block).  Fragments: lane (c, q) reads row
//     16 fb + c of a 32-row block, 16-byte chunk 4 s + q of k-step s (two k-steps of 32 per K-tile).  The two waves of a SIMD
//     belong to different groups and run one barrier apart ("ping-pong"): while one issues the MFMAs of a segment, the other
//     reads its next fragments from LDS and issues its DMA pieces, then they swap.  Two MFMA segments per K-tile.
//   * persistent: a block owns a contiguous range of 64-frame units of one 256-channel column and cuts it into tiles of 4, 3 or
//     2 units, as equal as possible (a partial round of fixed 256-row tiles would idle a quarter of the chip at B = 256).
//   * weights: row 16 cb + c of a wave's 64-channel block holds channel 4 c + cb (pack.hip), so a lane's four accumulators of a
//     frame hold four ADJACENT channels: the store epilogue writes 8 bytes per lane = whole 128-byte row segments of four frames
//     per instruction.
//   * epilogues (they run in the open: both waves of a SIMD are in theirs together).  Plain bf16 stores relu(z + bias') only --
//     the bias is in the accumulators (srcC of a tile's first MFMAs), the BatchNorm is deferred into the consumer's weights
//     (xvec_api.hip, refold), the ReLU is taken on the packed pair (relu_pk_bf16).  Layer 5 (POOL) emits pooling partials per
//     (block, utterance, frames half) instead of its [frames, 1500] output: plain bf16 forms the sums on the MATRIX pipe
//     (SegMx below), bf16x3 on the vector pipe at fp32 (Seg / pool_rows).  bf16x3 applies bias, ReLU and BatchNorm itself and
//     writes the hi and lo planes.
// The next tile's first K-tiles are requested before the epilogue, so the DMA flies under it.
#include "tdnn_common.h"

namespace xvec {
namespace pp16 {

constexpr int kRowB = 128;                        // one K-tile slab of one row: 64 bf16
constexpr int kAccRowB = 32 * kRowB;              // 32 frames: 4 KiB = 4 DMA pieces
constexpr int kABytes = 2 * 4 * kAccRowB;         // [group][acc row][32 frames]: 32 KiB
constexpr int kWBytes = 256 * kRowB;              // 256 channels: 32 KiB = 32 DMA pieces
constexpr int kBufBytes = kABytes + kWBytes;      // 64 KiB per K-tile buffer (K-tile parity); LDS image: A0 | A1 | W0 | W1, so
constexpr int kWOff = 2 * kABytes;                // that BOTH buffers of an operand lie within the 64 KiB reach of a ds_read's
constexpr int kConstOff = 2 * kBufBytes;          // offset field from one base register (A0 | W0 | A1 | W1 cost four more)
constexpr int kConstBytes = 3 * 256 * 4;          // bias | scale | shift of the block's 256 channels, natural order
constexpr int kParkOff = kConstOff + kConstBytes; // pooling variant: 48 bytes per thread (pivot | S1 | S2 of the lane's four
constexpr int kParkBytes = 512 * 48;              // channels), the running sums of a wave's current utterance between two epilogues
constexpr int kLdsBytes = kParkOff + kParkBytes;
constexpr int kThreads = 512;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ i32x4 make_srd(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    i32x4 d;
    d.x = (int)__builtin_amdgcn_readfirstlane((unsigned)v);
    d.y = (int)(__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) & 0xffffu);   // stride 0
    d.z = 0x7fffffff;                                                            // num_records (bytes)
    d.w = 0x00020000;
    return d;
}

// One DMA piece: 64 lanes x 16 B from per-lane source offsets to 1 KiB of LDS at `dst` (wave
// uniform).  Inline asm on purpose: hipcc would wait vmcnt(0) for the builtin form before the
// next ds_read; this way the pieces are invisible to its bookkeeping and are waited for by hand
// (counted vmcnt before the barrier that publishes them).
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned dst, int voff, int soff) {
#ifdef XVEC_DIAG   // diagnostic build (more scalar pressure): keep a folded constant out of soffset and the descriptor in SGPRs
    asm volatile("s_mov_b32 %0, %1" : "=s"(soff) : "s"(soff));
    i32x4 rs_;
    rs_.x = __builtin_amdgcn_readfirstlane(rsrc.x); rs_.y = __builtin_amdgcn_readfirstlane(rsrc.y);
    rs_.z = __builtin_amdgcn_readfirstlane(rsrc.z); rs_.w = __builtin_amdgcn_readfirstlane(rsrc.w);
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(voff), "s"(rs_), "s"(soff)
        : "memory", "m0");
    return;
#endif
    // dst / soff / rsrc are SALU results (no VALU-written SGPR feeds the load: no wait states needed
    // beyond the one after the M0 write); M0 is declared clobbered instead of saved and restored
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff)
        : "memory", "m0");
}

#ifdef XVEC_DIAG
// Diagnostic build only (make DIAG=1): s_memtime stamps of one wave per group, summed per segment kind.
// slot = 16 * group + kind; kinds: 0-5 phase 0 (issue, wait, barrier, mfma, barrier, -), 6-11 phase 1
__device__ unsigned long long g_pp16_diag[2 * 512 * 32];   // [pooling variant][block][group][kind]
__device__ unsigned long long g_pp16_clk[8];               // block 0: s_memtime / s_memrealtime at entry and exit, per variant
#if XVEC_DIAG == 2      // make DIAG=2: anatomy of the pooling epilogue only (kinds 0-5), K-loop stamps off
#define PP_STAMP(k_)
#define PP_ESTAMP(k_) PP_STAMP_DO(k_)
#else
#define PP_STAMP(k_) PP_STAMP_DO(k_)
#define PP_ESTAMP(k_)
#endif
#define PP_STAMP_DO(k_)                                                                            \
    {                                                                                              \
        SB();                                                                                      \
        unsigned long long now_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");              \
        dsum[k_] += now_ - dprev;                                                                  \
        dprev = now_;                                                                              \
        SB();                                                                                      \
    }
#else
#define PP_STAMP(k_)
#define PP_ESTAMP(k_)
#endif
#ifdef XVEC_KNOCK
// Timing-only knock-outs, compile time (-DXVEC_KNOCK=mask; results are garbage):
//   bit 0: no DMA pieces in the K loop     bit 1: no LDS fragment reads (fragments stay zero)
//   bit 2: no epilogue (stores / pooling)
#define PP_KNOCK_DMA ((XVEC_KNOCK & 1) != 0)
#define PP_KNOCK_RD ((XVEC_KNOCK & 2) != 0)
#define PP_KNOCK_EPI ((XVEC_KNOCK & 4) != 0 || ((XVEC_KNOCK & 2048) != 0 && POOL))   // bit 11: no epilogue in the pooling variant only (layers 2-4 keep theirs, so layer 5 still runs on real data)
#define PP_KNOCK_RDW ((XVEC_KNOCK & 8) != 0)     // bit 3: no W fragment reads only
#define PP_KNOCK_RDA ((XVEC_KNOCK & 16) != 0)    // bit 4: no A fragment reads only
#define PP_KNOCK_MASKED ((XVEC_KNOCK & 32) != 0) // bit 5: pooling epilogue without its masked path (code-size experiment)
#define PP_KNOCK_PARK ((XVEC_KNOCK & 64) != 0)   // bit 6: pooling epilogue without the LDS round trip of the running sums
#define PP_KNOCK_ROWS ((XVEC_KNOCK & 128) != 0)  // bit 7: pooling epilogue on acc row 0 only
#define PP_KNOCK_MXMF ((XVEC_KNOCK & 256) != 0)  // bit 8: matrix-pipe pooling without its MFMAs
#define PP_KNOCK_MXPK ((XVEC_KNOCK & 512) != 0)  // bit 9: ... without the packing (max / cvt / mask) of the deviations
#define PP_KNOCK_MXROWS ((XVEC_KNOCK & 1024) != 0) // bit 10: ... without the groups (restore / fold / park only)
#define PP_KNOCK_MXFOLD ((XVEC_KNOCK & 4096) != 0) // bit 12: ... without the fold + park at the end of a tile
#define PP_KNOCK_ATAP ((XVEC_KNOCK & 8192) != 0)   // bit 13: activation pieces requested for tap 0 only (what sharing one slab between a layer's taps would save)
#define PP_KNOCK_W2 ((XVEC_KNOCK & 16384) != 0)    // bit 14: every weight piece requested TWICE, counted waits adjusted (cost side of two frame halves with their own weight K-tiles)
#define PP_KNOCK_EPI4 ((XVEC_KNOCK & 32768) != 0)  // bit 15: no store epilogue in the single-tap store variant only (layer 4; layers 2-3 keep theirs, so it runs on real data)
#define PP_KNOCK_A5 ((XVEC_KNOCK & 65536) != 0)    // bit 16: no activation pieces in the pooling variant (layer 5 fed from LDS by a fused layer 4)
#else
#define PP_KNOCK_W2 false
#define PP_KNOCK_EPI4 false
#define PP_KNOCK_A5 false
#define PP_KNOCK_ATAP false
#define PP_KNOCK_MXFOLD false
#define PP_KNOCK_MXMF false
#define PP_KNOCK_MXPK false
#define PP_KNOCK_MXROWS false
#define PP_KNOCK_MASKED false
#define PP_KNOCK_PARK false
#define PP_KNOCK_ROWS false
#define PP_KNOCK_RDW false
#define PP_KNOCK_RDA false
#define PP_KNOCK_DMA false
#define PP_KNOCK_RD false
#define PP_KNOCK_EPI false
#endif
// lgkmcnt(0) as the BUILTIN (0xC07F = lgkmcnt 0, vmcnt / expcnt untouched): hipcc's wait-count pass sees it and
// knows every earlier LDS read is back.  As inline asm it did not, and put lgkmcnt(3..0) waits for
// fragments read a segment earlier in front of the MFMAs -- behind the freshly issued prefetch reads,
// which serialised those reads with the MFMAs they were meant to hide under.
#define PP_WAIT_LGKM()                          \
    {                                           \
        SB();                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);     \
        SB();                                   \
    }
#define PP_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")
// counted wait whose count is only known at run time (it depends on the next tile's height): one of the few
// values the request schedule can produce; anything else waits for everything (stricter, never wrong)
#define PP_WAIT_VM_RT(n_)                                        \
    {                                                            \
        const int nn_ = (n_);                                    \
        if (nn_ == 10) { PP_WAIT_VM(10); }                       \
        else if (nn_ == 9) { PP_WAIT_VM(9); }                    \
        else if (nn_ == 8) { PP_WAIT_VM(8); }                    \
        else if (nn_ == 7) { PP_WAIT_VM(7); }                    \
        else if (nn_ == 6) { PP_WAIT_VM(6); }                    \
        else if (nn_ == 2) { PP_WAIT_VM(2); }                    \
        else if (nn_ == 1) { PP_WAIT_VM(1); }                    \
        else { PP_WAIT_VM(0); }                                  \
    }
#define PP_BARRIER() \
    {                \
        SB();        \
        __builtin_amdgcn_s_barrier(); \
        SB();        \
    }

// Tile = `mr` accumulator rows per group (rows m0 .. m0 + 64*mr), of which rows below `valid_end`
// belong to this block.
struct Tile {
    int64_t m0;
    int64_t valid_end;
    int mr;
};

// activation source of one tile: descriptor at its first row + this lane's byte offsets of the wave's A piece of
// acc rows 0..3
struct Rows {
    i32x4 xrsrc;
    int av0, av1, av2, av3;
};

struct Stream {
    i32x4 wrsrc;
    Rows cur;                   // tile the requests are for
    int wv0;                    // and of its first W piece; the others are 64 channel rows (w64 bytes, scalar) apart
    int w64;
    unsigned lds_a, lds_w;      // LDS byte address (buffer 0) of this wave's A piece of acc row 0 / its first W piece
    int u_tile;                 // utterance holding the stream tile's first row, and where the next one starts
    int64_t off_next;
};

// per-lane source offsets of the wave's A pieces for the tile at row t.m0 (see set_tile_rows_impl in
// tdnn_layer.hip: compact output row p of utterance u reads input rows p + u*span; the utterance
// boundaries inside the tile are walked with block-uniform values, each lane counts the ones its
// rows have passed)
// (this lane's row within its group's acc row 0 and the swizzled 16-byte chunk it fetches are recomputed from the lane
// id here, once per tile: held in registers across the K loop they were the three registers over the budget)
template <bool RAGGED>
__device__ __forceinline__ void set_rows(const TdnnArgs& a, const Tile& t, int grp, int wc, Stream& st, Rows& out) {
    int lane_;            // volatile asm: as a builtin the compiler hoists it out of the tile loop and keeps it, again
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
    const int row_in_group = wc * 8 + (lane_ >> 3);
    const int a_chunk = ((lane_ & 7) ^ ((row_in_group >> 1) & 7)) * 16;
    const int n_last = a.out_map.n_utts - 1;
    const int64_t t_out = a.out_map.fixed_T - a.out_map.cum;
    auto next_off = [&](int u) -> int64_t {
        if (RAGGED) return sload_i64(a.out_map.offsets + __builtin_amdgcn_readfirstlane(u + 1)) - (int64_t)(u + 1) * a.out_map.cum;
        return (int64_t)(u + 1) * t_out;
    };
    while (t.m0 >= st.off_next && st.u_tile < n_last) {
        st.u_tile = __builtin_amdgcn_readfirstlane(st.u_tile + 1);
        st.off_next = next_off(st.u_tile);
    }
    const int rl = grp * 32 * t.mr + row_in_group;     // row of this lane's acc-row-0 piece, relative to m0
    const int64_t p = t.m0 + rl;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int u = st.u_tile;
    int64_t nxt = st.off_next;
    const int64_t t_end = t.m0 + 64 * t.mr;
    while (nxt < t_end && u < n_last) {
        c0 += (p >= nxt) ? 1 : 0;
        c1 += (p + 32 >= nxt) ? 1 : 0;
        c2 += (p + 64 >= nxt) ? 1 : 0;
        c3 += (p + 96 >= nxt) ? 1 : 0;
        u = __builtin_amdgcn_readfirstlane(u + 1);
        nxt = next_off(u);
    }
    const int rb = a.ldx * 2;
    const int base = rl * rb + a_chunk;
    out.av0 = base + (st.u_tile + c0) * a.span * rb;
    out.av1 = base + 32 * rb + (st.u_tile + c1) * a.span * rb;
    out.av2 = base + 64 * rb + (st.u_tile + c2) * a.span * rb;
    out.av3 = base + 96 * rb + (st.u_tile + c3) * a.span * rb;
    out.xrsrc = make_srd(static_cast<const char*>(a.X) + t.m0 * (int64_t)a.ldx * 2);
}

__device__ __forceinline__ void set_rows(const TdnnArgs& a, const Tile& t, int grp, int wc, Stream& st, Rows& out) {
    if (a.out_map.offsets == nullptr) set_rows<false>(a, t, grp, wc, st, out);
    else set_rows<true>(a, t, grp, wc, st, out);
}

// Scalar source offset of the activation K-tiles, stepped one K-tile at a time (taps innermost:
// tap 0, 1, .., then the next 64-channel block): no division in the loop.
struct KPos {
    int tap, so, ph;
};
// X3 (bf16x3: fp32 values carried as two bf16 planes hi + lo, x*W ~ x_hi*W_hi + x_hi*W_lo + x_lo*W_hi): every 64-channel
// slab of a tap is THREE K-tiles -- the hi slab against W_hi, the hi slab again against W_lo, the lo slab (x_plane_bytes
// further on: a scalar offset, the lanes' offsets do not change) against W_hi -- and the packed weights hold the
// matching sequence W_hi | W_lo | W_hi per slab (pack.hip).  The K loop itself does not know: same tiles, same schedule,
// three times as many K-tiles.
template <bool X3>
__device__ __forceinline__ void kstep(const TdnnArgs& a, KPos& k) {
    if constexpr (X3) {
        if (k.ph == 0) { k.ph = 1; return; }
        if (k.ph == 1) { k.ph = 2; k.so += a.x_plane_bytes; return; }
        k.ph = 0;
        k.so -= a.x_plane_bytes;
    }
    const int tapstep = a.tap_rows * a.ldx * 2;
    if (k.tap + 1 < a.n_taps) {
        k.tap += 1;
        k.so += tapstep;
    } else {
        k.so += 128 - k.tap * tapstep;
        k.tap = 0;
    }
}

// --- DMA piece groups of one wave (buffer b_ = parity of the K-tile) -------------------------
// W: the wave's four pieces (channel rows 8*(wave + 8t) ..+7); A01 / A23: its piece of acc rows 0,1 / 2,(3)
#define PP_ISSUE_W(b_, q_)                                                          \
    {                                                                               \
        const int so_ = (q_) * kWBytes;              /* K-tile major weights: 32 KiB per K-tile */ \
        dma16(st.wrsrc, st.lds_w + (b_) * kWBytes, st.wv0, so_);                  \
        dma16(st.wrsrc, st.lds_w + (b_) * kWBytes + 8 * 1024, st.wv0, so_ + st.w64);       \
        dma16(st.wrsrc, st.lds_w + (b_) * kWBytes + 16 * 1024, st.wv0, so_ + 2 * st.w64);  \
        dma16(st.wrsrc, st.lds_w + (b_) * kWBytes + 24 * 1024, st.wv0, so_ + 3 * st.w64);  \
    }
#define PP_ISSUE_A01(b_, so_)                                                       \
    {                                                                               \
        dma16(st.cur.xrsrc, st.lds_a + (b_) * kABytes, st.cur.av0, so_);                  \
        dma16(st.cur.xrsrc, st.lds_a + (b_) * kABytes + kAccRowB, st.cur.av1, so_);       \
    }
#define PP_ISSUE_A23(MR_, b_, so_)                                                  \
    {                                                                               \
        if ((MR_) > 2) dma16(st.cur.xrsrc, st.lds_a + (b_) * kABytes + 2 * kAccRowB, st.cur.av2, so_);   \
        if ((MR_) > 3) dma16(st.cur.xrsrc, st.lds_a + (b_) * kABytes + 3 * kAccRowB, st.cur.av3, so_); \
    }

// first K-tiles of a tile: part 1 = all of K-tile 0; part 2 = all of K-tile 1, in the K loop's request order
// (acc rows 0,1, then W, then acc rows 2,3), which keeps the loop's counted waits uniform from the first K-tile
template <bool POOL>
__device__ __forceinline__ void issue_head1(const TdnnArgs& a, const Stream& st, int mr) {
    PP_ISSUE_W(0, 0)
    PP_ISSUE_A01(0, 0)
    PP_ISSUE_A23(mr, 0, 0)
}
template <bool POOL, bool X3>
__device__ __forceinline__ void issue_head2(const TdnnArgs& a, const Stream& st, int mr) {
    KPos k1 = {0, 0, 0};
    kstep<X3>(a, k1);
    PP_ISSUE_A01(1, k1.so)
    PP_ISSUE_W(1, 1)
    PP_ISSUE_A23(mr, 1, k1.so)
}

struct Lane {
    int q, r;        // lane quad (l >> 4): k chunk of a fragment, frames 4q..4q+3 of an accumulator; r = l & 15
    int rd;          // r*128: row part of every fragment read
    int k0, k1;      // swizzled byte offset of this lane's 16-byte chunk for k-steps 0, 1
    unsigned a_rd;   // LDS byte offset (buffer 0) of this wave's group's acc row 0, + rd
    unsigned w_rd;   // LDS byte offset (buffer 0) of this wave's channel column 0, + rd
    // the four base registers of every fragment read (operand x k-step), made opaque to the compiler: left to itself it
    // re-associates the address sums around ONE base and then materialises a dozen "base + constant" registers for the
    // constants that do not fit a ds_read's 16-bit offset field
    unsigned a_k0, a_k1, w_k0, w_k1;
    int wave, grp, wc;
};

#define PP_RD(dst_, off_) if constexpr (!PP_KNOCK_RD) dst_ = *reinterpret_cast<const float4*>(smem + (off_));
#define PP_RDW(dst_, off_) if constexpr (!PP_KNOCK_RDW) PP_RD(dst_, off_)
constexpr int kBlk16 = 16 * kRowB;                // 16 rows of a 32-row block: 2 KiB
// W fragments of K-tile in buffer b_: 4 channel blocks x 2 k-steps
#define PP_READ_W(b_)                                                     \
    {                                                                     \
        constexpr unsigned o_ = (b_) * kWBytes;                           \
        PP_RDW(wf0_0, ln.w_k0 + o_) PP_RDW(wf0_1, ln.w_k1 + o_)               \
        PP_RDW(wf1_0, ln.w_k0 + o_ + kBlk16) PP_RDW(wf1_1, ln.w_k1 + o_ + kBlk16)          \
        PP_RDW(wf2_0, ln.w_k0 + o_ + 2 * kBlk16) PP_RDW(wf2_1, ln.w_k1 + o_ + 2 * kBlk16)  \
        PP_RDW(wf3_0, ln.w_k0 + o_ + 3 * kBlk16) PP_RDW(wf3_1, ln.w_k1 + o_ + 3 * kBlk16)  \
    }
// one MFMA: accumulator (acc row i_, frame block f_, channel block c_), k-step s_.  The activations are the MFMA A
// operand: frames in the accumulator's registers (frame 4q + e), the channel on the lane.
// Inline asm with the accumulator tied ("+v"): hipcc has no tied form of the 4-pass MFMAs (vdst may be any register),
// its allocator let the 32 accumulators wander from MFMA to MFMA and ended 50 registers over the budget (spills that
// wait on vmcnt(0) in the K loop).  What hipcc would otherwise look after, by hand: operands written by ds_read are
// waited for by the compiler (it sees the asm's register uses); the same accumulator is never used by two MFMAs less than
// four MFMAs apart; PP_MFMA_SETTLE() stands between the accumulators' initialisation / the K loop's last MFMA and
// the vector instructions that write / read them.
#define PP_MF(i_, f_, c_, s_) PP_MF_S##s_(i_, f_, c_)
#define PP_MF_S1(i_, f_, c_)                                                                                        \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc##i_##f_##c_) : "v"(__builtin_bit_cast(f32x4, af##i_##_##f_##1)), "v"(__builtin_bit_cast(f32x4, wf##c_##_1)));
// k-step 0: in a tile's FIRST K-tile (kt_first, a constant of the enclosing PP_KTILE) the MFMA takes the bias of its
// four-channel block as srcC and only WRITES the accumulator: the accumulators are never initialised (128 v_mov per
// wave and tile, in the open at the head of every tile: 1 k cycles of each SIMD per tile, 3.5 % of a K = 512 tile)
// (bf16x3 starts at ZERO and adds the bias in the epilogue, as the fp32 kernel does: on top of a large bias every MFMA's
//  sum would be rounded at the bias's ulp, and the hi*lo / lo*hi terms -- 2^-9 of the product -- mostly lost: seen as 2e-4
//  on the pooled standard deviations of a channel with |mean|/std = 4000, where the bar is 1e-4)
#define PP_MF_S0(i_, f_, c_)                                                                                        \
    if constexpr (kt_first && X3)                                                                                   \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc##i_##f_##c_) : "v"(__builtin_bit_cast(f32x4, af##i_##_##f_##0)), "v"(__builtin_bit_cast(f32x4, wf##c_##_0))); \
    else if constexpr (kt_first)                                                                                    \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc##i_##f_##c_) : "v"(__builtin_bit_cast(f32x4, af##i_##_##f_##0)), "v"(__builtin_bit_cast(f32x4, wf##c_##_0)), "v"(bias4_##c_)); \
    else                                                                                                            \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc##i_##f_##c_) : "v"(__builtin_bit_cast(f32x4, af##i_##_##f_##0)), "v"(__builtin_bit_cast(f32x4, wf##c_##_0)));
#define PP_MFMA_SETTLE() asm volatile("s_nop 15\n\ts_nop 3" ::: "memory")
// half a quad / a quad: (acc row, frame block, k-step) x channel blocks 0,1 / 2,3 / all four -- 2 x / 4 x 16 cycles
#define PP_H0(i_, f_, s_) PP_MF(i_, f_, 0, s_) PP_MF(i_, f_, 1, s_) SB();
#define PP_H1(i_, f_, s_) PP_MF(i_, f_, 2, s_) PP_MF(i_, f_, 3, s_) SB();
#define PP_Q(i_, f_, s_) PP_H0(i_, f_, s_) PP_H1(i_, f_, s_)
// one activation fragment read: acc row i_, frame block f_, k-step s_, from buffer b_.  A ds_read_b128 holds the
// wave's issue for ~30 cycles (stamps, profiles/diag/pp_stamps.py), which is free exactly while MFMAs of this wave
// are executing: ONE read behind every two MFMAs (32 cycles), never two in a row
#define PP_RA(i_, f_, s_, b_) if constexpr (!PP_KNOCK_RDA) PP_RD(af##i_##_##f_##s_, ln.a_k##s_ + ((b_) * kABytes + (i_) * kAccRowB + (f_) * kBlk16)) SB();
// half a quad followed by one read
#define PP_H0R(i_, f_, s_, ri_, rf_, rs_, b_) PP_H0(i_, f_, s_) PP_RA(ri_, rf_, rs_, b_)
#define PP_H1R(i_, f_, s_, ri_, rf_, rs_, b_) PP_H1(i_, f_, s_) PP_RA(ri_, rf_, rs_, b_)

// One K-tile held in LDS buffer b_ (odd_ = its parity).  k2.so / wq = activation source offset and index of
// the K-tile requested now, two K-tiles ahead; mr_req = height of the tile it belongs to.  The request stream
// does not stop at the end of a tile: in a tile's last two K-tiles ("last") the requests are the NEXT tile's
// K-tiles 0 and 1 (the stream state was switched to that tile just before), so a tile starts with its first
// K-tiles in LDS and its first fragments in registers.  Only a block's last tile requests nothing there.
//   load 0:  read the 8 W fragments; request acc rows 0,1 of K-tile q+2
//   mfma 0:  MR=4: acc rows 0,1 (32 MFMAs of 16 cycles); MR=3: six of the twelve (row, frame block, k-step) quads
//            of rows 0,1,2 (24); MR=2: k-step 0 (16) -- with the fragment reads of acc rows 2,(3) behind its first MFMAs
//   load 1:  request W and acc rows 2,(3) of K-tile q+2
//   mfma 1:  the other half -- with the fragment reads of acc rows 0,1 of K-tile q+1 behind its first MFMAs, each
//            after the last MFMA that uses the register it overwrites
// The two segments of a wave are equally long, so the SIMD partner's load segments have the same time to
// hide in.  Activation fragments are never read in a load segment, and never two reads behind one MFMA.
// Every segment ends with lgkmcnt(0) before its barrier (a slot may be refilled in any later segment).
// Counted vmcnt at the end of a load segment (request order per wave: [rows 0,1] | [W, rows 2,3] | ...; m = this
// tile's MR, m' = mr_req, equal except in the last two K-tiles):
//   load 0 must have acc rows 2,3 of K-tile q  (requested three load segments ago): 2 + (m'+2) + 2 younger
//   load 1 must have W of K-tile q+1 (two load segments ago) and rows 0,1 of q+1 (three): (m1-2) + 2 + (m'+2)
//          younger, m1 = height of the tile K-tile q+1 belongs to
// -- never a drain.  (The epilogue's stores sit in the same queue: the first waits of the next tile then
// wait for a few entries more than they need to, which have long completed.)
#define PP_KTILE(b_, odd_, first_)                                                  \
    {                                                                               \
        constexpr bool kt_first = first_;                                           \
        SB();                                                                       \
        PP_READ_W(b_)                                                               \
        SB();                                                                       \
        if (req && !PP_KNOCK_DMA && !(PP_KNOCK_A5 && POOL) && !(PP_KNOCK_ATAP && k2.tap != 0)) PP_ISSUE_A01(b_, k2.so) \
        SB();                                                                       \
        PP_WAIT_LGKM();                                                             \
        PP_STAMP(0)                                                                 \
        if (PP_KNOCK_A5 && POOL) { }   /* only weight pieces in the queue: nothing of this K-tile is outstanding here */ \
        else if (PP_KNOCK_W2 && !last) { if (MR == 4) { PP_WAIT_VM(14); } else if (MR == 3) { PP_WAIT_VM(13); } else { PP_WAIT_VM(12); } } \
        else if (!last) { if (MR == 4) { PP_WAIT_VM(10); } else if (MR == 3) { PP_WAIT_VM(9); } else { PP_WAIT_VM(8); } } \
        else if (!(odd_)) { PP_WAIT_VM_RT(req ? MR + 6 : MR + 4) }                  \
        else { PP_WAIT_VM_RT(req ? mr_req + 6 : 0) }                                \
        PP_STAMP(1)                                                                 \
        PP_BARRIER()                                                                \
        PP_STAMP(2)                                                                 \
        __builtin_amdgcn_s_setprio(1);                                              \
        if constexpr (MR == 2) {   /* k-step 0 of rows 0,1: nothing to read (no acc row 2) */ \
            PP_Q(0, 0, 0) PP_Q(0, 1, 0) PP_Q(1, 0, 0) PP_Q(1, 1, 0)                 \
        } else if constexpr (MR == 3) {                                             \
            PP_H0R(0, 0, 0, 2, 0, 0, b_) PP_H1R(0, 0, 0, 2, 1, 0, b_)               \
            PP_H0R(0, 1, 0, 2, 0, 1, b_) PP_H1R(0, 1, 0, 2, 1, 1, b_)               \
            PP_Q(1, 0, 0) PP_Q(1, 1, 0) PP_Q(0, 0, 1) PP_Q(0, 1, 1)                 \
        } else {                                                                    \
            PP_H0R(0, 0, 0, 2, 0, 0, b_) PP_H1R(0, 0, 0, 2, 1, 0, b_)               \
            PP_H0R(0, 1, 0, 2, 0, 1, b_) PP_H1R(0, 1, 0, 2, 1, 1, b_)               \
            PP_H0R(1, 0, 0, 3, 0, 0, b_) PP_H1R(1, 0, 0, 3, 1, 0, b_)               \
            PP_H0R(1, 1, 0, 3, 0, 1, b_) PP_H1R(1, 1, 0, 3, 1, 1, b_)               \
            PP_Q(0, 0, 1) PP_Q(0, 1, 1) PP_Q(1, 0, 1) PP_Q(1, 1, 1)                 \
        }                                                                           \
        __builtin_amdgcn_s_setprio(0);                                              \
        PP_WAIT_LGKM();                                                             \
        PP_STAMP(3)                                                                 \
        PP_BARRIER()                                                                \
        PP_STAMP(4)                                                                 \
        if (req && !PP_KNOCK_DMA) {                                                 \
            PP_ISSUE_W(b_, wq)                                                      \
            if (PP_KNOCK_W2) PP_ISSUE_W(b_, wq)                                     \
            if (!(PP_KNOCK_A5 && POOL) && !(PP_KNOCK_ATAP && k2.tap != 0)) PP_ISSUE_A23(mr_req, b_, k2.so) \
        }                                                                           \
        SB();                                                                       \
        PP_STAMP(6)                                                                 \
        if (PP_KNOCK_A5 && POOL) { if (req) { PP_WAIT_VM(4); } else { PP_WAIT_VM(0); } }   /* W of K-tile q+1 behind the four pieces just requested */ \
        else if (PP_KNOCK_W2 && !last) { if (MR == 4) { PP_WAIT_VM(14); } else if (MR == 3) { PP_WAIT_VM(12); } else { PP_WAIT_VM(10); } } \
        else if (!last) { if (MR == 4) { PP_WAIT_VM(10); } else if (MR == 3) { PP_WAIT_VM(8); } else { PP_WAIT_VM(6); } } \
        else if (!(odd_)) { PP_WAIT_VM_RT(req ? MR + mr_req + 2 : MR - 2) }         \
        else { PP_WAIT_VM_RT(req ? 2 * mr_req + 2 : 0) }                            \
        PP_STAMP(7)                                                                 \
        PP_BARRIER()                                                                \
        PP_STAMP(8)                                                                 \
        __builtin_amdgcn_s_setprio(1);                                              \
        /* (after a block's last K-tile these reads fetch stale bytes that nobody uses: cheaper than a branch */ \
        /*  around the MFMAs, which made hipcc keep two register assignments alive and spill) */ \
        if constexpr (MR == 2) {   /* k-step 1; the k-step 0 fragments are free, the k-step 1 ones after their quad */ \
            PP_H0R(0, 0, 1, 0, 0, 0, (b_) ^ 1) PP_H1R(0, 0, 1, 0, 1, 0, (b_) ^ 1)   \
            PP_H0R(0, 1, 1, 1, 0, 0, (b_) ^ 1) PP_H1R(0, 1, 1, 1, 1, 0, (b_) ^ 1)   \
            PP_H0R(1, 0, 1, 0, 0, 1, (b_) ^ 1) PP_H1R(1, 0, 1, 0, 1, 1, (b_) ^ 1)   \
            PP_H0(1, 1, 1) PP_H1R(1, 1, 1, 1, 0, 1, (b_) ^ 1)                       \
            PP_RA(1, 1, 1, (b_) ^ 1)                                                \
        } else if constexpr (MR == 3) {   /* row 1 k-step 1, then row 2; row 0 is free, row 1 after its two quads */ \
            PP_H0R(1, 0, 1, 0, 0, 0, (b_) ^ 1) PP_H1R(1, 0, 1, 0, 1, 0, (b_) ^ 1)   \
            PP_H0R(1, 1, 1, 0, 0, 1, (b_) ^ 1) PP_H1R(1, 1, 1, 0, 1, 1, (b_) ^ 1)   \
            PP_H0R(2, 0, 0, 1, 0, 0, (b_) ^ 1) PP_H1R(2, 0, 0, 1, 1, 0, (b_) ^ 1)   \
            PP_H0R(2, 1, 0, 1, 0, 1, (b_) ^ 1) PP_H1R(2, 1, 0, 1, 1, 1, (b_) ^ 1)   \
            PP_Q(2, 0, 1) PP_Q(2, 1, 1)                                             \
        } else {                                                                    \
            PP_H0R(2, 0, 0, 0, 0, 0, (b_) ^ 1) PP_H1R(2, 0, 0, 0, 1, 0, (b_) ^ 1)   \
            PP_H0R(2, 1, 0, 0, 0, 1, (b_) ^ 1) PP_H1R(2, 1, 0, 0, 1, 1, (b_) ^ 1)   \
            PP_H0R(3, 0, 0, 1, 0, 0, (b_) ^ 1) PP_H1R(3, 0, 0, 1, 1, 0, (b_) ^ 1)   \
            PP_H0R(3, 1, 0, 1, 0, 1, (b_) ^ 1) PP_H1R(3, 1, 0, 1, 1, 1, (b_) ^ 1)   \
            PP_Q(2, 0, 1) PP_Q(2, 1, 1) PP_Q(3, 0, 1) PP_Q(3, 1, 1)                 \
        }                                                                           \
        __builtin_amdgcn_s_setprio(0);                                              \
        PP_WAIT_LGKM();                                                             \
        PP_STAMP(9)                                                                 \
        PP_BARRIER()                                                                \
        PP_STAMP(10)                                                                \
        kstep<X3>(a, k2);                                                           \
        ++wq;                                                                       \
    }

// Fused statistics pooling of the pooling variant (main.py:59-63), VECTOR-pipe form (bf16x3; plain bf16 uses the matrix-pipe
// form, SegMx below): frames in the accumulator registers, the channel on the lane.  The epilogue runs in the open here (both
// waves of a SIMD are in it at the same time, the matrix pipe idles): per (segment of an utterance, channel) the pivoted sums
// S1 = sum (r - K), S2 = sum (r - K)^2 of r = relu(z + bias), K = frame 0 of the group in which the segment's first rows fall
// (tdnn_common.h, pool_group_impl: why a pivot) -- one v_max, one v_sub, one v_add and one v_fma per value.  Scale and shift of
// the folded BatchNorm are applied by pool_finalize.
// RAGGED is a template parameter and the utterance index is kept provably wave-uniform on purpose: with
// a run-time "offsets ? load : multiply" hipcc emitted VECTOR loads of the offsets followed by
// s_waitcnt vmcnt(0) -- on the fixed-length path too -- and every one of those waits drained the DMA
// queue (the next tile's first K-tiles) in the middle of the epilogue.
template <bool RAGGED>
__device__ __forceinline__ int64_t first_row(const RowMap& m, int u) {
    u = __builtin_amdgcn_readfirstlane(u);
    if (RAGGED) return sload_i64(m.offsets + u) - (int64_t)u * m.cum;
    return (int64_t)u * (m.fixed_T - m.cum);
}
// max(x, 0) as ONE instruction.  fmaxf() on a value hipcc cannot prove canonical (the accumulators come out of inline asm)
// becomes v_max_f32 t, x, x; v_max_f32 r, 0, t -- a quieting pass plus the max, the second waiting for the first: twice
// the instructions of the epilogues' ReLU, in dependent pairs.  (A NaN gives 0 here, as fmaxf does.)
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// ReLU of two packed bf16 values in ONE instruction: as 16-bit integers, negative floats (and -0) are negative and everything
// else keeps its order, so max(x, 0) per half is the ReLU.  (A NaN with the sign bit set gives 0, one without stays a NaN.)
__device__ __forceinline__ unsigned relu_pk_bf16(unsigned pk) {
    unsigned r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(pk));
    return r;
}

// --- cross-lane helpers for the 16x16 accumulator layout (frames on the four lane quads q = l >> 4) --------------
// {a summed over the two lane halves | b summed over the two lane halves}: v_permlane32_swap(a, b) returns
// {a.lo | b.lo, a.hi | b.hi}; their sum holds a's total in the lower half and b's in the upper half
__device__ __forceinline__ float swap32_add(float x, float y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the same one level down: v_permlane16_swap(a, b) returns {a.r0 | b.r0 | a.r2 | b.r2, a.r1 | b.r1 | a.r3 | b.r3} (rows of 16
// lanes); the sum holds a.r0+a.r1 | b.r0+b.r1 | a.r2+a.r3 | b.r2+b.r3
__device__ __forceinline__ float swap16_add(float x, float y) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// quad 0's x in every lane
__device__ __forceinline__ float quad0(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // r[0] = {r0 | r0 | r2 | r2}
    return lower_half(__uint_as_float(r[0]));
}

// ---- fused statistics pooling, accumulated per UTTERANCE SEGMENT ------------------------------------------------
// A block owns a contiguous row range of its column (~27 units of 64 frames at the bench batch: six utterances), and a
// wave sees the rows of its group (frames half) of every tile in increasing order.  So a wave keeps the running sums
// of its CURRENT utterance -- pivot K, S1 = sum (r - K), S2 = sum (r - K)^2 per lane and channel, r = relu(z + bias) --
// across tiles (parked in LDS during the K loops: the loop has no register to spare) and writes ONE partial per
// (block, group, utterance): 2 x (blocks per column + utterances) slots instead of one per 32-frame group (47 MB ->
// 11 MB at the bench batch), no cross-lane traffic and no store in the common case.  Layout of a slot: tdnn_common.h's three planes K | S1 | S2 of n_pad floats;
// slot of (block b of the column, utterance u, group g) = 2 (b + u) + g -- b and u both grow along the rows, so
// consecutive segments get distinct slots -- and the number of frames behind a partial goes to pool_cnt[slot]
// (pool_finalize_seg needs it to re-base the pivots).  Every wave writes a partial, possibly of zero frames, for EVERY
// utterance that overlaps its block's row range: pool_finalize_seg reads exactly those.
// The pivot of a segment is frame 0 of the 32-frame group in which the segment's first rows fall (a computed row
// of the same channel, this utterance's or its neighbour's: tdnn_common.h, pool_group_impl on why that is enough).
struct Seg {
    f32x4 k, s1, s2;            // per lane: its four channels
};

// flush: reduce S1, S2 over the four lane quads (reduce-scatter: four v_permlane32_swap + two v_permlane16_swap, six
// adds; afterwards quad 0 holds S1 of columns 0,1, quad 2 S1 of columns 2,3, quad 1 S2 of columns 0,1, quad 3 S2 of
// columns 2,3) and write the slot: ONE 8-byte store = 256 contiguous bytes of the S1 plane and 256 of the S2 plane;
// quad 0 writes the pivots (16 bytes per lane); one lane of the block's first column writes the frame count.
__device__ __forceinline__ void flush_seg(const TdnnArgs& a, Seg& sg, int slot, int n_rows, int q, int col0, bool cnt_writer) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int ld = a.ldy;
    // descriptor at the slot (64-bit base: no 2 GiB limit on the partials buffer; a flush is rare)
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(a.pool_part + (int64_t)slot * (kPoolPlanes * ld));
    const float y0 = swap32_add(sg.s1[0], sg.s1[2]), y1 = swap32_add(sg.s2[0], sg.s2[2]);
    const float y2 = swap32_add(sg.s1[1], sg.s1[3]), y3 = swap32_add(sg.s2[1], sg.s2[3]);
    const float z0 = swap16_add(y0, y1), z1 = swap16_add(y2, y3);
    const int soff = 0;
    const u32x2 v = {__float_as_uint(z0), __float_as_uint(z1)};
    __builtin_amdgcn_raw_buffer_store_b64(v, prs, (col0 + (q >> 1) * 2 + (1 + (q & 1)) * ld) * 4, soff, 0);
    if (q == 0) {
        const u32x4 kv = {__float_as_uint(sg.k[0]), __float_as_uint(sg.k[1]), __float_as_uint(sg.k[2]), __float_as_uint(sg.k[3])};
        __builtin_amdgcn_raw_buffer_store_b128(kv, prs, col0 * 4, soff, 0);
        asm volatile("s_nop 1" ::"v"(kv));       // the 128-bit-store data hazard (tdnn_common.h, store_acc)
    }
    if (cnt_writer) a.pool_cnt[slot] = n_rows;
    sg.k = f32x4{0.f, 0.f, 0.f, 0.f};
    sg.s1 = sg.k;
    sg.s2 = sg.k;
}

// cursor of a wave: utterance being accumulated, its end row, frames of it accumulated so far (all wave-uniform)
struct SegCur {
    int u;
    int n;
    int64_t end;
};

// One 32-frame group of this wave (acc row): v{f}{c} = accumulator of frame block f (frames 16 f + 4 q + e) and channel
// col0 + c, bias inside.  limit = first row that does not belong to this block (its range end, or the end of the batch).
// ADDB: the accumulators come WITHOUT the bias (bf16x3), bi = the bias of the lane's four channels.
template <bool RAGGED, bool ADDB>
__device__ __forceinline__ void pool_rows(const TdnnArgs& a, f32x4& v00, f32x4& v01, f32x4& v02, f32x4& v03, f32x4& v10,
                                          f32x4& v11, f32x4& v12, f32x4& v13, int64_t row_g, int64_t limit, int q, int col0,
                                          int blk, int grp, bool cnt_writer, SegCur& sc, Seg& sg, const f32x4& bi) {
    const RowMap& m = a.out_map;
    if (row_g >= limit) return;
    // r = relu(z + bias), IN PLACE: as an expression in both paths below hipcc computes the 32 values up front into
    // 32 more registers, next to 128 live accumulators
#define PQ_RELU(v_, c_) _Pragma("unroll") for (int e = 0; e < 4; ++e) v_[e] = relu1(ADDB ? v_[e] + bi[c_] : v_[e]);
    PQ_RELU(v00, 0) PQ_RELU(v01, 1) PQ_RELU(v02, 2) PQ_RELU(v03, 3) PQ_RELU(v10, 0) PQ_RELU(v11, 1) PQ_RELU(v12, 2) PQ_RELU(v13, 3)
#undef PQ_RELU
    const int64_t g_end = row_g + 32 < limit ? row_g + 32 : limit;
    // utterances that ended before this group (the current one, and any that lay wholly in the other group's rows)
    while (sc.end <= row_g && sc.u < m.n_utts - 1) {
        flush_seg(a, sg, 2 * (blk + sc.u) + grp, sc.n, q, col0, cnt_writer);
        sc.u = __builtin_amdgcn_readfirstlane(sc.u + 1);
        sc.n = 0;
        sc.end = first_row<RAGGED>(m, sc.u + 1);
    }
    int64_t lo = row_g;
    for (;;) {
        const int64_t hi = sc.end < g_end ? sc.end : g_end;           // rows [lo, hi) of the group belong to sc.u
        if (hi > lo) {
            if (sc.n == 0) {                                          // the segment's first rows: take the pivots
                sg.k = f32x4{quad0(v00[0]), quad0(v01[0]), quad0(v02[0]), quad0(v03[0])};
            }
            if (hi - lo == 32 || PP_KNOCK_MASKED) {                   // the whole group: no masks
                // plain v_sub / v_add / v_fma (v_pk_*_f32 issue at ~17 cycles each: MI355X_MICROARCH.md, price of fillers;
                // -fno-slp-vectorize keeps hipcc from re-packing them), the four channels' chains interleaved so that no
                // instruction waits for the one before it
                f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
#define PQ_STEP(f_, e_)                                                                                       \
                {                                                                                             \
                    const float d0 = v##f_##0[e_] - sg.k[0], d1 = v##f_##1[e_] - sg.k[1];                      \
                    const float d2 = v##f_##2[e_] - sg.k[2], d3 = v##f_##3[e_] - sg.k[3];                      \
                    t1[0] += d0; t1[1] += d1; t1[2] += d2; t1[3] += d3;                                       \
                    t2[0] = fmaf(d0, d0, t2[0]); t2[1] = fmaf(d1, d1, t2[1]);                                 \
                    t2[2] = fmaf(d2, d2, t2[2]); t2[3] = fmaf(d3, d3, t2[3]);                                 \
                }
                PQ_STEP(0, 0) PQ_STEP(0, 1) PQ_STEP(0, 2) PQ_STEP(0, 3) PQ_STEP(1, 0) PQ_STEP(1, 1) PQ_STEP(1, 2) PQ_STEP(1, 3)
#undef PQ_STEP
                sg.s1 += t1;
                sg.s2 += t2;
            } else {
                const int lo_l = (int)(lo - row_g), hi_l = (int)(hi - row_g);       // local rows [lo_l, hi_l), hi_l - lo_l < 32
                const unsigned below_hi = hi_l >= 32 ? 0xffffffffu : ((1u << hi_l) - 1u);
                const unsigned lm = (below_hi & ~((1u << lo_l) - 1u)) >> (4 * q);   // this lane's frames: bits 16 f + e
                // (the pivots go through an opaque copy here: with the same SSA value in both paths hipcc computes all 32
                //  differences v - k up front, shared by the two paths, into 32 more registers next to 128 live accumulators)
                f32x4 km = sg.k;
                asm volatile("" : "+v"(km));
#define PQ_MASKED(f_, c_)                                                                                     \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                               \
                    const bool in = (lm >> (16 * f_ + e)) & 1u;      /* a SELECT: rows outside may hold anything */ \
                    const float d = in ? v##f_##c_[e] - km[c_] : 0.f;                                          \
                    sg.s1[c_] += d;                                                                           \
                    sg.s2[c_] = fmaf(d, d, sg.s2[c_]);                                                        \
                }
                PQ_MASKED(0, 0) PQ_MASKED(1, 0) PQ_MASKED(0, 1) PQ_MASKED(1, 1)
                PQ_MASKED(0, 2) PQ_MASKED(1, 2) PQ_MASKED(0, 3) PQ_MASKED(1, 3)
#undef PQ_MASKED
            }
            sc.n += (int)(hi - lo);
        }
        if (sc.end >= g_end || sc.u >= m.n_utts - 1) break;          // the utterance goes on past this group
        flush_seg(a, sg, 2 * (blk + sc.u) + grp, sc.n, q, col0, cnt_writer);   // it ended inside the group
        sc.u = __builtin_amdgcn_readfirstlane(sc.u + 1);
        sc.n = 0;
        sc.end = first_row<RAGGED>(m, sc.u + 1);
        lo = hi;
    }
}

// end of the block: the current utterance, and every later one that still begins inside the block's rows (seen by the
// other group only), get their partial
template <bool RAGGED>
__device__ __forceinline__ void pool_finish(const TdnnArgs& a, int64_t limit, int q, int col0, int blk, int grp,
                                            bool cnt_writer, SegCur& sc, Seg& sg) {
    const RowMap& m = a.out_map;
    for (;;) {
        flush_seg(a, sg, 2 * (blk + sc.u) + grp, sc.n, q, col0, cnt_writer);
        sc.n = 0;
        if (sc.u >= m.n_utts - 1 || sc.end >= limit) break;
        sc.u = __builtin_amdgcn_readfirstlane(sc.u + 1);
        sc.end = first_row<RAGGED>(m, sc.u + 1);
    }
}


// ---- fused statistics pooling of PLAIN bf16 (POOL && !X3): the sums on the matrix pipe ---------------------------------
// The pooling epilogue above costs four vector instructions per value (ReLU, r - K, S1 +=, S2 fma) on both waves of every
// SIMD, with the matrix pipe idle, and ~25 KB of code per tile height (the kernel: 111 KB, beyond the instruction cache).
// bf16x3 keeps it (its partials must hold fp32-level sums).  Plain bf16 (bar 1e-2) does this instead:
//   * ONE pivot C per (wave, channel) for the whole block: the ReLU output of the first frame of the wave's first tile,
//     rounded to bf16.  From the block's second tile on it is inside the accumulators already -- their first MFMA takes
//     bias - C as srcC (PP_MF_S0) -- so a deviation is ONE instruction:  d = r - C = max(z + bias - C, -C).  (The first tile
//     subtracts C once it knows it.)  Every partial of the wave carries K = C: pool_finalize_seg re-bases in fp64.
//   * the deviations of a 32-frame group, rounded to bf16 and packed (8 per lane: frames 16 f + 4 q + e of the lane's channel
//     of channel block cb), are at once the B operand [32 frames x 16 channels] and the A operand [16 channels x 32 frames]
//     of a 16x16x32 MFMA:   ones x d = S1 in every row,   d^T x d = the Gram matrix, whose DIAGONAL is S2
//     (lane c + 16 (c >> 2), register c & 3).  Two MFMAs per channel block and group replace 96 vector instructions.
//   * why bf16 deviations are enough: d is small where it matters (|mean - C| is a few std: C is a sample of the channel),
//     each product is exact in the fp32 accumulator, and rounding d costs 2^-9 |d| of RANDOM error per frame (2e-4 of a
//     std on a mean over 286 frames); C itself is bf16-exact so that the frames where the channel is off (d = -C, all
//     alike) round with no error at all instead of a common one.
//   * rows outside the segment (another utterance, another block, past the batch) are cleared in the PACKED operand with a
//     bit mask (v_and: garbage, NaN included, becomes +0 in both operands).
// Between epilogues a wave parks C | S1 | S2 compactly (48 B per lane, as before); the MFMA accumulators restart at zero.
struct SegMx {
    f32x4 c;                    // pivot of the lane's four channels (bf16-exact)
    f32x4 a01, a23;             // S1 accumulators, two channel blocks each: blocks 0 / 2 in the even rows (registers 0, 2), blocks 1 / 3 in
                                // the odd ones -- their MFMAs take `ones` only in the even / odd ROWS of the A operand (lanes): half the registers
    f32x4 g0, g1, g2, g3;       // Gram accumulators
    // the lane's 48 bytes of LDS at park_base + 48 * lane: C | S1 | S2, the compact sums gathered before the last fold (needed at
    // a fold only: in registers they were 8 of the ~20 the epilogue spilled).  Everything lane-derived a fold or a flush needs
    // (its LDS slot, lane quad q, row r, first column) is recomputed THERE from an opaque lane id: computed once per epilogue
    // the compiler holds it in registers across the groups, and the epilogue has none to spare.
    char* park_base;            // wave-uniform: smem + kParkOff + 48 * 64 * wave
    int colw;                   // wave-uniform: first column of the wave, n0 + 64 * wc
    bool cnt_wave;              // wave-uniform: this wave's lane 0 writes the frame counts (first column block, first wave column)
};
__device__ __forceinline__ int mx_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ unsigned pk_min_u16(unsigned x, unsigned y) {
    unsigned r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// (volatile asm: as a plain conversion hipcc hoists all sixteen of a group above the first MFMA pair -- sixteen more live
//  registers in an epilogue that has two to spare)
__device__ __forceinline__ unsigned cvt_pk(float x, float y) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// conversion and minimum as ONE statement: behind an asm statement whose result the next instruction reads hipcc puts an
// s_nop 0 (it cannot know that the unknown instruction is not a transcendental one) -- sixteen per group, in an epilogue
// whose vector issue slots both waves of the SIMD are queueing for
__device__ __forceinline__ unsigned cvt_pk_min(float x, float y, unsigned m) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_min_u16 %0, %0, %3" : "=&v"(r) : "v"(x), "v"(y), "v"(m));
    return r;
}

// The pooling MFMAs are inline asm with the accumulator TIED, like the K loop's (PP_MF): with the builtin hipcc gave every
// MFMA a fresh destination and copied the 32 accumulator registers at every merge of the bookkeeping's control flow -- 160
// v_mov per 32-frame group next to 56 useful instructions (stamps: 1.4-2.0 k cycles per group, no better than the vector sums
// they replaced).  What hipcc would otherwise look after is looked after by hand, and so that it holds WHATEVER copies the
// register allocator still places around the asm statements (it does, where paths merge):
//   * vector write -> MFMA operand read: every MFMA pair is ONE asm statement that opens with two wait states (s_nop 1),
//     so a copy or a mask the compiler puts in front of the statement is two wait states old when the MFMA reads it;
//   * MFMA write -> vector read (4 passes: 7 wait states): PMX_SETTLE() closes every group, PP_MFMA_SETTLE() every fold;
//   * the same accumulator in two MFMAs of a group (a01 / a23): a dependent chain, interlocked by the hardware like the
//     K loop's.
#define PMX_MFS(pk_, ONES_, A_, G_)                                                                            \
    if constexpr (PP_KNOCK_MXMF) asm volatile("" : "+v"(A_), "+v"(G_) : "v"(ONES_), "v"(pk_)); else                 \
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %3, %1"     \
                 : "+v"(A_), "+v"(G_) : "v"(ONES_), "v"(pk_));
#define PMX_SETTLE() asm volatile("s_nop 7" ::: "memory")

// what the MFMA accumulators gathered goes to the compact sums; they restart at zero
__device__ __forceinline__ void fold_mx(SegMx& sg) {
    PP_MFMA_SETTLE();
    const int lane = mx_lane(), r = lane & 15;
    char* park = sg.park_base + lane * 48;
    // the diagonal element of the lane: register r & 3, taken with three selects on two bit tests (written as a chain of
    // `k == j ? g[j]` hipcc turns it into a vector element at a run-time index -- through scratch memory)
    const bool b0 = (r & 1) != 0, b1 = (r & 2) != 0;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#define PMX_DIAG(g_) (b1 ? (b0 ? g_[3] : g_[2]) : (b0 ? g_[1] : g_[0]))
    f32x4 s1 = *reinterpret_cast<const f32x4*>(park + 16), s2 = *reinterpret_cast<const f32x4*>(park + 32);
    s1[0] += sg.a01[0]; s1[1] += sg.a01[1]; s1[2] += sg.a23[0]; s1[3] += sg.a23[1];
    s2[0] += PMX_DIAG(sg.g0); s2[1] += PMX_DIAG(sg.g1); s2[2] += PMX_DIAG(sg.g2); s2[3] += PMX_DIAG(sg.g3);
    *reinterpret_cast<f32x4*>(park + 16) = s1;
    *reinterpret_cast<f32x4*>(park + 32) = s2;
#undef PMX_DIAG
    sg.a01 = z; sg.a23 = z;
    sg.g0 = z; sg.g1 = z; sg.g2 = z; sg.g3 = z;
    // (the zeros are vector writes of registers the next MFMAs read as srcC: they come a group's packing later at the earliest)
    asm volatile("" : "+v"(sg.a01), "+v"(sg.a23));
    asm volatile("" : "+v"(sg.g0), "+v"(sg.g1), "+v"(sg.g2), "+v"(sg.g3));
}

// write the segment's partial (slot layout: tdnn_common.h, three planes K | S1 | S2): the 16 lanes that hold the Gram
// diagonals (q == r >> 2) own four adjacent channels each -- three 16-byte stores, 256 contiguous bytes per plane
__device__ __forceinline__ void flush_mx(const TdnnArgs& a, SegMx& sg, int slot, int n_rows) {
    fold_mx(sg);
    const int lane = mx_lane(), q = lane >> 4, r = lane & 15;
    char* park = sg.park_base + lane * 48;
    const int col0 = sg.colw + 4 * r;
    const int ld = a.ldy;
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(a.pool_part + (int64_t)slot * (kPoolPlanes * ld));
    if (q == (r >> 2)) {
        const u32x4 kv = __builtin_bit_cast(u32x4, sg.c), v1 = *reinterpret_cast<const u32x4*>(park + 16),
                    v2 = *reinterpret_cast<const u32x4*>(park + 32);
        __builtin_amdgcn_raw_buffer_store_b128(kv, prs, col0 * 4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(v1, prs, (col0 + ld) * 4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(v2, prs, (col0 + 2 * ld) * 4, 0, 0);
        asm volatile("s_nop 1" ::"v"(kv), "v"(v1), "v"(v2));       // the 128-bit-store data hazard (tdnn_common.h, store_acc)
    }
    if (sg.cnt_wave && lane == 0) a.pool_cnt[slot] = n_rows;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(park + 16) = z;
    *reinterpret_cast<f32x4*>(park + 32) = z;
}

// first compact row of utterance u, fixed-length or ragged decided at RUN time by a scalar branch (the offsets through the
// scalar cache: sload_i64): one copy of the epilogue instead of two
__device__ __forceinline__ int64_t first_row_rt(const RowMap& m, int u) {
    u = __builtin_amdgcn_readfirstlane(u);
    if (m.offsets != nullptr) return sload_i64(m.offsets + u) - (int64_t)u * m.cum;
    return (int64_t)u * (m.fixed_T - m.cum);
}

// cursor of the matrix-pipe form: SegCur plus the end of the current utterance RELATIVE to the wave's first row of the tile,
// clamped to 32 bits -- the per-group bookkeeping is then 32-bit scalar arithmetic (a 64-bit compare is a vector instruction
// and a round trip through vcc on this ISA)
__device__ __forceinline__ int rel_row(int64_t row, int64_t base) {
    const int64_t d = row - base;
    return (int)(d < -(1 << 30) ? -(1 << 30) : d > (1 << 30) ? (1 << 30) : d);
}

// packed deviations of channel block cb_ of one 32-frame group: v0 / v1 = its accumulators of frame blocks 0 / 1 (z + bias - C).
// d = max(z + bias - C, -C) is taken AFTER the rounding, on the packed pair: -C is bf16-exact, so the two commute, and as
// unsigned 16-bit patterns the larger of a bf16 x and a bf16 m <= 0 is their MINIMUM (a positive x is below 0x8000 <= m; of two
// negative values the smaller pattern is the smaller magnitude) -- one v_pk_min_u16 per pair instead of two v_max_f32.  The same
// instruction masks: where a frame lies outside the segment the operand's half is 0x0000, and min(x, 0) = +0 whatever x holds.
#define PMX_PACKM(pk_, cb_, v0_, v1_, MASKED_)                                                                 \
    u32x4 pk_ = ones_e;                                                                                        \
    if constexpr (!PP_KNOCK_MXPK) {                                                                            \
        const u32x4 m_ = MASKED_ ? mk & ncp[cb_] : u32x4{ncp[cb_], ncp[cb_], ncp[cb_], ncp[cb_]};             \
        pk_ = u32x4{cvt_pk_min(v0_[0], v0_[1], m_[0]), cvt_pk_min(v0_[2], v0_[3], m_[1]),                      \
                    cvt_pk_min(v1_[0], v1_[1], m_[2]), cvt_pk_min(v1_[2], v1_[3], m_[3])};                     \
    }
// one group: the MFMAs of a channel block behind the packing of the next one.  Two copies of this code, for whole groups (no
// mask) and for the others; where the two paths merge the register allocator may copy accumulators -- PMX_MFS / PMX_SETTLE
// are written so that this is harmless
#define PMX_GROUP(MASKED_)                                                                                     \
    {                                                                                                          \
        PMX_PACKM(pk0, 0, v00, v10, MASKED_) SB();                                                             \
        PMX_PACKM(pk1, 1, v01, v11, MASKED_) SB();                                                             \
        PMX_MFS(pk0, ones_e, sg.a01, sg.g0) SB();                                                              \
        PMX_PACKM(pk2, 2, v02, v12, MASKED_) SB();                                                             \
        PMX_MFS(pk1, ones_o, sg.a01, sg.g1) SB();                                                              \
        PMX_PACKM(pk3, 3, v03, v13, MASKED_) SB();                                                             \
        PMX_MFS(pk2, ones_e, sg.a23, sg.g2) SB();                                                              \
        PMX_MFS(pk3, ones_o, sg.a23, sg.g3) SB();                                                              \
        PMX_SETTLE(); SB();                                                                                    \
    }

// One 32-frame group of this wave (acc row): v{f}{cb} = accumulator of frame block f and channel block cb, holding
// z + bias - C.  g0 = the group's first row relative to the wave's first row of the tile, lim_rel = first row (relative) that
// is not this block's, sc.end_rel likewise the end of the current utterance.
struct SegCurMx {
    int u, n;
    int64_t end;
    int end_rel;
};
__device__ __forceinline__ void pool_rows_mx(const TdnnArgs& a, const f32x4& v00, const f32x4& v01, const f32x4& v02,
                                             const f32x4& v03, const f32x4& v10, const f32x4& v11, const f32x4& v12,
                                             const f32x4& v13, int g0, int lim_rel, int64_t row_base, int blk, int grp, SegCurMx& sc,
                                             SegMx& sg, const u32x4& ones_e, const u32x4& ones_o) {
    // -C of the lane's four channels as packed bf16 pairs (C is bf16-exact: the upper half of its fp32 pattern, sign flipped)
    const u32x4 cb_ = __builtin_bit_cast(u32x4, sg.c);
    const u32x4 ncp = {((cb_[0] ^ 0x80000000u) >> 16) * 0x10001u, ((cb_[1] ^ 0x80000000u) >> 16) * 0x10001u,
                       ((cb_[2] ^ 0x80000000u) >> 16) * 0x10001u, ((cb_[3] ^ 0x80000000u) >> 16) * 0x10001u};
    const RowMap& m = a.out_map;
    // the common case first, with two compares of bookkeeping: the whole group belongs to this block and to the current utterance
    // (stamps and knock-outs: at ~60 scalar instructions per group the generic walk below cost as much as the arithmetic)
    if (sc.end_rel >= g0 + 32 && lim_rel >= g0 + 32) {
        const u32x4 mk = ones_e;                                      // (unused)
        PMX_GROUP(false)
        sc.n += 32;
        return;
    }
    if (g0 >= lim_rel) return;
    const int n_last = m.n_utts - 1;
    const int g_end = g0 + 32 < lim_rel ? g0 + 32 : lim_rel;
#define PMX_NEXT_UTT()                                                                                         \
    {                                                                                                          \
        flush_mx(a, sg, 2 * (blk + sc.u) + grp, sc.n);                                                         \
        sc.u = __builtin_amdgcn_readfirstlane(sc.u + 1);                                                       \
        sc.n = 0;                                                                                              \
        sc.end = first_row_rt(m, sc.u + 1);                                                                    \
        sc.end_rel = rel_row(sc.end, row_base);                                                                \
    }
    // utterances that ended before this group (the current one, and any that lay wholly in the other group's rows)
    while (sc.end_rel <= g0 && sc.u < n_last) PMX_NEXT_UTT()
    int lo = g0;
    for (;;) {
        const int hi = sc.end_rel < g_end ? sc.end_rel : g_end;       // rows [lo, hi) of the group belong to sc.u
        if (hi > lo) {
            {
                const int lo_l = lo - g0, hi_l = hi - g0;             // local rows [lo_l, hi_l) of the group
                const unsigned below_hi = hi_l >= 32 ? 0xffffffffu : ((1u << hi_l) - 1u);
                const unsigned lm = (below_hi & ~((1u << lo_l) - 1u)) >> (4 * (mx_lane() >> 4));   // this lane's frames (lane quad q): bit 16 f + e
                // packed register j holds frames (f, e) = (j >> 1, 2 (j & 1)) in its low half and (j >> 1, 2 (j & 1) + 1) in its high half
#define PMX_MK(b_) ((unsigned)(-(int)((lm >> (b_)) & 1u)) & 0xffffu) | ((unsigned)(-(int)((lm >> ((b_) + 1)) & 1u)) << 16)
                const u32x4 mk = {PMX_MK(0), PMX_MK(2), PMX_MK(16), PMX_MK(18)};
#undef PMX_MK
                PMX_GROUP(true)
            }
            sc.n += hi - lo;
        }
        if (sc.end_rel >= g_end || sc.u >= n_last) break;            // the utterance goes on past this group
        PMX_NEXT_UTT()                                                // it ended inside the group
        lo = hi;
    }
#undef PMX_NEXT_UTT
}

// end of the block (pool_finish for this form)
__device__ __forceinline__ void pool_finish_mx(const TdnnArgs& a, int64_t limit, int blk, int grp, SegCurMx& sc, SegMx& sg) {
    const RowMap& m = a.out_map;
    for (;;) {
        flush_mx(a, sg, 2 * (blk + sc.u) + grp, sc.n);
        sc.n = 0;
        if (sc.u >= m.n_utts - 1 || sc.end >= limit) break;
        sc.u = __builtin_amdgcn_readfirstlane(sc.u + 1);
        sc.end = first_row_rt(m, sc.u + 1);
    }
}

// One tile: K loop, request of the next tile's first K-tiles, epilogue.
template <int MR, bool POOL, bool X3>
__device__ __forceinline__ void process_tile(const TdnnArgs& a, char* smem, Stream& st, const Lane& ln,
                                             const Tile& t, const Tile& nxt, bool has_next, bool first, int n0, int nk,
                                             SegCur& sc, int64_t limit, int blk) {
    // source rows of the NEXT tile (its first K-tiles are requested during this tile's last two): worked out
    // here, before the accumulators exist, and parked in four registers
    Rows rows_next = st.cur;
    if (has_next) set_rows(a, nxt, ln.grp, ln.wc, st, rows_next);
    // the accumulators start at the bias of their lane's four channels (64*wc + 4r .. +3 of the block's column; the
    // constants live in a 3-KiB LDS table: registers held across the K loop were what a third tile height cost)
    const float* cst = reinterpret_cast<const float*>(smem + kConstOff) + ln.wc * 64 + 4 * ln.r;
#define PP_ACCS(i_) acc##i_##00, acc##i_##01, acc##i_##02, acc##i_##03, acc##i_##10, acc##i_##11, acc##i_##12, acc##i_##13
    f32x4 PP_ACCS(0), PP_ACCS(1), PP_ACCS(2), PP_ACCS(3);
    // the bias of the lane's four channels, one four-register tuple per channel block: srcC of every accumulator's first MFMA
    f32x4 bias4_0, bias4_1, bias4_2, bias4_3;
    {
        float4 bi = *reinterpret_cast<const float4*>(cst);
        if constexpr (POOL && !X3) {    // matrix-pipe pooling: the block's pivot C rides in the accumulators (0 in the first tile)
            int lane_p;       // opaque lane id: from ln.* the address is hoisted out of the tile loop and held (spilled) across it
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
            const f32x4 cp = *reinterpret_cast<const f32x4*>(smem + kParkOff + (ln.wave * 64 + lane_p) * 48);
            bi.x -= cp[0]; bi.y -= cp[1]; bi.z -= cp[2]; bi.w -= cp[3];
        }
        bias4_0 = f32x4{bi.x, bi.x, bi.x, bi.x};
        bias4_1 = f32x4{bi.y, bi.y, bi.y, bi.y};
        bias4_2 = f32x4{bi.z, bi.z, bi.z, bi.z};
        bias4_3 = f32x4{bi.w, bi.w, bi.w, bi.w};
    }
    // materialised HERE: left alone hipcc sinks the 16 moves to just in front of the first MFMA, which is inline asm -- so
    // nobody would pad the VALU-write -> MFMA-srcC-read hazard (seen: results that differ from run to run)
    asm volatile("" : "+v"(bias4_0), "+v"(bias4_1), "+v"(bias4_2), "+v"(bias4_3));
    float4 wf0_0, wf0_1, wf1_0, wf1_1, wf2_0, wf2_1, wf3_0, wf3_1;     // [channel block]_[k-step]
    float4 af0_00, af0_01, af0_10, af0_11, af1_00, af1_01, af1_10, af1_11;   // acc rows 0,1: [frame block][k-step] (read during the previous mfma 1)
    float4 af2_00, af2_01, af2_10, af2_11, af3_00, af3_01, af3_10, af3_11;   // acc rows 2,3 (read during mfma 0)
#ifdef XVEC_DIAG
    unsigned long long dsum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dprev, dstart;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dstart)::"memory");
#endif

#ifdef XVEC_KNOCK
    if (PP_KNOCK_RD || PP_KNOCK_RDW || PP_KNOCK_RDA) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        wf0_0 = wf0_1 = wf1_0 = wf1_1 = wf2_0 = wf2_1 = wf3_0 = wf3_1 = z;
        af0_00 = af0_01 = af0_10 = af0_11 = af1_00 = af1_01 = af1_10 = af1_11 = z;
        af2_00 = af2_01 = af2_10 = af2_11 = af3_00 = af3_01 = af3_10 = af3_11 = z;
    }
#endif
    // A block's first tile waits for its first K-tile (requested by the kernel prologue; its pieces are older
    // than the MR+4 of K-tile 1); later tiles find it in LDS, confirmed by the previous tile's last K-tiles.
    if (first) {
        if (MR == 4) { PP_WAIT_VM(8); } else if (MR == 3) { PP_WAIT_VM(7); } else { PP_WAIT_VM(6); }
        PP_BARRIER()
    }
    // acc rows 0,1 of K-tile 0: the only activation fragments read outside an MFMA segment (the previous tile's
    // last MFMA segment fetched them too, but keeping them in registers across the epilogue costs it 32 VGPRs)
    PP_RA(0, 0, 0, 0) PP_RA(0, 0, 1, 0) PP_RA(0, 1, 0, 0) PP_RA(0, 1, 1, 0)
    PP_RA(1, 0, 0, 0) PP_RA(1, 0, 1, 0) PP_RA(1, 1, 0, 0) PP_RA(1, 1, 1, 0)
    PP_WAIT_LGKM();
    // Their slots are the first ones the loop refills (load 0 of K-tile 0 requests K-tile 2 into them), and the
    // waves of a group leave the epilogue at different times: every wave must have read them before any wave
    // may request.  (Deferring that one request instead costs a branch in the loop, and with it hipcc's
    // register assignment: 160 spilled registers.)
    PP_BARRIER()
    if (ln.grp == 1) PP_BARRIER()          // ping-pong: the second group runs one barrier behind
#ifdef XVEC_DIAG
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dprev)::"memory");
    dsum[12] += dprev - dstart;            // head wait
#endif
    KPos k2 = {0, 0, 0};                    // K-tile requested now (two ahead of the one computed), its W index,
    kstep<X3>(a, k2);                       // the height of its tile, and whether there is anything to request
    kstep<X3>(a, k2);
    int wq = 2;
    int mr_req = MR;
    bool req = true;
    // (the first K-tile pair stands outside the loop: its k-step 0 MFMAs start the accumulators, see PP_MF_S0)
#define PP_LAST_SWITCH()                                                                           \
        if (last) {                         /* from here on the requests are the next tile's K-tiles 0 and 1 */ \
            req = has_next;                                                                        \
            if (has_next) {                                                                        \
                st.cur = rows_next;                                                                \
                mr_req = nxt.mr;                                                                   \
                k2.tap = 0;                                                                        \
                k2.so = 0;                                                                         \
                k2.ph = 0;                                                                         \
                wq = 0;                                                                            \
            }                                                                                      \
        }
    {
        const bool last = 2 >= nk;
        PP_LAST_SWITCH()
        PP_KTILE(0, false, true)
        PP_KTILE(1, true, false)
    }
    for (int q = 2; q < nk; q += 2) {
        const bool last = q + 2 >= nk;
        PP_LAST_SWITCH()
        PP_KTILE(0, false, false)
        PP_KTILE(1, true, false)
    }
#undef PP_LAST_SWITCH
    if (ln.grp == 0) PP_BARRIER()
    PP_MFMA_SETTLE();
    PP_STAMP(13)                            // tail barrier
    PP_ESTAMP(13)                           // (DIAG=2: everything up to here)
    PP_STAMP(5)

    const int64_t row0 = t.m0 + ln.grp * 32 * MR;
#define PP_ACCV(i_) "v"(acc##i_##00), "v"(acc##i_##01), "v"(acc##i_##02), "v"(acc##i_##03), "v"(acc##i_##10), "v"(acc##i_##11), "v"(acc##i_##12), "v"(acc##i_##13)
    if constexpr (PP_KNOCK_EPI) {
        asm volatile("" ::PP_ACCV(0), PP_ACCV(1));
        asm volatile("" ::PP_ACCV(2), PP_ACCV(3));
    } else if (PP_KNOCK_EPI4 && !POOL && a.n_taps == 1) {
        asm volatile("" ::PP_ACCV(0), PP_ACCV(1));
        asm volatile("" ::PP_ACCV(2), PP_ACCV(3));
    } else if constexpr (!POOL) {
        // ReLU + folded BatchNorm (tdnn_layer.py:30-39).  The lane's four accumulators of a frame hold the ADJACENT
        // channels 4r..4r+3 of the wave's 64-channel block (pack.hip, shape 16): two v_cvt_pk_bf16_f32 make the 8
        // bytes that belong at column 4r, and one store instruction writes whole 128-byte row segments of four
        // frames (lane quad q = frames 4q..4q+3 of a 16-frame block: register e of quad q is frame 4q + e).
        // (64-bit stores: the data hazard behind 128-bit buffer stores, tdnn_common.h store_acc, does not apply.)
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const __amdgpu_buffer_rsrc_t yrsrc = make_rsrc(static_cast<char*>(a.Y) + (t.m0 * (int64_t)a.ldy + n0) * 2);
        int lane_e;           // opaque lane id: see the pooling epilogue
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int y_voff = (4 * (lane_e >> 4) * a.ldy + ln.wc * 64 + 4 * (lane_e & 15)) * 2;
        const float* cst_e = reinterpret_cast<const float*>(smem + kConstOff) + ln.wc * 64 + 4 * (lane_e & 15);
        // bf16x3 only: scale / shift of the folded BatchNorm and the bias (not in its accumulators, PP_MF_S0).  Plain bf16 stores
        // relu(z + bias') and nothing else: its BatchNorm is deferred into the consumer's weights (xvec_api.hip, refold), the
        // bias is in the accumulators, and the ReLU is taken on the PACKED pair (relu_pk_bf16: one instruction per two values):
        // 1.25 vector instructions per value instead of 2.75, in an epilogue that runs in the open on both waves of a SIMD.
        float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc, bi_e = sc;
        if constexpr (X3) {
            sc = *reinterpret_cast<const float4*>(cst_e + 256);
            sh = *reinterpret_cast<const float4*>(cst_e + 512);
            bi_e = *reinterpret_cast<const float4*>(cst_e);
        }
        // Row (32 i + 16 f + e) of the wave's part of the tile: the 32 i go into the scalar offset, the 16 f + e into EIGHT lane
        // offsets made here from the opaque lane id.  (All 32 row offsets as scalars: loop-invariant, so hipcc computed them
        // before the tile loop, ran out of SGPRs, parked them in VGPR lanes and fetched each with v_readlane_b32 + s_nop 4
        // in front of its store.)
        int y_vo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y_vo[j] = y_voff + (16 * (j >> 2) + (j & 3)) * a.ldy * 2;
#define PP_STORE_F(i_, f_)                                                                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                               \
                const int so_ = (ln.grp * 32 * MR + 32 * i_) * a.ldy * 2;                                 \
                const int y_voff = y_vo[4 * f_ + e];                                                      \
                if constexpr (!X3) {                                                                      \
                    const u32x2 pk = {relu_pk_bf16(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc##i_##f_##0[e], acc##i_##f_##1[e]}, bf16x2))), \
                                      relu_pk_bf16(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc##i_##f_##2[e], acc##i_##f_##3[e]}, bf16x2)))}; \
                    __builtin_amdgcn_raw_buffer_store_b64(pk, yrsrc, y_voff, so_, 0);                     \
                } else {   /* ReLU + folded BatchNorm, then hi and the remainder v - hi (lo plane, y_plane_bytes further on) */ \
                    const float v0 = fmaf(relu1(acc##i_##f_##0[e] + bi_e.x), sc.x, sh.x);                 \
                    const float v1 = fmaf(relu1(acc##i_##f_##1[e] + bi_e.y), sc.y, sh.y);                 \
                    const float v2 = fmaf(relu1(acc##i_##f_##2[e] + bi_e.z), sc.z, sh.z);                 \
                    const float v3 = fmaf(relu1(acc##i_##f_##3[e] + bi_e.w), sc.w, sh.w);                 \
                    const u32x2 pk = {__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v0, v1}, bf16x2)), \
                                      __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v2, v3}, bf16x2))}; \
                    __builtin_amdgcn_raw_buffer_store_b64(pk, yrsrc, y_voff, so_, 0);                     \
                    const float l0 = v0 - __uint_as_float(pk[0] << 16), l1 = v1 - __uint_as_float(pk[0] & 0xffff0000u); \
                    const float l2 = v2 - __uint_as_float(pk[1] << 16), l3 = v3 - __uint_as_float(pk[1] & 0xffff0000u); \
                    const u32x2 pl = {__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{l0, l1}, bf16x2)), \
                                      __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{l2, l3}, bf16x2))}; \
                    __builtin_amdgcn_raw_buffer_store_b64(pl, yrsrc, y_voff, so_ + a.y_plane_bytes, 0);  \
                }                                                                                         \
            }
#define PP_STORE(i_)                                                                                   \
        if (MR > i_ && row0 + 32 * i_ < t.valid_end) { PP_STORE_F(i_, 0) PP_STORE_F(i_, 1) }
        PP_STORE(0) PP_STORE(1) PP_STORE(2) PP_STORE(3)
#undef PP_STORE
#undef PP_STORE_F
    } else if constexpr (!X3) {
        // plain bf16: the pooling sums on the matrix pipe (SegMx above)
        SegMx sg;
        sg.park_base = smem + kParkOff + ln.wave * (64 * 48);
        sg.colw = n0 + ln.wc * 64;
        sg.cnt_wave = n0 == 0 && ln.wc == 0;
        sg.c = *reinterpret_cast<const f32x4*>(sg.park_base + mx_lane() * 48);
        sg.a01 = f32x4{0.f, 0.f, 0.f, 0.f};
        sg.a23 = sg.a01;
        sg.g0 = sg.a01; sg.g1 = sg.a01; sg.g2 = sg.a01; sg.g3 = sg.a01;
        asm volatile("" : "+v"(sg.a01), "+v"(sg.a23));                                 // written here, not sunk to the first MFMA
        asm volatile("" : "+v"(sg.g0), "+v"(sg.g1), "+v"(sg.g2), "+v"(sg.g3));
        if (first && row0 < limit) {
            // the block's first tile: its accumulators hold z + bias (C was 0).  C = relu of the group's first frame (quad 0,
            // register 0 of frame block 0), rounded to bf16 so that -C is exact in the packed operand; subtracted here, ONCE
#define PP_PIV(c_) __uint_as_float(__float_as_uint(quad0(relu1(acc00##c_[0]))) + 0x8000u & 0xffff0000u)
            sg.c = f32x4{PP_PIV(0), PP_PIV(1), PP_PIV(2), PP_PIV(3)};
#undef PP_PIV
#define PP_SUBC(i_) if (MR > i_) { acc##i_##00 -= sg.c[0]; acc##i_##10 -= sg.c[0]; acc##i_##01 -= sg.c[1]; acc##i_##11 -= sg.c[1]; \
                                   acc##i_##02 -= sg.c[2]; acc##i_##12 -= sg.c[2]; acc##i_##03 -= sg.c[3]; acc##i_##13 -= sg.c[3]; }
            PP_SUBC(0) PP_SUBC(1) PP_SUBC(2) PP_SUBC(3)
#undef PP_SUBC
        }
        // all wave-uniform row bookkeeping below is 32-bit, relative to the wave's first row of this tile
        const int lim_rel = rel_row(limit, row0);
        SegCurMx scm = {sc.u, sc.n, sc.end, rel_row(sc.end, row0)};
        // A operands of the S1 MFMAs: bf16 ones in the even / the odd rows (row = lane & 15), zeros in the others
        const unsigned one_e = (mx_lane() & 1) ? 0u : 0x3f803f80u, one_o = one_e ^ 0x3f803f80u;
        u32x4 ones_e = {one_e, one_e, one_e, one_e}, ones_o = {one_o, one_o, one_o, one_o};
        asm volatile("" : "+v"(ones_e), "+v"(ones_o));        // materialised here, well before the first MFMA reads them
#define PP_POOLMX(i_)                                                                                  \
        if (MR > i_ && !PP_KNOCK_MXROWS)                                                                  \
            pool_rows_mx(a, acc##i_##00, acc##i_##01, acc##i_##02, acc##i_##03, acc##i_##10, acc##i_##11, acc##i_##12, \
                         acc##i_##13, 32 * i_, lim_rel, row0, blk, ln.grp, scm, sg, ones_e, ones_o);
        PP_ESTAMP(0)
        PP_POOLMX(0) PP_ESTAMP(1) PP_POOLMX(1) PP_ESTAMP(2) PP_POOLMX(2) PP_ESTAMP(3) PP_POOLMX(3) PP_ESTAMP(4)
#undef PP_POOLMX
        if (!has_next) pool_finish_mx(a, limit, blk, ln.grp, scm, sg);
        sc.u = scm.u;
        sc.n = scm.n;
        sc.end = scm.end;
        if constexpr (!PP_KNOCK_MXFOLD) {
            fold_mx(sg);
            *reinterpret_cast<f32x4*>(sg.park_base + mx_lane() * 48) = sg.c;
        } else {
            asm volatile("" ::"v"(sg.a01), "v"(sg.a23), "v"(sg.g0), "v"(sg.g1), "v"(sg.g2), "v"(sg.g3), "v"(sg.c));
        }
        PP_ESTAMP(5)
    } else {
        // (lane-derived values of this epilogue come from an opaque lane id: computed from ln.* the compiler hoists them
        //  out of the tile loop and carries them through the K loop, which has no register to spare)
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int q_e = lane_e >> 4, r_e = lane_e & 15;
        const int col0 = n0 + ln.wc * 64 + 4 * r_e;
        const bool cnt_writer = n0 == 0 && ln.wc == 0 && lane_e == 0;
        // the running sums of the wave's current utterance come back from their LDS slots
        char* park = smem + kParkOff + (ln.wave * 64 + lane_e) * 48;
        Seg sg;
        if constexpr (PP_KNOCK_PARK) {
            sg.k = f32x4{0.f, 0.f, 0.f, 0.f};
            sg.s1 = sg.k;
            sg.s2 = sg.k;
        } else {
            sg.k = *reinterpret_cast<const f32x4*>(park);
            sg.s1 = *reinterpret_cast<const f32x4*>(park + 16);
            sg.s2 = *reinterpret_cast<const f32x4*>(park + 32);
        }
        f32x4 bi_p = {0.f, 0.f, 0.f, 0.f};                       // bf16x3: the bias of the lane's four channels (not in the accumulators)
        if constexpr (X3) bi_p = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem + kConstOff) + ln.wc * 64 + 4 * r_e);
        PP_ESTAMP(0)
#define PP_POOL(RG_, i_)                                                                               \
        if (MR > i_ && !(PP_KNOCK_ROWS && i_ > 0))                                                        \
            pool_rows<RG_, X3>(a, acc##i_##00, acc##i_##01, acc##i_##02, acc##i_##03, acc##i_##10, acc##i_##11, acc##i_##12, \
                               acc##i_##13, row0 + 32 * i_, limit, q_e, col0, blk, ln.grp, cnt_writer, sc, sg, bi_p);
        if (a.out_map.offsets == nullptr) {
            PP_POOL(false, 0) PP_ESTAMP(1) PP_POOL(false, 1) PP_ESTAMP(2) PP_POOL(false, 2) PP_ESTAMP(3) PP_POOL(false, 3) PP_ESTAMP(4)
            if (!has_next) pool_finish<false>(a, limit, q_e, col0, blk, ln.grp, cnt_writer, sc, sg);
        } else {
            PP_POOL(true, 0) PP_POOL(true, 1) PP_POOL(true, 2) PP_POOL(true, 3)
            if (!has_next) pool_finish<true>(a, limit, q_e, col0, blk, ln.grp, cnt_writer, sc, sg);
        }
#undef PP_POOL
        if constexpr (!PP_KNOCK_PARK) {
            *reinterpret_cast<f32x4*>(park) = sg.k;
            *reinterpret_cast<f32x4*>(park + 16) = sg.s1;
            *reinterpret_cast<f32x4*>(park + 32) = sg.s2;
        } else {
            asm volatile("" ::"v"(sg.k), "v"(sg.s1), "v"(sg.s2));
        }
        PP_ESTAMP(5)
    }
#undef PP_ACCV
#undef PP_ACCS
#ifdef XVEC_DIAG
    {
        unsigned long long dend;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dend)::"memory");
        dsum[11] += dend - dprev;          // epilogue + second K-tile of the next tile
        dsum[14] += 1;
        dsum[15] += (unsigned long long)MR;
        if ((ln.wave & 3) == 0 && ln.r == 0 && ln.q == 0 && blockIdx.x < 512) {
            _Pragma("unroll") for (int k = 0; k < 16; ++k) g_pp16_diag[(POOL ? 512 * 32 : 0) + blockIdx.x * 32 + ln.grp * 16 + k] += dsum[k];
        }
    }
#endif
}

template <bool POOL, bool X3>
__global__ __launch_bounds__(kThreads, 2) void tdnn_pp_kernel(const TdnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int jcol = lid % a.n_tiles;                   // 256-channel column
    const int prange = lid / a.n_tiles;
    const int64_t u_begin = a.groups_total * (int64_t)prange / a.blocks_per_col;     // 64-frame units
    const int64_t u_end = a.groups_total * (int64_t)(prange + 1) / a.blocks_per_col;
    const int n0 = jcol * 256;
    const int nk = a.n_taps * a.cpt * (X3 ? 3 : 1);     // K-tiles of 64 (even); bf16x3: three per slab (kstep)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef XVEC_DIAG
    if (blockIdx.x == 0 && tid == 0) {
        g_pp16_clk[(POOL ? 4 : 0) + 0] = __builtin_amdgcn_s_memtime();
        g_pp16_clk[(POOL ? 4 : 0) + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    Lane ln;
    ln.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ln.grp = ln.wave >> 2;
    ln.wc = ln.wave & 3;
    ln.q = lane >> 4;
    ln.r = lane & 15;
    ln.rd = ln.r * kRowB;
    {
        // fragment of k-step s: k = 32 s + 8 q ..+7 = 16-byte chunk 4 s + q of the row's 128-byte slab, at the
        // swizzled position chunk ^ ((row >> 1) & 7); row = 16 fb + r, so the swizzle does not depend on fb
        const int sw = (ln.r >> 1) & 7;
        ln.k0 = ((0 + ln.q) ^ sw) << 4;
        ln.k1 = ((4 + ln.q) ^ sw) << 4;
    }
    ln.a_rd = ln.grp * 4 * kAccRowB + ln.rd;
    ln.w_rd = kWOff + ln.wc * 2 * kAccRowB + ln.rd;
    ln.a_k0 = ln.a_rd + ln.k0;
    ln.a_k1 = ln.a_rd + ln.k1;
    ln.w_k0 = ln.w_rd + ln.k0;
    ln.w_k1 = ln.w_rd + ln.k1;
    asm volatile("" : "+v"(ln.a_k0), "+v"(ln.a_k1), "+v"(ln.w_k0), "+v"(ln.w_k1));

    // per-channel constants of the block's column -> LDS
    if (tid < 192) {
        const int arr = tid >> 6, c4 = (tid & 63) * 4;
        const float* src = arr == 0 ? a.bias : arr == 1 ? a.scale : a.shift;
        *reinterpret_cast<float4*>(smem + kConstOff + arr * 1024 + c4 * 4) = *reinterpret_cast<const float4*>(src + n0 + c4);
    }

    // DMA map of this wave: piece row = lane >> 3 (8 rows per piece), LDS position lane & 7 holds the
    // source chunk (lane & 7) ^ swizzle(row), swizzle = (row >> 1) & 7 of the row's index in its 32-row block
    Stream st;
    const int prow = lane >> 3, ppos = lane & 7;
    {
        // (A pieces: row ln.wc * 8 + prow of the 32-frame acc row, chunk ppos ^ swizzle: set_rows)
        st.lds_a = (unsigned)(unsigned long long)(lds_ptr)(smem) + ln.grp * 4 * kAccRowB + ln.wc * 1024;
        const int wr = ln.wave * 8 + prow;                                // W: channel row of piece t = wr + 64*t
        const int w_chunk = (ppos ^ ((wr >> 1) & 7)) * 16;                // ((wr + 64t) >> 1) & 7 is the same for every t
        st.lds_w = (unsigned)(unsigned long long)(lds_ptr)(smem) + kWOff + ln.wave * 1024;
        st.wv0 = wr * kRowB + w_chunk;                                    // K-tile major: rows 128 B apart
        st.w64 = 64 * kRowB;
        st.wrsrc = make_srd(static_cast<const char*>(a.W) + (int64_t)jcol * nk * kWBytes);
        st.cur.av0 = st.cur.av1 = st.cur.av2 = st.cur.av3 = 0;
        st.u_tile = __builtin_amdgcn_readfirstlane(utt_of_row(a.out_map, u_begin * 64));
        st.off_next = row_off(a.out_map, st.u_tile + 1);
        st.cur.xrsrc = st.wrsrc;
        SegCur sc;
        sc.u = 0;
        sc.n = 0;
        sc.end = 0;

        // tiles of this block: n units cut into ceil(n/4) tiles of 4, 3 or 2 units, as equal as possible (5 = 3 + 2:
        // without the 2-unit tile a batch of 128 utterances ran slower than one of 96); n = 1: one 2-unit tile whose
        // second unit lies past the range and is masked
        const int n = (int)(u_end - u_begin);
        if (n <= 0) return;
        int nt = (n + 3) / 4;
        int base = n / nt, extra = n % nt;
        constexpr int kMinMr = 2;
        if (base < kMinMr) { base = kMinMr; extra = 0; nt = (n + kMinMr - 1) / kMinMr; }
        const int64_t range_end = u_end * 64;

        auto tile_at = [&](int idx, int64_t m0) {
            Tile t;
            t.m0 = m0;
            t.mr = idx < extra ? base + 1 : base;
            t.valid_end = range_end;
            return t;
        };
        Tile cur = tile_at(0, u_begin * 64);
        int64_t limit = range_end;                       // first row that is not this block's: its range end or the batch end
        if (POOL) {
            // both groups start at the utterance of the block's FIRST row (a wave flushes an empty partial for every
            // utterance its own rows skip)
            const PoolCur pc0 = pool_cursor(a, u_begin * 64);
            sc.u = pc0.u;
            sc.end = pc0.end;
            const int64_t total_rows = row_off(a.out_map, a.out_map.n_utts);
            limit = range_end < total_rows ? range_end : total_rows;
            f32x4* park = reinterpret_cast<f32x4*>(smem + kParkOff + tid * 48);
            park[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            park[1] = park[0];
            park[2] = park[0];
        }
        set_rows(a, cur, ln.grp, ln.wc, st, st.cur);
        __syncthreads();                                   // constants visible; nobody reads LDS buffers yet
        issue_head1<POOL>(a, st, cur.mr);
        issue_head2<POOL, X3>(a, st, cur.mr);
        for (int idx = 0; idx < nt; ++idx) {
            const bool has_next = idx + 1 < nt;
            Tile nxt = cur;
            if (has_next) nxt = tile_at(idx + 1, cur.m0 + 64 * cur.mr);
            if (cur.mr == 4)
                process_tile<4, POOL, X3>(a, smem, st, ln, cur, nxt, has_next, idx == 0, n0, nk, sc, limit, prange);
            else if (cur.mr == 3)
                process_tile<3, POOL, X3>(a, smem, st, ln, cur, nxt, has_next, idx == 0, n0, nk, sc, limit, prange);
            else
                process_tile<2, POOL, X3>(a, smem, st, ln, cur, nxt, has_next, idx == 0, n0, nk, sc, limit, prange);
            cur = nxt;
        }
#ifdef XVEC_DIAG
        if (blockIdx.x == 0 && tid == 0) {
            g_pp16_clk[(POOL ? 4 : 0) + 2] = __builtin_amdgcn_s_memtime();
            g_pp16_clk[(POOL ? 4 : 0) + 3] = __builtin_amdgcn_s_memrealtime();
        }
#endif
    }
}

}  // namespace pp16

#ifdef XVEC_DIAG
extern "C" int xvec_pp16_clk_read(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pp16::g_pp16_clk), 8 * 8);
}
extern "C" int xvec_pp16_diag_read(unsigned long long* host, int n_words, int reset) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(pp16::g_pp16_diag), (size_t)n_words * 8);
    if (e == hipSuccess && reset) e = hipMemset(nullptr, 0, 0);
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(pp16::g_pp16_diag)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 2 * 512 * 32);
    }
    return (int)e;
}
#endif

hipError_t launch_tdnn_pp16(const TdnnArgs& a, bool pool, hipStream_t s) {
    if (a.groups_total <= 0 || a.blocks_per_col <= 0 || a.blocks_per_col > a.groups_total || (a.cpt & 1) ||
        a.n_tiles <= 0)
        return hipErrorInvalidValue;
    const int grid = a.blocks_per_col * a.n_tiles;
#define PP_LAUNCH(POOL_, X3_)                                                                                       \
    {                                                                                                               \
        static LdsOptIn opt;                                                                                        \
        if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(pp16::tdnn_pp_kernel<POOL_, X3_>), pp16::kLdsBytes); \
            e != hipSuccess)                                                                                        \
            return e;                                                                                               \
        pp16::tdnn_pp_kernel<POOL_, X3_><<<dim3(grid), dim3(pp16::kThreads), pp16::kLdsBytes, s>>>(a);              \
    }
    const bool x3 = a.terms == 2;          // bf16x3: W = the three-K-tiles-per-slab packing, X / Y = hi and lo planes
    if (pool) {
        if (x3) PP_LAUNCH(true, true) else PP_LAUNCH(true, false)
    } else {
        if (x3) PP_LAUNCH(false, true) else PP_LAUNCH(false, false)
    }
#undef PP_LAUNCH
    return hipGetLastError();
}

}  // namespace xvec

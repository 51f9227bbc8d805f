// Frame-level TDNN layer in EXACT fp32 for large batches (BASELINE configs[1], the headline): the implicit GEMM of
// tdnn_layer.hip (tdnn_layer.py:26-41 of the reference: context gather -> Linear -> ReLU -> eval BatchNorm, optional
// fused statistics pooling, main.py:59-63) on the machine mapping that was built for the bf16 matrix rate
// (tdnn_pp16.hip; this file derives from its 32x32 form, profiles/experiments/tdnn_pp_32x32x16.hip) with
// v_mfma_f32_32x32x2_f32 in place of the bf16 MFMA: exact fp32 products and sums, bit for bit an fmaf chain.
//
// Why.  The 128x128 kernel of tdnn_layer.hip runs layers 2/3 at 146-147 TFLOP/s (0.93 of the 157.3 fp32 MFMA peak) but
// layer 5 at 138.6 and moves 3.7x the algorithmic bytes there: twelve 128-channel column blocks per row range re-read
// the activations through a 4 MiB L2 that also holds all 3 MB of layer-5 weights.  Here a block
// tile is 256 frames x 256 channels (half the column blocks, half the re-reads), both operands reach LDS by DMA with
// no staging registers, and a K-tile of 32 floats (the same 128-byte row slab as 64 bf16) carries 8x the matrix-pipe
// time of its bf16 counterpart (128 MFMAs of 64 cycles per wave against 16 of 32), so the load segments of the
// ping-pong -- the limit of the bf16 kernel -- vanish behind the MFMA segments and the per-tile epilogue is < 1 % of a tile.
//   * ONE 512-thread block per CU, tile = up to 256 frames x 256 channels, K in 32-wide tiles (128-byte rows)
//   * both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 1-KiB pieces of 8 rows x 128 B; the
//     16-byte-chunk XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied on the SOURCE
//     address.  Two 64-KiB LDS buffers (K-tile parity), refilled slot by slot two K-tiles ahead behind COUNTED
//     s_waitcnt vmcnt.
//   * 8 waves = 2 groups (frames halves) x 4 (64-channel columns); wave tile = MR x 2 accumulators of 32x32 (MR = 2, 3
//     or 4 per tile: 128, 192 or 256 frames).  The two waves of a SIMD belong to different groups and run one barrier
//     apart ("ping-pong").  A ds_read_b128 fragment holds four consecutive k of its row and feeds FOUR MFMAs (lane half
//     h owns k = 8s + 4h .. +3 of k-step s, for A and B alike, so element t of both fragments belongs to MFMA t).
//   * persistent: a block owns a contiguous range of 64-frame units of one 256-channel column, cut into tiles of 4, 3
//     or 2 units.
//   * epilogues: frames in the accumulator's registers, the channel on the lane, and a lane's two accumulators hold
//     ADJACENT channels (a row permutation of the packed weights, pack.hip); store variant: ReLU + folded BatchNorm,
//     8 bytes per lane = whole 256-byte row segments; pooling variant (layer 5): the pivoted partials of
//     tdnn_common.h per (32-frame group, utterance), as the 128x128 kernel writes them.
#include "tdnn_common.h"

namespace xvec {
namespace ppf {

constexpr int kRowB = 128;                        // one K-tile slab of one row: 32 fp32
constexpr int kAccRowB = 32 * kRowB;              // 32 frames: 4 KiB = 4 DMA pieces
constexpr int kABytes = 2 * 4 * kAccRowB;         // [group][acc row][32 frames]: 32 KiB
constexpr int kWBytes = 256 * kRowB;              // 256 channels: 32 KiB = 32 DMA pieces
constexpr int kBufBytes = kABytes + kWBytes;      // 64 KiB per K-tile buffer (K-tile parity); LDS image: A0 | A1 | W0 | W1, so
constexpr int kWOff = 2 * kABytes;                // that BOTH buffers of an operand lie within the 64 KiB reach of a ds_read's
constexpr int kConstOff = 2 * kBufBytes;          // offset field from one base register (tdnn_pp16.hip)
constexpr int kConstBytes = 3 * 256 * 4;          // bias | scale | shift of the block's 256 channels, natural order
constexpr int kLdsBytes = kConstOff + kConstBytes;
constexpr int kThreads = 512;

// a wave-uniform 64-bit value, provably so: kept in a scalar register pair instead of a vector one
__device__ __forceinline__ int64_t uni64(int64_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

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
__device__ unsigned long long g_ppf_diag[2 * 512 * 32];   // [pooling variant][block][group][kind]
__device__ unsigned long long g_ppf_clk[8];               // block 0: s_memtime / s_memrealtime at entry and exit, per variant
#define PP_STAMP(k_)                                                                               \
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
#endif
#ifdef XVEC_KNOCK
// Timing-only knock-outs, compile time (-DXVEC_KNOCK=mask; results are garbage):
//   bit 0: no DMA pieces in the K loop     bit 1: no LDS fragment reads (fragments stay zero)
//   bit 2: no epilogue (stores / pooling)
#define PP_KNOCK_DMA ((XVEC_KNOCK & 1) != 0)
#define PP_KNOCK_RD ((XVEC_KNOCK & 2) != 0)
#define PP_KNOCK_EPI ((XVEC_KNOCK & 4) != 0)
#define PP_KNOCK_RDW ((XVEC_KNOCK & 8) != 0)     // bit 3: no W fragment reads only
#define PP_KNOCK_RDA ((XVEC_KNOCK & 16) != 0)    // bit 4: no A fragment reads only
#else
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
// (this lane's row within its group's acc row 0 and the swizzled 16-byte chunk it fetches are recomputed from an opaque
// lane id here, once per tile, instead of being carried through the K loop: tdnn_pp16.hip)
template <bool RAGGED>
__device__ __forceinline__ void set_rows(const TdnnArgs& a, const Tile& t, int grp, int wc, Stream& st, Rows& out) {
    int lane_;
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
        st.off_next = uni64(next_off(st.u_tile));
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
    const int rb = a.ldx * 4;
    const int base = rl * rb + a_chunk;
    out.av0 = base + (st.u_tile + c0) * a.span * rb;
    out.av1 = base + 32 * rb + (st.u_tile + c1) * a.span * rb;
    out.av2 = base + 64 * rb + (st.u_tile + c2) * a.span * rb;
    out.av3 = base + 96 * rb + (st.u_tile + c3) * a.span * rb;
    out.xrsrc = make_srd(static_cast<const char*>(a.X) + t.m0 * (int64_t)a.ldx * 4);
}

__device__ __forceinline__ void set_rows(const TdnnArgs& a, const Tile& t, int grp, int wc, Stream& st, Rows& out) {
    if (a.out_map.offsets == nullptr) set_rows<false>(a, t, grp, wc, st, out);
    else set_rows<true>(a, t, grp, wc, st, out);
}

// Scalar source offset of the activation K-tiles, stepped one K-tile at a time (taps innermost:
// tap 0, 1, .., then the next 64-channel block): no division in the loop.
struct KPos {
    int tap, so;
};
__device__ __forceinline__ void kstep(const TdnnArgs& a, KPos& k) {
    const int tapstep = a.tap_rows * a.ldx * 4;
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
template <bool POOL>
__device__ __forceinline__ void issue_head2(const TdnnArgs& a, const Stream& st, int mr) {
    KPos k1 = {0, 0};
    kstep(a, k1);
    PP_ISSUE_A01(1, k1.so)
    PP_ISSUE_W(1, 1)
    PP_ISSUE_A23(mr, 1, k1.so)
}

struct Lane {
    int h, r;
    int rd;          // r*128: row part of every fragment read
    int k0, k1, k2, k3;   // swizzled byte offset of this lane's 16-byte chunk for k-steps 0..3
    unsigned a_rd;   // LDS byte offset (buffer 0) of this wave's group's acc row 0, + rd
    unsigned w_rd;   // LDS byte offset (buffer 0) of this wave's channel column 0, + rd
    unsigned a_k0, a_k1, a_k2, a_k3, w_k0, w_k1, w_k2, w_k3;   // the base registers of the fragment reads, opaque (tdnn_pp16.hip)
    int wave, grp, wc;
};

#define PP_RD(dst_, off_) if constexpr (!PP_KNOCK_RD) dst_ = *reinterpret_cast<const float4*>(smem + (off_));
#define PP_RDW(dst_, off_) if constexpr (!PP_KNOCK_RDW) PP_RD(dst_, off_)
// W fragments of K-tile in buffer b_: 2 columns x 4 k-steps
#define PP_READ_W(b_)                                                     \
    {                                                                     \
        constexpr unsigned o_ = (b_) * kWBytes;                           \
        PP_RDW(wf0_0, ln.w_k0 + o_) PP_RDW(wf0_1, ln.w_k1 + o_) PP_RDW(wf0_2, ln.w_k2 + o_) PP_RDW(wf0_3, ln.w_k3 + o_)      \
        PP_RDW(wf1_0, ln.w_k0 + o_ + kAccRowB) PP_RDW(wf1_1, ln.w_k1 + o_ + kAccRowB)                                     \
        PP_RDW(wf1_2, ln.w_k2 + o_ + kAccRowB) PP_RDW(wf1_3, ln.w_k3 + o_ + kAccRowB)                                     \
    }
// A fragments of acc row i_ into fragment set f_
#define PP_READ_A(f_, i_, b_)                                             \
    {                                                                     \
        constexpr unsigned o_ = (b_) * kABytes + (i_) * kAccRowB;         \
        PP_RD(af##f_##_0, ln.a_k0 + o_) PP_RD(af##f_##_1, ln.a_k1 + o_) PP_RD(af##f_##_2, ln.a_k2 + o_) PP_RD(af##f_##_3, ln.a_k3 + o_) \
    }
// one k-step of one accumulator (row i_, column j_), A fragment set f_, k-step s_: FOUR v_mfma_f32_32x32x2_f32, element t
// of both 16-byte fragments (k = 8s + 4h + t) each.  The activations are the MFMA A operand: frames in the accumulator's
// registers, the channel on the lane.
#define PP_MFT(i_, j_, f_, s_, t_)                                                                                 \
    acc##i_##j_ = __builtin_amdgcn_mfma_f32_32x32x2f32(af##f_##_##s_.t_, wf##j_##_##s_.t_, acc##i_##j_, 0, 0, 0);
// Consecutive MFMAs go to DIFFERENT accumulators (element t of the fragments outermost): four in a row on one
// accumulator wait for each other's results -- the first version of this kernel did that and ran layers 2/3 at 131
// TFLOP/s against the 128x128 kernel's 146.
// k-step s_ of acc rows i0_, i1_ (fragment sets of the same number) x both columns: 16 MFMAs on four accumulators
#define PP_MF4T(i0_, i1_, s_, t_) PP_MFT(i0_, 0, i0_, s_, t_) PP_MFT(i0_, 1, i0_, s_, t_) PP_MFT(i1_, 0, i1_, s_, t_) PP_MFT(i1_, 1, i1_, s_, t_)
#define PP_MF4(i0_, i1_, s_) PP_MF4T(i0_, i1_, s_, x) PP_MF4T(i0_, i1_, s_, y) PP_MF4T(i0_, i1_, s_, z) PP_MF4T(i0_, i1_, s_, w)
// k-step s_ of acc row i0_ x both columns: 8 MFMAs on two accumulators
#define PP_MF2T(i0_, s_, t_) PP_MFT(i0_, 0, i0_, s_, t_) PP_MFT(i0_, 1, i0_, s_, t_)
#define PP_MF2(i0_, s_) PP_MF2T(i0_, s_, x) PP_MF2T(i0_, s_, y) PP_MF2T(i0_, s_, z) PP_MF2T(i0_, s_, w)
// the same with two LDS reads in its shadow: r0_ after the first four MFMAs, r1_ after the last one (r1_ may overwrite
// a fragment these very MFMAs use; r0_ never does: see the sequences below).  A ds_read_b128 holds the wave's issue
// for ~30 cycles, free while an MFMA of this wave is executing
#define PP_M2R(i0_, s_, r0_, r1_)                                   \
    PP_MF2T(i0_, s_, x) PP_MF2T(i0_, s_, y) SB(); r0_ SB();        \
    PP_MF2T(i0_, s_, z) PP_MF2T(i0_, s_, w) SB(); r1_ SB();
// one fragment read: acc row i_ (= fragment set i_), k-step s_, from buffer b_
#define PP_RA(i_, s_, b_) if constexpr (!PP_KNOCK_RDA) PP_RD(af##i_##_##s_, ln.a_k##s_ + ((b_) * kABytes + (i_) * kAccRowB))

// One K-tile held in LDS buffer b_ (odd_ = its parity).  k2.so / wq = activation source offset and index of
// the K-tile requested now, two K-tiles ahead; mr_req = height of the tile it belongs to.  The request stream
// does not stop at the end of a tile: in a tile's last two K-tiles ("last") the requests are the NEXT tile's
// K-tiles 0 and 1 (the stream state was switched to that tile just before), so a tile starts with its first
// K-tiles in LDS and its first fragments in registers.  Only a block's last tile requests nothing there.
//   load 0:  read the 8 W fragments; request acc rows 0,1 of K-tile q+2
//   mfma 0:  MR=4: acc rows 0,1 (16 MFMAs); MR=3: acc rows 0,1, k-steps 0-2 (12) -- with the fragment reads of
//            acc rows 2,(3) behind its first MFMAs
//   load 1:  request W and acc rows 2,(3) of K-tile q+2
//   mfma 1:  MR=4: acc rows 2,3 (16); MR=3: acc rows 0,1 k-step 3 then acc row 2 (12) -- with the fragment reads
//            of acc rows 0,1 of K-tile q+1 behind the acc row 2 MFMAs
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
#define PP_KTILE(b_, odd_)                                                          \
    {                                                                               \
        SB();                                                                       \
        PP_READ_W(b_)                                                               \
        SB();                                                                       \
        if (req && !PP_KNOCK_DMA) PP_ISSUE_A01(b_, k2.so)                           \
        SB();                                                                       \
        PP_WAIT_LGKM();                                                             \
        PP_STAMP(0)                                                                 \
        if (!last) { if (MR == 4) { PP_WAIT_VM(10); } else if (MR == 3) { PP_WAIT_VM(9); } else { PP_WAIT_VM(8); } } \
        else if (!(odd_)) { PP_WAIT_VM_RT(req ? MR + 6 : MR + 4) }                  \
        else { PP_WAIT_VM_RT(req ? mr_req + 6 : 0) }                                \
        PP_STAMP(1)                                                                 \
        PP_BARRIER()                                                                \
        PP_STAMP(2)                                                                 \
        __builtin_amdgcn_s_setprio(1);                                              \
        if constexpr (MR == 2) {   /* rows 0,1, k-steps 0,1: nothing to read (no acc row 2) */ \
            PP_MF4(0, 1, 0) SB(); PP_MF4(0, 1, 1) SB();                             \
        } else {                                                                    \
        PP_M2R(0, 0, PP_RA(2, 0, b_), PP_RA(2, 1, b_))   \
        PP_M2R(1, 0, PP_RA(2, 2, b_), PP_RA(2, 3, b_))   \
        }                                                                           \
        if constexpr (MR == 4) {                                                    \
            PP_M2R(0, 1, PP_RA(3, 0, b_), PP_RA(3, 1, b_)) \
            PP_M2R(1, 1, PP_RA(3, 2, b_), PP_RA(3, 3, b_)) \
            PP_MF4(0, 1, 2) SB(); PP_MF4(0, 1, 3) SB();                             \
        } else if constexpr (MR == 3) {                                             \
            PP_MF4(0, 1, 1) SB(); PP_MF4(0, 1, 2) SB();                             \
        }                                                                           \
        __builtin_amdgcn_s_setprio(0);                                              \
        PP_WAIT_LGKM();                                                             \
        PP_STAMP(3)                                                                 \
        PP_BARRIER()                                                                \
        PP_STAMP(4)                                                                 \
        if (req && !PP_KNOCK_DMA) {                                                 \
            PP_ISSUE_W(b_, wq)                                                      \
            PP_ISSUE_A23(mr_req, b_, k2.so)                                         \
        }                                                                           \
        SB();                                                                       \
        PP_STAMP(6)                                                                 \
        if (!last) { if (MR == 4) { PP_WAIT_VM(10); } else if (MR == 3) { PP_WAIT_VM(8); } else { PP_WAIT_VM(6); } } \
        else if (!(odd_)) { PP_WAIT_VM_RT(req ? MR + mr_req + 2 : MR - 2) }         \
        else { PP_WAIT_VM_RT(req ? 2 * mr_req + 2 : 0) }                            \
        PP_STAMP(7)                                                                 \
        PP_BARRIER()                                                                \
        PP_STAMP(8)                                                                 \
        __builtin_amdgcn_s_setprio(1);                                              \
        if constexpr (MR == 3) { PP_MF4(0, 1, 3) SB(); }                            \
        if constexpr (MR == 2) {   /* k-steps 2,3; each read behind the last MFMA that uses the register it overwrites */ \
            PP_M2R(0, 2, PP_RA(0, 0, (b_) ^ 1), PP_RA(0, 1, (b_) ^ 1)) \
            PP_M2R(1, 2, PP_RA(1, 0, (b_) ^ 1), PP_RA(1, 1, (b_) ^ 1)) \
            PP_M2R(0, 3, PP_RA(0, 2, (b_) ^ 1), PP_RA(0, 3, (b_) ^ 1)) \
            PP_M2R(1, 3, PP_RA(1, 2, (b_) ^ 1), PP_RA(1, 3, (b_) ^ 1)) \
        } else {                                                                    \
        /* (after a block's last K-tile these reads fetch stale bytes that nobody uses: cheaper than a branch */ \
        /*  around the MFMAs, which made hipcc keep two register assignments alive and spill) */ \
        PP_M2R(2, 0, PP_RA(0, 0, (b_) ^ 1), PP_RA(0, 1, (b_) ^ 1)) \
        PP_M2R(2, 1, PP_RA(0, 2, (b_) ^ 1), PP_RA(0, 3, (b_) ^ 1)) \
        PP_M2R(2, 2, PP_RA(1, 0, (b_) ^ 1), PP_RA(1, 1, (b_) ^ 1)) \
        PP_M2R(2, 3, PP_RA(1, 2, (b_) ^ 1), PP_RA(1, 3, (b_) ^ 1)) \
        if constexpr (MR == 4) {                                                    \
            PP_MF2(3, 0) PP_MF2(3, 1) PP_MF2(3, 2) PP_MF2(3, 3) SB();               \
        }                                                                           \
        }                                                                           \
        __builtin_amdgcn_s_setprio(0);                                              \
        PP_WAIT_LGKM();                                                             \
        PP_STAMP(9)                                                                 \
        PP_BARRIER()                                                                \
        PP_STAMP(10)                                                                \
        kstep(a, k2);                                                               \
        ++wq;                                                                       \
    }

// Fused statistics pooling for the pooling variant (main.py:59-63), frames in the accumulator registers and
// the channel on the lane.  The epilogue runs in the open here (both waves of a SIMD are in it at the same
// time, the matrix pipe idles), so it is as short as the arithmetic allows: per (32-frame group, utterance)
// and channel the pivoted sums S1 = sum (r - K), S2 = sum (r - K)^2 of r = relu(z + bias) over the utterance's
// frames in the group, K = the group's frame 0 (tdnn_common.h, pool_group_impl: why a pivot) -- one v_max, half a
// v_pk_add for the pivot, half a v_pk_add and half a v_pk_fma per value.  Scale and shift of the folded BatchNorm
// are applied by pool_finalize.
// v0 / v1: this wave's two accumulators for the group at compact row row_g -- the lane's channels col0 and
// col0 + 1 (bias already inside: the accumulators start at it).
// Returns true when the group lay inside one utterance.
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
template <bool RAGGED>
__device__ __forceinline__ bool pool_raw_pair(const TdnnArgs& a, f32x16& v0, f32x16& v1, int64_t row_g,
                                              int h, int col0, PoolCur& pc) {
    const RowMap& m = a.out_map;
    // r = relu(z + bias) IN PLACE (as an expression in both paths below hipcc computes the 32 values up front into 32
    // more registers, next to 128 live accumulators: tdnn_pp16.hip)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        v0[e] = fmaxf(v0[e], 0.f);
        v1[e] = fmaxf(v1[e], 0.f);
    }
    while (pc.end <= row_g && pc.u < m.n_utts - 1) {
        pc.u = __builtin_amdgcn_readfirstlane(pc.u + 1);
        pc.end = uni64(first_row<RAGGED>(m, pc.u + 1));
    }
    const int64_t grp = row_g >> 5;
    const int ld = a.ldy;
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(a.pool_part);
    // pivots: the group's frame 0 of the lane's two channels (tdnn_common.h, pool_group_impl)
    const float ka = lower_half(v0[0]), kb = lower_half(v1[0]);
    if (pc.end >= row_g + 32) {               // the whole group belongs to utterance pc.u
        // plain v_sub / v_add / v_fma in four interleaved chains (v_pk_*_f32 issue at ~17 cycles each: tdnn_pp16.hip)
        float p1a0 = 0.f, p1a1 = 0.f, p2a0 = 0.f, p2a1 = 0.f, p1b0 = 0.f, p1b1 = 0.f, p2b0 = 0.f, p2b1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const float da0 = v0[e] - ka, da1 = v0[e + 1] - ka;
            const float db0 = v1[e] - kb, db1 = v1[e + 1] - kb;
            p1a0 += da0; p1a1 += da1; p1b0 += db0; p1b1 += db1;
            p2a0 = fmaf(da0, da0, p2a0); p2a1 = fmaf(da1, da1, p2a1);
            p2b0 = fmaf(db0, db0, p2b0); p2b1 = fmaf(db1, db1, p2b1);
        }
        const float s1a = add_halves(p1a0 + p1a1), s2a = add_halves(p2a0 + p2a1);
        const float s1b = add_halves(p1b0 + p1b1), s2b = add_halves(p2b0 + p2b1);
        store_partial2(prs, ld, grp + pc.u, h, col0, ka, kb, s1a, s2a, s1b, s2b);
        return true;
    }
    for (int u = pc.u; u < m.n_utts; u = __builtin_amdgcn_readfirstlane(u + 1)) {   // the group straddles utterances
        const int64_t off = first_row<RAGGED>(m, u);
        if (off >= row_g + 32) break;
        const int64_t end = first_row<RAGGED>(m, u + 1);
        const int64_t lo_r = off > row_g ? off : row_g;
        const int64_t hi_r = end < row_g + 32 ? end : row_g + 32;
        if (hi_r <= lo_r) continue;
        const int lo_l = (int)(lo_r - row_g), hi_l = (int)(hi_r - row_g);
        const unsigned below_hi = hi_l >= 32 ? 0xffffffffu : ((1u << hi_l) - 1u);
        const unsigned lm = (below_hi & ~((1u << lo_l) - 1u)) >> (4 * h);   // this lane's rows: bits (e&3) + 8*(e>>2)
        float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
        float kma = ka, kmb = kb;            // opaque copies: no sub-expressions shared with the unmasked path
        asm volatile("" : "+v"(kma), "+v"(kmb));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool in = (lm >> ((e & 3) + 8 * (e >> 2))) & 1u;      // a SELECT: rows outside may hold anything
            const float da = in ? v0[e] - kma : 0.f, db = in ? v1[e] - kmb : 0.f;
            s1a += da;
            s2a = fmaf(da, da, s2a);
            s1b += db;
            s2b = fmaf(db, db, s2b);
        }
        s1a = add_halves(s1a);
        s2a = add_halves(s2a);
        s1b = add_halves(s1b);
        s2b = add_halves(s2b);
        store_partial2(prs, ld, grp + u, h, col0, ka, kb, s1a, s2a, s1b, s2b);
    }
    return false;
}

// One tile: K loop, request of the next tile's first K-tiles, epilogue.
template <int MR, bool POOL>
__device__ __forceinline__ void process_tile(const TdnnArgs& a, char* smem, Stream& st, const Lane& ln,
                                             const Tile& t, const Tile& nxt, bool has_next, bool first, int n0, int nk,
                                             PoolCur& pc) {
    // source rows of the NEXT tile (its first K-tiles are requested during this tile's last two): worked out
    // here, before the accumulators exist, and parked in four registers
    Rows rows_next = st.cur;
    if (has_next) set_rows(a, nxt, ln.grp, ln.wc, st, rows_next);
    // the accumulators start at the bias of their lane's two channels (64*wc + 2r, +1 of the block's column; the
    // constants live in a 3-KiB LDS table: six registers held across the K loop were what a third tile height cost)
    const float* cst = reinterpret_cast<const float*>(smem + kConstOff) + ln.wc * 64 + 2 * ln.r;
    f32x16 acc00, acc01, acc10, acc11, acc20, acc21, acc30, acc31;
    {
        const float2 bi = *reinterpret_cast<const float2*>(cst);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc00[e] = bi.x; acc01[e] = bi.y; acc10[e] = bi.x; acc11[e] = bi.y;
            acc20[e] = bi.x; acc21[e] = bi.y; acc30[e] = bi.x; acc31[e] = bi.y;
        }
    }
    float4 wf0_0, wf0_1, wf0_2, wf0_3, wf1_0, wf1_1, wf1_2, wf1_3;
    float4 af0_0, af0_1, af0_2, af0_3, af1_0, af1_1, af1_2, af1_3;     // acc rows 0,1 (read during the previous mfma 1)
    float4 af2_0, af2_1, af2_2, af2_3, af3_0, af3_1, af3_2, af3_3;     // acc rows 2,3 (read during mfma 0)
#ifdef XVEC_DIAG
    unsigned long long dsum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dprev, dstart;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dstart)::"memory");
#endif

#ifdef XVEC_KNOCK
    if (PP_KNOCK_RD || PP_KNOCK_RDW || PP_KNOCK_RDA) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        wf0_0 = wf0_1 = wf0_2 = wf0_3 = wf1_0 = wf1_1 = wf1_2 = wf1_3 = z;
        af0_0 = af0_1 = af0_2 = af0_3 = af1_0 = af1_1 = af1_2 = af1_3 = z;
        af2_0 = af2_1 = af2_2 = af2_3 = af3_0 = af3_1 = af3_2 = af3_3 = z;
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
    PP_RA(0, 0, 0) PP_RA(0, 1, 0) PP_RA(0, 2, 0) PP_RA(0, 3, 0)
    PP_RA(1, 0, 0) PP_RA(1, 1, 0) PP_RA(1, 2, 0) PP_RA(1, 3, 0)
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
    KPos k2 = {0, 0};                       // K-tile requested now (two ahead of the one computed), its W index,
    kstep(a, k2);                           // the height of its tile, and whether there is anything to request
    kstep(a, k2);
    int wq = 2;
    int mr_req = MR;
    bool req = true;
    for (int q = 0; q < nk; q += 2) {
        const bool last = q + 2 >= nk;
        if (last) {                         // from here on the requests are the next tile's K-tiles 0 and 1
            req = has_next;                 // (a peeled copy of the last pair made hipcc spill ~250 registers)
            if (has_next) {
                st.cur = rows_next;
                mr_req = nxt.mr;
                k2.tap = 0;
                k2.so = 0;
                wq = 0;
            }
        }
        PP_KTILE(0, false)
        PP_KTILE(1, true)
    }
    if (ln.grp == 0) PP_BARRIER()
    PP_STAMP(13)                            // tail barrier
    PP_STAMP(5)

    const int64_t row0 = t.m0 + ln.grp * 32 * MR;
    if constexpr (PP_KNOCK_EPI) {
        asm volatile("" ::"v"(acc00), "v"(acc01), "v"(acc10), "v"(acc11), "v"(acc20), "v"(acc21), "v"(acc30), "v"(acc31));
    } else if constexpr (!POOL) {
        // ReLU + folded BatchNorm (tdnn_layer.py:30-39); the lane's two channels are adjacent (pack.hip): 8 bytes per lane,
        // a store instruction writes two whole 256-byte row segments (lane halves = rows 4 apart).  Element e of an
        // accumulator = frame (e&3) + 8*(e>>2) + 4*h.
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const __amdgpu_buffer_rsrc_t yrsrc = make_rsrc(static_cast<char*>(a.Y) + (t.m0 * (int64_t)a.ldy + n0) * 4);
        int lane_e;           // opaque lane id (tdnn_pp16.hip)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int y_voff = (4 * (lane_e >> 5) * a.ldy + ln.wc * 64 + 2 * (lane_e & 31)) * 4;
        const float* cst_e = reinterpret_cast<const float*>(smem + kConstOff) + ln.wc * 64 + 2 * (lane_e & 31);
        const float2 sc = *reinterpret_cast<const float2*>(cst_e + 256), sh = *reinterpret_cast<const float2*>(cst_e + 512);
#define PP_STORE(i_)                                                                                   \
        if (MR > i_ && row0 + 32 * i_ < t.valid_end) {                                                    \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                              \
                const float v0 = fmaf(fmaxf(acc##i_##0[e], 0.f), sc.x, sh.x);                             \
                const float v1 = fmaf(fmaxf(acc##i_##1[e], 0.f), sc.y, sh.y);                             \
                const u32x2 pk = {__float_as_uint(v0), __float_as_uint(v1)};                              \
                __builtin_amdgcn_raw_buffer_store_b64(pk, yrsrc, y_voff,                                  \
                                                      (ln.grp * 32 * MR + 32 * i_ + (e & 3) + 8 * (e >> 2)) * a.ldy * 4, 0); \
            }                                                                                             \
        }
        PP_STORE(0) PP_STORE(1) PP_STORE(2) PP_STORE(3)
#undef PP_STORE
    } else {
        const int col0 = n0 + ln.wc * 64 + 2 * ln.r;
#define PP_POOL(RG_, i_)                                                                               \
        if (MR > i_ && row0 + 32 * i_ < t.valid_end) {                                                    \
            pool_raw_pair<RG_>(a, acc##i_##0, acc##i_##1, row0 + 32 * i_, ln.h, col0, pc);                \
        }
        if (a.out_map.offsets == nullptr) {
            PP_POOL(false, 0) PP_POOL(false, 1) PP_POOL(false, 2) PP_POOL(false, 3)
        } else {
            PP_POOL(true, 0) PP_POOL(true, 1) PP_POOL(true, 2) PP_POOL(true, 3)
        }
#undef PP_POOL
    }
#ifdef XVEC_DIAG
    {
        unsigned long long dend;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dend)::"memory");
        dsum[11] += dend - dprev;          // epilogue + second K-tile of the next tile
        dsum[14] += 1;
        dsum[15] += (unsigned long long)MR;
        if ((ln.wave & 3) == 0 && ln.r == 0 && ln.h == 0 && blockIdx.x < 512) {
            _Pragma("unroll") for (int k = 0; k < 16; ++k) g_ppf_diag[(POOL ? 512 * 32 : 0) + blockIdx.x * 32 + ln.grp * 16 + k] += dsum[k];
        }
    }
#endif
}

template <bool POOL>
__global__ __launch_bounds__(kThreads, 2) void tdnn_pp_kernel(const TdnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int jcol = lid % a.n_tiles;                   // 256-channel column
    const int prange = lid / a.n_tiles;
    const int64_t u_begin = a.groups_total * (int64_t)prange / a.blocks_per_col;     // 64-frame units
    const int64_t u_end = a.groups_total * (int64_t)(prange + 1) / a.blocks_per_col;
    const int n0 = jcol * 256;
    const int nk = a.n_taps * a.cpt;                    // K-tiles of 64 (even)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef XVEC_DIAG
    if (blockIdx.x == 0 && tid == 0) {
        g_ppf_clk[(POOL ? 4 : 0) + 0] = __builtin_amdgcn_s_memtime();
        g_ppf_clk[(POOL ? 4 : 0) + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    Lane ln;
    ln.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ln.grp = ln.wave >> 2;
    ln.wc = ln.wave & 3;
    ln.h = lane >> 5;
    ln.r = lane & 31;
    ln.rd = ln.r * kRowB;
    {
        const int sw = (ln.r >> 1) & 7;
        ln.k0 = ((0 + ln.h) ^ sw) << 4;
        ln.k1 = ((2 + ln.h) ^ sw) << 4;
        ln.k2 = ((4 + ln.h) ^ sw) << 4;
        ln.k3 = ((6 + ln.h) ^ sw) << 4;
    }
    ln.a_rd = ln.grp * 4 * kAccRowB + ln.rd;
    ln.w_rd = kWOff + ln.wc * 2 * kAccRowB + ln.rd;
    ln.a_k0 = ln.a_rd + ln.k0; ln.a_k1 = ln.a_rd + ln.k1; ln.a_k2 = ln.a_rd + ln.k2; ln.a_k3 = ln.a_rd + ln.k3;
    ln.w_k0 = ln.w_rd + ln.k0; ln.w_k1 = ln.w_rd + ln.k1; ln.w_k2 = ln.w_rd + ln.k2; ln.w_k3 = ln.w_rd + ln.k3;
    asm volatile("" : "+v"(ln.a_k0), "+v"(ln.a_k1), "+v"(ln.a_k2), "+v"(ln.a_k3));
    asm volatile("" : "+v"(ln.w_k0), "+v"(ln.w_k1), "+v"(ln.w_k2), "+v"(ln.w_k3));

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
        st.lds_a = (unsigned)(unsigned long long)(lds_ptr)(smem) + ln.grp * 4 * kAccRowB + ln.wc * 1024;
        const int wr = ln.wave * 8 + prow;                                // W: channel row of piece t = wr + 64*t
        const int w_chunk = (ppos ^ ((wr >> 1) & 7)) * 16;                // ((wr + 64t) >> 1) & 7 is the same for every t
        st.lds_w = (unsigned)(unsigned long long)(lds_ptr)(smem) + kWOff + ln.wave * 1024;
        st.wv0 = wr * kRowB + w_chunk;                                    // K-tile major: rows 128 B apart
        st.w64 = 64 * kRowB;
        st.wrsrc = make_srd(static_cast<const char*>(a.W) + (int64_t)jcol * nk * kWBytes);
        st.cur.av0 = st.cur.av1 = st.cur.av2 = st.cur.av3 = 0;
        st.u_tile = __builtin_amdgcn_readfirstlane(utt_of_row(a.out_map, u_begin * 64));
        st.off_next = uni64(row_off(a.out_map, st.u_tile + 1));
        st.cur.xrsrc = st.wrsrc;
        PoolCur pc;
        pc.u = 0;
        pc.end = 0;

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
        if (POOL) pc = pool_cursor(a, cur.m0 + ln.grp * 32 * cur.mr);
        set_rows(a, cur, ln.grp, ln.wc, st, st.cur);
        __syncthreads();                                   // constants visible; nobody reads LDS buffers yet
        issue_head1<POOL>(a, st, cur.mr);
        issue_head2<POOL>(a, st, cur.mr);
        for (int idx = 0; idx < nt; ++idx) {
            const bool has_next = idx + 1 < nt;
            Tile nxt = cur;
            if (has_next) nxt = tile_at(idx + 1, cur.m0 + 64 * cur.mr);
            if (cur.mr == 4)
                process_tile<4, POOL>(a, smem, st, ln, cur, nxt, has_next, idx == 0, n0, nk, pc);
            else if (cur.mr == 3)
                process_tile<3, POOL>(a, smem, st, ln, cur, nxt, has_next, idx == 0, n0, nk, pc);
            else
                process_tile<2, POOL>(a, smem, st, ln, cur, nxt, has_next, idx == 0, n0, nk, pc);
            cur = nxt;
        }
#ifdef XVEC_DIAG
        if (blockIdx.x == 0 && tid == 0) {
            g_ppf_clk[(POOL ? 4 : 0) + 2] = __builtin_amdgcn_s_memtime();
            g_ppf_clk[(POOL ? 4 : 0) + 3] = __builtin_amdgcn_s_memrealtime();
        }
#endif
    }
}

}  // namespace ppf

#ifdef XVEC_DIAG
extern "C" int xvec_ppf_clk_read(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ppf::g_ppf_clk), 8 * 8);
}
extern "C" int xvec_ppf_diag_read(unsigned long long* host, int n_words, int reset) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(ppf::g_ppf_diag), (size_t)n_words * 8);
    if (e == hipSuccess && reset) e = hipMemset(nullptr, 0, 0);
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(ppf::g_ppf_diag)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 2 * 512 * 32);
    }
    return (int)e;
}
#endif

hipError_t launch_tdnn_ppf(const TdnnArgs& a, bool pool, hipStream_t s) {
    if (a.groups_total <= 0 || a.blocks_per_col <= 0 || a.blocks_per_col > a.groups_total || (a.cpt & 1) ||
        a.n_tiles <= 0)
        return hipErrorInvalidValue;
    const int grid = a.blocks_per_col * a.n_tiles;
    if (pool) {
        static LdsOptIn opt;
        if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(ppf::tdnn_pp_kernel<true>), ppf::kLdsBytes); e != hipSuccess)
            return e;
        ppf::tdnn_pp_kernel<true><<<dim3(grid), dim3(ppf::kThreads), ppf::kLdsBytes, s>>>(a);
    } else {
        static LdsOptIn opt;
        if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(ppf::tdnn_pp_kernel<false>), ppf::kLdsBytes); e != hipSuccess)
            return e;
        ppf::tdnn_pp_kernel<false><<<dim3(grid), dim3(ppf::kThreads), ppf::kLdsBytes, s>>>(a);
    }
    return hipGetLastError();
}

}  // namespace xvec

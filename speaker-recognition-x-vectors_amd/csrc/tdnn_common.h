// Device helpers shared by the frame-level kernels (tdnn_layer.hip, tdnn_bf16.hip).
#pragma once
#include "xvec_internal.h"

namespace xvec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks b, b+8, b+16.. share an XCD (round-robin dispatch); give each XCD a
    // contiguous run of logical ids (bijective for any nwg).
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// x + (x of lane ^ 32) in every lane: v_permlane32_swap exchanges the upper half of one copy with
// the lower half of the other in the vector ALU (ds_bpermute would be an LDS round trip plus a
// wait in the middle of the epilogue)
__device__ __forceinline__ float add_halves(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_n(const void* p, int num_bytes) {
    // uniform by construction (kernel argument + blockIdx-derived offset); readfirstlane makes
    // that provable so hipcc emits no waterfall loop around the buffer loads
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, (short)0, __builtin_amdgcn_readfirstlane(num_bytes), 0x00020000);
}

// descriptor without a range limit (the buffers behind it are padded for every over-read)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) { return make_rsrc_n(p, 0x7fffffff); }

// descriptor of base + off that ends where the buffer of `total` bytes ends: the hardware range
// check then returns zeros for every byte past the end, with no instruction spent on it
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_bounded(const void* base, int64_t off, int64_t total) {
    int64_t left = total - off;
    left = left < 0 ? 0 : (left > 0x7fffffff ? 0x7fffffff : left);
    return make_rsrc_n(static_cast<const char*>(base) + off, (int)left);
}

// ReLU + folded BatchNorm on one accumulator whose REGISTERS are channels (store variant): element e =
// channel (e&3) + 8*(e>>2) + 4*h of the 32-channel block.  The bias is already inside (the accumulators
// start at it); scale / shift of the lane's 16 channels are passed in registers, read from LDS once per
// column and tile (an LDS read holds a wave's issue ~30-40 cycles: twelve per accumulator were the
// largest single item of this epilogue).
__device__ __forceinline__ void store_acc(const f32x16& v, const float4 (&sc)[4], const float4 (&sh)[4],
                                          __amdgpu_buffer_rsrc_t yrsrc, int y_voff, int y_soff) {
    unsigned pk[8];
    typedef float f32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const float y0 = fmaf(fmaxf(v[4 * gq + 0], 0.f), sc[gq].x, sh[gq].x);
        const float y1 = fmaf(fmaxf(v[4 * gq + 1], 0.f), sc[gq].y, sh[gq].y);
        const float y2 = fmaf(fmaxf(v[4 * gq + 2], 0.f), sc[gq].z, sh[gq].z);
        const float y3 = fmaf(fmaxf(v[4 * gq + 3], 0.f), sc[gq].w, sh[gq].w);
        pk[2 * gq] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{y0, y1}, bf16x2));
        pk[2 * gq + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{y2, y3}, bf16x2));
    }
    // lane halves hold channels 8g..8g+3 (h=0) and 8g+4..8g+7 (h=1) of register group g: swapping the
    // upper half of group g with the lower half of group g+1 leaves 8 consecutive channels per lane
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        auto r0 = __builtin_amdgcn_permlane32_swap(pk[4 * pr + 0], pk[4 * pr + 2], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(pk[4 * pr + 1], pk[4 * pr + 3], false, false);
        const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, y_voff, y_soff + pr * 32, 0);
        // A 128-bit buffer store reads its data registers a cycle or two after it issues; a vector instruction
        // that overwrites them in the very next slot wins the race in some lanes.  hipcc (ROCm 7.2) pads that
        // hazard only when the store's soffset is NOT a register -- with the row offset in an SGPR, as here, it
        // scheduled `v_max_f32 v24, ...` straight behind `buffer_store_dwordx4 v[24:27], ..., s70 offen` and
        // lanes 12-15/28-31 of ~1 row in 10^4 stored the NEXT accumulator's raw bits (profiles/diag/swap_debug.py).
        // The data registers are kept alive across one wait state:
        asm volatile("s_nop 1" ::"v"(o));
    }
}

// x of lane (l & 31) in every lane (the lower half's value broadcast to both halves): v_permlane32_swap of a
// register with itself returns {lower | lower, upper | upper}
__device__ __forceinline__ float lower_half(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]);
}

// A pooling partial = three planes of n_pad floats per slot (32-row group + utterance): K | S1 | S2 with K the
// partial's pivot (pool_group_impl) and S1 = sum (r - K), S2 = sum (r - K)^2 over the utterance's frames in the group.
constexpr int kPoolPlanes = 3;

// Two of the three floats of one column's partial: as a buffer store (scalar slot offset + one 32-bit lane offset):
// a plain `part[...] = v` costs a 64-bit address in two VGPRs per store, in an epilogue that has none to spare.
// (The buffer is < 2 GiB: forward_rows.)  Both lane halves hold every value after add_halves / lower_half: half 0
// stores the S1 plane and half 1 the S2 plane with ONE instruction (two whole 128-byte segments), then half 0
// alone stores the K plane.
__device__ __forceinline__ void store_partial(__amdgpu_buffer_rsrc_t prs, int ld, int64_t slot, int h, int col, float k,
                                              float v0, float v1) {
    const int soff = (int)(slot * kPoolPlanes * ld) * 4;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h ? v1 : v0), prs, (col + (1 + h) * ld) * 4, soff, 0);
    if (h == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(k), prs, col * 4, soff, 0);
}

// The same for TWO adjacent columns (col, col+1) held by one lane: 8-byte stores, 256 contiguous bytes per lane
// half.  col is even.
__device__ __forceinline__ void store_partial2(__amdgpu_buffer_rsrc_t prs, int ld, int64_t slot, int h, int col, float ka,
                                               float kb, float s1a, float s2a, float s1b, float s2b) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int soff = (int)(slot * kPoolPlanes * ld) * 4;
    const u32x2 v = {__float_as_uint(h ? s2a : s1a), __float_as_uint(h ? s2b : s1b)};
    __builtin_amdgcn_raw_buffer_store_b64(v, prs, (col + (1 + h) * ld) * 4, soff, 0);
    if (h == 0) {
        const u32x2 kv = {__float_as_uint(ka), __float_as_uint(kb)};
        __builtin_amdgcn_raw_buffer_store_b64(kv, prs, col * 4, soff, 0);
    }
}

// Pooling cursor of a block: the utterance that holds the first row of the next 32-row group, and
// the compact row where it ends.  Rows only grow along a block's range, so it advances with a few
// scalar steps per group instead of a 64-bit division / binary search each time.
struct PoolCur {
    int u;
    int64_t end;
};

// Row offset of a ragged batch through the SCALAR cache.  hipcc loads `offsets[u]` with a vector load even
// for a provably uniform u (the pointer is not known to be read-only), and the s_waitcnt vmcnt(0) it puts
// behind that load drains every vector-memory operation the wave has in flight (the staging loads of the
// K loop, the LDS-DMA queue of tdnn_pp16.hip); the array is written by the host before the launch.
__device__ __forceinline__ int64_t sload_i64(const int64_t* p) {
    int64_t v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// first compact row of utterance u; u is made provably wave-uniform so that everything derived from it
// (64-bit row numbers, masks) lives in scalar registers
template <bool RAGGED>
__device__ __forceinline__ int64_t pool_first_row(const RowMap& m, int u) {
    u = __builtin_amdgcn_readfirstlane(u);
    if (RAGGED) return sload_i64(m.offsets + u) - (int64_t)u * m.cum;
    return (int64_t)u * (m.fixed_T - m.cum);
}

// max(x, m) as ONE instruction: fmaxf() on a value hipcc cannot prove canonical (an MFMA result) becomes
// v_max t, x, x; v_max r, m, t
__device__ __forceinline__ float max1(float x, float m) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
    return r;
}

// Fused statistics-pooling partial (main.py:59-63) of one 32-row group held in one accumulator: for every
// utterance overlapping compact rows [row_g, row_g+32), the sums S1 = sum (r - K), S2 = sum (r - K)^2 of this
// lane's column over the utterance's frames in the group, r = relu(z + bias), about a PIVOT K.  pool_finalize
// re-bases every partial's sums to the utterance's first pivot in fp64, and applies the folded BatchNorm there
// (mean = shift + scale*mean_r, std = |scale|*sqrt(M2_r/(n-1))).
// Conditioning: torch.std (main.py:61) is two-pass.  Raw fp32 sums (sum r, sum r^2), which round 2 used, lose
// ~1e-7*(mean/std)^2 of the variance inside every 32-row partial whatever the precision of the merge -- 1.5e-4
// of the std at mean/std = 40, an always-on low-variance post-ReLU channel (VERDICT r02 / ADVICE r02).  About a
// sample of the same channel the sums are of deviations: the loss is ~1e-7*(1 + ((mean - K)/std)^2), and K is
// within a few std of the mean.
// Round 4: ONE pivot per (block, channel) -- the block's first frame (this utterance's or a neighbour's frame of
// the same channel) -- and it rides in the accumulators: from the block's second tile on they start at
// bias - K (tdnn_layer.hip, process_tile), so v = z + bias - K and a deviation is d = max(v, -K): one v_max, one
// v_add and one v_fma per value (round 3: a per-group pivot subtracted here, after bias and ReLU: five).
// negk = -K; both lane halves hold it.
template <bool RAGGED>
__device__ __forceinline__ void pool_group_impl(const TdnnArgs& a, const f32x16& v, float negk, int64_t row_g, int h,
                                                int col, PoolCur& pc) {
    const int64_t grp = row_g >> 5;
    const RowMap& m = a.out_map;
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(a.pool_part);
    while (pc.end <= row_g && pc.u < m.n_utts - 1) {
        pc.u = __builtin_amdgcn_readfirstlane(pc.u + 1);
        pc.end = pool_first_row<RAGGED>(m, pc.u + 1);
    }
    const float K = -negk;
    if (pc.end >= row_g + 32) {               // whole group inside utterance pc.u: no masks
        // plain v_max / v_add / v_fma in four interleaved chains: v_pk_add_f32 / v_pk_fma_f32, which round 2 used here
        // ("two values per instruction"), issue at ~17 cycles each (MI355X_MICROARCH.md, price of fillers beside
        // MFMAs) -- twice the cost of the two scalar instructions they replace, in an epilogue that only issues in the
        // gaps of the partner wave's MFMA stream
        float a0 = max1(v[0], negk), a1 = max1(v[1], negk), a2 = max1(v[2], negk), a3 = max1(v[3], negk);
        float b0 = a0 * a0, b1 = a1 * a1, b2 = a2 * a2, b3 = a3 * a3;
#pragma unroll
        for (int e = 4; e < 16; e += 4) {
            const float d0 = max1(v[e], negk), d1 = max1(v[e + 1], negk), d2 = max1(v[e + 2], negk), d3 = max1(v[e + 3], negk);
            a0 += d0; a1 += d1; a2 += d2; a3 += d3;
            b0 = fmaf(d0, d0, b0); b1 = fmaf(d1, d1, b1); b2 = fmaf(d2, d2, b2); b3 = fmaf(d3, d3, b3);
        }
        const float s1 = add_halves((a0 + a1) + (a2 + a3)), s2 = add_halves((b0 + b1) + (b2 + b3));
        store_partial(prs, a.ldy, grp + pc.u, h, col, K, s1, s2);
        return;
    }
    for (int u = pc.u; u < m.n_utts; u = __builtin_amdgcn_readfirstlane(u + 1)) {
        const int64_t off = pool_first_row<RAGGED>(m, u);
        if (off >= row_g + 32) break;
        const int64_t end = pool_first_row<RAGGED>(m, u + 1);
        const int64_t lo_r = off > row_g ? off : row_g;
        const int64_t hi_r = end < row_g + 32 ? end : row_g + 32;
        if (hi_r <= lo_r) continue;
        const int lo_l = (int)(lo_r - row_g), hi_l = (int)(hi_r - row_g);   // local rows [lo_l, hi_l), 0 <= lo_l < hi_l <= 32
        // bit r of rowmask = local row r belongs to utterance u; this lane's rows are
        // (e&3) + 8*(e>>2) + 4*h, i.e. bits (e&3) + 8*(e>>2) of the mask shifted by 4*h
        const unsigned below_hi = hi_l >= 32 ? 0xffffffffu : ((1u << hi_l) - 1u);
        const unsigned rowmask = below_hi & ~((1u << lo_l) - 1u);
        const unsigned lm = rowmask >> (4 * h);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            // a SELECT, not a 0/1 weight: rows outside the utterance may be rows no layer wrote
            // (the tail of the last 32-row group), and 0 * Inf would poison the sums
            const float d = ((lm >> ((e & 3) + 8 * (e >> 2))) & 1u) ? max1(v[e], negk) : 0.f;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
        s1 = add_halves(s1);
        s2 = add_halves(s2);
        store_partial(prs, a.ldy, grp + u, h, col, K, s1, s2);
    }
}

// (two code paths: see set_tile_rows in tdnn_layer.hip)
__device__ __forceinline__ void pool_group(const TdnnArgs& a, const f32x16& v, float negk, int64_t row_g, int h, int col,
                                           PoolCur& pc) {
    if (a.out_map.offsets == nullptr) pool_group_impl<false>(a, v, negk, row_g, h, col, pc);
    else pool_group_impl<true>(a, v, negk, row_g, h, col, pc);
}

// cursor for a block whose first group starts at compact row `row`
__device__ __forceinline__ PoolCur pool_cursor(const TdnnArgs& a, int64_t row) {
    PoolCur pc;
    pc.u = __builtin_amdgcn_readfirstlane(utt_of_row(a.out_map, row));
    pc.end = row_off(a.out_map, pc.u + 1);
    return pc;
}

}  // namespace xvec

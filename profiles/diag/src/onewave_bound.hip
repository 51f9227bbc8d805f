// Bound for VERDICT r05 item 1(a): the bf16 K loop as ONE wave per SIMD with a 256-register accumulator tile in AGPRs.
// Timing only (nothing is checked, nothing is stored): what rate would a K loop of that shape reach on this chip, next to the
// shipped two-waves-per-SIMD ping-pong loop of csrc/tdnn_pp16.hip with the same things knocked out (XVEC_KNOCK builds)?
//
// Shape.  One 256-thread block per CU, block tile 256 frames x 256 channels, 2 x 2 waves, wave tile 128 x 128 on
// v_mfma_f32_16x16x32_bf16: 8 frame blocks x 8 channel blocks x 4 registers = 256 accumulator registers ("+a": AGPRs).
// Per 32-deep k-step and wave: 64 MFMAs, 8 + 8 fragment reads (ds_read_b128) -- 0.25 reads per MFMA against the shipped
// loop's 0.375 (wave tile 128 x 64: 8 + 4 per 32).  K in STAGES of 32 (64-byte rows) through a ring of four 32-KiB LDS slots
// (A 16 KiB | W 16 KiB), both operands by LDS-DMA in 1-KiB pieces of 16 rows x 64 B (8 pieces per wave and stage), requested
// three stages ahead behind a counted vmcnt, ONE barrier per stage.  A fragment's 1 KiB is its DMA piece; the 16-byte chunk
// swizzle q ^ 3 (r >> 3) on the SOURCE address makes the ds_read_b128 conflict-free (MI355X_MICROARCH.md, LDS: the four
// lanes of a 16-lane service group that share a 64-byte segment get four different chunks).
// In the one instruction stream of a wave: an A fragment two rows ahead and a next-stage W fragment behind MFMAs of every row,
// one DMA piece per row.
//   MODE bit 0: fragment reads in the loop     bit 1: DMA pieces in the loop (else the ring keeps its first fill)
// Operand statistics: A = relu(N(0,1)) as the stored activations are (half zeros), W = U(+-0.05); all-zero with argv "zero".
// build: hipcc -O3 --offload-arch=gfx950 onewave_bound.hip -o onewave_bound
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int kSlot = 32 * 1024;       // A 16 KiB | W 16 KiB
constexpr int kWOff = 16 * 1024;
constexpr int kLds = 4 * kSlot;
constexpr int kRowStride = 1024;       // activation row: 512 channels bf16

#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ i32x4 make_srd(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    i32x4 d;
    d.x = (int)__builtin_amdgcn_readfirstlane((unsigned)v);
    d.y = (int)(__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) & 0xffffu);
    d.z = 0x7fffffff;
    d.w = 0x00020000;
    return d;
}
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned dst, int voff, int soff) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff)
        : "memory", "m0");
}

#define MF(i_, j_, af_, wf_) \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(af_), "v"(wf_));

template <int MODE, int JS>
__device__ __forceinline__ void onewave_body(char* smem, const char* __restrict__ A, const char* __restrict__ W, float* __restrict__ out,
                                             int n_stage4, long a_block_bytes, int a_blocked, int a_share) {
    constexpr bool RD = (MODE & 1) != 0, DMA = (MODE & 2) != 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    // fragment read offsets within a slot (+ i * 1024 per block): A rows 128 wr + 16 i + r, W rows 128 wc + 16 j + r
    const unsigned frag = r * 64 + ((q ^ (3 * (r >> 3))) << 4);
    unsigned a_rd = wr * 8 * 1024 + frag, w_rd = kWOff + wc * 8 * 1024 + frag;
    asm volatile("" : "+v"(a_rd), "+v"(w_rd));
    // DMA: piece = 16 rows x 64 B; lane -> row lane >> 2, LDS position lane & 3 holds source chunk (lane & 3) ^ 3 ((row & 15) >> 3)
    const int prow = lane >> 2, ppos = lane & 3;
    const int pchunk = (ppos ^ (3 * (prow >> 3))) << 4;
    const i32x4 arsrc = make_srd(A + (long)blockIdx.x * a_block_bytes);
    const i32x4 wrsrc = make_srd(W);
    // A source, row-major (a_blocked = 0): 1-KiB rows, a stage takes 64 bytes of each (HALF a cache line per row: the other half
    // is the next stage's); blocked (1): [64-byte slab][row][64 B] as a producer could store it -- a piece is 1 KiB contiguous
    const int a_rs = a_blocked ? 64 : kRowStride;
    const int av = prow * a_rs + pchunk;            // + piece * 16 rows (scalar) + stage byte offset (scalar)
    const int wv = prow * 64 + pchunk;             // W stage-major: [stage][256 rows][64 B]
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr)(smem);
    // wave w issues A pieces 4w..4w+3 and W pieces 4w..4w+3 of every stage
    const unsigned a_dst = lds0 + wave * 4 * 1024, w_dst = lds0 + kWOff + wave * 4 * 1024;
    const int a_so0 = wave * 4 * 16 * a_rs, w_so0 = wave * 4 * 1024;

    // a stage's source: A byte offset 64 * (stage % 16) within the 1-KiB row (then the next rows: a tap), W stage-major
    auto a_soff = [&](int st) {
        return a_so0 + (a_blocked ? (st & 15) * (272 * 64) : (st & 15) * 64) + ((st >> 4) % 3) * 2 * a_rs;
    };
    // a_share = 1: activation pieces in one stage of three only (one slab held in LDS for a layer's three taps)
    auto a_due = [&](int st) { return !a_share || (st % 3) == 0; };
    auto w_soff = [&](int st) { return w_so0 + (st % 48) * 16 * 1024; };
#define ISSUE_PIECE(st_, slot_, p_)                                                                      \
    {                                                                                                    \
        if ((p_) < 4) { if (a_due(st_)) dma16(arsrc, a_dst + (slot_) * kSlot + (p_) * 1024, av, a_soff(st_) + (p_) * 16 * a_rs); } \
        else dma16(wrsrc, w_dst + (slot_) * kSlot + ((p_) - 4) * 1024, wv, w_soff(st_) + ((p_) - 4) * 1024);   \
    }
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc[i][j]));
        }
    // prologue: stages 0..3 -> slots 0..3 (the loop's first stage requests stage 3 again: harmless)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int p = 0; p < 8; ++p) ISSUE_PIECE(s, s, p)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4 wf[2][8], af[4];
#define RDF(dst_, off_) dst_ = *reinterpret_cast<const f32x4*>(smem + (off_));
#pragma unroll
    for (int j = 0; j < 8; ++j) RDF(wf[0][j], w_rd + j * 1024)
    RDF(af[0], a_rd)
    RDF(af[1], a_rd + 1024)
    if (!RD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[1][j] = wf[0][j];
        af[2] = af[0];
        af[3] = af[1];
    }

    // JS: the MFMA of a row behind which this wave issues its piece (stagger = 1: wave w at 2 w + 1 instead of all four
    // waves at 6 -- after every barrier the waves run in lockstep: four 1-KiB requests reach the CU's address path together)
    constexpr int jslot = JS;
    int st = 0;                                    // stage being computed; requests go out for st + 3
    // STAGE(slot): rows i = 0..7; behind the MFMAs of row i: A fragment of row i + 2 (rows 8, 9 = the next stage's 0, 1),
    // next-stage W fragment i, DMA piece i of stage st + 3 into slot (slot + 3) & 3
#define STAGE(slot_, wcur_, wnxt_)                                                                      \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                              \
                MF(i, j, af[i % 4], wf[wcur_][j])                                                        \
                if (j == 1 && RD) {                                                                      \
                    SB();                                                                                \
                    if (i < 6) { RDF(af[(i + 2) % 4], a_rd + (slot_) * kSlot + (i + 2) * 1024) }         \
                    else { RDF(af[(i + 2) % 4], a_rd + (((slot_) + 1) & 3) * kSlot + (i - 6) * 1024) }   \
                    SB();                                                                                \
                }                                                                                        \
                if (j == 4 && RD) {                                                                      \
                    SB();                                                                                \
                    RDF(wf[wnxt_][i], w_rd + (((slot_) + 1) & 3) * kSlot + i * 1024)                     \
                    SB();                                                                                \
                }                                                                                        \
                if (DMA && j == jslot) {                                                                 \
                    SB();                                                                                \
                    ISSUE_PIECE(st + 3, ((slot_) + 3) & 3, i)                                            \
                    SB();                                                                                \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        SB();                                                                                            \
        if (DMA) {                                                                                       \
            if (!a_share || a_due(st + 3)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");              \
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                      \
        }                                                                                                \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                              \
        __builtin_amdgcn_s_barrier();                                                                    \
        SB();                                                                                            \
        ++st;                                                                                            \
    }
    // (A fragments in a ring of four registers sets: row i in af[i % 4], eight rows per stage, so every stage starts at af[0])
    for (int it = 0; it < n_stage4; ++it) {
        STAGE(0, 0, 1)
        STAGE(1, 1, 0)
        STAGE(2, 0, 1)
        STAGE(3, 1, 0)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" ::"a"(acc[i][j]));
    if (out == reinterpret_cast<float*>(1)) out[0] = af[0].x + wf[0][0].x + wf[1][0].x;
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void onewave_kernel(
    const char* __restrict__ A, const char* __restrict__ W, float* __restrict__ out, int n_stage4, long a_block_bytes,
    int a_blocked, int a_share, int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int sel = stagger ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 4;
    if (sel == 0) onewave_body<MODE, 1>(smem, A, W, out, n_stage4, a_block_bytes, a_blocked, a_share);
    else if (sel == 1) onewave_body<MODE, 3>(smem, A, W, out, n_stage4, a_block_bytes, a_blocked, a_share);
    else if (sel == 2) onewave_body<MODE, 5>(smem, A, W, out, n_stage4, a_block_bytes, a_blocked, a_share);
    else if (sel == 3) onewave_body<MODE, 7>(smem, A, W, out, n_stage4, a_block_bytes, a_blocked, a_share);
    else onewave_body<MODE, 6>(smem, A, W, out, n_stage4, a_block_bytes, a_blocked, a_share);
}

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

template <int MODE>
static double run(const char* A, const char* W, float* out, int n_stage4, long a_block_bytes, int reps, float* ms_out,
                  int a_blocked = 0, int a_share = 0, int stagger = 0) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(onewave_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> t;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        onewave_kernel<MODE><<<256, 256, kLds>>>(A, W, out, n_stage4, a_block_bytes, a_blocked, a_share, stagger);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    *ms_out = t[t.size() / 2];
    const double flop = 256.0 * n_stage4 * 4 * 4 * 64 * (2.0 * 16 * 16 * 32);
    return flop / (*ms_out * 1e-3) * 1e-15;
}

int main(int argc, char** argv) {
    const bool zero = argc > 1 && !strcmp(argv[1], "zero");
    // layer 2 of the bench batch is 54.75 K-tiles of 64 per CU = 109.5 stages: 27 x 4 stages per launch
    const int n_stage4 = argc > 2 ? atoi(argv[2]) : 27;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    const long a_block_bytes = 300L * kRowStride;                       // blocks read overlapping row windows of a 77 MB tensor
    const size_t a_bytes = 255 * a_block_bytes + (256 + 16) * (size_t)kRowStride * 2 + (1 << 20);
    const size_t w_bytes = 48 * 16 * 1024 + (1 << 16);
    std::vector<unsigned short> ha(a_bytes / 2), hw(w_bytes / 2);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(-0.05f, 0.05f);
    if (!zero) {
        for (auto& v : ha) { const float x = nd(rng); v = f2bf(x > 0.f ? x : 0.f); }
        for (auto& v : hw) v = f2bf(ud(rng));
    }
    char *A, *W;
    float* out;
    hipMalloc(&A, a_bytes);
    hipMalloc(&W, w_bytes);
    hipMalloc(&out, 4);
    hipMemcpy(A, ha.data(), a_bytes, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), w_bytes, hipMemcpyHostToDevice);
    float ms;
    // warm the clock state: ~2 s of the full loop
    for (int i = 0; i < 200; ++i) run<3>(A, W, out, n_stage4, a_block_bytes, 50, &ms);
    // the DMA stream's variants: activation source row-major / blocked, one slab per three taps
    for (int stagger = 0; stagger < 2; ++stagger)
        for (int blocked = 0; blocked < 2; ++blocked)
            for (int share = 0; share < 2; ++share) {
                float m2, m3;
                run<2>(A, W, out, n_stage4, a_block_bytes, 200, &m2, blocked, share, stagger);
                const double p3 = run<3>(A, W, out, n_stage4, a_block_bytes, 200, &m3, blocked, share, stagger);
                printf("%s | A %s, %s: DMA only %.4f ms | reads+DMA %.4f ms %.3f PF\n", stagger ? "waves' DMA slots staggered" : "all waves' DMA at MFMA 6 ",
                       blocked ? "blocked (1-KiB contiguous pieces)" : "row-major (64 B of each row)   ",
                       share ? "one slab per 3 taps" : "every stage        ", m2, m3, p3);
            }
    printf("onewave_bound: %s operands, %d x 4 stages of 32 per launch (layer 2 of B = 256: 109.5 stages per CU)\n", zero ? "ZERO" : "relu-normal / uniform", n_stage4);
    for (int rd = 0; rd < rounds; ++rd) {
        const double p0 = run<0>(A, W, out, n_stage4, a_block_bytes, 200, &ms);
        const float m0 = ms;
        const double p1 = run<1>(A, W, out, n_stage4, a_block_bytes, 200, &ms);
        const float m1 = ms;
        const double p2 = run<2>(A, W, out, n_stage4, a_block_bytes, 200, &ms);
        const float m2 = ms;
        const double p3 = run<3>(A, W, out, n_stage4, a_block_bytes, 200, &ms);
        const float m3 = ms;
        printf("round %d  MFMA+barrier only %.4f ms %.3f PF | +fragment reads %.4f ms %.3f PF | +DMA only %.4f ms %.3f PF | reads+DMA %.4f ms %.3f PF\n",
               rd, m0, p0, m1, p1, m2, p2, m3, p3);
        fflush(stdout);
    }
    return 0;
}

// MFCC front end (next-row N3: the step in front of the hot path).
//
// What the reference computes per utterance on the CPU, inside its DataLoader workers
// (reference dataset.py:128): python_speech_features.mfcc(signal, 16000, numcep=24, nfilt=26,
// nfft=512) -- pre-emphasis, 25 ms / 10 ms rectangular frames (zero padded tail), |rfft|^2 / nfft,
// 26 triangular mel filters, log, orthonormal DCT-II, sinusoidal lifter, c0 := log frame energy.
// Here: one wave per pair of frames, the whole chain in one kernel, so a batch of waveforms on
// the GPU becomes the [B, frames, numcep] tensor XVectorModel consumes with one launch and one
// pass over the samples (HBM traffic: the samples once -- frames overlap 2.5x but the re-reads hit
// L2 -- plus numcep floats per frame out).
//
// The filterbank and the DCT x lifter matrices are built on the host in double precision with the
// package's formulas (floor((nfft+1)*hz/samplerate) bin edges etc.) and kept on the device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/xvec_hip.h"
#include "xvec_internal.h"

namespace {

constexpr int kMaxNfft = 4096;
constexpr int kThreads = 256;
constexpr float kEps = 2.220446049250313e-16f;   // numpy.finfo(float).eps, what the package substitutes for 0

struct MfccDev {
    // one table blob, copied into LDS by every block:
    //   twiddles per pass (cos, -sin pairs) | dctl[numcep][nfilt|1] | fb_w[fb_nnz] | fb_lo[nfilt] | fb_off[nfilt+1]
    const float* tables;
    int tw_off, dctl_off, fbw_off, fblo_off, fboff_off, table_floats;
    int frame_len, frame_step, nfft, log2n, nbins, nfilt, numcep, append_energy;
    float preemph;
    // nfft == 512 kernel (fft512 below): per-lane twiddles, filterbank and DCT x lifter as MFMA B fragments
    const float* f_tw1;     // [7][64] complex: W_512^(lane * k), k = 1..7
    const float* f_tw2;     // [7][8] complex: W_64^(c * k)
    const float* f_fb;      // [2 filter tiles][16 bin groups][64 lanes][4]
    const float* f_dct;     // [2 cepstrum tiles][2 filter groups][64 lanes][4]
    int f_n0, f_lo0, f_n1, f_lo1;   // bin groups with a non-zero weight: tile 0 [lo0, lo0+n0), tile 1 [lo1, lo1+n1)
    // the filterbank as a BANDED product on v_mfma_f32_4x4x1_16b_f32 (fft512::kBand*; f_band == nullptr: the dense form)
    const float* f_band;    // [4 waves][6][64 lanes][4]: 20 weights per lane (instruction n = 4 q + c), then 4 LDS byte offsets
    const int* f_gat;       // [8][32]: per filter, LDS byte offsets of the partial sums that hold it (frame quad 0)
    int f_gat_n;            // the longest list of a filter (entries past a filter's list point at zeros)
    // frame index -> (utterance, frame of the utterance) without a division: b = (g * fr_magic) >> (31 + fr_shift), exact for
    // g < 2^31 (fr_magic = ceil(2^(31 + fr_shift) / n_frames), fr_shift = ceil(log2 n_frames); set per launch by xvec_mfcc)
    unsigned fr_magic;
    int fr_shift;
    float in_scale;         // 16-bit PCM input (xvec_mfcc_i16): sample = (float)s * in_scale, one fp32 rounding (set per launch)
};

__device__ __forceinline__ unsigned bitrev(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

// LDS position of complex point i: one pad slot per 32 points.  Power-of-two strides (the
// bit-reversed store, the butterflies of the first stages) otherwise land on a handful of banks:
// rocprofv3 counted 53 % of all LDS cycles as bank conflicts without it.
__host__ __device__ constexpr int zpos(int i) { return i + (i >> 5); }
// floats per wave: padded complex image + two power spectra
__host__ __device__ constexpr int wave_floats(int nfft, int nbins) { return 2 * zpos(nfft) + 2 * ((nbins + 1) & ~1); }

// order this wave's LDS traffic: earlier ds_writes are complete and visible to the wave's later
// ds_reads (one wave owns its LDS region, so no block barrier is ever needed)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

constexpr int kWavesPerBlock = kThreads / 64;

// One WAVE per PAIR of frames (eight frames per 256-thread block).  The block first copies the
// small tables (twiddles, DCT x lifter rows, the non-zero filterbank weights) into LDS.  Each wave
// then packs its two real frames into ONE complex signal (frame A real part, frame B imaginary
// part), runs a single FFT on the image in its private, padded LDS region (8-byte ds_read/ds_write,
// two radix-2 stages per pass, no block barriers) and separates the two
// spectra by conjugate symmetry.  Lanes 0-31 finish frame A, lanes 32-63 frame B: mel filters
// (one lane per filter walks its short run of bins), log, DCT x lifter rows, store.
// L2N: log2(nfft) fixed at compile time (9: the reference's nfft = 512), or 0 = taken from `d`.  With
// constant trip counts the sample loads of a frame, the two butterfly groups of a pass and the
// bins of the split are issued together instead of one LDS / memory round trip at a time.
// Samples: fp32, or (i16 != 0) 16-bit PCM as scipy.io.wavfile.read yields it (reference dataset.py:125), every sample entering as
// (float)s * d.in_scale -- exactly what the float path is given when the caller converts on the host.  A RUN-TIME flag on
// purpose: one instantiation per L2N, so everything behind the sample fetch is the same instructions for both input types
// and the two results are equal bit for bit by construction (as two template instances hipcc contracted the multiply-adds
// of the two differently: 1 ulp apart).
template <int L2N>
__global__ __launch_bounds__(kThreads) void mfcc_kernel(const void* __restrict__ sig_v, int i16, int64_t n_samples,
                                                        int n_frames, MfccDev d, float* __restrict__ out) {
    // (__fmul_rn: a product of its own, rounded once -- as a plain `*` hipcc may fuse it into the pre-emphasis' multiply-add)
    auto smp = [&](int64_t i_) -> float {
        return i16 ? __fmul_rn((float)static_cast<const int16_t*>(sig_v)[i_], d.in_scale) : static_cast<const float*>(sig_v)[i_];
    };
    const int log2n = L2N ? L2N : d.log2n;
    const int nfft = L2N ? (1 << L2N) : d.nfft;
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tab = lds;
    for (int i = threadIdx.x; i < d.table_floats; i += kThreads) tab[i] = d.tables[i];
    __syncthreads();
    const float2* tw2 = reinterpret_cast<const float2*>(tab + d.tw_off);
    const float* dctl = tab + d.dctl_off;
    const float* fb_w = tab + d.fbw_off;
    const int* fb_lo = reinterpret_cast<const int*>(tab + d.fblo_off);
    const int* fb_off = reinterpret_cast<const int*>(tab + d.fboff_off);

    const int nb_pad = (d.nbins + 1) & ~1;
    const int per_wave = wave_floats(nfft, d.nbins);
    float2* z = reinterpret_cast<float2*>(lds + d.table_floats + (size_t)wave * per_wave);
    float* work = reinterpret_cast<float*>(z + zpos(nfft));   // [2][nb_pad] power spectra of frames A, B

    const int fa = 2 * (blockIdx.x * kWavesPerBlock + wave), b = blockIdx.y;
    if (fa >= n_frames) return;                               // whole wave leaves together (no later block barrier)
    const bool has_b = fa + 1 < n_frames;
    const int64_t s0 = (int64_t)b * n_samples;             // the utterance's first sample
    const int64_t start = (int64_t)fa * d.frame_step;
    const int used = d.frame_len < nfft ? d.frame_len : nfft;   // rfft(frame, nfft) truncates long frames

    // ---- the two pre-emphasised frames, zero padded to nfft, as one complex signal in bit-reversed order
#pragma unroll
    for (int n = lane; n < nfft; n += 64) {
        float va = 0.f, vb = 0.f;
        if (n < used) {
            const int64_t ga = start + n, gb = ga + d.frame_step;
            if (ga < n_samples) va = (ga == 0) ? smp(s0) : smp(s0 + ga) - d.preemph * smp(s0 + ga - 1);
            if (has_b && gb < n_samples) vb = smp(s0 + gb) - d.preemph * smp(s0 + gb - 1);
        }
        z[zpos((int)bitrev((unsigned)n, log2n))] = make_float2(va, vb);
    }
    wave_lds_sync();

    // ---- decimation-in-time FFT on the bit-reversed image, two radix-2 stages per pass: the four
    // points {p, p+h, p+2h, p+3h} (h = 4^pass, bits of weight h and 2h clear in p) form a closed
    // two-stage butterfly, so a pass touches every point once instead of twice.  The twiddles of a
    // pass are stored contiguously per j, {W_2h^j, W_4h^j, W_4h^(j+h)}: consecutive lanes read
    // consecutive 24-byte records (strided reads of one N/2-entry table were up to 16-way conflicts).
    const float2* twp = tw2;
    int st = 0;
    for (; st + 1 < log2n; st += 2) {
        const int h = 1 << st;
#pragma unroll
        for (int i = lane; i < nfft / 4; i += 64) {
            const int j = i & (h - 1);
            const int p0 = ((i >> st) << (st + 2)) + j;
            const float2 w1 = twp[3 * j], w2 = twp[3 * j + 1], w3 = twp[3 * j + 2];
            const int q0 = zpos(p0), q1 = zpos(p0 + h), q2 = zpos(p0 + 2 * h), q3 = zpos(p0 + 3 * h);
            const float2 a0 = z[q0], a1 = z[q1], a2 = z[q2], a3 = z[q3];
            // first stage: (a0, a1) and (a2, a3) with W_2h^j
            const float t1r = a1.x * w1.x - a1.y * w1.y, t1i = a1.x * w1.y + a1.y * w1.x;
            const float t3r = a3.x * w1.x - a3.y * w1.y, t3i = a3.x * w1.y + a3.y * w1.x;
            const float b0r = a0.x + t1r, b0i = a0.y + t1i, b1r = a0.x - t1r, b1i = a0.y - t1i;
            const float b2r = a2.x + t3r, b2i = a2.y + t3i, b3r = a2.x - t3r, b3i = a2.y - t3i;
            // second stage: (b0, b2) with W_4h^j and (b1, b3) with W_4h^(j+h)
            const float u2r = b2r * w2.x - b2i * w2.y, u2i = b2r * w2.y + b2i * w2.x;
            const float u3r = b3r * w3.x - b3i * w3.y, u3i = b3r * w3.y + b3i * w3.x;
            z[q0] = make_float2(b0r + u2r, b0i + u2i);
            z[q2] = make_float2(b0r - u2r, b0i - u2i);
            z[q1] = make_float2(b1r + u3r, b1i + u3i);
            z[q3] = make_float2(b1r - u3r, b1i - u3i);
        }
        twp += 3 * h;
        wave_lds_sync();
    }
    if (st < log2n) {                                                // odd log2(nfft): one plain stage, W_N^j
        const int half = nfft >> 1;
#pragma unroll
        for (int i = lane; i < half; i += 64) {
            const float2 w = twp[i];
            const int ql = zpos(i), qh = zpos(i + half);
            const float2 x = z[qh], u = z[ql];
            const float tr = x.x * w.x - x.y * w.y, ti = x.x * w.y + x.y * w.x;
            z[ql] = make_float2(u.x + tr, u.y + ti);
            z[qh] = make_float2(u.x - tr, u.y - ti);
        }
        wave_lds_sync();
    }

    // ---- split the spectra (A = (Z[k] + conj Z[N-k])/2, B = (Z[k] - conj Z[N-k])/(2i)), power, energies
    float ea = 0.f, eb = 0.f;
    const float scale = 0.25f / (float)nfft;                // (1/2)^2 from the split, 1/nfft from powspec
    for (int k = lane; k < d.nbins; k += 64) {
        const float2 p = z[zpos(k)], q = z[zpos((nfft - k) & (nfft - 1))];
        const float ar = p.x + q.x, ai = p.y - q.y, br = p.y + q.y, bi = q.x - p.x;
        const float pa = (ar * ar + ai * ai) * scale, pb = (br * br + bi * bi) * scale;
        work[k] = pa;
        work[nb_pad + k] = pb;
        ea += pa;
        eb += pb;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ea += __shfl_xor(ea, o);
        eb += __shfl_xor(eb, o);
    }
    const int sel = lane >> 5;                                // 0: frame A, 1: frame B
    float energy = sel ? eb : ea;
    if (energy == 0.f) energy = kEps;
    wave_lds_sync();

    // ---- mel filterbank and log (log energies go to the now dead complex image)
    float* le = reinterpret_cast<float*>(z) + sel * d.nfilt;
    const float* pw = work + sel * nb_pad;
    for (int j = lane & 31; j < d.nfilt; j += 32) {
        const int lo = fb_lo[j], o = fb_off[j], len = fb_off[j + 1] - o;
        float acc = 0.f;
        for (int k = 0; k < len; ++k) acc = fmaf(pw[lo + k], fb_w[o + k], acc);
        if (acc == 0.f) acc = kEps;
        le[j] = logf(acc);
    }
    wave_lds_sync();

    // ---- DCT-II (orthonormal) x lifter; c0 replaced by the log energy
    const int f = fa + sel;
    if (f < n_frames) {
        for (int c = lane & 31; c < d.numcep; c += 32) {
            const float* row = dctl + c * (d.nfilt | 1);       // odd row stride: lanes hit distinct banks
            float v = 0.f;
            for (int j = 0; j < d.nfilt; ++j) v = fmaf(le[j], row[j], v);
            if (c == 0 && d.append_energy) v = logf(energy);
            out[((int64_t)b * n_frames + f) * d.numcep + c] = v;
        }
    }
}


// ---- nfft == 512 (the reference's call, dataset.py:128) ------------------------------------------------------------
// The kernel above walks the FFT through LDS one radix-4 pass at a time (five read-modify-write passes with a
// wave-wide wait each) and gives the mel filters one LANE each (filter 25 walks 66 bins one LDS round trip at a
// time): 118 us per 256 x 3 s, 26 k cycles per pair of frames, latency from end to end.  Here:
//  * the 512-point FFT of a pair of frames is three radix-8 steps IN REGISTERS (8 complex points per lane:
//    n = 64a + 8b + c, k = k0 + 8 k1 + 64 k2; DFT-8 over a, x W_512^((8b+c) k0), DFT-8 over b, x W_64^(c k1),
//    DFT-8 over c) with two transposes through the wave's private LDS region between them (index maps chosen
//    so that every ds_write_b64 / ds_read_b64 is conflict-free by the bank rules: rows of 8 padded to 9,
//    planes of 64 padded to 72); the twiddles never change: step 1's seven per lane live in registers, step 2's 56 in LDS;
//  * a block of four waves owns 16 consecutive frames of the batch (frames are numbered through the whole batch:
//    76 544 = 4784 x 16 for 256 x 299, no ragged last tile per utterance), two pairs per wave, and leaves their
//    power spectra in LDS as sixteen rows of 256 (eight in an array, eight in the waves' exchange regions: kExRow);
//  * the mel filterbank is a BANDED product on v_mfma_f32_4x4x1_16b_f32 (kBand* below; round 6) and, for a filterbank that
//    form cannot hold, the dense one described next; the DCT x lifter is dense either way.
//  * dense: mel filterbank and DCT x lifter are matrix products on v_mfma_f32_16x16x4_f32 (exact fp32): P[16 x 256] x
//    FB^T[256 x 32] with the filterbank's zero blocks skipped (filters 0-15 end at bin 87, filters 16-25 begin
//    at bin 77: 18 of 32 blocks of 16 bins), split over the four waves and summed in wave order; log;
//    [16 x 32] x DCTL^T[32 x 32].  The B fragments are packed per lane on the host and fetched per tile (L2-resident):
//    blocks are persistent (FIVE per CU since round 6: 96 registers, 30 016 bytes of LDS -- the SIMDs were half busy with
//    four waves each, 51.5 / 42.3 / 38.3 us at two / three / four blocks) and walk the tiles with a grid stride.
namespace fft512 {

#ifdef XVEC_MFKNOCK
// Timing-only knock-outs of mfcc512_kernel (-DXVEC_MFKNOCK=mask; results are garbage):
//   (bit 0, round 6: the split without its LDS round trip -- the bound of what ships now as the ds_bpermute split)
//   bit 1: the DCT fragments are not fetched per tile      bit 2: no tile tail at all (filterbank, log, DCT, two barriers)
#define MF_KNOCK_DCTLD ((XVEC_MFKNOCK & 2) != 0)
#define MF_KNOCK_TAIL ((XVEC_MFKNOCK & 4) != 0)
#else
#define MF_KNOCK_DCTLD false
#define MF_KNOCK_TAIL false
#endif

constexpr int kTile = 16;                   // frames per block pass
constexpr int kEx = 576;                    // complex slots of a wave's exchange region (8 x 72)
constexpr int kPS = 264;                    // floats per row of the power-spectrum tile (256 + 8) and of the log-mel tile: row strides of 8 mod 64
constexpr int kLS = 40;                     // dwords put the sixteen lanes of a ds_read_b128 service group (rows r, lane quads q: 16-byte
                                            // fragments at 4 q) on sixteen distinct 4-bank windows (260 / 36: one 2-way conflict per group)
constexpr int kMaxItems = 5;                // (filter tile, bin group) products per wave: 20 per block (the default filterbank has 18)
// The power rows of a wave's SECOND pair of frames live in its own exchange region (idle from that pair's last exchange to the
// next tile's first), its partial sums behind them; only the first pairs' eight rows have storage of their own: 30 016 bytes a
// block, five blocks on a CU (round 6; sixteen rows of their own were 38 464 bytes, four blocks).
constexpr int kExRow = 16;                  // floats: the rows in wave w's region start at 16 w + 4: with the row stride (8 mod 64) the
constexpr int kExRow0 = 4;                  // eight rows of the array sit on bank offsets 0, 8, ..., 56 and these eight on 4, 12, ..., 60
constexpr int kExPart = 592;                // floats: the partial sums (2 x 64 lanes x 4) behind the rows (3 x 16 + 4 + 2 x 264 = 580)
static_assert(3 * kExRow + kExRow0 + 2 * kPS <= kExPart && kExPart + 512 <= 2 * kEx && kExPart % 4 == 0 && kExRow0 % 4 == 0,
              "layout of an exchange region");
constexpr int kLdsFloats = 4 * kEx * 2 + (kTile / 2) * kPS + kTile * kLS + 2 * kTile + 2 * 56;
// ---- the banded filterbank (round 6).  A bin carries weight for two neighbouring triangles, the dense form multiplies it
// with sixteen: 18 products of 16 x 16 x 16 per tile, 32 cycles of the matrix pipe per 16 x 16 x 4 step, 640 cycles per wave
// and tile -- a fifth of a SIMD's busy time.  v_mfma_f32_4x4x1_16b_f32 is SIXTEEN independent 4 x 4 x 1 products (8 cycles):
// block 4 fq + s takes frames 4 fq .. + 3 (rows) of ONE bin and the bin's weights for FOUR neighbouring filters (columns).
// The host cuts the bins into at most 15 GROUPS of at most 20 consecutive bins whose filters fit one window of four (a .. a + 3);
// wave w owns groups 4 w .. 4 w + 3, SLOT s of its instructions is group 4 w + s, instruction n the group's bin n: 20
// instructions of 8 cycles per wave and tile, every lane ends with a group's [4 frames] x one filter, nothing to add across
// lanes.  The summing threads gather per filter the (at most eight) groups that hold it, in bin order; an absent entry points
// at group 15, whose weights are all zero.
constexpr int kBandN = 20;                  // instructions per tile and wave = bins a group's lanes read
constexpr int kBandCap = 19;                // bins of a group that may carry weight: one bin of play, so that the four groups of a wave
                                            // can start their reads on four different residues mod 4 -- with the sixteen rows on bank
                                            // offsets 0, 4, ..., 60 the 64 lanes of a read then fall on 64 different banks
constexpr int kBandGroups = 15;             // (+ the all-zero group 15)
constexpr int kBandGat = 8;                 // groups a filter can collect from
constexpr int kBandLdsFloats = kLdsFloats + kBandGat * 32;   // + the gather table
static_assert(kBandLdsFloats * 4 <= 32 * 1024, "five blocks per CU");
typedef float f32x4v __attribute__((ext_vector_type(4)));

// A complex value is a register pair and the arithmetic below is PACKED fp32 (v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32: both halves per instruction; measured here 52.4 -> 49.7 us per batch against the scalar form -- the
// kernel is bound by issue slots and LDS cycles together, not by the VALU alone).  A complex add is one instruction;
// the op_sel / neg modifiers of the packed forms fold the multiplications by -i and the conjugates of the
// butterflies into the adds, and make a complex product two instructions with no swapped copy of the twiddle.
typedef float c32 __attribute__((ext_vector_type(2)));
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ c32 cadd_mi(c32 a, c32 b) {
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a - (-i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ c32 csub_mi(c32 a, c32 b) {
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ c32 cadd_conj(c32 a, c32 b) {
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ c32 csub_conj(c32 a, c32 b) {
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x w = (x.x w.x - x.y w.y, x.x w.y + x.y w.x)
__device__ __forceinline__ c32 cmul(c32 x, c32 w) {
    c32 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(x), "v"(w), "v"(t));
    return r;
}

// sum over the 64 lanes, in every lane, without LDS: within each row of 16 lanes by DPP (mirror, half mirror, the two quad
// permutations), across the four rows by v_permlane16_swap / v_permlane32_swap of the value with itself
__device__ __forceinline__ float wave_sum(float v) {
#define MF_DPP_ADD(ctrl_) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl_, 0xf, 0xf, false));
    MF_DPP_ADD(0x140)          // row_mirror:        lane i <-> 15 - i
    MF_DPP_ADD(0x141)          // row_half_mirror:   i <-> 7 - i within each half row
    MF_DPP_ADD(0xb1)           // quad_perm [1,0,3,2]
    MF_DPP_ADD(0x4e)           // quad_perm [2,3,0,1]
#undef MF_DPP_ADD
    // the two swaps as inline asm: given the SAME value twice and a use of BOTH results, hipcc 7.2 folds the sum of the two
    // results of __builtin_amdgcn_permlane16_swap / permlane32_swap into 2 x the first one in this kernel (seen as log-energies
    // off by a few per cent; with two different operands the builtins are fine).  s_nop 1: the two wait states between a vector
    // write of an operand and the swap, which the compiler adds for the builtin.
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));          // rows {0|0|2|2}, {1|1|3|3}
    a += b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));          // halves {lo|lo}, {hi|hi}
    return a + b;
}

// natural-order DFT-8, forward sign (W_8 = exp(-2 pi i / 8)): 28 packed instructions
__device__ __forceinline__ void dft8(c32 (&x)[8]) {
    constexpr float c = 0.70710678118654752f;
    // x_j +- x_(j+4)
    const c32 a0 = x[0] + x[4], b0 = x[0] - x[4];
    const c32 a1 = x[1] + x[5], t1 = x[1] - x[5];
    const c32 a2 = x[2] + x[6], t2 = x[2] - x[6];
    const c32 a3 = x[3] + x[7], t3 = x[3] - x[7];
    // odd branch: b_j = (x_j - x_(j+4)) W_8^j;  W_8 = (1 - i)/sqrt2, W_8^2 = -i (folded into the adds below),
    // W_8^3 = -(1 + i)/sqrt2
    const c32 b1 = cadd_mi(t1, t1) * c;
    const c32 b3 = csub_mi(t3, t3) * (-c);
    {   // DFT-4 of a -> X[0], X[2], X[4], X[6]
        const c32 s0 = a0 + a2, d0 = a0 - a2, s1 = a1 + a3, d1 = a1 - a3;
        x[0] = s0 + s1;
        x[4] = s0 - s1;
        x[2] = cadd_mi(d0, d1);
        x[6] = csub_mi(d0, d1);
    }
    {   // DFT-4 of b -> X[1], X[3], X[5], X[7]
        const c32 s0 = cadd_mi(b0, t2), d0 = csub_mi(b0, t2), s1 = b1 + b3, d1 = b1 - b3;
        x[1] = s0 + s1;
        x[5] = s0 - s1;
        x[3] = cadd_mi(d0, d1);
        x[7] = csub_mi(d0, d1);
    }
}

// w[(k - 1) * stride] = the twiddle of output k (registers: stride 1; the LDS table of step 2: stride 8)
template <int STRIDE>
__device__ __forceinline__ void twiddle(c32 (&x)[8], const c32* w) {
#pragma unroll
    for (int k = 1; k < 8; ++k) x[k] = cmul(x[k], w[(k - 1) * STRIDE]);
}

// Descriptors of one frame's samples.  The frame index is uniform: the base is the frame's first sample (one before
// it when there is one, for the pre-emphasis) and the range ends with the frame or the signal -- reads outside
// return zeros, so the sixteen loads of a frame are issued back to back without a branch or a wait between them
// (a first version loaded under `if (n < lim)`: every value was waited for where it was computed, 28 memory round
// trips per tile).  `prev` ends one sample earlier than `cur`, so x[n-1] of the first sample past the frame reads 0 too.
struct FrameSrc {
    __amdgpu_buffer_rsrc_t cur, prev;
    int back;
};
// (utterance b, frame f of the utterance) of global frame g.  Round 5: ONE magic division per wave and tile -- the wave's four
// frames of a tile are consecutive, the other three are stepped from the first (frame_step_bf) -- and everything below in 32-bit
// scalar arithmetic where the launch allows it: in-kernel stamps put 17 % of a wave's time at the head of a pair, in the scalar
// work of two descriptors built from scratch (154 scalar instructions and 42 wait states per pair).
struct FramePos {
    unsigned b, f;
};
__device__ __forceinline__ FramePos frame_pos(int64_t gframe, int64_t total_frames, int n_frames, const MfccDev& d) {
    const unsigned gf = gframe < total_frames ? (unsigned)gframe : 0u;     // the launcher keeps total_frames below 2^31
    // (a 32-bit division by a run-time divisor is ~30 scalar instructions; by the launch's magic number a multiply-high and a shift)
    FramePos p;
    p.b = (unsigned)(((unsigned long long)gf * d.fr_magic) >> (31 + d.fr_shift));
    p.f = gf - p.b * (unsigned)n_frames;
    return p;
}
__device__ __forceinline__ FramePos frame_step_bf(FramePos p, int n_frames) {
    p.f += 1;
    if (p.f >= (unsigned)n_frames) {
        p.f = 0;
        p.b += 1;
    }
    return p;
}
// (n_samples < 2^30 on this path -- xvec_mfcc sends longer signals to the generic kernel -- so a frame's position inside its
//  signal is 32-bit scalar arithmetic; only the utterance's base is a 64-bit product)
template <int ES>      // bytes per sample: 4 (fp32) or 2 (16-bit PCM)
__device__ __forceinline__ FrameSrc frame_src(const void* __restrict__ sig, int n_samples, int64_t total_frames,
                                              int64_t gframe, FramePos fp, const MfccDev& d, int used) {
    const bool live = gframe < total_frames;
    const int start = (int)fp.f * d.frame_step;
    const int left = n_samples - start;                        // samples from `start` to the end of the signal
    const int lim = live && left > 0 ? (left < used ? left : used) : 0;   // (a frame step longer than the frame can start past the end)
    FrameSrc f;
    f.back = fp.f > 0 ? 1 : 0;                                 // x[start - 1] exists
    const unsigned long long base = reinterpret_cast<unsigned long long>(sig) + ((int64_t)fp.b * n_samples + (start - f.back)) * ES;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base);
    const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)bhi << 32) | blo);
    const int n_cur = (lim + f.back) * ES, n_prev = n_cur >= ES ? n_cur - ES : 0;
    f.cur = __builtin_amdgcn_make_buffer_rsrc(q, (short)0, __builtin_amdgcn_readfirstlane(n_cur), 0x00020000);
    f.prev = __builtin_amdgcn_make_buffer_rsrc(q, (short)0, __builtin_amdgcn_readfirstlane(n_prev), 0x00020000);
    return f;
}
__device__ __forceinline__ float ldf(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

template <bool I16, bool BAND>   // I16: 16-bit PCM samples converted on the way in ((float)s * d.in_scale); BAND: the banded filterbank
__global__ __launch_bounds__(256, 5) void mfcc512_kernel(const void* __restrict__ sig, int64_t n_samples, int n_frames,
                                                         int64_t total_frames, int n_tiles, MfccDev d,
                                                         float* __restrict__ out) {
    constexpr int ES = I16 ? 2 : 4;
    __shared__ __attribute__((aligned(16))) float smem[BAND ? kBandLdsFloats : kLdsFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c32* ex = reinterpret_cast<c32*>(smem) + wave * kEx;            // this wave's exchange region
    float* P0 = smem + 4 * kEx * 2;                                 // [8][kPS] power spectra of the first pairs: row 2 wave + i
    float* LE = P0 + (kTile / 2) * kPS;                             // [16][kLS] log mel energies
    float* en = LE + kTile * kLS;                                   // [2][16] frame energies (by tile parity)
    const int hi = lane >> 3, lo = lane & 7, q = lane >> 4, row = lane & 15;
    // the split's partner lane (holds Z[512 - k] of this lane's bins k = hi + 8 lo + 64 k2, k2 < 4) as a ds_bpermute byte address
    const int part_addr = 4 * (8 * ((8 - hi) & 7) + (hi ? 7 - lo : (8 - lo) & 7));
    const int bin0 = hi + 8 * lo;

    // twiddles: step 1's depend on the lane and stay in registers, step 2's (W_64^(c k), 56 values) sit in LDS
    c32 w1[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) w1[k] = reinterpret_cast<const c32*>(d.f_tw1)[k * 64 + lane];
    c32* w2 = reinterpret_cast<c32*>(en + 2 * kTile);
    if constexpr (!BAND) {                                          // (the banded form keeps its lane's seven in registers: w2r)
        if (tid < 56) w2[tid] = reinterpret_cast<const c32*>(d.f_tw2)[tid];
    }
    int* gat = reinterpret_cast<int*>(smem + kLdsFloats);           // BAND: [kBandGat][32] byte offsets
    if constexpr (BAND) gat[tid] = d.f_gat[tid];                    // kBandGat * 32 = 256 = the block's threads
    const __amdgpu_buffer_rsrc_t band_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BAND ? d.f_band : d.f_fb), (short)0,
                                                                             (4 * 5 + 1) * 64 * 16, 0x00020000);
    // BAND: LDS byte address of this lane's row (frame 4 fq + t) at its group's first bin
    int band_a = 0;
    if constexpr (BAND) band_a = __builtin_amdgcn_raw_buffer_load_b32(band_rs, lane * 4 + wave * 256, 4 * 5 * 1024, 0);
    __syncthreads();
    // resident B fragments: products wave, wave + 4, ... of the list {tile 0 groups, tile 1 groups}
    const int n_items = d.f_n0 + d.f_n1;
    int a_off[kMaxItems];                                           // float offset of the A fragment in a P row
    int b_off[kMaxItems];                                           // its B fragment in d.f_fb (bytes, uniform; lane l reads 16 bytes at 16 l)
    const __amdgpu_buffer_rsrc_t fb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.f_fb), (short)0, 32 * 64 * 16, 0x00020000);
#pragma unroll
    for (int s = 0; s < kMaxItems; ++s) {
        const int it = wave + 4 * s;
        a_off[s] = 0;
        int bo = 0;                                                 // past the list: any product, it is left out of the sums
        if (it < n_items) {
            const int t = it < d.f_n0 ? 0 : 1;
            const int g = t ? d.f_lo1 + it - d.f_n0 : d.f_lo0 + it;
            bo = (t * 16 + g) * 64;
            a_off[s] = 16 * g;
        }
        b_off[s] = __builtin_amdgcn_readfirstlane(bo * 16);
    }
    // tile row r = lane & 15 (frame r of the tile) as the matrix products read it: wave r >> 2, pair (r >> 1) & 1, frame r & 1
    const float* prow = ((row >> 1) & 1) ? smem + (row >> 2) * (2 * kEx + kExRow) + kExRow0 + (row & 1) * kPS
                                         : P0 + (2 * (row >> 2) + (row & 1)) * kPS;
    const int used = d.frame_len < 512 ? d.frame_len : 512;
    // The power spectrum's factor 2^-11 ((1/2)^2 from the split, 1/nfft from powspec) costs no multiply per bin: 2^-6 rides in
    // the second step's twiddles (the host's table; k1 = 0 has none: one packed multiply), which makes the squares 2^-12, and the
    // 2 that is missing sits in the filterbank weights (host) and on the two frame energies -- powers of two all: exact
    // rescalings, the results are the straight form's to the last bit or two (seven of nine test shapes bit for bit; the
    // compiler's choice of fused multiply-adds differs).  -8 vector multiplies per pair, 39.19 -> 38.73 us interleaved.
    constexpr float kTwoM6 = 0.015625f;

    // raw samples of a pair of frames (x[n] and x[n-1] of both).  Requesting them one pair ahead (during the split of
    // the previous pair / the previous tile's matrix products) measured the same, interleaved on one box, and would cost
    // 32 registers that a fifth wave per SIMD uses better (VALU 50 %, LDS array 54 % busy at four waves).
    // (every load writes a SCALAR of its own: an asm output that is one half of a register pair goes through a temporary and a
    //  copy, which hipcc places right behind the load -- before the data has landed)
    float cur_a[8], cur_b[8], prev_a[8], prev_b[8];                 // x[n] and x[n-1] of frame A (real part) and frame B (imaginary part)
    // Sample n = 64 a + lane of a frame sits at byte (n + back) * ES of its descriptor (x[n-1] one sample lower): ONE lane offset
    // per frame (ES * (lane + back)) and the instruction's immediate offset do that.  The loads are asm so that the immediates
    // stay immediates (given the offsets as expressions hipcc built 32 offset registers per pair with 48 vector instructions --
    // a fifth of the pair's vector work); hipcc does not see asm loads, so the wait for them is written out behind the request.
    // (the lane offset of x[n-1] is -ES in lane 0 at a signal's start: added to the immediates 64 a ES it addresses x[64a - 1]
    //  for a >= 1 and, as an unsigned offset, lands far outside the descriptor's range for a = 0, which reads as 0 -- the
    //  hardware sums lane offset and immediate BEFORE the range check; tests/test_mfcc.py, test_first_frame_previous_samples,
    //  pins exactly these samples)
    const int lES = lane * ES;
    const int n32 = (int)n_samples;
#define MF_LD(dst_, rs_, vo_, imm_)                                                                                              \
    if constexpr (I16) asm volatile("buffer_load_sshort %0, %1, %2, 0 offen offset:%3" : "=v"(dst_) : "v"(vo_), "s"(rs_), "i"((imm_) / 2) : "memory"); \
    else asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%3" : "=v"(dst_) : "v"(vo_), "s"(rs_), "i"(imm_) : "memory");
#define MF_LOAD_FRAME(S_, C_, P_)                                                                              \
    {                                                                                                          \
        const int vc_ = lES + ES * S_.back, vp_ = vc_ - ES;                                                    \
        asm volatile("s_nop 4" ::: "memory");   /* descriptor words may come out of v_readfirstlane: 5 wait states before a load reads them (hipcc pads nothing in front of asm) */ \
        MF_LD(C_[0], S_.cur, vc_, 0) MF_LD(C_[1], S_.cur, vc_, 256) MF_LD(C_[2], S_.cur, vc_, 512) MF_LD(C_[3], S_.cur, vc_, 768) \
        MF_LD(C_[4], S_.cur, vc_, 1024) MF_LD(C_[5], S_.cur, vc_, 1280) MF_LD(C_[6], S_.cur, vc_, 1536) MF_LD(C_[7], S_.cur, vc_, 1792) \
        MF_LD(P_[0], S_.prev, vp_, 0) MF_LD(P_[1], S_.prev, vp_, 256) MF_LD(P_[2], S_.prev, vp_, 512) MF_LD(P_[3], S_.prev, vp_, 768) \
        MF_LD(P_[4], S_.prev, vp_, 1024) MF_LD(P_[5], S_.prev, vp_, 1280) MF_LD(P_[6], S_.prev, vp_, 1536) MF_LD(P_[7], S_.prev, vp_, 1792) \
    }
    // The wait for the 32 loads carries their destination registers as operands: hipcc sees neither the asm loads nor a plain asm
    // wait, so only a DATA dependence keeps every use of the values behind it (two statements: 30 operands at
    // most).  tests/test_kernel_resources.py checks on the assembly that nothing touches a destination before the wait.
#define MF_W8(A_) "+v"(A_[0]), "+v"(A_[1]), "+v"(A_[2]), "+v"(A_[3]), "+v"(A_[4]), "+v"(A_[5]), "+v"(A_[6]), "+v"(A_[7])
#define MF_REQUEST(fa_)                                                                                        \
    {                                                                                                          \
        const FrameSrc sa = frame_src<ES>(sig, n32, total_frames, (fa_), fpos, d, used);                 \
        fpos = frame_step_bf(fpos, n_frames);                                                                  \
        const FrameSrc sb = frame_src<ES>(sig, n32, total_frames, (fa_) + 1, fpos, d, used);             \
        fpos = frame_step_bf(fpos, n_frames);                                                                  \
        MF_LOAD_FRAME(sa, cur_a, prev_a)                                                                       \
        MF_LOAD_FRAME(sb, cur_b, prev_b)                                                                       \
        asm volatile("s_waitcnt vmcnt(0)" : MF_W8(cur_a), MF_W8(prev_a) :: "memory");                        \
        asm volatile("" : MF_W8(cur_b), MF_W8(prev_b) :: "memory");                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if constexpr (I16) {   /* sign-extended 16-bit integers -> (float)s * in_scale, as the host would have converted them */ \
            _Pragma("unroll") for (int a_ = 0; a_ < 8; ++a_) {                                                 \
                cur_a[a_] = __fmul_rn((float)__builtin_bit_cast(int, cur_a[a_]), d.in_scale);                            \
                cur_b[a_] = __fmul_rn((float)__builtin_bit_cast(int, cur_b[a_]), d.in_scale);                            \
                prev_a[a_] = __fmul_rn((float)__builtin_bit_cast(int, prev_a[a_]), d.in_scale);                          \
                prev_b[a_] = __fmul_rn((float)__builtin_bit_cast(int, prev_b[a_]), d.in_scale);                          \
            }                                                                                                  \
        }                                                                                                      \
    }
    // BAND leaves registers free (80 of the 96 that five waves per SIMD allow): the second step's twiddles of this lane stay in
    // registers -- seven 8-byte LDS reads per pair less, 38.33 -> 37.59 us: at five waves per SIMD the LDS unit is what the waves
    // queue on (a sixth wave made it 44 us, profiles/experiments/README.md).  (The DCT's fragments resident instead: -0.3 us.)
    c32 w2r[7] = {};
    if constexpr (BAND) {
#pragma unroll
        for (int k = 0; k < 7; ++k) w2r[k] = reinterpret_cast<const c32*>(d.f_tw2)[k * 8 + lo];
    }
    int par = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1) {
        // ---- FFT of this wave's two pairs of frames, power spectra into P
        FramePos fpos = frame_pos((int64_t)tile * kTile + 4 * wave, total_frames, n_frames, d);     // the wave's first frame of the tile
#pragma unroll 1
        for (int pp = 0; pp < 2; ++pp) {
            const int r0 = 4 * wave + 2 * pp;                       // rows of the pair in the tile
            const int64_t fa = (int64_t)tile * kTile + r0;
            // frame A is the real part, frame B the imaginary part: x[a] = sample 64a + lane, pre-emphasised
            MF_REQUEST(fa)
            c32 x[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) x[a] = c32{cur_a[a], cur_b[a]} - c32{prev_a[a], prev_b[a]} * d.preemph;
            // An all-zero frame (digital silence, or a frame that starts past the end of the signal) must come out as
            // exact zeros -- the package substitutes eps for 0 before the log, -36.04 -- but packed with a live frame it
            // picks up that frame's rounding noise through the split (1e-14 of its power: log ~ -30).  Its power
            // scale is therefore 0 (uniform: one OR over the lane's samples and a ballot per frame).
            unsigned ora = 0u, orb = 0u;
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                ora |= __float_as_uint(x[a].x);
                orb |= __float_as_uint(x[a].y);
            }
            const bool live_a = __ballot((ora & 0x7fffffffu) != 0u) != 0ull, live_b = __ballot((orb & 0x7fffffffu) != 0u) != 0ull;
            const bool mixed = live_a != live_b;                    // (uniform, rare) an all-zero frame packed with a live one
            // lane = 8b + c holds x[64a + 8b + c], a = 0..7
            dft8(x);                                                // over a -> k0
            twiddle<1>(x, w1);                                      // W_512^((8b + c) k0)
#pragma unroll
            for (int k0 = 0; k0 < 8; ++k0) ex[k0 * 72 + lane] = x[k0];
            wave_lds_sync();
#pragma unroll
            for (int b = 0; b < 8; ++b) x[b] = ex[hi * 72 + b * 8 + lo];   // lane = 8 k0 + c
            dft8(x);                                                // over b -> k1
            if constexpr (BAND) twiddle<1>(x, w2r);                 // 2^-6 W_64^(c k1), resident
            else twiddle<8>(x, w2 + lo);
            x[0] = x[0] * kTwoM6;
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) ex[hi * 72 + k1 * 9 + lo] = x[k1];
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = ex[hi * 72 + lo * 9 + c];   // lane = 8 k0 + k1
            dft8(x);                                                // over c -> k2
            // lane 8 k0 + k1 now holds Z[k0 + 8 k1 + 64 k2] in x[k2].  The split needs Z[k] and Z[512 - k] for k < 256 (k2 < 4):
            // 512 - k = (8 - k0) + 8 (7 - k1) + 64 (7 - k2) for k0 != 0, 8 (8 - k1) + 64 (7 - k2) for k0 = 0 != k1 -- ONE
            // partner lane (part_addr) holds all four, in x[7], x[6], x[5], x[4] -- and 64 (8 - k2) in lane 0 itself (its own
            // x[0], x[7], x[6], x[5]).  Eight ds_bpermute_b32 fetch them (round 6; until then a third transpose through LDS:
            // eight 8-byte writes, a wait and ten reads per pair, 16.7 % of a wave's time with the power rows).
            // (the elements go through float temporaries: __builtin_bit_cast(int, v.y) on an ELEMENT of an ext_vector reads
            //  element 0 with hipcc 7.2 -- the imaginary parts came back as copies of the real parts)
            c32 f[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float sx = x[7 - m].x, sy = x[7 - m].y;
                f[m].x = __int_as_float(__builtin_amdgcn_ds_bpermute(part_addr, __float_as_int(sx)));
                f[m].y = __int_as_float(__builtin_amdgcn_ds_bpermute(part_addr, __float_as_int(sy)));
            }
            // (second pair: into this wave's exchange region -- its last read of the region is in program order before these writes)
            float* pw = pp ? smem + wave * (2 * kEx + kExRow) + kExRow0 : P0 + 2 * wave * kPS;
            float ea = 0.f, eb = 0.f;
            // (one uniform branch per pair, not one per bin: the mixed form is the rare one)
            auto power_rows = [&](auto mixed_c) {
                constexpr bool MIXED = decltype(mixed_c)::value;
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    c32 z;           // Z[512 - k]: lane 0 pairs 64 k2 with 64 (8 - k2)
                    if (k2 == 0) { z.x = lane == 0 ? x[0].x : f[0].x; z.y = lane == 0 ? x[0].y : f[0].y; }
                    else { z.x = lane == 0 ? f[k2 - 1].x : f[k2].x; z.y = lane == 0 ? f[k2 - 1].y : f[k2].y; }
                    const c32 sa2 = cadd_conj(x[k2], z), sb2 = csub_conj(x[k2], z);   // 2A, 2iB
                    const c32 qa = sa2 * sa2, qb = sb2 * sb2;
                    float pa = qa.x + qa.y, pb = qb.x + qb.y;       // half the power (see kTwoM6)
                    if constexpr (MIXED) {
                        pa = live_a ? pa : 0.f;
                        pb = live_b ? pb : 0.f;
                    }
                    pw[bin0 + 64 * k2] = pa;
                    pw[kPS + bin0 + 64 * k2] = pb;
                    ea += pa;
                    eb += pb;
                }
                if (lane == 0) {     // bin 256 = Z[256] = lane 0's x[4], its own partner: A[256] = Re, B[256] = Im (energies only)
                    const float ta = 4.f * x[4].x * x[4].x, tb = 4.f * x[4].y * x[4].y;
                    ea += MIXED && !live_a ? 0.f : ta;
                    eb += MIXED && !live_b ? 0.f : tb;
                }
            };
            if (mixed) power_rows(std::true_type{});
            else power_rows(std::false_type{});
            ea = wave_sum(ea);                                     // (vector-ALU cross-lane adds: __shfl_xor is six LDS round trips per value)
            eb = wave_sum(eb);
            if (lane == 0) {
                en[par * kTile + r0] = ea == 0.f ? kEps : 2.f * ea;
                en[par * kTile + r0 + 1] = eb == 0.f ? kEps : 2.f * eb;
            }
        }
        // the filterbank's B fragments: requested here, used behind the barrier (resident they were twenty registers of the 96
        // that five waves per SIMD leave a lane)
        f32x4v fb[kMaxItems];
#pragma unroll
        for (int s = 0; s < kMaxItems; ++s)
            fb[s] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(BAND ? band_rs : fb_rs, lane * 16,
                                                                                    BAND ? (wave * 5 + s) * 1024 : b_off[s], 0));
        __syncthreads();                                            // power rows and en complete
        if constexpr (MF_KNOCK_TAIL) continue;
        // ---- mel filterbank: this wave's share of the (filter tile, bin group) products
        // Independent accumulators, the k-steps outermost: an MFMA's accumulator is then two or three MFMAs old when the next
        // one needs it (v_mfma_f32_16x16x4_f32: 32 cycles to issue, 40 until a dependent one may start).  Round 4 chained all
        // twenty MFMAs of a wave through two accumulators: twenty waits of one MFMA's latency per tile with nothing between
        // them.  Two rounds (products 0-2, then 3-4): all five at once needs 40 registers and spills.  (Products past n_items
        // repeat product 0 and are left out of the sums.)
        if constexpr (BAND) {
            // ---- banded filterbank: 20 x v_mfma_f32_4x4x1_16b_f32 into four accumulators in turn (an accumulator is four
            // instructions old when its next one needs it), added at the end in a fixed order
            static_assert(kMaxItems * 4 == kBandN, "20 weights per lane in five fragments");
            f32x4v acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = f32x4v{0.f, 0.f, 0.f, 0.f};
            const char* lds = reinterpret_cast<const char*>(smem) + band_a;
#pragma unroll
            for (int n = 0; n < kBandN; ++n) {
                const float av = *reinterpret_cast<const float*>(lds + 4 * n);
                acc[n & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, fb[n >> 2][n & 3], acc[n & 3], 0, 0, 0);
            }
            const f32x4v sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            int lo4 = lane;                                         // (opaque: the address below is made here, not held across the loop)
            asm volatile("" : "+v"(lo4));
            reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(ex) + kExPart)[lo4] = sum;   // [frame quad][group][filter] x 4 frames
        } else {
        f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#define MF_FB_ROUND(S0_, N_)                                                                                       \
        {                                                                                                          \
            f32x4v pa[N_], pacc[N_];                                                                               \
            _Pragma("unroll") for (int s = 0; s < N_; ++s) {                                                       \
                pa[s] = *reinterpret_cast<const f32x4v*>(prow + a_off[S0_ + s] + 4 * q);                           \
                pacc[s] = f32x4v{0.f, 0.f, 0.f, 0.f};                                                              \
            }                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
                _Pragma("unroll") for (int s = 0; s < N_; ++s)                                                     \
                    pacc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s][j], fb[S0_ + s][j], pacc[s], 0, 0, 0);     \
            _Pragma("unroll") for (int s = 0; s < N_; ++s) {        /* in product order */                         \
                const int it = wave + 4 * (S0_ + s);                                                               \
                if (it < d.f_n0) acc0 += pacc[s];                   /* uniform */                                  \
                else if (it < n_items) acc1 += pacc[s];                                                            \
            }                                                                                                      \
        }
        static_assert(kMaxItems == 5, "the two rounds below cover products 0-2 and 3-4");
        MF_FB_ROUND(0, 3)
        MF_FB_ROUND(3, 2)
#undef MF_FB_ROUND
        // partial sums into the wave's exchange region, behind its two power rows (which other waves may still be reading)
        float* part = reinterpret_cast<float*>(ex) + kExPart;
        reinterpret_cast<f32x4v*>(part)[lane] = acc0;
        reinterpret_cast<f32x4v*>(part)[64 + lane] = acc1;
        }
        __syncthreads();
        if constexpr (BAND) {
            if (tid < 128) {                                        // frames 4 fq .. + 3 of filter f: its groups in bin order, log
                const int f = tid & 31, fq64 = (tid >> 5) * 256;
                static_assert(kBandGat == 8, "two halves of four");
                int off[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) off[e] = gat[e * 32 + f];
                const char* lds = reinterpret_cast<const char*>(smem) + fq64;
                f32x4v v = *reinterpret_cast<const f32x4v*>(lds + off[0]);
#pragma unroll
                for (int e = 1; e < 4; ++e) v += *reinterpret_cast<const f32x4v*>(lds + off[e]);
                if (d.f_gat_n > 4) {                                // (uniform; the reference's filterbank: four at most)
#pragma unroll
                    for (int e = 0; e < 4; ++e) off[e] = gat[(4 + e) * 32 + f];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v += *reinterpret_cast<const f32x4v*>(lds + off[e]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) LE[((tid >> 5) * 4 + r) * kLS + f] = logf(v[r] == 0.f ? kEps : v[r]);
            }
        } else if (tid < 128) {                                            // tile t = tid >> 6: sum in wave order, log
            const f32x4v* p = reinterpret_cast<const f32x4v*>(smem + kExPart) + tid;
            f32x4v v = p[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += p[w * (kEx / 2)];
            const int col = 16 * (tid >> 6) + row;                  // filter; rows = frames 4q + r
#pragma unroll
            for (int r = 0; r < 4; ++r) LE[(4 * q + r) * kLS + col] = logf(v[r] == 0.f ? kEps : v[r]);
        }
        __syncthreads();
        // ---- DCT-II x lifter: waves 0 and 1, one tile of 16 cepstra each; the others go on to the next tile
        if (wave < 2) {
            f32x4v db[2];
            if constexpr (MF_KNOCK_DCTLD) {
                db[0] = f32x4v{1.f, 0.5f, 0.25f, 2.f};
                db[1] = db[0];
                asm volatile("" : "+v"(db[0]), "+v"(db[1]));
            } else {
                db[0] = reinterpret_cast<const f32x4v*>(d.f_dct)[(wave * 2 + 0) * 64 + lane];
                db[1] = reinterpret_cast<const f32x4v*>(d.f_dct)[(wave * 2 + 1) * 64 + lane];
            }
            f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x4v a = *reinterpret_cast<const f32x4v*>(LE + row * kLS + 16 * g + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], db[g][j], acc, 0, 0, 0);
            }
            const int c = 16 * wave + row;
            if (c < d.numcep) {
                // (a descriptor over the tile's live frames: the range check is the test against total_frames, the lane's part one
                //  32-bit offset -- as pointer arithmetic this was five register pairs held across the whole tile loop)
                const int64_t left = total_frames - (int64_t)tile * kTile;
                const int live = left < kTile ? (int)left : kTile;
                const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(
                    out + (int64_t)tile * kTile * d.numcep, (short)0, live * d.numcep * 4, 0x00020000);
                int q4 = lane;                                      // (opaque: addresses made from it are made here, not held in
                asm volatile("" : "+v"(q4));                        //  registers across the tile loop)
                q4 = (q4 >> 4) * 4;
                const int vo = (q4 * d.numcep + c) * 4;
                if (c == 0 && d.append_energy) {                    // cepstrum 0 <- log of the frame energy
                    const f32x4v e4 = *reinterpret_cast<const f32x4v*>(en + par * kTile + q4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = logf(e4[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[r];                         // (never bit_cast an ELEMENT of a vector: hipcc 7.2 reads element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), o_rs, vo, r * d.numcep * 4, 0);
                }
            }
        }
    }
}

#undef MF_REQUEST
#undef MF_W8
#undef MF_LOAD_FRAME
#undef MF_LD

}  // namespace fft512

thread_local char g_merr[256] = "";
int mfail(int code, const char* msg) {
    snprintf(g_merr, sizeof(g_merr), "%s", msg);
    return code;
}

int round_half_up(double v) { return (int)std::floor(v + 0.5); }   // sigproc.round_half_up for v >= 0

}  // namespace

struct xvec_mfcc_plan {
    xvec_mfcc_cfg cfg;
    MfccDev dev;
    void* blob;   // one device allocation holding every table
    bool fast;    // nfft == 512, nfilt <= 32, numcep <= 32: fft512::mfcc512_kernel
    int num_cu;
};

extern "C" {

const char* xvec_mfcc_last_error(void) { return g_merr; }

int xvec_mfcc_create(const xvec_mfcc_cfg* cfg, xvec_mfcc_plan** out) {
    if (!cfg || !out) return mfail(XVEC_ERR_ARG, "null argument");
    const int nfft = cfg->nfft;
    int log2n = 0;
    while ((1 << log2n) < nfft) ++log2n;
    if (nfft < 64 || nfft > kMaxNfft || (1 << log2n) != nfft) return mfail(XVEC_ERR_ARG, "nfft must be a power of two in [64, 4096]");
    if (cfg->samplerate < 1 || cfg->nfilt < 1 || cfg->nfilt > kThreads || cfg->nfilt > nfft / 2 + 1 ||
        cfg->numcep < 1 || cfg->numcep > cfg->nfilt)
        return mfail(XVEC_ERR_ARG, "need 1 <= numcep <= nfilt <= min(256, nfft/2+1) and a positive sample rate");
    const int frame_len = round_half_up((double)cfg->winlen * cfg->samplerate);
    const int frame_step = round_half_up((double)cfg->winstep * cfg->samplerate);
    if (frame_len < 1 || frame_step < 1) return mfail(XVEC_ERR_ARG, "window length/step too small");
    if (hipSetDevice(cfg->device) != hipSuccess) return mfail(XVEC_ERR_HIP, "hipSetDevice failed");

    const int nbins = nfft / 2 + 1, nfilt = cfg->nfilt, numcep = cfg->numcep;
    // python_speech_features.base.get_filterbanks, in double
    const double high = cfg->highfreq > 0 ? cfg->highfreq : cfg->samplerate / 2.0, low = cfg->lowfreq;
    auto hz2mel = [](double hz) { return 2595.0 * std::log10(1.0 + hz / 700.0); };
    auto mel2hz = [](double mel) { return 700.0 * (std::pow(10.0, mel / 2595.0) - 1.0); };
    std::vector<double> bins(nfilt + 2);
    for (int i = 0; i < nfilt + 2; ++i) {
        const double mel = hz2mel(low) + (hz2mel(high) - hz2mel(low)) * i / (nfilt + 1);
        bins[i] = std::floor((nfft + 1) * mel2hz(mel) / cfg->samplerate);
    }
    // non-zero weights only: filter j covers bins [lo_j, lo_j + len_j)
    std::vector<float> fbw;
    std::vector<int> lo(nfilt), off(nfilt + 1);
    for (int j = 0; j < nfilt; ++j) {
        const int b0 = std::min((int)bins[j], nbins), b1 = std::min((int)bins[j + 1], nbins),
                  b2 = std::min((int)bins[j + 2], nbins);
        lo[j] = b0;
        off[j] = (int)fbw.size();
        for (int i = b0; i < b1; ++i) fbw.push_back((float)((i - bins[j]) / (bins[j + 1] - bins[j])));
        for (int i = b1; i < b2; ++i) fbw.push_back((float)((bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])));
    }
    off[nfilt] = (int)fbw.size();
    // scipy dct(type=2, norm='ortho') rows times base.lifter
    const int dct_ld = nfilt | 1;
    std::vector<float> dctl((size_t)numcep * dct_ld, 0.f);
    for (int k = 0; k < numcep; ++k) {
        const double scale = std::sqrt((k == 0 ? 1.0 : 2.0) / nfilt);
        const double lift = cfg->ceplifter > 0 ? 1.0 + (cfg->ceplifter / 2.0) * std::sin(M_PI * k / cfg->ceplifter) : 1.0;
        for (int m = 0; m < nfilt; ++m)
            dctl[(size_t)k * dct_ld + m] = (float)(std::cos(M_PI * k * (2 * m + 1) / (2.0 * nfilt)) * scale * lift);
    }
    // twiddles W_M^j = exp(-2 pi i j / M), grouped per pass of the kernel: for h = 1, 4, 16, ...
    // records {W_2h^j, W_4h^j, W_4h^(j+h)}, j < h; then, for odd log2(nfft), W_N^j, j < N/2
    std::vector<float> tw;
    auto push_w = [&](int j, int M) {
        tw.push_back((float)std::cos(2.0 * M_PI * j / M));
        tw.push_back((float)(-std::sin(2.0 * M_PI * j / M)));
    };
    int st = 0;
    for (; st + 1 < log2n; st += 2) {
        const int h = 1 << st;
        for (int j = 0; j < h; ++j) {
            push_w(j, 2 * h);
            push_w(j, 4 * h);
            push_w(j + h, 4 * h);
        }
    }
    if (st < log2n)
        for (int j = 0; j < nfft / 2; ++j) push_w(j, nfft);

    xvec_mfcc_plan* p = new (std::nothrow) xvec_mfcc_plan();
    if (!p) return mfail(XVEC_ERR_STATE, "out of host memory");
    memset(p, 0, sizeof(*p));
    p->cfg = *cfg;
    // one blob: twiddle | dctl | fb_w | fb_lo | fb_off (ints stored bit-for-bit in the float array)
    std::vector<float> blob;
    auto app_f = [&](const std::vector<float>& v) { int o = (int)blob.size(); blob.insert(blob.end(), v.begin(), v.end()); return o; };
    auto app_i = [&](const std::vector<int>& v) {
        int o = (int)blob.size();
        for (int x : v) { float fv; memcpy(&fv, &x, 4); blob.push_back(fv); }
        return o;
    };
    p->dev.tw_off = app_f(tw);
    p->dev.dctl_off = app_f(dctl);
    p->dev.fbw_off = app_f(fbw);
    p->dev.fblo_off = app_i(lo);
    p->dev.fboff_off = app_i(off);
    if (blob.size() & 1) blob.push_back(0.f);            // keep the per-wave regions 8-byte aligned
    p->dev.table_floats = (int)blob.size();
    // tables of the nfft == 512 kernel, after the LDS image (16-byte aligned)
    int f_tw1 = 0, f_tw2 = 0, f_fb = 0, f_dct = 0, f_band = -1, f_gat = 0;
    p->fast = (nfft == 512 && nfilt <= 32 && numcep <= 32);
    if (p->fast) {
        while (blob.size() & 3) blob.push_back(0.f);
        f_tw1 = (int)blob.size();
        for (int k = 1; k < 8; ++k)
            for (int l = 0; l < 64; ++l) {
                blob.push_back((float)std::cos(2.0 * M_PI * (l * k) / 512.0));
                blob.push_back((float)(-std::sin(2.0 * M_PI * (l * k) / 512.0)));
            }
        f_tw2 = (int)blob.size();
        for (int k = 1; k < 8; ++k)
            for (int c = 0; c < 8; ++c) {
                blob.push_back((float)std::cos(2.0 * M_PI * (c * k) / 64.0) * 0.015625f);   // (x 2^-6: fft512, kTwoM6)
                blob.push_back((float)(-std::sin(2.0 * M_PI * (c * k) / 64.0)) * 0.015625f);
            }
        // dense filterbank [32][256] (bin 256 never carries a weight: the last edge is exclusive) -> B fragments
        // of v_mfma_f32_16x16x4_f32: lane l of product (tile t, group g) holds FB[16t + (l & 15)][16g + 4(l >> 4) + j]
        std::vector<float> dense(32 * 256, 0.f);
        for (int j = 0; j < nfilt; ++j)
            for (int k = 0; k < off[j + 1] - off[j]; ++k)
                if (lo[j] + k < 256) dense[j * 256 + lo[j] + k] = 2.f * fbw[off[j] + k];   // (x 2: the kernel's power rows are halves, kTwoM6)
        f_fb = (int)blob.size();
        int g_lo[2] = {16, 16}, g_hi[2] = {0, 0};
        for (int t = 0; t < 2; ++t)
            for (int g = 0; g < 16; ++g)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        const float w = dense[(16 * t + (l & 15)) * 256 + 16 * g + 4 * (l >> 4) + j];
                        blob.push_back(w);
                        if (w != 0.f) {
                            g_lo[t] = std::min(g_lo[t], g);
                            g_hi[t] = std::max(g_hi[t], g + 1);
                        }
                    }
        for (int t = 0; t < 2; ++t)
            if (g_hi[t] <= g_lo[t]) g_lo[t] = g_hi[t] = 0;
        p->dev.f_lo0 = g_lo[0];
        p->dev.f_n0 = g_hi[0] - g_lo[0];
        p->dev.f_lo1 = g_lo[1];
        p->dev.f_n1 = g_hi[1] - g_lo[1];
        if (p->dev.f_n0 + p->dev.f_n1 > 4 * fft512::kMaxItems) p->fast = false;   // an unusually dense filterbank
        f_dct = (int)blob.size();
        for (int t = 0; t < 2; ++t)
            for (int g = 0; g < 2; ++g)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        const int c = 16 * t + (l & 15), m = 16 * g + 4 * (l >> 4) + j;
                        blob.push_back(c < numcep && m < nfilt ? dctl[(size_t)c * dct_ld + m] : 0.f);
                    }
        // ---- the banded form (fft512::kBand*): groups of consecutive bins whose filters fit a window of four
        {
            using namespace fft512;
            int fmin[256], fmax[256];
            for (int k = 0; k < 256; ++k) {
                fmin[k] = 32;
                fmax[k] = -1;
                for (int j = 0; j < 32; ++j)
                    if (dense[j * 256 + k] != 0.f) {
                        fmin[k] = std::min(fmin[k], j);
                        fmax[k] = std::max(fmax[k], j);
                    }
            }
            int g_k[16] = {}, g_cnt[16] = {}, g_a[16] = {}, n_groups = 0;   // (group 15 stays empty: all-zero weights)
            bool ok = true;
            for (int k = 0; k < 256 && ok;) {
                if (n_groups == kBandGroups) { ok = false; break; }
                int a = -1, cnt = 0;
                const int k0 = k;
                while (k < 256 && cnt < kBandCap) {
                    if (fmax[k] >= 0) {                            // (a bin without weight joins any group)
                        if (a < 0) a = std::min(fmin[k], 28);
                        if (fmax[k] > a + 3) break;
                    }
                    ++k;
                    ++cnt;
                }
                if (cnt == 0) { ok = false; break; }               // one bin wider than a window
                g_k[n_groups] = k0;
                g_cnt[n_groups] = cnt;
                g_a[n_groups++] = a < 0 ? 0 : a;
            }
            // where the partial sums of group g, frame quad 0, filter column j sit (bytes from the start of the block's LDS):
            // wave g >> 2 writes lane 16 fq + 4 (g & 3) + j
            auto part_byte = [&](int g, int j) { return ((g >> 2) * 2 * kEx + kExPart) * 4 + ((g & 3) * 4 + j) * 16; };
            std::vector<int> gat(kBandGat * 32, part_byte(15, 0));  // absent: a column of the all-zero group
            int gat_n = 0;
            for (int f = 0; f < 32 && ok; ++f) {
                int e = 0;
                for (int g = 0; g < n_groups; ++g) {
                    if (f < g_a[g] || f > g_a[g] + 3) continue;
                    bool any = false;
                    for (int k = g_k[g]; k < g_k[g] + g_cnt[g]; ++k) any = any || dense[f * 256 + k] != 0.f;
                    if (!any) continue;
                    if (e == kBandGat) { ok = false; break; }
                    gat[e++ * 32 + f] = part_byte(g, f - g_a[g]);
                }
                gat_n = std::max(gat_n, e);
            }
            p->dev.f_gat_n = gat_n;
            // (XVEC_MFCC_FILTERBANK=dense keeps the dense products for a filterbank the banded form can hold: tests and A/B timing)
            const char* force = getenv("XVEC_MFCC_FILTERBANK");
            if (force && strcmp(force, "dense") == 0) ok = false;
            if (ok) {
                while (blob.size() & 3) blob.push_back(0.f);
                f_band = (int)blob.size();
                // first bin a group's lanes read: anywhere in [k + cnt - 20, k] (bins in front of the group carry weight 0), inside
                // the row, and -- where that leaves a choice -- on a residue mod 4 no earlier group of the wave reads on
                int k_rd[16];
                for (int w = 0; w < 4; ++w) {
                    bool used[4] = {false, false, false, false};
                    for (int sl = 0; sl < 4; ++sl) {
                        const int g = 4 * w + sl;
                        const int hi_k = g < n_groups ? std::min(g_k[g], 256 - kBandN) : 256 - kBandN;
                        const int lo_k = g < n_groups ? std::max(0, g_k[g] + g_cnt[g] - kBandN) : 0;
                        int pick = hi_k;
                        for (int kk = hi_k; kk >= lo_k; --kk)
                            if (!used[kk & 3]) { pick = kk; break; }
                        used[pick & 3] = true;
                        k_rd[g] = pick;
                    }
                }
                auto k_read = [&](int g) { return k_rd[g]; };
                for (int w = 0; w < 4; ++w)
                    for (int q5 = 0; q5 < 5; ++q5)
                        for (int l = 0; l < 64; ++l)
                            for (int c = 0; c < 4; ++c) {          // weight of instruction n = 4 q5 + c: bin n of group 4 w + slot
                                const int g = 4 * w + ((l >> 2) & 3), t = l & 3, kk = k_read(g) + 4 * q5 + c;
                                const bool live = g < n_groups && kk >= g_k[g] && kk < g_k[g] + g_cnt[g];
                                blob.push_back(live ? dense[(g_a[g] + t) * 256 + kk] : 0.f);
                            }
                for (int w = 0; w < 4; ++w)                        // LDS byte address of frame 4 fq + t at the group's first bin
                    for (int l = 0; l < 64; ++l) {
                        const int fq = l >> 4, g = 4 * w + ((l >> 2) & 3), t = l & 3;
                        const int rowf = (t >> 1) ? fq * (2 * kEx + kExRow) + kExRow0 + (t & 1) * kPS : 4 * kEx * 2 + (2 * fq + (t & 1)) * kPS;
                        const int byte = (rowf + k_read(g)) * 4;
                        float fv;
                        memcpy(&fv, &byte, 4);
                        blob.push_back(fv);
                    }
                f_gat = app_i(gat);
            }
        }
    }
    if (hipMalloc(&p->blob, blob.size() * 4) != hipSuccess) {
        delete p;
        return mfail(XVEC_ERR_HIP, "hipMalloc failed");
    }
    if (hipMemcpy(p->blob, blob.data(), blob.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p->blob);
        delete p;
        return mfail(XVEC_ERR_HIP, "hipMemcpy failed");
    }
    p->dev.tables = static_cast<const float*>(p->blob);
    p->dev.f_tw1 = p->dev.tables + f_tw1;
    p->dev.f_tw2 = p->dev.tables + f_tw2;
    p->dev.f_fb = p->dev.tables + f_fb;
    p->dev.f_dct = p->dev.tables + f_dct;
    p->dev.f_band = f_band >= 0 ? p->dev.tables + f_band : nullptr;
    p->dev.f_gat = reinterpret_cast<const int*>(p->dev.tables + f_gat);
    p->dev.frame_len = frame_len;
    p->dev.frame_step = frame_step;
    p->dev.nfft = nfft;
    p->dev.log2n = log2n;
    p->dev.nbins = nbins;
    p->dev.nfilt = nfilt;
    p->dev.numcep = numcep;
    p->dev.append_energy = cfg->append_energy;
    p->dev.preemph = cfg->preemph;
    {
        hipDeviceProp_t prop;
        p->num_cu = hipGetDeviceProperties(&prop, cfg->device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    *out = p;
    return XVEC_OK;
}

void xvec_mfcc_destroy(xvec_mfcc_plan* p) {
    if (!p) return;
    if (p->blob) (void)hipFree(p->blob);
    delete p;
}

int32_t xvec_mfcc_kernel_form(const xvec_mfcc_plan* p) {
    if (!p) return -1;
    return !p->fast ? 0 : p->dev.f_band == nullptr ? 1 : 2;
}

int32_t xvec_mfcc_frames(const xvec_mfcc_plan* p, int64_t n_samples) {
    if (!p || n_samples < 1) return 0;
    if (n_samples <= p->dev.frame_len) return 1;
    return 1 + (int32_t)((n_samples - p->dev.frame_len + p->dev.frame_step - 1) / p->dev.frame_step);
}

static int mfcc_run(xvec_mfcc_plan* p, const void* signal, bool i16, float in_scale, int32_t B, int64_t n_samples, float* out,
                    xvec_stream stream) {
    if (!p || !signal || !out) return mfail(XVEC_ERR_ARG, "null argument");
    if (B < 1 || B > 65535 || n_samples < 1) return mfail(XVEC_ERR_ARG, "need 1 <= B <= 65535 and n_samples >= 1");
    const int n_frames = xvec_mfcc_frames(p, n_samples);
    const int64_t total_frames = (int64_t)B * n_frames;
    MfccDev dv = p->dev;
    dv.in_scale = in_scale;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (p->fast && total_frames < (int64_t(1) << 31) - fft512::kTile && n_samples < (int64_t(1) << 30)) {
        const int n_tiles = (int)((total_frames + fft512::kTile - 1) / fft512::kTile);
        const int grid = std::min(n_tiles, 5 * p->num_cu);   // five blocks of 30 016 bytes of LDS and 96 registers per CU
        dv.fr_shift = 0;
        while ((1u << dv.fr_shift) < (unsigned)n_frames) ++dv.fr_shift;
        dv.fr_magic = (unsigned)((((unsigned long long)1 << (31 + dv.fr_shift)) + n_frames - 1) / (unsigned)n_frames);
        const bool band = dv.f_band != nullptr;                 // (else: a filterbank the banded form cannot hold; the dense products)
        if (i16 && band) fft512::mfcc512_kernel<true, true><<<grid, 256, 0, hs>>>(signal, n_samples, n_frames, total_frames, n_tiles, dv, out);
        else if (i16) fft512::mfcc512_kernel<true, false><<<grid, 256, 0, hs>>>(signal, n_samples, n_frames, total_frames, n_tiles, dv, out);
        else if (band) fft512::mfcc512_kernel<false, true><<<grid, 256, 0, hs>>>(signal, n_samples, n_frames, total_frames, n_tiles, dv, out);
        else fft512::mfcc512_kernel<false, false><<<grid, 256, 0, hs>>>(signal, n_samples, n_frames, total_frames, n_tiles, dv, out);
        if (hipGetLastError() != hipSuccess) return mfail(XVEC_ERR_HIP, "mfcc kernel launch failed");
        return XVEC_OK;
    }
    const int per_wave = wave_floats(p->dev.nfft, p->dev.nbins);
    const size_t lds = ((size_t)per_wave * kWavesPerBlock + p->dev.table_floats) * 4;
    if (lds > 64 * 1024) {   // nfft 4096: opt in to the larger dynamic LDS once
        static xvec::LdsOptIn opt;
        if (opt.ensure(reinterpret_cast<const void*>(mfcc_kernel<0>), 160 * 1024) != hipSuccess)
            return mfail(XVEC_ERR_HIP, "hipFuncSetAttribute failed");
    }
    const int grid_x = (n_frames + 2 * kWavesPerBlock - 1) / (2 * kWavesPerBlock);
    const dim3 grid(grid_x, B);
    if (p->dev.log2n == 9) mfcc_kernel<9><<<grid, kThreads, lds, hs>>>(signal, i16 ? 1 : 0, n_samples, n_frames, dv, out);
    else mfcc_kernel<0><<<grid, kThreads, lds, hs>>>(signal, i16 ? 1 : 0, n_samples, n_frames, dv, out);
    if (hipGetLastError() != hipSuccess) return mfail(XVEC_ERR_HIP, "mfcc kernel launch failed");
    return XVEC_OK;
}

int xvec_mfcc(xvec_mfcc_plan* p, const float* signal, int32_t B, int64_t n_samples, float* out, xvec_stream stream) {
    return mfcc_run(p, signal, false, 1.0f, B, n_samples, out, stream);
}

int xvec_mfcc_i16(xvec_mfcc_plan* p, const int16_t* signal, float scale, int32_t B, int64_t n_samples, float* out,
                  xvec_stream stream) {
    return mfcc_run(p, signal, true, scale, B, n_samples, out, stream);
}

}  // extern "C"

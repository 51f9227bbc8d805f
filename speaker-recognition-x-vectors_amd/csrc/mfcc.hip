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

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
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
template <int L2N>
__global__ __launch_bounds__(kThreads) void mfcc_kernel(const float* __restrict__ sig, int64_t n_samples,
                                                        int n_frames, MfccDev d, float* __restrict__ out) {
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
    const float* s = sig + (int64_t)b * n_samples;
    const int64_t start = (int64_t)fa * d.frame_step;
    const int used = d.frame_len < nfft ? d.frame_len : nfft;   // rfft(frame, nfft) truncates long frames

    // ---- the two pre-emphasised frames, zero padded to nfft, as one complex signal in bit-reversed order
#pragma unroll
    for (int n = lane; n < nfft; n += 64) {
        float va = 0.f, vb = 0.f;
        if (n < used) {
            const int64_t ga = start + n, gb = ga + d.frame_step;
            if (ga < n_samples) va = (ga == 0) ? s[0] : s[ga] - d.preemph * s[ga - 1];
            if (has_b && gb < n_samples) vb = s[gb] - d.preemph * s[gb - 1];
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
    p->dev.frame_len = frame_len;
    p->dev.frame_step = frame_step;
    p->dev.nfft = nfft;
    p->dev.log2n = log2n;
    p->dev.nbins = nbins;
    p->dev.nfilt = nfilt;
    p->dev.numcep = numcep;
    p->dev.append_energy = cfg->append_energy;
    p->dev.preemph = cfg->preemph;
    *out = p;
    return XVEC_OK;
}

void xvec_mfcc_destroy(xvec_mfcc_plan* p) {
    if (!p) return;
    if (p->blob) (void)hipFree(p->blob);
    delete p;
}

int32_t xvec_mfcc_frames(const xvec_mfcc_plan* p, int64_t n_samples) {
    if (!p || n_samples < 1) return 0;
    if (n_samples <= p->dev.frame_len) return 1;
    return 1 + (int32_t)((n_samples - p->dev.frame_len + p->dev.frame_step - 1) / p->dev.frame_step);
}

int xvec_mfcc(xvec_mfcc_plan* p, const float* signal, int32_t B, int64_t n_samples, float* out, xvec_stream stream) {
    if (!p || !signal || !out) return mfail(XVEC_ERR_ARG, "null argument");
    if (B < 1 || B > 65535 || n_samples < 1) return mfail(XVEC_ERR_ARG, "need 1 <= B <= 65535 and n_samples >= 1");
    const int n_frames = xvec_mfcc_frames(p, n_samples);
    const int per_wave = wave_floats(p->dev.nfft, p->dev.nbins);
    const size_t lds = ((size_t)per_wave * kWavesPerBlock + p->dev.table_floats) * 4;
    if (lds > 64 * 1024) {   // nfft 4096: opt in to the larger dynamic LDS once
        static xvec::LdsOptIn opt;
        if (opt.ensure(reinterpret_cast<const void*>(mfcc_kernel<0>), 160 * 1024) != hipSuccess)
            return mfail(XVEC_ERR_HIP, "hipFuncSetAttribute failed");
    }
    const int grid_x = (n_frames + 2 * kWavesPerBlock - 1) / (2 * kWavesPerBlock);
    if (p->dev.log2n == 9)
        mfcc_kernel<9><<<dim3(grid_x, B), kThreads, lds, static_cast<hipStream_t>(stream)>>>(signal, n_samples, n_frames,
                                                                                           p->dev, out);
    else
        mfcc_kernel<0><<<dim3(grid_x, B), kThreads, lds, static_cast<hipStream_t>(stream)>>>(signal, n_samples, n_frames,
                                                                                           p->dev, out);
    if (hipGetLastError() != hipSuccess) return mfail(XVEC_ERR_HIP, "mfcc kernel launch failed");
    return XVEC_OK;
}

}  // extern "C"

/* xvec_hip.h -- C ABI of libxvec_hip.so: the MI355X (gfx950) x-vector extraction path.
 *
 * The reference (TorbenHellriegel/Speaker-Recognition-x-vectors) has no FFI, plugin or
 * operator interface: its boundary for this path is the Python method surface of
 * XVectorModel (reference main.py:23-94).  The entry points below are what a ctypes
 * binding of that surface needs (INTEGRATION.md shows the binding); each comment names
 * the reference code the entry point replaces.
 *
 * Conventions
 *   - every tensor pointer is a DEVICE pointer (HIP) unless the name ends in _host;
 *   - tensors are contiguous, batch-first, time-major, channels-last, fp32 -- exactly the
 *     [B,T,C] layout the reference feeds (main.py:99,137);
 *   - all work is enqueued on the caller's stream and is asynchronous w.r.t. the host; nothing
 *     allocates device, pinned or host heap memory after xvec_create (the fixed-length path is therefore
 *     capturable into a hipGraph).  What is synchronous: xvec_create / xvec_destroy; xvec_get_timings
 *     (waits for the last recorded event); and the ragged entry points (lengths_host / offsets_host
 *     given) copy the offsets through a ring of two pinned staging slots owned by the handle, so
 *     only a third ragged call issued before the first one's offset copy has run waits for that copy;
 *   - every entry point that takes a handle runs on the handle's device and leaves the calling
 *     thread's current device as it found it (xvec_create included);
 *   - limits: at most 65535 utterances per call; input_size, hidden_size <= 8192; rows are addressed
 *     with 32-bit byte offsets, so (utterances x context + 264) x channels x element size must stay
 *     below 2^31 (XVEC_ERR_TOO_LARGE otherwise -- split the batch);
 *   - every function returns XVEC_OK (0) or an error code and never throws; the message
 *     for the last error on the calling thread is available from xvec_last_error();
 *   - one handle per device; distinct handles may be used from distinct threads.
 */
#ifndef XVEC_HIP_H
#define XVEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xvec_handle xvec_handle;
typedef void* xvec_stream; /* hipStream_t */

enum {
    XVEC_OK = 0,
    XVEC_ERR_ARG = 1,
    XVEC_ERR_HIP = 2,
    XVEC_ERR_STATE = 3,
    XVEC_ERR_WORKSPACE = 4,
    XVEC_ERR_TOO_LARGE = 5 /* the batch exceeds one of the per-call size limits below: the same data in several smaller calls works */
};

/* arithmetic of the frame-level stack (accumulation and pooling are fp32 in all three; the segment-level affines
 * are fp32 MFMAs in XVEC_F32 and XVEC_BF16X3, and bf16x3 products -- fp32 operands as bf16 pairs, 6e-6 of fp64 -- in
 * XVEC_BF16) */
enum {
    XVEC_F32 = 0,   /* exact fp32 on v_mfma_f32_32x32x2_f32 (the reference's arithmetic) */
    XVEC_BF16 = 1,  /* bf16 activations and weights (parity bar 1e-2).  Between layers this mode keeps the ReLU outputs (bf16) and
                     * DEFERS each layer's eval BatchNorm (tdnn_layer.py:36-39) into the next layer's weights and bias -- an affine
                     * map of a valid convolution's input folds exactly; layer 5's goes to the pooling merge.  The per-layer entry
                     * points below still take and return the reference's tensors (BatchNorm applied): they take x back through the
                     * producing layer's BatchNorm, r = (x - shift) / scale; a channel whose folded scale is 0 (gamma = 0) or below
                     * 1e-18 in magnitude is read as x = shift, which is what the reference's BatchNorm gives there for any r.
                     * Pooling in this mode sums bf16-rounded deviations about ONE pivot per block of ~27 x 64 frames, so the same
                     * utterance gives x-vectors that differ by up to ~1e-3 (bar 1e-2) with its position in the batch, the batch's
                     * make-up and the way a job is cut into calls; XVEC_F32 / XVEC_BF16X3: <= 1e-5.  Repeat calls are bit-identical. */
    /* fp32 values carried as two bf16 planes (hi + lo), three bf16 products per k-step
     * (x_hi*W_hi + x_lo*W_hi + x_hi*W_lo; every product is exact in the fp32 accumulator, what is
     * dropped is 2^-16 relative): fp32-level results (parity bar 1e-4, as XVEC_F32) at bf16 matrix rates */
    XVEC_BF16X3 = 2
};

/* what xvec_forward returns */
enum {
    XVEC_MODE_LOGITS = 0, /* XVectorModel.forward        main.py:66-75  -> [B,num_classes] */
    XVEC_MODE_POOLED = 5, /* stat_pool(time_context_layers(x)), main.py:68-69 / 83-84 -> [B,3000] = mean | std:
                           * the frame-level stack with its fused pooling alone (segment-level weights not needed) */
    XVEC_MODE_XVEC6 = 6,  /* extract_x_vec, layer 6      main.py:86-87  -> [B,x_vector_size] */
    XVEC_MODE_XVEC7 = 7   /* extract_x_vec, layer 7      main.py:88-90  -> [B,x_vector_size] */
};

/* which segment-level affine (main.py:45-47) */
enum { XVEC_SEG6 = 6, XVEC_SEG7 = 7, XVEC_OUTPUT = 8 };

#define XVEC_NUM_TDNN 5        /* main.py:38-44 */
#define XVEC_POOL_CHANNELS 1500 /* main.py:43 (hard-coded in the reference) */
#define XVEC_TOTAL_CONTEXT 14   /* frames consumed by the valid convolutions */
#define XVEC_MAX_TIMINGS 16

/* XVectorModel.__init__ shape arguments (main.py:24-34; defaults config.py:4-12) */
typedef struct {
    int32_t input_size;    /* 24   */
    int32_t hidden_size;   /* 512  */
    int32_t num_classes;   /* 1211 */
    int32_t x_vector_size; /* 512  */
    int32_t batch_norm;    /* 1: TdnnLayer has BatchNorm1d after the ReLU (tdnn_layer.py:36-39) */
    int32_t device;        /* HIP device ordinal the handle lives on */
} xvec_cfg;

/* ---- lifetime ---------------------------------------------------------------------- */
int xvec_create(const xvec_cfg* cfg, xvec_handle** out);
void xvec_destroy(xvec_handle* h);
const char* xvec_last_error(void);
/* "xvec_hip gfx950 build <id>": id = hash of the sources the library was built from (csrc/Makefile, BUILD_ID) */
const char* xvec_version(void);

/* ---- parameters (replaces nn.Module state: load_state_dict, main.py:213) -------------
 * Pointers are device fp32 tensors in PyTorch layout; the library re-packs them on the
 * device (tap-major padded weights, BatchNorm folded to scale/shift, bf16 copies).
 *   layer 0..4: time_context_layers.{layer}.linear.{weight[out, in*|ctx|], bias[out]}
 *               time_context_layers.{layer}.norm.{weight,bias,running_mean,running_var}[out]
 *               (all four NULL when cfg.batch_norm == 0); eps = BatchNorm1d.eps (1e-5). */
/* Layers may be loaded in any order and on different streams: loading layer l also re-packs the bf16 copies of layer l + 1
 * (whose weights carry layer l's BatchNorm in XVEC_BF16) from what an earlier xvec_load_tdnn(l + 1) left on ITS stream, and
 * waits for that load's event first.  A forward call must still be ordered behind the loads by the caller (same stream, or a
 * synchronisation), like any consumer of stream-ordered work. */
int xvec_load_tdnn(xvec_handle* h, int layer, const float* weight, const float* bias,
                   const float* bn_weight, const float* bn_bias, const float* bn_mean,
                   const float* bn_var, float eps, xvec_stream stream);
/* which = XVEC_SEG6 / XVEC_SEG7 / XVEC_OUTPUT: {segment_layer6,segment_layer7,output}.{weight,bias} */
int xvec_load_affine(xvec_handle* h, int which, const float* weight, const float* bias,
                     xvec_stream stream);

/* ---- whole path --------------------------------------------------------------------
 * Bytes of scratch xvec_forward needs for B utterances totalling total_frames input
 * frames (B*T for a fixed-length batch).  */
size_t xvec_workspace_bytes(const xvec_handle* h, int64_t total_frames, int32_t n_utts);

/* XVectorModel.forward / extract_x_vec (main.py:66-94) on x[B,T,input_size].
 *   lengths_host: NULL (every utterance has T frames, the reference's only case) or a
 *                 HOST array of B valid-frame counts, 15 <= lengths[i] <= T: utterance i
 *                 is x[i, :lengths[i]] and the result equals the reference run on that
 *                 un-padded slice alone (BASELINE config 3).
 *   mode:         XVEC_MODE_*;  dtype: XVEC_F32 / XVEC_BF16 / XVEC_BF16X3 (the last one: at most
 *                 ~1M frames per call at 512 channels -- its second plane is addressed with 30 bits)
 *   out:          [B, num_classes] (logits), [B, x_vector_size] or [B, 2*XVEC_POOL_CHANNELS] (XVEC_MODE_POOLED)
 * Errors: T (or a length) < 15 -> XVEC_ERR_ARG (the reference silently yields empty
 * tensors / NaN there); weights not loaded -> XVEC_ERR_STATE. */
int xvec_forward(xvec_handle* h, const float* x, const int32_t* lengths_host, int32_t B,
                 int32_t T, int mode, int dtype, float* out, void* workspace,
                 size_t workspace_bytes, xvec_stream stream);

/* Same on a packed ragged batch: x_packed[offsets_host[B], input_size] holds utterance i
 * in rows [offsets_host[i], offsets_host[i+1]). */
int xvec_forward_packed(xvec_handle* h, const float* x_packed, const int64_t* offsets_host,
                        int32_t B, int mode, int dtype, float* out, void* workspace,
                        size_t workspace_bytes, xvec_stream stream);

/* ---- per-stage entry points (unit tests; each mirrors one reference function) --------
 * TdnnLayer.forward (tdnn_layer.py:26-41), eval mode, for time_context_layers.{layer}:
 * x[B,T,in] -> y[B,T-(c[-1]-c[0]),out].  workspace >= xvec_workspace_bytes(h, B*T, B). */
int xvec_tdnn_layer(xvec_handle* h, int layer, const float* x, int32_t B, int32_t T, int dtype,
                    float* y, void* workspace, size_t workspace_bytes, xvec_stream stream);
/* stat_pool(time_context_layers[4](x)) (main.py:44,59-63 as forward composes them, main.py:68-69): the last
 * frame-level layer with the statistics pooling fused into its epilogue, exactly as xvec_forward runs it, on a
 * caller-given input x[B,T,hidden] -> out[B, 2*XVEC_POOL_CHANNELS].  workspace >= xvec_workspace_bytes(h, B*T, B). */
int xvec_tdnn_pool_layer(xvec_handle* h, const float* x, int32_t B, int32_t T, int dtype, float* out,
                         void* workspace, size_t workspace_bytes, xvec_stream stream);
/* XVectorModel.stat_pool (main.py:59-63): x[B,T,C] -> out[B,2C] = mean ‖ unbiased std.
 * lengths_dev: NULL or DEVICE int32[B] valid-frame counts (mask).  Stand-alone: no handle. */
int xvec_stat_pool(const float* x, const int32_t* lengths_dev, int32_t B, int32_t T, int32_t C,
                   float* out, xvec_stream stream);
/* nn.Linear (+ optional F.relu) of segment_layer6 / segment_layer7 / output
 * (main.py:72-75,87-90): x[M,in] -> y[M,out]. */
int xvec_affine(xvec_handle* h, int which, const float* x, int32_t M, int relu, float* y,
                xvec_stream stream);

/* ---- test introspection -------------------------------------------------------------
 * Where the regions of a workspace lie that a test may want to look into after a call (tests/test_segmx_exact_gpu.py reads
 * layer 5's pooling partials): byte offsets for a batch of n_utts utterances totalling total_frames frames, exactly as
 * xvec_forward / xvec_tdnn_pool_layer lay them out.  Nothing in the product calls this.
 *   part:     pooling partials [part_slots][3 planes K | S1 | S2][pool_n_pad] fp32 (csrc/tdnn_common.h; which slot holds
 *             what depends on the kernel layer 5 went to: xvec_get_dispatch)
 *   part_cnt: int32 frames behind each segment partial of the large-batch kernel (csrc/tdnn_pp16.hip)
 *   act_a / act_b: the two frame-level activation buffers, rows_alloc rows each (the per-stage entries stage their input in act_a) */
typedef struct {
    size_t act_a, act_b, part, part_cnt, pooled, bytes;
    int64_t rows_alloc, part_slots;
    int32_t pool_n_pad, hidden_n_pad, num_cu;
} xvec_ws_layout;
int xvec_workspace_layout(const xvec_handle* h, int64_t total_frames, int32_t n_utts, xvec_ws_layout* out);

/* ---- measurement --------------------------------------------------------------------
 * With profiling on, xvec_forward brackets every kernel with hipEvents on the caller's
 * stream.  xvec_get_timings synchronises on the last event and returns milliseconds:
 * ms[0..4] TDNN layers 1-5 (layer 5 includes its fused pooling epilogue), ms[5] pooling
 * finalize / stand-alone pooling, ms[6..8] segment_layer6 / 7 / output (0 if not run),
 * ms[9] input packing (ragged batches, channel padding, the bf16x3 hi/lo split; 0 when the first layer
 * reads the caller's tensor directly, as in fp32 and bf16 on fixed-length batches); a segment-layer figure
 * covers both launches of its split-K form; *n = 10. */
int xvec_set_profiling(xvec_handle* h, int on);
int xvec_get_timings(xvec_handle* h, float* ms, int* n);
/* Which kernel the LAST launch of each frame-level layer went to (the choice depends on arithmetic, widths and
 * batch size): kernels[0..4], *n = 5.  bench.py names its roofline kernel and keys profiles/traffic.json by this. */
enum {
    XVEC_KERNEL_NONE = 0,
    XVEC_KERNEL_TILE128 = 1, /* xvec::tdnn_kernel<...>          128x128 tiles (csrc/tdnn_layer.hip) */
    XVEC_KERNEL_PP = 2,      /* xvec::pp16::tdnn_pp_kernel<POOL, X3>  256-channel LDS-DMA mapping, bf16 / bf16x3 at large batches (csrc/tdnn_pp16.hip) */
    XVEC_KERNEL_FIRST = 3    /* xvec::first::tdnn_first_kernel / first3::tdnn_first3_kernel  layer 1 of the bf16 / bf16x3 path, streaming (csrc/tdnn_first.hip) */
};
int xvec_get_dispatch(const xvec_handle* h, int* kernels, int* n);

/* ---- next row N3: MFCC front end (the step in front of the path) ----------------------------
 * python_speech_features.mfcc as the reference calls it in its DataLoader workers
 * (reference dataset.py:128: mfcc(signal, 16000, numcep=24, nfilt=26, nfft=512)): pre-emphasis,
 * rectangular 25 ms / 10 ms frames with a zero-padded tail, |rfft|^2/nfft, triangular mel
 * filterbank, log, orthonormal DCT-II, lifter, c0 := log frame energy.  One kernel, one launch per
 * batch of equal-length waveforms.  Parity is UNPINNED: the package is absent from the build image
 * (oracle/mfcc_oracle.py restates its published algorithm). */
typedef struct xvec_mfcc_plan xvec_mfcc_plan;
typedef struct {
    int32_t samplerate;    /* 16000 */
    float winlen;          /* 0.025 s */
    float winstep;         /* 0.01 s  */
    int32_t numcep;        /* 24 in the reference (package default 13) */
    int32_t nfilt;         /* 26 */
    int32_t nfft;          /* 512; power of two in [64, 4096] */
    float lowfreq;         /* 0 */
    float highfreq;        /* 0 = samplerate/2 */
    float preemph;         /* 0.97 */
    int32_t ceplifter;     /* 22; 0 = no liftering */
    int32_t append_energy; /* 1 */
    int32_t device;
} xvec_mfcc_cfg;
int xvec_mfcc_create(const xvec_mfcc_cfg* cfg, xvec_mfcc_plan** out);
void xvec_mfcc_destroy(xvec_mfcc_plan* plan);
const char* xvec_mfcc_last_error(void);
/* Which kernel serves the plan (introspection for tests and benchmarks; no reference counterpart): 0 = the general kernel
 * (nfft != 512, nfilt or numcep > 32), 1 = the nfft-512 kernel with the mel filterbank as dense 16 x 16 x 16 products,
 * 2 = the same with the filterbank as a banded product (every bin group's filters inside a window of four; the reference's
 * call takes this one).  -1 for a null plan.  XVEC_MFCC_FILTERBANK=dense in the environment at create time forces 1 over 2. */
int32_t xvec_mfcc_kernel_form(const xvec_mfcc_plan* plan);
/* frames produced for n_samples samples: 1 + ceil((n - frame_len)/frame_step), 1 if n <= frame_len */
int32_t xvec_mfcc_frames(const xvec_mfcc_plan* plan, int64_t n_samples);
/* signal[B, n_samples] fp32 (device) -> out[B, frames, numcep] fp32 (device) */
int xvec_mfcc(xvec_mfcc_plan* plan, const float* signal, int32_t B, int64_t n_samples, float* out,
              xvec_stream stream);
/* The same for 16-bit PCM as scipy.io.wavfile.read yields it (reference dataset.py:125): every sample enters as
 * (float)s * scale -- one fp32 rounding, exactly what xvec_mfcc is given when the caller converts first, so the two agree bit
 * for bit; half the bytes over PCIe and out of HBM.  scale = 1 keeps the raw PCM range the reference's call sees. */
int xvec_mfcc_i16(xvec_mfcc_plan* plan, const int16_t* signal, float scale, int32_t B, int64_t n_samples, float* out,
                  xvec_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* XVEC_HIP_H */

// C-ABI layer of libxvec_hip.so (see include/xvec_hip.h): handle, parameter packing,
// workspace planning and the launch sequence of the extraction path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/xvec_hip.h"
#include "xvec_internal.h"

using namespace xvec;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(XVEC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                  \
    } while (0)

// reference main.py:38-44
const int kCtxLen[XVEC_NUM_TDNN] = {5, 3, 3, 1, 1};
const int kCtxDil[XVEC_NUM_TDNN] = {1, 2, 3, 0, 0};

enum { T_L1 = 0, T_POOL = 5, T_SEG6 = 6, T_SEG7 = 7, T_OUT = 8, T_PACK = 9, T_COUNT = 10 };
constexpr int kMaxUtts = 65535;   // utterances per call (16-bit utterance counters in the kernels' row maps)

// RAII: the calling thread's current device is whatever it was before the call
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t enter(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev != dev) {
            e = hipSetDevice(dev);
            changed = e == hipSuccess;
        }
        return e;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct xvec_handle {
    xvec_cfg cfg;
    int cin_pad;                       // input_size rounded up to 4 (row stride of layer-1 input)
    TdnnGeom geo[XVEC_NUM_TDNN];
    TdnnGeom geo16[XVEC_NUM_TDNN];     // bf16 packing of layers 2-5: 64-element chunks (layer 1 stays fp32)
    void* Wp16[XVEC_NUM_TDNN];         // bf16, fragment-major
    void* Wp16b;                       // layer 1 only: the same with the bias in the two spare k slots (tdnn_first.hip)
    void* Wr16[XVEC_NUM_TDNN];         // bf16, K-tile major [n_pad/256][k_pad/64][256][64] (tdnn_pp16.hip: both operands reach LDS by DMA)
    bool use_pp;                       // large-batch bf16 mapping enabled (XVEC_PP=0 disables it: A/B runs)
    int pp_min_tenths;                 // ... from this many tenths of a 64-frame unit per CU on (18; XVEC_PP_MIN_TENTHS: crossover sweeps)
    int pp_cu_pct;                     // XVEC_PP_CU_PCT (diagnostic): percentage of the CUs the large-batch kernel's grid covers (0 = all)
    void* Wp48[XVEC_NUM_TDNN];         // bf16x3: per chunk W_hi then W_lo fragments (2x the size), fragment-major
    void* Wr48[XVEC_NUM_TDNN];         // bf16x3, K-tile major for tdnn_pp16.hip: per K-tile W_hi | W_lo | W_hi (3x the size of Wr16)
    float* Wp[XVEC_NUM_TDNN];
    float* vec[XVEC_NUM_TDNN];         // bias | scale | shift, n_pad each
    // Plain bf16 DEFERS every layer's BatchNorm into its consumer (refold below): layer l stores relu(z + bias'), layer l+1's
    // bf16 weights carry scale_l and its bias' the shift_l (layer 5's BatchNorm goes to pool_finalize as in every mode).
    float* Wraw[XVEC_NUM_TDNN];        // the caller's fp32 weight [cout][taps*cin] and bias, kept for re-folding when the
    float* braw[XVEC_NUM_TDNN];        // producing layer's BatchNorm is (re)loaded later
    float* vec16[XVEC_NUM_TDNN];       // bias' | 1 | 0 (n_pad each): epilogue constants of the plain-bf16 kernels
    bool folded[XVEC_NUM_TDNN];        // Wp16 / Wr16 / vec16 of the layer are consistent with the producing layer's BatchNorm
    hipEvent_t load_evt[XVEC_NUM_TDNN]; // recorded behind a layer's load on the stream that carried it: a re-fold on ANOTHER stream waits for it
    bool tdnn_loaded[XVEC_NUM_TDNN];
    float* affW[3];
    float* affB[3];
    void* affW3[3];                    // affW as bf16 (hi, lo) pairs for the bf16 modes' segment layers (launch_split_pairs); K % 4 == 0
    int affN[3], affK[3];
    bool aff_loaded[3];
    // ragged batches: host offsets staged through pinned memory
    // (allocated once in xvec_create: two slots of kMaxUtts+1 entries, so the ragged entry points
    // never allocate and block only when a third ragged call arrives before the first one's copy left)
    int64_t* offs_pinned[2];
    hipEvent_t offs_evt[2];
    bool offs_pending[2];
    int offs_next;
    int num_cu;
    int blocks_per_cu;                 // persistent TDNN blocks per CU (LDS allows 2)
    int last_kernel[XVEC_NUM_TDNN];    // XVEC_KERNEL_* the last launch of each frame-level layer went to (xvec_get_dispatch)
    // profiling
    bool profiling;
    hipEvent_t ev0[T_COUNT], ev1[T_COUNT];
    bool ev_used[T_COUNT];
};

namespace {

struct Plan {
    int64_t total, m_pad, rows_alloc;
    size_t xpad, x16, actA, actB, act5, part, part_cnt, pooled, seg6, seg7, offs, bytes;
    int64_t part_slots;
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

Plan make_plan(const xvec_handle* h, int64_t total, int B) {
    Plan p;
    p.total = total;
    p.m_pad = round_up64(total > 0 ? total : 1, 128);
    p.rows_alloc = p.m_pad + kRowPadTail;
    const int nh = h->geo[0].n_pad, n5 = h->geo[4].n_pad;
    size_t o = 0;
    p.xpad = o;   o += align_up((size_t)p.rows_alloc * h->cin_pad * 4);
    p.x16 = o;    o += align_up((size_t)p.rows_alloc * h->cin_pad * 2 * 2);   // bf16 rows; bf16x3: hi and lo planes
    p.actA = o;   o += align_up((size_t)p.rows_alloc * nh * 4);
    p.actB = o;   o += align_up((size_t)p.rows_alloc * nh * 4);
    p.act5 = o;   o += align_up((size_t)p.rows_alloc * (n5 > nh ? n5 : nh) * 4);   // also the fp32 output of xvec_tdnn_layer
    // pooling partials: one per (32-row group, utterance) -- or, tdnn_pp16.hip, two per (block of a column, utterance)
    p.part_slots = std::max<int64_t>(p.m_pad / 32 + B + 1, 2 * ((int64_t)h->num_cu + B) + 2);
    p.part = o;   o += align_up((size_t)p.part_slots * 3 * n5 * 4);   // (addressed with 32-bit offsets: forward_rows checks < 2 GiB)
    p.part_cnt = o; o += align_up((size_t)2 * (h->num_cu + B + 2) * 4);     // tdnn_pp16.hip: frames behind each segment partial
    p.pooled = o; o += align_up((size_t)B * 2 * XVEC_POOL_CHANNELS * 4);
    p.seg6 = o;   o += align_up((size_t)B * h->cfg.x_vector_size * 4);
    p.seg7 = o;   o += align_up((size_t)B * h->cfg.x_vector_size * 4);
    p.offs = o;   o += align_up((size_t)(B + 1) * 8);
    p.bytes = o;
    return p;
}

// chunk_pair = K elements of two 128-byte chunks (64 for fp32, 128 for bf16): the kernel's main
// loop consumes whole chunk pairs per tap
void fill_geometry(xvec_handle* h, TdnnGeom* geo, int chunk_pair) {
    const int hid = h->cfg.hidden_size;
    for (int i = 0; i < XVEC_NUM_TDNN; ++i) {
        TdnnGeom& g = geo[i];
        g.src_taps = kCtxLen[i];
        g.src_cin = (i == 0) ? h->cfg.input_size : hid;
        g.cin = g.src_cin;
        g.cout = (i == 4) ? XVEC_POOL_CHANNELS : hid;
        g.n_pad = round_up(g.cout, 128);
        g.ctx_span = (kCtxLen[i] - 1) * kCtxDil[i];
        if (kCtxLen[i] == 1 || kCtxDil[i] == 1) {
            // contiguous context: the taps of one output row are one contiguous run of the
            // input buffer (row stride ldx), so they fold into a single tap of K = taps*ldx
            const int ldx = (i == 0) ? h->cin_pad : hid;
            g.n_taps = 1;
            g.tap_rows = 0;
            g.tap_stride_src = (kCtxLen[i] == 1) ? round_up(g.cin, chunk_pair) : ldx;
            g.kpt = (kCtxLen[i] == 1) ? g.cin : kCtxLen[i] * ldx;
        } else {
            g.n_taps = kCtxLen[i];
            g.tap_rows = kCtxDil[i];
            g.kpt = g.cin;
            g.tap_stride_src = round_up(g.cin, chunk_pair);
        }
        g.chunk_k = chunk_pair / 2;
        g.kpt_pad = round_up(g.kpt, chunk_pair);
        g.k_pad = g.n_taps * g.kpt_pad;
    }
}

int aff_index(int which) { return which == XVEC_SEG6 ? 0 : which == XVEC_SEG7 ? 1 : which == XVEC_OUTPUT ? 2 : -1; }

struct StageTimer {
    xvec_handle* h;
    int idx;
    hipStream_t s;
    StageTimer(xvec_handle* h_, int idx_, hipStream_t s_) : h(h_), idx(idx_), s(s_) {
        if (h->profiling) (void)hipEventRecord(h->ev0[idx], s);
    }
    ~StageTimer() {
        if (h->profiling) {
            (void)hipEventRecord(h->ev1[idx], s);
            h->ev_used[idx] = true;
        }
    }
};

// The large-batch mapping (tdnn_pp16.hip) takes a layer when its column count is a multiple of 256 and every CU gets at least
// pp_min_tenths / 10 units of 64 frames (1.8: the crossover with the 128x128 kernel measured in round 3 -- 51 utterances of
// 300 frames for layers 2-4, 18 for layer 5; the same for bf16x3).  Returns the blocks per 256-channel column, 0 if not.
int pp_blocks_per_col(const xvec_handle* h, int n_pad, int64_t rows_out) {
    if (!h->use_pp || n_pad % 256 != 0) return 0;
    int bpc = h->num_cu / (n_pad / 256);
    if (h->pp_cu_pct > 0 && h->pp_cu_pct < 100) bpc = (bpc * h->pp_cu_pct + 50) / 100;      // diagnostic knob: fewer, longer row ranges
    const int64_t units = (rows_out + 63) / 64;
    return (bpc >= 1 && units >= bpc && 10 * units >= h->pp_min_tenths * (int64_t)bpc) ? bpc : 0;
}

// What run_tdnn launched: kernel family and, for the large-batch mapping, the geometry its pooling partials were written
// with.  Returned to the caller and handed to finalize_pool explicitly (taken from handle state it was right only
// as long as layer 5 happened to be the last launch before the finalize, on the one thread using the handle).
struct Dispatch {
    int kernel = XVEC_KERNEL_NONE;
    int64_t pool_units = 0;
    int pool_bpc = 0;
};

// Launch one frame-level layer on flat rows.  The variant selects arithmetic and epilogue; bf16
// variants use the bf16 packing (64-element chunks) of the layer's weights.
// x3: bf16x3 arithmetic -- X (and Y, when it is bf16) are two bf16 planes `x_plane` / `y_plane` bytes apart
int run_tdnn(xvec_handle* h, int layer, TdnnVariant v, const void* X, int ldx, int64_t x_rows, void* Y,
             int64_t rows_out, const RowMap& out_map, float* part, hipStream_t s, bool x3 = false,
             int64_t x_plane = 0, int64_t y_plane = 0, int* part_cnt = nullptr, Dispatch* disp = nullptr) {
    Dispatch d_local;
    Dispatch& d = disp ? *disp : d_local;
    const bool in16 = v == TdnnVariant::kBf16 || v == TdnnVariant::kBf16Pool || v == TdnnVariant::kBf16ToF32 ||
                      v == TdnnVariant::kBf16First || v == TdnnVariant::kBf16FirstToF32 || v == TdnnVariant::kBf16FirstSrc32;
    const TdnnGeom& g = in16 ? h->geo16[layer] : h->geo[layer];
    TdnnArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X;
    a.W = h->Wp[layer];
    a.Wf = h->Wp16[layer];
    // plain bf16: the folded constants (bias', 1, 0) -- BatchNorm is deferred into the consumer's weights (refold)
    const float* vecs = (in16 && !x3) ? h->vec16[layer] : h->vec[layer];
    a.bias = vecs;
    a.scale = vecs + g.n_pad;
    a.shift = vecs + 2 * g.n_pad;
    a.Y = Y;
    a.x_rows = x_rows;
    a.ldx = ldx;
    a.ldy = g.n_pad;
    a.n_taps = g.n_taps;
    a.tap_rows = g.tap_rows;
    a.kpt = g.kpt;
    a.cpt = g.kpt_pad / (in16 ? 2 * kBK : kBK);
    a.k_pad = g.k_pad;
    a.n_tiles = g.n_pad / 128;
    a.groups_total = (rows_out + 31) / 32;
    {
        int64_t per_col = (int64_t)h->num_cu * h->blocks_per_cu / a.n_tiles;
        if (per_col < 1) per_col = 1;
        if (per_col > a.groups_total) per_col = a.groups_total;
        a.blocks_per_col = (int)per_col;
        // CU-pair-aware range sizes need the full 2-blocks-per-CU grid and whole XCD runs per half
        const int nwg = a.blocks_per_col * a.n_tiles;
        const bool ok = h->blocks_per_cu == 2 && nwg == 2 * h->num_cu && nwg % 16 == 0 &&
                        (nwg / 8) % (2 * a.n_tiles) == 0;
        a.pair_period = ok ? (nwg / 8) / a.n_tiles : 0;
    }
    a.pool_part = part;
    a.pool_cnt = part_cnt;
    a.out_map = out_map;
    a.span = h->geo[layer].ctx_span;
    a.terms = 1;
    {
        // the kernel addresses its input with 32-bit byte offsets from a per-tile descriptor: the
        // largest is (rows of a tile + look-ahead + u*span re-basing) * row bytes (+ the lo plane)
        const int64_t es = (in16 && v != TdnnVariant::kBf16FirstSrc32) ? 2 : 4;   // element size of the rows READ
        const int64_t reach = ((int64_t)out_map.n_utts * a.span + kRowPadTail) * ldx * es + (x3 ? x_plane : 0);
        if (reach > 0x7fffffff)
            return fail(XVEC_ERR_TOO_LARGE, "layer %d: %d utterances x %d channels exceed 32-bit row offsets; split the batch",
                        layer, out_map.n_utts, ldx);
    }
    if (x3) {
        if (x_plane > 0x3fffffff || y_plane > 0x3fffffff)
            return fail(XVEC_ERR_TOO_LARGE, "batch too large for bf16x3 (plane offsets must fit 30 bits); split it");
        a.terms = 2;
        a.Wf = h->Wp48[layer];
        a.k_pad = 2 * g.k_pad;
        a.x_plane_bytes = (int)x_plane;
        a.y_plane_bytes = (int)y_plane;
        a.x_bytes = x_rows > 0 ? x_plane + x_rows * (int64_t)ldx * 2 : 0;
        if (v == TdnnVariant::kBf16FirstSrc32) a.x_bytes = 0;       // the caller's fp32 rows (tdnn_first3): x_rows * ldx * 4
    }
    StageTimer t(h, T_L1 + layer, s);
    // bf16, wide layers, enough rows to give every CU about two 64-frame units: the 256-channel
    // ping-pong mapping (tdnn_pp16.hip; bf16x3: the same kernel over three K-tiles per 64-channel slab); everything else
    // (small batches, layer 1, narrow models, fp32) runs the 128x128 kernel
    if (layer > 0 && (v == TdnnVariant::kBf16 || v == TdnnVariant::kBf16Pool)) {
        if (const int bpc = pp_blocks_per_col(h, g.n_pad, rows_out)) {
            const int64_t units = (rows_out + 63) / 64;
            a.W = x3 ? h->Wr48[layer] : h->Wr16[layer];
            a.n_tiles = g.n_pad / 256;
            a.blocks_per_col = bpc;
            a.groups_total = units;
            a.pair_period = 0;
            d.pool_units = units;
            d.pool_bpc = bpc;
            HIP_TRY(launch_tdnn_pp16(a, v == TdnnVariant::kBf16Pool, s));
            h->last_kernel[layer] = d.kernel = XVEC_KERNEL_PP;
            return XVEC_OK;
        }
    }
    if (x3 && v == TdnnVariant::kBf16FirstSrc32) {      // bf16x3 layer 1 straight from the fp32 rows: only the streaming kernel does that
        if (!(h->use_pp && layer == 0 && tdnn_first3_applicable(a)))
            return fail(XVEC_ERR_STATE, "internal: bf16x3 layer 1 from fp32 rows needs the streaming kernel's shapes");
        HIP_TRY(launch_tdnn_first3(a, h->num_cu, s));
        h->last_kernel[layer] = d.kernel = XVEC_KERNEL_FIRST;
        return XVEC_OK;
    }
    if (h->use_pp && layer == 0 && v == TdnnVariant::kBf16FirstSrc32 && tdnn_first_applicable(a)) {
        a.Wf = h->Wp16b;
        HIP_TRY(launch_tdnn_first(a, h->num_cu, s));
        h->last_kernel[layer] = d.kernel = XVEC_KERNEL_FIRST;
        return XVEC_OK;
    }
    // pooling partials of the 128x128 kernels: one slot per (32-row group, utterance), addressed with 32-bit offsets
    // (tdnn_pp16.hip's segment partials take a 64-bit base per slot and have no such limit)
    if (part && (size_t)((rows_out + 31) / 32 + out_map.n_utts + 1) * 3 * g.n_pad * 4 > 0x7fffffffull)
        return fail(XVEC_ERR_TOO_LARGE, "batch too large: pooling partials exceed 2 GiB; split it");
    HIP_TRY(launch_tdnn(a, v, s));
    h->last_kernel[layer] = d.kernel = XVEC_KERNEL_TILE128;
    return XVEC_OK;
}

// (Re)build the plain-bf16 copies of one layer: weights scaled per input channel by the PRODUCING layer's folded BatchNorm
// scale, bias' = bias + W . shift_prev (reference order Linear -> ReLU -> BatchNorm, tdnn_layer.py:30-39: the BatchNorm of
// layer l is an affine map of layer l+1's input, and the valid convolutions have no padded frames where that would differ).
// Needs the layer itself and its producer loaded; xvec_load_tdnn calls it for the layer it loads and for its consumer.
int refold(xvec_handle* h, int layer, hipStream_t s) {
    if (layer < 0 || layer >= XVEC_NUM_TDNN || !h->tdnn_loaded[layer]) return XVEC_OK;
    if (layer > 0 && !h->tdnn_loaded[layer - 1]) { h->folded[layer] = false; return XVEC_OK; }
    // the raw weights of this layer and the folded BatchNorm of its producer may have been written by loads on other streams
    // (ADVICE r04): wait for their events -- no-ops on the stream that recorded them
    HIP_TRY(hipStreamWaitEvent(s, h->load_evt[layer], 0));
    if (layer > 0) HIP_TRY(hipStreamWaitEvent(s, h->load_evt[layer - 1], 0));
    const TdnnGeom& g = h->geo16[layer];
    const float* sc = layer > 0 ? h->vec[layer - 1] + h->geo[layer - 1].n_pad : nullptr;
    const float* sh = layer > 0 ? h->vec[layer - 1] + 2 * h->geo[layer - 1].n_pad : nullptr;
    HIP_TRY(launch_pack_tdnn_bf16(h->Wraw[layer], sc, g, h->Wp16[layer], s));
    HIP_TRY(launch_pack_tdnn_rows_bf16(h->Wraw[layer], sc, g, h->Wr16[layer], s));
    HIP_TRY(launch_fold_bias(h->Wraw[layer], h->braw[layer], sh, g, h->vec16[layer], s));
    if (layer == 0 && g.kpt + 2 <= g.k_pad) {      // the streaming kernel's copy: bias in the spare k slots
        HIP_TRY(launch_pack_tdnn_bf16(h->Wraw[layer], nullptr, g, h->Wp16b, s));
        HIP_TRY(launch_patch_bias_kslots(h->braw[layer], g, h->Wp16b, s));
    }
    h->folded[layer] = true;
    return XVEC_OK;
}

int check_loaded(const xvec_handle* h, int mode) {
    for (int i = 0; i < XVEC_NUM_TDNN; ++i)
        if (!h->tdnn_loaded[i]) return fail(XVEC_ERR_STATE, "time_context_layers.%d weights not loaded", i);
    for (int i = 0; i < XVEC_NUM_TDNN; ++i)
        if (!h->folded[i]) return fail(XVEC_ERR_STATE, "internal: time_context_layers.%d not folded", i);
    if (mode == XVEC_MODE_POOLED) return XVEC_OK;
    if (!h->aff_loaded[0]) return fail(XVEC_ERR_STATE, "segment_layer6 weights not loaded");
    if ((mode == XVEC_MODE_XVEC7 || mode == XVEC_MODE_LOGITS) && !h->aff_loaded[1])
        return fail(XVEC_ERR_STATE, "segment_layer7 weights not loaded");
    if (mode == XVEC_MODE_LOGITS && !h->aff_loaded[2]) return fail(XVEC_ERR_STATE, "output weights not loaded");
    return XVEC_OK;
}

// merge the pooling partials layer 5 left (in the form of the kernel that wrote them: `d`, from its run_tdnn) into pooled[B, 3000]
int finalize_pool(xvec_handle* h, const Dispatch& d, const float* part, const int* part_cnt, const RowMap& map, float* pooled,
                  hipStream_t s) {
    PoolFinalizeArgs f;
    memset(&f, 0, sizeof(f));
    f.part = part;
    f.out = pooled;
    f.map = map;
    f.C = XVEC_POOL_CHANNELS;
    f.n_pad = h->geo[4].n_pad;
    f.sub_rows = 32;
    f.scale = h->vec[4] + f.n_pad;              // the pooling epilogues leave sums of r = relu(z + bias)
    f.shift = h->vec[4] + 2 * f.n_pad;
    if (d.kernel == XVEC_KERNEL_PP) {               // tdnn_pp16.hip: one partial per (block, utterance, half)
        f.cnt = part_cnt;
        f.units_total = d.pool_units;
        f.blocks_per_col = d.pool_bpc;
    }
    HIP_TRY(launch_pool_finalize(f, s));
    return XVEC_OK;
}

// bf16x3 layer 1 can read the caller's fp32 rows itself (tdnn_first3: no hi/lo split pass) when the streaming kernel's
// shapes hold (the reference's 24 MFCCs x 5 frames -> 512 channels do); run_tdnn checks the same through tdnn_first3_applicable
bool first3_ok(const xvec_handle* h, const void* x_rows, int ldx) {
    const TdnnGeom& g = h->geo16[0];
    return h->use_pp && g.n_pad == 512 && g.k_pad == 128 && g.n_taps == 1 && g.kpt <= 128 && (ldx * 4) % 16 == 0 &&
           (reinterpret_cast<uintptr_t>(x_rows) & 15) == 0;
}

// x_rows: [total, ldx] packed rows (offs_host == nullptr: B utterances of fixed_T rows each)
int forward_rows(xvec_handle* h, const float* x_rows, int ldx, const int64_t* offs_dev, int B, int fixed_T,
                 const Plan& p, int mode, int dtype, float* out, char* ws, hipStream_t s) {
    float* actA = reinterpret_cast<float*>(ws + p.actA);
    float* actB = reinterpret_cast<float*>(ws + p.actB);
    float* part = reinterpret_cast<float*>(ws + p.part);
    int* part_cnt = reinterpret_cast<int*>(ws + p.part_cnt);
    float* pooled = mode == XVEC_MODE_POOLED ? out : reinterpret_cast<float*>(ws + p.pooled);
    float* s6 = reinterpret_cast<float*>(ws + p.seg6);
    float* s7 = reinterpret_cast<float*>(ws + p.seg7);
    const int nh = h->geo[0].n_pad;
    int rc;
    const bool x3 = dtype == XVEC_BF16X3;
    const bool b16 = dtype == XVEC_BF16 || x3;
    // layer 1 reads the caller's rows (guarded against the end of the buffer and the K tail); in
    // bf16 mode the MFCC rows are first rounded to bf16 into the workspace (bf16x3: split into a hi
    // and a lo plane; every activation buffer then holds two bf16 planes in the space of one fp32).
    // Layer 5 carries the statistics-pooling epilogue: its [frames,1500] output stays on chip.
    // (plain bf16: layer 1 reads the fp32 rows itself and rounds them in its staging path; bf16x3 too when the
    // streaming kernel's shapes hold -- first3_ok -- and otherwise needs the hi/lo split pass)
    const bool first3 = x3 && first3_ok(h, x_rows, ldx);
    const TdnnVariant v1 = x3 && !first3 ? TdnnVariant::kBf16First : b16 ? TdnnVariant::kBf16FirstSrc32 : TdnnVariant::kF32First;
    const TdnnVariant vm = b16 ? TdnnVariant::kBf16 : TdnnVariant::kF32;
    const TdnnVariant v5 = b16 ? TdnnVariant::kBf16Pool : TdnnVariant::kF32Pool;
    // every layer's output is compact: utterance u keeps len_u - cum frames after `cum` frames of
    // context have been consumed (4, 8, 14, 14, 14 after layers 1..5)
    RowMap map;
    map.offsets = offs_dev;
    map.n_utts = B;
    map.fixed_T = fixed_T;
    map.cum = 0;
    void* bufs[2] = {actA, actB};
    const void* in = x_rows;
    int ld_in = ldx;
    const int64_t act_plane = p.rows_alloc * (int64_t)nh * 2;        // bytes of one bf16 plane of an activation buffer
    int64_t in_plane = 0;
    if (b16 && p.total > 0x7fffffff) return fail(XVEC_ERR_TOO_LARGE, "too many frames for one bf16 batch; split it");
    if (x3 && !first3) {    // [total, ldx] fp32 -> bf16 hi and lo planes (same row stride in elements)
        StageTimer t(h, T_PACK, s);
        void* x16 = ws + p.x16;
        in_plane = p.rows_alloc * (int64_t)ldx * 2;
        HIP_TRY(launch_pack_rows_split(x_rows, p.total, ldx, ldx, in_plane / 2, x16, s));
        in = x16;
    }
    Dispatch d5;                                                     // what layer 5 went to: finalize_pool reads its partials
    for (int l = 0; l < XVEC_NUM_TDNN; ++l) {
        map.cum += h->geo[l].ctx_span;
        const int64_t rows_out = p.total - (int64_t)B * map.cum;
        const TdnnVariant v = l == 0 ? v1 : l == 4 ? v5 : vm;
        void* out_buf = l == 4 ? nullptr : bufs[l & 1];
        if ((rc = run_tdnn(h, l, v, in, ld_in, l == 0 ? p.total : 0, out_buf, rows_out, map, l == 4 ? part : nullptr, s,
                           x3, in_plane, l == 4 ? 0 : act_plane, l == 4 ? part_cnt : nullptr, l == 4 ? &d5 : nullptr)))
            return rc;
        in = out_buf;
        ld_in = nh;
        in_plane = act_plane;
    }
    {
        StageTimer t(h, T_POOL, s);
        if ((rc = finalize_pool(h, d5, part, part_cnt, map, pooled, s))) return rc;
    }
    if (mode == XVEC_MODE_POOLED) return XVEC_OK;
    const int xv = h->cfg.x_vector_size, K6 = 2 * XVEC_POOL_CHANNELS;
    // plain bf16 (parity bar 1e-2): the segment layers' products as bf16x3 (affine.hip); XVEC_BF16X3 promises the fp32
    // bar end to end and keeps the fp32 MFMAs here (x3 segment layers measured 3x the fp32 ones' error: 6e-6)
    const bool w3 = b16 && !x3;
    // the frame-level activations are dead from here on: layers 1-4's buffers serve as split-K scratch
    float* scr = actA;
    const size_t scr_bytes = (p.actB - p.actA) * 2;                 // actA and actB are adjacent
    if (mode == XVEC_MODE_XVEC6) {
        StageTimer t(h, T_SEG6, s);
        HIP_TRY(launch_affine_f32(pooled, h->affW[0], h->affB[0], out, B, xv, K6, 0, s, scr, scr_bytes, w3 ? h->affW3[0] : nullptr));
        return XVEC_OK;
    }
    {
        StageTimer t(h, T_SEG6, s);
        HIP_TRY(launch_affine_f32(pooled, h->affW[0], h->affB[0], s6, B, xv, K6, 1, s, scr, scr_bytes, w3 ? h->affW3[0] : nullptr));
    }
    if (mode == XVEC_MODE_XVEC7) {
        StageTimer t(h, T_SEG7, s);
        HIP_TRY(launch_affine_f32(s6, h->affW[1], h->affB[1], out, B, xv, xv, 0, s, scr, scr_bytes, w3 ? h->affW3[1] : nullptr));
        return XVEC_OK;
    }
    {
        StageTimer t(h, T_SEG7, s);
        HIP_TRY(launch_affine_f32(s6, h->affW[1], h->affB[1], s7, B, xv, xv, 1, s, scr, scr_bytes, w3 ? h->affW3[1] : nullptr));
    }
    {
        StageTimer t(h, T_OUT, s);
        HIP_TRY(launch_affine_f32(s7, h->affW[2], h->affB[2], out, B, h->cfg.num_classes, xv, 0, s, scr, scr_bytes, w3 ? h->affW3[2] : nullptr));
    }
    return XVEC_OK;
}

// host offsets -> device (stream ordered, via the handle's ring of two pinned staging slots).
// Synchronous part: only when both slots still hold copies that have not left them (a third ragged
// call enqueued before the first one's copy ran) does this wait for the older one.
int stage_offsets(xvec_handle* h, const int64_t* offs_host, const int32_t* lengths_host, int B, int64_t* dev,
                  hipStream_t s) {
    const size_t n = (size_t)B + 1;
    const int slot = h->offs_next;
    h->offs_next ^= 1;
    if (h->offs_pending[slot]) {
        HIP_TRY(hipEventSynchronize(h->offs_evt[slot]));
        h->offs_pending[slot] = false;
    }
    int64_t* pin = h->offs_pinned[slot];        // free: its last copy has left it
    if (offs_host) {
        memcpy(pin, offs_host, n * 8);
    } else {                                    // prefix sum of the lengths, built in place (no host allocation)
        pin[0] = 0;
        for (int i = 0; i < B; ++i) pin[i + 1] = pin[i] + lengths_host[i];
    }
    HIP_TRY(hipMemcpyAsync(dev, pin, n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(h->offs_evt[slot], s));
    h->offs_pending[slot] = true;
    return XVEC_OK;
}

int common_checks(xvec_handle* h, const void* x, int B, int mode, int dtype, const void* out, const void* ws) {
    if (!h) return fail(XVEC_ERR_ARG, "null handle");
    if (!x || !out || !ws) return fail(XVEC_ERR_ARG, "null tensor pointer");
    if (B < 1) return fail(XVEC_ERR_ARG, "B must be in [1, %d] (got %d)", kMaxUtts, B);
    if (B > kMaxUtts) return fail(XVEC_ERR_TOO_LARGE, "B must be in [1, %d] (got %d); split larger batches", kMaxUtts, B);
    if (mode != XVEC_MODE_LOGITS && mode != XVEC_MODE_XVEC6 && mode != XVEC_MODE_XVEC7 && mode != XVEC_MODE_POOLED)
        return fail(XVEC_ERR_ARG, "unknown mode %d", mode);
    if (dtype != XVEC_F32 && dtype != XVEC_BF16 && dtype != XVEC_BF16X3) return fail(XVEC_ERR_ARG, "unknown dtype %d", dtype);
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(ws) & 255))
        return fail(XVEC_ERR_ARG, "x must be 16-byte and workspace 256-byte aligned");
    return check_loaded(h, mode);
}

}  // namespace

extern "C" {

const char* xvec_last_error(void) { return g_err; }
#ifndef XVEC_BUILD_ID
#define XVEC_BUILD_ID "unknown"
#endif
const char* xvec_version(void) { return "xvec_hip gfx950 build " XVEC_BUILD_ID; }

int xvec_create(const xvec_cfg* cfg, xvec_handle** out) {
    if (!cfg || !out) return fail(XVEC_ERR_ARG, "null argument");
    if (cfg->input_size < 1 || cfg->hidden_size < 1 || cfg->num_classes < 1 || cfg->x_vector_size < 1)
        return fail(XVEC_ERR_ARG, "sizes must be positive");
    if (cfg->hidden_size > 8192 || cfg->input_size > 8192)
        return fail(XVEC_ERR_ARG, "input_size / hidden_size above 8192 are not supported (32-bit row offsets)");
    DeviceGuard guard;                 // allocate on cfg->device, leave the caller's current device as it was
    HIP_TRY(guard.enter(cfg->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(XVEC_ERR_STATE, "device %d is %s; this library is built for gfx950 only", cfg->device,
                    prop.gcnArchName);
    xvec_handle* h = new (std::nothrow) xvec_handle();
    if (!h) return fail(XVEC_ERR_STATE, "out of host memory");
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->num_cu = prop.multiProcessorCount;
    {
        const char* e = getenv("XVEC_BLOCKS_PER_CU");   // diagnostic knob (profiles/ab_env.sh); 2 = what the LDS allows
        h->blocks_per_cu = e ? atoi(e) : 2;
        if (h->blocks_per_cu < 1) h->blocks_per_cu = 1;
        const char* p = getenv("XVEC_PP");
        h->use_pp = !(p && atoi(p) == 0);
        const char* mt = getenv("XVEC_PP_MIN_TENTHS");
        h->pp_min_tenths = mt && atoi(mt) > 0 ? atoi(mt) : 18;
        const char* cp = getenv("XVEC_PP_CU_PCT");
        h->pp_cu_pct = cp ? atoi(cp) : 0;
    }
    h->cin_pad = round_up(cfg->input_size, 4);
    fill_geometry(h, h->geo, 2 * kBK);
    fill_geometry(h, h->geo16, 4 * kBK);
    for (int i = 0; i < XVEC_NUM_TDNN; ++i) {
        const TdnnGeom& g = h->geo[i];
        if (hipMalloc(&h->Wp16[i], (size_t)h->geo16[i].n_pad * h->geo16[i].k_pad * 2) != hipSuccess ||
            hipMalloc(&h->Wr16[i], (size_t)h->geo16[i].n_pad * h->geo16[i].k_pad * 2) != hipSuccess ||
            (i == 0 && hipMalloc(&h->Wp16b, (size_t)h->geo16[i].n_pad * h->geo16[i].k_pad * 2) != hipSuccess) ||
            hipMalloc(&h->Wp48[i], (size_t)h->geo16[i].n_pad * h->geo16[i].k_pad * 2 * 2) != hipSuccess ||
            hipMalloc(&h->Wr48[i], (size_t)h->geo16[i].n_pad * h->geo16[i].k_pad * 2 * 3) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&h->Wp[i]), (size_t)g.n_pad * g.k_pad * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&h->Wraw[i]), (size_t)g.cout * g.src_taps * g.src_cin * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&h->braw[i]), (size_t)g.cout * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&h->vec16[i]), (size_t)3 * g.n_pad * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&h->vec[i]), (size_t)3 * g.n_pad * 4) != hipSuccess) {
            xvec_destroy(h);
            return fail(XVEC_ERR_HIP, "hipMalloc of packed weights failed");
        }
    }
    const int N[3] = {cfg->x_vector_size, cfg->x_vector_size, cfg->num_classes};
    const int K[3] = {2 * XVEC_POOL_CHANNELS, cfg->x_vector_size, cfg->x_vector_size};
    for (int i = 0; i < 3; ++i) {
        h->affN[i] = N[i];
        h->affK[i] = K[i];
        if (hipMalloc(reinterpret_cast<void**>(&h->affW[i]), (size_t)N[i] * K[i] * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&h->affB[i]), (size_t)N[i] * 4) != hipSuccess ||
            (K[i] % 4 == 0 && hipMalloc(&h->affW3[i], (size_t)N[i] * K[i] * 4) != hipSuccess)) {
            xvec_destroy(h);
            return fail(XVEC_ERR_HIP, "hipMalloc of affine weights failed");
        }
    }
    bool ok = true;
    for (int i = 0; i < 2 && ok; ++i)
        ok = hipEventCreateWithFlags(&h->offs_evt[i], hipEventDisableTiming) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void**>(&h->offs_pinned[i]), (size_t)(kMaxUtts + 1) * 8,
                           hipHostMallocDefault) == hipSuccess;
    for (int i = 0; i < T_COUNT && ok; ++i)
        ok = hipEventCreate(&h->ev0[i]) == hipSuccess && hipEventCreate(&h->ev1[i]) == hipSuccess;
    for (int i = 0; i < XVEC_NUM_TDNN && ok; ++i)
        ok = hipEventCreateWithFlags(&h->load_evt[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        xvec_destroy(h);
        return fail(XVEC_ERR_HIP, "hipEventCreate / hipHostMalloc failed");
    }
    *out = h;
    return XVEC_OK;
}

void xvec_destroy(xvec_handle* h) {
    if (!h) return;
    for (int i = 0; i < XVEC_NUM_TDNN; ++i) {
        if (h->Wp[i]) (void)hipFree(h->Wp[i]);
        if (h->Wp16[i]) (void)hipFree(h->Wp16[i]);
        if (h->Wr16[i]) (void)hipFree(h->Wr16[i]);
        if (h->Wp48[i]) (void)hipFree(h->Wp48[i]);
        if (h->Wr48[i]) (void)hipFree(h->Wr48[i]);
        if (h->vec[i]) (void)hipFree(h->vec[i]);
        if (h->Wraw[i]) (void)hipFree(h->Wraw[i]);
        if (h->braw[i]) (void)hipFree(h->braw[i]);
        if (h->vec16[i]) (void)hipFree(h->vec16[i]);
    }
    if (h->Wp16b) (void)hipFree(h->Wp16b);
    for (int i = 0; i < 3; ++i) {
        if (h->affW[i]) (void)hipFree(h->affW[i]);
        if (h->affB[i]) (void)hipFree(h->affB[i]);
        if (h->affW3[i]) (void)hipFree(h->affW3[i]);
    }
    for (int i = 0; i < 2; ++i) {
        if (h->offs_pinned[i]) (void)hipHostFree(h->offs_pinned[i]);
        if (h->offs_evt[i]) (void)hipEventDestroy(h->offs_evt[i]);
    }
    for (int i = 0; i < XVEC_NUM_TDNN; ++i)
        if (h->load_evt[i]) (void)hipEventDestroy(h->load_evt[i]);
    for (int i = 0; i < T_COUNT; ++i) {
        if (h->ev0[i]) (void)hipEventDestroy(h->ev0[i]);
        if (h->ev1[i]) (void)hipEventDestroy(h->ev1[i]);
    }
    delete h;
}

int xvec_load_tdnn(xvec_handle* h, int layer, const float* weight, const float* bias, const float* bn_weight,
                   const float* bn_bias, const float* bn_mean, const float* bn_var, float eps, xvec_stream stream) {
    if (!h || layer < 0 || layer >= XVEC_NUM_TDNN) return fail(XVEC_ERR_ARG, "bad handle or layer %d", layer);
    if (!weight || !bias) return fail(XVEC_ERR_ARG, "null weight/bias");
    DeviceGuard guard;                 // launches go to the handle's device whatever the caller's current one is
    HIP_TRY(guard.enter(h->cfg.device));
    const bool has_bn = bn_weight && bn_bias && bn_mean && bn_var;
    const bool none_bn = !bn_weight && !bn_bias && !bn_mean && !bn_var;
    if (h->cfg.batch_norm ? !has_bn : !none_bn)
        return fail(XVEC_ERR_ARG, "BatchNorm tensors must be %s for batch_norm=%d",
                    h->cfg.batch_norm ? "all given" : "all NULL", h->cfg.batch_norm);
    const TdnnGeom& g = h->geo[layer];
    // A RE-load of this layer while a neighbour's load is still in flight on another stream: that load's re-folds read
    // Wraw / vec of this layer (refold(layer) reads the producer's BatchNorm, refold(layer + 1) this layer's), which the writes
    // below replace -- they wait for the neighbours' loads first (a first load has nothing to wait for).
    if (layer > 0 && h->tdnn_loaded[layer - 1]) HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->load_evt[layer - 1], 0));
    if (layer + 1 < XVEC_NUM_TDNN && h->tdnn_loaded[layer + 1])
        HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->load_evt[layer + 1], 0));
    HIP_TRY(launch_pack_tdnn(weight, bias, bn_weight, bn_bias, bn_mean, bn_var, eps, g, h->Wp[layer],
                             h->vec[layer], h->vec[layer] + g.n_pad, h->vec[layer] + 2 * g.n_pad,
                             static_cast<hipStream_t>(stream)));
    HIP_TRY(launch_pack_tdnn_rows_bf16x3(weight, h->geo16[layer], h->Wr48[layer], static_cast<hipStream_t>(stream)));
    {
        TdnnGeom g3 = h->geo16[layer];
        g3.terms = 2;
        HIP_TRY(launch_pack_tdnn_bf16(weight, nullptr, g3, h->Wp48[layer], static_cast<hipStream_t>(stream)));
    }
    // plain bf16 (BatchNorm deferred into the consumer): this layer's copies depend on its producer's BatchNorm, its
    // consumer's on this layer's -- whatever the order the layers are loaded in
    HIP_TRY(hipMemcpyAsync(h->Wraw[layer], weight, (size_t)g.cout * g.src_taps * g.src_cin * 4, hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    HIP_TRY(hipMemcpyAsync(h->braw[layer], bias, (size_t)g.cout * 4, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    h->tdnn_loaded[layer] = true;
    HIP_TRY(hipEventRecord(h->load_evt[layer], static_cast<hipStream_t>(stream)));
    if (int rc = refold(h, layer, static_cast<hipStream_t>(stream))) return rc;
    if (int rc = refold(h, layer + 1, static_cast<hipStream_t>(stream))) return rc;
    // (the re-folds above read what this call wrote: later loads of the neighbours on other streams wait for them too)
    HIP_TRY(hipEventRecord(h->load_evt[layer], static_cast<hipStream_t>(stream)));
    return XVEC_OK;
}

int xvec_load_affine(xvec_handle* h, int which, const float* weight, const float* bias, xvec_stream stream) {
    const int i = aff_index(which);
    if (!h || i < 0) return fail(XVEC_ERR_ARG, "bad handle or affine id %d", which);
    if (!weight || !bias) return fail(XVEC_ERR_ARG, "null weight/bias");
    DeviceGuard guard;                 // launches go to the handle's device whatever the caller's current one is
    HIP_TRY(guard.enter(h->cfg.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(h->affW[i], weight, (size_t)h->affN[i] * h->affK[i] * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->affB[i], bias, (size_t)h->affN[i] * 4, hipMemcpyDeviceToDevice, s));
    if (h->affW3[i]) HIP_TRY(launch_split_pairs(h->affW[i], h->affW3[i], (int64_t)h->affN[i] * h->affK[i], s));
    h->aff_loaded[i] = true;
    return XVEC_OK;
}

size_t xvec_workspace_bytes(const xvec_handle* h, int64_t total_frames, int32_t n_utts) {
    if (!h || total_frames < 1 || n_utts < 1) return 0;
    return make_plan(h, total_frames, n_utts).bytes;
}

int xvec_workspace_layout(const xvec_handle* h, int64_t total_frames, int32_t n_utts, xvec_ws_layout* out) {
    if (!h || !out || total_frames < 1 || n_utts < 1) return fail(XVEC_ERR_ARG, "bad argument");
    const Plan p = make_plan(h, total_frames, n_utts);
    out->act_a = p.actA;
    out->act_b = p.actB;
    out->part = p.part;
    out->part_cnt = p.part_cnt;
    out->pooled = p.pooled;
    out->bytes = p.bytes;
    out->rows_alloc = p.rows_alloc;
    out->part_slots = p.part_slots;
    out->pool_n_pad = h->geo[4].n_pad;
    out->hidden_n_pad = h->geo[0].n_pad;
    out->num_cu = h->num_cu;
    return XVEC_OK;
}

int xvec_forward(xvec_handle* h, const float* x, const int32_t* lengths_host, int32_t B, int32_t T, int mode,
                 int dtype, float* out, void* workspace, size_t workspace_bytes, xvec_stream stream) {
    int rc = common_checks(h, x, B, mode, dtype, out, workspace);
    if (rc) return rc;
    DeviceGuard guard;                 // launches go to the handle's device whatever the caller's current one is
    HIP_TRY(guard.enter(h->cfg.device));
    if (T <= XVEC_TOTAL_CONTEXT)
        return fail(XVEC_ERR_ARG, "T=%d: need at least %d frames (receptive field of the TDNN stack)", T,
                    XVEC_TOTAL_CONTEXT + 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    const int C = h->cfg.input_size;

    if (!lengths_host) {
        const Plan p = make_plan(h, (int64_t)B * T, B);
        if (workspace_bytes < p.bytes)
            return fail(XVEC_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, p.bytes);
        const float* rows = x;
        if (h->cin_pad != C) {   // channel count not a multiple of 4: pad rows once
            StageTimer t(h, T_PACK, s);
            float* xp = reinterpret_cast<float*>(ws + p.xpad);
            HIP_TRY(launch_pack_rows(x, nullptr, B, T, C, h->cin_pad, xp, false, s));
            rows = xp;
        }
        return forward_rows(h, rows, h->cin_pad, nullptr, B, T, p, mode, dtype, out, ws, s);
    }

    // ragged: pack the valid frames, run the stack on sum(lengths) rows only
    int64_t total = 0;
    for (int i = 0; i < B; ++i) {
        const int n = lengths_host[i];
        if (n <= XVEC_TOTAL_CONTEXT || n > T)
            return fail(XVEC_ERR_ARG, "lengths[%d]=%d outside [%d, T=%d]", i, n, XVEC_TOTAL_CONTEXT + 1, T);
        total += n;
    }
    const Plan p = make_plan(h, total, B);
    if (workspace_bytes < p.bytes)
        return fail(XVEC_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, p.bytes);
    if ((rc = stage_offsets(h, nullptr, lengths_host, B, reinterpret_cast<int64_t*>(ws + p.offs), s))) return rc;
    const int64_t* offs_dev = reinterpret_cast<const int64_t*>(ws + p.offs);
    float* xp = reinterpret_cast<float*>(ws + p.xpad);
    {
        StageTimer t(h, T_PACK, s);
        HIP_TRY(launch_pack_rows(x, offs_dev, B, T, C, h->cin_pad, xp, false, s));
    }
    return forward_rows(h, xp, h->cin_pad, offs_dev, B, 0, p, mode, dtype, out, ws, s);
}

int xvec_forward_packed(xvec_handle* h, const float* x_packed, const int64_t* offsets_host, int32_t B, int mode,
                        int dtype, float* out, void* workspace, size_t workspace_bytes, xvec_stream stream) {
    int rc = common_checks(h, x_packed, B, mode, dtype, out, workspace);
    if (rc) return rc;
    DeviceGuard guard;                 // launches go to the handle's device whatever the caller's current one is
    HIP_TRY(guard.enter(h->cfg.device));
    if (!offsets_host) return fail(XVEC_ERR_ARG, "null offsets");
    if (offsets_host[0] != 0) return fail(XVEC_ERR_ARG, "offsets[0] must be 0");
    for (int i = 0; i < B; ++i) {
        const int64_t n = offsets_host[i + 1] - offsets_host[i];
        if (n <= XVEC_TOTAL_CONTEXT)
            return fail(XVEC_ERR_ARG, "utterance %d has %lld frames; need at least %d", i, (long long)n,
                        XVEC_TOTAL_CONTEXT + 1);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    const int64_t total = offsets_host[B];
    const Plan p = make_plan(h, total, B);
    if (workspace_bytes < p.bytes)
        return fail(XVEC_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, p.bytes);
    if ((rc = stage_offsets(h, offsets_host, nullptr, B, reinterpret_cast<int64_t*>(ws + p.offs), s))) return rc;
    const int64_t* offs_dev = reinterpret_cast<const int64_t*>(ws + p.offs);
    const float* rows = x_packed;
    const int C = h->cfg.input_size;
    if (h->cin_pad != C) {
        if (total > 0x7fffffff) return fail(XVEC_ERR_ARG, "too many frames");
        StageTimer t(h, T_PACK, s);
        float* xp = reinterpret_cast<float*>(ws + p.xpad);
        HIP_TRY(launch_pack_rows(x_packed, nullptr, 1, (int)total, C, h->cin_pad, xp, false, s));
        rows = xp;
    }
    return forward_rows(h, rows, h->cin_pad, offs_dev, B, 0, p, mode, dtype, out, ws, s);
}

int xvec_tdnn_layer(xvec_handle* h, int layer, const float* x, int32_t B, int32_t T, int dtype, float* y,
                    void* workspace, size_t workspace_bytes, xvec_stream stream) {
    if (!h || layer < 0 || layer >= XVEC_NUM_TDNN) return fail(XVEC_ERR_ARG, "bad handle or layer %d", layer);
    if (!x || !y || !workspace) return fail(XVEC_ERR_ARG, "null tensor pointer");
    if (dtype != XVEC_F32 && dtype != XVEC_BF16 && dtype != XVEC_BF16X3) return fail(XVEC_ERR_ARG, "unknown dtype %d", dtype);
    if (!h->tdnn_loaded[layer]) return fail(XVEC_ERR_STATE, "time_context_layers.%d weights not loaded", layer);
    if (dtype == XVEC_BF16 && !h->folded[layer])
        return fail(XVEC_ERR_STATE, "time_context_layers.%d: plain bf16 folds the BatchNorm of layer %d into it; load that layer too", layer, layer - 1);
    DeviceGuard guard;                 // launches go to the handle's device whatever the caller's current one is
    HIP_TRY(guard.enter(h->cfg.device));
    const TdnnGeom& g = h->geo[layer];
    if (B < 1 || T <= g.ctx_span) return fail(XVEC_ERR_ARG, "need B>=1 and T>%d (got B=%d T=%d)", g.ctx_span, B, T);
    const Plan p = make_plan(h, (int64_t)B * T, B);
    if (workspace_bytes < p.bytes)
        return fail(XVEC_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, p.bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    // bf16 mode: activations between layers are bf16 (layer 1 still reads fp32 MFCCs); the result is
    // widened back to fp32 for the caller
    const bool x3 = dtype == XVEC_BF16X3;
    const bool b16 = dtype == XVEC_BF16;
    const bool in16 = b16 || x3;
    // stage the compact input into the layer's native row layout (stride = producer's n_pad)
    const int ldx = (layer == 0) ? h->cin_pad : h->geo[layer - 1].n_pad;
    // (plain bf16 layer 1 reads fp32 rows and rounds them itself, exactly as in xvec_forward)
    const bool rows16 = in16 && !(b16 && layer == 0);
    void* xin = ws + (layer == 0 ? (rows16 ? p.x16 : p.xpad) : p.actA);
    int64_t x_plane = 0;
    const bool l0_first3 = x3 && layer == 0 && first3_ok(h, ws + p.actB, ldx);   // layer 1 as xvec_forward runs it (tdnn_first3)
    if (x3) {   // fp32 rows first (padding to ldx), then the hi/lo split; bf16x3 returns fp32 directly
        void* x32 = ws + p.actB;
        HIP_TRY(launch_pack_rows(x, nullptr, B, T, g.src_cin, ldx, x32, false, s));
        x_plane = p.rows_alloc * (int64_t)ldx * 2;
        if (!l0_first3) HIP_TRY(launch_pack_rows_split(static_cast<const float*>(x32), p.total, ldx, ldx, x_plane / 2, xin, s));
    } else {
        // plain bf16, layers 2-5: the kernel reads the producing layer's relu output and carries that layer's BatchNorm in its
        // weights (refold), so the caller's x -- the reference's layer input, BatchNorm applied -- is taken back through it
        const float* un = (b16 && layer > 0) ? h->vec[layer - 1] : nullptr;
        const int npp = layer > 0 ? h->geo[layer - 1].n_pad : 0;
        HIP_TRY(launch_pack_rows(x, nullptr, B, T, g.src_cin, ldx, xin, rows16, s, un ? un + npp : nullptr, un ? un + 2 * npp : nullptr));
    }
    void* yflat = ws + (layer == 4 || x3 ? p.act5 : p.actB);
    const TdnnVariant v = layer == 0 ? (b16 ? TdnnVariant::kBf16FirstSrc32 : x3 ? TdnnVariant::kBf16FirstToF32 : TdnnVariant::kF32First)
                                     : (b16 ? TdnnVariant::kBf16 : x3 ? TdnnVariant::kBf16ToF32 : TdnnVariant::kF32);
    RowMap map;
    map.offsets = nullptr;
    map.n_utts = B;
    map.fixed_T = T;
    map.cum = g.ctx_span;
    const int To = T - g.ctx_span;
    if (l0_first3) {    // fp32 rows in, the two bf16 planes out, joined for the caller
        const int64_t y_plane = p.rows_alloc * (int64_t)g.n_pad * 2;
        void* y16 = ws + p.actA;
        int rc = run_tdnn(h, layer, TdnnVariant::kBf16FirstSrc32, ws + p.actB, ldx, p.total, y16, (int64_t)B * To, map, nullptr,
                          s, true, 0, y_plane);
        if (rc) return rc;
        HIP_TRY(launch_unpack_rows_split(y16, y_plane / 2, g.n_pad, B, To, To, g.cout, y, s));
        return XVEC_OK;
    }
    // bf16x3, layers 2-4 at the sizes xvec_forward gives to the large-batch kernel: that kernel, with its output as
    // the two bf16 planes the next layer would read, joined (hi + lo) for the caller -- so that the per-layer entry runs
    // what the whole path runs; otherwise the 128x128 kernel writes fp32 directly
    if (x3 && layer > 0 && layer < XVEC_NUM_TDNN - 1) {
        if (pp_blocks_per_col(h, h->geo16[layer].n_pad, (int64_t)B * To)) {
            const int64_t y_plane = p.rows_alloc * (int64_t)g.n_pad * 2;
            void* y16 = ws + p.actB;
            int rc = run_tdnn(h, layer, TdnnVariant::kBf16, xin, ldx, p.total, y16, (int64_t)B * To, map, nullptr, s, true,
                              x_plane, y_plane);
            if (rc) return rc;
            HIP_TRY(launch_unpack_rows_split(y16, y_plane / 2, g.n_pad, B, To, To, g.cout, y, s));
            return XVEC_OK;
        }
    }
    int rc = run_tdnn(h, layer, v, xin, ldx, p.total, yflat, (int64_t)B * To, map, nullptr, s, x3, x_plane, 0);
    if (rc) return rc;
    // (plain bf16 stored relu(z + bias'): this layer's BatchNorm is applied here, in fp32, as its consumer's weights would)
    HIP_TRY(launch_unpack_rows(yflat, b16, g.n_pad, B, To, To, g.cout, y, s, b16 ? h->vec[layer] + g.n_pad : nullptr,
                               b16 ? h->vec[layer] + 2 * g.n_pad : nullptr));
    return XVEC_OK;
}

int xvec_tdnn_pool_layer(xvec_handle* h, const float* x, int32_t B, int32_t T, int dtype, float* out, void* workspace,
                         size_t workspace_bytes, xvec_stream stream) {
    const int layer = XVEC_NUM_TDNN - 1;
    if (!h) return fail(XVEC_ERR_ARG, "null handle");
    if (!x || !out || !workspace) return fail(XVEC_ERR_ARG, "null tensor pointer");
    if (dtype != XVEC_F32 && dtype != XVEC_BF16 && dtype != XVEC_BF16X3) return fail(XVEC_ERR_ARG, "unknown dtype %d", dtype);
    if (!h->tdnn_loaded[layer]) return fail(XVEC_ERR_STATE, "time_context_layers.%d weights not loaded", layer);
    if (dtype == XVEC_BF16 && !h->folded[layer])
        return fail(XVEC_ERR_STATE, "time_context_layers.%d: plain bf16 folds the BatchNorm of layer %d into it; load that layer too", layer, layer - 1);
    DeviceGuard guard;
    HIP_TRY(guard.enter(h->cfg.device));
    const TdnnGeom& g = h->geo[layer];
    if (B < 1 || B > kMaxUtts || T <= g.ctx_span) return fail(XVEC_ERR_ARG, "need 1<=B<=%d and T>%d (got B=%d T=%d)", kMaxUtts, g.ctx_span, B, T);
    const Plan p = make_plan(h, (int64_t)B * T, B);
    if (workspace_bytes < p.bytes)
        return fail(XVEC_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, p.bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    const bool x3 = dtype == XVEC_BF16X3, in16 = x3 || dtype == XVEC_BF16;
    // the input in the layer's native row layout (the producer's n_pad; bf16 / two bf16 planes as in xvec_forward)
    const int ldx = h->geo[layer - 1].n_pad;
    void* xin = ws + p.actA;
    int64_t x_plane = 0;
    if (x3) {
        void* x32 = ws + p.actB;
        HIP_TRY(launch_pack_rows(x, nullptr, B, T, g.src_cin, ldx, x32, false, s));
        x_plane = p.rows_alloc * (int64_t)ldx * 2;
        HIP_TRY(launch_pack_rows_split(static_cast<const float*>(x32), p.total, ldx, ldx, x_plane / 2, xin, s));
    } else {
        const float* un = dtype == XVEC_BF16 ? h->vec[layer - 1] : nullptr;     // plain bf16: see xvec_tdnn_layer
        const int npp = h->geo[layer - 1].n_pad;
        HIP_TRY(launch_pack_rows(x, nullptr, B, T, g.src_cin, ldx, xin, in16, s, un ? un + npp : nullptr, un ? un + 2 * npp : nullptr));
    }
    RowMap map;
    map.offsets = nullptr;
    map.n_utts = B;
    map.fixed_T = T;
    map.cum = g.ctx_span;
    float* part = reinterpret_cast<float*>(ws + p.part);
    int* part_cnt = reinterpret_cast<int*>(ws + p.part_cnt);
    Dispatch d5;
    int rc = run_tdnn(h, layer, in16 ? TdnnVariant::kBf16Pool : TdnnVariant::kF32Pool, xin, ldx, 0, nullptr,
                      (int64_t)B * (T - g.ctx_span), map, part, s, x3, x_plane, 0, part_cnt, &d5);
    if (rc) return rc;
    return finalize_pool(h, d5, part, part_cnt, map, out, s);
}

int xvec_stat_pool(const float* x, const int32_t* lengths_dev, int32_t B, int32_t T, int32_t C, float* out,
                   xvec_stream stream) {
    if (!x || !out) return fail(XVEC_ERR_ARG, "null tensor pointer");
    if (B < 1 || T < 1 || C < 1) return fail(XVEC_ERR_ARG, "B, T, C must be positive");
    if (B > 65535) return fail(XVEC_ERR_ARG, "B > 65535 not supported by the stand-alone pooling kernel");
    PoolArgs a;
    memset(&a, 0, sizeof(a));
    a.X = x;
    a.out = out;
    a.lengths = lengths_dev;
    a.B = B;
    a.T = T;
    a.C = C;
    HIP_TRY(launch_stat_pool(a, static_cast<hipStream_t>(stream)));
    return XVEC_OK;
}

int xvec_affine(xvec_handle* h, int which, const float* x, int32_t M, int relu, float* y, xvec_stream stream) {
    const int i = aff_index(which);
    if (!h || i < 0) return fail(XVEC_ERR_ARG, "bad handle or affine id %d", which);
    if (!x || !y || M < 1) return fail(XVEC_ERR_ARG, "bad x/y/M");
    if (!h->aff_loaded[i]) return fail(XVEC_ERR_STATE, "affine %d weights not loaded", which);
    DeviceGuard guard;                 // launches go to the handle's device whatever the caller's current one is
    HIP_TRY(guard.enter(h->cfg.device));
    HIP_TRY(launch_affine_f32(x, h->affW[i], h->affB[i], y, M, h->affN[i], h->affK[i], relu,
                              static_cast<hipStream_t>(stream)));
    return XVEC_OK;
}

int xvec_get_dispatch(const xvec_handle* h, int* kernels, int* n) {
    if (!h || !kernels || !n) return fail(XVEC_ERR_ARG, "null argument");
    for (int i = 0; i < XVEC_NUM_TDNN; ++i) kernels[i] = h->last_kernel[i];
    *n = XVEC_NUM_TDNN;
    return XVEC_OK;
}

int xvec_set_profiling(xvec_handle* h, int on) {
    if (!h) return fail(XVEC_ERR_ARG, "null handle");
    h->profiling = on != 0;
    for (int i = 0; i < T_COUNT; ++i) h->ev_used[i] = false;
    return XVEC_OK;
}

int xvec_get_timings(xvec_handle* h, float* ms, int* n) {
    if (!h || !ms || !n) return fail(XVEC_ERR_ARG, "null argument");
    for (int i = 0; i < T_COUNT; ++i) {
        ms[i] = 0.f;
        if (h->ev_used[i]) {
            HIP_TRY(hipEventSynchronize(h->ev1[i]));
            HIP_TRY(hipEventElapsedTime(&ms[i], h->ev0[i], h->ev1[i]));
        }
    }
    *n = T_COUNT;
    return XVEC_OK;
}

}  // extern "C"

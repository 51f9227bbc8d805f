/* Plain-C restatement of the reference's extraction path.  TEST INFRASTRUCTURE ONLY:
 * an independent checker (double-precision accumulation, naive loops) for the PyTorch
 * oracle in xvector_oracle.py and for the HIP kernels at small sizes.  Nothing in the
 * product links or calls this file.
 *
 * Follows, function by function:
 *   xvo_tdnn_layer  reference tdnn_layer.py:26-41 + get_time_context :43-60 (eval mode)
 *   xvo_stat_pool   reference main.py:59-63 (mean ‖ unbiased std)
 *   xvo_linear      nn.Linear (+ F.relu) as used in main.py:72-75,87-90
 * Parity pin: checked against the golden fixtures generated from the reference itself
 * (tests/golden/, tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stddef.h>

/* x[B,T,C] -> y[B,T-(ctx[n-1]-ctx[0]),N];  W[N, nctx*C] with column j*C+c (tap-major);
 * gamma == NULL means no BatchNorm. */
void xvo_tdnn_layer(const float* x, int B, int T, int C, const float* W, const float* b,
                    const float* gamma, const float* beta, const float* mean, const float* var,
                    float eps, const int* ctx, int nctx, int N, float* y) {
    const int span = ctx[nctx - 1] - ctx[0];
    const int To = T - span;
    for (int u = 0; u < B; ++u)
        for (int t = 0; t < To; ++t) {
            float* yo = y + ((size_t)u * To + t) * N;
#pragma omp parallel for
            for (int n = 0; n < N; ++n) {
                double acc = b[n];
                for (int j = 0; j < nctx; ++j) {
                    /* slice j of get_time_context starts at frame ctx[nctx-1]+ctx[j] ... but the
                     * reference's slices are taken relative to c[-1]: frame t of slice j is
                     * input frame t + (ctx[j]-ctx[0]) */
                    const float* xi = x + ((size_t)u * T + t + (ctx[j] - ctx[0])) * C;
                    const float* wj = W + (size_t)n * nctx * C + (size_t)j * C;
                    for (int c = 0; c < C; ++c) acc += (double)xi[c] * (double)wj[c];
                }
                if (acc < 0.0) acc = 0.0;
                if (gamma) acc = (acc - mean[n]) / sqrt((double)var[n] + (double)eps) * gamma[n] + beta[n];
                yo[n] = (float)acc;
            }
        }
}

/* x[B,T,C] -> out[B,2C] */
void xvo_stat_pool(const float* x, int B, int T, int C, float* out) {
    for (int u = 0; u < B; ++u)
        for (int c = 0; c < C; ++c) {
            double s = 0.0;
            for (int t = 0; t < T; ++t) s += x[((size_t)u * T + t) * C + c];
            const double m = s / T;
            double q = 0.0;
            for (int t = 0; t < T; ++t) {
                const double d = x[((size_t)u * T + t) * C + c] - m;
                q += d * d;
            }
            out[(size_t)u * 2 * C + c] = (float)m;
            out[(size_t)u * 2 * C + C + c] = (float)sqrt(q / (T - 1)); /* T==1 -> 0/0 = NaN like torch.std */
        }
}

/* y[M,N] = act(x[M,K] W[N,K]^T + b) */
void xvo_linear(const float* x, int M, int K, const float* W, const float* b, int N, int relu, float* y) {
    for (int m = 0; m < M; ++m) {
#pragma omp parallel for
        for (int n = 0; n < N; ++n) {
            double acc = b[n];
            for (int k = 0; k < K; ++k) acc += (double)x[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
            if (relu && acc < 0.0) acc = 0.0;
            y[(size_t)m * N + n] = (float)acc;
        }
    }
}

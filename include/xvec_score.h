/* xvec_score.h -- C ABI of the scoring back end in libxvec_hip.so (next row N4).
 *
 * The step behind the extraction path: the reference scores every test x-vector against every
 * other one with speechbrain's fast_PLDA_scoring (reference plda_classifier.py:81-87, called from
 * plda_score_stat.py:59) in numpy float64.  The arithmetic lives in the un-vendored dependency
 * speechbrain==0.5.12 (requirements.txt:55), absent from the build image: parity is UNPINNED
 * against the package itself; oracle/plda_oracle.py restates its published algorithm and is
 * checked against the closed-form two-covariance log-likelihood ratio (tests/test_scoring.py).
 *
 * Everything is fp64 (as the reference), row-major, DEVICE pointers, asynchronous on the caller's
 * stream, no allocation.  The [n_enroll, n_test] score matrix is an "NT" GEMM on
 * v_mfma_f64_16x16x4_f64.  Return codes as xvec_hip.h (0 = OK); message from
 * xvec_score_last_error().
 */
#ifndef XVEC_SCORE_H
#define XVEC_SCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* xvec_stream; /* hipStream_t */

const char* xvec_score_last_error(void);

/* C[M,N] = scale * ( A[M,K] . B[N,K]^T + rowv[m] + colv[n] + cst );  rowv / colv may be NULL.
 * lda/ldb/ldc are row strides in elements.  The building block of both scorers. */
int xvec_gemm_nt_f64(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t M, int64_t N,
                     int32_t K, const double* rowv, const double* colv, double cst, double scale,
                     double* C, int64_t ldc, xvec_stream stream);

/* Scratch for the two scorers below. */
size_t xvec_score_workspace_bytes(int64_t n_enroll, int64_t n_test, int32_t dim);

/* fast_PLDA_scoring (speechbrain.processing.PLDA_LDA, as called at plda_classifier.py:86):
 *   e = enroll - mean, t = test - mean
 *   scores[i,j] = scaling * ( 0.5 e_i' Phi e_i + 0.5 t_j' Phi t_j + e_i' Psi t_j + plda_cst )
 * (when phi_t == psi_t + dim*dim, i.e. the caller keeps [Psi^T ; Phi^T] stacked in one buffer, [e Psi | e Phi] is formed in one launch)
 * psi_t / phi_t are the TRANSPOSES of Psi / Phi ([dim,dim], row-major), plda_cst the Gaussian
 * constant; the host derives them from (F, Sigma) once per model (scoring.PldaScorer).
 * test == NULL scores enroll against itself (the reference's use: plda_score_stat.py:19-20). */
int xvec_plda_score(const double* enroll, int64_t n_enroll, const double* test, int64_t n_test,
                    int32_t dim, const double* mean, const double* psi_t, const double* phi_t,
                    double plda_cst, double scaling_factor, double* scores, void* workspace,
                    size_t workspace_bytes, xvec_stream stream);

/* The same scores through the model's low-rank structure: with tot = F F' + Sigma, L = tot^-1 F [dim, rank], G = F' L,
 * W = (I - G^2)^-1, Z = W G the matrices above are Phi = -L Z L' and Psi = L W L' exactly, so with y = (x - mean) L
 *   scores[i,j] = scaling * ( -0.5 y_i Z y_i' - 0.5 y_j Z y_j' + (y_i W) y_j' + plda_cst )
 * and the [n_enroll, n_test] product runs over rank instead of dim (the reference trains rank_f = 50 .. 200 on 512-d
 * x-vectors, main.py:293-308).  l_t = L^T [rank, dim]; wz_t = [ W^T ; (-Z)^T ] stacked [2 rank, rank]; both row-major,
 * derived once per model by the host (scoring.plda_lowrank).  Same workspace as xvec_plda_score; test == NULL as above. */
int xvec_plda_score_lowrank(const double* enroll, int64_t n_enroll, const double* test, int64_t n_test,
                            int32_t dim, int32_t rank, const double* mean, const double* l_t, const double* wz_t,
                            double plda_cst, double scaling_factor, double* scores, void* workspace,
                            size_t workspace_bytes, xvec_stream stream);

/* Cosine scoring: scores[i,j] = <enroll_i, test_j> / (|enroll_i| |test_j|).  test == NULL as above. */
int xvec_cosine_score(const double* enroll, int64_t n_enroll, const double* test, int64_t n_test,
                      int32_t dim, double* scores, void* workspace, size_t workspace_bytes,
                      xvec_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* XVEC_SCORE_H */

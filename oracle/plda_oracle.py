"""CPU oracle for the scoring back end (next-row N4).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against the package: the reference scores x-vectors with
`speechbrain.processing.PLDA_LDA.fast_PLDA_scoring` (`from speechbrain.processing.PLDA_LDA import *`,
reference plda_classifier.py:4; call site plda_classifier.py:86:
    fast_PLDA_scoring(en_stat, te_stat, ndx, plda.mean, plda.F, plda.Sigma, p_known=0.0)
with en_stat = te_stat = all test x-vectors, plda_score_stat.py:19-20,59).  speechbrain==0.5.12
(requirements.txt:55) is not installed in the build image and cannot be fetched, and the reference
holds no test or golden vector for it.  This file restates the published algorithm of that function
(speechbrain 0.5.12, PLDA_LDA.py, itself taken from SIDEKIT's iv_scoring.fast_PLDA_scoring) in numpy
float64, step by step; it has NOT been run against the package.

What pins the math instead (tests/test_scoring.py): for the two-covariance model the package
implements -- x = mean + F y + eps, y ~ N(0, I), eps ~ N(0, Sigma) -- the score must equal the
log-likelihood ratio  log p(e, t | same speaker) - log p(e) - log p(t)  computed directly from
Gaussian log-densities (`llr_direct` below, scipy-free, via Cholesky).  The restatement reproduces
that closed form to 1e-9, constant term included.
"""
from __future__ import annotations

import numpy as np


def plda_constants(F, Sigma, scaling_factor=1.0):
    """The model-only part of fast_PLDA_scoring: (Phi, Psi, plda_cst)."""
    F = np.asarray(F, dtype=np.float64)
    Sigma = np.asarray(Sigma, dtype=np.float64)
    invSigma = np.linalg.inv(Sigma)
    I_spk = np.eye(F.shape[1], dtype="float")
    K = F.T.dot(invSigma * scaling_factor).dot(F)
    K1 = np.linalg.inv(K + I_spk)
    K2 = np.linalg.inv(2 * K + I_spk)
    alpha1 = np.linalg.slogdet(K1)[1]
    alpha2 = np.linalg.slogdet(K2)[1]
    plda_cst = alpha2 / 2.0 - alpha1
    Sigma_ac = np.dot(F, F.T)
    Sigma_tot = Sigma_ac + Sigma
    Sigma_tot_inv = np.linalg.inv(Sigma_tot)
    Tmp = np.linalg.inv(Sigma_tot - Sigma_ac.dot(Sigma_tot_inv).dot(Sigma_ac))
    Phi = Sigma_tot_inv - Tmp
    Psi = Sigma_tot_inv.dot(Sigma_ac).dot(Tmp)
    return Phi, Psi, plda_cst


def fast_plda_scoring(enroll, test, mu, F, Sigma, scaling_factor=1.0):
    """Score matrix [n_enroll, n_test] float64 (p_known = 0, the reference's only setting; enrol
    models unique, so no per-model averaging -- the reference builds modelset from the unique ids)."""
    enroll = np.asarray(enroll, dtype=np.float64) - np.asarray(mu, dtype=np.float64)      # center_stat1
    test = np.asarray(test, dtype=np.float64) - np.asarray(mu, dtype=np.float64)
    Phi, Psi, plda_cst = plda_constants(F, Sigma, scaling_factor)
    model_part = 0.5 * np.einsum("ij, ji->i", enroll.dot(Phi), enroll.T)
    seg_part = 0.5 * np.einsum("ij, ji->i", test.dot(Phi), test.T)
    scoremat = model_part[:, np.newaxis] + seg_part + plda_cst
    scoremat += enroll.dot(Psi).dot(test.T)
    scoremat *= scaling_factor
    return scoremat


def cosine_scoring(enroll, test):
    e = np.asarray(enroll, dtype=np.float64)
    t = np.asarray(test, dtype=np.float64)
    e = e / np.linalg.norm(e, axis=1, keepdims=True)
    t = t / np.linalg.norm(t, axis=1, keepdims=True)
    return e @ t.T


def _gauss_logpdf(x, cov):
    """log N(x; 0, cov) for the rows of x."""
    L = np.linalg.cholesky(cov)
    z = np.linalg.solve(L, x.T)
    return -0.5 * np.sum(z * z, axis=0) - np.sum(np.log(np.diag(L))) - 0.5 * cov.shape[0] * np.log(2 * np.pi)


def llr_direct(enroll, test, mu, F, Sigma):
    """The definition the fast formula must equal (scaling_factor 1): same-speaker vs
    different-speaker log-likelihood ratio of the pair (e_i, t_j) under x = mu + F y + eps."""
    e = np.asarray(enroll, dtype=np.float64) - mu
    t = np.asarray(test, dtype=np.float64) - mu
    ac = F @ F.T
    tot = ac + Sigma
    joint = np.block([[tot, ac], [ac, tot]])
    le = _gauss_logpdf(e, tot)
    lt = _gauss_logpdf(t, tot)
    out = np.empty((e.shape[0], t.shape[0]))
    for i in range(e.shape[0]):
        pair = np.concatenate([np.repeat(e[i:i + 1], t.shape[0], 0), t], axis=1)
        out[i] = _gauss_logpdf(pair, joint) - le[i] - lt
    return out


def make_plda(dim, rank, seed):
    """A random well-conditioned PLDA model (mean, F, Sigma) for tests and benchmarks."""
    rng = np.random.default_rng(seed)
    mean = rng.normal(0, 1, dim)
    F = rng.normal(0, 1 / np.sqrt(dim), (dim, rank))
    A = rng.normal(0, 1 / np.sqrt(dim), (dim, dim))
    Sigma = A @ A.T + 0.5 * np.eye(dim)
    return mean, F, Sigma

"""Scoring back end on the GPU (next-row N4): the step behind the extraction path.

The reference scores all test x-vectors against each other with speechbrain's
`fast_PLDA_scoring(en_stat, te_stat, ndx, plda.mean, plda.F, plda.Sigma, p_known=0.0)` in numpy
float64 (reference plda_classifier.py:81-87, plda_score_stat.py:59).  Here the [n_enroll, n_test]
score matrix is one fp64 MFMA GEMM (include/xvec_score.h); the model-only constants (two 512x512
inverses and two log-determinants) are derived once per PLDA model on the host, as the reference
does on every call.

    scorer = PldaScorer(plda.mean, plda.F, plda.Sigma)            # any object with these arrays
    S = scorer.score(x_vecs)                                      # [N, N] float64 on the device
    scores = plda_scores(plda, en_stat, te_stat)                  # drop-in for plda_classifier.plda_scores

Parity with speechbrain itself is unpinned (not installed in the build image); the kernels are
checked against oracle/plda_oracle.py, which is in turn checked against the closed-form
log-likelihood ratio of the two-covariance model.
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip as _hip


def _check(rc: int):
    if rc != _hip.OK:
        raise _hip.XvecError(rc, _hip.lib.xvec_score_last_error().decode())


def _dev_f64(a, device) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        t = a.detach().to(device=device, dtype=torch.float64)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(device)
    return t.contiguous()


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_device(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("scoring runs on a HIP device only (no CPU path)")
    return device


def gemm_nt(a: torch.Tensor, b: torch.Tensor, rowv=None, colv=None, cst=0.0, scale=1.0) -> torch.Tensor:
    """scale * (a @ b.T + rowv[:,None] + colv[None,:] + cst) in fp64 (xvec_gemm_nt_f64)."""
    if not (a.is_cuda and b.is_cuda):
        raise RuntimeError("gemm_nt: expected tensors on a HIP device")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1] or a.shape[1] < 1:
        raise ValueError(f"gemm_nt: shapes {tuple(a.shape)} x {tuple(b.shape)}^T do not match")
    a = a.to(torch.float64).contiguous()
    b = b.to(torch.float64).contiguous()
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), dtype=torch.float64, device=a.device)
    rv = None if rowv is None else rowv.to(device=a.device, dtype=torch.float64).contiguous()
    cv = None if colv is None else colv.to(device=a.device, dtype=torch.float64).contiguous()
    if (rv is not None and rv.numel() != M) or (cv is not None and cv.numel() != N):
        raise ValueError("gemm_nt: rowv / colv length mismatch")
    with torch.cuda.device(a.device):
        _check(_hip.lib.xvec_gemm_nt_f64(a.data_ptr(), K, b.data_ptr(), K, M, N, K,
                                         rv.data_ptr() if rv is not None else None,
                                         cv.data_ptr() if cv is not None else None,
                                         float(cst), float(scale), out.data_ptr(), N, _stream(a.device)))
    return out


def plda_constants(F, Sigma, scaling_factor=1.0):
    """(Phi, Psi, plda_cst) of fast_PLDA_scoring: model-only, host, numpy float64."""
    F = np.asarray(F, dtype=np.float64)
    Sigma = np.asarray(Sigma, dtype=np.float64)
    inv = np.linalg.inv
    K = F.T @ (inv(Sigma) * scaling_factor) @ F
    eye = np.eye(F.shape[1])
    cst = np.linalg.slogdet(inv(2 * K + eye))[1] / 2.0 - np.linalg.slogdet(inv(K + eye))[1]
    ac = F @ F.T
    tot = ac + Sigma
    tot_inv = inv(tot)
    tmp = inv(tot - ac @ tot_inv @ ac)
    return tot_inv - tmp, tot_inv @ ac @ tmp, float(cst)


def plda_lowrank(F, Sigma, scaling_factor=1.0):
    """(L, W, Z, plda_cst) with Phi = -L Z L' and Psi = L W L' exactly (Woodbury): tot = F F' + Sigma, L = tot^-1 F [D, R],
    G = F' L, W = (I - G^2)^-1, Z = W G.  Model-only, host, numpy float64; what xvec_plda_score_lowrank takes instead of the
    dense Phi, Psi of `plda_constants` (the same scores to rounding over K = R instead of D)."""
    F = np.asarray(F, dtype=np.float64)
    Sigma = np.asarray(Sigma, dtype=np.float64)
    _, _, cst = plda_constants(F, Sigma, scaling_factor)
    L = np.linalg.solve(F @ F.T + Sigma, F)
    G = F.T @ L
    W = np.linalg.inv(np.eye(F.shape[1]) - G @ G)
    return L, W, W @ G, cst


class PldaScorer:
    """PLDA model (mean, F, Sigma) prepared for scoring on one HIP device.  `lowrank=None` takes the low-rank form whenever
    F has fewer columns than rows (the reference trains rank_f = 50 .. 200 on 512-d x-vectors), `False` forces the dense
    Phi / Psi products of the package's formulation."""

    def __init__(self, mean, F, Sigma, scaling_factor=1.0, device="cuda:0", lowrank=None):
        self.device = _require_device(device)
        F = np.asarray(F, dtype=np.float64)
        self.dim = int(F.shape[0])
        if np.asarray(mean).shape != (self.dim,) or np.asarray(Sigma).shape != (self.dim, self.dim):
            raise ValueError("PldaScorer: mean[D], F[D,R], Sigma[D,D] expected")
        phi, psi, cst = plda_constants(F, Sigma, scaling_factor)
        self.scaling_factor = float(scaling_factor)
        self.plda_cst = cst
        self._mean = _dev_f64(mean, self.device)
        # Psi^T and Phi^T stacked in ONE [2 dim, dim] buffer: the library then forms [e Psi | e Phi] in one launch
        # (xvec_plda_score: centring, the product and the row dots 0.5 e Phi e' in that launch, the score matrix in a second)
        self._psiphi_t = _dev_f64(np.concatenate([np.asarray(psi).T, np.asarray(phi).T], 0), self.device)
        self._psi_t = self._psiphi_t[: psi.shape[0]]
        self._phi_t = self._psiphi_t[psi.shape[0]:]
        self.rank = int(F.shape[1])
        self.lowrank = (self.rank < self.dim) if lowrank is None else bool(lowrank)
        if self.lowrank:
            if self.rank > self.dim:
                raise ValueError("PldaScorer: the low-rank form needs rank <= dim")
            L, W, Z, _ = plda_lowrank(F, Sigma, scaling_factor)
            self._l_t = _dev_f64(L.T, self.device)
            self._wz_t = _dev_f64(np.concatenate([W.T, -Z.T], 0), self.device)
        self._ws = None

    def _workspace(self, ne, nt):
        need = int(_hip.lib.xvec_score_workspace_bytes(ne, nt, self.dim))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def score(self, enroll, test=None) -> torch.Tensor:
        """[n_enroll, n_test] float64 on the device; test=None scores enroll against itself."""
        e = _dev_f64(enroll, self.device)
        t = None if test is None else _dev_f64(test, self.device)
        if e.dim() != 2 or e.shape[1] != self.dim or (t is not None and (t.dim() != 2 or t.shape[1] != self.dim)):
            raise ValueError(f"PldaScorer.score: expected [N, {self.dim}] x-vectors")
        ne, nt = e.shape[0], (e.shape[0] if t is None else t.shape[0])
        ws = self._workspace(ne, 0 if t is None else nt)
        out = torch.empty((ne, nt), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            if self.lowrank:
                _check(_hip.lib.xvec_plda_score_lowrank(e.data_ptr(), ne, t.data_ptr() if t is not None else None, nt,
                                                        self.dim, self.rank, self._mean.data_ptr(), self._l_t.data_ptr(),
                                                        self._wz_t.data_ptr(), self.plda_cst, self.scaling_factor,
                                                        out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(self.device)))
            else:
                _check(_hip.lib.xvec_plda_score(e.data_ptr(), ne, t.data_ptr() if t is not None else None, nt, self.dim,
                                                self._mean.data_ptr(), self._psi_t.data_ptr(), self._phi_t.data_ptr(),
                                                self.plda_cst, self.scaling_factor, out.data_ptr(), ws.data_ptr(),
                                                ws.numel(), _stream(self.device)))
        return out


def cosine_scores(enroll, test=None, device="cuda:0") -> torch.Tensor:
    device = _require_device(device)
    e = _dev_f64(enroll, device)
    t = None if test is None else _dev_f64(test, device)
    if e.dim() != 2 or (t is not None and (t.dim() != 2 or t.shape[1] != e.shape[1])):
        raise ValueError("cosine_scores: expected [N, D] x-vectors of one dimension")
    ne, nt, dim = e.shape[0], (e.shape[0] if t is None else t.shape[0]), e.shape[1]
    ws = torch.empty(int(_hip.lib.xvec_score_workspace_bytes(ne, 0 if t is None else nt, dim)), dtype=torch.uint8,
                     device=device)
    out = torch.empty((ne, nt), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        _check(_hip.lib.xvec_cosine_score(e.data_ptr(), ne, t.data_ptr() if t is not None else None, nt, dim,
                                          out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(device)))
    return out


class Scores:
    """What the reference reads from speechbrain's Scores object (plda_score_stat.py:60-87):
    modelset, segset, scoremask, scoremat (numpy float64 [n_models, n_segs])."""

    def __init__(self, modelset, segset, scoremat, scoremask=None):
        self.modelset = modelset
        self.segset = segset
        self.scoremat = scoremat
        self.scoremask = np.ones(scoremat.shape, dtype=bool) if scoremask is None else scoremask


def fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma, p_known=0.0, scaling_factor=1.0, device="cuda:0") -> Scores:
    """Drop-in for the call at plda_classifier.py:86.  enroll / test: objects with `.modelset`,
    `.segset`, `.stat1` (the reference's StatObject_SB); ndx: object with `.modelset`, `.segset`
    (and optionally `.trialmask`) or None.  Models and segments are scored in ndx order; names
    missing from the stats are dropped, as the package's check_missing does."""
    if p_known != 0:
        raise NotImplementedError("open-set scoring (p_known != 0) is not on the reference's path (it passes 0.0)")
    e_names = np.asarray(enroll.modelset)
    t_names = np.asarray(test.segset)
    e_x = np.asarray(enroll.stat1, dtype=np.float64)
    t_x = np.asarray(test.stat1, dtype=np.float64)
    if len(set(e_names.tolist())) != len(e_names):
        raise NotImplementedError("several enrolment vectors per model (per-model averaging) is not on the "
                                  "reference's path: it enrols every utterance under its own id")
    mask = None
    if ndx is not None:
        e_pos = {n: i for i, n in enumerate(e_names.tolist())}
        t_pos = {n: i for i, n in enumerate(t_names.tolist())}
        m_keep = [i for i, n in enumerate(np.asarray(ndx.modelset).tolist()) if n in e_pos]
        s_keep = [j for j, n in enumerate(np.asarray(ndx.segset).tolist()) if n in t_pos]
        m_names = np.asarray(ndx.modelset)[m_keep]
        s_names = np.asarray(ndx.segset)[s_keep]
        e_x = e_x[[e_pos[n] for n in m_names.tolist()]]
        t_x = t_x[[t_pos[n] for n in s_names.tolist()]]
        tm = getattr(ndx, "trialmask", None)
        if tm is not None:
            mask = np.asarray(tm)[np.ix_(m_keep, s_keep)]
        e_names, t_names = m_names, s_names
    # the reference scores the test set against ITSELF through two stat objects built from the same arrays
    # (plda_score_stat.py:19-20): same names and same vectors take the library's self path (upper triangle of tiles only)
    same = e_names.tolist() == t_names.tolist() and (e_x is t_x or (e_x.shape == t_x.shape and np.array_equal(e_x, t_x)))
    scorer = PldaScorer(mu, F, Sigma, scaling_factor=scaling_factor, device=device)
    mat = scorer.score(e_x, None if same else t_x)
    return Scores(e_names, t_names, mat.cpu().numpy(), mask)


class _Ndx:
    def __init__(self, models, testsegs):
        self.modelset = np.asarray(models)
        self.segset = np.asarray(testsegs)
        self.trialmask = np.ones((len(self.modelset), len(self.segset)), dtype=bool)


def plda_scores(plda, en_stat, te_stat, device="cuda:0") -> Scores:
    """plda_classifier.plda_scores (plda_classifier.py:81-87): every enrolment model against every
    test segment, `plda` any object with `.mean`, `.F`, `.Sigma`."""
    ndx = _Ndx(models=en_stat.modelset, testsegs=te_stat.modelset)
    return fast_PLDA_scoring(en_stat, te_stat, ndx, plda.mean, plda.F, plda.Sigma, p_known=0.0, device=device)

"""Scoring back end (next-row N4): oracle identities on the CPU, HIP kernels vs the oracle on the GPU.

Reference: plda_classifier.py:81-87 -> speechbrain fast_PLDA_scoring (numpy float64).  The package
is absent from the build image (parity with it is unpinned); the oracle restates its algorithm and
is pinned here to the closed-form log-likelihood ratio of the model the package implements."""
import numpy as np
import pytest
import torch

import plda_oracle as po

F64_TOL = 1e-10     # fp64 GEMM, K <= 512: summation order is the only difference


def _xvecs(n, dim, seed, mean=None):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1, (n, dim))
    return x if mean is None else x + mean


# ------------------------------------------------------------------ CPU: the oracle itself

@pytest.mark.parametrize("dim,rank", [(8, 3), (24, 24), (40, 10)])
def test_oracle_equals_direct_llr(dim, rank):
    mean, F, Sigma = po.make_plda(dim, rank, seed=dim)
    e, t = _xvecs(5, dim, 1, mean), _xvecs(7, dim, 2, mean)
    fast = po.fast_plda_scoring(e, t, mean, F, Sigma)
    direct = po.llr_direct(e, t, mean, F, Sigma)
    np.testing.assert_allclose(fast, direct, rtol=1e-9, atol=1e-9)


def test_oracle_properties():
    mean, F, Sigma = po.make_plda(16, 6, seed=3)
    x = _xvecs(9, 16, 4, mean)
    s = po.fast_plda_scoring(x, x, mean, F, Sigma)
    np.testing.assert_allclose(s, s.T, rtol=1e-12, atol=1e-12)           # Psi, Phi symmetric
    phi, psi, _ = po.plda_constants(F, Sigma)
    np.testing.assert_allclose(psi, psi.T, atol=1e-12)
    np.testing.assert_allclose(phi, phi.T, atol=1e-12)
    # scaling_factor multiplies the score and rescales K inside the constant
    s2 = po.fast_plda_scoring(x, x, mean, F, Sigma, scaling_factor=0.5)
    _, _, c1 = po.plda_constants(F, Sigma, 1.0)
    _, _, c2 = po.plda_constants(F, Sigma, 0.5)
    np.testing.assert_allclose(s2, 0.5 * (s - c1 + c2), rtol=1e-12, atol=1e-12)
    c = po.cosine_scoring(x, x)
    np.testing.assert_allclose(np.diag(c), 1.0, atol=1e-14)


def test_host_constants_match_oracle():
    """The product's host-side constant derivation against the oracle's (no GPU involved)."""
    from xvector_amd import scoring
    mean, F, Sigma = po.make_plda(32, 12, seed=5)
    phi, psi, cst = scoring.plda_constants(F, Sigma, 0.7)
    phi_o, psi_o, cst_o = po.plda_constants(F, Sigma, 0.7)
    np.testing.assert_allclose(phi, phi_o, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(psi, psi_o, rtol=1e-12, atol=1e-14)
    assert abs(cst - cst_o) < 1e-12


def test_scoring_refuses_cpu():
    from xvector_amd import scoring
    mean, F, Sigma = po.make_plda(8, 3, seed=0)
    with pytest.raises(RuntimeError):
        scoring.PldaScorer(mean, F, Sigma, device="cpu")
    with pytest.raises(RuntimeError):
        scoring.gemm_nt(torch.zeros(2, 2), torch.zeros(2, 2))
    with pytest.raises(RuntimeError):
        scoring.cosine_scores(np.zeros((2, 2)), device="cpu")


def test_score_abi_argument_errors_without_gpu():
    from xvector_amd import hip
    assert hip.lib.xvec_gemm_nt_f64(None, 4, None, 4, 2, 2, 0, None, None, 0.0, 1.0, None, 2, None) == hip.ERR_ARG
    assert b"shape" in hip.lib.xvec_score_last_error()
    assert hip.lib.xvec_plda_score(None, 3, None, 3, 8, None, None, None, 0.0, 1.0, None, None, 0, None) == hip.ERR_ARG
    assert hip.lib.xvec_score_workspace_bytes(-1, 0, 8) == 0
    assert hip.lib.xvec_score_workspace_bytes(10, 0, 8) > 0


# ------------------------------------------------------------------ GPU: kernels vs oracle

def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.gpu
# the last two shapes take the 128 x 128 tiles (812 / 810 of them, two rounds of the 512 block slots, against four rounds of the
# 1024 slots of the 64 x 64 tiles: at K > 256 a round of those counts 0.53, gemm_nt's rule) and their supertile walk, with ragged
# last bands / supertiles (29 x 28 and 45 x 18 tiles; even K the 16-byte loads, odd K the 8-byte ones); the same shapes at short K
# and the others run on the 64 x 64 variant, four blocks per CU
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (3, 5, 7), (128, 128, 16), (129, 127, 33), (200, 300, 512), (64, 1000, 150),
                                   (3601, 3500, 48), (5700, 2300, 17), (3601, 3500, 272), (5700, 2300, 257)])
def test_gemm_nt_f64_vs_numpy(M, N, K):
    from xvector_amd import scoring
    rng = np.random.default_rng(M * 1000 + N)
    a, b = rng.normal(0, 1, (M, K)), rng.normal(0, 1, (N, K))
    rv, cv = rng.normal(0, 1, M), rng.normal(0, 1, N)
    dev = "cuda:0"
    got = scoring.gemm_nt(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(rv).to(dev),
                          torch.from_numpy(cv).to(dev), cst=0.25, scale=-1.5).cpu().numpy()
    ref = -1.5 * (a @ b.T + rv[:, None] + cv[None, :] + 0.25)
    assert _rel(got, ref) < F64_TOL
    got0 = scoring.gemm_nt(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    assert _rel(got0, a @ b.T) < F64_TOL


def test_lowrank_factors_reproduce_phi_and_psi():
    """Phi = -L Z L', Psi = L W L' (the identity the low-rank scorer rests on), on the host in float64."""
    from xvector_amd import scoring
    for dim, rank in [(512, 200), (512, 50), (64, 20), (25, 25), (40, 1)]:
        mean, F, Sigma = po.make_plda(dim, rank, seed=3 + rank)
        phi, psi, cst = po.plda_constants(F, Sigma)
        L, W, Z, cst_lr = scoring.plda_lowrank(F, Sigma)
        assert L.shape == (dim, rank) and W.shape == Z.shape == (rank, rank) and cst_lr == pytest.approx(cst, rel=1e-14)
        assert np.abs(phi + L @ Z @ L.T).max() <= 1e-12 * np.abs(phi).max()
        assert np.abs(psi - L @ W @ L.T).max() <= 1e-12 * np.abs(psi).max()


@pytest.mark.gpu
@pytest.mark.parametrize("lowrank", [True, False])
@pytest.mark.parametrize("dim,rank,ne,nt", [(512, 150, 300, 517), (512, 200, 129, 129), (200, 50, 64, 33), (25, 25, 7, 300),
                                            (512, 1, 70, 40), (96, 37, 130, 131)])
def test_plda_score_vs_oracle(dim, rank, ne, nt, lowrank):
    from xvector_amd import scoring
    mean, F, Sigma = po.make_plda(dim, rank, seed=dim + rank)
    e, t = _xvecs(ne, dim, 1, mean), _xvecs(nt, dim, 2, mean)
    scorer = scoring.PldaScorer(mean, F, Sigma, lowrank=lowrank)
    assert scorer.lowrank == lowrank
    got = scorer.score(e, t).cpu().numpy()
    ref = po.fast_plda_scoring(e, t, mean, F, Sigma)
    assert got.shape == (ne, nt) and got.dtype == np.float64
    assert _rel(got, ref) < 1e-9
    # enrol against itself (the reference's use, plda_score_stat.py:19-20) walks the upper triangle of tiles only and
    # writes every element twice: symmetric bit for bit
    got_self = scorer.score(e).cpu().numpy()
    assert _rel(got_self, po.fast_plda_scoring(e, e, mean, F, Sigma)) < 1e-9
    assert np.array_equal(got_self, got_self.T)
    # ... and the two-set path given the same vectors twice (full tile walk, no mirroring) agrees with it to rounding
    got_two = scorer.score(e, e).cpu().numpy()
    assert _rel(got_two, got_self) < 1e-12
    # scaling factor
    got_s = scoring.PldaScorer(mean, F, Sigma, scaling_factor=0.5, lowrank=lowrank).score(e, t).cpu().numpy()
    assert _rel(got_s, po.fast_plda_scoring(e, t, mean, F, Sigma, scaling_factor=0.5)) < 1e-9


@pytest.mark.gpu
def test_plda_scores_dropin_surface():
    """plda_classifier.plda_scores' I/O: stat objects in, Scores(modelset, segset, scoremat) out;
    x-vectors arrive from the CSV as float64 widened from fp32 (main.py:142-145)."""
    from xvector_amd import scoring

    class Stat:      # the fields of StatObject_SB the path reads (plda_classifier.py:71-79)
        def __init__(self, ids, x):
            self.modelset = np.array(ids, dtype="|O")
            self.segset = np.array(ids, dtype="|O")
            self.stat1 = x

    class Plda:
        pass

    dim, n = 512, 200
    plda = Plda()
    plda.mean, plda.F, plda.Sigma = po.make_plda(dim, 150, seed=11)
    x = _xvecs(n, dim, 3, plda.mean).astype(np.float32).astype(np.float64)
    ids = [f"id1{i:04d}/clip/{i % 7:05d}.wav" for i in range(n)]
    en, te = Stat(ids, x), Stat(ids, x)
    sc = scoring.plda_scores(plda, en, te)
    assert list(sc.modelset) == ids and list(sc.segset) == ids
    assert sc.scoremat.shape == (n, n) and sc.scoremat.dtype == np.float64 and sc.scoremask.all()
    assert _rel(sc.scoremat, po.fast_plda_scoring(x, x, plda.mean, plda.F, plda.Sigma)) < 1e-9
    # en_stat and te_stat are two objects built from the same vectors (plda_score_stat.py:19-20): the self path
    assert np.array_equal(sc.scoremat, sc.scoremat.T)
    # the lookup the reference does per trial (plda_score_stat.py:66-74)
    i = int(np.where(sc.modelset == ids[5])[0][0])
    j = int(np.where(sc.segset == ids[17])[0][0])
    assert (i, j) == (5, 17)
    with pytest.raises(NotImplementedError):
        scoring.fast_PLDA_scoring(en, te, None, plda.mean, plda.F, plda.Sigma, p_known=0.1)


@pytest.mark.gpu
def test_cosine_vs_oracle():
    from xvector_amd import scoring
    e, t = _xvecs(100, 512, 1), _xvecs(77, 512, 2)
    got = scoring.cosine_scores(e, t).cpu().numpy()
    assert _rel(got, po.cosine_scoring(e, t)) < F64_TOL
    self_scores = scoring.cosine_scores(e).cpu().numpy()
    np.testing.assert_allclose(np.diag(self_scores), 1.0, atol=1e-13)
    assert np.array_equal(self_scores, self_scores.T)
    assert _rel(self_scores, po.cosine_scoring(e, e)) < F64_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("lowrank", [True, False])
@pytest.mark.parametrize("n", [4874, 5600])
def test_full_size_properties(lowrank, n):
    """VoxCeleb1 test-set size (4874 utterances, 512-d): symmetry, diagonal = self-score formula, and
    a sampled block against the oracle.  4874 walks 64 x 64 tiles in both forms (780 tiles of 128 x 128 would fill the 512
    block slots one and a half times: csrc/score.hip, gemm_nt); 5600 walks the 128 x 128 tiles in the dense form (990 of them: two
    full rounds, K = 512) and 64 x 64 in the low-rank form (K = 200: the small tiles win whenever they pack as well) -- symmetric
    walks all."""
    from xvector_amd import scoring
    dim = 512
    mean, F, Sigma = po.make_plda(dim, 200, seed=21)
    x = _xvecs(n, dim, 5, mean)
    scorer = scoring.PldaScorer(mean, F, Sigma, lowrank=lowrank)
    s = scorer.score(x)
    assert s.shape == (n, n)
    assert torch.equal(s, s.T)              # only tiles on or above the diagonal are computed, the rest are their mirror images
    phi, psi, cst = po.plda_constants(F, Sigma)
    xc = x - mean
    diag_ref = np.einsum("ij,ij->i", xc @ (phi + psi), xc) + cst
    assert _rel(torch.diagonal(s).cpu().numpy(), diag_ref) < 1e-9
    rows, cols = slice(n - 74, n), slice(1000, 1100)
    ref = po.fast_plda_scoring(x[rows], x[cols], mean, F, Sigma)
    assert _rel(s[rows, cols].cpu().numpy(), ref) < 1e-9
    ref_d = po.fast_plda_scoring(x[n - 84:n], x[n - 84:n], mean, F, Sigma)          # a diagonal tile and its ragged edge
    assert _rel(s[n - 84:, n - 84:].cpu().numpy(), ref_d) < 1e-9
    # the non-self path (full tile walk) is unchanged by the self path's shortcut
    two = scorer.score(x[:700], x[300:1500])
    assert _rel(two.cpu().numpy(), s[:700, 300:1500].cpu().numpy()) < 1e-12

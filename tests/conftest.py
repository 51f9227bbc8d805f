import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def assert_parity(got, ref, tol=1e-4, what="", elem_tol=None):
    """The path's tolerance (BASELINE north_star: 1e-4 relative fp32): per-row
    norm-wise relative error <= tol AND allclose(rtol=tol, atol=tol*mean|ref|).  Element-wise
    relative error alone is ill-conditioned: the reference differs from itself by 9e-3 on
    near-zero components across batch sizes."""
    got = torch.as_tensor(np.asarray(got) if not torch.is_tensor(got) else got).double().cpu()
    ref = torch.as_tensor(np.asarray(ref) if not torch.is_tensor(ref) else ref).double().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    g2, r2 = got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])
    rel = (g2 - r2).norm(dim=1) / r2.norm(dim=1).clamp_min(1e-30)
    assert rel.max().item() <= tol, f"{what}: row-wise relative error {rel.max().item():.3e} > {tol}"
    et = tol if elem_tol is None else elem_tol
    atol = et * ref.abs().mean().item()
    bad = (got - ref).abs() > (atol + et * ref.abs())
    assert not bad.any(), f"{what}: {int(bad.sum())} elements outside rtol={et}, atol={atol:.3e}"


def assert_parity_masked(got, ref, tol, what, elem_tol, exclude, max_excluded=2.5e-2, atol_scale=None):
    """assert_parity with an EXPLICIT list of ill-conditioned elements instead of a widened element-wise tolerance (VERDICT r03
    item 5): the row-wise norm check runs on everything, the element-wise check at `elem_tol` on every element that is not in
    `exclude` (bool, same shape), and `exclude` may cover at most `max_excluded` of the elements.  Returns the excluded share.
    `atol_scale`: magnitude the absolute part of the element-wise bound refers to (default: mean |ref|) -- for the std half of
    the pooled statistics checked alone, the scale of the ACTIVATIONS (mean |pooled row|, means included): 45 % of the seed-42
    stds are exactly 0 and deflate mean |std| to half a per cent of an activation, while a channel that is off in an utterance
    but shares a 32-row group (and with it the pooling pivot K, tdnn_common.h) with a neighbour in which it is on comes out at
    ~3e-4 K instead of exactly 0 (fp32 sums of n identical deviations -K): 1e-6 against a bound of 5.7e-7."""
    got = torch.as_tensor(np.asarray(got) if not torch.is_tensor(got) else got).double().cpu()
    ref = torch.as_tensor(np.asarray(ref) if not torch.is_tensor(ref) else ref).double().cpu()
    exclude = torch.as_tensor(exclude).cpu().bool()
    assert got.shape == ref.shape == exclude.shape, f"{what}: shapes {tuple(got.shape)} {tuple(ref.shape)} {tuple(exclude.shape)}"
    share = float(exclude.double().mean())
    assert share <= max_excluded, f"{what}: {share:.2e} of the elements excluded as ill-conditioned (limit {max_excluded:.0e})"
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    g2, r2 = got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])
    rel = (g2 - r2).norm(dim=1) / r2.norm(dim=1).clamp_min(1e-30)
    assert rel.max().item() <= tol, f"{what}: row-wise relative error {rel.max().item():.3e} > {tol}"
    atol = elem_tol * (ref.abs().mean().item() if atol_scale is None else float(atol_scale))
    bad = ((got - ref).abs() > (atol + elem_tol * ref.abs())) & ~exclude
    worst = [(tuple(int(v) for v in i), float(got[tuple(i)]), float(ref[tuple(i)])) for i in bad.nonzero()[:5]]
    assert not bad.any(), (f"{what}: {int(bad.sum())} elements outside rtol={elem_tol}, atol={atol:.3e} ({int(exclude.sum())} excluded); "
                           f"first (index, got, ref): {worst}")
    line = f"[mask] {what}: {int(exclude.sum())} of {exclude.numel()} elements excluded ({share:.2e}; limit {max_excluded:.2e})"
    print(line)
    if os.environ.get("XVEC_MASK_LOG"):              # profiles/r05_mask_shares.txt is such a log of one full GPU run
        with open(os.environ["XVEC_MASK_LOG"], "a") as f:
            f.write(line + "\n")
    return share


def nearly_off_channels(relu_frames, ref_std=None, pre_act=None, min_on=8, tiny=1e-2, near_zero=1e-3):
    """[U, C] bool: the (utterance, channel) pairs whose pooled standard deviation is ill-conditioned, LISTED instead of covered by
    a wide tolerance: channels whose ReLU output (reference tdnn_layer.py:31, before the BatchNorm) is above zero in at least one
    but fewer than `min_on` of the utterance's frames [U, T, C], and -- with `ref_std` [U, C] -- those whose reference std is
    under `tiny` of the mean non-zero std, and -- with `pre_act` [U, T, C], the values in front of the ReLU -- channels that are
    never on but come within `near_zero` of the mean |pre-activation| of it.  Such a std hangs on a handful of values near zero: the fp32 reference disagrees with
    its own fp64 run there by more than 1e-4 (all 31 such elements of a 16-utterance sample are on in 1..3 frames and under
    0.8 % of the mean std).  With the seed-42 weights 45 % of layer 5's channels are never on in an utterance (std
    exactly 0, checked like any other element), 1.5 % are on in 1..7 of 286 frames and 0.3 % have a tiny std: the share this
    mask may take is bounded by assert_parity_masked: every call site passes 1.5 x the share measured in a full run
    (profiles/r05_mask_shares.txt)."""
    cnt = (relu_frames > 1e-12).sum(dim=1)
    m = (cnt > 0) & (cnt < min_on)
    if ref_std is not None:
        ref_std = torch.as_tensor(ref_std).double().cpu()
        nz = ref_std[ref_std > 0]
        if nz.numel():
            m = m | ((ref_std > 0) & (ref_std < tiny * nz.mean()))
    if pre_act is not None:
        # never on in the reference, but some frame's pre-activation is within rounding of zero: one rounding flips it on
        # (seen: fp32 std 1.0e-6 against an fp64 reference of exactly 0)
        pre_act = torch.as_tensor(pre_act).double().cpu()
        m = m | ((cnt == 0) & (pre_act.max(dim=1).values > -near_zero * pre_act.abs().mean()))
    return m


@pytest.fixture(scope="session")
def synth():
    import xvector_amd
    return xvector_amd.synth


@pytest.fixture(scope="session")
def sd42(synth):
    """Full-width synthetic weights, seed 42 (the ones the golden fixtures were made with)."""
    return {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(seed=42).items()}


def float_params(sd):
    return {k: v for k, v in sd.items() if v.is_floating_point()}


@pytest.fixture(scope="session")
def c_oracle():
    path = os.path.join(ROOT, "oracle", "libxvec_oracle.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(path)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    lib.xvo_tdnn_layer.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, fp, fp,
                                   ctypes.c_float, ip, ctypes.c_int, ctypes.c_int, fp]
    lib.xvo_stat_pool.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
    lib.xvo_linear.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int, ctypes.c_int, fp]
    for f in (lib.xvo_tdnn_layer, lib.xvo_stat_pool, lib.xvo_linear):
        f.restype = None
    return lib


@pytest.fixture(scope="session")
def gpu_model(sd42):
    """Full-width XVectorModel on cuda:0 with the seed-42 weights."""
    import xvector_amd as xa
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    m = xa.XVectorModel()
    m.load_state_dict(sd42)
    return m.to("cuda:0").eval()

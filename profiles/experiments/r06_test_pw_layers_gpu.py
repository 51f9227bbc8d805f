"""Whole-tensor parity of the one-wave-per-SIMD kernel `pw::tdnn_pw_kernel` (csrc/tdnn_pw.hip: the store layers 2-4 of plain bf16
at large batches), in the manner of tests/test_large_batch_layers_gpu.py: every ELEMENT of every layer's output, at sizes
that give its persistent blocks every tile height (2, 3, 4 units of 64 frames, masked last units) and utterance boundaries
inside tiles (the activation slab of a tile then holds the rows of two or more utterances, its fragment reads skip `span` rows
at each boundary), against
  1. the fp64 oracle (reference tdnn_layer.py:26-41) at the bf16 bar,
  2. the same arithmetic on the 128x128 kernel (XVEC_PP=0), in the domain the kernels store (ReLU outputs before BatchNorm),
  3. itself (repeat runs are bit-identical),
and, end to end, fixed-length and ragged batches against the oracle with the dispatch asserted.  Short utterances (fewer than
32 output rows) must fall back to the ping-pong kernel.
"""
import os

import numpy as np
import pytest
import torch

import xvector_oracle as oracle
from conftest import assert_parity, float_params
from test_large_batch_layers_gpu import _model, _oracle_layer, _pre_bn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(52, 300), (63, 300), (100, 300), (128, 300), (160, 300), (256, 300), (70, 517), (900, 47)]


def _model_pw(sd, on=True):
    import xvector_amd as xa
    m = xa.XVectorModel(precision="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV)
    old = os.environ.get("XVEC_PW")
    try:
        os.environ["XVEC_PW"] = "1" if on else "0"
        m._engine(torch.device(DEV))          # creates the handle now, under this environment
    finally:
        if old is None:
            os.environ.pop("XVEC_PW", None)
        else:
            os.environ["XVEC_PW"] = old
    return m


@pytest.fixture(scope="module")
def models(sd42):
    return _model_pw(sd42), _model(sd42, "bf16", pp=False)


@pytest.mark.parametrize("B,T", SHAPES)
def test_pw_every_layer_every_element(gpu_model, sd42, synth, models, B, T):
    m_pw, m_old = models
    p64 = oracle.cast_params(float_params(sd42), torch.float64)
    h = torch.as_tensor(synth.make_mfcc(B, T, seed=1000 + B)).to(DEV)
    for i in range(4):
        if i > 0:
            got = m_pw.time_context_layers[i](h)
            assert m_pw.last_dispatch()[i] == "pw", f"layer {i}: {m_pw.last_dispatch()}"
            assert torch.equal(got, m_pw.time_context_layers[i](h)), f"layer {i}: repeat run differs"
            ref = _oracle_layer(h.cpu(), p64, i, key=("chain", B, T))
            assert_parity(got, ref, 1e-2, f"pw layer {i} B={B} T={T} vs oracle", elem_tol=4e-2)
            old = m_old.time_context_layers[i](h)
            assert m_old.last_dispatch()[i] == "tile128"
            # The two kernels sum the same bf16 products in another order (32-deep stages with the taps innermost against 64-deep
            # chunks): about one element in 10^4 rounds to the other bf16 neighbour (measured: 0.008 %, profiles/diag/pw_check.py),
            # so ELEMENT by element: at most one bf16 step of the larger value, plus the fp32 sums' own noise -- a value that
            # missed a single product term is tens of times further off.  Row-wise: one flipped large element moves a row by
            # up to 1.5e-3 (measured), a corrupted row is at 2e-2 and more.
            a, b = _pre_bn(got, sd42, i), _pre_bn(old, sd42, i)
            d = (a - b).abs()
            bound = 1.01 * 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 3e-4
            assert bool((d <= bound).all()), (f"pw layer {i} B={B} T={T}: {int((d > bound).sum())} elements more than a bf16 step from the "
                                               f"128x128 kernel's, worst {float((d - bound).max()):.3e} over the bound")
            assert float((d > 0).double().mean()) < 1e-3, "more than rounding flips differ"
            assert_parity(a, b, 3e-3, f"pw layer {i} B={B} T={T} vs 128x128 kernel", elem_tol=2e-2)
        h = gpu_model.time_context_layers[i](h)       # next layer's input: the fp32 path's output


@pytest.mark.parametrize("B,T", [(256, 300), (96, 299)])
def test_pw_end_to_end_fixed(sd42, synth, models, B, T):
    m_pw, _ = models
    x = torch.as_tensor(synth.make_mfcc(B, T, seed=7 + B))
    got = m_pw.extract_x_vec(x.to(DEV))
    assert m_pw.last_dispatch() == ["first", "pw", "pw", "pw", "pp"]
    idx = [0, 1, B // 2, B - 1]
    p64 = oracle.cast_params(float_params(sd42), torch.float64)
    assert_parity(got[idx], oracle.extract_x_vec(x[idx].double(), p64), 1e-2, "pw x-vectors, fixed length")
    assert torch.equal(got, m_pw.extract_x_vec(x.to(DEV)))


def test_pw_end_to_end_ragged_and_short_utterance_fallback(sd42, synth, models):
    m_pw, _ = models
    B, T = 200, 700
    rng = np.random.default_rng(11)
    lens = rng.integers(46, T + 1, B).tolist()           # >= 46 frames: 32 output rows behind every store layer
    lens[0], lens[1], lens[2] = 46, 47, T
    x = torch.as_tensor(synth.make_mfcc(B, T, seed=99))
    got = m_pw.extract_x_vec(x.to(DEV), lengths=lens)
    assert m_pw.last_dispatch() == ["first", "pw", "pw", "pw", "pp"]
    p64 = oracle.cast_params(float_params(sd42), torch.float64)
    idx = [0, 1, 2, 50, 199]
    ref = torch.cat([oracle.extract_x_vec(x[i:i + 1, :lens[i]].double(), p64) for i in idx])
    assert_parity(got[idx], ref, 1e-2, "pw x-vectors, ragged")
    # one utterance of 40 frames: 26 output rows behind layer 3 -- the slab of a tile could meet more boundaries than it has
    # room for; those layers go back to the ping-pong kernel (layer 2 keeps 32 rows: 40 - 8)
    lens[5] = 40
    got2 = m_pw.extract_x_vec(x.to(DEV), lengths=lens)
    assert m_pw.last_dispatch() == ["first", "pw", "pp", "pp", "pp"]
    ref5 = oracle.extract_x_vec(x[5:6, :40].double(), p64)
    assert_parity(got2[[5]], ref5, 1e-2, "x-vector of the short utterance")
    assert_parity(got2[idx[3:]], ref[3:], 1e-2, "x-vectors beside it")

"""The oracle (PyTorch restatement + plain-C restatement) against the golden vectors that
were produced by running the reference itself (tests/golden/make_golden.py).  CPU only."""
import ctypes

import numpy as np
import pytest
import torch

import xvector_oracle as oracle
from conftest import assert_parity, float_params, load_golden


def _p(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def test_g1_time_context_known_answers():
    g = load_golden("g1_time_context.npz")
    x = torch.from_numpy(g["x15"])
    for i in range(4):
        got = torch.cat(oracle.get_time_context(x, g[f"ctx{i}"].tolist()), 2)
        assert torch.equal(got, torch.from_numpy(g[f"out{i}"]))
    # shapes and first row quoted in SURVEY.md §4 from extra/time_context_test.py
    assert g["out0"].shape == (5, 11, 5) and g["out1"].shape == (5, 11, 2)
    assert g["out2"].shape == (5, 7, 5) and g["out3"].shape == (5, 5, 11)
    assert g["out0"][0, 0, :, ].tolist() == [1, 2, 3, 4, 5] and g["out0"][0, -1].tolist() == [11, 12, 13, 14, 15]
    got = torch.cat(oracle.get_time_context(torch.from_numpy(g["xdoc"]), [-1, 0, 1]), 2)
    assert torch.equal(got, torch.from_numpy(g["outdoc"])) and got.shape == (1, 3, 6)


def test_g2_per_layer(sd42, synth):
    g = load_golden("g2_layers.npz")
    p = float_params(sd42)
    h = torch.from_numpy(synth.make_mfcc(int(g["B"]), int(g["T"]), seed=int(g["seed_x"])))
    for i in range(5):
        h = oracle.tdnn_layer(h, p, f"time_context_layers.{i}.", oracle.CONTEXTS[i])
        assert list(h.shape) == g[f"l{i}_shape"].tolist()
        assert_parity(h[:, g[f"l{i}_frames"].tolist(), :], g[f"l{i}_rows"], 1e-5, f"layer {i} rows")
        assert_parity(h.double().sum(dim=(0, 1))[None], g[f"l{i}_sum"][None], 1e-5, f"layer {i} sums")


def test_g2_per_layer_c_oracle(sd42, synth, c_oracle):
    g = load_golden("g2_layers.npz")
    B, T = int(g["B"]), int(g["T"])
    h = synth.make_mfcc(B, T, seed=int(g["seed_x"]))
    for i in range(5):
        pre = f"time_context_layers.{i}."
        W = sd42[pre + "linear.weight"].numpy()
        ctx = oracle.CONTEXTS[i]
        N, C = W.shape[0], h.shape[2]
        To = h.shape[1] - (ctx[-1] - ctx[0])
        y = np.empty((B, To, N), dtype=np.float32)
        cctx = (ctypes.c_int * len(ctx))(*ctx)
        c_oracle.xvo_tdnn_layer(_p(h), B, h.shape[1], C, _p(W), _p(sd42[pre + "linear.bias"].numpy()),
                                _p(sd42[pre + "norm.weight"].numpy()), _p(sd42[pre + "norm.bias"].numpy()),
                                _p(sd42[pre + "norm.running_mean"].numpy()), _p(sd42[pre + "norm.running_var"].numpy()),
                                1e-5, cctx, len(ctx), N, _p(y))
        assert_parity(y[:, g[f"l{i}_frames"].tolist(), :], g[f"l{i}_rows"], 1e-5, f"C layer {i}")
        h = y


def test_g3_stat_pool(c_oracle):
    g = load_golden("g3_stat_pool.npz")
    rng = np.random.default_rng(int(g["seed"]))
    x = (rng.standard_normal((4, 286, 1500)) * float(g["scale"]) + float(g["shift"])).astype(np.float32)
    assert_parity(oracle.stat_pool(torch.from_numpy(x)), g["pool_286"], 1e-5, "pool 286")
    for n in (2, 3):
        xs = np.ascontiguousarray(x[:, :n, :64])
        assert_parity(oracle.stat_pool(torch.from_numpy(xs)), g[f"pool_{n}"], 1e-5, f"pool {n}")
        out = np.empty((4, 128), dtype=np.float32)
        c_oracle.xvo_stat_pool(_p(xs), 4, n, 64, _p(out))
        assert_parity(out, g[f"pool_{n}"], 1e-5, f"C pool {n}")
    # n == 1: unbiased std is NaN in the reference; the oracle mirrors it
    one = oracle.stat_pool(torch.from_numpy(x[:, :1, :8]))
    assert torch.isnan(one[:, 8:]).all() and torch.isfinite(one[:, :8]).all()


@pytest.mark.parametrize("B,T", [(1, 299), (8, 299), (1, 300), (8, 300)])
def test_g4_full_path(sd42, synth, B, T):
    g = load_golden("g4_full.npz")
    p = float_params(sd42)
    key = f"B{B}_T{T}"
    x = torch.from_numpy(synth.make_mfcc(B, T, seed=int(g[key + "_seed_x"])))
    with torch.no_grad():
        assert_parity(oracle.extract_x_vec(x, p, 6), g[key + "_xvec6"], 1e-5, "xvec6")
        assert_parity(oracle.extract_x_vec(x, p, 7), g[key + "_xvec7"], 1e-5, "xvec7")
        assert_parity(oracle.forward(x, p), g[key + "_logits"], 1e-5, "logits")


def test_g4_fp64_oracle_agrees(sd42, synth):
    """The fp64 evaluation of the oracle (used as the high-precision yardstick for the
    GPU kernels) is within fp32 rounding of the reference's fp32 output."""
    g = load_golden("g4_full.npz")
    p64 = oracle.cast_params(float_params(sd42), torch.float64)
    x = torch.from_numpy(synth.make_mfcc(8, 300, seed=int(g["B8_T300_seed_x"]))).double()
    assert_parity(oracle.extract_x_vec(x, p64), g["B8_T300_xvec6"], 1e-5, "fp64 oracle")


def test_g5_ragged_definition(sd42, synth):
    g = load_golden("g5_ragged.npz")
    p = float_params(sd42)
    x = torch.from_numpy(synth.make_mfcc(3, 1000, seed=int(g["seed_x"])))
    with torch.no_grad():
        got = oracle.extract_x_vec_ragged(x, g["lengths"].tolist(), p)
    assert_parity(got, g["xvec6"], 1e-5, "ragged xvec6")


@pytest.mark.parametrize("tag", ["bn", "nobn"])
def test_g6_tiny_model(synth, c_oracle, tag):
    g = load_golden("g6_tiny.npz")
    bn = tag == "bn"
    if bn:
        p = {k[len("bn/w/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("bn/w/")}
        # the stored weights ARE what the generator still produces (fixture self-consistency)
        regen = synth.make_state_dict(seed=int(g["bn/seed_w"]), input_size=24, hidden_size=32, num_classes=10,
                                      x_vector_size=16)
        for k, v in regen.items():
            assert np.array_equal(v, g["bn/w/" + k]), k
    else:
        p = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(
            seed=int(g["nobn/seed_w"]), input_size=24, hidden_size=32, num_classes=10, x_vector_size=16,
            batch_norm=False).items()}
    p = float_params(p)
    x = torch.from_numpy(g[f"{tag}/x"])
    with torch.no_grad():
        fr = oracle.time_context_layers(x, p, batch_norm=bn)
        assert_parity(fr[:, g[f"{tag}/frames_t"].tolist()], g[f"{tag}/frames"], 1e-5, "frames")
        assert_parity(oracle.extract_x_vec(x, p, 6, bn), g[f"{tag}/xvec6"], 1e-5, "xvec6")
        assert_parity(oracle.extract_x_vec(x, p, 7, bn), g[f"{tag}/xvec7"], 1e-5, "xvec7")
        assert_parity(oracle.forward(x, p, bn), g[f"{tag}/logits"], 1e-5, "logits")
    # C restatement of the segment-level affine on the oracle's pooled vector
    pooled = oracle.stat_pool(fr).numpy()
    out = np.empty((4, 16), dtype=np.float32)
    c_oracle.xvo_linear(_p(pooled), 4, 3000, _p(p["segment_layer6.weight"].numpy()),
                        _p(p["segment_layer6.bias"].numpy()), 16, 0, _p(out))
    assert_parity(out, g[f"{tag}/xvec6"], 1e-5, "C seg6")


def test_flops_formula():
    assert oracle.flops_per_utt(300) == 1_539_518_240

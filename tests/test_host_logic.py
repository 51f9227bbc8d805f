"""CPU-only checks of the boundary: the C-ABI library loads and exports what include/xvec_hip.h
declares, the host mirror keeps the reference's surface (constructor, state_dict, helper
functions), and the product path refuses to run anywhere but on the HIP device."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import xvector_oracle as oracle
from conftest import ROOT, load_golden


def _header_functions():
    import glob
    names = set()
    for path in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        src = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        names |= set(re.findall(r"\b(xvec_[a-z_0-9]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from xvector_amd import hip
    declared = _header_functions()
    assert len(declared) >= 24
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert sorted(hip.EXPORTS) == declared, "ctypes binding and header disagree"
    assert "gfx950" in hip.version()
    assert hip.last_error() == ""


def test_binding_loads_torch_before_the_library():
    """libxvec_hip.so has to bind to the HIP runtime torch ships: the ctypes binding imports torch
    before it opens the library (two runtimes in one process: "no ROCm-capable device")."""
    src = open(os.path.join(ROOT, "speaker-recognition-x-vectors_amd", "hip.py")).read()
    assert 0 <= src.index("\nimport torch") < src.index("C.CDLL(")


def test_c_abi_argument_errors_without_gpu():
    """Error paths that never touch the device: null arguments."""
    from xvector_amd import hip
    assert hip.lib.xvec_create(None, None) == hip.ERR_ARG
    assert "null" in hip.last_error()
    assert hip.lib.xvec_workspace_bytes(None, 100, 1) == 0
    assert hip.lib.xvec_set_profiling(None, 1) == hip.ERR_ARG
    hip.lib.xvec_destroy(None)      # no-op


def test_get_time_context_matches_reference_known_answers():
    import xvector_amd as xa
    g = load_golden("g1_time_context.npz")
    x = torch.from_numpy(g["x15"])
    for i in range(4):
        got = torch.cat(xa.get_time_context(x, g[f"ctx{i}"].tolist()), 2)
        assert torch.equal(got, torch.from_numpy(g[f"out{i}"]))
    got = torch.cat(xa.get_time_context(torch.from_numpy(g["xdoc"]), [-1, 0, 1]), 2)
    assert torch.equal(got, torch.from_numpy(g["outdoc"]))
    # same function as the oracle's restatement on random data and the model's own contexts
    xr = torch.randn(2, 40, 3)
    for ctx in oracle.CONTEXTS:
        a, b = xa.get_time_context(xr, ctx), oracle.get_time_context(xr, ctx)
        assert all(torch.equal(u, v) for u, v in zip(a, b))


def test_state_dict_surface_matches_reference(synth):
    """Key names and shapes of main.py:38-47: a reference checkpoint's
    state_dict loads unchanged."""
    import xvector_amd as xa
    m = xa.XVectorModel()
    sd = m.state_dict()
    want = synth.make_state_dict(seed=1)
    assert set(sd.keys()) == set(want.keys())
    for k, v in want.items():
        assert tuple(sd[k].shape) == tuple(np.asarray(v).shape), k
    assert sum(p.numel() for p in m.parameters()) == 5_095_503      # SURVEY.md §8a4 [probe]
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in want.items()})
    assert not m.training                                            # extraction-only: eval by default
    # constructor keeps the reference's keyword arguments (main.py:24-34)
    m2 = xa.XVectorModel(input_size=24, hidden_size=64, num_classes=10, x_vector_size=32, x_vec_extract_layer=7,
                         batch_size=512, learning_rate=0.001, batch_norm=False, dropout_p=0.0,
                         augmentations_per_sample=2, data_folder_path='data')
    assert "time_context_layers.0.norm.weight" not in m2.state_dict()
    assert m2.time_context_layers[1].context == [-2, 0, 2] and m2.time_context_layers[4].output_size == 1500


def test_no_cpu_fallback():
    import xvector_amd as xa
    m = xa.XVectorModel(hidden_size=32, num_classes=5, x_vector_size=8)
    x = torch.zeros(1, 50, 24)
    for fn in (m, m.extract_x_vec, m.stat_pool, m.time_context_layers[0]):
        with pytest.raises(RuntimeError, match="HIP device|no CPU path"):
            fn(x)
    with pytest.raises(RuntimeError):
        m.affine("segment_layer6", torch.zeros(1, 3000))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "speaker-recognition-x-vectors_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "xvector_oracle" not in text and "libxvec_oracle" not in text, f


def test_synth_is_deterministic(synth):
    a, b = synth.make_state_dict(seed=42), synth.make_state_dict(seed=42)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert float(a["segment_layer6.weight"][0, 0]) == float(b["segment_layer6.weight"][0, 0])
    assert synth.make_lengths(256).min() >= 200 and synth.make_lengths(256).max() <= 1000
    x = synth.make_mfcc(2, 300, seed=0)
    assert x.shape == (2, 300, 24) and x.dtype == np.float32


def test_csv_record_format_roundtrip(tmp_path):
    """N1: same CSV the reference writes (main.py:246-247) and reads back (main.py:276-279)."""
    from xvector_amd import extract
    vecs = torch.randn(4, 512)
    recs = extract.to_records(vecs, torch.tensor([3, 1, 4, 1]), ["a/b/1", "c/d/2", "e/f/3", "g/h/4"])
    assert [r[0] for r in recs] == ["a/b/1", "c/d/2", "e/f/3", "g/h/4"] and recs[2][1] == 4
    assert recs[0][2].dtype == np.float64 and np.array_equal(recs[0][2], vecs[0].numpy().astype(np.float64))
    path = str(tmp_path / "x_vector_test.csv")
    extract.write_x_vector_csv(recs, path)
    assert open(path).readline().strip() == ",0,1,2"
    ids, labels, back = extract.read_x_vector_csv(path)
    assert ids.tolist() == [r[0] for r in recs] and labels.tolist() == [3, 1, 4, 1]
    assert back.shape == (4, 512)
    np.testing.assert_allclose(back, vecs.numpy(), rtol=1e-6, atol=1e-7)     # numpy prints 8 digits
    np.testing.assert_array_equal(np.load(path + ".npy"), np.stack([r[2] for r in recs]))


def test_shard_bounds_and_balancing():
    from xvector_amd import extract
    for n, w in ((100000, 8), (10, 4), (7, 8), (256, 1)):
        spans = [extract.shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert extract.shard_bounds(100000, 3, 8) == (37500, 50000)
    lens = np.random.default_rng(0).integers(200, 1001, 256)
    buckets = extract.balanced_order(lens, 8)
    assert sorted(i for b in buckets for i in b) == list(range(256))
    loads = [int(lens[b].sum()) for b in buckets]
    assert max(loads) - min(loads) <= 1000


def test_lightning_checkpoint_ingestion(tmp_path, synth):
    """N2: a reference-style Lightning .ckpt (state_dict + hyper_parameters, plus harness state
    this build ignores) loads without Lightning installed."""
    import xvector_amd as xa
    hp = dict(input_size=24, hidden_size=32, num_classes=10, x_vector_size=16, x_vec_extract_layer=7,
              batch_size=512, learning_rate=0.001, batch_norm=True, dropout_p=0.0,
              augmentations_per_sample=2, data_folder_path="data")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(
        seed=3, hidden_size=32, num_classes=10, x_vector_size=16).items()}
    ckpt = {"epoch": 3, "global_step": 100, "pytorch-lightning_version": "1.6.4", "state_dict": dict(sd),
            "hyper_parameters": hp, "optimizer_states": [{}], "callbacks": {}}
    ckpt["state_dict"]["accuracy.correct"] = torch.tensor(0)      # torchmetrics state: not part of the path
    path = str(tmp_path / "last.ckpt")
    torch.save(ckpt, path)
    m = xa.XVectorModel.load_from_checkpoint(path)
    assert m.x_vec_extract_layer == 7 and m.hparams["hidden_size"] == 32
    for k, v in sd.items():
        assert torch.equal(m.state_dict()[k], v), k

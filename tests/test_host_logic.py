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


def test_csv_writer_matches_reference_file_byte_for_byte(tmp_path):
    """N1 pinned to the reference's own writer: g7_caller.npz holds the text pandas produced from the
    record list the reference accumulated (main.py:142-146, 246-247) and what the reference's reader
    (plda_score_stat.py:16-17 = main.py:276-279) parsed back.  Our writer must emit the same bytes from
    the same records, our reader the same arrays from that text."""
    from xvector_amd import extract
    g = load_golden("g7_caller.npz")
    recs = [(str(i), int(l), np.asarray(v, dtype=np.float64))
            for i, l, v in zip(g["out_ids"], g["out_labels"], g["out_vecs"])]
    path = str(tmp_path / "x_vector_test.csv")
    extract.write_x_vector_csv(recs, path, npy_sidecar=False)
    assert open(path, newline="").read() == str(g["csv_text"])
    ref_path = str(tmp_path / "ref.csv")
    with open(ref_path, "w", newline="") as f:
        f.write(str(g["csv_text"]))
    ids, labels, vecs = extract.read_x_vector_csv(ref_path)
    assert ids.tolist() == g["read_ids"].tolist() and labels.tolist() == g["out_labels"].tolist()
    assert np.array_equal(vecs, g["read_vecs"])
    # to_records on the fp32 vectors gives the reference's records (exact widening), hence the same file
    recs2 = extract.to_records(torch.from_numpy(g["out_vecs"].astype(np.float32)), torch.from_numpy(g["labels"]),
                               g["ids"].tolist())
    extract.write_x_vector_csv(recs2, path, npy_sidecar=False)
    assert open(path, newline="").read() == str(g["csv_text"])


def test_checkpoint_with_unimportable_lightning_classes(tmp_path, synth):
    """N2: a real Lightning 1.6 checkpoint pickles `hyper_parameters` as
    pytorch_lightning.utilities.parsing.AttributeDict (a dict subclass) and callback state keyed by
    classes of pytorch_lightning.callbacks; none of them is importable here (main.py:56,198,213).
    Build such a pickle from a temporary module of that name, remove the module, load."""
    import sys
    import types
    import xvector_amd as xa
    names = ["pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing",
             "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.model_checkpoint"]
    assert not any(n in sys.modules for n in names), "test needs Lightning to be absent"
    mods = {n: types.ModuleType(n) for n in names}

    class AttributeDict(dict):                      # Lightning's: a dict with attribute access
        def __getattr__(self, key):
            try:
                return self[key]
            except KeyError as exc:
                raise AttributeError(key) from exc
    AttributeDict.__module__, AttributeDict.__qualname__ = names[2], "AttributeDict"
    mods[names[2]].AttributeDict = AttributeDict

    class ModelCheckpoint:                          # object state restored through __setstate__/__dict__
        def __init__(self):
            self.best_model_score = torch.tensor(0.5)
            self.monitor = "val_step_loss"
    ModelCheckpoint.__module__, ModelCheckpoint.__qualname__ = names[4], "ModelCheckpoint"
    mods[names[4]].ModelCheckpoint = ModelCheckpoint

    hp = AttributeDict(input_size=24, hidden_size=32, num_classes=10, x_vector_size=16, x_vec_extract_layer=7,
                       batch_size=512, learning_rate=0.001, batch_norm=True, dropout_p=0.0,
                       augmentations_per_sample=2, data_folder_path="data")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(
        seed=3, hidden_size=32, num_classes=10, x_vector_size=16).items()}
    ckpt = {"epoch": 3, "global_step": 100, "pytorch-lightning_version": "1.6.4", "state_dict": dict(sd),
            "hyper_parameters": hp, "optimizer_states": [{}], "callbacks": {"ModelCheckpoint": ModelCheckpoint()},
            "hparams_name": "kwargs"}
    path = str(tmp_path / "epoch=3.ckpt")
    sys.modules.update(mods)
    try:
        torch.save(ckpt, path)
    finally:
        for n in names:
            sys.modules.pop(n, None)
    with pytest.raises(Exception):                  # the stock loader cannot resolve the classes ...
        torch.load(path, map_location="cpu", weights_only=False)
    m = xa.XVectorModel.load_from_checkpoint(path)  # ... the lenient one reads them as plain dicts
    assert m.x_vec_extract_layer == 7 and m.hparams["hidden_size"] == 32 and m.hparams["num_classes"] == 10
    for k, v in sd.items():
        assert torch.equal(m.state_dict()[k], v), k
    # overrides win over stored hyper-parameters (Lightning's load_from_checkpoint(**kwargs))
    assert xa.XVectorModel.load_from_checkpoint(path, x_vec_extract_layer=6).x_vec_extract_layer == 6


def test_call_ranges_split_a_batch_that_one_call_cannot_hold():
    """bf16x3 addresses its planes with 30-bit offsets (~1 M frames per call); the host splits the
    batch instead of surfacing XVEC_ERR_ARG.  The same for more utterances than the library's offsets ring holds."""
    import xvector_amd as xa
    R = xa.XVectorModel._call_ranges
    assert R(256, 300, None, 65535) == [(0, 256)]
    assert R(256, 300, 1_000_000, 65535) == [(0, 256)]
    assert R(5000, 300, 1_000_000, 65535) == [(0, 3333), (3333, 5000)]
    assert R(70000, 20, None, 65535) == [(0, 65535), (65535, 70000)]
    assert R(7, 400, 1000, 65535) == [(0, 2), (2, 4), (4, 6), (6, 7)]
    assert R(20000, 300, None, 65535) == [(0, 11229), (11229, 20000)]     # pooling partials < 2 GiB per call
    assert R(20000, 300, None, 65535, pool_pad=1536, partial_limit=False) == [(0, 20000)]   # large-batch partials: no such limit
    assert R(20000, 300, None, 65535, pool_pad=3072) == [(0, 5614), (5614, 11228), (11228, 16842), (16842, 20000)]   # from the model's width, not a literal
    with pytest.raises(ValueError, match="exceeds"):
        R(2, 2000, 1000, 65535)
    m = xa.XVectorModel(precision="bf16x3")
    assert 1_000_000 < m._max_frames_per_call() < (1 << 20)
    assert xa.XVectorModel(precision="bf16")._max_frames_per_call() is None


def _profile_build(got, want, what):
    msg = f"{what} was collected on {got!r}, the tree builds {want!r}: re-run profiles/run_round.sh on the shipped build"
    if os.environ.get("XVEC_STRICT_PROFILES") == "1":
        assert got == want, msg
    elif got != want:
        import warnings
        warnings.warn(msg)


def test_traffic_json_is_tied_to_the_built_library():
    """profiles/traffic.json (the PMC bytes bench.py quotes as roofline.traffic; counters cannot be read inside the bench
    process) must describe THIS build: every kernel key bench.py can look up, and every key in the file, is a kernel of the
    freshly built libxvec_hip.so (demangled symbols), the file's `source` names the newest committed profile round, and the
    library is a build of the sources in the tree (csrc/Makefile, BUILD_ID; profiles/build_id.py recomputes the hash here).
    Renaming a kernel without re-running profiles/run_round.sh turns this red.
    Whether the committed profile set was COLLECTED on exactly this build (its `build` fields) is checked strictly only with
    XVEC_STRICT_PROFILES=1 -- profiles/run_round.sh sets it after a collection -- and is a warning naming the stale id
    otherwise: a comment-only edit of a kernel source must not turn the CPU suite red until a GPU box has re-profiled it
    (ADVICE r05)."""
    import glob
    import json
    import re
    import shutil
    import subprocess
    import bench
    if shutil.which("nm") is None:
        pytest.skip("needs binutils nm")
    lib = os.path.join(ROOT, "speaker-recognition-x-vectors_amd", "libxvec_hip.so")
    syms = subprocess.run(["nm", "-C", lib], capture_output=True, text=True, check=True).stdout
    syms = syms.replace("(anonymous namespace)::", "")       # rocprofv3's names, as summarize_pmc.py keeps them, drop it too
    kernels = {m.group(1) for m in re.finditer(r"__device_stub__(\S[^\n]*?)\(", syms)}   # name incl. template arguments
    full = {re.sub(r"__device_stub__", "", m.group(0)[:-1]) for m in re.finditer(r"\S*__device_stub__[^\n]*?\(", syms)}
    assert len(kernels) > 30, "could not list the library's kernels"
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    import sys
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import build_id
    from xvector_amd import hip
    want_build = f"xvec_hip gfx950 build {build_id.source_build_id()}"
    assert hip.version() == want_build, f"the built library ({hip.version()}) is not a build of the sources in the tree ({want_build}): run build()"
    base = lambda key: key.split(" [grid")[0]          # the fp64 GEMM is listed per launch size: "name [grid N]"
    rounds = sorted(int(m.group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats*.csv"))
                    for m in [re.match(r"r(\d+)_", os.path.basename(f))] if m)
    newest = f"r{rounds[-1]:02d}"
    for dtype in ("fp32", "bf16", "bf16x3"):
        sec = tj[dtype]
        assert f"gpurun_out/{newest}" in sec["source"], f"traffic.json[{dtype}] comes from {sec['source']!r}, newest profile set is {newest}"
        _profile_build(sec.get("build"), want_build, f"traffic.json[{dtype}]")
        for pp in (True, False):
            key = bench.traffic_key(dtype, pp)
            assert any(key in k for k in full), f"bench.py's traffic key {key!r} is not a kernel of the built library"
        assert bench.traffic_key(dtype, dtype != "fp32") in sec, f"traffic.json[{dtype}] lacks the dominant kernel of the bench batch"
        for key in sec:
            if key not in ("source", "build"):
                assert any(base(key) in k for k in full), f"traffic.json[{dtype}] names {key!r}, which the built library does not contain"
    nxt = tj.get("next_rows")                 # N3 / N4: the MFCC kernel and the fp64 score GEMM
    if newest >= "r04":
        assert nxt is not None and f"gpurun_out/{newest}" in nxt["source"]
        _profile_build(nxt.get("build"), want_build, "traffic.json[next_rows]")
        assert any("mfcc512_kernel" in k for k in nxt) and any("gemm_nt_f64_kernel" in k for k in nxt)
        for key in nxt:
            if key not in ("source", "build"):
                assert any(base(key) in k for k in full), f"traffic.json[next_rows] names {key!r}, which the built library does not contain"
        if newest >= "r05":      # the [n, n] score matrix has an entry of its own, with its ratio to the algorithmic bytes
            # (64 x 64 tiles at N = 4874 since round 6: gemm_nt's rule by rounds of the block slots, csrc/score.hip)
            big = [v for k, v in nxt.items() if "gemm_nt_f64_kernel<true, 2, false>" in k or "gemm_nt_f64_kernel<true, 4" in k]
            assert big and "ratio_to_algorithmic" in big[0]

"""BASELINE configs[3] (utterance-sharded extraction + all-gather of the embeddings) and the next
row N2 (Lightning checkpoint ingestion) on the one GPU a test box has: the sharded job with the
REAL extractor, the balanced (ragged) job, and the RCCL leg itself with a one-rank communicator.
The reference has no counterpart (main.py:220: devices=[0]); the numerics bar is the path's own
(bit-for-bit against direct calls of the same kernels)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_parity, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_extract_sharded_with_the_real_extractor(gpu_model, synth):
    """extract.extract_sharded at world 1, N=1000 in batches of 256 (last one partial): equals the
    direct calls bit for bit, rows in input order."""
    from xvector_amd import extract
    n = 1000
    xs = torch.from_numpy(synth.make_mfcc(n, 300, seed=90)).to(DEV)
    got = extract.extract_sharded(gpu_model.extract_x_vec, lambda lo, hi: xs[lo:hi], n, batch_size=256)
    assert got.shape == (n, 512)
    want = torch.cat([gpu_model.extract_x_vec(xs[i:i + 256]) for i in range(0, n, 256)])
    assert torch.equal(got, want)
    # a row does not depend on how the job was batched (utterance independence)
    assert_parity(got[777:778], gpu_model.extract_x_vec(xs[777:778]), 1e-5, "row 777 alone")


def test_balanced_ragged_job_with_real_embeddings(gpu_model, synth):
    """balanced_order + gather_embeddings(order=...) with real embeddings: the job is cut into W
    frame-balanced shards exactly as W ranks would take them; each shard is extracted as its own
    padded batch + lengths; the assembly puts every row back in input order."""
    from xvector_amd import extract
    lens = synth.make_lengths(48)
    T = int(lens.max())
    x = torch.from_numpy(synth.make_mfcc(48, T, seed=91)).to(DEV)
    direct = gpu_model.extract_x_vec(x, lengths=lens.tolist())
    for world in (1, 3, 8):
        order = extract.balanced_order(lens.tolist(), world)
        loads = [int(lens[o].sum()) for o in order]
        assert max(loads) - min(loads) <= 1000
        shards = [gpu_model.extract_x_vec(x[o], lengths=lens[o].tolist()) for o in order]
        # what all_gather_into_tensor delivers: equal-size (padded) shards back to back
        per = max(len(o) for o in order)
        recv = torch.zeros((world * per, 512), device=DEV)
        for r, s in enumerate(shards):
            recv[r * per: r * per + s.shape[0]] = s
        out = torch.empty((48, 512), device=DEV)
        for r, o in enumerate(order):
            out[torch.as_tensor(o, device=DEV)] = recv[r * per: r * per + len(o)]
        assert_parity(out, direct, 1e-5, f"balanced job, {world} shards")
    # the product's single-rank form of the same call
    got = extract.extract_balanced(lambda idx: gpu_model.extract_x_vec(x[idx], lengths=lens[idx].tolist()),
                                   lens.tolist())
    assert_parity(got, direct, 1e-5, "extract_balanced world 1")


_RCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "{port}")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)       # before any other GPU work, as bench.py does
import xvector_amd as xa
sd = {{k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}}
model = xa.XVectorModel(); model.load_state_dict(sd); model = model.to(dev).eval()
B, K = 64, 3
x = torch.from_numpy(xa.synth.make_mfcc(B, 300, seed=5)).to(dev)
emb = torch.empty((K * B, 512), device=dev)
gathered = torch.empty((1 * K * B, 512), device=dev)
for k in range(K):
    emb[k * B:(k + 1) * B] = model.extract_x_vec(x)
dist.all_gather_into_tensor(gathered, emb)            # bench.py's collective leg, world 1
torch.cuda.synchronize(dev)
assert torch.equal(gathered, emb)
# the product's job entry point with the collective forced
xs = torch.from_numpy(xa.synth.make_mfcc(150, 300, seed=6)).to(dev)
full = xa.extract.extract_sharded(model.extract_x_vec, lambda lo, hi: xs[lo:hi], 150, batch_size=64,
                                  force_collective=True)
want = torch.cat([model.extract_x_vec(xs[i:i + 64]) for i in range(0, 150, 64)])
assert torch.equal(full, want)
lens = xa.synth.make_lengths(12)
xr = torch.from_numpy(xa.synth.make_mfcc(12, int(lens.max()), seed=7)).to(dev)
got = xa.extract.extract_balanced(lambda idx: model.extract_x_vec(xr[idx], lengths=lens[idx].tolist()), lens.tolist(),
                                  force_collective=True)
assert torch.equal(got, model.extract_x_vec(xr, lengths=lens.tolist()))
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", dist.is_nccl_available())
"""


def test_rccl_all_gather_single_rank():
    """The RCCL calls of bench.py / extract.gather_embeddings executed for real: a one-rank
    "nccl" process group (initialised before any other GPU work, in its own process) runs
    all_gather_into_tensor on the embeddings and the forced-collective job entry points."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    port = 32100 + os.getpid() % 1500
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT.format(root=ROOT, port=port)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_multi_rank_code_path_world1(tmp_path):
    """bench.py launched the way the driver launches N>1 (torch.distributed.run), with one rank:
    init_process_group("nccl"), barriers, the in-region all-gather and the MAX all-reduce all run."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    port = 33700 + os.getpid() % 1500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--cpu-budget", "0", "--force-collective"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and "all-gather" in out["config"]["sharding"]
    # attribution: the rank's own time and the collective's, by events on the launch stream
    assert 0 < out["config"]["per_rank_compute_s"]["min"] <= out["config"]["per_rank_compute_s"]["max"] < out["steps"] * out["ms_per_step"] * 1e-3 * 1.05
    assert out["config"]["all_gather_ms"]["max"] > 0


# ------------------------------------------------------------------------------- N2 on the GPU
def test_checkpoint_to_gpu_matches_reference_outputs(tmp_path, sd42, synth):
    """N2 end to end: a Lightning-style .ckpt holding the seed-42 weights -> load_from_checkpoint
    -> .to(cuda) -> HIP path == the REFERENCE's outputs for those weights (fixture g4,
    main.py:56,198,213)."""
    import xvector_amd as xa
    g = load_golden("g4_full.npz")
    hp = dict(input_size=24, hidden_size=512, num_classes=1211, x_vector_size=512, x_vec_extract_layer=6,
              batch_size=512, learning_rate=0.001, batch_norm=True, dropout_p=0.0, augmentations_per_sample=2,
              data_folder_path="data")
    sd = dict(sd42)
    sd["accuracy.correct"] = torch.tensor(0)            # torchmetrics state rides along in real checkpoints
    path = str(tmp_path / "last.ckpt")
    torch.save({"epoch": 9, "global_step": 7830, "pytorch-lightning_version": "1.6.4", "state_dict": sd,
                "hyper_parameters": hp, "optimizer_states": [{}], "lr_schedulers": [], "callbacks": {}}, path)
    m = xa.XVectorModel.load_from_checkpoint(path).to(DEV)
    for B, T in ((8, 300), (1, 299)):
        key = f"B{B}_T{T}"
        x = torch.from_numpy(synth.make_mfcc(B, T, seed=int(g[key + "_seed_x"]))).to(DEV)
        assert_parity(m.extract_x_vec(x), g[key + "_xvec6"], 1e-4, "ckpt xvec6")
        assert_parity(m(x), g[key + "_logits"], 1e-4, "ckpt logits")
    m7 = xa.XVectorModel.load_from_checkpoint(path, x_vec_extract_layer=7, precision="bf16x3").to(DEV)
    x = torch.from_numpy(synth.make_mfcc(8, 300, seed=int(g["B8_T300_seed_x"]))).to(DEV)
    assert_parity(m7.extract_x_vec(x), g["B8_T300_xvec7"], 1e-4, "ckpt xvec7 (bf16x3)")


# ------------------------------------------------------------------------------- scratch hygiene
@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_stale_workspace_contents_never_leak(sd42, synth, precision):
    """The scratch buffer is uninitialised memory shared across calls and precisions.  Rows of the
    last 32-row group past the batch's last frame are computed from whatever it holds; the fused
    pooling must SELECT its rows, not weight them (0 * Inf = NaN).  Poison every byte with Inf / NaN
    bit patterns of each element type, run ragged and fixed batches: same bits as the clean run."""
    import xvector_amd as xa
    m = xa.XVectorModel(precision=precision)
    m.load_state_dict(sd42)
    m = m.to(DEV)
    lens = [300, 271, 299, 15, 16, 203, 47, 290]        # total rows not a multiple of 32 at any layer
    x = torch.from_numpy(synth.make_mfcc(8, 300, seed=92)).to(DEV)
    clean_r = m.extract_x_vec(x, lengths=lens)
    clean_f = m.extract_x_vec(x[:, :299])
    eng = m._engine(torch.device(DEV))
    ws = eng.workspace
    assert ws is not None
    for pattern, dt in ((0x7F80, torch.int16), (-128, torch.int16), (0x7F800000, torch.int32), (-1, torch.int32),
                        (0x7FC00000, torch.int32)):
        n = ws.numel() // (2 if dt == torch.int16 else 4)
        ws.view(dt)[:n].fill_(pattern)                  # bf16 +Inf, bf16 -Inf/NaN-ish 0xFF80, fp32 +Inf, all ones (NaN), quiet NaN
        got_r = m.extract_x_vec(x, lengths=lens)
        assert eng.workspace is ws
        ws.view(dt)[:n].fill_(pattern)
        got_f = m.extract_x_vec(x[:, :299])
        # utterance 3 has 15 frames: one pooled frame -> NaN std in the reference too (main.py:61)
        ok = [0, 1, 2, 4, 5, 6, 7]
        assert torch.isfinite(got_r[ok]).all() and torch.isfinite(got_f).all(), hex(pattern & 0xFFFFFFFF)
        assert torch.equal(got_r[ok], clean_r[ok]) and torch.equal(got_f, clean_f), hex(pattern & 0xFFFFFFFF)


def test_graphed_path_owns_its_scratch(gpu_model, synth):
    """A captured graph holds raw pointers into its scratch buffer.  Growing the engine's shared
    workspace afterwards (a larger eager batch) frees the old block; a graph that pointed into it
    would scribble over whatever the allocator hands that memory to next."""
    x = torch.from_numpy(synth.make_mfcc(4, 300, seed=11)).to(DEV)
    eager = gpu_model.extract_x_vec(x)
    import xvector_amd as xa
    m = xa.XVectorModel()
    m.load_state_dict(gpu_model.state_dict())
    m = m.to(DEV)
    gp = m.graphed(x)
    eng = m._engine(torch.device(DEV))
    before = None if eng.workspace is None else eng.workspace.data_ptr()
    big = torch.from_numpy(synth.make_mfcc(96, 300, seed=12)).to(DEV)
    m.extract_x_vec(big)                                # (re)allocates the shared workspace
    assert eng.workspace is not None and eng.workspace.data_ptr() != gp._ws.data_ptr()
    torch.cuda.synchronize()
    # fill the allocator's free blocks with a sentinel, replay, check nothing was overwritten
    canaries = [torch.full((1 << 22,), 7.0, device=DEV) for _ in range(8)]
    out = gp(x).clone()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    assert all(bool((c == 7.0).all()) for c in canaries)
    del before


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
def test_batches_larger_than_one_call_are_split(precision, monkeypatch):
    """A batch that exceeds what one library call can carry (bf16x3: 30-bit plane offsets; any mode: 65 535
    utterances) is cut into several calls by the host; results equal the unsplit call's to rounding."""
    import numpy as np
    import torch
    import xvector_amd as xa
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
    m = xa.XVectorModel(precision=precision)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = torch.from_numpy(xa.synth.make_mfcc(11, 120, seed=5)).to(dev)
    lens = [120, 60, 90, 33, 120, 47, 101, 120, 15 + 1, 80, 64]
    whole, whole_r, whole_logits = m.extract_x_vec(x), m.extract_x_vec(x, lengths=lens), m(x)
    monkeypatch.setattr(xa.XVectorModel, "MAX_UTTS_PER_CALL", 4)
    if precision == "bf16x3":
        monkeypatch.setattr(xa.XVectorModel, "_max_frames_per_call", lambda self: 3 * 120)
    split, split_r = m.extract_x_vec(x), m.extract_x_vec(x, lengths=lens)
    # same arithmetic per frame; the pooling partials are cut at 32-row groups of the CALL's row layout, so the
    # fp32 summation order over an utterance's frames differs between the two: rounding-level differences only
    def same(a, b):
        return a.shape == b.shape and float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    assert same(split, whole) and same(split_r, whole_r) and same(m(x), whole_logits)

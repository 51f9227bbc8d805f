"""bench.py prints ONE JSON line with the fields the driver and the judge read (task contract ④)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--dtype", "bf16"], ["--workload", "ragged"]])
def test_bench_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-budget", "1"] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert d["dtype"] == ("bf16" if "bf16" in extra else "f32")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 256 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 0.01     # value = units / time
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "embeddings/s" and c["sample"]
    # SURVEY 8(d) protocol: B=1 and B=64, all usable cores and one thread
    assert set(c["legs"]) == {"b64_all", "b1_all", "b64_1t", "b1_1t"}
    assert c["legs"]["b64_1t"]["threads"] == 1 and c["legs"]["b1_all"]["batch"] == 1
    # the headline CPU figure is the FASTEST honest leg (B=64 on all cores was the slowest per utterance)
    assert c["value"] == max(l["embeddings_per_s"] for l in c["legs"].values())
    assert c["value"] == c["legs"][c["leg"]]["embeddings_per_s"] and c["cores"] == c["legs"][c["leg"]]["threads"]
    cfg = d["config"]
    if not extra:      # the default line carries the other single-GPU configs as secondary fields (never `value`)
        for key in ("bf16_embeddings_per_s", "bf16_roofline_frac", "bf16x3_embeddings_per_s", "ragged_utt_per_s", "ragged_valid_frames_per_s"):
            assert isinstance(cfg[key], float) and cfg[key] > 0, key
        assert cfg["bf16_embeddings_per_s"] > cfg["bf16x3_embeddings_per_s"] > d["value"]   # bf16 matrix rate is 16x the fp32 one
        # next rows N3 / N4 in the driver's own line
        for key in ("wave_utt_per_s", "wave_bf16_utt_per_s", "mfcc_us_per_batch", "plda_score_ms_n4874",
                    # round 6: configs[2] x [4], configs[3] x [4] at its own size, host waveforms -> bf16
                    "ragged_bf16_utt_per_s", "ragged_bf16_valid_frames_per_s", "job100k_bf16_embeddings_per_s",
                    "wave_bf16_pcie_inclusive_utt_per_s", "wave_bf16_pcie_inclusive_f32_samples_utt_per_s", "plda_dense_score_ms_n4874"):
            assert isinstance(cfg[key], float) and cfg[key] > 0, key
        assert cfg["ragged_bf16_utt_per_s"] > cfg["ragged_utt_per_s"] and cfg["job100k_bf16_embeddings_per_s"] > cfg["job100k_embeddings_per_s"]
        assert cfg["plda_score_ms_n4874"] < cfg["plda_dense_score_ms_n4874"]          # rank 200 of 512
        assert cfg["wave_bf16_utt_per_s"] > cfg["wave_utt_per_s"] and cfg["mfcc_us_per_batch"] < 500 and cfg["plda_score_ms_n4874"] < 20
        assert cfg["mfcc_kernel_form"] == 2          # the reference's MFCC call runs the nfft-512 kernel with the banded filterbank
    else:
        assert "bf16_embeddings_per_s" not in cfg


def _one_line(cmd, timeout=600):
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,scaling", [([], "weak"), (["--workload", "job", "--utterances", "37"], "strong")])
def test_plain_python_launch_of_two_ranks_dry_run(extra, scaling):
    """`python bench.py --gpus 2` WITHOUT a launcher (how a driver may call it): bench.py starts the two ranks itself
    as a child torch.distributed.run, passes rank 0's line through and exits with its code.  On CPU the ranks
    rehearse the control flow only (--dry-run: gloo, constant embeddings)."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--batch", "8"] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env_clean)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["data"] == "dry-run" and d["scaling"] == scaling and d["value"] > 0
    # attribution fields of a multi-rank line: one compute time per rank, the collective's span
    pr = d["config"]["per_rank_compute_s"]
    assert len(pr["ranks"]) == 2 and pr["min"] == min(pr["ranks"]) and pr["max"] == max(pr["ranks"]) and pr["min"] >= 0
    assert d["config"]["all_gather_ms"]["max"] >= d["config"]["all_gather_ms"]["min"] >= 0


def test_self_launch_propagates_failure():
    """A rank that dies must fail the bench, not print a line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "job", "--utterances", "0"]
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env_clean)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_self_launch_on_the_gpu_with_the_collective():
    """The self-launch path with real kernels and a real (one-rank) RCCL group: `--gpus 1 --force-collective` run
    through the launcher child exactly as N>1 would be."""
    from bench import self_launch  # noqa: F401  (importable without side effects)
    cmd = [sys.executable, "-c",
           "import sys, bench; sys.exit(bench.self_launch(1, ['--gpus','1','--force-collective','--steps','3','--warmup','1',"
           "'--cpu-budget','0','--preroll','0.1']))"]
    d = _one_line(cmd)
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and "all-gather" in d["config"]["sharding"] and d["value"] > 0
    # attribution of a --gpus N line: every rank's own time and the collective's, one entry per rank
    assert d["config"]["per_rank_compute_s"]["max"] >= d["config"]["per_rank_compute_s"]["min"] > 0
    assert len(d["config"]["per_rank_compute_s"]["ranks"]) == 1 and d["config"]["all_gather_ms"]["max"] > 0


@pytest.mark.gpu
def test_job_workload_on_the_gpu():
    """BASELINE configs[3] in its job form inside the bench: extract_sharded over generated batches, one all-gather
    (forced on the single rank), strong scaling."""
    d = _one_line([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "job", "--utterances", "2000",
                   "--force-collective", "--cpu-budget", "0"])
    assert d["scaling"] == "strong" and d["value"] > 0 and d["n_gpus"] == 1
    assert "configs[3]" in d["config"]["workload"] and "2000 utterances" in d["config"]["workload"]


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_torchrun_form_sets_the_ipc_variable_in_every_rank():
    """The contract's OTHER launch form -- `python -m torch.distributed.run ... bench.py --gpus 2` -- with
    HSA_ENABLE_IPC_MODE_LEGACY absent from the parent's environment: every rank must still see it as "0" (RCCL across
    processes fails at communicator creation without it on this driver; VERDICT r03: only self_launch() set it)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2",
           "--batch", "4", "--report-env"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    env_lines = [l for l in out.stderr.splitlines() if l.startswith("ENV ")]
    assert len(env_lines) == 1, out.stderr[-2000:]
    assert json.loads(env_lines[0][4:]) == {"HSA_ENABLE_IPC_MODE_LEGACY": ["0", "0"]}
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


@pytest.mark.parametrize("form", ["self", "torchrun"])
def test_a_rank_that_dies_before_the_first_barrier_ends_the_job(form):
    """One rank raises before the first barrier: the job exits non-zero well inside the process-group timeout instead
    of leaving its peer in the barrier for torch's default 10 minutes, and no JSON line is printed."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    tail = ["--gpus", "2", "--dry-run", "--steps", "2", "--batch", "4", "--fail-rank", "1", "--dist-timeout", "20"]
    if form == "self":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + tail
    t0 = time.time()
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0, out.stdout[-500:]
    assert time.time() - t0 < 120, "the surviving rank was not released by the timeout / the launcher"
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "simulated rank failure" in out.stderr

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
    assert c["value"] == c["legs"]["b64_all"]["embeddings_per_s"] and c["cores"] == c["legs"]["b64_all"]["threads"]
    cfg = d["config"]
    if not extra:      # the default line carries the other single-GPU configs as secondary fields (never `value`)
        for key in ("bf16_embeddings_per_s", "bf16_roofline_frac", "ragged_utt_per_s", "ragged_valid_frames_per_s"):
            assert isinstance(cfg[key], float) and cfg[key] > 0, key
        assert cfg["bf16_embeddings_per_s"] > d["value"]          # bf16 matrix rate is 16x the fp32 one
    else:
        assert "bf16_embeddings_per_s" not in cfg

#!/usr/bin/env python3
"""profiles/traffic.json from the per-arithmetic PMC summaries (rocprofv3 FETCH_SIZE / WRITE_SIZE passes):
    python profiles/make_traffic.py fp32=gpurun_out/pmc_<tag> bf16=gpurun_out/pmc_<tag> > profiles/traffic.json
bench.py reads roofline.traffic of its dominant kernel from here (counters cannot be collected from inside it)."""
import json, subprocess, sys, os, tempfile
here = os.path.dirname(os.path.abspath(__file__))
out = {}
for arg in sys.argv[1:]:
    name, root = arg.split("=", 1)
    with tempfile.NamedTemporaryFile(suffix=".json") as f:
        what = "profiles/diag/next_rows_pmc.py" if name == "next_rows" else "bench.py --steps 3 --warmup 1"
        subprocess.run([sys.executable, os.path.join(here, "summarize_pmc.py"), root, f.name, what], check=True, stdout=subprocess.DEVNULL)
        out[name] = json.load(open(f.name))
# which build the counters belong to (tests/test_host_logic.py compares it with the built library's xvec_version())
sys.path.insert(0, os.path.dirname(here))
import xvector_amd as xa
for sec in out.values():
    sec["build"] = xa.hip.version()
# next rows: bytes per launch against the algorithmic bytes of the workload profiles/diag/next_rows_pmc.py runs
nxt = out.get("next_rows")
if nxt:
    n, dim = 4874, 512
    alg = {"mfcc512_kernel": 256 * (48000 * 4 + 299 * 24 * 4),              # 256 waveforms of 3 s in, [256, 299, 24] out
           "gemm_nt_f64_kernel<true, 4> [grid 512]": (2 * n * dim + n * n) * 8,   # the [n, n] score matrix: two operands in, scores out
           "gemm_nt_f64_kernel<true, 2> [grid 512]": (n * dim + 2 * dim * dim + 2 * n * dim) * 8}   # [e Psi | e Phi]: e and the stacked matrices in, [n, 2 dim] out
    for key, ent in nxt.items():
        for frag, b in alg.items():
            if isinstance(ent, dict) and frag in key:
                ent["algorithmic_bytes"] = b
                ent["ratio_to_algorithmic"] = round(ent["hbm_bytes_per_launch"] / b, 3)
json.dump(out, sys.stdout, indent=1)

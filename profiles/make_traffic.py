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
    n, dim, R = 4874, 512, 200             # the PLDA model of next_rows_pmc.py: rank 200 of 512, the low-rank scorer (round 6)
    alg = {"mfcc512_kernel": 256 * (48000 * 4 + 299 * 24 * 4),              # 256 waveforms of 3 s in, [256, 299, 24] out
           # the [n, n] score matrix over the upper triangle of tiles: y W and y in (K = R), every score out (64 x 64 tiles, four
           # blocks per CU, at this size since round 6: 780 tiles of 128 x 128 fill 512 block slots one and a half times)
           "gemm_nt_f64_kernel<true, 2, false> [grid 1024]": (2 * n * R + n * n) * 8,
           # y = (x - mean) L: x and L^T in, y out (77 x 4 tiles of 64 x 64)
           "gemm_nt_f64_kernel<true, 2, true> [grid 308]": (n * dim + R * dim + n * R) * 8,
           # [y W | row dots of y (-Z) with y]: y and the stacked [2 R, R] matrices in, y W and seven partials per row out
           "gemm_nt_f64_kernel<true, 2, true> [grid 539]": (n * R + 2 * R * R + n * R + 7 * n) * 8}
    for key, ent in nxt.items():
        for frag, b in alg.items():
            if isinstance(ent, dict) and frag in key:
                ent["algorithmic_bytes"] = b
                ent["ratio_to_algorithmic"] = round(ent["hbm_bytes_per_launch"] / b, 3)
json.dump(out, sys.stdout, indent=1)

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
json.dump(out, sys.stdout, indent=1)

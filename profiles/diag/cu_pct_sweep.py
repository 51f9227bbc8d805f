"""Diagnostic: per-layer times of the bf16 path with the large-batch kernel's grid covering a percentage of the CUs
(XVEC_PP_CU_PCT, read at handle creation): fewer, longer row ranges -> taller tiles on fewer CUs."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = os.path.join(root, "profiles", "diag", "pp_knock.py")
for rnd in range(2):
    for pct in (sys.argv[1:] or ["100", "75", "67", "50"]):
        env = dict(os.environ); env["XVEC_PP_CU_PCT"] = pct
        out = subprocess.run([sys.executable, child], env=env, capture_output=True, text=True, timeout=300)
        print(pct, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)

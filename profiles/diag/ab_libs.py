"""Interleaved A/B of two builds of the library on one GPU: python profiles/diag/ab_libs.py libA.so libB.so [rounds]
(each round = a fresh subprocess per library running the bf16 path 60 times; reports per-layer medians)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import os, sys, json, time
sys.path.insert(0, %r)
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision=os.environ.get("AB_DTYPE", "bf16")); m.load_state_dict(sd); m = m.to(dev).eval()
x = torch.from_numpy(xa.synth.make_mfcc(256, 300, seed=0)).to(dev)
for _ in range(15): m.extract_x_vec(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(60): m.extract_x_vec(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
m.set_profiling(True); acc = {}
for _ in range(20):
    m.extract_x_vec(x)
    for k, v in m.timings_ms().items(): acc[k] = acc.get(k, 0) + v / 20
print(json.dumps({"ms": dt * 1e3, **{k: v for k, v in acc.items() if v > 0}}))
''' % root
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ); env["XVEC_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", out.stderr[-500:]); continue
        res[l].append(json.loads(line[-1]))
for l in libs:
    if not res[l]: continue
    keys = res[l][0].keys()
    med = {k: sorted(d[k] for d in res[l])[len(res[l]) // 2] for k in keys}
    print(os.path.basename(l), {k: round(v, 4) for k, v in med.items()}, f"-> {256 / med['ms'] * 1e3:.0f} emb/s")

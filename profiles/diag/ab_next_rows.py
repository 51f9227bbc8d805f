"""Interleaved A/B of two library builds on one GPU, next rows N3 / N4: python profiles/diag/ab_next_rows.py libA.so libB.so [rounds]
Per round and library a fresh subprocess: the MFCC kernel on 256 x 3 s (us per batch, 100 launches), the PLDA score matrix of
4874 x-vectors and the score GEMM alone (ms, median of 20 by hipEvents)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch
import xvector_amd as xa
from xvector_amd import scoring
dev = torch.device("cuda:0")
fe = xa.MfccFrontEnd(device=dev)
w = 0.1 * torch.randn(256, 48000, device=dev)
for _ in range(20): fe(w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): fe(w)
e1.record(); torch.cuda.synchronize()
mf = e0.elapsed_time(e1) / 100 * 1e3
mean, F, Sigma = xa.synth.make_plda(512, 200, seed=21)
sc = scoring.PldaScorer(mean, F, Sigma, device=dev)
x = torch.randn(4874, 512, device=dev).double() + torch.from_numpy(mean).to(dev)
def ev(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
print(json.dumps({"mfcc_us": mf, "plda_ms": ev(lambda: sc.score(x)), "gemm_ms": ev(lambda: scoring.gemm_nt(x, x)), "build": xa.hip.version()}))
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
            print(l, "FAILED", out.stderr[-800:]); continue
        res[l].append(json.loads(line[-1]))
        print(os.path.basename(l), res[l][-1], flush=True)
for l in libs:
    if res[l]:
        print(os.path.basename(l), "median:", {k: round(sorted(d[k] for d in res[l])[len(res[l]) // 2], 4) for k in ("mfcc_us", "plda_ms", "gemm_ms")})

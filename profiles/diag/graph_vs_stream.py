"""Whole path at the bench size: stream launches vs ONE hipGraph replay per batch (model.graphed).
python profiles/diag/graph_vs_stream.py [precision]"""
import os, sys, json, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision=prec); m.load_state_dict(sd); m = m.to(dev).eval()
x = torch.from_numpy(xa.synth.make_mfcc(256, 300, seed=0)).to(dev)
g = m.graphed(x)
def run(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t_end = time.time() + 0.5
while time.time() < t_end: m.extract_x_vec(x)
res = {"stream": [], "graph": []}
n = 200 if prec != "fp32" else 40
for r in range(4):
    res["stream"].append(round(run(lambda: m.extract_x_vec(x), n), 4))
    res["graph"].append(round(run(lambda: g._graph.replay(), n), 4))
assert torch.equal(g._y, m.extract_x_vec(x))
print(json.dumps({"precision": prec, **res}))

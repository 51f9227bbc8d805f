"""Diagnostic (DIAG=1 build of libxvec_hip.so only): per-block prologue / main-loop / epilogue
cycles from in-kernel s_memtime stamps (wave 0 of each block), one TDNN layer at B=256, T=300."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
from xvector_amd import hip
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(); m.load_state_dict(sd); m = m.to("cuda:0")
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(256, 300, 24 if layer == 0 else 512, device="cuda:0")
for _ in range(3):
    y = m.time_context_layers[layer](x)
torch.cuda.synchronize()
nblk = 600 * (12 if layer == 4 else 4)
n = 8 * min(nblk, 8192)
buf = (C.c_ulonglong * n)()
hip.lib.xvec_diag_read.argtypes = [C.c_void_p, C.c_int]
assert hip.lib.xvec_diag_read(buf, n) == 0
d = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
nch = d[0, 5]
print("blocks", len(d), "chunks", nch)
for i, nme in enumerate(["prologue", "loop", "epilogue"]):
    v = d[:, i]
    print(f"{nme:9s} cycles: mean {v.mean():10.1f}  p10 {np.percentile(v,10):10.1f}  p50 {np.percentile(v,50):10.1f} p90 {np.percentile(v,90):10.1f}")
print("loop per chunk: mean %.1f" % (d[:, 1].mean() / nch), " (64 MFMAs = 4096 pipe cycles)")
t0, t1 = d[:, 3], d[:, 4]
span = t1.max() - t0.min()
print("kernel span (s_memtime ticks): %.0f ; sum of block times / span = %.1f concurrent blocks" % (span, (t1 - t0).sum() / span))
# how the tail looks: number of blocks running over time (20 bins)
edges = np.linspace(t0.min(), t1.max(), 21)
occ = [(np.minimum(t1, edges[i + 1]) - np.maximum(t0, edges[i])).clip(0).sum() / (edges[i + 1] - edges[i]) for i in range(20)]
print("resident blocks per 5% time bin:", " ".join(f"{o:.0f}" for o in occ))

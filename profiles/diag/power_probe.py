"""Is the bf16 path power-limited?  Same kernels, same schedule, same memory traffic -- only the VALUES differ:
random MFCCs and weights (as benchmarked) against all-zero inputs (no toggling in the matrix pipe, LDS and fabric).
usage: python profiles/diag/power_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision=sys.argv[1] if len(sys.argv) > 1 else "bf16"); m.load_state_dict(sd); m = m.to(dev).eval()
sd0 = {k: torch.zeros_like(v) if v.dtype.is_floating_point and "running_var" not in k else v for k, v in sd.items()}
m0 = xa.XVectorModel(precision=m.precision); m0.load_state_dict(sd0); m0 = m0.to(dev).eval()
xr = torch.from_numpy(xa.synth.make_mfcc(256, 300, seed=0)).to(dev)
xz = torch.zeros_like(xr)
def t(model, x, n=300):
    for _ in range(100): model.extract_x_vec(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): model.extract_x_vec(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for r in range(3):
    a, b, c = t(m, xr), t(m, xz), t(m0, xz)
    print(f"round {r}: random weights + random input {a:.4f} ms ({256 / a:.0f} k emb/s) | random weights, zero input {b:.4f} ms "
          f"({256 / b:.0f} k) | zero weights, zero input {c:.4f} ms ({256 / c:.0f} k)", flush=True)

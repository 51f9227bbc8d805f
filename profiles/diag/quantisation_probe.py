"""fp32 layers 2-4 at batch sizes whose 32-row groups do / do not divide among the 128 row ranges of a column:
TFLOP/s per layer from the library's events.  python profiles/diag/quantisation_probe.py"""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision="fp32"); m.load_state_dict(sd); m = m.to(dev).eval()
T = 300
out = {}
for B in (256, 512, 1024):
    x = torch.from_numpy(xa.synth.make_mfcc(B, T, seed=0)).to(dev)
    for _ in range(10): m.extract_x_vec(x)
    m.set_profiling(True); acc = {}
    n = 10
    for _ in range(n):
        m.extract_x_vec(x)
        for k, v in m.timings_ms().items(): acc[k] = acc.get(k, 0) + v / n
    m.set_profiling(False)
    fl = {"tdnn2": 2 * 512 * 1536 * (T - 8) * B, "tdnn3": 2 * 512 * 1536 * (T - 14) * B, "tdnn4": 2 * 512 * 512 * (T - 14) * B,
          "tdnn5_pool": 2 * 1500 * 512 * (T - 14) * B}
    rows = {"tdnn2": (T - 8) * B, "tdnn3": (T - 14) * B, "tdnn4": (T - 14) * B, "tdnn5_pool": (T - 14) * B}
    out[B] = {k: {"tflops": round(fl[k] / acc[k] / 1e9, 1), "groups_per_block": round(-(-rows[k] // 32) / (128 if k != "tdnn5_pool" else 512 / 12), 2)} for k in fl}
print(json.dumps(out))

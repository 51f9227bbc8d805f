"""Diagnostic (DIAG=1 build of libxvec_hip.so only): per-block K-loop / epilogue cycles from
in-kernel s_memtime stamps (wave 0 of each block), one TDNN layer at B=256, T=300."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
from xvector_amd import hip
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
m = xa.XVectorModel(precision=prec); m.load_state_dict(sd); m = m.to("cuda:0")
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T_in = [300, 296, 292, 286, 286][layer]
x = torch.randn(256, T_in, 24 if layer == 0 else 512, device="cuda:0")
for _ in range(int(os.environ.get("STAMP_ITERS", "3"))):      # STAMP_ITERS=200: past the clock ramp of a cold chip
    if layer == 4 and len(sys.argv) > 3:      # whole path: the stamps left are layer 5's fused-pooling launch
        y = m.extract_x_vec(torch.randn(256, 300, 24, device="cuda:0"))
    else:
        y = m.time_context_layers[layer](x)
torch.cuda.synchronize()
nblk = 512 if layer < 4 else 504
buf = (C.c_ulonglong * (8 * nblk))()
hip.lib.xvec_diag_read.argtypes = [C.c_void_p, C.c_int]
assert hip.lib.xvec_diag_read(buf, 8 * nblk) == 0
d = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
total = d[:, 1] - d[:, 0]
tiles = d[:, 5]
print(f"{prec} layer {layer + 1}: blocks {len(d)}  groups/block {d[:,2].mean():.2f}  tiles/block {tiles.mean():.2f}")
print(f"block lifetime cycles: mean {total.mean():.0f} max {total.max():.0f}")
print(f"K-loop per tile: {(d[:,3]/tiles).mean():.0f}   epilogue per tile: {(d[:,4]/tiles).mean():.0f}   "
      f"outside (prologue, tile switch, waits): {((total - d[:,3] - d[:,4])/tiles).mean():.0f} per tile")
real = d[:, 7] - d[:, 6]
print(f"core clock during the launch (s_memtime cycles per s_memrealtime tick of 10 ns): {np.median(total / (real / 100.0)) / 1e3:.3f} GHz "
      f"(block lifetime {np.median(real) / 100.0:.1f} us)")
print(f"shares: loop {d[:,3].sum()/total.sum()*100:.1f}%  epilogue {d[:,4].sum()/total.sum()*100:.1f}%  other {(1-(d[:,3].sum()+d[:,4].sum())/total.sum())*100:.1f}%")
# first-slot (b < grid/2) vs second-slot blocks of a CU pair: start/end relative to the earliest start
h = len(d) // 2
t0 = d[:, 0].min()
for name, sl in (("first-slot blocks", slice(0, h)), ("second-slot blocks", slice(h, None))):
    print(f"{name}: start {np.mean(d[sl, 0] - t0):.0f}  end mean {np.mean(d[sl, 1] - t0):.0f} max {np.max(d[sl, 1] - t0):.0f}  "
          f"lifetime {np.mean(total[sl]):.0f}  groups {d[sl, 2].mean():.2f}")
q = np.percentile(d[:, 1] - t0, [5, 25, 50, 75, 95, 100])
print("end-time percentiles 5/25/50/75/95/100:", " ".join(f"{v:.0f}" for v in q))

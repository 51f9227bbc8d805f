"""per-layer bf16 output of the 128x128 kernel against the fp32 layer: where are the wrong values?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
os.environ["XVEC_PP"] = "0"
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m32 = xa.XVectorModel(precision="fp32"); m32.load_state_dict(sd); m32 = m32.to(dev).eval()
m16 = xa.XVectorModel(precision="bf16"); m16.load_state_dict(sd); m16 = m16.to(dev).eval()
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 100
x = torch.from_numpy(xa.synth.make_mfcc(B, T, seed=0)).to(dev)
h = x
for i in range(2):
    ref = m32.time_context_layers[i](h)
    o = m16.time_context_layers[i](h)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(o) | ((o - ref).abs() > 0.05 * ref.abs().max())
    print(f"layer {i}: shape {tuple(o.shape)} bad {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        rows = bad.reshape(-1, bad.shape[-1])
        print("  bad per row (first 40 rows):", rows.sum(1)[:40].tolist())
        print("  bad per column (first 64):", rows.sum(0)[:64].tolist())
        br = rows.any(1).nonzero().flatten()
        print("  bad rows:", br[:48].tolist(), "... n =", len(br))
        print("  bad rows mod 128:", sorted(set((br % 128).tolist()))[:64])
        print("  cols bad in first bad row:", rows[br[0]].nonzero().flatten().tolist()[:40])
        r0 = int(rows.any(1).nonzero()[0])
        print(f"  row {r0}: got {o.reshape(-1, o.shape[-1])[r0, :16].tolist()}")
        print(f"  row {r0}: ref {ref.reshape(-1, o.shape[-1])[r0, :16].tolist()}")
    h = ref

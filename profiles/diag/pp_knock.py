"""Per-layer times of the bf16 path with whatever library XVEC_LIB points at (knock-out builds:
profiles/diag/build_variants.sh; results of those are garbage, only the times count)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision="bf16"); m.load_state_dict(sd); m = m.to(dev).eval()
x = torch.from_numpy(xa.synth.make_mfcc(256, 300, seed=0)).to(dev)
for _ in range(10): m.extract_x_vec(x)
m.set_profiling(True)
acc = {}
for _ in range(30):
    m.extract_x_vec(x)
    for k, v in m.timings_ms().items(): acc[k] = acc.get(k, 0) + v / 30
m.set_profiling(False)
print(os.path.basename(os.environ.get("XVEC_LIB", "shipping")), {k: round(v, 4) for k, v in acc.items() if k.startswith("tdnn")}, flush=True)

"""Per-layer times of the bf16 path by batch size, one-wave kernel (XVEC_PW=1) against the ping-pong kernel (diagnostic)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
def mk(pw):
    m = xa.XVectorModel(precision="bf16"); m.load_state_dict(sd); m = m.to(dev).eval()
    os.environ["XVEC_PW"] = pw; m._engine(dev); os.environ.pop("XVEC_PW")
    return m
ms = {"pp": mk("0"), "pw": mk("1")}
for B in (256, 384, 448, 512, 672, 896):
    x = torch.from_numpy(xa.synth.make_mfcc(B, 300, seed=0)).to(dev)
    for rnd in range(2):
        for name, m in ms.items():
            for _ in range(10): m.extract_x_vec(x)
            m.set_profiling(True); acc = {}
            for _ in range(20):
                m.extract_x_vec(x)
                for k, v in m.timings_ms().items(): acc[k] = acc.get(k, 0) + v / 20
            m.set_profiling(False)
            print(B, name, m.last_dispatch()[1:4], {k: round(v, 4) for k, v in acc.items() if k in ("tdnn2", "tdnn3", "tdnn4")}, flush=True)

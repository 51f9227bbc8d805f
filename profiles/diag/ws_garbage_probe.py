#!/usr/bin/env python3
"""Do results depend on what the workspace held before the call?  (They must not: xvec_hip.h hands the library an
uninitialised scratch buffer.)  Every arithmetic at a few shapes: the engine's own workspace against fresh ones filled
with zeros, 0xFF bytes (NaNs) and 0x7F bytes (huge values)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import xvector_amd as xa
from xvector_amd import hip

dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
bad = 0
for dt in ("fp32", "bf16", "bf16x3"):
    m = xa.XVectorModel(precision=dt)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    for B, T in ((64, 299), (64, 300), (52, 300), (7, 157), (96, 300), (256, 300)):
        x = torch.from_numpy(xa.synth.make_mfcc(B, T, seed=B + T)).to(dev)
        eng = m._engine(dev)
        need = eng.workspace_bytes(B * T, B)
        res = {}
        for name, fill in (("zeros", 0), ("ff", 0xFF), ("7f", 0x7F), ("zeros2", 0)):
            ws = torch.full((need,), fill, dtype=torch.uint8, device=dev)
            res[name] = m._run(x, hip.MODE_XVEC6, workspace=ws).clone()
            torch.cuda.synchronize()
        ref = res["zeros"]
        line = f"{dt} B={B} T={T} dispatch={m.last_dispatch(dev)}:"
        for name in ("ff", "7f", "zeros2"):
            same = torch.equal(res[name], ref)
            fin = bool(torch.isfinite(res[name]).all())
            d = (res[name] - ref).abs().max().item() if fin else float("nan")
            line += f"  {name}: equal={same} finite={fin} maxdiff={d:.3e}"
            bad += int(not same)
        print(line, flush=True)
print("DEPENDS ON WORKSPACE CONTENTS" if bad else "independent of workspace contents")

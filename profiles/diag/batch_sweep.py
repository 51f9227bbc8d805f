"""Throughput of the path against the batch size (300-frame utterances), per precision.
usage: python profiles/diag/batch_sweep.py [bf16|fp32|bf16x3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision=prec); m.load_state_dict(sd); m = m.to(dev).eval()
for _ in range(200): m.extract_x_vec(torch.zeros(256, 300, 24, device=dev))      # clock ramp
sizes = [int(v) for v in os.environ.get("SWEEP_B", "1,8,16,32,64,96,128,192,256,384,512,1024").split(",")]
for B in sizes:
    x = torch.from_numpy(xa.synth.make_mfcc(B, 300, seed=B)).to(dev)
    n = max(30, min(400, 40000 // B))
    for _ in range(20): m.extract_x_vec(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.extract_x_vec(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{prec} B={B:5d}: {dt * 1e3:8.4f} ms/batch  {B / dt:10.0f} emb/s  {m.last_dispatch()[1:]}", flush=True)

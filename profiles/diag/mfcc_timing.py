"""Throughput of the MFCC front-end kernel: B=256 waveforms of 3 s at 16 kHz (the reference's crop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
fe = xa.MfccFrontEnd()
B, n = 256, 48000
w = torch.randn(B, n, device="cuda:0") * 0.1
for _ in range(5): fe(w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): out = fe(w)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
bytes_alg = B * (n * 4 + 299 * 24 * 4)
print(f"mfcc B={B} n={n}: {ms*1e3:.1f} us per batch -> {B/ms*1e3:.0f} utt/s, {bytes_alg/ms/1e6:.1f} GB/s algorithmic (in once + out)")

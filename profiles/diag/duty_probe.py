#!/usr/bin/env python3
"""Is the bf16 path limited by AVERAGE power?  Interleaved on one box, one process:
  (a) the plain extract_x_vec loop (layers 1-5, pooling merge, segment_layer6),
  (b) model.pooled(x): the same without segment_layer6 (two launches, ~14 us alone),
  (c) (a) with an idle kernel of N us between batches (the chip rests; profiles/diag/src/idle_kernel.hip).
If the path were limited by its schedule alone, (b) saves the launches' own time and (c) costs exactly N us per batch;
if the chip gives idle time back as clock (or takes removed idle time away), the differences are smaller."""
import ctypes as C
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import xvector_amd as xa

so = os.path.join(ROOT, "build", "idle_kernel.so")
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                    os.path.join(ROOT, "profiles", "diag", "src", "idle_kernel.hip")], check=True)
idle = C.CDLL(so)
idle.idle_us.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision=dt)
m.load_state_dict(sd)
m = m.to(dev).eval()
x = torch.randn((256, 300, 24), device=dev)
sp = torch.cuda.current_stream(dev).cuda_stream
STEPS = 300 if dt != "fp32" else 60


def loop(fn, gap_us):
    for _ in range(STEPS):
        fn(x)
        if gap_us:
            idle.idle_us(sp, gap_us)


arms = {"xvec6": (m.extract_x_vec, 0), "pooled (no segment6)": (m.pooled, 0), "xvec6 + 20 us idle": (m.extract_x_vec, 20),
        "xvec6 + 50 us idle": (m.extract_x_vec, 50), "xvec6 + 150 us idle": (m.extract_x_vec, 150)}
t_end = time.perf_counter() + 1.5
while time.perf_counter() < t_end:
    loop(m.extract_x_vec, 0)
    torch.cuda.synchronize()
res = {k: [] for k in arms}
for r in range(5):
    for k, (fn, gap) in arms.items():
        loop(fn, gap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(fn, gap)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / STEPS * 1e6)
base = statistics.median(res["xvec6"])
for k in arms:
    med = statistics.median(res[k])
    print(f"{dt} {k:24s}: {med:8.1f} us per batch ({med - base:+7.1f} vs xvec6; idle asked {arms[k][1]} us)   rounds: " + " ".join(f"{v:.1f}" for v in res[k]), flush=True)

#!/usr/bin/env python3
"""A/B on one box, interleaved rounds in ONE process: the plain loop (model.extract_x_vec on the current stream) against
XVectorModel.pipelined() -- every batch's tail on the library's own stream beside the next batch's layer 1
(model.py, PipelinedPath; include/xvec_hip.h, xvec_set_tail_overlap).
    python3 profiles/diag/pipeline_probe.py [--rounds 5] [--steps 200]
Prints embeddings/s per arm and round, the medians, and whether the pipelined results equal the plain ones bit for bit."""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import xvector_amd as xa

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtypes", default="bf16,fp32")
args = ap.parse_args()
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
B = args.batch
x = torch.randn((B, 300, 24), device=dev, generator=torch.Generator(device=dev).manual_seed(1000))
for dt in args.dtypes.split(","):
    m = xa.XVectorModel(precision=dt)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    steps = args.steps if dt != "fp32" else max(args.steps // 5, 20)
    ref = m.extract_x_vec(x).clone()
    arms = {}
    for n in (1, 2):                      # 1 = plain loop, 2 = pipelined (tail overlapped with the next batch's layer 1)
        if n == 1:
            def run(k, m=m):
                keep = None
                for _ in range(k):
                    keep = m.extract_x_vec(x)
                return keep
        else:
            pipe = m.pipelined()
            outs = pipe.map([x] * 6)
            torch.cuda.synchronize(dev)
            same = all(torch.equal(o, ref) for o in outs)
            print(f"{dt} arm={n}: results equal the plain loop's bit for bit: {same}", flush=True)

            def run(k, pipe=pipe):
                pend = [pipe.submit(x) for _ in range(k)]
                return [p.result() for p in pend][-1]
        arms[n] = run
    t_end = time.perf_counter() + 1.0           # leave the idle power state
    while time.perf_counter() < t_end:
        arms[1](16)
        torch.cuda.synchronize(dev)
    res = {n: [] for n in arms}
    for r in range(args.rounds):
        for n, run in arms.items():
            run(10)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            run(steps)
            torch.cuda.synchronize(dev)
            dtm = time.perf_counter() - t0
            res[n].append(steps * B / dtm)
        print(f"{dt} round {r}: " + "  ".join(f"arm={n}: {res[n][-1] / 1e3:8.1f} k/s" for n in arms), flush=True)
    base = statistics.median(res[1]) if 1 in res else None
    for n in arms:
        med = statistics.median(res[n])
        print(f"{dt} arm={n}: median {med / 1e3:.1f} k embeddings/s" + (f"  ({med / base - 1:+.1%} vs plain)" if base else ""), flush=True)

"""Host-buffer pipeline, per-batch timeline: where do the milliseconds between `resident` and `overlapped` go?
  A  extract.stream_x_vectors as shipped (three streams, event per hop)
  B  one stream, nothing blocking: pinned H2D, extract, pinned D2H ring, harvest `depth` later
  C  side-stream H2D only; D2H on the compute stream
usage: python profiles/diag/stream_probe5.py [fp32|bf16] [batches] [rounds]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import xvector_amd as xa

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision=prec)
m.load_state_dict(sd)
m = m.to(dev).eval()
B = 256
x_dev = torch.randn(B, 300, 24, device=dev)
x_host = x_dev.cpu().pin_memory()
for _ in range(10):
    m.extract_x_vec(x_dev)


def resident():
    for _ in range(N):
        m.extract_x_vec(x_dev)


def naive():
    for _ in range(N):
        m.extract_x_vec(x_host.to(dev, non_blocking=True)).cpu()


def shipped():
    for _ in xa.extract.stream_x_vectors(m, (x_host for _ in range(N))):
        pass


ring = [torch.empty(B, 512, pin_memory=True) for _ in range(4)]
slots = [torch.empty_like(x_dev) for _ in range(4)]


def one_stream(depth=3):
    inflight = []
    for k in range(N):
        i = k & 3
        slots[i].copy_(x_host, non_blocking=True)
        out = m.extract_x_vec(slots[i])
        ring[i].copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        inflight.append((ev, ring[i]))
        if len(inflight) > depth:
            e, b = inflight.pop(0)
            e.synchronize()
            b.numpy().copy()
    for e, b in inflight:
        e.synchronize()
        b.numpy().copy()


h2d = torch.cuda.Stream(dev)


def side_h2d(depth=3):
    compute = torch.cuda.current_stream(dev)
    consumed = [None] * 4
    inflight = []
    for k in range(N):
        i = k & 3
        with torch.cuda.stream(h2d):
            if consumed[i] is not None:
                h2d.wait_event(consumed[i])
            slots[i].copy_(x_host, non_blocking=True)
            arrived = torch.cuda.Event()
            arrived.record(h2d)
        compute.wait_event(arrived)
        out = m.extract_x_vec(slots[i])
        ring[i].copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        consumed[i] = ev
        inflight.append((ev, ring[i]))
        if len(inflight) > depth:
            e, b = inflight.pop(0)
            e.synchronize()
            b.numpy().copy()
    for e, b in inflight:
        e.synchronize()
        b.numpy().copy()
    compute.wait_stream(h2d)


variants = {"resident": resident, "naive": naive, "shipped": shipped, "one_stream": one_stream, "side_h2d": side_h2d}
times = {k: [] for k in variants}
for fn in variants.values():
    fn()
    torch.cuda.synchronize()
for r in range(R):
    for name, fn in variants.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / N * 1e3)
print(f"precision {prec}, {N} batches per call, {R} rounds, ms/batch")
for name, v in times.items():
    print(f"{name:12s} " + " ".join(f"{t:.3f}" for t in v) + f"   median {sorted(v)[len(v) // 2]:.3f}", flush=True)

# host-side cost of one iteration of the shipped pipeline with the GPU idle-ish: time the enqueue calls alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    e = torch.cuda.Event()
    e.record()
t1 = time.perf_counter()
for _ in range(200):
    h2d.wait_event(e)
t2 = time.perf_counter()
for _ in range(200):
    with torch.cuda.stream(h2d):
        pass
t3 = time.perf_counter()
print(f"host cost: event create+record {(t1 - t0) / 200 * 1e6:.1f} us, wait_event {(t2 - t1) / 200 * 1e6:.1f} us, "
      f"stream ctx {(t3 - t2) / 200 * 1e6:.1f} us")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    m.extract_x_vec(x_dev)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue cost of extract_x_vec: {(t1 - t0) / 50 * 1e6:.1f} us")

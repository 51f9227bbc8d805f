"""Where does the host-buffer pipeline lose time?  (overlapped < naive < resident.)
Variants of the per-batch loop, all in one process, interleaved rounds, pinned fp32 input [256,300,24]:
  resident   x on the device, results stay on the device
  naive      to(dev, non_blocking) / extract / .cpu()             (one stream)
  r1         extract.stream_x_vectors as shipped in round 1       (side-stream H2D, blocking .cpu() per batch)
  h2d_only   side-stream H2D, results stay on the device
  d2h_pin    same-stream H2D; D2H enqueued behind the batch into a torch-pinned ring, harvested `depth` later
  d2h_reg    the same with a hipHostRegister'ed ordinary tensor as ring (CPU-cacheable pages)
  both_reg   side-stream H2D + d2h_reg
usage: python profiles/diag/stream_probe3.py [rounds] [batches]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import xvector_amd as xa

dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel()
m.load_state_dict(sd)
m = m.to(dev).eval()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 256
x_host = torch.randn(B, 300, 24).pin_memory()
x_dev = x_host.to(dev)
rt = torch.cuda.cudart()


def registered(shape):
    t = torch.empty(shape, dtype=torch.float32)
    rc = rt.cudaHostRegister(t.data_ptr(), t.numel() * 4, 0)
    assert int(rc) == 0, rc
    return t


def v_resident():
    for _ in range(N):
        m.extract_x_vec(x_dev)


def v_naive():
    acc = 0.0
    for _ in range(N):
        acc += float(m.extract_x_vec(x_host.to(dev, non_blocking=True)).cpu()[0, 0])
    return acc


def v_r1():
    acc = 0.0
    for h in xa.extract.stream_x_vectors_r1(m, (x_host for _ in range(N))):
        acc += float(h[0, 0])
    return acc


def v_h2d_only():
    compute = torch.cuda.current_stream(dev)
    h2d = torch.cuda.Stream(dev)
    slots = [torch.empty_like(x_dev) for _ in range(3)]
    consumed = [None] * 3
    outs = []
    for k in range(N):
        i = k % 3
        with torch.cuda.stream(h2d):
            if consumed[i] is not None:
                h2d.wait_event(consumed[i])
            slots[i].copy_(x_host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(h2d)
        compute.wait_event(ev)
        outs.append(m.extract_x_vec(slots[i]))
        consumed[i] = torch.cuda.Event()
        consumed[i].record(compute)
    compute.wait_stream(h2d)


def _pipeline(ring, side_h2d, depth=2, clone=False, stats=None):
    compute = torch.cuda.current_stream(dev)
    h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    slots = [torch.empty_like(x_dev) for _ in range(depth + 1)]
    consumed = [None] * (depth + 1)
    inflight = []
    acc = 0.0

    def harvest():
        ev, buf = inflight.pop(0)
        t0 = time.perf_counter()
        ev.synchronize()
        t1 = time.perf_counter()
        if clone == "numpy":
            c = torch.from_numpy(buf.numpy().copy())
            t2 = time.perf_counter()
            if stats is not None:
                stats.append((t1 - t0, t2 - t1))
            return float(c[0, 0]) + float(c[B - 1, 511])
        if clone:
            c = buf.clone()
            t2 = time.perf_counter()
            if stats is not None:
                stats.append((t1 - t0, t2 - t1))
            return float(c[0, 0]) + float(c[B - 1, 511])
        return float(buf[0, 0]) + float(buf[B - 1, 511])

    for k in range(N):
        i = k % (depth + 1)
        if side_h2d:
            with torch.cuda.stream(h2d):
                if consumed[i] is not None:
                    h2d.wait_event(consumed[i])
                slots[i].copy_(x_host, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(h2d)
            compute.wait_event(ev)
            xin = slots[i]
        else:
            xin = x_host.to(dev, non_blocking=True)
        out = m.extract_x_vec(xin)
        done = torch.cuda.Event()
        done.record(compute)
        consumed[i] = done
        buf = ring[k % len(ring)]
        with torch.cuda.stream(d2h):
            d2h.wait_event(done)
            buf.copy_(out, non_blocking=True)
            landed = torch.cuda.Event()
            landed.record(d2h)
        out.record_stream(d2h)
        inflight.append((landed, buf))
        if len(inflight) > depth:
            acc += harvest()
    while inflight:
        acc += harvest()
    compute.wait_stream(h2d)
    return acc


ring_pin = [torch.empty(B, 512).pin_memory() for _ in range(4)]
ring_reg = [registered((B, 512)) for _ in range(4)]
variants = {
    "resident": v_resident, "naive": v_naive, "h2d_only": v_h2d_only,
    "d2h_pin": lambda: _pipeline(ring_pin, False), "d2h_reg": lambda: _pipeline(ring_reg, False),
    "both_pin": lambda: _pipeline(ring_pin, True), "both_reg": lambda: _pipeline(ring_reg, True),
    "both_reg_d4": lambda: _pipeline(ring_reg, True, depth=3),
    "both_pin_clone": lambda: _pipeline(ring_pin, True, depth=3, clone=True, stats=clone_stats),
    "both_reg_clone": lambda: _pipeline(ring_reg, True, depth=3, clone=True, stats=clone_stats2),
    "both_pin_npcopy": lambda: _pipeline(ring_pin, True, depth=3, clone="numpy", stats=clone_stats3),
}
clone_stats3 = []
clone_stats, clone_stats2 = [], []
if hasattr(xa.extract, "stream_x_vectors_r1"):
    variants["r1"] = v_r1
variants["shipped"] = lambda: sum(float(h[0, 0]) for h in xa.extract.stream_x_vectors(m, (x_host for _ in range(N))))

times = {k: [] for k in variants}
for name, fn in variants.items():      # warm every variant once
    try:
        fn()
        torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001
        print(f"{name}: FAILED {type(e).__name__}: {e}", flush=True)
        times.pop(name)
for r in range(R):
    for name in list(times):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        variants[name]()
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / N * 1e3)
for name, v in times.items():
    v = sorted(v)
    print(f"{name:12s} ms/batch median {v[len(v) // 2]:.3f}  min {v[0]:.3f}  max {v[-1]:.3f}   -> {B / v[len(v) // 2] * 1e3:9.0f} emb/s",
          flush=True)

print("torch threads", torch.get_num_threads(), "cpu_count", os.cpu_count())
for nm, st in (("pin", clone_stats), ("reg", clone_stats2), ("pin numpy copy", clone_stats3)):
    if st:
        print(f"{nm}: event wait mean {1e3 * sum(a for a, _ in st) / len(st):.3f} ms, clone mean {1e3 * sum(b for _, b in st) / len(st):.3f} ms, clone max {1e3 * max(b for _, b in st):.3f} ms")
# CPU read cost of the two kinds of result buffer
for nm, ring in (("pinned", ring_pin), ("registered", ring_reg)):
    t0 = time.perf_counter()
    for _ in range(20):
        c = ring[0].clone()
    print(f"cpu clone of a {nm} [256,512] fp32 buffer: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")

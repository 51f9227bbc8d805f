"""Where does extract.stream_x_vectors spend host time?  (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(); m.load_state_dict(sd); m = m.to(dev).eval()
x_host = torch.randn(256, 300, 24).pin_memory()
for _ in range(3): m.extract_x_vec(x_host.to(dev))
torch.cuda.synchronize()
compute = torch.cuda.current_stream(dev)
h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
xbuf = torch.empty_like(x_host, device=dev)
res = torch.empty(256, 512).pin_memory()
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
evs = []
t_all = time.perf_counter()
for k in range(20):
    t0 = time.perf_counter()
    with torch.cuda.stream(h2d):
        xbuf.copy_(x_host, non_blocking=True)
        arrived = torch.cuda.Event(); arrived.record(h2d)
    tick("h2d enqueue", t0); t0 = time.perf_counter()
    compute.wait_event(arrived)
    tick("wait_event", t0); t0 = time.perf_counter()
    out = m.extract_x_vec(xbuf)
    tick("extract enqueue", t0); t0 = time.perf_counter()
    done = torch.cuda.Event(); done.record(compute)
    tick("record", t0); t0 = time.perf_counter()
    with torch.cuda.stream(d2h):
        d2h.wait_event(done)
        res.copy_(out, non_blocking=True)
        back = torch.cuda.Event(); back.record(d2h)
    tick("d2h enqueue", t0); t0 = time.perf_counter()
    evs.append((back, out))
    if len(evs) > 2:
        evs.pop(0)[0].synchronize()
    tick("retire", t0)
torch.cuda.synchronize()
print("total ms/batch", (time.perf_counter() - t_all) / 20 * 1e3)
for k, v in T.items():
    print(f"  {k:16s} {v / 20 * 1e3:8.3f} ms")
# same on one stream
t_all = time.perf_counter()
for k in range(20):
    res.copy_(m.extract_x_vec(x_host.to(dev, non_blocking=True)), non_blocking=True)
torch.cuda.synchronize()
print("single-stream ms/batch", (time.perf_counter() - t_all) / 20 * 1e3)
# two streams: H2D on a side stream, compute + D2H on the current one
xb = [torch.empty_like(x_host, device=dev) for _ in range(3)]
rs = [torch.empty(256, 512).pin_memory() for _ in range(4)]
last = [None] * 3
for rep in range(2):
    evs = []
    t_all = time.perf_counter()
    for k in range(40):
        s = k % 3
        with torch.cuda.stream(h2d):
            if last[s] is not None:
                h2d.wait_event(last[s])
            xb[s].copy_(x_host, non_blocking=True)
            arrived = torch.cuda.Event(); arrived.record(h2d)
        compute.wait_event(arrived)
        out = m.extract_x_vec(xb[s])
        rs[k % 4].copy_(out, non_blocking=True)
        done = torch.cuda.Event(); done.record(compute)
        last[s] = done
        evs.append(done)
        if len(evs) > 2:
            evs.pop(0).synchronize()
    torch.cuda.synchronize()
    print("two-stream ms/batch", (time.perf_counter() - t_all) / 40 * 1e3)

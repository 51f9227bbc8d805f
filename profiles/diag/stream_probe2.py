import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(); m.load_state_dict(sd); m = m.to(dev).eval()
x_host = torch.randn(256, 300, 24).pin_memory()
for _ in xa.extract.stream_x_vectors(m, (x_host for _ in range(3))): pass
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in xa.extract.stream_x_vectors(m, (x_host for _ in range(30))): pass
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr.disable()
print("ms/batch", dt / 30 * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
# pageable float64 input (what a DataLoader hands over), results consumed
x64 = torch.randn(256, 299, 24, dtype=torch.float64)
for _ in xa.extract.stream_x_vectors(m, (x64 for _ in range(3))): pass
t0 = time.perf_counter(); acc = 0.0
for h in xa.extract.stream_x_vectors(m, (x64 for _ in range(30))): acc += float(h[0, 0])
torch.cuda.synchronize()
print("pageable f64 input: ms/batch", (time.perf_counter() - t0) / 30 * 1e3)
t0 = time.perf_counter()
for _ in range(30): acc += float(m.extract_x_vec(x64.to(dev).float()).cpu()[0, 0])
print("serial (to(dev) / extract / cpu()): ms/batch", (time.perf_counter() - t0) / 30 * 1e3)

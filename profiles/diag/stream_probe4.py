"""bench.py's overlapped leg in isolation, a few call patterns (why is it slower there than in stream_probe3?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(); m.load_state_dict(sd); m = m.to(dev).eval()
gen = torch.Generator(device=dev).manual_seed(1000)
x = torch.randn((256, 300, 24), generator=gen, device=dev)
for _ in range(10): m.extract_x_vec(x)
x_host = x.cpu().pin_memory()
def t(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def naive(n):
    for k in range(n): out_host = m.extract_x_vec(x_host.to(dev, non_blocking=True)).cpu()
def ovl(n, depth=3):
    for _ in xa.extract.stream_x_vectors(m, (x_host for _ in range(n)), depth=depth): pass
ovl(3)
for n in (30, 50, 200):
    print(f"n={n}: naive {t(lambda: naive(n), n):.3f}  overlapped d3 {t(lambda: ovl(n), n):.3f}  d2 {t(lambda: ovl(n, 2), n):.3f}  d6 {t(lambda: ovl(n, 6), n):.3f} ms/batch", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); ovl(50); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)

"""Timing of the scoring back end (next row N4): PLDA score matrix of N x-vectors against themselves
and N x M, fp64.  Prints ms (median of 20, hipEvents) and TFLOP/s of the N x M x 512 GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch, time
import xvector_amd as xa
from xvector_amd import scoring
import plda_oracle as po          # model generator + CPU timing only

dev = "cuda:0"
dim = 512
mean, F, Sigma = po.make_plda(dim, 200, seed=21)
scorer = scoring.PldaScorer(mean, F, Sigma)                      # low-rank form (rank 200 < 512)
dense = scoring.PldaScorer(mean, F, Sigma, lowrank=False)        # the package's dense Phi / Psi products

def ev_time(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))

for n in (4874, 16384):
    x = torch.randn((n, dim), device=dev, dtype=torch.float64) + torch.from_numpy(mean).to(dev)
    ms = ev_time(lambda: scorer.score(x))
    fl = 2.0 * n * n * dim + 2 * 2.0 * n * dim * dim
    print(f"plda self-score N={n}, low-rank: {ms:.3f} ms  {n * n / ms / 1e6:.1f} G scores/s")
    ms = ev_time(lambda: dense.score(x))
    print(f"plda self-score N={n}, dense: {ms:.3f} ms  {fl / 2 / ms / 1e9:.1f} TFLOP/s fp64 (upper triangle), {n * n / ms / 1e6:.1f} G scores/s")
    y = x[: n // 2].contiguous(); z = x[n // 2:].contiguous()
    ms = ev_time(lambda: scorer.score(y, z))
    print(f"plda two sets {n // 2} x {n - n // 2}, low-rank: {ms:.3f} ms;  dense: {ev_time(lambda: dense.score(y, z)):.3f} ms")
    a = x.contiguous()
    ms = ev_time(lambda: scoring.gemm_nt(a, a))
    print(f"   gemm_nt alone: {ms:.3f} ms  {2.0 * n * n * dim / ms / 1e9:.1f} TFLOP/s")
n = 4874
x = np.random.default_rng(0).normal(0, 1, (n, dim)) + mean
t0 = time.perf_counter(); ref = po.fast_plda_scoring(x, x, mean, F, Sigma); dt = time.perf_counter() - t0
print(f"CPU oracle (numpy float64, {torch.get_num_threads()} threads) N={n}: {dt * 1e3:.1f} ms")
got = scorer.score(x).cpu().numpy()
print("max rel diff vs oracle:", np.abs(got - ref).max() / np.abs(ref).max())

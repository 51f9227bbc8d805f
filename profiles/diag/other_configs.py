"""Timing of the BASELINE configs that are not the bench line: ragged batch (config 2), batch-of-one
latency, bf16 (config 4).  Prints ms per call (median of 30, after warm-up)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = "cuda:0"
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
def model(prec):
    m = xa.XVectorModel(precision=prec); m.load_state_dict(sd); return m.to(dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
for prec in ("fp32", "bf16"):
    m = model(prec)
    lens = xa.synth.make_lengths(256)
    x = torch.from_numpy(xa.synth.make_mfcc(256, int(lens.max()), seed=2)).to(dev)
    ll = lens.tolist()
    ms = timeit(lambda: m.extract_x_vec(x, lengths=ll))
    print(f"{prec} config2 ragged B=256 frames={int(lens.sum())}: {ms:.3f} ms  -> {256/ms*1e3:.0f} utt/s, {lens.sum()/ms*1e3/1e6:.2f} M valid frames/s")
    x300 = x[:, :300].contiguous()
    ms = timeit(lambda: m.extract_x_vec(x300))
    print(f"{prec} config1 B=256 T=300: {ms:.3f} ms -> {256/ms*1e3:.0f} emb/s")
    for B in (1, 8, 64):
        xb = x300[:B].contiguous()
        ms = timeit(lambda: m.extract_x_vec(xb))
        print(f"{prec} B={B} T=300 latency: {ms*1e3:.0f} us")

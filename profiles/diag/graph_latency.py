"""Small-batch latency: eager launches vs one hipGraph replay of the whole path (torch.cuda.graph
captures the library's launches: nothing in xvec_forward synchronises or allocates)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
for prec in ("fp32", "bf16"):
    m = xa.XVectorModel(precision=prec); m.load_state_dict(sd); m = m.to(dev).eval()
    for B in (1, 8, 64):
        x = torch.randn(B, 300, 24, device=dev)
        for _ in range(3): ref = m.extract_x_vec(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m.extract_x_vec(x)
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(out, ref), "graph replay differs from the eager result"
        def lat(fn, n=200):
            ts = []
            for _ in range(n):
                t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return 1e6 * float(np.median(ts))
        e = lat(lambda: m.extract_x_vec(x)); r = lat(g.replay)
        print(f"{prec} B={B}: eager {e:.0f} us   graph replay {r:.0f} us")

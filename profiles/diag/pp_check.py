"""New bf16 mapping (tdnn_pp.hip) against the 128x128 bf16 kernel and the fp32 path, layer by layer and end to end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
def mk(prec, pp):
    os.environ["XVEC_PP"] = pp
    m = xa.XVectorModel(precision=prec); m.load_state_dict(sd); m = m.to(dev).eval()
    m.extract_x_vec(torch.zeros(1, 32, 24, device=dev))   # create the engine under this env
    return m
m32, mold, mnew = mk("fp32", "0"), mk("bf16", "0"), mk("bf16", "1")
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 300
x = torch.from_numpy(xa.synth.make_mfcc(B, T, seed=0)).to(dev)
def rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).reshape(-1, a.shape[-1]).norm(dim=1) / b.reshape(-1, b.shape[-1]).norm(dim=1).clamp_min(1e-30)).max())
h = x
for i in range(5):
    ref = m32.time_context_layers[i](h)
    o = mold.time_context_layers[i](h); n = mnew.time_context_layers[i](h)
    torch.cuda.synchronize()
    print(f"layer {i}: old-vs-fp32 {rel(o, ref):.3e}  new-vs-fp32 {rel(n, ref):.3e}  new-vs-old {rel(n, o):.3e}  finite {bool(torch.isfinite(n).all())}", flush=True)
    h = ref
e32, eo, en = m32.extract_x_vec(x), mold.extract_x_vec(x), mnew.extract_x_vec(x)
torch.cuda.synchronize()
print(f"xvec: old-vs-fp32 {rel(eo, e32):.3e} new-vs-fp32 {rel(en, e32):.3e} new-vs-old {rel(en, eo):.3e} deterministic {bool(torch.equal(en, mnew.extract_x_vec(x)))}", flush=True)
lens = xa.synth.make_lengths(B)
xr = torch.from_numpy(xa.synth.make_mfcc(B, int(lens.max()), seed=2)).to(dev)
r32, rn = m32.extract_x_vec(xr, lengths=lens.tolist()), mnew.extract_x_vec(xr, lengths=lens.tolist())
torch.cuda.synchronize()
print(f"ragged xvec: new-vs-fp32 {rel(rn, r32):.3e}", flush=True)
for name, m in (("old", mold), ("new", mnew)):
    for _ in range(10): m.extract_x_vec(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): m.extract_x_vec(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    m.set_profiling(True)
    acc = {}
    for _ in range(20):
        m.extract_x_vec(x)
        for k, v in m.timings_ms().items(): acc[k] = acc.get(k, 0) + v / 20
    m.set_profiling(False)
    print(name, f"{dt*1e3:.4f} ms/batch {B/dt:.0f} emb/s", {k: round(v, 4) for k, v in acc.items() if v > 0}, flush=True)

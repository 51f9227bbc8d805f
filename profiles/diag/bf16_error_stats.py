"""Error statistics of the bf16 path against the fp32 path (same inputs), per layer and end to end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m32 = xa.XVectorModel(); m32.load_state_dict(sd); m32 = m32.to("cuda:0")
m16 = xa.XVectorModel(precision="bf16"); m16.load_state_dict(sd); m16 = m16.to("cuda:0")
def stats(name, got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    g2, r2 = got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])
    rel = ((g2 - r2).norm(dim=1) / r2.norm(dim=1)).max().item()
    atol = ref.abs().mean().item()
    ratio = ((got - ref).abs() / (atol + ref.abs())).max().item()
    print(f"{name:14s} max row rel {rel:.3e}   max |d|/(mean|ref|+|ref|) {ratio:.3e}  finite {bool(torch.isfinite(got).all())}")
h = torch.from_numpy(xa.synth.make_mfcc(3, 150, seed=21)).to("cuda:0")
for i in range(5):
    ref = m32.time_context_layers[i](h); got = m16.time_context_layers[i](h)
    stats(f"layer {i}", got, ref); h = ref
for B in (8, 256):
    x = torch.from_numpy(xa.synth.make_mfcc(B, 300, seed=31)).to("cuda:0")
    stats(f"xvec B={B}", m16.extract_x_vec(x), m32.extract_x_vec(x))
    stats(f"logits B={B}", m16(x), m32(x))

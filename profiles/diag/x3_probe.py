import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, torch
import xvector_amd as xa
import xvector_oracle as oracle
dev = "cuda:0"
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
p64 = oracle.cast_params({k: v for k, v in sd.items() if v.is_floating_point()}, torch.float64)
x = torch.from_numpy(xa.synth.make_mfcc(8, 300, seed=3))
ref = oracle.extract_x_vec(x.double(), p64)
for prec in ("fp32", "bf16x3", "bf16"):
    m = xa.XVectorModel(precision=prec); m.load_state_dict(sd); m = m.to(dev).eval()
    got = m.extract_x_vec(x.to(dev)).double().cpu()
    rel = ((got - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
    el = ((got - ref).abs() / (ref.abs().mean())).max().item()
    print(prec, "row rel err", rel, "max abs err / mean|ref|", el)
    # per-layer
    h = x.to(dev)
    hr = x.double()
    for i, layer in enumerate(m.time_context_layers):
        h = layer(h)
        hr = oracle.time_context_layers(x.double(), p64, upto=i + 1)
        print("   layer", i + 1, "rel", ((h.double().cpu() - hr).norm() / hr.norm()).item())

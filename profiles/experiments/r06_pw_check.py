"""pw::tdnn_pw_kernel against the 128x128 kernel, element by element in bf16 steps (diagnostic)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np, torch
import xvector_amd as xa
DEV = "cuda:0"
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
def mk(env):
    m = xa.XVectorModel(precision="bf16"); m.load_state_dict(sd); m = m.to(DEV)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env); m._engine(torch.device(DEV))
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    return m
m_pw, m_pp, m_old = mk({"XVEC_PW": "1"}), mk({"XVEC_PW": "0"}), mk({"XVEC_PP": "0"})
m32 = xa.XVectorModel(); m32.load_state_dict(sd); m32 = m32.to(DEV).eval()
def pre_bn(y, i):
    k = f"time_context_layers.{i}.norm."
    sc = (sd[k + "weight"].double() / torch.sqrt(sd[k + "running_var"].double() + 1e-5)).to(y.device)
    sh = (sd[k + "bias"].double().to(y.device) - sd[k + "running_mean"].double().to(y.device) * sc)
    return (y.double() - sh) / sc
for B, T in [(63, 300), (256, 300)]:
    h = torch.as_tensor(xa.synth.make_mfcc(B, T, seed=1000 + B)).to(DEV)
    for i in range(4):
        if i > 0:
            outs = {n: pre_bn(m.time_context_layers[i](h), i) for n, m in (("pw", m_pw), ("pp", m_pp), ("t128", m_old))}
            print(B, T, "layer", i, m_pw.last_dispatch()[i], m_pp.last_dispatch()[i], m_old.last_dispatch()[i])
            for a, b in (("pw", "t128"), ("pp", "t128"), ("pw", "pp")):
                d = (outs[a] - outs[b]).abs()
                ulp = d / (outs[b].abs().clamp_min(1e-6) * 2.0 ** -8)
                rows = d.reshape(-1, d.shape[-1]).norm(dim=1) / outs[b].reshape(-1, d.shape[-1]).norm(dim=1).clamp_min(1e-30)
                print(f"   {a} vs {b}: differing elements {int((d > 0).sum())} of {d.numel()}, max {float(ulp.max()):.2f} bf16 steps, "
                      f"elements > 1.01 steps {int((ulp > 1.01).sum())}, worst row {float(rows.max()):.2e}")
        h = m32.time_context_layers[i](h)

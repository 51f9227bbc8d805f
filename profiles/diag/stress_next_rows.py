"""Random-shape stress of the two kernels that changed in round 6 (not a test: a one-off run; failures print and exit 1)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np, torch
import xvector_amd as xa
from xvector_amd import scoring
import mfcc_oracle as mo, plda_oracle as po
rng = np.random.default_rng(12345)
bad = 0
# ---- MFCC: random batch / length, float and int16, both filterbank forms, sampled rows against the oracle
fes = {"auto": xa.MfccFrontEnd()}
os.environ["XVEC_MFCC_FILTERBANK"] = "dense"; fes["dense"] = xa.MfccFrontEnd(); os.environ.pop("XVEC_MFCC_FILTERBANK")
for it in range(60):
    B = int(rng.integers(1, 400)); n = int(rng.integers(1, 60000))
    w = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
    if it % 5 == 0: w[rng.integers(0, B)] = 0
    if it % 7 == 0: w[:, : n // 3] = 0
    wt = torch.from_numpy(w).to("cuda:0")
    outs = {k: fe(wt) for k, fe in fes.items()}
    a, d = outs["auto"], outs["dense"]
    if not torch.isfinite(a).all() or not torch.isfinite(d).all(): print("non-finite", B, n); bad += 1
    rel = ((a - d).double().norm(dim=-1) / d.double().norm(dim=-1).clamp_min(1e-30)).max().item()
    if rel > 5e-6: print("banded vs dense", B, n, rel); bad += 1
    for b in rng.integers(0, B, size=min(B, 2)):
        ref = mo.mfcc(w[b].astype(np.float64), 16000, numcep=24, nfilt=26, nfft=512)
        got = a[b].cpu().numpy()
        r = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
        if got.shape != ref.shape or r > 2e-4: print("mfcc vs oracle", B, n, b, got.shape, ref.shape, r); bad += 1
    pcm = torch.from_numpy(np.clip(np.round(w * 20000), -32768, 32767).astype(np.int16)).to("cuda:0")
    for k, fe in fes.items():
        if not torch.equal(fe(pcm, scale=1 / 20000), fe(pcm.float() * (1 / 20000))): print("i16 vs float", k, B, n); bad += 1
print("mfcc done, bad =", bad, flush=True)
# ---- scoring: random N (both tilings by the rule), symmetric exactly, sampled blocks against the oracle
for it in range(24):
    dim = int(rng.choice([64, 200, 512])); rank = int(rng.integers(1, dim + 1)) if it % 3 else dim
    n = int(rng.integers(1, 7000))
    mean, F, Sigma = po.make_plda(dim, rank, seed=100 + it)
    x = rng.standard_normal((n, dim)) + mean
    xt = torch.from_numpy(x).to("cuda:0")
    for lowrank in (True, False):
        sc = scoring.PldaScorer(mean, F, Sigma, lowrank=lowrank)
        s = sc.score(xt)
        if not torch.equal(s, s.T): print("asymmetric", n, dim, rank, lowrank); bad += 1
        i0 = int(rng.integers(0, max(1, n - 60))); j0 = int(rng.integers(0, max(1, n - 70)))
        ref = po.fast_plda_scoring(x[i0:i0 + 60], x[j0:j0 + 70], mean, F, Sigma)
        got = s[i0:i0 + 60, j0:j0 + 70].cpu().numpy()
        r = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)
        if r > 1e-9: print("plda vs oracle", n, dim, rank, lowrank, r); bad += 1
        m = int(rng.integers(1, 3000))
        y = rng.standard_normal((m, dim)) + mean
        two = sc.score(xt, torch.from_numpy(y).to("cuda:0"))
        ref2 = po.fast_plda_scoring(x[:50], y[-40:], mean, F, Sigma)
        r2 = np.abs(two[:50, -40:].cpu().numpy() - ref2).max() / max(np.abs(ref2).max(), 1e-300)
        if r2 > 1e-9: print("plda two sets vs oracle", n, m, dim, rank, lowrank, r2); bad += 1
    print("score", it, n, dim, rank, "bad =", bad, flush=True)
print("TOTAL bad", bad)
sys.exit(1 if bad else 0)

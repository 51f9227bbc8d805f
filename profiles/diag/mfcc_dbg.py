import os, sys, subprocess, json
root = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
child = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch
import xvector_amd as xa
fe = xa.MfccFrontEnd(device="cuda:0", preemph=0.0)
n = np.arange(400)
out = {}
for k in list(range(0, 257, 1)):
    w = np.zeros((1, 400), np.float32); w[0] = np.cos(2 * np.pi * k * n / 512.0)
    out[k] = fe(torch.from_numpy(w).to("cuda:0")).cpu().numpy()[0, 0].tolist()
print(json.dumps(out))
''' % root
res = {}
for lib in sys.argv[1:3]:
    env = dict(os.environ); env["XVEC_LIB"] = os.path.abspath(lib)
    o = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
    line = [x for x in o.stdout.splitlines() if x.startswith("{")]
    if not line: print(lib, "FAILED", o.stderr[-500:]); sys.exit(1)
    res[lib] = json.loads(line[-1])
a, b = [res[l] for l in sys.argv[1:3]]
import numpy as np
bad = []
for k in a:
    d = np.abs(np.array(a[k]) - np.array(b[k])).max()
    if d > 1e-3: bad.append((int(k), round(float(d), 3)))
print("bins whose single-tone frame differs:", len(bad))
for k in ("10", "100"):
    print(k, "old", np.round(a[k][:8], 3).tolist()); print(k, "new", np.round(b[k][:8], 3).tolist())

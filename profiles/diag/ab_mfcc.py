import json, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
child = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
import xvector_amd as xa
fe = xa.MfccFrontEnd(device="cuda:0")
w = 0.1 * torch.randn(256, 48000, device="cuda:0")
for _ in range(30): fe(w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): fe(w)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"mfcc_us": e0.elapsed_time(e1) / 200 * 1e3}))
''' % root
libs = sys.argv[1:]
res = {l: [] for l in libs}
for r in range(6):
    for l in libs:
        env = dict(os.environ); env["XVEC_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if line: res[l].append(json.loads(line[-1])["mfcc_us"])
        else: print(l, "FAILED", out.stderr[-300:])
for l in libs:
    v = sorted(res[l]); print(os.path.basename(l), "median %.2f  min %.2f  all" % (v[len(v)//2], v[0]), [round(x, 1) for x in res[l]])

#!/usr/bin/env python3
"""Timeline of consecutive batches, plain loop or pipelined (tail overlapped), for `rocprofv3 --kernel-trace`:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o tl -- python3 profiles/diag/pipeline_timeline.py run [plain|pipe] [dtype]
    python3 profiles/diag/pipeline_timeline.py show <dir>       # start / end of every launch of the last batches, relative"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import numpy as np
    import torch
    import xvector_amd as xa
    mode = sys.argv[2] if len(sys.argv) > 2 else "pipe"
    dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
    m = xa.XVectorModel(precision=dt)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = torch.randn((256, 300, 24), device=dev)
    for _ in range(300):
        m.extract_x_vec(x)
    torch.cuda.synchronize()
    if mode == "plain":
        for _ in range(40):
            keep = m.extract_x_vec(x)
    else:
        pipe = m.pipelined()
        pend = [pipe.submit(x) for _ in range(40)]
        keep = [p.result() for p in pend]
    torch.cuda.synchronize()
else:
    path = glob.glob(sys.argv[2].rstrip("/") + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if "xvec::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-7 * 6 - 3:-7 * 3]           # three batches from the middle of the last run
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("xvec::", "")[:60]
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  "
              f"({(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f})  q={r.get('Queue_Id', '?')}  {name}")

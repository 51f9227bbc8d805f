#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel name.
usage: python profiles/summarize_pmc.py gpurun_out/pmc_<tag> > profiles/<tag>_pmc_summary.txt"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if "xvec::" not in k and "mfcc" not in k:       # (the MFCC kernels live in an anonymous namespace)
                continue
            k = k.replace("void xvec::", "").replace("void ", "").replace("(xvec::TdnnArgs)", "").replace("(anonymous namespace)::", "").split("(")[0]
            # the fp64 score GEMM runs at two very different sizes in one PLDA score (two [n,512] x [512,512] products, then
            # the [n,n] score matrix): one entry per grid, not an average over both
            if "gemm_nt_f64_kernel" in k:
                k += f" [grid {int(row['Grid_Size']) // max(int(row['Workgroup_Size']), 1)}]"
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
# HBM traffic per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
# FETCH_SIZE (KiB) under-reports wide coalesced reads by exactly 2x, WRITE_SIZE (KiB) is exact.
what = sys.argv[3] if len(sys.argv) > 3 else "bench.py --steps 3 --warmup 1"
traffic = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes in {root} ({what}); "
                     "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, mean per launch"}
for k in agg:
    if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
        f = sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"])
        w = sum(agg[k]["WRITE_SIZE"]) / len(agg[k]["WRITE_SIZE"])
        traffic[k] = {"fetch_kib_raw": f, "write_kib": w, "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
if len(sys.argv) > 2:
    import json
    json.dump(traffic, open(sys.argv[2], "w"), indent=1)
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"    {c:32s} mean {sum(v) / len(v):16.1f}   n={len(v)}")

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
            if "xvec::" not in k:
                continue
            k = k.replace("void xvec::", "").replace("(xvec::TdnnArgs)", "").split("(")[0]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"    {c:32s} mean {sum(v) / len(v):16.1f}   n={len(v)}")

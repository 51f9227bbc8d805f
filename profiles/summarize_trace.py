"""Per-kernel mean duration over the TIMED steps of a `rocprofv3 --kernel-trace` run of bench.py (the last `steps` batches of
the trace; the `--stats` CSV beside it averages every launch of the process, preroll and warm-up included).
python3 profiles/summarize_trace.py <dir with *_kernel_trace.csv> [steps=20]"""
import csv, glob, statistics, sys

path = glob.glob(sys.argv[1].rstrip("/") + "/*kernel_trace.csv")[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = [r for r in csv.DictReader(open(path)) if "xvec::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one batch = the launches from one layer-1 kernel to the next; keep the last `steps` whole batches
def is_first(r):
    n = r["Kernel_Name"]
    return "tdnn_first" in n or "tdnn_kernel<1," in n or "tdnn_kernel<2," in n
starts = [i for i, r in enumerate(rows) if is_first(r)]
per_batch = len(rows) - starts[-1]
lo = starts[-steps]
sel = rows[lo:starts[-1] + per_batch]
acc = {}
for r in sel:
    acc.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"# {path}: last {steps} batches ({len(sel)} launches of {len(rows)}); us per launch")
print(f"{'kernel':90s} {'calls':>6s} {'mean':>9s} {'min':>9s} {'max':>9s}")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:90]:90s} {len(v):6d} {statistics.mean(v):9.1f} {min(v):9.1f} {max(v):9.1f}")
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3 / steps
print(f"# wall per batch over the region UNDER THE PROFILER (tracing adds gaps between launches): {span:.1f} us; sum of the means above: {sum(statistics.mean(v) * len(v) for v in acc.values()) / steps:.1f} us")

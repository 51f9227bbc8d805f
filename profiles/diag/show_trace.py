"""Timeline of the last N kernel launches of a `rocprofv3 --kernel-trace --output-format csv` run:
python3 profiles/diag/show_trace.py <dir> [N=20]   (start -> end in us relative to the first one shown, duration, blocks, name)"""
import csv, glob, sys
path = glob.glob(sys.argv[1].rstrip("/") + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 20):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[-70:]
    blocks = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))), 1)
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} -> {e / 1e3:9.1f} us ({(e - s) / 1e3:7.1f})  blocks {blocks:5d}  {name}")

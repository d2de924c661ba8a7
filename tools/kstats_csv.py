"""Per-kernel statistics of a rocprofv3 --kernel-trace csv, split by grid size (the levels of a Gauss-Newton chain).
    python tools/kstats_csv.py <dir>/p_kernel_trace.csv [frames]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
acc = defaultdict(list)
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void mmf::", "").replace("mmf::", "")[:40]
    acc[(n, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r.get("Grid_Size_Y", 1) or 1))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values())
print(f"kernel time per frame: {tot / frames / 1e3:.1f} us over {frames} frames")
for (n, gx, gy), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:40s} wgs {gx:5d} x {gy:2d} n/frame={len(v) / frames:6.2f} avg={sum(v) / len(v) / 1e3:7.2f}us min={min(v) / 1e3:6.2f} us/frame={sum(v) / frames / 1e3:7.1f}")

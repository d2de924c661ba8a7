"""Host API calls and GPU kernels of one frame on one time axis, from rocprofv3 --hip-runtime-trace --kernel-trace (csv).
    python tools/api_timeline.py <dir> [frame] [from_us] [to_us]"""
import csv
import sys

d = sys.argv[1]
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 80
lo = float(sys.argv[3]) if len(sys.argv) > 3 else -1e9
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
k = list(csv.DictReader(open(d + "/p_kernel_trace.csv")))
k.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(k) if "odom_begin_kernel" in r["Kernel_Name"]]
t0, t1 = int(k[starts[frame]]["Start_Timestamp"]), int(k[starts[frame + 1]]["Start_Timestamp"])
ev = []
for r in k:
    s = int(r["Start_Timestamp"])
    if t0 - 100000 <= s <= t1:
        name = r["Kernel_Name"].split("(")[0].replace("void mmf::", "").replace("mmf::", "")[:28]
        ev.append((s, "GPU q%s %-28s dur %.1f" % (r["Queue_Id"], name, (int(r["End_Timestamp"]) - s) / 1e3)))
for r in csv.DictReader(open(d + "/p_hip_api_trace.csv")):
    s = int(r["Start_Timestamp"])
    if t0 - 100000 <= s <= t1 and r["Function"] not in ("hipGetLastError", "hipSetDevice"):
        ev.append((s, "   host %-30s %.1f us" % (r["Function"], (int(r["End_Timestamp"]) - s) / 1e3)))
ev.sort()
for s, txt in ev:
    t = (s - t0) / 1e3
    if lo <= t <= hi:
        print("%8.1f %s" % (t, txt))

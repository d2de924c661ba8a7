"""The model's stream of the headline loop from a rocprofv3 kernel trace (csv): per kernel name, launches per frame and
the time from the end of the previous launch on that queue (or its own start, whichever is later) to its own end, i.e.
what the launch adds to a gap-free stream; averaged over the steady-state frames.
   python tools/q1_sum.py <dir>/p_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
begins = [i for i, r in enumerate(rows) if "odom_begin_kernel" in r["Kernel_Name"]]
q1 = rows[begins[0]]["Queue_Id"]
first, last = begins[len(begins) // 4], begins[-2]
nframes = sum(1 for b in begins if first <= b < last)
dur = defaultdict(float)
cnt = defaultdict(int)
for r in rows[first:last]:
    if r["Queue_Id"] != q1:
        continue
    n = r["Kernel_Name"].split("(")[0].replace("void mmf::", "").replace("mmf::", "")
    if "prep_batch" in n:
        n += " grid %s" % r["Grid_Size_X"]
    dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[n] += 1
total = 0.0
for n in sorted(dur, key=lambda k: -dur[k]):
    print("%-52s n/frame %5.2f  avg %6.2f us  per frame %6.1f us" % (n[:52], cnt[n] / nframes, dur[n] / cnt[n], dur[n] / nframes))
    total += dur[n] / nframes
print("kernel time on the model's stream per frame: %.1f us over %d frames" % (total, nframes))

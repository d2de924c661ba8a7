"""Timeline of one steady-state frame from a rocprofv3 kernel trace (csv): start, end, duration, gap to the previous
kernel on the same queue.   python tools/timeline.py <dir>/p_kernel_trace.csv [frame-index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 100
# a frame starts where its tracking begins: odom_begin_kernel, or -- when that rode the last launch of the preparation enqueued
# ahead of the frame -- that launch (prep_batch_begin_kernel)
starts = [i for i, r in enumerate(rows) if "odom_begin_kernel" in r["Kernel_Name"] or "prep_batch_begin_kernel" in r["Kernel_Name"]]
i0, i1 = starts[which], starts[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
j = i0
while j > 0 and int(rows[j - 1]["Start_Timestamp"]) > t0 - 150000:
    j -= 1
prev_end = {}
for r in rows[j:i1 + 2]:
    n = r["Kernel_Name"].split("(")[0].replace("void mmf::", "").replace("mmf::", "")[:34]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r["Queue_Id"]
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    print("%8.1f %7.1f dur %6.1f gap %6.1f q%s %-34s grid %s" % (s, e, e - s, gap, q, n, r["Grid_Size_X"]))

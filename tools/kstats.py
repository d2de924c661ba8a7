"""Summarise a rocprofv3 --kernel-trace results database (rocpd sqlite): per-kernel and per-grid-size stats.

    python tools/kstats.py gpurun_out/prof/x_results.db [frames] [name-pattern ...]
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = db.execute("select name, count(*), avg(end-start), sum(end-start), min(end-start) from kernels "
                  "group by name order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
print(f"kernel time per frame: {tot / frames / 1e3:.1f} us over {frames} frames")
for r in rows[:int(30)]:
    n = re.sub(r"\(.*", "", r[0]).replace("void mmf::", "").replace("mmf::", "")[:44]
    print(f"{n:44s} n/frame={r[1] / frames:5.1f} avg={r[2] / 1e3:7.2f}us min={r[4] / 1e3:6.2f} "
          f"us/frame={r[3] / frames / 1e3:6.1f} ({100 * r[3] / tot:4.1f}%)")
for pat in sys.argv[3:]:
    g = db.execute("select grid_x, count(*), avg(end-start), min(end-start) from kernels where name like ? "
                   "group by grid_x order by grid_x", (f"%{pat}%",)).fetchall()
    print(pat, [(r[0], r[1], round(r[2] / 1e3, 2), round(r[3] / 1e3, 2)) for r in g])

"""median / min duration of the kernels whose name holds a pattern, from a rocprofv3 --kernel-trace output directory
    python tools/kmedian.py <dir> pattern..."""
import glob
import sqlite3
import statistics
import sys

dbs = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
db = sqlite3.connect(dbs[0])
for pat in sys.argv[2:]:
    d = [r[0] / 1e3 for r in db.execute("select end-start from kernels where name like ?", (f"%{pat}%",))]
    print(f"{pat}: n={len(d)} median={statistics.median(d):.2f} min={min(d):.2f} us", end="  ")
print()

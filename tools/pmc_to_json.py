"""profiles/r02_pmc_summary.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) of
tools/profile_frames.py: HBM-side traffic per launch of the Gauss-Newton kernels, per pyramid level.

    python tools/pmc_to_json.py <fetch dir> <write dir> <width> <height> [out.json]

Corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
tallies 128-byte read requests at 64 bytes, so it is doubled (calibrated there for 16-B-per-lane streams; the producer
reads 8 B per lane and gathers 8 / 16 B records -- round 1 found the same factor for the 16-B build of the ICP kernel);
WRITE_SIZE is exact.  Counters are summed over the 8 XCDs by rocprofv3."""
import glob
import json
import os
import sys

import pandas as pd


def load(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    df = pd.concat(pd.read_csv(f) for f in files)
    df["Kernel_Name"] = df["Kernel_Name"].str.replace(r"^void mmf::", "", regex=True).str.replace(r"\(.*$", "", regex=True)
    return df


fetch, write = load(sys.argv[1]), load(sys.argv[2])
W, H = int(sys.argv[3]), int(sys.argv[4])
out = sys.argv[5] if len(sys.argv) > 5 else "profiles/r02_pmc_summary.json"
recs = []
for kern in ("track_producer_kernel", "rgb_step_kernel"):
    f = fetch[fetch["Kernel_Name"].str.contains(kern) & (fetch["Counter_Name"] == "FETCH_SIZE")]
    w = write[write["Kernel_Name"].str.contains(kern) & (write["Counter_Name"] == "WRITE_SIZE")]
    grids = sorted(f["Grid_Size"].unique(), reverse=True)[:3]
    for level, grid in enumerate(grids):
        fk = f[f["Grid_Size"] == grid]["Counter_Value"]
        wk = w[w["Grid_Size"] == grid]["Counter_Value"]
        if not len(fk) or not len(wk):
            continue
        recs.append({"kernel": kern, "width": W, "height": H, "level": level, "grid_threads": int(grid), "launches": int(len(fk)),
                     "FETCH_SIZE_KiB_mean": float(fk.mean()), "WRITE_SIZE_KiB_mean": float(wk.mean()),
                     "traffic_bytes_per_launch": float((2.0 * fk.mean() + wk.mean()) * 1024.0),
                     "correction": "2 x FETCH_SIZE (gfx950: 128-byte requests tallied at 64 bytes) + WRITE_SIZE, KiB -> bytes"})
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/profile_frames.py 30 "
                     f"{W}x{H} 1 0", "kernels": recs}, open(out, "w"), indent=1)
for r in recs:
    print(r["kernel"], "level", r["level"], "grid", r["grid_threads"], f"fetch {r['FETCH_SIZE_KiB_mean']:.1f} KiB write {r['WRITE_SIZE_KiB_mean']:.1f} KiB "
          f"-> {r['traffic_bytes_per_launch'] / 1e6:.2f} MB per launch")

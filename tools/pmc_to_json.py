"""profiles/r04_pmc_summary.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) of
tools/profile_frames.py: HBM-side traffic per launch of the Gauss-Newton kernel (per pyramid level = per grid size) and of the
surfel / preparation kernels of a frame.

    python tools/pmc_to_json.py <fetch dir> <write dir> <width> <height> [out.json] [source stamp]

Corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
tallies 128-byte read requests at 64 bytes, so it is doubled (calibrated there for 16-B-per-lane streams; round 1 found the
same factor for the 8-B and 16-B builds of the ICP kernel; narrower gathers are uncalibrated: ratios between kernels of one
access shape hold, absolutes are indicative); WRITE_SIZE is exact.  Counters are summed over the 8 XCDs by rocprofv3."""
import glob
import json
import os
import sys

import pandas as pd

PER_LEVEL = ("gn_iter_kernel", "track_producer_kernel", "rgb_step_kernel")
PER_KERNEL = ("index_map_kernel", "index_resolve_kernel", "splat_kernel", "splat_resolve_fill_kernel", "fuse_data_kernel",
              "fuse_update_kernel", "fuse_update_index_kernel", "clean_flag_kernel", "clean_scatter_kernel", "prep_batch_kernel", "bilateral_filter2_kernel",
              "so3_kernel", "gn_final_kernel", "odom_publish_kernel")


def load(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    df = pd.concat(pd.read_csv(f) for f in files)
    df["Kernel_Name"] = df["Kernel_Name"].str.replace(r"^(void )?mmf::", "", regex=True).str.replace(r"\(.*$", "", regex=True)
    return df


fetch, write = load(sys.argv[1]), load(sys.argv[2])
W, H = int(sys.argv[3]), int(sys.argv[4])
out = sys.argv[5] if len(sys.argv) > 5 else "profiles/r04_pmc_summary.json"
STAMP = sys.argv[6] if len(sys.argv) > 6 else None  # bench.source_stamp() of the kernels that were measured
CORR = "2 x FETCH_SIZE (gfx950: 128-byte requests tallied at 64 bytes) + WRITE_SIZE, KiB -> bytes"
recs = []


def rec(kern, f, w, **extra):
    if not len(f) or not len(w):
        return
    r = {"kernel": kern, "width": W, "height": H, "launches": int(len(f)), "FETCH_SIZE_KiB_mean": float(f.mean()),
         "WRITE_SIZE_KiB_mean": float(w.mean()), "traffic_bytes_per_launch": float((2.0 * f.mean() + w.mean()) * 1024.0), "correction": CORR}
    r.update(extra)
    recs.append(r)


for kern in PER_LEVEL:
    f = fetch[fetch["Kernel_Name"].str.contains(kern) & (fetch["Counter_Name"] == "FETCH_SIZE")]
    w = write[write["Kernel_Name"].str.contains(kern) & (write["Counter_Name"] == "WRITE_SIZE")]
    for level, grid in enumerate(sorted(f["Grid_Size"].unique(), reverse=True)[:3]):
        rec(kern, f[f["Grid_Size"] == grid]["Counter_Value"], w[w["Grid_Size"] == grid]["Counter_Value"], level=level, grid_threads=int(grid))
for kern in PER_KERNEL:
    f = fetch[fetch["Kernel_Name"].str.startswith(kern) & (fetch["Counter_Name"] == "FETCH_SIZE")]["Counter_Value"]
    w = write[write["Kernel_Name"].str.startswith(kern) & (write["Counter_Name"] == "WRITE_SIZE")]["Counter_Value"]
    rec(kern, f, w, level=None)
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/profile_frames.py 60 "
                     f"{W}x{H} 1 0 headline", "source_stamp": STAMP, "kernels": recs}, open(out, "w"), indent=1)
for r in recs:
    print(f"{r['kernel']:28s} level {r['level']} launches {r['launches']:4d} fetch {r['FETCH_SIZE_KiB_mean']:9.1f} KiB write "
          f"{r['WRITE_SIZE_KiB_mean']:9.1f} KiB -> {r['traffic_bytes_per_launch'] / 1e6:7.2f} MB per launch")

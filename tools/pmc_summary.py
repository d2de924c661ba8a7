"""Mean counter value (and duration) per kernel from a rocprofv3 --pmc ... --output-format csv run.
usage: python tools/pmc_summary.py <dir with *counter_collection.csv> [kernel-name substring]"""
import glob
import os
import sys

import pandas as pd

files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
df = pd.concat(pd.read_csv(f) for f in files)
if len(sys.argv) > 2:
    df = df[df["Kernel_Name"].str.contains(sys.argv[2], regex=False)]
df["Kernel_Name"] = df["Kernel_Name"].str.replace(r"^void mmf::", "", regex=True).str.replace(r"\(.*$", "", regex=True)
df["key"] = df["Kernel_Name"] + " grid=" + df["Grid_Size"].astype(str)
t = df.pivot_table(index="key", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
t["launches"] = df.groupby("key")["Dispatch_Id"].nunique()
if "End_Timestamp" in df.columns:
    df["us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
    t["us"] = df.groupby("key")["us"].mean()
if "GRBM_GUI_ACTIVE" in t.columns and "us" in t.columns:
    t["MHz"] = t["GRBM_GUI_ACTIVE"] / 8 / t["us"]  # rocprofv3 reports the sum over the 8 XCDs
if "GRBM_GUI_ACTIVE" in t.columns and "SQ_VALU_MFMA_BUSY_CYCLES" in t.columns:
    t["MfmaUtil%"] = 100.0 * t["SQ_VALU_MFMA_BUSY_CYCLES"] / (t["GRBM_GUI_ACTIVE"] / 8 * 1024)  # 1024 SIMDs
pd.set_option("display.width", 250, "display.max_columns", 30, "display.max_colwidth", 70, "display.float_format", "{:.1f}".format)
print(t.to_string())

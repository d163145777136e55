"""Digest of a rocprofv3 --pmc counter_collection.csv: per kernel (shortened name, grid, workgroup size), the mean of every
counter over its dispatches and the mean dispatch duration (the CSV's own begin / end stamps; under --pmc every dispatch runs
alone).  usage: pmc_summary.py <dir or csv> [name filter, default "wm::"]"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "wm::"
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if flt and flt not in name:
                continue
            short = name.split("(")[0].replace("void ", "")[:60]
            key = (short, row["Grid_Size"], row["Workgroup_Size"])
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[key][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
for key, ctrs in sorted(acc.items()):
    short, grid, wg = key
    d = list(dur[key].values())
    print(f"{short} grid={grid} wg={wg} dispatches={len(d)} mean_us={sum(d) / len(d):.1f} vgpr/lds see kernel-trace")
    for c, v in sorted(ctrs.items()):
        print(f"    {c:36s} mean {sum(v) / len(v):18.1f}")

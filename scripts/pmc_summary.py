"""Digest of a rocprofv3 --pmc counter_collection.csv: per kernel name (shortened), the mean of every counter over its
dispatches.  usage: pmc_summary.py <dir or csv> [name filter]"""
import csv, glob, os, sys, collections
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if flt and flt not in name:
                continue
            short = name.split("(")[0].replace("void ", "")[:60]
            acc[(short, row["Grid_Size"], row["Workgroup_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (short, grid, wg), ctrs in sorted(acc.items()):
    n = max(len(v) for v in ctrs.values())
    print(f"{short} grid={grid} wg={wg} dispatches={n}")
    for c, v in sorted(ctrs.items()):
        print(f"    {c:36s} mean {sum(v) / len(v):18.1f}")

"""Distribution of one kernel's durations in a rocprofv3 --kernel-trace CSV (the --stats summary gives the mean only):
    python scripts/trace_kernel_hist.py <kernel_trace.csv> [kernel-name substring]
Prints count, mean, percentiles, the share of launches that overlap another launch of the same kernel (begin < previous end on
another queue) and the mean of the ones that do not.  profiles/r5d_cross_attn_trace_hist.txt is its output for the driver's command."""
import csv
import sys

import numpy as np

path, key = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "attn_cross_kernel<1, false, 0")
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        if key in r.get("Kernel_Name", ""):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")))
rows.sort()
b = np.array([r[0] for r in rows], dtype=np.int64)
e = np.array([r[1] for r in rows], dtype=np.int64)
d = (e - b) / 1e3
prev_end = np.maximum.accumulate(np.concatenate([[0], e[:-1]]))
next_begin = np.concatenate([b[1:], [np.iinfo(np.int64).max]])
overlapped = (b < prev_end) | (next_begin < e)
print(f"{path}: {len(d)} launches of *{key}*")
print(f"mean {d.mean():.1f} us, median {np.median(d):.1f}, p10 {np.percentile(d, 10):.1f}, p90 {np.percentile(d, 90):.1f}, p99 {np.percentile(d, 99):.1f}, max {d.max():.1f}")
print(f"launches that overlap another launch of the kernel in time: {overlapped.mean() * 100:.2f} % (mean {d[overlapped].mean() if overlapped.any() else 0:.1f} us)")
print(f"launches alone on the chip: mean {d[~overlapped].mean():.1f} us, median {np.median(d[~overlapped]):.1f} us ({(~overlapped).sum()} launches)")
for lo, hi in ((0, 212), (212, 216), (216, 220), (220, 230), (230, 260), (260, 400), (400, 1e9)):
    print(f"  {lo:>4.0f} - {hi if hi < 1e8 else float('inf'):>6} us: {((d >= lo) & (d < hi)).mean() * 100:6.2f} %")

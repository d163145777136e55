"""Where a decode step's time goes, launch by launch, from a rocprofv3 --kernel-trace CSV of `bench.py --batch B`:
    python scripts/step_timeline.py <kernel_trace.csv>
Takes the steady part of the trace (the decode loop: the most frequent kernels), and prints per kernel name the launches per token
step, the mean duration and the mean gap to the end of the launch before it (begin - previous end: what the launch boundary
costs under the profiler, which serialises dispatches -- an upper bound of the graph's own gaps)."""
import csv
import sys
from collections import defaultdict

import numpy as np

path = sys.argv[1]
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
greedy = [i for i, n in enumerate(names) if "greedy" in n]
per = defaultdict(lambda: [0, 0.0, 0.0])
n_steps = 0
# a token step = the launches between two greedy kernels that are a whole decoder pass apart (>= 100 launches)
for a, b in zip(greedy[:-1], greedy[1:]):
    if b - a < 100 or b - a > 400:
        continue
    n_steps += 1
    for i in range(a + 1, b + 1):
        short = names[i].split("(")[0][:90]
        per[short][0] += 1
        per[short][1] += (rows[i][1] - rows[i][0]) / 1e3
        per[short][2] += max(0, rows[i][0] - rows[i - 1][1]) / 1e3
print(f"{path}: {n_steps} token steps")
tot_d = tot_g = 0.0
for k, (n, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print(f"{n / n_steps:7.1f} per step  dur {d / n:7.2f} us  gap before {g / n:6.2f} us  per step {d / n_steps:8.1f} + {g / n_steps:7.1f} us  {k}")
    tot_d += d / n_steps; tot_g += g / n_steps
print(f"per step: kernels {tot_d:.1f} us + gaps {tot_g:.1f} us = {tot_d + tot_g:.1f} us")

"""Gaps between consecutive kernels of one queue, from a rocprofv3 --kernel-trace CSV (decode loop analysis).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline
    python scripts/trace_gaps.py /tmp/tr

Per queue, kernels sorted by start time; for each (previous kernel -> next kernel) pair the gap start(next) - end(previous)
and the durations.  Only pairs inside the decode chain are reported (both names from the decode kernels).
"""
import csv, glob, os, sys, collections
import numpy as np

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
assert files, f"no kernel_trace.csv under {root}"
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
print(f"{len(rows)} kernels in {len(files)} file(s)")

def short(n):
    n = n.replace("void ", "").replace("wm::", "")
    return n.split("(")[0][:40]

DECODE = ("gemm_skinny", "row_finish", "attn_self", "attn_cross_kernel<1", "embed", "greedy", "step_advance", "layernorm")
by_q = collections.defaultdict(list)
for q, s, e, n in rows:
    by_q[q].append((s, e, short(n)))
gaps = collections.defaultdict(list)
durs = collections.defaultdict(list)
for q, ks in by_q.items():
    ks.sort()
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        if any(d in n0 for d in DECODE) and any(d in n1 for d in DECODE):
            gaps[(n0, n1)].append((s1 - e0) / 1e3)
    for s, e, n in ks:
        if any(d in n for d in DECODE):
            durs[n].append((e - s) / 1e3)
print("\nkernel durations (us): count, mean, median, p90")
for n, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    print(f"  {n:42s} {len(v):8d} {v.mean():9.1f} {np.median(v):9.1f} {np.percentile(v, 90):9.1f}")
print("\ngaps end(prev) -> start(next) on one queue (us): count, mean, median, p90")
tot = 0.0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    if len(v) < 50:
        continue
    tot += v.sum()
    print(f"  {k[0]:34s} -> {k[1]:34s} {len(v):7d} {v.mean():8.1f} {np.median(v):8.1f} {np.percentile(v, 90):8.1f}")
print(f"\nqueues: {len(by_q)}; kernels per queue: {sorted(len(v) for v in by_q.values())[-8:]}")

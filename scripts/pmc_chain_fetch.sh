#!/bin/bash
# HBM bytes of the one-launch token step (gemv_chain_kernel): rocprofv3 --pmc FETCH_SIZE (counters only, a run of its own) over bench.py
# --batch B, summarised per launch in the format bench.py reads back as roofline.traffic (profiles/r<N>*_pmc_b<B>_chain_fetch.txt).
#   bash scripts/pmc_chain_fetch.sh <batch> <out.txt>
set -e
B=${1:-1}; OUT=${2:-/dev/stdout}
export TMPDIR=/tmp
D=$(mktemp -d /tmp/wm_pmc_chain_XXXX)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D -- python3 bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --length-dist forced --encoder-cus 0 > /dev/null 2>&1
python3 - "$D" "$B" > "$OUT" <<'PY'
import csv, glob, sys
import numpy as np
d, B = sys.argv[1], int(sys.argv[2])
vals = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if "gemv_chain_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == "FETCH_SIZE":
            vals.append(float(row["Counter_Value"]))
v = np.array(vals)
print(f"gemv_chain_kernel launches {len(v)}")
print(f"FETCH_SIZE per launch (KB): median {np.median(v):.0f}  min {v.min():.0f}  max {v.max():.0f}")
print(f"HBM bytes per launch = KB x 1024 x 2 (gfx950 correction for 16 B/lane streaming reads): {np.median(v) * 1024 * 2 / 1e6:.1f} MB")
print(f"algorithmic bytes of a batch-{B} token step inside the launch (large-v2 int8): Linear weights 734.0 MB + cross K/V {B} x 245.76 MB + "
      f"self-attention cache ~ {B} x 2 x 32 x 1280 x T B (T ~ 70: {B * 5.7:.1f} MB) = ~ {734.0 + B * 245.76 + B * 5.7:.0f} MB")
PY
rm -rf $D

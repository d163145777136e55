#!/bin/bash
# round 6: (1) HBM bytes of the one-launch token step at 1..8 rows (rocprofv3 --pmc FETCH_SIZE, a run of its own per batch) so that roofline.traffic is never null;
# (2) the overlap closure table (scripts/overlap_closure.py)
export TMPDIR=/tmp
for b in 1 2 3 4 5 6 7 8; do
  bash scripts/pmc_chain_fetch.sh $b gpurun_out/r6h_pmc_b${b}_chain_fetch.txt; cat gpurun_out/r6h_pmc_b${b}_chain_fetch.txt
done
python scripts/overlap_closure.py 576 128 12 > gpurun_out/r6h_overlap_closure.txt 2> gpurun_out/r6h_overlap_closure.err; cat gpurun_out/r6h_overlap_closure.txt; tail -3 gpurun_out/r6h_overlap_closure.err

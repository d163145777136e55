#!/bin/bash
# B = 32: per-kernel statistics of the row-split form (WM_ROWS_MIN=16) next to the split-K chain (r3ae_b32_g1_kernel_stats.csv)
out=gpurun_out/r3ak; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "b32_rows|--batch 32 --groups 1|WM_ROWS_MIN=16" "b32_small32|--batch 32 --groups 1|WM_SMALL_PATH=32"; do
  IFS='|' read name args envs <<< "$cfg"
  export $envs
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$name -o $name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --no-roofline --encoder-cus 0 --length-dist forced $args > $R/$out/bench_$name.json 2> $R/$out/bench_$name.err
  unset ${envs%%=*}
  f=$(find $R/$out/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/$out/${name}_kernel_stats.csv && head -12 $f | cut -c1-150
  rm -rf $R/$out/prof_$name
done

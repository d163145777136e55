#!/bin/bash
# per-kernel statistics of the small-batch decode loops (B = 32 in one group / two groups, B = 8)
out=gpurun_out/r3ae; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "b32_g1|--batch 32 --groups 1" "b32_g2|--batch 32 --groups 2" "b8|--batch 8"; do
  name=${cfg%%|*}; args=${cfg#*|}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$name -o $name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --no-roofline --encoder-cus 0 --length-dist forced $args > $R/$out/bench_$name.json 2> $R/$out/bench_$name.err
  f=$(find $R/$out/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/$out/${name}_kernel_stats.csv && head -14 $f | cut -c1-150
  rm -rf $R/$out/prof_$name
done

#!/bin/bash
# A/B runs of bench.py in one gpurun call (one box: the engines are built once, the variants share them).
#   scripts/ab_bench.sh <tag> "<ENV=.. ENV=..>|<extra args>" ...      each variant: "name|env assignments|bench args"
# Lines land in gpurun_out/<tag>_<name>.json; digest by scripts/sumline.py.
tag=$1; shift
mkdir -p gpurun_out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced"
for v in "$@"; do
    name=$(echo "$v" | cut -d'|' -f1); envs=$(echo "$v" | cut -d'|' -f2); args=$(echo "$v" | cut -d'|' -f3)
    env $envs timeout 900 python bench.py $common $args > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err
    python scripts/sumline.py gpurun_out/${tag}_${name}.json
done

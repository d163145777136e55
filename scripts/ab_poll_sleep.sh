#!/bin/bash
# Lab: the sleep between two polling passes of the one-launch step's sweeps (s_sleep N in sweep_granules16): builds a variant per N and
# runs bench.py --batch 1 / 2 against each, interleaved.   bash scripts/ab_poll_sleep.sh "0 1 2 4"
set -e
for n in $1; do
  sed "s/__builtin_amdgcn_s_sleep(2);/__builtin_amdgcn_s_sleep($n);/" eddie-wang-hackathon2023_amd/csrc/gemv_chain.hip > build/lab/gemv_chain_sleep$n.hip
  SRC=build/lab/gemv_chain_sleep$n.hip scripts/lab/build_variant.sh sleep$n gemv_chain.hip > /dev/null
done

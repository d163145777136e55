#!/bin/bash
out=gpurun_out/r3i; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python scripts/chain_probe.py --rows-path 1 > $out/chain_rows.log 2>&1; tail -16 $out/chain_rows.log
timeout 600 python scripts/chain_probe.py --rows-path 0 > $out/chain_splitk.log 2>&1; tail -16 $out/chain_splitk.log
timeout 600 python scripts/chain_probe.py --rows-path 1 --batch 192 --groups 1 > $out/chain_rows_alone.log 2>&1; tail -16 $out/chain_rows_alone.log
timeout 600 python scripts/chain_probe.py --rows-path 0 --batch 192 --groups 1 > $out/chain_splitk_alone.log 2>&1; tail -16 $out/chain_splitk_alone.log
for r in 0 1 0 1; do echo "WM_GEMM_TILE_ROWS=$r"; WM_GEMM_TILE_ROWS=$r timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep TFLOP; done > $out/bench_gemm_ab.log 2>&1; cat $out/bench_gemm_ab.log
for r in 0 1; do
  d=/tmp/pmc_gemm_$r; rm -rf $d
  REPS=2 WM_GEMM_TILE_ROWS=$r timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -- python3 scripts/bench_gemm.py 128 > $out/pmc_gemm_$r.log 2>&1
  echo "# REPS=2 WM_GEMM_TILE_ROWS=$r rocprofv3 --pmc FETCH_SIZE -- python3 scripts/bench_gemm.py 128   (0 = the round-3 tile order, 1 = plain row-major)" > $out/pmc_gemm_fetch_tile_rows_$r.txt
  python3 scripts/pmc_summary.py $d >> $out/pmc_gemm_fetch_tile_rows_$r.txt; cat $out/pmc_gemm_fetch_tile_rows_$r.txt
done

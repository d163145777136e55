#!/bin/bash
# L1 <-> L2 traffic of the encoder GEMM (rocprofv3 --pmc, one pass per counter set): does a K stage of 64-byte rows fetch every 128-byte
# line twice?  scripts/pmc_gemm_l2.sh <tag> [clips]   (M = 1500 x clips)
tag=${1:-r5ai}; clips=${2:-128}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" | sort -u > $out/counters_avail.txt
i=0
for pass in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TA_DATA_STALL_CYCLES_sum" "FETCH_SIZE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
    i=$((i + 1))
    d=/tmp/pmc_${tag}_$i; rm -rf $d
    REPS=2 timeout 600 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 scripts/bench_gemm.py $clips > $out/pass$i.log 2>&1
    echo "# rocprofv3 --pmc $pass -- python3 scripts/bench_gemm.py $clips (REPS=2)" > $out/pmc_pass$i.txt
    python3 scripts/pmc_summary.py $d gemm_f16p >> $out/pmc_pass$i.txt 2>&1
    cat $out/pmc_pass$i.txt
done

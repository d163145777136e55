#!/bin/bash
# Round 3 counter passes (rocprofv3 --pmc, one pass per counter set, nothing else traced) over ONE group-sized batch through the
# engines: scripts/stage_times.py --batch 192 (encoder at M = 288 000, cross-K/V projection, language pass, prefill and a few
# token steps of a 192-row group = the bench's launch shapes).  Digests land in gpurun_out/<tag>/pmc_<pass>.txt.
#   scripts/pmc_round3.sh <tag> [decode steps]
tag=${1:-r3pmc}; steps=${2:-6}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python3 scripts/stage_times.py --batch 192 --decode-steps $steps --reps 1 > $out/plain_run.log 2>&1      # engines built and cached first
i=0
for pass in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"; do
    i=$((i + 1))
    d=/tmp/pmc_${tag}_$i; rm -rf $d
    timeout 900 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 scripts/stage_times.py --batch 192 --decode-steps $steps --reps 1 > $out/pass$i.log 2>&1
    echo "# rocprofv3 --pmc $pass -- python3 scripts/stage_times.py --batch 192 --decode-steps $steps --reps 1" > $out/pmc_pass$i.txt
    python3 scripts/pmc_summary.py $d >> $out/pmc_pass$i.txt 2>&1
    tail -2 $out/pass$i.log
done

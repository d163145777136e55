#!/bin/bash
out=gpurun_out/r3aj; mkdir -p $out
timeout 300 python scripts/host_issue_probe.py 1 8 > $out/host_issue_cap0.log 2>&1; tail -6 $out/host_issue_cap0.log
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout 300 python scripts/host_issue_probe.py 1 8 > $out/host_issue_cap1.log 2>&1; tail -6 $out/host_issue_cap1.log
timeout 300 python scripts/bench_gemv_warm.py > $out/gemv_warm.log 2>&1; tail -4 $out/gemv_warm.log

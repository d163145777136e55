#!/bin/bash
bash scripts/ab_bench.sh r3o "g3_e96|WM_CROSS_PERSIST_WGS=240|" "g2_e96|WM_CROSS_PERSIST_WGS=240|--groups 2" "g3_e8|WM_CROSS_PERSIST_WGS=240|--encoder-cus 8" "g3_e96_q8|WM_CROSS_PERSIST_WGS=240 GPU_MAX_HW_QUEUES=8|" "g2_e128|WM_CROSS_PERSIST_WGS=240|--groups 2 --encoder-cus 128"

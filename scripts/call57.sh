#!/bin/bash
# persistent GEMM with whole-stage intervals (two barriers per stage, 32-MFMA bursts: -DWM_GEMM_WHOLE_STAGE) against the product kernel
out=gpurun_out/r3az; mkdir -p $out
cd eddie-wang-hackathon2023_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc -I../../include -DWM_GEMM_WHOLE_STAGE -c gemm_f16p.hip -o /tmp/gemm_f16p_ws.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libwm_ws.so engine.o gemm_f16.o /tmp/gemm_f16p_ws.o gemm_skinny.o gemv_small.o gemm_rows.o rowops.o attn_encoder.o attn_decode.o greedy.o frontend.o flac_decode.o || exit 1
cd ../..
for r in 1; do
  echo "== product"; REPS=20 timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep -v amdgpu.ids | tee -a $out/bench_gemm_product.log
  echo "== whole-stage intervals"; WM_LIBRARY_PATH=/tmp/libwm_ws.so REPS=20 timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep -v amdgpu.ids | tee -a $out/bench_gemm_whole_stage.log
done
echo "== product, zero operands"; ZERO_DATA=1 REPS=20 timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep -v amdgpu.ids | tee -a $out/bench_gemm_product_zero.log
echo "== whole-stage intervals, zero operands"; ZERO_DATA=1 WM_LIBRARY_PATH=/tmp/libwm_ws.so REPS=20 timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep -v amdgpu.ids | tee -a $out/bench_gemm_whole_stage_zero.log
echo "== kernel tests on the variant"; WM_LIBRARY_PATH=/tmp/libwm_ws.so timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -3

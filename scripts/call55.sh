#!/bin/bash
# interval-by-interval time stamps of the persistent GEMM's K loop (diagnostic builds next to the product library)
out=gpurun_out/r3ay; mkdir -p $out
cd eddie-wang-hackathon2023_amd/csrc
for v in "0xA5|0" "0xA5|1" "0xA5|2" "0xA5|3" "0x367|0" "0x367|3"; do
  mask=${v%%|*}; skip=${v##*|}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc -I../../include -DWM_GEMM_STAMPS=$mask -DWM_GEMM_STAMP_SKIP=$skip -c gemm_f16p.hip -o /tmp/gemm_f16p_st_${mask}_$skip.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libwm_stamps_${mask}_$skip.so engine.o gemm_f16.o /tmp/gemm_f16p_st_${mask}_$skip.o gemm_skinny.o gemv_small.o gemm_rows.o rowops.o attn_encoder.o attn_decode.o greedy.o frontend.o flac_decode.o || exit 1
done
cd ../..
timeout 300 python scripts/gemm_interval_stamps.py > $out/gemm_intervals_product.log 2>&1; cat $out/gemm_intervals_product.log | grep -v amdgpu.ids
for v in "0xA5|0" "0xA5|1" "0xA5|2" "0xA5|3" "0x367|0" "0x367|3"; do
  mask=${v%%|*}; skip=${v##*|}
  echo "== stamps $mask, skipped in the stamped stages: $skip (1 = DMA requests, 2 = fragment reads, 3 = both)"
  WM_LIBRARY_PATH=/tmp/libwm_stamps_${mask}_$skip.so STAMP_MASK=$mask timeout 300 python scripts/gemm_interval_stamps.py > $out/gemm_intervals_${mask}_skip$skip.log 2>&1; cat $out/gemm_intervals_${mask}_skip$skip.log | grep -v amdgpu.ids
done

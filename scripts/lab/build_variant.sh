#!/bin/bash
# Lab: build a variant of the library in which ONE translation unit is compiled with extra flags, for A/B runs next to the
# product library (WM_LIBRARY_PATH=build/lab/libwm_<name>.so python scripts/bench_....py).  Nothing here ships.
#   scripts/lab/build_variant.sh <name> <file.hip> [extra hipcc flags...]
#   SRC=<path to a modified copy> scripts/lab/build_variant.sh <name> <file.hip> ...   compiles the copy in place of csrc/<file.hip>
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$root/eddie-wang-hackathon2023_amd/csrc
out=$root/build/lab; mkdir -p $out
make -C $csrc >/dev/null
base=${src%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc -I$root/include "$@" \
    -I$csrc -c ${SRC:-$csrc/$src} -o $out/${base}_$name.o
others=$(ls $csrc/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libwm_$name.so $others $out/${base}_$name.o
echo $out/libwm_$name.so

#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/exp13
mkdir -p $OUT
cd $ROOT
T=tests/native/vrdx_selftest
for v in base w4 w12 w16 w24; do
  if [ $v = base ]; then L=$ROOT/vulkan_radix_sort_amd; else L=$ROOT/build/variants/$v; fi
  for c in 1024x32x2 1024x32; do
    echo "=== window=$v config=$c" | tee -a $OUT/bench.log
    LD_LIBRARY_PATH=$L VRDX_TILE_CONFIG=$c timeout 120 $T bench 24 25 26 27 2>&1 | grep -E "keys|kv" | tee -a $OUT/bench.log
  done
done

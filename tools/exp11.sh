#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/exp11
mkdir -p $OUT
cd $ROOT
T=tests/native/vrdx_selftest
timeout 200 $T bench 18 19 20 21 22 23 24 25 26 27 2>&1 | tee $OUT/bench_default.log | tail -22
for c in 1024x32 1024x32x2; do
  echo "=== bench $c" | tee -a $OUT/bench.log
  VRDX_TILE_CONFIG=$c timeout 200 $T bench 24 25 26 2>&1 | grep -E "keys|kv" | tee -a $OUT/bench.log
done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.log

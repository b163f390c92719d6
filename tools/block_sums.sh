#!/bin/bash
# Block sums against the classic look-back over the sorts of one round, ON the GPU box:
#   gpurun -- 'bash tools/block_sums.sh [out-name]'
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-block_sums}.txt
mkdir -p "$ROOT/gpurun_out"
for mode in keys kv; do
  for knob in 1 0; do
    for rep in 1 2; do
      echo "=== $mode VRDX_BLOCK_SUMS=$knob (run $rep)" | tee -a "$OUT"
      VRDX_BLOCK_SUMS=$knob $ROOT/tests/native/vrdx_selftest lsweep 7864320 17301504 19 $mode 2>&1 | tail -n +3 | tee -a "$OUT"
    done
  done
done

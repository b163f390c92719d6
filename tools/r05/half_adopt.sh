#!/bin/bash
# After adopting the half-size bucket kernel (keys-only and key+value from 8.14 M): GPU tests, native parity, the plan
# table around the old and new boundaries.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT; OUT=gpurun_out/r05_half; mkdir -p $OUT
T=tests/native/vrdx_selftest
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 900 $T parity > $OUT/native_parity.txt 2>&1; tail -2 $OUT/native_parity.txt
timeout 900 $T msd 8144129 12000003 18325504 18325505 > $OUT/msd_parity.txt 2>&1; tail -2 $OUT/msd_parity.txt
timeout 600 $T bench 8144128 8144129 8388608 9437184 10485760 12582912 14680064 16252928 16777216 18325504 18325505 18874368 > $OUT/table_adopted.txt 2>&1
cat $OUT/table_adopted.txt

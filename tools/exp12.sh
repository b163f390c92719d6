#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/exp12
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_batched_gpu.py -x -q -m gpu 2>&1 | tail -25 | tee $OUT/pytest_batched.log
timeout 200 tests/native/vrdx_selftest bench 24 25 2>&1 | tee $OUT/bench_default.log | tail -6
timeout 300 python bench.py --steps 5 2>&1 | tail -3 | tee $OUT/bench_py.log

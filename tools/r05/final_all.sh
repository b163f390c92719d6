#!/bin/bash
# Everything that has to be re-measured on the frozen tree: GPU tests, bench line + rocprofv3 + PMC, plan table, adversarial,
# the bench driver's sweep for this library.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r05final; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
bash tools/r05/final_numbers.sh > $OUT/final_numbers.log 2>&1; tail -30 $OUT/final_numbers.log
bash tools/r05/msd_plan.sh > /dev/null 2>&1
VRDX_SELFTEST_PATTERNS=6 timeout 600 tests/native/vrdx_selftest adversarial 25 > $OUT/adversarial_msd.txt 2>&1
VRDX_SELFTEST_PATTERNS=6 VRDX_MSD=0 timeout 600 tests/native/vrdx_selftest adversarial 25 > $OUT/adversarial_four_passes.txt 2>&1
timeout 900 bench/bench hip --no-verify -o $OUT/bench_driver_hip.csv > $OUT/bench_driver_hip.log 2>&1
timeout 300 tests/native/vrdx_selftest soak 120 > $OUT/soak.txt 2>&1; tail -1 $OUT/soak.txt
timeout 900 tests/native/vrdx_selftest bench 15 16 17 18 19 20 21 22 23 24 25 26 27 > $OUT/native_sweep.txt 2>&1

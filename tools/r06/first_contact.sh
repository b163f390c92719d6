#!/bin/bash
# First contact of the window-choosing MSD plan with the GPU: the quick parity battery, the MSD battery at two sizes, the
# adversarial table at 2^25, the headline sizes.
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_first}; mkdir -p $OUT
cd $ROOT
timeout 600 tests/native/vrdx_selftest quick > $OUT/quick.txt 2>&1; echo "quick rc=$?" | tee -a $OUT/summary.txt
timeout 900 tests/native/vrdx_selftest msd ${SIZES:-8144129 33554432} > $OUT/msd.txt 2>&1; echo "msd rc=$?" | tee -a $OUT/summary.txt
timeout 600 tests/native/vrdx_selftest adversarial 25 > $OUT/adversarial.txt 2>&1; echo "adversarial rc=$?" | tee -a $OUT/summary.txt
timeout 300 tests/native/vrdx_selftest bench 23 24 25 26 > $OUT/bench.txt 2>&1
tail -5 $OUT/quick.txt; grep -v "^ok" $OUT/msd.txt | tail -30; cat $OUT/adversarial.txt $OUT/bench.txt

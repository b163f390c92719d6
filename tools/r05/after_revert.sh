#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_after_revert; mkdir -p $OUT
cd $ROOT
T=tests/native/vrdx_selftest
VRDX_SELFTEST_PATTERNS=6 timeout 600 $T adversarial 25 > $OUT/adversarial_msd.txt 2>&1
VRDX_SELFTEST_PATTERNS=6 VRDX_MSD=0 timeout 600 $T adversarial 25 > $OUT/adversarial_four_passes.txt 2>&1
cat $OUT/adversarial_msd.txt $OUT/adversarial_four_passes.txt
# where does the MSD plan start to pay?  19 sizes from 7.9 M to 17.3 M, with the plan recorded from 1 element up and as shipped
for mode in keys kv; do
  for from in 1 0; do
    echo "=== $mode VRDX_MSD_FROM=$from" >> $OUT/threshold.txt
    if [ $from = 1 ]; then VRDX_MSD_FROM=1 timeout 600 $T lsweep 7864320 17301504 19 $mode >> $OUT/threshold.txt 2>&1; else timeout 600 $T lsweep 7864320 17301504 19 $mode >> $OUT/threshold.txt 2>&1; fi
  done
done
cat $OUT/threshold.txt
bash tools/r05/ceiling.sh > $OUT/ceiling.log 2>&1
tail -60 $OUT/ceiling.log

#!/bin/bash
# bucket_sort2_half_kernel (512 threads, buckets <= 16384, two workgroups per CU) against the full-size kernel: parity first,
# then the plan table at 8.4 ... 16.2 M with VRDX_MSD_HALF=1 | 0, keys-only and (VRDX_MSD_FROM lowered) key+value.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT; OUT=gpurun_out/r05_half; mkdir -p $OUT
T=tests/native/vrdx_selftest
VRDX_MSD_FROM=8150000 timeout 600 $T msd 8388608 12000003 16252928 > $OUT/parity.txt 2>&1; tail -3 $OUT/parity.txt
for half in 1 0; do
  echo "== VRDX_MSD_HALF=$half (keys-only: default thresholds)"
  VRDX_MSD_HALF=$half timeout 300 $T bench 8388608 9437184 10485760 12582912 14680064 16252928 2>&1 | grep -E "keys"
  echo "== VRDX_MSD_HALF=$half VRDX_MSD_FROM=8150000 (key+value)"
  VRDX_MSD_HALF=$half VRDX_MSD_FROM=8150000 timeout 300 $T bench 8388608 9437184 10485760 12582912 14680064 16252928 2>&1 | grep -E " kv "
done > $OUT/table.txt 2>&1
echo "== default (nine-bit hybrid / four passes for key+value)" >> $OUT/table.txt
timeout 300 $T bench 8388608 9437184 10485760 12582912 14680064 16252928 2>&1 | grep -E " kv " >> $OUT/table.txt
cat $OUT/table.txt

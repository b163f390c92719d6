#!/bin/bash
# Round 5, first contact of the MSD plan with the hardware: parity, then timings with and without it, then per-kernel stats.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_msd_first
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
T=tests/native/vrdx_selftest
timeout 900 $T msd 16252929 20000003 33554432 > $OUT/parity.txt 2>&1
tail -3 $OUT/parity.txt
for msd in 1 0; do
  echo "=== VRDX_MSD=$msd" >> $OUT/bench.txt
  VRDX_MSD=$msd timeout 600 $T bench 20000000 25 >> $OUT/bench.txt 2>&1
done
cat $OUT/bench.txt
cd /tmp && rm -rf /tmp/pr_msd && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_msd -o t -- $ROOT/$T bench 25 > /dev/null 2>&1
S=$(find /tmp/pr_msd -name '*kernel_stats.csv' | head -1)
cp "$S" $OUT/kernel_stats_2pow25.csv
cut -d, -f1-8 "$S" | head -20

#!/bin/bash
# Round 5: the persistent scatter / bucket kernels of the MSD plan -- parity, per-kernel averages, grid sizes.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_msd_fourth
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
T=$ROOT/tests/native/vrdx_selftest
timeout 1200 $T msd 16252929 20000003 33554432 > $OUT/parity.txt 2>&1
tail -4 $OUT/parity.txt
prof() {
  local tag=$1; shift
  rm -rf /tmp/pr_$tag
  (cd /tmp && export "$@" && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$tag -o t -- $T bench 25 > $OUT/bench_$tag.txt 2>&1)
  local S=$(find /tmp/pr_$tag -name '*kernel_stats.csv' | head -1)
  echo "=== $tag ($*)" >> $OUT/kernels.txt
  python3 - "$S" >> $OUT/kernels.txt <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Name']).replace('void vrdx::', '')
    if 'msd' in name or 'sort2' in name or 'onesweep' in name or 'histogram' in name:
        print(f"{name:55s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
}
prof counted VRDX_MSD=1
cat $OUT/kernels.txt
for g in 256 512 128; do
  echo "=== VRDX_MSD_GRID=$g" >> $OUT/bench.txt
  VRDX_MSD_GRID=$g timeout 600 $T bench 20000000 25 >> $OUT/bench.txt 2>&1
done
cat $OUT/bench.txt

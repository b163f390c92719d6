#!/bin/bash
# Round 5: the MSD plan after the first fixes -- parity, then a matrix of (bits, tile) x (keys, kv) with per-kernel averages.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_msd_second
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
T=$ROOT/tests/native/vrdx_selftest
timeout 1200 $T msd 16252929 20000003 33554432 > $OUT/parity.txt 2>&1
tail -4 $OUT/parity.txt
VRDX_MSD_TILE=16384 timeout 600 $T msd 20000003 > $OUT/parity_tile16384.txt 2>&1
tail -2 $OUT/parity_tile16384.txt
VRDX_MSD_BITS=11 timeout 600 $T msd 20000003 > $OUT/parity_bits11.txt 2>&1
tail -2 $OUT/parity_bits11.txt
prof() {  # prof <tag> <env...> -- per-kernel averages of `bench 25` under rocprofv3
  local tag=$1; shift
  rm -rf /tmp/pr_$tag
  (cd /tmp && env "$@" true; for kv in 0; do :; done)
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
  grep -E "^33554432" $OUT/bench_$tag.txt >> $OUT/kernels.txt
}
prof b10_t32 VRDX_MSD=1
prof b10_t16 VRDX_MSD_TILE=16384
prof b11_t32 VRDX_MSD_BITS=11
prof b11_t16 VRDX_MSD_BITS=11 VRDX_MSD_TILE=16384
prof off VRDX_MSD=0
cat $OUT/kernels.txt
# where does the plan start to pay?  (VRDX_MSD_FROM=1: recorded for every size the capacity allows)
for msd in 0 1; do
  echo "=== VRDX_MSD=$msd VRDX_MSD_FROM=1" >> $OUT/sizes.txt
  VRDX_MSD=$msd VRDX_MSD_FROM=1 timeout 900 $T bench 21 22 23 12582912 24 20000000 25165824 25 40000000 26 >> $OUT/sizes.txt 2>&1
done
cat $OUT/sizes.txt

#!/bin/bash
# Round 5, item 3 of the round-4 review: the ceiling of the pass-shaped kernels re-measured on the tree that ships.
# NOTE (round 6): the VRDX_ABLATE sites this script builds with were removed from vrdx_kernels.hip; run it on the tree of commit 9caa359.
# Builds timing-ablation variants of the library (VRDX_ABLATE, vrdx_kernels.hip) ON the box, runs the four-pass plan
# (VRDX_MSD=0) at 2^25 under rocprofv3 for each, then the product at 2^26 and 2^27.  Writes gpurun_out/r05_ceiling/.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_ceiling
mkdir -p $OUT
cd $ROOT/vulkan_radix_sort_amd/csrc
export TMPDIR=/tmp
T=$ROOT/tests/native/vrdx_selftest
for v in base:0 nolookback_linear:257 nolookback_noticket_linear:265 linear:256 noticket:8; do
  name=${v%%:*}; bits=${v#*:}
  mkdir -p /tmp/ceil_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVRDX_ABLATE=$bits -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o /tmp/ceil_$name/libvrdx_hip.so 2> $OUT/build_$name.err &
done
wait
summ() {  # summ <kernel_stats.csv>
  python3 - "$1" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Name']).replace('void vrdx::', '')
    if ('onesweep' in name or 'histogram' in name or 'msd' in name or 'sort2' in name) and float(r['AverageNs']) > 20000:
        print(f"    {name:55s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
}
: > $OUT/ceiling.txt
for v in base nolookback_linear nolookback_noticket_linear linear noticket; do
  rm -rf /tmp/pc_$v
  (cd /tmp && LD_LIBRARY_PATH=/tmp/ceil_$v VRDX_MSD=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$v -o t -- $T bench 25 > $OUT/bench_$v.txt 2>&1)
  echo "=== four passes at 2^25, variant $v" >> $OUT/ceiling.txt
  summ "$(find /tmp/pc_$v -name '*kernel_stats.csv' | head -1)" >> $OUT/ceiling.txt
done
for lg in 26 27; do
  for msd in 0 1; do
    rm -rf /tmp/pc_big
    (cd /tmp && VRDX_MSD=$msd timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_big -o t -- $T bench $lg > $OUT/bench_2pow${lg}_msd$msd.txt 2>&1)
    echo "=== product library at 2^$lg, VRDX_MSD=$msd" >> $OUT/ceiling.txt
    summ "$(find /tmp/pc_big -name '*kernel_stats.csv' | head -1)" >> $OUT/ceiling.txt
    grep -E "^[0-9]+ +(keys|kv)" $OUT/bench_2pow${lg}_msd$msd.txt | sed 's/^/    /' >> $OUT/ceiling.txt
  done
done
cat $OUT/ceiling.txt

#!/bin/bash
# Timing ablations of the MSD plan's kernels: builds variant libraries (on the box), runs `bench 25` under rocprofv3 for each.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_ablate
mkdir -p $OUT
cd $ROOT/vulkan_radix_sort_amd/csrc
export TMPDIR=/tmp
T=$ROOT/tests/native/vrdx_selftest
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p /tmp/var_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o /tmp/var_$name/libvrdx_hip.so 2> $OUT/build_$name.err &
done
wait
for v in "$@"; do
  name=${v%%:*}
  rm -rf /tmp/pr_$name
  (cd /tmp && LD_LIBRARY_PATH=/tmp/var_$name timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$name -o t -- $T bench 25 > $OUT/bench_$name.txt 2>&1)
  S=$(find /tmp/pr_$name -name '*kernel_stats.csv' | head -1)
  echo "=== $name ($v)" >> $OUT/kernels.txt
  python3 - "$S" >> $OUT/kernels.txt <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Name']).replace('void vrdx::', '')
    if 'msd' in name or 'sort2' in name or 'histogram' in name:
        print(f"{name:55s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
done
cat $OUT/kernels.txt

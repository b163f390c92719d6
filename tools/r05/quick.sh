#!/bin/bash
# bench 25 under rocprofv3 with the product library and given environment settings: quick.sh tag VAR=.. VAR=..
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_quick; mkdir -p $OUT
tag=$1; shift
rm -rf /tmp/pq_$tag
(cd /tmp && export TMPDIR=/tmp "$@" && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq_$tag -o t -- $ROOT/tests/native/vrdx_selftest bench ${SIZES:-25} > $OUT/bench_$tag.txt 2>&1)
S=$(find /tmp/pq_$tag -name '*kernel_stats.csv' | head -1)
echo "=== $tag ($*)" | tee -a $OUT/kernels.txt
python3 - "$S" <<'PY' | tee -a $OUT/kernels.txt
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Name']).replace('void vrdx::', '')
    if 'msd' in name or 'sort2' in name or 'histogram' in name or ('onesweep' in name and float(r['AverageNs']) > 20000):
        print(f"{name:55s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY

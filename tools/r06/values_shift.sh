#!/bin/bash
# Does the placement of the values scratch relative to the keys scratch (4 N bytes apart by default) matter to the key+value kernels?
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_values_shift}; mkdir -p $OUT
export TMPDIR=/tmp
SELF=$ROOT/tests/native/vrdx_selftest
for shift in 0 4096 65536 1052672 0 2101248 4198400 36864 0; do
  rm -rf /tmp/vs
  (cd /tmp && VRDX_X_VALUES_SHIFT=$shift timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vs -o t -- $SELF backtoback 25 kv 10) > $OUT/b2b_$shift.log 2>&1
  echo "== shift $shift: $(grep 'back to back' $OUT/b2b_$shift.log)"
  python3 - "$(find /tmp/vs -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"]).replace("void vrdx::", "")
    if "scatter" in name or "bucket" in name:
        print(f"   {name[:40]:40s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
PY
done

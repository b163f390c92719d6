#!/bin/bash
# NOTE: needs the experiment build in front of commit a1208b4 (VRDX_X_BUCKET_GRID=1 selected one bucket per workgroup); the product has no such knob.
# Two buckets per workgroup in the bucket launches of their own (key+value; the half-size kernel) against one (VRDX_X_BUCKET_GRID=1).
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_bucket_grid2}; mkdir -p $OUT
export TMPDIR=/tmp
SELF=$ROOT/tests/native/vrdx_selftest
for rep in 1 2; do
for whole in 0 1; do
  for cfg in "25 kv" "24 kv" "24 keys" "26 kv"; do
  set -- $cfg
  rm -rf /tmp/bg
  (cd /tmp && VRDX_X_BUCKET_GRID=$whole timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bg -o t -- $SELF backtoback $1 $2 10) > $OUT/b2b.log 2>&1
  echo "== one-bucket-per-workgroup=$whole: $(grep 'back to back' $OUT/b2b.log)"
  python3 - "$(find /tmp/bg -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"]).replace("void vrdx::", "")
    if "bucket" in name:
        print(f"   {name[:48]:48s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
  done
done
done
$SELF msd 8144200 20000003 33554432 45000000 | tail -1

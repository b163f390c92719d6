#!/bin/bash
# NOTE: needs the experiment build in front of commit a1208b4 (VRDX_X_BUCKET_GRID=1 selected one bucket per workgroup); the product has no such knob.
# The keys-only bucket launch that is also pass 1 of the fallback: 512 workgroups of two buckets each (default) against 1024 of one
# (VRDX_X_BUCKET_GRID=1, experiment knob of the build under test): uniform keys (the bucket role) and inputs the device turns down (the pass role).
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_bucket_grid}; mkdir -p $OUT
export TMPDIR=/tmp
SELF=$ROOT/tests/native/vrdx_selftest
for rep in 1 2; do
for whole in 0 1; do
  for lg in 25; do
  rm -rf /tmp/bg
  (cd /tmp && VRDX_X_BUCKET_GRID=$whole timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bg -o t -- $SELF backtoback $lg keys 10) > $OUT/b2b_$whole.log 2>&1
  echo "== one-bucket-per-workgroup=$whole 2^$lg: $(grep 'back to back' $OUT/b2b_$whole.log)"
  python3 - "$(find /tmp/bg -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"]).replace("void vrdx::", "")
    if "bucket" in name:
        print(f"   {name[:48]:48s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
  done
  VRDX_X_BUCKET_GRID=$whole VRDX_SELFTEST_PATTERNS=9 timeout 300 $SELF adversarial 25 | grep " keys " | awk '{print "   ", $1, $2, $3, $8, $9}'
done
done
VRDX_SELFTEST_PATTERNS=1 $SELF sweep 24.2 25.1 6 | grep "^[0-9]"
VRDX_X_BUCKET_GRID=1 $SELF sweep 24.2 25.1 6 | grep "^[0-9]"
$SELF msd 20000003 33554432 | tail -1

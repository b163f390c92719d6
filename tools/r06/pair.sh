#!/bin/bash
# The two-tile scatter (scatter_msd_pair_kernel, VRDX_MSD_PAIR=1) against the shipped one-tile scatter (VRDX_MSD_FUSED=0, so that
# NOTE: run on commit f1a3069, where the two-tile scatter is an experiment behind VRDX_MSD_PAIR=1 next to the one-tile form; from da5c39e on it IS the keys-only scatter.
# both run as kernels of their own): parity battery, kernel durations (rocprofv3 --kernel-trace --stats, ten keys-only sorts back to
# back), HBM bytes per launch (--pmc FETCH_SIZE / WRITE_SIZE in passes of their own).  -> gpurun_out/r06_pair/
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/${TAG:-r06_pair}; mkdir -p $OUT
export TMPDIR=/tmp
SELF=$ROOT/tests/native/vrdx_selftest
VRDX_MSD_PAIR=1 timeout 900 $SELF msd ${SIZES:-8144200 20000003 33554432 36000001} > $OUT/parity.txt 2>&1; echo "parity rc=$?"; tail -3 $OUT/parity.txt
for lg in 25 24; do
for pair in 0 1; do
  rm -rf /tmp/pp
  (cd /tmp && VRDX_MSD_PAIR=$pair VRDX_MSD_FUSED=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- $SELF backtoback $lg keys 10) > $OUT/b2b_${lg}_pair$pair.log 2>&1
  echo "== 2^$lg pair=$pair: $(grep 'back to back' $OUT/b2b_${lg}_pair$pair.log)"
  python3 - "$(find /tmp/pp -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"]).replace("void vrdx::", "")
    if "scatter" in name or "bucket" in name or "histogram" in name:
        print(f"   {name[:56]:56s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
done
done
for pair in 0 1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pc
    (cd /tmp && VRDX_MSD_PAIR=$pair VRDX_MSD_FUSED=0 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pc -o c -- $SELF backtoback 25 keys 3) > $OUT/pmc_${ctr}_pair$pair.log 2>&1
    python3 - "$(find /tmp/pc -name '*counter_collection.csv' | head -1)" $ctr $pair <<'PY'
import csv, sys, re, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void vrdx::", "")
    if "scatter" in name: acc[name].append(float(r["Counter_Value"]))
for name, v in acc.items():
    # units: FETCH_SIZE in 64-byte units x 32 (gfx950: 2048 B per count for wide reads, tools/pmc_report.py), WRITE_SIZE 1024 B... raw counts printed
    print(f"   pair={sys.argv[3]} {sys.argv[2]:10s} {name[:48]:48s} launches {len(v)} raw counts per launch {sum(v)/len(v):.0f}")
PY
  done
done

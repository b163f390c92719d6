#!/bin/bash
# NOTE: profiles/r05_msd_plan.txt was produced by this script on the tree of commit ea48965 (before the nine-bit plan was removed), where VRDX_MSD=0 still
# recorded round 4's nine-bit hybrid plan up to 16.2 M elements; on later trees VRDX_MSD=0 gives the four passes at every size and
# the "nine bits" block of part (2) shows only the pass kernels.
# profiles/r05_msd_plan.txt: (1) 20 sizes from 2^23 to 2^26, keys-only and key+value, reference protocol, with the per-stage
# times of the 15-slot timestamp contract; (2) what the scatter through memory costs by digit width -- nine bits (the nine-bit
# hybrid plan's scatter9_kernel), ten and eleven (scatter_msd_kernel) -- at 16 M elements, and ten against eleven at 2^25.
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_msd_plan; mkdir -p $OUT
export TMPDIR=/tmp
T=$ROOT/tests/native/vrdx_selftest
F=$OUT/msd_plan.txt
echo "# (1) vrdx_selftest bench: 1 warm-up + 10 timed runs on fresh mt19937 data, median; stage ms from the 15 timestamps" > $F
echo "#     (MSD plan: histogram | spine scatter buckets fallback; other plans: histogram | the four pass launches)" >> $F
SIZES="8388608 9437184 10485760 12582912 14680064 16252928 16777216 18325504 18325505 18874368 20971520 23068672 25165824 27262976 29360128 31457280 33554432 36600000 41943040 67108864"
timeout 1500 $T bench $SIZES >> $F 2>&1
echo "# the same sizes with VRDX_MSD=0 (round 4's plans: nine-bit hybrid up to 16.2 M, four passes beyond)" >> $F
VRDX_MSD=0 timeout 1500 $T bench $SIZES >> $F 2>&1
prof() {  # prof <n> <label> <env...>
  local n=$1 label=$2; shift 2
  rm -rf /tmp/pmp
  (cd /tmp && export "$@" && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmp -o t -- $T bench $n > /dev/null 2>&1)
  echo "=== N = $n, $label ($*)" >> $F
  python3 - "$(find /tmp/pmp -name '*kernel_stats.csv' | head -1)" >> $F <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Name']).replace('void vrdx::', '')
    if ('msd' in name or 'sort2' in name or 'scatter9' in name or 'bucket_sort_kernel' in name or 'histogram' in name) and int(r['Calls']) >= 11:
        print(f"    {name:55s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
}
echo "# (2) the scatter through memory by digit width (rocprofv3 --kernel-trace --stats of vrdx_selftest bench N; keys-only = ...false>, key+value = ...true>)" >> $F
prof 16000000 "nine bits: scatter9_kernel + bucket_sort_kernel<...,512> (three in-LDS passes)" VRDX_MSD=0
prof 16000000 "ten bits: scatter_msd_kernel<10> + bucket_sort2_half_kernel<10> (two in-LDS passes, two workgroups per CU)" VRDX_MSD=1
prof 16000000 "ten bits with the full-size bucket kernel, one workgroup per CU" VRDX_MSD_HALF=0
prof 16000000 "eleven bits (full-size bucket kernel)" VRDX_MSD_BITS=11
prof 33554432 "ten bits" VRDX_MSD=1
prof 33554432 "eleven bits" VRDX_MSD_BITS=11
cat $F

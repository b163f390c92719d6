#!/bin/bash
# Per-pass kernel durations (rocprofv3 --kernel-trace) of 2^25 sorts with the storage handed over at different offsets:
#   gpurun -- 'bash tools/pass_parity.sh "0 96" 20 [out-name]'
set -u
OFFSETS=${1:-"0 96"}
RUNS=${2:-20}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${3:-pass_parity}.txt
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
for mode in kv keys; do
  for off in $OFFSETS; do
    echo "=== $mode, storage offset $off" | tee -a "$OUT"
    rm -rf /tmp/pp_prof
    VRDX_SELFTEST_STORAGE_OFFSET=$off timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp_prof -o t -- \
        $ROOT/tests/native/vrdx_selftest passes 25 $mode $RUNS 2>&1 | grep -v "^W2\|^E2\|rocprof" | tee -a "$OUT"
    f=$(find /tmp/pp_prof -name "*kernel_trace.csv" | head -1)
    echo "  rocprofv3 kernel durations (us):" | tee -a "$OUT"
    python3 $ROOT/tools/pass_parity.py "$f" 1 | tee -a "$OUT"
  done
done

#!/bin/bash
# What the histogram kernel costs BEHIND another sort, and whether non-temporal stores in the previous sort's last pass
# change that (measurement variant build/variants/ntlast, -DVRDX_NT_LAST_PASS=1), ON the GPU box:
#   gpurun -- 'bash tools/hist_busy.sh [out-name]'
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-hist_busy}.txt
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
for mode in keys kv; do
  for lib in product ntlast; do
    for rep in 1 2; do
      echo "=== $mode, $lib library (run $rep): 10 sorts of 2^25 back to back" | tee -a "$OUT"
      rm -rf /tmp/hb_prof
      if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$ROOT/build/variants/ntlast; fi
      timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/hb_prof -o t -- \
          $ROOT/tests/native/vrdx_selftest backtoback 25 $mode 10 2>&1 | grep -v "^W2\|^E2\|rocprof" | tee -a "$OUT"
      f=$(find /tmp/hb_prof -name "*kernel_trace.csv" | head -1)
      python3 $ROOT/tools/pass_parity.py "$f" 1 | tee -a "$OUT"
    done
  done
done

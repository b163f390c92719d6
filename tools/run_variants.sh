#!/bin/bash
# Runs prebuilt variants (tools/build_variants.sh) ON the GPU box with the native selftest:
#   gpurun -- 'bash tools/run_variants.sh "base h0" "auto 512x32x2" "bench 24 25" [out-name]'
# $1 variant names, $2 VRDX_TILE_CONFIG values ("auto" = size-adaptive), $3 selftest arguments.
# PROF=1: under rocprofv3 --kernel-trace --stats, the per-kernel averages of the vrdx kernels are appended.
set -u
VARIANTS=$1
CONFIGS=${2:-auto}
ARGS=${3:-"bench 25"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${4:-variants}.log
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
for name in $VARIANTS; do
  d=$ROOT/build/variants/$name
  for c in $CONFIGS; do
    echo "=== variant=$name ($(cat $d/flags.txt)) config=$c args=$ARGS" | tee -a "$OUT"
    if [ "$c" = auto ]; then unset VRDX_TILE_CONFIG; else export VRDX_TILE_CONFIG=$c; fi
    if [ "${PROF:-0}" = 1 ]; then
      rm -rf /tmp/rv_prof
      (cd /tmp && LD_LIBRARY_PATH=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rv_prof -o s -- \
          $ROOT/tests/native/vrdx_selftest $ARGS 2>&1 | grep -v "^W2\|^E2\|rocprof" | tee -a "$OUT")
      f=$(find /tmp/rv_prof -name "*kernel_stats.csv" | head -1)
      python3 $ROOT/tools/kernel_stats.py "$f" | tee -a "$OUT"
    else
      LD_LIBRARY_PATH=$d timeout 300 $ROOT/tests/native/vrdx_selftest $ARGS 2>&1 | tee -a "$OUT"
    fi
  done
done

#!/bin/bash
# Hybrid plan of mid-size sorts: bucket headroom rules and the run-time slot counts of the bucket sort, ON the GPU box:
#   gpurun -- 'bash tools/hybrid_headroom.sh'      (build/variants/head = the previous commit's library, optional)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/hybrid_headroom.txt
T=$ROOT/tests/native/vrdx_selftest
: > $OUT
for hl in "200 200" "200 105"; do
  set -- $hl
  echo "== parity VRDX_HYBRID_HEADROOM=$1 VRDX_HYBRID_HEADROOM_LAST=$2" | tee -a $OUT
  VRDX_HYBRID_HEADROOM=$1 VRDX_HYBRID_HEADROOM_LAST=$2 timeout 600 $T parity 2>&1 | tail -2 | tee -a $OUT
done
for kind in keys kv; do
  if [ -f $ROOT/build/variants/head/libvrdx_hip.so ]; then
    echo "== sweep $kind head" | tee -a $OUT
    LD_LIBRARY_PATH=$ROOT/build/variants/head timeout 600 $T sweep 14.2 23.2 37 $kind 2>&1 | tee -a $OUT
  fi
  for hl in "200 200" "200 110" "150 110" "125 110" "110 105"; do
    set -- $hl
    echo "== sweep $kind H=$1 L=$2" | tee -a $OUT
    VRDX_HYBRID_HEADROOM=$1 VRDX_HYBRID_HEADROOM_LAST=$2 timeout 600 $T sweep 14.2 23.2 37 $kind 2>&1 | tee -a $OUT
  done
done

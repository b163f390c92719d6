#!/bin/bash
# Size sweep (the reference's points N = 2^18 k) of forced tile geometries with the tail / even split on and off,
# ON the GPU box:   gpurun -- 'bash tools/tail_split.sh keys|kv [from_k to_k step_k] [out-name]'
set -u
MODE=${1:-keys}
FROM=${2:-8}; TO=${3:-136}; STEP=${4:-2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${5:-tail_split_$MODE}.txt
mkdir -p "$ROOT/gpurun_out"
POINTS=$(( (TO - FROM) / STEP + 1 ))
LO=$(( FROM * 262144 )); HI=$(( (FROM + (POINTS - 1) * STEP) * 262144 ))
run() {  # label, env...
  local label=$1; shift
  echo "=== $label" | tee -a "$OUT"
  env "$@" $ROOT/tests/native/vrdx_selftest lsweep $LO $HI $POINTS $MODE 2>&1 | tail -n +3 | tee -a "$OUT"
}
run "auto" VRDX_NOP=1
run "auto, no tail split" VRDX_TAIL_SPLIT=0
run "auto, no hybrid" VRDX_HYBRID=0
run "1024x32" VRDX_TILE_CONFIG=1024x32
run "1024x32, no tail split" VRDX_TILE_CONFIG=1024x32 VRDX_TAIL_SPLIT=0
run "1024x32, no split at all" VRDX_TILE_CONFIG=1024x32 VRDX_TAIL_SPLIT=0 VRDX_EVEN_SPLIT=0
run "1024x16" VRDX_TILE_CONFIG=1024x16
if [ "$MODE" = keys ]; then
  run "1024x32x2" VRDX_TILE_CONFIG=1024x32x2
  run "1024x32x2, no tail split" VRDX_TILE_CONFIG=1024x32x2 VRDX_TAIL_SPLIT=0
fi

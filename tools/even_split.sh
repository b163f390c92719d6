#!/bin/bash
# Even-split tiles (OnesweepArgs::slots) against tiles of full capacity, by forced geometry, ON the GPU box:
#   gpurun -- 'bash tools/even_split.sh'
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/even_split.txt
T=$ROOT/tests/native/vrdx_selftest
: > $OUT
for c in auto 1024x32 1024x32x2; do
  if [ "$c" = auto ]; then unset VRDX_TILE_CONFIG; else export VRDX_TILE_CONFIG=$c; fi
  echo "== parity config=$c" | tee -a $OUT
  timeout 600 $T parity 2>&1 | tail -3 | tee -a $OUT
done
for c in 1024x16 1024x32 1024x32x2 auto; do
  if [ "$c" = auto ]; then unset VRDX_TILE_CONFIG; else export VRDX_TILE_CONFIG=$c; fi
  for e in 0 1; do
    echo "== sweep keys config=$c VRDX_EVEN_SPLIT=$e" | tee -a $OUT
    VRDX_EVEN_SPLIT=$e timeout 600 $T sweep 21.6 24.2 27 keys 2>&1 | tee -a $OUT
  done
done

#!/bin/bash
# Timing comparison of compile-time variants of the kernels, run ON the GPU box (hipcc is in the image):
#   gpurun -- 'bash tools/variants.sh "base: w8:-DVRDX_LOOKBACK_WINDOW=8" "512x32 1024x16" 25'
# Each variant "name:flags" is built into /tmp and benchmarked with the native selftest.
set -u
VARIANTS=${1:-"base:"}
CONFIGS=${2:-"512x32"}
LOGS=${3:-"25"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/variants.log
mkdir -p "$ROOT/gpurun_out"
: > "$OUT"
for v in $VARIANTS; do
  name=${v%%:*}
  flags=${v#*:}
  flags=${flags//,/ }
  d=/tmp/vrdx_variant_$name
  mkdir -p $d
  (cd $ROOT/vulkan_radix_sort_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
      $flags -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o $d/libvrdx_hip.so) || exit 1
  for c in $CONFIGS; do
    echo "=== variant=$name ($flags) config=$c" | tee -a "$OUT"
    if [ "$c" = auto ]; then unset VRDX_TILE_CONFIG; else export VRDX_TILE_CONFIG=$c; fi  # auto: the size-adaptive choice
    LD_LIBRARY_PATH=$d timeout 120 $ROOT/tests/native/vrdx_selftest bench $LOGS 2>&1 | tail -n +3 | tee -a "$OUT"
  done
done

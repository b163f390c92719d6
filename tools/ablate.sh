#!/bin/bash
# Timing-only ablations of the onesweep kernel, run ON the GPU box (hipcc is in the image):
#   gpurun -- 'bash tools/ablate.sh "0 1 2 4 8 16 31" "512x16 1024x8" 25'
# Builds one libvrdx_hip.so per VRDX_ABLATE mask into /tmp and runs the native bench against it.
# Masks != 0 produce WRONG sort results by construction; only the timings mean anything.
set -u
MASKS=${1:-"0 1 2 4 8 16"}
CONFIGS=${2:-"512x16"}
LOGS=${3:-"25"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/ablate.log
mkdir -p "$ROOT/gpurun_out"
: > "$OUT"
for m in $MASKS; do
  d=/tmp/vrdx_ablate_$m
  mkdir -p $d
  (cd $ROOT/vulkan_radix_sort_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
      -DVRDX_ABLATE=$m -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o $d/libvrdx_hip.so) || exit 1
  for c in $CONFIGS; do
    echo "=== ablate=$m config=$c" | tee -a "$OUT"
    LD_LIBRARY_PATH=$d VRDX_TILE_CONFIG=$c timeout 120 $ROOT/tests/native/vrdx_selftest bench $LOGS 2>&1 | tail -n +3 | tee -a "$OUT"
  done
done

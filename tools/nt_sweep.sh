#!/bin/bash
# Non-temporal tile loads (VRDX_STREAMING_LOADS: 0 never, 1 by size = the product, 2 always), native size sweep,
# run ON the GPU box:   gpurun -- 'bash tools/nt_sweep.sh "0 1 2" 23.5 26.5 25 kv'
MODES=${1:-"0 1"}; LO=${2:-23.5}; HI=${3:-26.5}; POINTS=${4:-25}; KIND=${5:-kv}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for m in $MODES; do
  d=/tmp/vrdx_streaming_$m; mkdir -p $d
  (cd $ROOT/vulkan_radix_sort_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVRDX_STREAMING_LOADS=$m \
      -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o $d/libvrdx_hip.so) || exit 1
  echo "=== VRDX_STREAMING_LOADS=$m $KIND sweep"
  LD_LIBRARY_PATH=$d timeout 300 $ROOT/tests/native/vrdx_selftest sweep $LO $HI $POINTS $KIND 2>&1 | tail -n +2
done

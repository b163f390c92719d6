#!/bin/bash
# Cross-compiles compile-time variants of the library HERE (no GPU needed) into build/variants/<name>/,
# which travel to the GPU box with the snapshot:   bash tools/build_variants.sh "base: w16:-DVRDX_LOOKBACK_WINDOW=16"
# Run them there with tools/run_variants.sh.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
pids=()
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  d=$ROOT/build/variants/$name; mkdir -p $d
  echo "$flags" > $d/flags.txt
  (cd $ROOT/vulkan_radix_sort_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
      $flags -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o $d/libvrdx_hip.so 2> $d/build.log || echo "BUILD FAILED: $name") &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls -la $ROOT/build/variants/*/libvrdx_hip.so

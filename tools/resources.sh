#!/bin/bash
# Compact register / scratch / LDS report of every kernel (cross-compiles, no GPU):  bash tools/resources.sh [extra hipcc flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/vulkan_radix_sort_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -Rpass-analysis=kernel-resource-usage -x hip -c vrdx_kernels.hip -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /LDS Size/ {print v, s, $(NF-1), name}' |
  while read v s l name; do printf "%4s VGPR %4s scratch %7s LDS  %s\n" "$v" "$s" "$l" "$(echo $name | c++filt | sed 's/void vrdx:://; s/(.*//')"; done

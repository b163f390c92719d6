#!/bin/bash
# Histogram kernel duration for compile-time variants, run ON the GPU box:
#   gpurun -- 'bash tools/hist_variants.sh "base: c16:-DVRDX_HIST_COPIES_LARGE=16" "25 23"'
VARIANTS=${1:-"base:"}
LOGS=${2:-"25"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
for v in $VARIANTS; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  d=/tmp/vrdx_hv_$name; mkdir -p $d
  (cd $ROOT/vulkan_radix_sort_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
      $flags -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o $d/libvrdx_hip.so) || exit 1
  for lg in $LOGS; do
    rm -rf /tmp/hv_prof
    (cd /tmp && LD_LIBRARY_PATH=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hv_prof -o s -- \
        $ROOT/tests/native/vrdx_selftest trace $lg keys > /dev/null 2>&1)
    f=$(find /tmp/hv_prof -name "*kernel_stats.csv" | head -1)
    echo "variant=$name n=2^$lg $(grep histogram $f | awk -F, '{print "hist avg ns", $(NF-4), "min", $(NF-2)}')"
  done
done

#!/bin/bash
# Phase-level timeline of the onesweep kernel, run ON the GPU box:
#   gpurun -- 'bash tools/trace.sh 512x32 25 [keys|kv] [uniform|ascending|equal]'
# Builds a -DVRDX_TRACE library into /tmp, sorts once (native selftest "trace" mode), and prints
# per-phase statistics from the stamps.  Timing diagnostics only.
set -u
CONFIG=${1:-512x16}
LOG2N=${2:-25}
KV=${3:-keys}
PATTERN=${4:-uniform}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/vrdx_trace
mkdir -p $d "$ROOT/gpurun_out"
# A library cross-compiled beforehand (make -C vulkan_radix_sort_amd/csrc trace -> build/trace/) saves
# the compile time on the GPU box; it must be newer than the sources.
pre=$ROOT/build/trace/libvrdx_hip.so
if [ -z "${EXTRA_FLAGS:-}" ] && [ -f $pre ] && [ $pre -nt $ROOT/vulkan_radix_sort_amd/csrc/vrdx_kernels.hip ] \
    && [ $pre -nt $ROOT/vulkan_radix_sort_amd/csrc/vrdx_api.cpp ]; then
  cp $pre $d/libvrdx_hip.so
else
  (cd $ROOT/vulkan_radix_sort_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
      -DVRDX_TRACE ${EXTRA_FLAGS:-} -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o $d/libvrdx_hip.so) || exit 1
fi
LD_LIBRARY_PATH=$d VRDX_TILE_CONFIG=$CONFIG VRDX_TRACE_FILE=$d/trace.bin timeout 120 \
    $ROOT/tests/native/vrdx_selftest trace $LOG2N $KV $PATTERN || exit 1
{
  case "$CONFIG" in
    *x2) echo "# two-sub-tile kernel: the phase labels below read  ticket | load A + rank A | scan A + regroup A + rank B | scan B | look-back | scatter A + regroup B + scatter B" ;;
  esac
  python3 $ROOT/tools/trace_report.py $d/trace.bin
} | tee "$ROOT/gpurun_out/trace_${CONFIG}_${KV}_${PATTERN}.txt"

#!/bin/bash
# What does a returning fallback launch cost by grid size?  Scratch builds with the passes behind the MSD plan launched with
# 256 / 64 / 8 workgroups instead of one per tile (results of DECLINED inputs would be wrong; uniform keys never get there).
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/r05_return_grid; mkdir -p $OUT; export TMPDIR=/tmp
T=$ROOT/tests/native/vrdx_selftest
for g in tiles 256 64 8; do
  rm -rf /tmp/rg_$g; mkdir -p /tmp/rg_$g/x/csrc; cp $ROOT/vulkan_radix_sort_amd/csrc/* /tmp/rg_$g/x/csrc/; cp -r $ROOT/include /tmp/rg_$g/include
  if [ $g != tiles ]; then sed -i "s/LaunchOnesweep(stream, configIndex, tiles, keyValue/LaunchOnesweep(stream, configIndex, (msdBits != 0 ? ${g}u : tiles), keyValue/" /tmp/rg_$g/x/csrc/vrdx_api.cpp; fi
  (cd /tmp/rg_$g/x/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o /tmp/rg_$g/libvrdx_hip.so 2> $OUT/build_$g.err) &
done
wait
for g in tiles 256 64 8; do
  echo "== fallback launches with grid $g"
  LD_LIBRARY_PATH=/tmp/rg_$g timeout 300 $T bench 16777216 33554432 2>&1 | grep -E "^[0-9]"
done | tee $OUT/return_grid.txt

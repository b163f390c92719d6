#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/r05_out_nt; mkdir -p $OUT
cd $ROOT/vulkan_radix_sort_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVRDX_MSD_OUT_NT=1 -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o /tmp/libvrdx_outnt.so 2> $OUT/build.err
cd $ROOT
for rep in 1 2 3; do
for lib in "" /tmp/libvrdx_outnt.so; do
  VRDX_LIBRARY=$lib python3 bench.py --no-sweep --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib'.ljust(24), 'keys', round(d['value'],1), 'kv', round(d['key_value']['value'],1), 'median keys', round(d['median_gitems_per_s'],1), 'kv', round(d['key_value']['median_gitems_per_s'],1))"
done; done
mkdir -p /tmp/nt; cp /tmp/libvrdx_outnt.so /tmp/nt/libvrdx_hip.so
echo "--- selftest bench (reference protocol), plain then nt"
tests/native/vrdx_selftest bench 24 25 | grep -E "^[0-9]"
LD_LIBRARY_PATH=/tmp/nt tests/native/vrdx_selftest bench 24 25 | grep -E "^[0-9]"

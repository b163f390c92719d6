#!/bin/bash
# bench.py's back-to-back loop with the bucket kernel's output stores plain / non-temporal (variant library on the box)
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/r05_out_nt; mkdir -p $OUT
cd $ROOT/vulkan_radix_sort_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVRDX_MSD_OUT_NT=1 -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o /tmp/libvrdx_outnt.so 2> $OUT/build.err
cd $ROOT
for lib in "" /tmp/libvrdx_outnt.so; do
  echo "=== VRDX_LIBRARY=$lib"
  VRDX_LIBRARY=$lib python3 bench.py --no-sweep --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('keys', round(d['value'],1), 'kv', round(d['key_value']['value'],1), 'median keys', round(d['median_gitems_per_s'],1), 'kv', round(d['key_value']['median_gitems_per_s'],1))
for m,ks in (('keys',d['roofline']['kernels']),('kv',d['roofline']['key_value']['kernels'])):
    print(m, {k: round(v['avg_launch_ms']*1e3,1) for k,v in ks.items()})"
done

#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT/vulkan_radix_sort_amd/csrc
mkdir -p /tmp/nodyn
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVRDX_MSD_SCATTER_DYN=0 -x hip vrdx_kernels.hip vrdx_api.cpp -shared -o /tmp/nodyn/libvrdx_hip.so 2>/dev/null
cd $ROOT
for rep in 1 2; do
echo "--- product (DYN, even tiles)"; tests/native/vrdx_selftest bench 25 | grep -E "^[0-9]"
echo "--- product, VRDX_MSD_EVEN=0"; VRDX_MSD_EVEN=0 tests/native/vrdx_selftest bench 25 | grep -E "^[0-9]"
echo "--- non-DYN scatter, VRDX_MSD_EVEN=0"; LD_LIBRARY_PATH=/tmp/nodyn VRDX_MSD_EVEN=0 tests/native/vrdx_selftest bench 25 | grep -E "^[0-9]"
done

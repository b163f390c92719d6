// PMC calibration + sort workload for the HBM-traffic measurement (run under rocprofv3 --pmc):
//   1. copy_dword : N u32 read + N u32 written with 4 B/lane accesses   (known bytes: 4N + 4N)
//   2. copy_uint4 : the same bytes with 16 B/lane accesses
//   3. one keys-only sort and one key+value sort of N = 2^25 through the C-ABI
// FETCH_SIZE / WRITE_SIZE of the copy kernels give the counter -> bytes factors for each access
// width (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide reads by 2x on gfx950, other widths are
// uncalibrated), which tools/pmc_report.py applies to the sort kernels' counters.
// Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 pmc_calibrate.hip -I../../include -L../../vulkan_radix_sort_amd -lvrdx_hip \
//         -Wl,-rpath,$PWD/../../vulkan_radix_sort_amd -o /tmp/pmc_calibrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>
#include "vk_radix_sort.h"

__global__ void copy_dword(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy_uint4(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t n4) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) out[i] = in[i];
}

int main(int argc, char** argv) {
  const uint32_t n = 1u << (argc > 1 ? atoi(argv[1]) : 25);
  std::vector<uint32_t> h(2 * (size_t)n);
  std::mt19937 g(1);
  for (auto& x : h) x = g();
  uint32_t *a, *b, *storage;
  hipMalloc(&a, 2 * (size_t)n * 4);
  hipMalloc(&b, 2 * (size_t)n * 4);
  VrdxSorter sorter;
  VrdxSorterCreateInfo info = {};
  if (vrdxCreateSorter(&info, &sorter) != VK_SUCCESS) return 2;
  VrdxSorterStorageRequirements req;
  vrdxGetSorterKeyValueStorageRequirements(sorter, n, &req);
  hipMalloc(&storage, req.size);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(a, h.data(), 2 * (size_t)n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(copy_dword, dim3(2048), dim3(512), 0, 0, a, b, n);
    hipLaunchKernelGGL(copy_uint4, dim3(2048), dim3(512), 0, 0, (const uint4*)a, (uint4*)b, n / 4);
    hipDeviceSynchronize();
    vrdxCmdSort(nullptr, sorter, n, (VkBuffer)a, 0, (VkBuffer)storage, 0, nullptr, 0);
    hipDeviceSynchronize();
    hipMemcpy(a, h.data(), 2 * (size_t)n * 4, hipMemcpyHostToDevice);
    vrdxCmdSortKeyValue(nullptr, sorter, n, (VkBuffer)a, 0, (VkBuffer)a, (VkDeviceSize)n * 4, (VkBuffer)storage, 0, nullptr, 0);
    hipDeviceSynchronize();
  }
  printf("pmc_calibrate done n=%u\n", n);
  return 0;
}

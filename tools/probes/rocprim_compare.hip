// Comparator only (the analogue of the reference's bench/cuda_benchmark.cu
// which times cub::DeviceRadixSort): rocPRIM's device radix sort on the same box, same protocol
// (uniform mt19937 u32, 1 warm-up + 10 timed runs on fresh data, median, hipEvents around the call).
// Never linked into the product.  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 rocprim_compare.hip -o /tmp/rocprim_compare
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/rocprim_version.hpp>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  printf("%-10s %-6s %10s %12s   (rocPRIM %d.%d.%d radix_sort_keys / radix_sort_pairs, out-of-place)\n", "n", "sort", "gpu_ms",
         "GItems/s", ROCPRIM_VERSION_MAJOR, ROCPRIM_VERSION_MINOR, ROCPRIM_VERSION_PATCH);
  for (int a = 1; a < (argc > 1 ? argc : 2); ++a) {
    const int lg = argc > 1 ? atoi(argv[a]) : 25;
    const size_t n = (size_t)1 << lg;
    uint32_t *kin, *kout, *vin, *vout;
    OK(hipMalloc(&kin, n * 4)); OK(hipMalloc(&kout, n * 4)); OK(hipMalloc(&vin, n * 4)); OK(hipMalloc(&vout, n * 4));
    for (int kv = 0; kv < 2; ++kv) {
      size_t tempBytes = 0;
      if (kv) OK(rocprim::radix_sort_pairs(nullptr, tempBytes, kin, kout, vin, vout, n, 0, 32));
      else OK(rocprim::radix_sort_keys(nullptr, tempBytes, kin, kout, n, 0, 32));
      void* temp; OK(hipMalloc(&temp, tempBytes));
      hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
      std::vector<float> times;
      for (int run = 0; run < 11; ++run) {
        std::mt19937 g(run + 1);
        std::vector<uint32_t> h(n), hv(n);
        for (auto& x : h) x = g();
        for (auto& x : hv) x = g();
        OK(hipMemcpy(kin, h.data(), n * 4, hipMemcpyHostToDevice));
        OK(hipMemcpy(vin, hv.data(), n * 4, hipMemcpyHostToDevice));
        OK(hipDeviceSynchronize());
        OK(hipEventRecord(e0, 0));
        if (kv) OK(rocprim::radix_sort_pairs(temp, tempBytes, kin, kout, vin, vout, n, 0, 32, 0));
        else OK(rocprim::radix_sort_keys(temp, tempBytes, kin, kout, n, 0, 32, 0));
        OK(hipEventRecord(e1, 0));
        OK(hipDeviceSynchronize());
        float ms; OK(hipEventElapsedTime(&ms, e0, e1));
        if (run > 0) times.push_back(ms);
        if (run == 1) {  // sanity: sorted
          std::vector<uint32_t> out(n);
          OK(hipMemcpy(out.data(), kout, n * 4, hipMemcpyDeviceToHost));
          if (!std::is_sorted(out.begin(), out.end())) { printf("rocprim output not sorted?!\n"); return 1; }
        }
      }
      std::nth_element(times.begin(), times.begin() + times.size() / 2, times.end());
      const float ms = times[times.size() / 2];
      printf("%-10zu %-6s %10.4f %12.3f   temp %zu bytes\n", n, kv ? "kv" : "keys", ms, n / (ms * 1e-3) / 1e9, tempBytes);
      fflush(stdout);
      OK(hipFree(temp));
    }
    OK(hipFree(kin)); OK(hipFree(kout)); OK(hipFree(vin)); OK(hipFree(vout));
  }
  return 0;
}

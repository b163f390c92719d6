// rocPRIM comparator backend of the bench driver -- the analogue of the reference's CUB backend
// (/root/reference/bench/cuda_benchmark.cu:37-126: out-of-place DeviceRadixSort over bits 0..32 timed
// by events).  Comparator only: nothing in vulkan_radix_sort_amd/ uses rocPRIM.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/rocprim_version.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "backends.h"

namespace {

#define ROC_OK(x)                                                                               \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      std::exit(2);                                                                             \
    }                                                                                           \
  } while (0)

class RocprimBenchmark : public BenchmarkBase {
 public:
  RocprimBenchmark() {
    ROC_OK(hipEventCreate(&start_));
    ROC_OK(hipEventCreate(&end_));
  }
  ~RocprimBenchmark() override {
    for (void* p : {(void*)kin_, (void*)kout_, (void*)vin_, (void*)vout_, temp_})
      if (p) (void)hipFree(p);
  }
  std::string LibraryVersion() const override {
    return "rocPRIM " + std::to_string(ROCPRIM_VERSION_MAJOR) + "." + std::to_string(ROCPRIM_VERSION_MINOR) + "." +
           std::to_string(ROCPRIM_VERSION_PATCH);
  }
  Results Sort(const std::vector<uint32_t>& keys) override { return Run(keys, nullptr); }
  Results SortKeyValue(const std::vector<uint32_t>& keys, const std::vector<uint32_t>& values) override {
    return Run(keys, &values);
  }

 private:
  Results Run(const std::vector<uint32_t>& keys, const std::vector<uint32_t>* values) {
    const size_t n = keys.size();
    if (n > cap_) {
      for (uint32_t** p : {&kin_, &kout_, &vin_, &vout_}) {
        if (*p) ROC_OK(hipFree(*p));
        ROC_OK(hipMalloc(reinterpret_cast<void**>(p), n * 4));
      }
      cap_ = n;
    }
    size_t bytes = 0;
    if (values)
      ROC_OK(rocprim::radix_sort_pairs(nullptr, bytes, kin_, kout_, vin_, vout_, n, 0, 32));
    else
      ROC_OK(rocprim::radix_sort_keys(nullptr, bytes, kin_, kout_, n, 0, 32));
    if (bytes > tempCap_) {
      if (temp_) ROC_OK(hipFree(temp_));
      ROC_OK(hipMalloc(&temp_, bytes));
      tempCap_ = bytes;
    }
    ROC_OK(hipMemcpy(kin_, keys.data(), n * 4, hipMemcpyHostToDevice));
    if (values) ROC_OK(hipMemcpy(vin_, values->data(), n * 4, hipMemcpyHostToDevice));
    ROC_OK(hipDeviceSynchronize());
    const auto c0 = std::chrono::steady_clock::now();
    ROC_OK(hipEventRecord(start_, 0));
    if (values)
      ROC_OK(rocprim::radix_sort_pairs(temp_, bytes, kin_, kout_, vin_, vout_, n, 0, 32, 0));
    else
      ROC_OK(rocprim::radix_sort_keys(temp_, bytes, kin_, kout_, n, 0, 32, 0));
    ROC_OK(hipEventRecord(end_, 0));
    ROC_OK(hipDeviceSynchronize());
    const auto c1 = std::chrono::steady_clock::now();
    float ms = 0;
    ROC_OK(hipEventElapsedTime(&ms, start_, end_));
    Results r;
    r.keys.resize(n);
    ROC_OK(hipMemcpy(r.keys.data(), kout_, n * 4, hipMemcpyDeviceToHost));
    if (values) {
      r.values.resize(n);
      ROC_OK(hipMemcpy(r.values.data(), vout_, n * 4, hipMemcpyDeviceToHost));
    }
    r.total_time = static_cast<uint64_t>(double(ms) * 1e6);
    r.cpu_time = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(c1 - c0).count());
    return r;
  }

  hipEvent_t start_ = nullptr, end_ = nullptr;
  uint32_t *kin_ = nullptr, *kout_ = nullptr, *vin_ = nullptr, *vout_ = nullptr;
  void* temp_ = nullptr;
  size_t cap_ = 0, tempCap_ = 0;
};

}  // namespace

std::unique_ptr<BenchmarkBase> CreateRocprimBenchmark() { return std::make_unique<RocprimBenchmark>(); }

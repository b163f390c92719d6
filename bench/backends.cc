// hip and cpu backends of the bench driver.
//   cpu: restates /root/reference/bench/cpu_benchmark.cc:19-53 (std::sort on a copy; std::stable_sort
//        of an index vector by key, then gather; only the sort call is timed).
//   hip: the in-repo analogue of bench/vulkan_benchmark.cc:253-339 (Sort) and :341-433 (SortKeyValue):
//        upload, vrdxCmdSort / vrdxCmdSortKeyValueIndirect with keys, values and the count in ONE
//        buffer, wall time around submit -> completion, read-back, 15 timestamps.
#include "backends.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "../include/vk_radix_sort.h"

namespace {

int64_t Now() { return std::chrono::high_resolution_clock::now().time_since_epoch().count(); }

#define BENCH_HIP_OK(x)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      std::exit(2);                                                                             \
    }                                                                                           \
  } while (0)

class CpuBenchmark : public BenchmarkBase {
 public:
  Results Sort(const std::vector<uint32_t>& keys) override {
    Results r;
    r.keys = keys;
    const auto start = Now();
    std::sort(r.keys.begin(), r.keys.end());
    const auto end = Now();
    r.total_time = r.cpu_time = static_cast<uint64_t>(end - start);
    return r;
  }
  Results SortKeyValue(const std::vector<uint32_t>& keys, const std::vector<uint32_t>& values) override {
    std::vector<uint32_t> indices(keys.size());
    std::iota(indices.begin(), indices.end(), 0u);
    const auto start = Now();
    std::stable_sort(indices.begin(), indices.end(), [&](uint32_t l, uint32_t r) { return keys[l] < keys[r]; });
    const auto end = Now();
    Results r;
    r.keys.reserve(keys.size());
    r.values.reserve(keys.size());
    for (uint32_t i : indices) {
      r.keys.push_back(keys[i]);
      r.values.push_back(values[i]);
    }
    r.total_time = r.cpu_time = static_cast<uint64_t>(end - start);
    return r;
  }
};

class HipBenchmark : public BenchmarkBase {
 public:
  explicit HipBenchmark(bool graph) : graph_(graph) {
    VrdxSorterCreateInfo info = {};
    const VkResult r = vrdxCreateSorter(&info, &sorter_);
    if (r != VK_SUCCESS) {
      std::fprintf(stderr, "vrdxCreateSorter failed: %d\n", static_cast<int>(r));
      std::exit(2);
    }
    BENCH_HIP_OK(hipStreamCreate(&stream_));
    if (vrdxHipCreateQueryPool(15, &pool_) != VK_SUCCESS) std::exit(2);
    BENCH_HIP_OK(hipEventCreate(&start_));
    BENCH_HIP_OK(hipEventCreate(&end_));
    // VRDX_BENCH_RESERVE=<elements>: allocate for that many pairs up front instead of growing with the sweep (the reference
    // grows, bench/vulkan_benchmark.cc:225-250, but from VMA's 256 MiB blocks; hipMalloc hands out fresh pages at every
    // size, and what mapping a size gets shows in its ten sorts -- tools/r05/jitter.sh).
    if (const char* reserve = std::getenv("VRDX_BENCH_RESERVE")) {
      const uint32_t n = static_cast<uint32_t>(std::strtoul(reserve, nullptr, 10));
      VrdxSorterStorageRequirements req;
      vrdxGetSorterKeyValueStorageRequirements(sorter_, n, &req);
      Reserve(size_t(2) * Align(n * 4u) + 16, req.size);
    }
  }
  ~HipBenchmark() override {
    (void)hipStreamSynchronize(stream_);
    DropGraph();
    if (keys_) (void)hipFree(keys_);
    if (storage_) (void)hipFree(storage_);
    vrdxHipDestroyQueryPool(pool_);
    vrdxDestroySorter(sorter_);
    (void)hipEventDestroy(start_);
    (void)hipEventDestroy(end_);
    (void)hipStreamDestroy(stream_);
  }
  std::string LibraryVersion() const override {
    return std::string(vrdxHipVersionString()) + (graph_ ? " [hipGraph replay]" : "");
  }
  bool Healthy() override { return vrdxHipReadSorterStatus(sorter_, (VkCommandBuffer)stream_) == 0; }

  Results Sort(const std::vector<uint32_t>& keys) override {
    const uint32_t n = static_cast<uint32_t>(keys.size());
    const uint32_t inout = Align(n * 4u);
    VrdxSorterStorageRequirements req;
    vrdxGetSorterStorageRequirements(sorter_, n, &req);
    Reserve(inout, req.size);
    BENCH_HIP_OK(hipMemcpy(keys_, keys.data(), size_t(n) * 4, hipMemcpyHostToDevice));
    return Run(n, inout, false, [&](VkQueryPool pool) {
      vrdxCmdSort((VkCommandBuffer)stream_, sorter_, n, (VkBuffer)keys_, 0, (VkBuffer)storage_, 0, pool, 0);
    });
  }

  Results SortKeyValue(const std::vector<uint32_t>& keys, const std::vector<uint32_t>& values) override {
    const uint32_t n = static_cast<uint32_t>(keys.size());
    const uint32_t inout = Align(n * 4u);
    VrdxSorterStorageRequirements req;
    vrdxGetSorterKeyValueStorageRequirements(sorter_, n, &req);
    Reserve(size_t(2) * inout + 16, req.size);
    // keys | values | count in one buffer, like bench/vulkan_benchmark.cc:356-358
    BENCH_HIP_OK(hipMemcpy(keys_, keys.data(), size_t(n) * 4, hipMemcpyHostToDevice));
    BENCH_HIP_OK(hipMemcpy(keys_ + inout, values.data(), size_t(n) * 4, hipMemcpyHostToDevice));
    BENCH_HIP_OK(hipMemcpy(keys_ + size_t(2) * inout, &n, 4, hipMemcpyHostToDevice));
    return Run(n, inout, true, [&](VkQueryPool pool) {
      vrdxCmdSortKeyValueIndirect((VkCommandBuffer)stream_, sorter_, n, (VkBuffer)keys_, size_t(2) * inout, (VkBuffer)keys_, 0,
                                  (VkBuffer)keys_, inout, (VkBuffer)storage_, 0, pool, 0);
    });
  }

 private:
  static uint32_t Align(uint32_t x) { return (x + 15u) / 16u * 16u; }

  void Reserve(size_t keysBytes, size_t storageBytes) {
    if (keysBytes > keysCap_ || storageBytes > storageCap_) DropGraph();  // the captured sort holds the old addresses
    if (keysBytes > keysCap_) {
      if (keys_) BENCH_HIP_OK(hipFree(keys_));
      BENCH_HIP_OK(hipMalloc(reinterpret_cast<void**>(&keys_), keysBytes));
      keysCap_ = keysBytes;
    }
    if (storageBytes > storageCap_) {
      if (storage_) BENCH_HIP_OK(hipFree(storage_));
      BENCH_HIP_OK(hipMalloc(reinterpret_cast<void**>(&storage_), storageBytes));
      storageCap_ = storageBytes;
    }
  }

  // The timed sort carries no query pool: total_time is two events around the enqueue (the 15
  // hipEventRecords of the timestamp contract cost ~50 us per sort).  The per-stage split comes
  // from a second, untimed sort (of the already sorted buffer) with the timestamps active.
  void DropGraph() {
    if (exec_) (void)hipGraphExecDestroy(exec_);
    exec_ = nullptr;
    graphN_ = 0;
  }

  template <typename Record>
  Results Run(uint32_t n, uint32_t inout, bool keyValue, Record record) {
    if (graph_ && (exec_ == nullptr || graphN_ != n || graphKeyValue_ != keyValue)) {
      // record once: vrdxCmdSort* into a capturing stream == gpuSort() into a command buffer
      DropGraph();
      hipGraph_t graph = nullptr;
      BENCH_HIP_OK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
      record(VK_NULL_HANDLE);
      BENCH_HIP_OK(hipStreamEndCapture(stream_, &graph));
      BENCH_HIP_OK(hipGraphInstantiate(&exec_, graph, nullptr, nullptr, 0));
      BENCH_HIP_OK(hipGraphDestroy(graph));
      graphN_ = n;
      graphKeyValue_ = keyValue;
    }
    BENCH_HIP_OK(hipDeviceSynchronize());
    const auto cpuStart = std::chrono::steady_clock::now();
    BENCH_HIP_OK(hipEventRecord(start_, stream_));
    if (graph_)
      BENCH_HIP_OK(hipGraphLaunch(exec_, stream_));  // submit many
    else
      record(VK_NULL_HANDLE);
    BENCH_HIP_OK(hipEventRecord(end_, stream_));
    BENCH_HIP_OK(hipStreamSynchronize(stream_));
    const auto cpuEnd = std::chrono::steady_clock::now();
    float ms = 0;
    BENCH_HIP_OK(hipEventElapsedTime(&ms, start_, end_));

    Results r;
    r.keys.resize(n);
    BENCH_HIP_OK(hipMemcpy(r.keys.data(), keys_, size_t(n) * 4, hipMemcpyDeviceToHost));
    if (keyValue) {
      r.values.resize(n);
      BENCH_HIP_OK(hipMemcpy(r.values.data(), keys_ + inout, size_t(n) * 4, hipMemcpyDeviceToHost));
    }
    r.total_time = static_cast<uint64_t>(double(ms) * 1e6);
    r.cpu_time = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(cpuEnd - cpuStart).count());
    {
      // what the timed sort moved: the plan the host recorded if the device took it, the four passes if it did not
      VrdxHipPlanInfo plan;
      vrdxHipDescribePlan(sorter_, n, keyValue ? 1 : 0, &plan);
      const uint32_t verdict = vrdxHipReadPlanVerdict((VkCommandBuffer)stream_, (VkBuffer)storage_, 0);
      const bool decidedOnDevice = plan.plan == VRDX_HIP_PLAN_MSD || plan.plan == VRDX_HIP_PLAN_HYBRID8;
      const bool taken = verdict == VRDX_HIP_VERDICT_MSD_RUNS || verdict == VRDX_HIP_VERDICT_MSD_SORTED ||
                         verdict == VRDX_HIP_VERDICT_HYBRID8_RUNS;
      r.bytes_per_element = decidedOnDevice && !taken ? plan.fallbackBytesPerElement : plan.bytesPerElement;
    }

    // stage split (bench/vulkan_benchmark.cc:330-337) from one more sort of the now sorted data
    // with the 15-slot timestamp contract active; not part of total_time
    record(pool_);
    BENCH_HIP_OK(hipStreamSynchronize(stream_));
    uint64_t ts[15];
    if (vrdxHipGetQueryPoolResults(pool_, 0, 15, ts) == VK_SUCCESS) {
      for (int pass = 0; pass < 4; ++pass) {
        r.upsweep_ns += ts[2 + 3 * pass] - ts[1 + 3 * pass];
        r.spine_ns += ts[3 + 3 * pass] - ts[2 + 3 * pass];
        r.downsweep_ns += ts[4 + 3 * pass] - ts[3 + 3 * pass];
      }
    }
    return r;
  }

  VrdxSorter sorter_ = nullptr;
  hipStream_t stream_ = nullptr;
  VkQueryPool pool_ = nullptr;
  hipEvent_t start_ = nullptr, end_ = nullptr;
  uint8_t* keys_ = nullptr;
  uint8_t* storage_ = nullptr;
  size_t keysCap_ = 0, storageCap_ = 0;
  const bool graph_;
  hipGraphExec_t exec_ = nullptr;  // the captured sort of (graphN_, graphKeyValue_) on keys_ / storage_
  uint32_t graphN_ = 0;
  bool graphKeyValue_ = false;
};

}  // namespace

std::unique_ptr<BenchmarkBase> CreateBenchmark(const std::string& type, bool graph) {
  if (type == "cpu") return std::make_unique<CpuBenchmark>();
  if (type == "hip") return std::make_unique<HipBenchmark>(graph);
  if (type == "rocprim") return CreateRocprimBenchmark();
  return nullptr;
}

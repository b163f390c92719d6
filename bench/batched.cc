// bench hip --devices G: the batched many-arrays variant of BASELINE.json (configs[4]) from ONE host
// process in C++ -- the reference's host side is C++, and it has a single device and a single queue
// (/root/reference/bench/vulkan_benchmark.cc:103,128-135); this extends that driver to G devices.
//
// A independent key+value arrays of 2^L elements; array i lives on GPU i mod G.  Per GPU: one
// VrdxSorter, one stream, one storage buffer.  All sorts are enqueued from this
// one thread (vrdxCmdSort* never block), then every stream is synchronised.  No collective touches
// the data; the only communication is one ncclAllGather (RCCL, xGMI between the GPUs of a node) of a
// 16-byte {status, elapsed_ns} record per GPU, after which every GPU holds every GPU's record -- the
// batch is complete for everybody when that returns.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <unordered_map>
#include <vector>

#include "../include/vk_radix_sort.h"
#include "backends.h"

namespace {

#define BATCH_HIP_OK(x)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                                 \
    }                                                                                           \
  } while (0)
#define BATCH_NCCL_OK(x)                                                                          \
  do {                                                                                            \
    ncclResult_t r_ = (x);                                                                        \
    if (r_ != ncclSuccess) {                                                                      \
      std::fprintf(stderr, "RCCL error %s at %s:%d\n", ncclGetErrorString(r_), __FILE__, __LINE__); \
      return 2;                                                                                   \
    }                                                                                             \
  } while (0)

struct Device {
  VrdxSorter sorter = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t first = nullptr, last = nullptr;
  uint8_t* storage = nullptr;
  uint64_t* record = nullptr;    // {status, elapsed_ns} of this GPU
  uint64_t* gathered = nullptr;  // the records of all GPUs
  std::vector<int> arrays;       // indices of the arrays it owns
  std::vector<uint8_t*> buffers; // keys | values of each
};

}  // namespace

int RunBatched(int devices, int arrays, int log2n, bool verify) {
  int present = 0;
  BATCH_HIP_OK(hipGetDeviceCount(&present));
  if (devices > present) {
    std::fprintf(stderr, "--devices %d but this node exposes %d GPU(s)\n", devices, present);
    return 2;
  }
  const uint32_t n = 1u << log2n;
  const size_t inout = (size_t(n) * 4 + 15) / 16 * 16;
  std::vector<Device> gpu(static_cast<size_t>(devices));
  std::vector<std::vector<uint32_t>> keys(static_cast<size_t>(arrays)), values(static_cast<size_t>(arrays));

  for (int i = 0; i < arrays; ++i) {  // array i: the raw mt19937(i + 1) stream, keys then values (bench/data_generator.cc:12-26)
    DataGenerator gen(i + 1);
    SortData d = gen.Generate(n);
    keys[size_t(i)] = std::move(d.keys);
    values[size_t(i)] = std::move(d.values);
    gpu[size_t(i % devices)].arrays.push_back(i);
  }
  for (int g = 0; g < devices; ++g) {
    Device& dev = gpu[size_t(g)];
    BATCH_HIP_OK(hipSetDevice(g));
    VrdxSorterCreateInfo info = {};
    info.device = VRDX_HIP_DEVICE(g);
    if (vrdxCreateSorter(&info, &dev.sorter) != VK_SUCCESS) {
      std::fprintf(stderr, "vrdxCreateSorter failed on device %d\n", g);
      return 2;
    }
    BATCH_HIP_OK(hipStreamCreate(&dev.stream));
    BATCH_HIP_OK(hipEventCreate(&dev.first));
    BATCH_HIP_OK(hipEventCreate(&dev.last));
    VrdxSorterStorageRequirements req;
    vrdxGetSorterKeyValueStorageRequirements(dev.sorter, n, &req);
    BATCH_HIP_OK(hipMalloc(reinterpret_cast<void**>(&dev.storage), req.size));
    BATCH_HIP_OK(hipMalloc(reinterpret_cast<void**>(&dev.record), 2 * sizeof(uint64_t)));
    BATCH_HIP_OK(hipMalloc(reinterpret_cast<void**>(&dev.gathered), size_t(devices) * 2 * sizeof(uint64_t)));
    for (int i : dev.arrays) {
      uint8_t* b = nullptr;
      BATCH_HIP_OK(hipMalloc(reinterpret_cast<void**>(&b), 2 * inout));
      BATCH_HIP_OK(hipMemcpy(b, keys[size_t(i)].data(), size_t(n) * 4, hipMemcpyHostToDevice));
      BATCH_HIP_OK(hipMemcpy(b + inout, values[size_t(i)].data(), size_t(n) * 4, hipMemcpyHostToDevice));
      dev.buffers.push_back(b);
    }
    // warm-up: code objects loaded, clocks up (one sort of the first array's copy would disturb the data:
    // sort a scratch copy living in the storage-sized buffer instead -- simply sort array 0 and re-upload)
    if (!dev.buffers.empty()) {
      vrdxCmdSortKeyValue((VkCommandBuffer)dev.stream, dev.sorter, n, (VkBuffer)dev.buffers[0], 0, (VkBuffer)dev.buffers[0], inout,
                          (VkBuffer)dev.storage, 0, VK_NULL_HANDLE, 0);
      BATCH_HIP_OK(hipStreamSynchronize(dev.stream));
      const int i = dev.arrays[0];
      BATCH_HIP_OK(hipMemcpy(dev.buffers[0], keys[size_t(i)].data(), size_t(n) * 4, hipMemcpyHostToDevice));
      BATCH_HIP_OK(hipMemcpy(dev.buffers[0] + inout, values[size_t(i)].data(), size_t(n) * 4, hipMemcpyHostToDevice));
    }
  }

  std::vector<ncclComm_t> comms(static_cast<size_t>(devices));
  std::vector<int> ordinals(static_cast<size_t>(devices));
  std::iota(ordinals.begin(), ordinals.end(), 0);
  BATCH_NCCL_OK(ncclCommInitAll(comms.data(), devices, ordinals.data()));

  // ---- the batch: every sort of every GPU enqueued from this thread, then one wait per GPU ---------------
  const auto wallStart = std::chrono::steady_clock::now();
  for (int g = 0; g < devices; ++g) {
    BATCH_HIP_OK(hipSetDevice(g));
    BATCH_HIP_OK(hipEventRecord(gpu[size_t(g)].first, gpu[size_t(g)].stream));
  }
  size_t longest = 0;
  for (const Device& dev : gpu) longest = std::max(longest, dev.buffers.size());
  for (size_t slot = 0; slot < longest; ++slot)  // round-robin over the GPUs, so that all of them start at once
    for (int g = 0; g < devices; ++g) {
      Device& dev = gpu[size_t(g)];
      if (slot >= dev.buffers.size()) continue;
      vrdxCmdSortKeyValue((VkCommandBuffer)dev.stream, dev.sorter, n, (VkBuffer)dev.buffers[slot], 0, (VkBuffer)dev.buffers[slot],
                          inout, (VkBuffer)dev.storage, 0, VK_NULL_HANDLE, 0);
    }
  for (int g = 0; g < devices; ++g) {
    BATCH_HIP_OK(hipSetDevice(g));
    BATCH_HIP_OK(hipEventRecord(gpu[size_t(g)].last, gpu[size_t(g)].stream));
  }
  for (int g = 0; g < devices; ++g) BATCH_HIP_OK(hipStreamSynchronize(gpu[size_t(g)].stream));
  const auto wallEnd = std::chrono::steady_clock::now();

  // ---- the end-of-batch records, exchanged over RCCL --------------------------------------------------
  for (int g = 0; g < devices; ++g) {
    Device& dev = gpu[size_t(g)];
    BATCH_HIP_OK(hipSetDevice(g));
    float ms = 0;
    BATCH_HIP_OK(hipEventElapsedTime(&ms, dev.first, dev.last));
    const uint64_t record[2] = {vrdxHipReadSorterStatus(dev.sorter, (VkCommandBuffer)dev.stream),
                                static_cast<uint64_t>(double(ms) * 1e6)};
    BATCH_HIP_OK(hipMemcpy(dev.record, record, sizeof(record), hipMemcpyHostToDevice));
  }
  BATCH_NCCL_OK(ncclGroupStart());
  for (int g = 0; g < devices; ++g)
    BATCH_NCCL_OK(ncclAllGather(gpu[size_t(g)].record, gpu[size_t(g)].gathered, 2, ncclUint64, comms[size_t(g)], gpu[size_t(g)].stream));
  BATCH_NCCL_OK(ncclGroupEnd());
  for (int g = 0; g < devices; ++g) BATCH_HIP_OK(hipStreamSynchronize(gpu[size_t(g)].stream));
  std::vector<uint64_t> records(size_t(devices) * 2);
  BATCH_HIP_OK(hipSetDevice(devices - 1));  // any GPU holds all of them: read the last one's copy
  BATCH_HIP_OK(hipMemcpy(records.data(), gpu[size_t(devices - 1)].gathered, records.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));

  int failures = 0;
  uint64_t slowest = 0;
  std::printf("batched: %d key+value arrays of 2^%d elements over %d GPU(s), %s\n", arrays, log2n, devices, vrdxHipVersionString());
  for (int g = 0; g < devices; ++g) {
    const uint64_t status = records[size_t(g) * 2], ns = records[size_t(g) * 2 + 1];
    const size_t mine = gpu[size_t(g)].arrays.size();
    slowest = std::max(slowest, ns);
    if (status != 0) ++failures;
    std::printf("  gpu %d: %zu array(s), status %llu, %.3f ms, %.3f GItems/s\n", g, mine, (unsigned long long)status, ns / 1e6,
                ns ? double(mine) * n / double(ns) : 0.0);
  }
  const double wallNs = double(std::chrono::duration_cast<std::chrono::nanoseconds>(wallEnd - wallStart).count());
  std::printf("  aggregate: %.3f GItems/s by the slowest GPU's device time, %.3f GItems/s by host wall time (%.3f ms)\n",
              slowest ? double(arrays) * n / double(slowest) : 0.0, double(arrays) * n / wallNs, wallNs / 1e6);

  if (verify) {
    // Every array: ascending keys, an intact multiset of (key, value) PAIRS -- a per-pair hash, so that values swapped
    // between two pairs change the sum -- and stability: among equal keys the values keep their input order, which
    // is checked against the order in which the input holds that key's values.  Array 0 of every GPU in addition:
    // the reference's own predicate (bench/bench.cc:41-64), element by element against the cpu backend.
    auto pairHash = [](uint32_t key, uint32_t value) {
      uint64_t x = (uint64_t(key) << 32) | value;
      x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
      return x;
    };
    std::unique_ptr<BenchmarkBase> cpu = CreateBenchmark("cpu");
    for (int g = 0; g < devices; ++g) {
      Device& dev = gpu[size_t(g)];
      BATCH_HIP_OK(hipSetDevice(g));
      for (size_t slot = 0; slot < dev.buffers.size(); ++slot) {
        const int i = dev.arrays[slot];
        std::vector<uint32_t> k(n), v(n);
        BATCH_HIP_OK(hipMemcpy(k.data(), dev.buffers[slot], size_t(n) * 4, hipMemcpyDeviceToHost));
        BATCH_HIP_OK(hipMemcpy(v.data(), dev.buffers[slot] + inout, size_t(n) * 4, hipMemcpyDeviceToHost));
        bool ok = std::is_sorted(k.begin(), k.end());
        uint64_t sumBefore = 0, sumAfter = 0;
        for (uint32_t j = 0; j < n; ++j) {
          sumBefore += pairHash(keys[size_t(i)][j], values[size_t(i)][j]);
          sumAfter += pairHash(k[j], v[j]);
        }
        ok = ok && sumBefore == sumAfter;
        if (ok) {
          // stability: the values of every run of equal keys, in output order, must be the input's values of that
          // key in input order (uniform 32-bit keys: ~n^2 / 2^33 short runs).  One pass over the output collects the
          // duplicated keys, one pass over the input replays their values in input order.
          std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> runs;  // key -> (next output index, end of its run)
          for (uint32_t j = 0; j + 1 < n; ++j) {
            if (k[j] != k[j + 1]) continue;
            uint32_t end = j + 1;
            while (end < n && k[end] == k[j]) ++end;
            runs.emplace(k[j], std::make_pair(j, end));
            j = end - 1;
          }
          for (uint32_t q = 0; ok && q < n; ++q) {
            const auto it = runs.find(keys[size_t(i)][q]);
            if (it == runs.end()) continue;
            ok = it->second.first < it->second.second && v[it->second.first] == values[size_t(i)][q];
            ++it->second.first;
          }
          for (const auto& r : runs) ok = ok && r.second.first == r.second.second;
        }
        if (ok && slot == 0) {
          const auto want = cpu->SortKeyValue(keys[size_t(i)], values[size_t(i)]);
          ok = want.keys == k && want.values == v;
        }
        if (!ok) {
          ++failures;
          std::fprintf(stderr, "array %d on gpu %d is wrong\n", i, g);
        }
      }
    }
    if (failures == 0) std::printf("Correctness check passed (%d arrays)\n", arrays);
  }

  for (int g = 0; g < devices; ++g) {
    Device& dev = gpu[size_t(g)];
    (void)hipSetDevice(g);
    (void)ncclCommDestroy(comms[size_t(g)]);
    for (uint8_t* b : dev.buffers) (void)hipFree(b);
    (void)hipFree(dev.storage);
    (void)hipFree(dev.record);
    (void)hipFree(dev.gathered);
    (void)hipEventDestroy(dev.first);
    (void)hipEventDestroy(dev.last);
    (void)hipStreamDestroy(dev.stream);
    vrdxDestroySorter(dev.sorter);
  }
  return failures == 0 ? 0 : 1;
}

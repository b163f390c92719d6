// Backends of the bench driver.  Mirrors the reference's BenchmarkBase / BenchmarkFactory /
// DataGenerator interfaces (/root/reference/bench/benchmark_base.h:9-29, benchmark_factory.h,
// data_generator.h:8-24) so that bench.cc reads like the reference's driver.
#ifndef VRDX_BENCH_BACKENDS_H
#define VRDX_BENCH_BACKENDS_H

#include <cstdint>
#include <memory>
#include <random>
#include <string>
#include <vector>

class BenchmarkBase {
 public:
  struct Results {
    std::vector<uint32_t> keys;
    std::vector<uint32_t> values;
    uint64_t total_time = 0;    // ns, device time of the whole sort
    uint64_t cpu_time = 0;      // ns, wall clock submit -> completion
    uint64_t upsweep_ns = 0;    // ns, summed over the 4 passes (ours: the fused histogram)
    uint64_t spine_ns = 0;      // ns (ours: 0, the scan is the look-back inside downsweep)
    uint64_t downsweep_ns = 0;  // ns, summed over the 4 passes
    // HBM bytes per element of the sort that ran (hip: the recorded plan's, vrdxHipDescribePlan, when the device took it --
    // vrdxHipReadPlanVerdict --, the four passes' otherwise); 0 = not known: priced as four passes
    uint32_t bytes_per_element = 0;
  };
  virtual ~BenchmarkBase() = default;
  virtual std::string LibraryVersion() const { return ""; }
  // false if any sort since the last call failed on the device (hip: vrdxHipReadSorterStatus)
  virtual bool Healthy() { return true; }
  virtual Results Sort(const std::vector<uint32_t>& keys) = 0;
  virtual Results SortKeyValue(const std::vector<uint32_t>& keys, const std::vector<uint32_t>& values) = 0;
};

struct SortData {
  std::vector<uint32_t> keys;
  std::vector<uint32_t> values;
};

// N keys first, then N values, raw mt19937 outputs (what libstdc++'s uniform_int_distribution
// over the full uint32 range yields; bench/data_generator.cc:12-26).
class DataGenerator {
 public:
  DataGenerator() : gen_(std::random_device{}()) {}
  explicit DataGenerator(int seed) : gen_(seed) {}
  SortData Generate(uint32_t size, uint32_t bits = 32) {
    SortData d;
    d.keys.resize(size);
    d.values.resize(size);
    for (auto& k : d.keys) {
      const uint32_t x = gen_();
      k = bits >= 32 ? x : (bits == 0 ? 0u : x >> (32 - bits));
    }
    for (auto& v : d.values) v = gen_();
    return d;
  }

 private:
  std::mt19937 gen_;
};

// "hip" (libvrdx_hip.so through the C-ABI), "cpu" (std::sort / std::stable_sort), and "rocprim"
// when the driver was built with the comparator (bench/rocprim_backend.hip).
// graph (hip only): every sort is captured ONCE per (N, mode) into a hipGraph and the timed runs replay it -- the
// reference's record-once / submit-many model (a command buffer is recorded once and submitted per sort,
// /root/reference/bench/vulkan_benchmark.cc:292-302): one host call per sort instead of seven or eight enqueues.
std::unique_ptr<BenchmarkBase> CreateBenchmark(const std::string& type, bool graph = false);
std::unique_ptr<BenchmarkBase> CreateRocprimBenchmark();  // nullptr when not compiled in

// bench hip --devices G: returns the process exit code (batched.cc).
int RunBatched(int devices, int arrays, int log2n, bool verify);

#endif  // VRDX_BENCH_BACKENDS_H

// bench -- benchmark driver for the HIP backend, following the reference's protocol and CSV format
// (/root/reference/bench/bench.cc:15-20,41-112,116-207) so that results are directly comparable with
// the reference's README table and its tools/plot.py:
//
//   bench <type> [-o results.csv] [--no-verify] [--points K] [--graph]
//     <type>      hip | cpu | rocprim            (reference: vulkan | cpu | cuda | fuchsia)
//     -o          output CSV (default results.csv)
//     --no-verify skip the one-shot correctness check against the cpu backend at the first point
//     --points K  number of sweep points between N = 2^18 and 2^25 (default 128, like the reference)
//     --min-log2n A / --max-log2n B   other sweep ends (the reference hard-codes 18 and 25)
//     --graph     hip: capture every sort once per (N, mode) into a hipGraph and time its replay (the reference's
//                 record-once / submit-many model, bench/vulkan_benchmark.cc:292-302); cpu_ms is then one host call
//   bench hip --devices G [--arrays A] [--log2n L]
//     the batched many-arrays variant: A independent key+value arrays of 2^L elements, array i on
//     GPU i mod G, one VrdxSorter + stream + storage per GPU, all enqueued from this one host thread;
//     the per-GPU {status, elapsed} records are exchanged with ncclAllGather (RCCL) -- see batched.cc
//
// Per (N, keys|kv): 1 warm-up + 10 timed runs on FRESH data each run, median.  CSV columns are the
// reference's seven (backend,n,sort,gpu_ms,cpu_ms,gpu_gitems_s,cpu_gitems_s) followed by
// achieved_GBps and hbm_fraction (algorithmic bytes 36 B/key, 68 B/pair over 8 TB/s).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../include/vk_radix_sort.h"
#include "backends.h"

namespace {

// protocol constants of the reference's driver (bench/bench.cc:15-20)
constexpr int kWarmupRuns = 1;
constexpr int kTimedRuns = 10;
uint32_t kNMin = 1u << 18;
uint32_t kNMax = 1u << 25;

// One (N, sort) line of the sweep: the five nanosecond series of its timed runs, reduced on demand.
class Series {
 public:
  enum Column { kDevice, kWall, kUpsweep, kSpine, kDownsweep, kColumns };
  void Add(const BenchmarkBase::Results& r) {
    const uint64_t sample[kColumns] = {r.total_time, r.cpu_time, r.upsweep_ns, r.spine_ns, r.downsweep_ns};
    for (int c = 0; c < kColumns; ++c) ns_[c].push_back(sample[c]);
    if (r.bytes_per_element != 0) bytesPerElement_ = r.bytes_per_element;
  }
  // of the last run that said so (the plan is a function of N and the mode; the device's verdict of the data); 0: unknown
  uint32_t BytesPerElement() const { return bytesPerElement_; }
  // upper median, like the reference's (element size/2 of the sorted series)
  double MedianMs(Column c) const {
    std::vector<uint64_t> sorted(ns_[c]);
    if (sorted.empty()) return 0.0;
    std::sort(sorted.begin(), sorted.end());
    return 1e-6 * static_cast<double>(sorted[sorted.size() / 2]);
  }

 private:
  std::vector<uint64_t> ns_[kColumns];
  uint32_t bytesPerElement_ = 0;
};

struct Line {
  uint32_t n;
  const char* sort;  // "keys" | "kv"
  Series series;
  double Ms(Series::Column c) const { return series.MedianMs(c); }
  double GItemsPerSecond(Series::Column c) const {
    const double ms = Ms(c);
    return ms > 0.0 ? static_cast<double>(n) / (ms * 1e6) : 0.0;
  }
  // HBM bytes the sort moved over its device time: the bytes of the plan that RAN where the backend says which
  // (hip: 20 B/key and 36 B/pair for the two-trip plans, vrdxHipDescribePlan + vrdxHipReadPlanVerdict), else the four
  // passes' of SURVEY.md section 8(d): 36 B/key, 68 B/pair.  (Column names as tools/plot.py expects them,
  // /root/reference/tools/plot.py:25-50.)
  double AlgorithmicGBps() const {
    const double ms = Ms(Series::kDevice);
    const double bytes = series.BytesPerElement() != 0 ? series.BytesPerElement() : (std::strcmp(sort, "keys") == 0 ? 36.0 : 68.0);
    return ms > 0.0 ? bytes * n / (ms * 1e6) : 0.0;
  }
};

// The reference's predicate (bench/bench.cc:41-64): the backend under test and the cpu backend agree on
// every key, and on every (key, value) pair of the stable key+value sort.
bool AgreesWithCpu(BenchmarkBase& tested, BenchmarkBase& cpu, uint32_t n, DataGenerator& gen) {
  const SortData input = gen.Generate(n);
  const auto firstDifference = [](const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    return static_cast<size_t>(std::mismatch(a.begin(), a.end(), b.begin(), b.end()).first - a.begin());
  };
  {
    const auto got = tested.Sort(input.keys), want = cpu.Sort(input.keys);
    const size_t at = firstDifference(got.keys, want.keys);
    if (at != n || got.keys.size() != want.keys.size()) {
      std::cerr << "keys-only sort differs from the cpu backend at element " << at << " of " << n << std::endl;
      return false;
    }
  }
  {
    const auto got = tested.SortKeyValue(input.keys, input.values), want = cpu.SortKeyValue(input.keys, input.values);
    const size_t at = std::min(firstDifference(got.keys, want.keys), firstDifference(got.values, want.values));
    if (at != n || got.keys.size() != want.keys.size() || got.values.size() != want.values.size()) {
      std::cerr << "key+value sort differs from the cpu backend at element " << at << " of " << n << std::endl;
      return false;
    }
  }
  std::cout << "Correctness check passed (N=" << n << ")" << std::endl;
  return true;
}

// kWarmupRuns + kTimedRuns sorts, every one on freshly generated data (bench/bench.cc:66-112).
Line Measure(BenchmarkBase& bench, uint32_t n, const char* sort, DataGenerator& gen) {
  Line line{n, sort, {}};
  const bool keysOnly = std::strcmp(sort, "keys") == 0;
  for (int run = -kWarmupRuns; run < kTimedRuns; ++run) {
    const SortData input = gen.Generate(n);
    const BenchmarkBase::Results r = keysOnly ? bench.Sort(input.keys) : bench.SortKeyValue(input.keys, input.values);
    if (run >= 0) line.series.Add(r);
  }
  return line;
}

}  // namespace

int main(int argc, char** argv) {
  std::string type, output = "results.csv";
  bool verify = true, graph = false;
  int points = 128;  // bench/bench.cc:19 kNCount
  int devices = 0, arrays = 8, batchLog2n = 25;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-o" || a == "--output") {
      if (++i < argc) output = argv[i];
    } else if (a == "--no-verify") {
      verify = false;
    } else if (a == "--graph") {
      graph = true;
    } else if (a == "--points") {
      if (++i < argc) points = std::max(2, std::atoi(argv[i]));
    } else if (a == "--min-log2n") {
      if (++i < argc) kNMin = 1u << std::min(29, std::max(0, std::atoi(argv[i])));
    } else if (a == "--max-log2n") {
      if (++i < argc) kNMax = 1u << std::min(29, std::max(0, std::atoi(argv[i])));
    } else if (a == "--devices") {
      if (++i < argc) devices = std::max(1, std::atoi(argv[i]));
    } else if (a == "--arrays") {
      if (++i < argc) arrays = std::max(1, std::atoi(argv[i]));
    } else if (a == "--log2n") {
      if (++i < argc) batchLog2n = std::min(29, std::max(10, std::atoi(argv[i])));
    } else if (a == "-h" || a == "--help") {
      type.clear();
      break;
    } else if (type.empty()) {
      type = a;
    }
  }
  if (type.empty()) {
    std::cout << "usage: bench <hip|cpu|rocprim> [-o results.csv] [--no-verify] [--points K] [--min-log2n A] [--max-log2n B] [--graph]\n"
                 "       bench hip --devices G [--arrays A] [--log2n L] [--no-verify]   (batched: A key+value arrays of 2^L over G GPUs)"
              << std::endl;
    return 0;
  }

  if (devices > 0) {  // the batched many-arrays variant (BASELINE.json configs[4]): see batched.cc
    if (type != "hip") {
      std::cerr << "--devices needs the hip backend" << std::endl;
      return 1;
    }
    return RunBatched(devices, arrays, batchLog2n, verify);
  }

  if (graph && type != "hip") {
    std::cerr << "--graph needs the hip backend" << std::endl;
    return 1;
  }
  std::unique_ptr<BenchmarkBase> bench = CreateBenchmark(type, graph);
  if (!bench) {
    std::cerr << "unknown or unavailable backend: " << type << std::endl;
    return 1;
  }
  std::unique_ptr<BenchmarkBase> cpu = CreateBenchmark("cpu");
  DataGenerator gen;  // random_device seeded, like the reference (bench/bench.cc:158)

  const uint32_t step = (kNMax - kNMin) / static_cast<uint32_t>(points - 1);  // :20
  std::vector<Line> lines;
  for (int i = 0; i < points; ++i) {
    const uint32_t n = i == points - 1 ? kNMax : kNMin + step * static_cast<uint32_t>(i);
    if (i == 0 && verify && type != "cpu" && !AgreesWithCpu(*bench, *cpu, n, gen)) return 1;
    for (const char* sort : {"keys", "kv"}) {
      lines.push_back(Measure(*bench, n, sort, gen));
      const Line& l = lines.back();
      std::cout << std::fixed << std::setprecision(3) << type << " n=" << l.n << " " << l.sort << "  gpu "
                << l.Ms(Series::kDevice) << " ms (" << l.GItemsPerSecond(Series::kDevice) << " GItems/s)  wall "
                << l.Ms(Series::kWall) << " ms";
      const double up = l.Ms(Series::kUpsweep), sp = l.Ms(Series::kSpine), dn = l.Ms(Series::kDownsweep);
      if (up + sp + dn > 0)
        std::cout << "  [up " << 100.0 * up / (up + sp + dn) << "% sp " << 100.0 * sp / (up + sp + dn) << "% dn "
                  << 100.0 * dn / (up + sp + dn) << "%]";
      std::cout << std::endl;
    }
  }
  if (!bench->Healthy()) {
    std::cerr << "the device reported a failed sort (bounded look-back spin expired)" << std::endl;
    return 1;
  }

  std::ofstream csv(output);
  const std::string version = bench->LibraryVersion();
  if (!version.empty()) csv << "# version: " << version << "\n";  // bench/bench.cc:197-198, read by tools/plot.py:53-57
  csv << "backend,n,sort,gpu_ms,cpu_ms,gpu_gitems_s,cpu_gitems_s,achieved_GBps,hbm_fraction\n";
  for (const Line& l : lines)
    csv << type << "," << l.n << "," << l.sort << "," << std::setprecision(6) << l.Ms(Series::kDevice) << "," << l.Ms(Series::kWall)
        << "," << l.GItemsPerSecond(Series::kDevice) << "," << l.GItemsPerSecond(Series::kWall) << "," << l.AlgorithmicGBps() << ","
        << l.AlgorithmicGBps() / 8000.0 << "\n";
  std::cout << "wrote " << output << std::endl;
  return 0;
}

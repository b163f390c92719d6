// bench -- benchmark driver for the HIP backend, following the reference's protocol and CSV format
// (/root/reference/bench/bench.cc:15-20,41-112,116-207) so that results are directly comparable with
// the reference's README table and its tools/plot.py:
//
//   bench <type> [-o results.csv] [--no-verify] [--points K]
//     <type>      hip | cpu | rocprim            (reference: vulkan | cpu | cuda | fuchsia)
//     -o          output CSV (default results.csv)
//     --no-verify skip the one-shot correctness check against the cpu backend at the first point
//     --points K  number of sweep points between N = 2^18 and 2^25 (default 128, like the reference)
//     --min-log2n A / --max-log2n B   other sweep ends (the reference hard-codes 18 and 25)
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

constexpr int kWarmupRuns = 1;                  // bench/bench.cc:15
constexpr int kTimedRuns = 10;                  // :16
uint32_t kNMin = 1u << 18;                      // :17
uint32_t kNMax = 1u << 25;                      // :18

double toMs(uint64_t ns) { return static_cast<double>(ns) / 1e6; }
double toGItemsS(uint32_t n, uint64_t ns) { return ns ? (static_cast<double>(n) / 1e9) / (static_cast<double>(ns) / 1e9) : 0.0; }

uint64_t median(std::vector<uint64_t>& v) {
  auto mid = static_cast<std::ptrdiff_t>(v.size() / 2);
  std::nth_element(v.begin(), v.begin() + mid, v.end());
  return v[static_cast<size_t>(mid)];
}

struct Row {
  uint32_t n;
  std::string sort;
  double gpu_ms, cpu_ms, gpu_gitems_s, cpu_gitems_s, upsweep_ms, spine_ms, downsweep_ms;
};

bool checkCorrectness(BenchmarkBase* bench, BenchmarkBase* cpu, uint32_t n, DataGenerator& gen) {
  auto data = gen.Generate(n);
  auto r0 = bench->Sort(data.keys);
  auto r1 = cpu->Sort(data.keys);
  for (uint32_t i = 0; i < n; ++i)
    if (r0.keys[i] != r1.keys[i]) {
      std::cerr << "Sort correctness failed at index " << i << std::endl;
      return false;
    }
  auto r2 = bench->SortKeyValue(data.keys, data.values);
  auto r3 = cpu->SortKeyValue(data.keys, data.values);
  for (uint32_t i = 0; i < n; ++i)
    if (r2.keys[i] != r3.keys[i] || r2.values[i] != r3.values[i]) {
      std::cerr << "SortKeyValue correctness failed at index " << i << std::endl;
      return false;
    }
  std::cout << "Correctness check passed (N=" << n << ")" << std::endl;
  return true;
}

Row measure(BenchmarkBase* bench, uint32_t n, const std::string& sort, DataGenerator& gen) {
  auto once = [&](const SortData& d) { return sort == "keys" ? bench->Sort(d.keys) : bench->SortKeyValue(d.keys, d.values); };
  for (int i = 0; i < kWarmupRuns; ++i) once(gen.Generate(n));
  std::vector<uint64_t> gpu, cpu, up, sp, dn;
  for (int i = 0; i < kTimedRuns; ++i) {
    auto r = once(gen.Generate(n));
    gpu.push_back(r.total_time);
    cpu.push_back(r.cpu_time);
    up.push_back(r.upsweep_ns);
    sp.push_back(r.spine_ns);
    dn.push_back(r.downsweep_ns);
  }
  const uint64_t g = median(gpu), c = median(cpu);
  return Row{n, sort, toMs(g), toMs(c), toGItemsS(n, g), toGItemsS(n, c), toMs(median(up)), toMs(median(sp)), toMs(median(dn))};
}

}  // namespace

int main(int argc, char** argv) {
  std::string type, output = "results.csv";
  bool verify = true;
  int points = 128;  // bench/bench.cc:19 kNCount
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-o" || a == "--output") {
      if (++i < argc) output = argv[i];
    } else if (a == "--no-verify") {
      verify = false;
    } else if (a == "--points") {
      if (++i < argc) points = std::max(2, std::atoi(argv[i]));
    } else if (a == "--min-log2n") {
      if (++i < argc) kNMin = 1u << std::min(29, std::max(0, std::atoi(argv[i])));
    } else if (a == "--max-log2n") {
      if (++i < argc) kNMax = 1u << std::min(29, std::max(0, std::atoi(argv[i])));
    } else if (a == "-h" || a == "--help") {
      type.clear();
      break;
    } else if (type.empty()) {
      type = a;
    }
  }
  if (type.empty()) {
    std::cout << "usage: bench <hip|cpu|rocprim> [-o results.csv] [--no-verify] [--points K] [--min-log2n A] [--max-log2n B]"
              << std::endl;
    return 0;
  }

  std::unique_ptr<BenchmarkBase> bench = CreateBenchmark(type);
  if (!bench) {
    std::cerr << "unknown or unavailable backend: " << type << std::endl;
    return 1;
  }
  std::unique_ptr<BenchmarkBase> cpu = CreateBenchmark("cpu");
  DataGenerator gen;  // random_device seeded, like the reference (bench/bench.cc:158)

  const uint32_t step = (kNMax - kNMin) / static_cast<uint32_t>(points - 1);  // :20
  std::vector<Row> rows;
  for (int i = 0; i < points; ++i) {
    const uint32_t n = i == points - 1 ? kNMax : kNMin + step * static_cast<uint32_t>(i);
    if (i == 0 && verify && type != "cpu" && !checkCorrectness(bench.get(), cpu.get(), n, gen)) return 1;
    for (const char* sort : {"keys", "kv"}) {
      Row r = measure(bench.get(), n, sort, gen);
      std::cout << std::fixed << std::setprecision(3) << type << " n=" << r.n << " " << r.sort << "  gpu " << r.gpu_ms << " ms ("
                << r.gpu_gitems_s << " GItems/s)  wall " << r.cpu_ms << " ms";
      const double stages = r.upsweep_ms + r.spine_ms + r.downsweep_ms;
      if (stages > 0)
        std::cout << "  [up " << 100.0 * r.upsweep_ms / stages << "% sp " << 100.0 * r.spine_ms / stages << "% dn "
                  << 100.0 * r.downsweep_ms / stages << "%]";
      std::cout << std::endl;
      rows.push_back(r);
    }
  }

  std::ofstream csv(output);
  const std::string version = bench->LibraryVersion();
  if (!version.empty()) csv << "# version: " << version << "\n";  // bench/bench.cc:197-198, read by tools/plot.py:53-57
  csv << "backend,n,sort,gpu_ms,cpu_ms,gpu_gitems_s,cpu_gitems_s,achieved_GBps,hbm_fraction\n";
  for (const Row& r : rows) {
    const double bytes = (r.sort == "keys" ? 36.0 : 68.0) * r.n;
    const double gbps = r.gpu_ms > 0 ? bytes / (r.gpu_ms * 1e-3) / 1e9 : 0.0;
    csv << type << "," << r.n << "," << r.sort << "," << std::setprecision(6) << r.gpu_ms << "," << r.cpu_ms << "," << r.gpu_gitems_s
        << "," << r.cpu_gitems_s << "," << gbps << "," << gbps / 8000.0 << "\n";
  }
  std::cout << "wrote " << output << std::endl;
  return 0;
}

// ref_shim.cc -- extern "C" doorway into the REFERENCE's own CPU backend, compiled from the
// sources where they lie under /root/reference (never copied into this repository):
//   bench/cpu_benchmark.{h,cc}, bench/data_generator.{h,cc}, bench/benchmark_base.h
// Built by oracle/Makefile into oracle/_ref/libvrdx_ref.so (git-ignored, travels to the GPU box).
// TEST / BASELINE INFRASTRUCTURE ONLY.
#include <cstdint>
#include <cstring>
#include <vector>

#include "cpu_benchmark.h"    // -I/root/reference/bench
#include "data_generator.h"

extern "C" {

// CpuBenchmark::Sort (bench/cpu_benchmark.cc:19-28).  Returns Results::total_time (ns).
int64_t vrdx_ref_sort_keys(uint32_t* keys, uint64_t n) {
  std::vector<uint32_t> in(keys, keys + n);
  CpuBenchmark cpu;
  auto r = cpu.Sort(in);
  if (n) std::memcpy(keys, r.keys.data(), n * sizeof(uint32_t));
  return (int64_t)r.total_time;
}

// CpuBenchmark::SortKeyValue (bench/cpu_benchmark.cc:30-53).  Returns Results::total_time (ns).
int64_t vrdx_ref_sort_key_value(uint32_t* keys, uint32_t* values, uint64_t n) {
  std::vector<uint32_t> k(keys, keys + n), v(values, values + n);
  CpuBenchmark cpu;
  auto r = cpu.SortKeyValue(k, v);
  if (n) {
    std::memcpy(keys, r.keys.data(), n * sizeof(uint32_t));
    std::memcpy(values, r.values.data(), n * sizeof(uint32_t));
  }
  return (int64_t)r.total_time;
}

// DataGenerator(seed).Generate(size, bits) (bench/data_generator.cc:8,12-26).
void vrdx_ref_generate(int32_t seed, uint32_t size, uint32_t bits, uint32_t* keys, uint32_t* values) {
  DataGenerator gen(seed);
  SortData d = gen.Generate(size, bits);
  if (size) {
    std::memcpy(keys, d.keys.data(), size * sizeof(uint32_t));
    if (values) std::memcpy(values, d.values.data(), size * sizeof(uint32_t));
  }
}

}  // extern "C"

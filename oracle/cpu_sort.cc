// cpu_sort.cc -- our C++ restatement of the reference's CPU backend and data generator.
// TEST / BASELINE INFRASTRUCTURE ONLY (see the header of vrdx_oracle.c for who may load it).
//
//   vrdx_port_sort_keys       <- CpuBenchmark::Sort          (bench/cpu_benchmark.cc:19-28)
//   vrdx_port_sort_key_value  <- CpuBenchmark::SortKeyValue  (bench/cpu_benchmark.cc:30-53)
//   vrdx_port_generate        <- DataGenerator::Generate     (bench/data_generator.cc:12-26)
//
// Timing follows the reference: only the std::sort / std::stable_sort call is inside the clock
// (bench/cpu_benchmark.cc:22-25, 38-41); single-threaded, like the reference.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <numeric>
#include <random>
#include <vector>

namespace {
int64_t Now() { return std::chrono::high_resolution_clock::now().time_since_epoch().count(); }
}  // namespace

extern "C" {

// Sorts keys[0..n) ascending in place; returns the nanoseconds spent inside std::sort.
int64_t vrdx_port_sort_keys(uint32_t* keys, uint64_t n) {
  const int64_t start = Now();
  std::sort(keys, keys + n);
  const int64_t end = Now();
  return end - start;
}

// Stable sort of (key, value) pairs by key: std::stable_sort of an index vector, then gather.
// Returns the nanoseconds spent inside std::stable_sort.
int64_t vrdx_port_sort_key_value(uint32_t* keys, uint32_t* values, uint64_t n) {
  std::vector<uint32_t> indices(n);
  std::iota(indices.begin(), indices.end(), 0u);
  const int64_t start = Now();
  std::stable_sort(indices.begin(), indices.end(),
                   [&](uint32_t lhs, uint32_t rhs) { return keys[lhs] < keys[rhs]; });
  const int64_t end = Now();
  std::vector<uint32_t> k(n), v(n);
  for (uint64_t i = 0; i < n; ++i) {
    k[i] = keys[indices[i]];
    v[i] = values[indices[i]];
  }
  std::copy(k.begin(), k.end(), keys);
  std::copy(v.begin(), v.end(), values);
  return end - start;
}

// DataGenerator(seed).Generate(size, bits): N keys first, then N values, from ONE mt19937 stream.
// With libstdc++ >= 11 uniform_int_distribution<uint32_t> over the full range returns the raw
// engine output, and over [0, 2^bits) it returns the top `bits` bits of one output (Lemire's
// multiply-shift never rejects for a power-of-two range).  tests/test_oracle.py checks this
// against the reference's own generator (oracle/_ref) wherever that is built.
void vrdx_port_generate(int32_t seed, uint32_t size, uint32_t bits, uint32_t* keys, uint32_t* values) {
  std::mt19937 gen(seed);
  for (uint32_t i = 0; i < size; ++i) {
    const uint32_t x = gen();
    keys[i] = bits >= 32 ? x : (bits == 0 ? 0u : (x >> (32 - bits)));
  }
  if (values != nullptr)
    for (uint32_t i = 0; i < size; ++i) values[i] = gen();
}

}  // extern "C"

// Host-only pieces under AddressSanitizer + UndefinedBehaviorSanitizer (CPU test run; GPU sanitizers are
// not available on this pool): the storage-layout math shared by host and device (vrdx_layout.h), the
// oracle's restatement of the reference's three shaders (vrdx_oracle.c) and our port of the reference's
// CPU backend / data generator (cpu_sort.cc), on ragged sizes, against std::stable_sort.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#include "../../vulkan_radix_sort_amd/csrc/vrdx_layout.h"

extern "C" {
int vrdx_oracle_sort(uint32_t* keys, uint32_t* values, uint32_t elementCount, uint32_t* globalHistogramOut);
uint64_t vrdx_oracle_storage_size(uint32_t maxElementCount, uint32_t align, int keyValue);
void vrdx_oracle_digit_counts(const uint32_t* keys, uint32_t elementCount, uint32_t out[4 * 256]);
uint64_t vrdx_port_sort_keys(uint32_t* keys, uint64_t n);
uint64_t vrdx_port_sort_key_value(uint32_t* keys, uint32_t* values, uint64_t n);
}

int main() {
  int failures = 0;
  std::mt19937 g(99);
  for (uint32_t n : {0u, 1u, 2u, 63u, 4095u, 4096u, 4097u, 12411u, 65539u, 200001u}) {
    std::vector<uint32_t> k(n), v(n);
    for (auto& x : k) x = (n % 3 == 0) ? (g() & 0xFFu) : g();
    std::iota(v.begin(), v.end(), 0u);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
    std::vector<uint32_t> ok(k), ov(v), hist(4 * 256), counts(4 * 256);
    if (vrdx_oracle_sort(ok.data(), ov.data(), n, hist.data()) != 0) ++failures;
    vrdx_oracle_digit_counts(k.data(), n, counts.data());
    std::vector<uint32_t> pk(k), pv(v);
    vrdx_port_sort_key_value(pk.data(), pv.data(), n);
    std::vector<uint32_t> qk(k);
    vrdx_port_sort_keys(qk.data(), n);
    for (uint32_t i = 0; i < n; ++i)
      if (ok[i] != k[order[i]] || ov[i] != order[i] || pk[i] != ok[i] || pv[i] != ov[i] || qk[i] != ok[i]) {
        ++failures;
        break;
      }
    uint64_t total = 0;
    for (uint32_t c = 0; c < 256; ++c) total += counts[c];
    if (total != n) ++failures;
    for (uint32_t tile : {8192u, 16384u, 32768u, 65536u}) {
      const vrdx::StorageLayout l = vrdx::MakeLayout(n, VRDX_STORAGE_ALIGN, vrdx::RoundUp(n, tile), 0x1000u + 16u * (n % 8));
      if (l.keysOnlySize != vrdx_oracle_storage_size(n, VRDX_STORAGE_ALIGN, 0) ||
          l.keyValueSize != vrdx_oracle_storage_size(n, VRDX_STORAGE_ALIGN, 1))
        ++failures;
    }
  }
  const vrdx::StorageLayout top = vrdx::MakeLayout(VRDX_MAX_ELEMENTS, VRDX_STORAGE_ALIGN, vrdx::RoundUp(VRDX_MAX_ELEMENTS, 8192));
  if (top.keyValueSize != vrdx_oracle_storage_size(VRDX_MAX_ELEMENTS, VRDX_STORAGE_ALIGN, 1)) ++failures;
  std::printf("sanitized host check: %d failures\n", failures);
  return failures != 0;
}

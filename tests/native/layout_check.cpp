// CPU-only check of vrdx_layout.h: for a sweep of element counts and every tile size in use, the
// device state (histogram table, two status regions, tickets) stays inside the region the
// reference's storage formula provides, the tickets keep a 128-byte line to themselves, and the
// totals equal the oracle's restatement of the reference formulas.  Built and run by
// tests/test_abi.py (no GPU, no HIP).
#include <cstdint>
#include <cstdio>

#include "../../vulkan_radix_sort_amd/csrc/vrdx_layout.h"

extern "C" uint64_t vrdx_oracle_storage_size(uint32_t n, uint32_t align, int key_value);

int main() {
  // the kernels' capacities and strides in between (tiles start a multiple of 256 keys >= 8192 apart, PlanTiles)
  const uint32_t tileSizes[] = {8192, 8448, 10240, 12288, 14336, 16384, 23808, 32768, 35840, 49408, 65536};
  uint64_t cases = 0;
  int failures = 0;
  auto check = [&](uint32_t n) {
    for (uint32_t t : tileSizes) {
      const vrdx::StorageLayout l = vrdx::MakeLayout(n, VRDX_STORAGE_ALIGN, t);
      ++cases;
      const uint64_t partitions = ((uint64_t)n + VRDX_REF_PARTITION_SIZE - 1) / VRDX_REF_PARTITION_SIZE;
      const uint64_t areaEnd = l.statusOffset + partitions * 1024;  // end of the reference's partition histograms
      const uint64_t region1End = l.statusOffset + 2 * l.statusRows * 1024;
      bool ok = true;
      ok = ok && l.keysOnlySize == vrdx_oracle_storage_size(n, VRDX_STORAGE_ALIGN, 0);
      ok = ok && l.keyValueSize == vrdx_oracle_storage_size(n, VRDX_STORAGE_ALIGN, 1);
      ok = ok && l.histogramOffset == 16 && l.statusOffset == 16 + 4096;
      ok = ok && l.clearBytes == l.statusOffset + l.statusRows * 1024;
      if (n > 0) {
        ok = ok && region1End <= areaEnd;                       // both status regions fit
        ok = ok && l.ticketOffset >= region1End + 128;          // a line of their own, after the status words
        ok = ok && l.ticketOffset + 8 + 120 <= areaEnd + 16;    // ... and still inside the area (+ its 16 B slack)
        ok = ok && l.ticketOffset + 8 <= l.inoutOffset;
      }
      ok = ok && l.inoutOffset == 16 + vrdx::HistogramSize(n, VRDX_STORAGE_ALIGN);
      if (!ok) {
        if (failures < 10) std::printf("FAIL n=%u tile=%u\n", n, t);
        ++failures;
      }
    }
  };
  for (uint32_t n = 0; n <= 70000; ++n) check(n);
  for (uint64_t n = 70001; n <= VRDX_MAX_ELEMENTS; n += 1 + n / 977) check((uint32_t)n);
  for (uint32_t lg = 10; lg < 30; ++lg)
    for (int d = -2; d <= 2; ++d) check((1u << lg) + d);
  check(VRDX_MAX_ELEMENTS);
  std::printf("layout: %llu cases, %d failures\n", (unsigned long long)cases, failures);
  return failures != 0;
}

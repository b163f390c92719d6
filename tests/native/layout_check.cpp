// CPU-only check of vrdx_layout.h: for a sweep of element counts, every tile geometry in use with every tile plan
// PlanTiles can make of it (full tiles, even split, tail split; several CU counts) and every 16-byte alignment of the
// storage address, the device state (histogram table, two status regions, ticket line) and the 128-byte aligned
// scratch arrays stay inside the size the reference's storage formula provides, nothing overlaps, the plan covers
// every key exactly once, and the totals equal the oracle's restatement of the reference formulas.  Built and run by
// tests/test_abi.py (no GPU, no HIP).
#include <cstdint>
#include <cstdio>
#include <initializer_list>

#include "../../vulkan_radix_sort_amd/csrc/vrdx_layout.h"

extern "C" uint64_t vrdx_oracle_storage_size(uint32_t n, uint32_t align, int key_value);

int main() {
  struct Geometry {
    uint32_t threads, keysPerThread, subTiles;
    bool splitForms;
  };
  const Geometry geometries[] = {{1024, 8, 1, false}, {1024, 16, 1, false}, {1024, 32, 1, false}, {1024, 32, 1, true},
                                 {1024, 32, 2, false}, {1024, 32, 2, true}};
  const uint32_t cuCounts[] = {256, 304, 64, 8};
  uint64_t cases = 0;
  int failures = 0;
  auto check = [&](uint32_t n, bool allAlignments) {
    for (const Geometry& g : geometries) {
      for (uint32_t cus : cuCounts) {
        if (!g.splitForms && cus != 256) continue;  // the plan does not depend on the CU count then
        const vrdx::TilePlan plan = vrdx::PlanTiles(n, cus, g.threads, g.keysPerThread, g.subTiles, g.splitForms, true, 100);
        // the plan covers [0, n) with tiles of a multiple of four slots, none larger than the kernel's capacity
        bool ok = true;
        const uint32_t slotKeys = g.threads * g.subTiles;
        if (plan.slots == 0) {
          ok = ok && plan.tiles == vrdx::RoundUp(n, slotKeys * g.keysPerThread);
        } else {
          ok = ok && g.splitForms && plan.slots % 4 == 0 && plan.tailSlots % 4 == 0 && plan.tailSlots >= 4;
          ok = ok && plan.slots <= g.keysPerThread && plan.tailSlots <= g.keysPerThread;
          const uint64_t frameA = (uint64_t)plan.slots * slotKeys, frameB = (uint64_t)plan.tailSlots * slotKeys;
          if (plan.fullTiles == ~0u) {
            ok = ok && (uint64_t)plan.tiles * frameA >= n && (uint64_t)(plan.tiles - 1) * frameA < n;
            ok = ok && plan.tiles <= cus;
          } else {
            const uint64_t keysA = (uint64_t)plan.fullTiles * frameA;
            ok = ok && plan.slots == g.keysPerThread && plan.fullTiles % cus == 0 && keysA < n && plan.tiles > plan.fullTiles;
            ok = ok && keysA + (uint64_t)(plan.tiles - plan.fullTiles) * frameB >= n;
            ok = ok && keysA + (uint64_t)(plan.tiles - plan.fullTiles - 1) * frameB < n;
            ok = ok && plan.tiles - plan.fullTiles <= cus;
          }
        }
        // the MSD plan (MsdBits in vrdx_api.cpp: recorded from 8144129 elements up while the mean bucket of the top ten
        // -- or else eleven -- bits leaves 3 % of room in the bucket capacity, at most 2048 tiles of 32768 keys): its per-tile counts
        // (16 bits per tile and bucket) and its bucket table sit in front of the status regions of whatever passes are
        // recorded behind it, and all of it must fit at every alignment
        if (g.keysPerThread == 32 && n >= 8144129u && vrdx::RoundUp(n, 32768u) <= 2048u) {  // (keys-only and key+value from 8.14 M)
          uint32_t bits = 0;
          for (uint32_t b = 10; b <= 11 && bits == 0; ++b)
            if ((((uint64_t)n + (1u << b) - 1) >> b) * 103 / 100 <= 36864) bits = b;
          // (equal tiles filling whole rounds of one workgroup per CU: up to rounds x cus of them; keys-only sorts by ten bits
          // fill whole rounds of PAIRS of tiles: smaller tiles, a larger table)
          for (int two = 0; two < (bits == 10 ? 2 : 1); ++two) {
          const uint32_t msdTileKeys = vrdx::MsdTileKeysFor(n, cus, 2048u, two != 0);
          const uint64_t msdTiles = vrdx::RoundUp(n, msdTileKeys);
          ok = ok && msdTileKeys % 4096u == 0 && msdTileKeys >= 4096u && msdTileKeys <= 32768u && msdTiles <= 2048u;
          for (uint32_t address = 0; bits != 0 && address < 128; address += 16) {
            const vrdx::StorageLayout lm = vrdx::MakeLayout(n, VRDX_STORAGE_ALIGN, plan.tiles, 0x7f0000001000ull + address,
                                                            plan.blockSums, bits, msdTiles);
            ++cases;
            // bucket sizes (inside the prefix the fill zeroes) and bucket bases on the first line behind the table, the per-tile
            // counts behind them, status region 0 behind those
            ok = ok && vrdx::LayoutFits(lm, n) && (address + lm.msdBucketOffset) % 128 == 0 && lm.msdBucketOffset >= 16 + 4096 &&
                 lm.msdBucketOffset < 16 + 4096 + 128 && lm.clearBytes == lm.msdBucketOffset + ((uint64_t)4 << bits);
            ok = ok && lm.msdCountsOffset == lm.msdBucketOffset + ((uint64_t)8 << bits) && lm.msdCountsBytes == msdTiles * ((uint64_t)2 << bits);
            ok = ok && lm.statusOffset == lm.msdCountsOffset + lm.msdCountsBytes && (address + lm.statusOffset) % 128 == 0;
            ok = ok && lm.statusClearOffset == lm.statusOffset && lm.statusClearBytes == lm.regionBytes;
            ok = ok && (address + lm.inoutOffset) % 128 == 0 && lm.valuesOffset >= lm.inoutOffset + (uint64_t)n * 4;
          }
          }
        }
        for (uint32_t address = 0; address < 128; address += 16) {
          if (!allAlignments && address != 0 && address != 32 && address != 112) continue;
          const vrdx::StorageLayout l =
              vrdx::MakeLayout(n, VRDX_STORAGE_ALIGN, plan.tiles, 0x7f0000001000ull + address, plan.blockSums);
          ++cases;
          // block sums: sorts of one round of 64 ... cus tiles of 32768 keys and more; one more row per 32 tiles
          ok = ok && l.blockRows == (plan.blockSums ? (plan.tiles + 31) / 32 : 0);
          ok = ok && (!plan.blockSums || (plan.tiles >= 64 && plan.tiles <= cus && plan.fullTiles == ~0u));
          ok = ok && l.regionBytes == (l.statusRows + l.blockRows) * 1024;
          const uint64_t region1End = l.statusOffset + 2 * l.regionBytes;
          const uint64_t inoutBytes = (uint64_t)n * 4;
          ok = ok && l.keysOnlySize == vrdx_oracle_storage_size(n, VRDX_STORAGE_ALIGN, 0);
          ok = ok && l.keyValueSize == vrdx_oracle_storage_size(n, VRDX_STORAGE_ALIGN, 1);
          ok = ok && l.histogramOffset == 16 && l.statusOffset >= 16 + 4096 && l.statusOffset < 16 + 4096 + 128;
          ok = ok && l.statusRows == (plan.tiles > 0 ? plan.tiles - 1 : 0);
          ok = ok && l.clearBytes == 16 + 4096 && l.statusClearBytes == l.regionBytes;
          ok = ok && (address + l.statusOffset) % 128 == 0;            // status rows start on a line
          ok = ok && l.ticketOffset == region1End;                     // the ticket line: its own, right behind them
          ok = ok && l.inoutOffset == l.ticketOffset + 128;
          ok = ok && (address + l.inoutOffset) % 128 == 0 && (address + l.valuesOffset) % 128 == 0;
          ok = ok && l.valuesOffset >= l.inoutOffset + inoutBytes;     // the scratch arrays do not overlap
          if (n > 0) {  // (an empty sort touches nothing: vrdx_api.cpp returns before it looks at the layout)
            ok = ok && l.inoutOffset + inoutBytes <= l.keysOnlySize;   // ... and end inside what the caller allocated
            ok = ok && l.valuesOffset + inoutBytes <= l.keyValueSize;
          }
        }
        if (!ok) {
          if (failures < 10)
            std::printf("FAIL n=%u geometry %ux%ux%u split=%d cus=%u: tiles %u slots %u fullTiles %u tailSlots %u\n", n, g.threads,
                        g.keysPerThread, g.subTiles, (int)g.splitForms, cus, plan.tiles, plan.slots, plan.fullTiles, plan.tailSlots);
          ++failures;
        }
      }
    }
  };
  for (uint32_t n = 0; n <= 70000; ++n) check(n, n % 97 == 0);
  for (uint64_t n = 70001; n <= VRDX_MAX_ELEMENTS; n += 1 + n / 977) check((uint32_t)n, false);
  for (uint32_t lg = 10; lg < 30; ++lg)
    for (int d = -2; d <= 2; ++d) check((1u << lg) + d, true);
  // the edges of the rounds: multiples of one round of every geometry and CU count, +- a few keys and +- one granule
  for (uint32_t cus : cuCounts)
    for (uint32_t capacity : {32768u, 65536u})
      for (uint32_t rounds = 1; rounds <= 6; ++rounds)
        for (int d : {-4097, -4096, -1, 0, 1, 2, 4095, 4096, 4097, 8192, 8193})
          check((uint32_t)((int64_t)rounds * cus * capacity + d), true);
  check(VRDX_MAX_ELEMENTS, true);
  // the MSD plan's range: its first size, where it goes from ten to eleven bits, its last size, tile edges
  for (uint32_t n : {8144129u, 8388608u, 16252929u, 16252930u, 35790291u, 35790292u, 36651000u, 67108863u, 67108864u, 67108865u})
    for (int d : {-32769, -32768, -1, 0, 1, 32767, 32768})
      check((uint32_t)((int64_t)n + d), true);
  std::printf("layout: %llu cases, %d failures\n", (unsigned long long)cases, failures);
  return failures != 0;
}

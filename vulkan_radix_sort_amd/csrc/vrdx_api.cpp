// Host recorder of the HIP backend: the eight vrdx* entry points of include/vk_radix_sort.h.
//
// Mirrors the reference's host side (src/vk_radix_sort.h.in:141-507) in behaviour:
//   * vrdxCreateSorter builds an immutable sorter and is the only call that can fail;
//   * the storage calculators are the reference's integer formulas, bit for bit;
//   * vrdxCmdSort* validate nothing, allocate nothing, never block the host: they append
//     stream-ordered work to the hipStream_t passed as VkCommandBuffer, the way gpuSort()
//     (:344-507) appends commands to a VkCommandBuffer.
// What differs is the recorded work: 1 clear + 1 fused histogram + 4 onesweep launches (+ the one or two launches of a
// hybrid plan for mid-size sorts) instead of 2 transfers + 12 dispatches + 12 barriers.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/vk_radix_sort.h"
#include "vrdx_kernels.h"
#include "vrdx_layout.h"

struct VrdxSorter_T {
  int device = 0;
  int computeUnits = 0;
  // (tile geometry is chosen per sort from the element count, see ConfigIndex)
  // LDS returning atomics proven lane-ordered on this device (at creation; vrdxHipRecheck may revise it)
  std::atomic<bool> atomicRank{false};
  // sorts recorded on the general path: behind every 65536th one a small, stream-ordered repeat of the lane-order check
  // is recorded (bit 1 of the sticky word) -- the property is not in the ISA manual, a driver or firmware update under a
  // long-lived process must not turn into silently unstable sorts
  mutable std::atomic<uint32_t> sortsRecorded{0};
  // One device word owned by the sorter: kernels OR their failure bit into it and nothing but
  // vrdxHipReadSorterStatus clears it, so a caller that reuses ONE storage buffer for many sorts (each
  // of which clears the storage's own failure word) still learns about a failure in any of them.
  uint32_t* stickyStatus = nullptr;
  // Second device word, right behind it: MSD plans the DEVICE turned down (a bucket beyond the capacity, a key outside the
  // sampled prefix, a sample that ruled the plan out) -- those sorts ran the four passes recorded behind the plan.  Next to
  // it the number of sorts recorded with the plan in front (host side).  vrdxHipReadPlanCounters.
  uint32_t* declinedPlans = nullptr;
  mutable std::atomic<uint32_t> plansRecorded{0};
  // Host side of the same diagnosis: set when an enqueue of a sort (fill, copy or kernel launch) was
  // refused by the runtime -- the entry points return void, so this is the only place it can go.
  // Reported as bit 31 by vrdxHipReadSorterStatus, which clears it.
  mutable std::atomic<uint32_t> enqueueFailed{0};
  // A sort was recorded with more than VRDX_MAX_ELEMENTS elements and clamped (bit 30 of vrdxHipReadSorterStatus).
  mutable std::atomic<uint32_t> countClamped{0};
  // reference: VrdxSorter_T::minStorageBufferOffsetAlignment (src/vk_radix_sort.h.in:134)
  uint32_t minStorageBufferOffsetAlignment = VRDX_STORAGE_ALIGN;
};

struct VrdxHipQueryPool {
  uint32_t count = 0;
  hipEvent_t* events = nullptr;
  uint8_t* recorded = nullptr;
  // source[slot]: the slot whose event holds this slot's time.  Slots the sort writes back to back, with
  // no device work between them, share ONE event record (an event record costs the stream ~3 us).
  uint32_t* source = nullptr;
};

namespace {

int DeviceOrdinalFromHandle(const void* handle, int* ordinal) {
  if (handle == nullptr) return hipGetDevice(ordinal) == hipSuccess ? 0 : -1;
  *ordinal = (int)((uintptr_t)handle - 1);
  return 0;
}

// "1024x32", or "1024x32x2" for the two-sub-tile kernel
void ConfigName(const vrdx::TileConfig& c, char* out, size_t size) {
  if (c.subTiles == 1)
    std::snprintf(out, size, "%dx%d", c.threads, c.keysPerThread);
  else
    std::snprintf(out, size, "%dx%dx%d", c.threads, c.keysPerThread, c.subTiles);
}

int ForcedConfigIndex() {
  static const int forced = [] {
    const char* env = std::getenv("VRDX_TILE_CONFIG");  // e.g. "512x16": one geometry for everything (tuning/testing)
    if (env == nullptr) return -1;
    for (int i = 0; i < vrdx::kNumTileConfigs; ++i) {
      char name[32];
      ConfigName(vrdx::kTileConfigs[i], name, sizeof(name));
      if (std::strcmp(env, name) == 0) return i;
    }
    std::fprintf(stderr, "vrdx-hip: unknown VRDX_TILE_CONFIG '%s', using the defaults\n", env);
    return -1;
  }();
  return forced;
}

// Tile geometry by problem size, measured on MI355X (tools: `vrdx_selftest sweep`, tables in
// profiles/r01_sweep_*.txt).  Three regimes:
//  * small sorts want many small tiles (parallelism across the CUs; six launches cost ~45 us);
//  * beyond that throughput grows with the tile (fewer look-backs per key, longer digit runs) up to
//    the 32768 keys whose staging buffer fits the CU's LDS ONCE -- so these tiles run one workgroup
//    per CU in lock-step ROUNDS of computeUnits tiles, and a sort whose tile count is just above a
//    multiple of the CU count pays for a whole extra round.  f below is the size in such rounds.
//    (16384-key tiles, two workgroups per CU, degrade gracefully in a partial round and used to win
//    just past the round boundaries; on the final kernels they no longer do);
//  * the two-sub-tile kernel (65536 keys, keys-only) halves the rounds again: best when f is in
//    (1, 2], just below 4 or 6.
enum : int { kCfg1024x8 = 0, kCfg1024x16 = 1, kCfg1024x32 = 2, kCfg1024x32x2 = 3 };

// msd: the MSD plan is recorded in front of the passes, which are then only the fallback for skewed keys; its per-tile counts
// take a quarter to a half of the reference's partition-histogram area, so the passes must not take tiles of 16384.
int ConfigIndex(const VrdxSorter_T* sorter, bool keyValue, uint32_t elementCount, bool atomicRank, bool msd = false) {
  const int forced = ForcedConfigIndex();
  // (the two-sub-tile kernel is keys-only: a key+value sort under a forced 1024x32x2 takes 1024x32)
  if (forced >= 0) return forced == kCfg1024x32x2 && (keyValue || !atomicRank) ? kCfg1024x32 : forced;
  const double f = (double)elementCount / ((double)sorter->computeUnits * 32768.0);
  if (keyValue) {
    if (f <= 0.26) return kCfg1024x8;
    if (f <= 0.53) return kCfg1024x16;
    // just past one round of 32768-element tiles, two workgroups of 16384 per CU fill the second round's gap
    // (1.2-2.6 % at 1.07 <= f <= 1.32, profiles/r03_sweep_by_geometry.txt; still so with the tail split of round 4,
    // profiles/r04_tail_split_kv.txt)
    if (f > 1.0 && f <= 1.35 && !msd) return kCfg1024x16;
    return kCfg1024x32;
  }
  // Behind the MSD plan the passes are the fallback only, and the plan's own launches double as its first two (one kernel, two
  // roles): those fused kernels exist for the two-sub-tile geometry, which is then taken at every size (at one round of tiles
  // and below as even-split tiles of two half-size sub-tiles).
  if (msd && atomicRank) return kCfg1024x32x2;
  if (f <= 0.125) return kCfg1024x8;
  if (f <= 0.5) return kCfg1024x16;   // beyond: even-split 1024x32 tiles (PlanTiles), 7 % faster at f = 0.536
  if (f <= 1.0) return kCfg1024x32;
  // Beyond one round the two-sub-tile kernel (65536 keys per workgroup: half the look-backs per key), whose last,
  // partial round is cut into small equal tiles (tail split, PlanTiles): with that it is the fastest geometry at every
  // size from one round up (profiles/r04_tail_split_keys.txt; without it, it lost a whole 65536-key round to the
  // 1024x32 tiles whenever the tile count passed a multiple of the CU count -- f in (2, 3.3] and beyond 4 in round 3).
  // It holds two sub-tiles' keys in registers: only with the one-atomic ranking.
  return atomicRank ? kCfg1024x32x2 : kCfg1024x32;
}

#ifdef VRDX_TRACE
// tools/trace.sh only: one device buffer of 8 stamps per (pass, tile), dumped to $VRDX_TRACE_FILE
// by vrdxDestroySorter.  Holds the LAST sort recorded before the dump.
unsigned long long* g_trace = nullptr;
uint32_t g_traceTiles = 0;
constexpr uint32_t kTraceMaxTiles = 1u << 16;
unsigned long long* TraceBuffer(uint32_t pass, uint32_t tiles) {
  if (g_trace == nullptr) {
    if (hipMalloc(reinterpret_cast<void**>(&g_trace), 4ull * kTraceMaxTiles * 8 * sizeof(unsigned long long)) !=
        hipSuccess)
      return nullptr;
  }
  if (tiles > kTraceMaxTiles) return nullptr;
  g_traceTiles = tiles;
  return g_trace + (size_t)pass * kTraceMaxTiles * 8;
}
void DumpTrace() {
  const char* path = std::getenv("VRDX_TRACE_FILE");
  if (g_trace == nullptr || path == nullptr) return;
  (void)hipDeviceSynchronize();
  const size_t words = 4ull * kTraceMaxTiles * 8;
  unsigned long long* host = new unsigned long long[words];
  if (hipMemcpy(host, g_trace, words * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
    if (FILE* f = std::fopen(path, "wb")) {
      std::fwrite(&g_traceTiles, sizeof(g_traceTiles), 1, f);
      for (uint32_t pass = 0; pass < 4; ++pass)
        std::fwrite(host + (size_t)pass * kTraceMaxTiles * 8, sizeof(unsigned long long), (size_t)g_traceTiles * 8, f);
      std::fclose(f);
    }
  }
  delete[] host;
}
#endif

// Integer environment knob for the tuning scripts (read once); -1 when unset.
int TuningKnob(const char* name) {
  const char* env = std::getenv(name);
  return env != nullptr ? std::atoi(env) : -1;
}

// Mid-size sorts record the hybrid plan (vrdx_kernels.hip, PassPlan) next to the four passes: launch 0 scatters by the
// keys' highest byte that varies and bucket_sort_kernel finishes every bucket inside one workgroup -- if the DEVICE finds that no bucket
// exceeds the capacity returned here; otherwise the four passes run as usual and the bucket launch is empty.  The
// capacity is the smallest of 4096 / 8192 / 16384 / 32768 (the last one with the one-atomic ranking only) that leaves a
// bucket twice the room of its mean N / 256; the largest one is recorded as long as it leaves 3 % (a bucket sort
// costs what the bucket's elements cost, whatever the capacity; uniform keys spread by half a percent at these sizes --
// mean 31800, sigma 178 at 8.1 M: the capacity is 5 sigma away -- and a plan that does not apply costs one empty
// launch, 3 us, where one that does saves 17-25 %): N <= 8.1 M elements (4.0 M with the ballot ranking).
// 0 = the plan is not recorded (larger N, a forced tile geometry, VRDX_HYBRID=0).
uint32_t HybridCapacity(bool atomicRank, uint32_t elementCount) {
  static const bool enabled = [] {
    const char* env = std::getenv("VRDX_HYBRID");  // "0": always the four-pass plan (testing / measurements)
    return env == nullptr || env[0] != '0';
  }();
  if (!enabled || elementCount <= vrdx::kSmallSortMaxElements) return 0;
  // percent of the mean bucket a bucket may hold (tools: VRDX_HYBRID_HEADROOM); the LARGEST capacity is tried with
  // less room than the others: failing costs one empty launch, the plan is worth a fifth to a third of the sort
  static const int knob = TuningKnob("VRDX_HYBRID_HEADROOM");
  static const int knobLast = TuningKnob("VRDX_HYBRID_HEADROOM_LAST");
  const uint64_t mean = (elementCount + VRDX_RADIX - 1) / VRDX_RADIX;
  const uint32_t need = (uint32_t)(mean * (uint64_t)(knob > 0 ? knob : 200) / 100u);
  const uint32_t needLast = (uint32_t)(mean * (uint64_t)(knobLast > 0 ? knobLast : 103) / 100u);
  // 32768-element buckets: the one-atomic ranking only (the ballot forms of that kernel would spill); key+value stages
  // keys and values through one buffer there (SharedStage in vrdx_kernels.hip)
  const uint32_t largest = atomicRank ? 32768u : 16384u;
  if (need <= 4096u) return 4096u;
  if (need <= 8192u) return 8192u;
  if (need <= 16384u) return 16384u;
  if (need <= largest) return largest;
  return needLast <= largest ? largest : 0u;
}

// The MSD plan (vrdx_kernels.hip, "MSD plan"): one stable scatter by the keys' top 10 or 11 bits, then every bucket by its
// remaining bits in two passes inside one workgroup -- three ranking steps and two trips through memory instead of four
// and four.  Recorded, in front of the four passes (which return on its verdict), for sorts beyond the eight-bit
// plan's reach whose mean bucket leaves 3 % of room in the bucket kernel's capacity (uniform keys spread by half a percent
// at these sizes): ten bits up to 36.6 M keys / 32.5 M pairs, eleven bits up to twice that.  Returns the bits or 0.
// One-atomic ranking only.  VRDX_MSD=0 switches it off (VRDX_HYBRID=0 and a forced tile geometry as well); VRDX_MSD_FROM=n
// records it from n elements up instead (measurements: below its default range it replaces the other two plans).
// Keys per tile of the MSD plan's histogram and scatter: equal tiles that fill whole rounds of one workgroup per CU
// (vrdx_layout.h).  VRDX_MSD_EVEN=0: tiles of full capacity (measurements).
// twoPerWorkgroup: keys-only sorts by ten bits (their scatter takes two consecutive tiles per workgroup, vrdx_kernels.hip)
uint32_t MsdTileKeys(uint32_t elementCount, uint32_t cus, bool twoPerWorkgroup) {
  static const int even = TuningKnob("VRDX_MSD_EVEN");
  return even == 0 ? vrdx::kMsdTileKeys : vrdx::MsdTileKeysFor(elementCount, cus, vrdx::kMsdMaxTiles, twoPerWorkgroup);
}

uint32_t MsdBits(bool atomicRank, bool keyValue, uint32_t elementCount, uint32_t hybridCap, uint32_t* capacity) {
  static const bool enabled = [] {
    const char* all = std::getenv("VRDX_HYBRID");
    const char* msd = std::getenv("VRDX_MSD");
    return (all == nullptr || all[0] != '0') && (msd == nullptr || msd[0] != '0');
  }();
  static const int from = TuningKnob("VRDX_MSD_FROM");
  static const int forcedBits = TuningKnob("VRDX_MSD_BITS");  // measurements: 10 | 11 wherever the capacity allows
  static const int knobLast = TuningKnob("VRDX_HYBRID_HEADROOM_LAST");
  if (!enabled || !atomicRank) return 0;
  // From where the EIGHT-bit plan ends (8.1 M: hybridCap == 0), keys-only and key+value.  Up to 18.1 M elements the buckets
  // hold at most 18432 and the half-size bucket kernel sorts them, two workgroups to a CU: with it the plan is 8-15 % faster
  // than round 4's nine-bit hybrid plan and the four passes at one round of tiles, which key+value sorts of these sizes
  // took before (profiles/r05_msd_half_buckets.txt); that plan's kernels are gone since.
  const uint32_t lowest = from > 0 ? (uint32_t)from : (hybridCap == 0 ? vrdx::kSmallSortMaxElements + 1u : ~0u);
  if (elementCount < lowest || vrdx::RoundUp(elementCount, vrdx::kMsdTileKeys) > vrdx::kMsdMaxTiles) return 0;
  const uint32_t cap = keyValue ? vrdx::kMsdCapKeyValue : vrdx::kMsdCapKeys;
  *capacity = cap;
  for (uint32_t bits = 10; bits <= 11; ++bits) {
    if (forcedBits > 0 && (uint32_t)forcedBits != bits) continue;
    const uint64_t mean = ((uint64_t)elementCount + (1u << bits) - 1u) >> bits;
    if (mean * (uint64_t)(knobLast > 0 ? knobLast : 103) / 100u <= cap) {
      // buckets of half the size: the bucket kernel of 512 threads, two workgroups per CU (bucket_sort2_half_kernel)
      static const int half = TuningKnob("VRDX_MSD_HALF");
      // (4 % of headroom here: 5.3 sigma of a uniform bucket of 17700; the 3 % of the full size would be 4 sigma at this
      // capacity, and with 1024 buckets one sort in thirty at the top of the range would be turned down)
      if (bits == 10 && half != 0 && mean * (uint64_t)(knobLast > 0 ? knobLast : 104) / 100u <= vrdx::kMsdHalfCap)
        *capacity = vrdx::kMsdHalfCap;
      return bits;
    }
  }
  return 0;
}

bool SmallSortEnabled() {
  static const bool enabled = [] {
    const char* env = std::getenv("VRDX_SMALL_SORT");  // "0": always take the general path (testing)
    return env == nullptr || env[0] != '0';
  }();
  return enabled;
}

inline uint8_t* BufferAddress(VkBuffer buffer, VkDeviceSize offset) {
  return reinterpret_cast<uint8_t*>(buffer) + offset;
}

// VRDX_DEBUG=1: report HIP errors of the enqueues on stderr (the entry points themselves return
// void and validate nothing, like the reference's vrdxCmd*).
bool DebugEnabled() {
  static const bool enabled = std::getenv("VRDX_DEBUG") != nullptr;
  return enabled;
}
// After every enqueue of a sort: a launch / fill / copy the runtime REFUSED is latched in the sorter (and
// printed under VRDX_DEBUG).  Only the value returned by that very call is looked at -- never the calling
// thread's last-error state, which is sticky on ROCm and may hold an unrelated, older failure of the caller's.
void EnqueueCheck(const VrdxSorter_T* sorter, const char* what, hipError_t returned) {
#ifdef VRDX_TESTING
  // test build only (tests/enqueue_error_check.py builds it): every check reports a refusal, the work itself is
  // enqueued as usual
  static const bool inject = std::getenv("VRDX_TEST_INJECT_ENQUEUE_ERROR") != nullptr;
  if (inject) returned = hipErrorUnknown;
#endif
  if (returned == hipSuccess) return;
  sorter->enqueueFailed.store(1u, std::memory_order_relaxed);
  if (DebugEnabled()) std::fprintf(stderr, "vrdx-hip: %s -> %s\n", what, hipGetErrorString(returned));
}

void Stamp(VrdxHipQueryPool* pool, uint32_t slot, hipStream_t stream) {
  if (pool == nullptr || slot >= pool->count) return;
  if (hipEventRecord(pool->events[slot], stream) == hipSuccess) {
    pool->recorded[slot] = 1;
    pool->source[slot] = slot;
  }
}
// The same point of the stream as slot `same` (stamped just before, nothing enqueued since): no second event.
void StampSame(VrdxHipQueryPool* pool, uint32_t slot, uint32_t same) {
  if (pool == nullptr || slot >= pool->count || same >= pool->count || !pool->recorded[same]) return;
  pool->recorded[slot] = 1;
  pool->source[slot] = pool->source[same];
}

// The tile plan of a sort (vrdx_layout.h, PlanTiles): even-split tiles for sorts of one round, tail-split tiles behind
// the whole rounds of a longer one -- where the kernels' forms with run-time slot counts exist and where they were
// measured to pay (profiles/r04_tail_split_keys.txt, r04_tail_split_kv.txt; f = size in rounds of CUs x 32768):
//   keys-only 1024x32x2   even split (-9.5 % at f = 1.07) and tail split at every size (f = 2.06: 0.191 instead of
//                         0.230 ms; 3.06: 0.261 / 0.272; 4.06: 0.340 / 0.388)
//   keys-only 1024x32     even split (-3.5 % at f = 0.5); NO tail split (+0 ... +4 %: these tiles are short enough that
//                         a few of them in a last round cost what 256 small ones cost)
//   key+value 1024x32     NO even split (its split form fetches the values late, vrdx_kernels.hip: +2 ... +7 % at
//                         0.55 < f < 1); tail split while the rest is at most half a round (f = 1.06: 0.184 / 0.193 ms,
//                         2.06: 0.294 / 0.303, 3.06: 0.399 / 0.408, 4.06: 0.524 / 0.532; beyond half a round -1 ... +4 %)
// VRDX_EVEN_SPLIT=0 / VRDX_TAIL_SPLIT=0 turn them off, VRDX_EVEN_SPLIT=1 / VRDX_TAIL_SPLIT=p (percent of a round) force
// them wherever the forms exist (measurements).
vrdx::TilePlan PlanTiles(const VrdxSorter_T* sorter, int configIndex, bool keyValue, uint32_t elementCount, bool atomicRank) {
  static const int evenKnob = TuningKnob("VRDX_EVEN_SPLIT");
  static const int tailKnob = TuningKnob("VRDX_TAIL_SPLIT");
  const vrdx::TileConfig& c = vrdx::kTileConfigs[configIndex];
  const bool pair = configIndex == kCfg1024x32x2;
  const bool splitForms = pair ? (!keyValue && atomicRank) : configIndex == kCfg1024x32;
  const bool evenSplit = evenKnob >= 0 ? evenKnob != 0 : !keyValue;
  const uint32_t tailPercent = tailKnob >= 0 ? (uint32_t)tailKnob : (pair ? 100u : (keyValue ? 50u : 0u));
  return vrdx::PlanTiles(elementCount, (uint32_t)sorter->computeUnits, (uint32_t)c.threads, (uint32_t)c.keysPerThread,
                         (uint32_t)c.subTiles, splitForms, evenSplit, tailPercent);
}

// Everything the host decides about a sort, in ONE place: RecordSort records it and vrdxHipDescribePlan reports it.
// storageAddress: the absolute address the storage is handed over at -- only its low seven bits matter (the pads in front
// of the 128-byte aligned regions); 0 is the worst case for what fits, which is what vrdxHipDescribePlan assumes.
struct SortPlan {
  bool atomicRank = false;
  bool oneWorkgroup = false;    // small_sort_kernel: one launch, no storage layout
  uint32_t hybridCap = 0;       // the eight-bit hybrid plan is recorded with this bucket capacity
  uint32_t msdBits = 0;         // the MSD plan is recorded in front of the passes (10 | 11)
  uint32_t msdCap = 0;
  uint32_t msdTileKeys = 0;
  uint32_t msdTiles = 0;
  uint32_t msdFused = 0;        // how many of the plan's launches double as the fallback's first passes (0 | 1 | 2)
  int configIndex = 0;
  vrdx::TilePlan tilePlan{};
  bool blockSums = false;
  bool fits = true;             // false: not even tiles of full capacity fit the caller's storage (refused)
  vrdx::StorageLayout layout{};
  uint32_t launches = 0;        // kernels + fills enqueued
};

SortPlan PlanSort(const VrdxSorter_T* sorter, bool keyValue, uint32_t elementCount, uint64_t storageAddress) {
  SortPlan p;
  p.atomicRank = sorter->atomicRank.load(std::memory_order_relaxed);  // one answer for the whole sort
  const bool adaptive = ForcedConfigIndex() < 0;
  if (elementCount == 0) return p;
  if (elementCount <= vrdx::kSmallSortMaxElements && adaptive && SmallSortEnabled()) {
    p.oneWorkgroup = true;
    p.launches = 1;
    p.layout = vrdx::MakeLayout(elementCount, sorter->minStorageBufferOffsetAlignment, 0, storageAddress);  // (the failure word)
    return p;
  }
  p.hybridCap = adaptive ? HybridCapacity(p.atomicRank, elementCount) : 0u;
  p.msdBits = adaptive ? MsdBits(p.atomicRank, keyValue, elementCount, p.hybridCap, &p.msdCap) : 0u;
  if (p.hybridCap != 0) p.msdBits = 0;  // (VRDX_MSD_FROM below the eight-bit plan's end: that plan keeps its sizes)
  p.configIndex = ConfigIndex(sorter, keyValue, elementCount, p.atomicRank, p.msdBits != 0);
  p.tilePlan = PlanTiles(sorter, p.configIndex, keyValue, elementCount, p.atomicRank);
  // Block sums instead of the look-back chain: sorts of one round (PlanTiles) on the four-pass plan -- with a hybrid
  // plan recorded, launch 0 may rank by another byte than its pass index, which the block-sum form does not look up.
  // VRDX_BLOCK_SUMS=0 keeps the classic look-back (measurements).
  static const int blockSumsKnob = TuningKnob("VRDX_BLOCK_SUMS");
  p.blockSums = p.tilePlan.blockSums && p.hybridCap == 0 && blockSumsKnob != 0;
  p.msdTileKeys = MsdTileKeys(elementCount, (uint32_t)sorter->computeUnits, !keyValue && p.msdBits == 10);
  p.msdTiles = vrdx::RoundUp(elementCount, p.msdTileKeys);
  const uint32_t align = sorter->minStorageBufferOffsetAlignment;
  p.layout = vrdx::MakeLayout(elementCount, align, p.tilePlan.tiles, storageAddress, p.blockSums, p.msdBits, p.msdTiles);
  if (p.msdBits != 0 && !vrdx::LayoutFits(p.layout, elementCount)) {
    // (cannot happen for the sizes MsdBits admits -- tests/native/layout_check.cpp sweeps them -- but the storage is the
    // caller's: without the plan's rows in front of the status regions the layout fits for every N)
    p.msdBits = 0;
    p.layout = vrdx::MakeLayout(elementCount, align, p.tilePlan.tiles, storageAddress, p.blockSums);
  }
  if (!vrdx::LayoutFits(p.layout, elementCount)) {
    // EVERY sort of the general path is checked, not only those with a plan in front: the layout depends on the tile plan, and the
    // measurement knobs (VRDX_TAIL_SPLIT, VRDX_EVEN_SPLIT, VRDX_TILE_CONFIG) can select plans the offline sweep of
    // tests/native/layout_check.cpp never saw.  Tiles of the kernel's full capacity without block rows fit for every N
    // (2 (tiles - 1) KiB <= (P - 1) KiB from 8192 keys per tile up); smaller tiles cannot be helped: the scratch arrays
    // must not leave the caller's allocation, so that sort is refused and says so (VRDX_HIP_STATUS_ENQUEUE_REFUSED).
    const vrdx::TileConfig& c = vrdx::kTileConfigs[p.configIndex];
    p.tilePlan = vrdx::PlanTiles(elementCount, (uint32_t)sorter->computeUnits, (uint32_t)c.threads, (uint32_t)c.keysPerThread,
                                 (uint32_t)c.subTiles, false, false, 0);
    p.blockSums = false;
    p.msdBits = 0;
    p.layout = vrdx::MakeLayout(elementCount, align, p.tilePlan.tiles, storageAddress, false, 0);
    p.fits = vrdx::LayoutFits(p.layout, elementCount);
  }
  // The MSD plan's scatter launch is ALSO pass 0 of the fallback and its bucket launch pass 1 (one branch on the verdict, on
  // the device): only passes 2 and 3 remain as launches that return when the plan runs.  The fused kernels exist for the
  // geometries the recorder selects at these sizes (ConfigIndex: the two-sub-tile kernel keys-only, 1024x32 key+value --
  // since round 6: vrdx_kernels.hip, msd_scatter_or_pass0_kernel); with buckets of the half-size kernel (512 threads; the
  // passes' bodies need 1024) only the scatter launch has a second role.  VRDX_MSD_FUSED=0: none (measurements).
  static const int fusedKnob = TuningKnob("VRDX_MSD_FUSED");
  if (p.msdBits != 0 && fusedKnob != 0 && p.configIndex == (keyValue ? kCfg1024x32 : kCfg1024x32x2))
    p.msdFused = p.msdCap == (keyValue ? vrdx::kMsdCapKeyValue : vrdx::kMsdCapKeys) ? 2u : 1u;
  // kernels: histogram + four passes (+ the eight-bit plan's bucket launch); the MSD plan: + spine, and those of its scatter
  // and bucket launches that are not also a pass
  p.launches = 1u + VRDX_PASSES + (p.msdBits != 0 ? 3u - p.msdFused : 0u) + (p.hybridCap != 0 ? 1u : 0u);
  return p;
}

// reference: gpuSort, src/vk_radix_sort.h.in:344-507
void RecordSort(VkCommandBuffer commandBuffer, VrdxSorter sorter, uint32_t elementCount,
                VkBuffer indirectBuffer, VkDeviceSize indirectOffset, VkBuffer keysBuffer,
                VkDeviceSize keysOffset, VkBuffer valuesBuffer, VkDeviceSize valuesOffset,
                VkBuffer storageBuffer, VkDeviceSize storageOffset, VkQueryPool queryPool,
                uint32_t query) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(commandBuffer);
  VrdxHipQueryPool* pool = reinterpret_cast<VrdxHipQueryPool*>(queryPool);
  const bool keyValue = valuesBuffer != nullptr;
  if (elementCount > VRDX_MAX_ELEMENTS) {
    // The reference's uint32 byte math wraps above 2^30 - 4 elements (src/vk_radix_sort.h.in:105-115): no storage
    // requirement exists for such a count.  The first 2^30 - 4 elements are sorted, the tail is left alone, and the
    // sorter's status word says so (VRDX_HIP_STATUS_COUNT_CLAMPED) -- like every other failure mode, never silently.
    elementCount = VRDX_MAX_ELEMENTS;
    sorter->countClamped.store(1u, std::memory_order_relaxed);
  }

  // Launches go to the sorter's device (a Vulkan command buffer belongs to one device too); the
  // calling thread's current device is put back afterwards.
  struct DeviceScope {
    int previous = -1;
    explicit DeviceScope(int wanted) {
      int current = -1;
      if (hipGetDevice(&current) == hipSuccess && current != wanted && hipSetDevice(wanted) == hipSuccess)
        previous = current;
    }
    ~DeviceScope() {
      if (previous >= 0) (void)hipSetDevice(previous);
    }
  } deviceScope(sorter->device);

  uint8_t* const storage = BufferAddress(storageBuffer, storageOffset);
  const SortPlan plan = PlanSort(sorter, keyValue, elementCount, (uint64_t)reinterpret_cast<uintptr_t>(storage));
  const bool atomicRank = plan.atomicRank;
  const uint32_t hybridCap = plan.hybridCap, msdBits = plan.msdBits, msdCap = plan.msdCap;
  const int configIndex = plan.configIndex;
  const vrdx::TilePlan& tilePlan = plan.tilePlan;
  const bool blockSums = plan.blockSums;
  const uint32_t msdTileKeys = plan.msdTileKeys, msdTiles = plan.msdTiles;
  const vrdx::StorageLayout& layout = plan.layout;
  uint32_t* const keys = reinterpret_cast<uint32_t*>(BufferAddress(keysBuffer, keysOffset));
  uint32_t* const values =
      keyValue ? reinterpret_cast<uint32_t*>(BufferAddress(valuesBuffer, valuesOffset)) : nullptr;
  const uint32_t* const countPtr =
      indirectBuffer != nullptr
          ? reinterpret_cast<const uint32_t*>(BufferAddress(indirectBuffer, indirectOffset))
          : nullptr;

  Stamp(pool, query + 0, stream);

  if (elementCount == 0) {
    // reference: zero partitions -> every dispatch is empty (:353,448,465,487)
    for (uint32_t s = 1; s < 15; ++s) StampSame(pool, query + s, query + 0);
    return;
  }

  // Small sorts: one workgroup, one launch, nothing but the caller's keys / values and the failure
  // word touched (the
  // general path below costs six launches = 30-45 us however small N is).  Forcing a tile geometry
  // (VRDX_TILE_CONFIG) also forces the general path, which is how the tests reach it at small sizes.
  if (plan.oneWorkgroup) {
    for (uint32_t s = 1; s < 14; ++s) StampSame(pool, query + s, query + 0);
    EnqueueCheck(sorter, "small_sort_kernel",
                 vrdx::LaunchSmallSort(stream, atomicRank, keys, values, elementCount, countPtr,
                                       reinterpret_cast<uint32_t*>(storage + layout.failureOffset)));
    Stamp(pool, query + 14, stream);
    return;
  }

  // (behind the empty sort and the one-workgroup sort above: neither touches the scratch arrays)
  if (!plan.fits) {
    // every slot of the timestamp contract is recorded, like in the empty sort: a caller that reads the pool after a
    // refused sort must not wait on events that never were
    for (uint32_t s = 1; s < 15; ++s) StampSame(pool, query + s, query + 0);
    EnqueueCheck(sorter, "storage layout (status rows do not fit the reference's partition-histogram area)", hipErrorInvalidValue);
    return;
  }
  uint32_t* const globalHistogram = reinterpret_cast<uint32_t*>(storage + layout.histogramOffset);
  uint32_t* const status = reinterpret_cast<uint32_t*>(storage + layout.statusOffset);
  uint32_t* const tickets = reinterpret_cast<uint32_t*>(storage + layout.ticketOffset);
  uint32_t* const failure = reinterpret_cast<uint32_t*>(storage + layout.failureOffset);
  uint32_t* const keysScratch = reinterpret_cast<uint32_t*>(storage + layout.inoutOffset);
  uint32_t* const valuesScratch = reinterpret_cast<uint32_t*>(storage + layout.valuesOffset);
  const uint32_t statusRows = (uint32_t)layout.statusRows;

  // the MSD plan's arguments: spine (prefixes over the tiles, bucket table, verdict), scatter by the window bits, one
  // workgroup per bucket; the histogram kernel takes the same structure
  vrdx::MsdArgs m;
  std::memset(&m, 0, sizeof(m));
  if (msdBits != 0) {
    m.keysCaller = keys;
    m.keysScratch = keysScratch;
    m.valuesCaller = keyValue ? values : nullptr;
    m.valuesScratch = keyValue ? valuesScratch : nullptr;
    m.maxCount = elementCount;
    m.countPtr = countPtr;
    m.histogramTable = globalHistogram;
    m.tileCounts = reinterpret_cast<uint32_t*>(storage + layout.msdCountsOffset);
    m.bucketCount = reinterpret_cast<uint32_t*>(storage + layout.msdBucketOffset);
    m.bucketBase = m.bucketCount + ((size_t)1 << msdBits);
    m.overflowWord = reinterpret_cast<uint32_t*>(storage + VRDX_OFF_MSD_OVERFLOW);
    m.planWord = reinterpret_cast<uint32_t*>(storage + VRDX_OFF_PLAN);
    m.bits = msdBits;
    m.cap = msdCap;
    m.tiles = msdTiles;
    m.tileKeys = msdTileKeys;
    m.statusClear = storage + layout.statusClearOffset;
    m.statusVecs = (uint32_t)(layout.statusClearBytes / 16u);
    m.tickets = tickets;
    m.declinedPlans = sorter->declinedPlans;
    sorter->plansRecorded.fetch_add(1u, std::memory_order_relaxed);
  }

  // Clear count / plan / failure word and the 4x256 global histogram (reference :382) in one fill of 4112 bytes; status
  // region 0 is zeroed by the histogram kernel behind it.  Indirect: also copy the device-side count to where the reference keeps
  // it (:368-379); the kernels themselves read it straight from the caller's buffer.  (Direct: the
  // count travels as a kernel argument, the slot stays 0 -- storage contents are scratch.)
  // (With the MSD plan recorded the fill also covers the plan's bucket sizes, 4-8 KiB behind the table: its histogram kernel
  // adds them up.)
  EnqueueCheck(sorter, "hipMemsetAsync(state)", hipMemsetAsync(storage, 0, layout.clearBytes, stream));
  if (countPtr != nullptr)
    EnqueueCheck(sorter, "hipMemcpyAsync(count)",
                 hipMemcpyAsync(storage + layout.countOffset, countPtr, sizeof(uint32_t), hipMemcpyDeviceToDevice,
                                stream));
  Stamp(pool, query + 1, stream);

  // upsweep of all four passes at once
  {
    // Every workgroup ends with up to 1024 global atomics on the same 1024 words, so few, long-lived
    // workgroups win for large inputs: one per CU and at least two groups of 16384 keys each (tools/hist_grid.sh:
    // 17.4 us with 256 workgroups against 21.1 us with 512 at N = 2^23; equal at 2^25).  Small inputs
    // want the opposite -- the kernel is one memory latency long, so up to 128 workgroups of at least
    // 4096 keys share it: 6.9 instead of 9.7 us at 2^18, 7.9 instead of 9.8 us at 2^20, same at 2^22.
    uint32_t grid = vrdx::RoundUp(elementCount, 2 * vrdx::kHistGroupKeys);
    const uint32_t wide = std::min<uint32_t>(128u, vrdx::RoundUp(elementCount, 4096u));
    if (grid < wide) grid = wide;
    const uint32_t cap = (uint32_t)sorter->computeUnits * vrdx::kHistWorkgroupsPerCu;
    if (grid > cap) grid = cap;
    if (grid == 0) grid = 1;
    static const int forcedGrid = TuningKnob("VRDX_HIST_GRID");  // tools/hist_grid.sh
    if (forcedGrid > 0) grid = (uint32_t)forcedGrid;
    if (msdBits != 0) {
      // the MSD plan's form: window-bits counts per tile of up to 32768 keys (a workgroup takes whole tiles); the spine
      // kernel clears status region 0
      if (forcedGrid <= 0) grid = std::min<uint32_t>(msdTiles, cap);
      EnqueueCheck(sorter, "histogram_msd_kernel", vrdx::LaunchHistogramMsd(stream, grid, m));
    } else {
      EnqueueCheck(sorter, "histogram_kernel",
                   vrdx::LaunchHistogram(stream, grid, keys, elementCount, countPtr, globalHistogram, tickets,
                                         storage + layout.statusClearOffset, (uint32_t)layout.statusClearBytes));
    }
  }

  const uint32_t tiles = tilePlan.tiles;
  // Key+value tiles fetch their values early (right after the ranking: they land during the scan and the
  // regroup) -- on the final kernels that is as fast as or faster than fetching them after the
  // look-back at every size (0-8 %, vrdx_selftest sweep with VRDX_KV_EARLY_VALUES=0|1); the late form
  // stays selectable for measurements.
  bool earlyValues = true;
  static const int forcedEarly = TuningKnob("VRDX_KV_EARLY_VALUES");  // 0 | 1: tuning/testing
  if (forcedEarly >= 0) earlyValues = forcedEarly != 0;
  // the arguments of pass `pass` of the four passes (also handed to the MSD plan's launches, whose second role they are)
  auto passArgs = [&](uint32_t pass) {
    vrdx::OnesweepArgs args;
    // which pair of arrays the pass reads is settled on the device (vrdx_kernels.h); the reference
    // switches in->out to out->in for pass 1, pass 3 (:417-427) and so do four ranking passes here
    args.keysCaller = keys;
    args.keysScratch = keysScratch;
    args.valuesCaller = keyValue ? values : nullptr;
    args.valuesScratch = keyValue ? valuesScratch : nullptr;
    args.maxCount = elementCount;
    args.countPtr = countPtr;
    args.histogramTable = globalHistogram;
    // region r of the two: [statusRows tile rows][blockRows block rows]
    const size_t regionWords = (size_t)(layout.regionBytes / sizeof(uint32_t));
    uint32_t* const regionCur = status + (size_t)(pass & 1u) * regionWords;
    uint32_t* const regionNext = status + (size_t)((pass + 1) & 1u) * regionWords;
    args.statusCur = regionCur;
    args.statusNext = pass + 1 < VRDX_PASSES ? regionNext : nullptr;
    args.statusRows = statusRows;
    args.blockCur = blockSums ? regionCur + (size_t)statusRows * VRDX_RADIX : nullptr;
    args.blockNext = blockSums ? regionNext + (size_t)statusRows * VRDX_RADIX : nullptr;
    args.blockRows = blockSums ? (uint32_t)layout.blockRows : 0u;
    args.ticketCur = tickets + (pass & 1u);
    args.ticketNext = tickets + ((pass + 1) & 1u);
    args.failure = failure;
    args.stickyFailure = sorter->stickyStatus;
    args.pass = pass;
    args.hybridCap = hybridCap;
    args.planWord = reinterpret_cast<uint32_t*>(storage + VRDX_OFF_PLAN);
    args.spinLimit = vrdx::kSpinLimit;
#ifdef VRDX_TESTING
    // test build only: VRDX_TEST_SPIN_LIMIT=0 makes the first look-back trip that has to wait give up, which is how
    // tests/sticky_status_check.py sees the device-side failure path (failure word + the sorter's sticky word)
    static const int testSpinLimit = TuningKnob("VRDX_TEST_SPIN_LIMIT");
    if (testSpinLimit >= 0) args.spinLimit = (uint32_t)testSpinLimit;
#endif
    args.earlyValues = earlyValues ? 1u : 0u;
    args.planInFront = msdBits != 0 ? 1u : 0u;  // the MSD plan in front may have taken the sort (verdict 3)
    args.slots = tilePlan.slots;
    args.fullTiles = tilePlan.fullTiles;
    args.tailSlots = tilePlan.tailSlots;
    args.trace = nullptr;
#ifdef VRDX_TRACE
    args.trace = TraceBuffer(pass, tiles);
#endif
    return args;
  };
  const uint32_t msdFused = plan.msdFused;  // how many of the plan's launches double as the fallback's first passes (PlanSort)
  // The MSD plan, recorded in front of the passes: spine (prefixes over the tiles, bucket table,
  // verdict), scatter by the window bits, one workgroup per bucket.  The passes behind return on the verdict word.
  if (msdBits != 0) {
    // Timestamps: the plan's own three stages take the names they have in the reference -- slot 2 "upsweep" = the
    // histogram, 3 "spine", 4 "downsweep" = the scatter -- and the bucket sorts are pass 1's "upsweep" (slot 5, like the
    // eight-bit plan's); the four returning passes share the slots behind.
    Stamp(pool, query + 2, stream);
    EnqueueCheck(sorter, "spine_msd_kernel", vrdx::LaunchSpineMsd(stream, m));
    Stamp(pool, query + 3, stream);
    if (msdFused >= 1)
      EnqueueCheck(sorter, "msd_scatter_or_pass0_kernel", vrdx::LaunchMsdFused(stream, false, keyValue, m, passArgs(0), tilePlan.tiles));
    else
      EnqueueCheck(sorter, "scatter_msd_kernel", vrdx::LaunchScatterMsd(stream, keyValue, m));
    Stamp(pool, query + 4, stream);
    if (msdFused >= 2)
      EnqueueCheck(sorter, "msd_buckets_or_pass1_kernel", vrdx::LaunchMsdFused(stream, true, keyValue, m, passArgs(1), tilePlan.tiles));
    else
      EnqueueCheck(sorter, "bucket_sort2_kernel", vrdx::LaunchBucketSort2(stream, keyValue, m));
    Stamp(pool, query + 5, stream);
  }
  for (uint32_t pass = 0; pass < VRDX_PASSES; ++pass) {
    if (pass < msdFused) {  // ran (or returned) inside the plan's own launches
      if (pass == 1) {
        StampSame(pool, query + 6, query + 5);
        StampSame(pool, query + 7, query + 5);
      }
      continue;
    }
    // "upsweep" of this pass: the fused histogram kernel for pass 0, nothing for the others -- the same
    // point of the stream as the previous pass's "downsweep" stamp
    if (msdBits != 0) {
      // (slots 2-5 are the MSD plan's, above; launch 0 and launch 1 fall into slot 7)
      if (pass == 1) StampSame(pool, query + 6, query + 5);
      if (pass >= 2) {
        StampSame(pool, query + 2 + 3 * pass + 0, query + 2 + 3 * (pass - 1) + 2);
        StampSame(pool, query + 2 + 3 * pass + 1, query + 2 + 3 * pass + 0);
      }
    } else if (pass == 0) {
      Stamp(pool, query + 2, stream);
    } else if (pass == 1 && hybridCap != 0) {
      // the hybrid plan's second half, between launch 0 and launch 1 (which is empty when the plan applies): its time
      // is this pass's "upsweep" slot
      vrdx::BucketSortArgs b;
      b.keysScratch = keysScratch;
      b.keysCaller = keys;
      b.valuesScratch = keyValue ? valuesScratch : nullptr;
      b.valuesCaller = keyValue ? values : nullptr;
      b.maxCount = elementCount;
      b.countPtr = countPtr;
      b.histogramTable = globalHistogram;
      b.hybridCap = hybridCap;
      b.planWord = reinterpret_cast<const uint32_t*>(storage + VRDX_OFF_PLAN);
      EnqueueCheck(sorter, "bucket_sort_kernel", vrdx::LaunchBucketSort(stream, keyValue, atomicRank, b));
      Stamp(pool, query + 2 + 3 * pass + 0, stream);
    } else {
      StampSame(pool, query + 2 + 3 * pass + 0, query + 2 + 3 * (pass - 1) + 2);
    }
    if (msdBits == 0) StampSame(pool, query + 2 + 3 * pass + 1, query + 2 + 3 * pass + 0);  // "spine" (fused into the look-back)

    const vrdx::OnesweepArgs args = passArgs(pass);
    EnqueueCheck(sorter, "onesweep_kernel",
                 vrdx::LaunchOnesweep(stream, configIndex, tiles, keyValue, atomicRank, args));

    if (msdBits == 0 || pass != 0) Stamp(pool, query + 2 + 3 * pass + 2, stream);  // "downsweep"
  }
  StampSame(pool, query + 14, query + 13);  // end of the sort = end of the last pass

  // Behind every 65536th sort: 8 workgroups repeat the lane-order check of vrdxCreateSorter (~20 us, never blocks; a
  // mismatch sets VRDX_HIP_STATUS_RANK_ORDER in the sorter's status word, which vrdxHipReadSorterStatus and
  // vrdxDestroySorter report).
  // Not into a stream capture: the check would be baked into the graph and run with every replay.  The count is not
  // advanced then, so the first sort recorded outside a capture makes up for it.
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (atomicRank && (sorter->sortsRecorded.load(std::memory_order_relaxed) & 0xFFFFu) == 0xFFFFu &&
      (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone))
    return;
  if (atomicRank && (sorter->sortsRecorded.fetch_add(1u, std::memory_order_relaxed) & 0xFFFFu) == 0xFFFFu)
    EnqueueCheck(sorter, "lds_order_check_kernel", vrdx::LaunchLdsOrderRecheck(stream, sorter->stickyStatus));
}

}  // namespace

extern "C" {

VkResult vrdxCreateSorter(const VrdxSorterCreateInfo* pCreateInfo, VrdxSorter* pSorter) {
  if (pCreateInfo == nullptr || pSorter == nullptr) return VK_ERROR_INITIALIZATION_FAILED;

  int deviceCount = 0;
  if (hipGetDeviceCount(&deviceCount) != hipSuccess || deviceCount <= 0)
    return VK_ERROR_INITIALIZATION_FAILED;

  int ordinal = 0;
  const void* handle = pCreateInfo->device != nullptr ? (const void*)pCreateInfo->device
                                                      : (const void*)pCreateInfo->physicalDevice;
  if (DeviceOrdinalFromHandle(handle, &ordinal) != 0) return VK_ERROR_INITIALIZATION_FAILED;
  if (ordinal < 0 || ordinal >= deviceCount) return VK_ERROR_INITIALIZATION_FAILED;

  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) return VK_ERROR_INITIALIZATION_FAILED;
  // The code object holds gfx950 ISA only (wave64, 160 KiB LDS, sc1 status protocol).
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VK_ERROR_FEATURE_NOT_PRESENT;

  VrdxSorter sorter = new (std::nothrow) VrdxSorter_T;
  if (sorter == nullptr) return VK_ERROR_OUT_OF_HOST_MEMORY;
  sorter->device = ordinal;
  sorter->computeUnits = prop.multiProcessorCount;

  int previous = 0;
  (void)hipGetDevice(&previous);
  hipError_t e = hipSetDevice(ordinal);
  for (int i = 0; i < vrdx::kNumTileConfigs && e == hipSuccess; ++i) e = vrdx::PrepareKernels(i);
  if (e == hipSuccess) e = vrdx::PrepareSmallSort();
  if (e == hipSuccess) e = vrdx::PrepareBucketSort();
  if (e == hipSuccess) e = vrdx::PrepareMsd();
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&sorter->stickyStatus), 2 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(sorter->stickyStatus, 0, 2 * sizeof(uint32_t));
  if (e == hipSuccess) sorter->declinedPlans = sorter->stickyStatus + 1;
  if (e == hipSuccess) {
    // Ranking mode: the single-atomic form needs a hardware property the ISA manual does not
    // promise, so it is verified here, once, on this very device; VRDX_RANK=ballot|atomic|auto.
    const char* mode = std::getenv("VRDX_RANK");
    if (mode != nullptr && std::strcmp(mode, "ballot") == 0) {
      sorter->atomicRank = false;
    } else {
      bool ordered = false;
      e = vrdx::LdsOrderCheck(&ordered);
      sorter->atomicRank = ordered;
      if (e == hipSuccess && !ordered) {
        // Never silent: the ballot form is correct everywhere but 1.2-1.7x slower (profiles/r03_ballot_ranking.txt).
        std::fprintf(stderr,
                     "vrdx-hip: LDS returning atomics are not lane-ordered on device %d: ranking with wave ballots "
                     "instead (same results, 1.2-1.7x the time)%s\n",
                     ordinal, mode != nullptr && std::strcmp(mode, "atomic") == 0 ? "; VRDX_RANK=atomic refused" : "");
      }
    }
  }
  (void)hipSetDevice(previous);
  if (e != hipSuccess) {
    if (sorter->stickyStatus != nullptr) (void)hipFree(sorter->stickyStatus);
    delete sorter;  // reference cleanup(): nothing half-built survives (:153-158)
    return VK_ERROR_INITIALIZATION_FAILED;
  }

  *pSorter = sorter;
  return VK_SUCCESS;
}

void vrdxDestroySorter(VrdxSorter sorter) {
  if (sorter == nullptr) return;  // reference :268
#ifdef VRDX_TRACE
  DumpTrace();
#endif
  // Last chance to be heard: the vrdxCmdSort* entry points return void like the reference's, so a caller that never
  // asks vrdxHipReadSorterStatus would not learn that a sort gave up a look-back (result unspecified), that the
  // lane-order re-check failed or that the runtime refused an enqueue.  One line on stderr, only if something did.
  if (sorter->stickyStatus != nullptr) {
    // on the sorter's device, whatever the calling thread's current one is (one thread may own a sorter per GPU)
    int previous = -1;
    const bool switched = hipGetDevice(&previous) == hipSuccess && previous != sorter->device &&
                          hipSetDevice(sorter->device) == hipSuccess;
    struct Restore {
      bool on;
      int device;
      ~Restore() {
        if (on) (void)hipSetDevice(device);
      }
    } restore{switched, previous};
    uint32_t word = 0;
    if (hipMemcpy(&word, sorter->stickyStatus, sizeof(word), hipMemcpyDeviceToHost) != hipSuccess) word = 0;  // (synchronises)
    if (sorter->enqueueFailed.load(std::memory_order_relaxed) != 0) word |= VRDX_HIP_STATUS_ENQUEUE_REFUSED;
    if (sorter->countClamped.load(std::memory_order_relaxed) != 0) word |= VRDX_HIP_STATUS_COUNT_CLAMPED;
    if (word != 0)
      std::fprintf(stderr,
                   "vrdx-hip: sorter destroyed with unreported failures (status 0x%08x:%s%s%s%s) -- see vrdxHipReadSorterStatus\n",
                   word, (word & VRDX_HIP_STATUS_LOOKBACK_GAVE_UP) ? " a look-back spin expired, that sort's result is unspecified;" : "",
                   (word & VRDX_HIP_STATUS_RANK_ORDER) ? " LDS atomics were seen out of lane order, call vrdxHipRecheck;" : "",
                   (word & VRDX_HIP_STATUS_COUNT_CLAMPED) ? " an element count beyond 2^30 - 4 was clamped, that sort's tail is unsorted;" : "",
                   (word & VRDX_HIP_STATUS_ENQUEUE_REFUSED) ? " the HIP runtime refused an enqueue;" : "");
    (void)hipFree(sorter->stickyStatus);
  }
  delete sorter;
}

void vrdxGetSorterStorageRequirements(VrdxSorter sorter, uint32_t maxElementCount,
                                      VrdxSorterStorageRequirements* requirements) {
  const vrdx::StorageLayout layout =
      vrdx::MakeLayout(maxElementCount, sorter->minStorageBufferOffsetAlignment, 0);  // the size is the reference's formula
  requirements->size = layout.keysOnlySize;
  requirements->usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT | VK_BUFFER_USAGE_TRANSFER_DST_BIT;
}

void vrdxGetSorterKeyValueStorageRequirements(VrdxSorter sorter, uint32_t maxElementCount,
                                              VrdxSorterStorageRequirements* requirements) {
  const vrdx::StorageLayout layout =
      vrdx::MakeLayout(maxElementCount, sorter->minStorageBufferOffsetAlignment, 0);
  requirements->size = layout.keyValueSize;
  requirements->usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT | VK_BUFFER_USAGE_TRANSFER_DST_BIT;
}

void vrdxCmdSort(VkCommandBuffer commandBuffer, VrdxSorter sorter, uint32_t elementCount,
                 VkBuffer keysBuffer, VkDeviceSize keysOffset, VkBuffer storageBuffer,
                 VkDeviceSize storageOffset, VkQueryPool queryPool, uint32_t query) {
  RecordSort(commandBuffer, sorter, elementCount, nullptr, 0, keysBuffer, keysOffset, nullptr, 0,
             storageBuffer, storageOffset, queryPool, query);
}

void vrdxCmdSortIndirect(VkCommandBuffer commandBuffer, VrdxSorter sorter, uint32_t maxElementCount,
                         VkBuffer indirectBuffer, VkDeviceSize indirectOffset, VkBuffer keysBuffer,
                         VkDeviceSize keysOffset, VkBuffer storageBuffer,
                         VkDeviceSize storageOffset, VkQueryPool queryPool, uint32_t query) {
  RecordSort(commandBuffer, sorter, maxElementCount, indirectBuffer, indirectOffset, keysBuffer,
             keysOffset, nullptr, 0, storageBuffer, storageOffset, queryPool, query);
}

void vrdxCmdSortKeyValue(VkCommandBuffer commandBuffer, VrdxSorter sorter, uint32_t elementCount,
                         VkBuffer keysBuffer, VkDeviceSize keysOffset, VkBuffer valuesBuffer,
                         VkDeviceSize valuesOffset, VkBuffer storageBuffer,
                         VkDeviceSize storageOffset, VkQueryPool queryPool, uint32_t query) {
  RecordSort(commandBuffer, sorter, elementCount, nullptr, 0, keysBuffer, keysOffset, valuesBuffer,
             valuesOffset, storageBuffer, storageOffset, queryPool, query);
}

void vrdxCmdSortKeyValueIndirect(VkCommandBuffer commandBuffer, VrdxSorter sorter,
                                 uint32_t maxElementCount, VkBuffer indirectBuffer,
                                 VkDeviceSize indirectOffset, VkBuffer keysBuffer,
                                 VkDeviceSize keysOffset, VkBuffer valuesBuffer,
                                 VkDeviceSize valuesOffset, VkBuffer storageBuffer,
                                 VkDeviceSize storageOffset, VkQueryPool queryPool,
                                 uint32_t query) {
  RecordSort(commandBuffer, sorter, maxElementCount, indirectBuffer, indirectOffset, keysBuffer,
             keysOffset, valuesBuffer, valuesOffset, storageBuffer, storageOffset, queryPool, query);
}

VkResult vrdxHipCreateQueryPool(uint32_t queryCount, VkQueryPool* pQueryPool) {
  if (pQueryPool == nullptr || queryCount == 0) return VK_ERROR_INITIALIZATION_FAILED;
  VrdxHipQueryPool* pool = new (std::nothrow) VrdxHipQueryPool;
  if (pool == nullptr) return VK_ERROR_OUT_OF_HOST_MEMORY;
  pool->events = new (std::nothrow) hipEvent_t[queryCount];
  pool->recorded = new (std::nothrow) uint8_t[queryCount];
  pool->source = new (std::nothrow) uint32_t[queryCount];
  if (pool->events == nullptr || pool->recorded == nullptr || pool->source == nullptr) {
    delete[] pool->events;
    delete[] pool->recorded;
    delete[] pool->source;
    delete pool;
    return VK_ERROR_OUT_OF_HOST_MEMORY;
  }
  std::memset(pool->recorded, 0, queryCount);
  for (uint32_t i = 0; i < queryCount; ++i) pool->source[i] = i;
  for (uint32_t i = 0; i < queryCount; ++i) {
    if (hipEventCreate(&pool->events[i]) != hipSuccess) {
      for (uint32_t j = 0; j < i; ++j) (void)hipEventDestroy(pool->events[j]);
      delete[] pool->events;
      delete[] pool->recorded;
      delete[] pool->source;
      delete pool;
      return VK_ERROR_INITIALIZATION_FAILED;
    }
    pool->count = i + 1;
  }
  *pQueryPool = reinterpret_cast<VkQueryPool>(pool);
  return VK_SUCCESS;
}

void vrdxHipDestroyQueryPool(VkQueryPool queryPool) {
  VrdxHipQueryPool* pool = reinterpret_cast<VrdxHipQueryPool*>(queryPool);
  if (pool == nullptr) return;
  for (uint32_t i = 0; i < pool->count; ++i) (void)hipEventDestroy(pool->events[i]);
  delete[] pool->events;
  delete[] pool->recorded;
  delete[] pool->source;
  delete pool;
}

VkResult vrdxHipGetQueryPoolResults(VkQueryPool queryPool, uint32_t firstQuery, uint32_t queryCount,
                                    uint64_t* pData) {
  VrdxHipQueryPool* pool = reinterpret_cast<VrdxHipQueryPool*>(queryPool);
  if (pool == nullptr || pData == nullptr || firstQuery + queryCount > pool->count)
    return VK_ERROR_INITIALIZATION_FAILED;
  for (uint32_t i = 0; i < queryCount; ++i) {
    if (!pool->recorded[firstQuery + i]) return VK_NOT_READY;
    float ms = 0.0f;
    const hipError_t e = hipEventElapsedTime(&ms, pool->events[pool->source[firstQuery]],
                                             pool->events[pool->source[firstQuery + i]]);
    if (e == hipErrorNotReady) return VK_NOT_READY;
    if (e != hipSuccess) return VK_ERROR_DEVICE_LOST;
    pData[i] = ms <= 0.0f ? 0ull : (uint64_t)((double)ms * 1.0e6 + 0.5);
  }
  return VK_SUCCESS;
}

uint32_t vrdxHipReadStatus(VkCommandBuffer commandBuffer, VkBuffer storageBuffer,
                           VkDeviceSize storageOffset) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(commandBuffer);
  uint32_t word = 0xFFFFFFFFu;
  if (hipMemcpyAsync(&word, BufferAddress(storageBuffer, storageOffset) + VRDX_OFF_FAILURE,
                     sizeof(word), hipMemcpyDeviceToHost, stream) != hipSuccess)
    return 0xFFFFFFFFu;
  if (hipStreamSynchronize(stream) != hipSuccess) return 0xFFFFFFFFu;
  return word;
}

uint32_t vrdxHipReadSorterStatus(VrdxSorter sorter, VkCommandBuffer commandBuffer) {
  if (sorter == nullptr || sorter->stickyStatus == nullptr) return 0xFFFFFFFFu;
  hipStream_t stream = reinterpret_cast<hipStream_t>(commandBuffer);
  uint32_t word = 0xFFFFFFFFu;
  if (hipMemcpyAsync(&word, sorter->stickyStatus, sizeof(word), hipMemcpyDeviceToHost, stream) != hipSuccess)
    return 0xFFFFFFFFu;
  const hipError_t cleared = hipMemsetAsync(sorter->stickyStatus, 0, sizeof(word), stream);
  // the copy above targets `word` on this stack frame: never return while it may still be in flight
  if (hipStreamSynchronize(stream) != hipSuccess || cleared != hipSuccess) return 0xFFFFFFFFu;
  if (sorter->enqueueFailed.exchange(0u, std::memory_order_relaxed) != 0) word |= VRDX_HIP_STATUS_ENQUEUE_REFUSED;
  if (sorter->countClamped.exchange(0u, std::memory_order_relaxed) != 0) word |= VRDX_HIP_STATUS_COUNT_CLAMPED;
  return word;
}

VkResult vrdxHipRecheck(VrdxSorter sorter) {
  if (sorter == nullptr) return VK_ERROR_INITIALIZATION_FAILED;
  const char* mode = std::getenv("VRDX_RANK");
  if (mode != nullptr && std::strcmp(mode, "ballot") == 0) return VK_SUCCESS;  // nothing rests on the property
  int previous = 0;
  (void)hipGetDevice(&previous);
  if (hipSetDevice(sorter->device) != hipSuccess) return VK_ERROR_DEVICE_LOST;
  bool ordered = false;
  const hipError_t e = vrdx::LdsOrderCheck(&ordered);
  (void)hipSetDevice(previous);
  if (e != hipSuccess) return VK_ERROR_DEVICE_LOST;
  const bool was = sorter->atomicRank.exchange(ordered, std::memory_order_relaxed);
  if (was && !ordered)
    std::fprintf(stderr,
                 "vrdx-hip: LDS returning atomics are no longer lane-ordered on device %d: sorts recorded from now on rank "
                 "with wave ballots (same results, 1.2-1.7x the time); results of earlier sorts may be unstable\n",
                 sorter->device);
  return VK_SUCCESS;
}

uint64_t vrdxHipEventOverheadNs(VkCommandBuffer commandBuffer) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(commandBuffer);
  int device = 0, clockKhz = 0;
  unsigned long long* stamps = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  uint64_t result = ~0ull;
  // the stream's device, not the calling thread's current one (the null stream: the current device)
  if ((stream != nullptr ? hipStreamGetDevice(stream, &device) : hipGetDevice(&device)) != hipSuccess ||
      hipDeviceGetAttribute(&clockKhz, hipDeviceAttributeWallClockRate, device) != hipSuccess || clockKhz <= 0)
    return result;
  int previousDevice = -1;
  const bool switchedDevice = hipGetDevice(&previousDevice) == hipSuccess && previousDevice != device &&
                              hipSetDevice(device) == hipSuccess;
  struct RestoreDevice {
    bool on;
    int device;
    ~RestoreDevice() {
      if (on) (void)hipSetDevice(device);
    }
  } restoreDevice{switchedDevice, previousDevice};
  if (hipMalloc(reinterpret_cast<void**>(&stamps), 2 * sizeof(unsigned long long)) != hipSuccess) return result;
  if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
    // a kernel that demonstrably runs 40 us, right behind another one (a busy stream, like the passes of a sort)
    const uint32_t ticks = (uint32_t)(40ull * (uint64_t)clockKhz / 1000ull);
    std::vector<uint64_t> extra;
    for (int run = 0; run < 9; ++run) {
      unsigned long long host[2] = {0, 0};
      float ms = 0.0f;
      if (vrdx::LaunchSpin(stream, stamps, ticks) != hipSuccess || hipEventRecord(e0, stream) != hipSuccess ||
          vrdx::LaunchSpin(stream, stamps, ticks) != hipSuccess || hipEventRecord(e1, stream) != hipSuccess ||
          hipStreamSynchronize(stream) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess ||
          hipMemcpy(host, stamps, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess)
        break;
      const double ran = (double)(host[1] - host[0]) * 1.0e6 / (double)clockKhz;  // ns
      const double between = (double)ms * 1.0e6;
      if (run > 0) extra.push_back(between > ran ? (uint64_t)(between - ran + 0.5) : 0ull);
    }
    if (!extra.empty()) {
      std::sort(extra.begin(), extra.end());
      result = extra[extra.size() / 2];
    }
  }
  if (e0 != nullptr) (void)hipEventDestroy(e0);
  if (e1 != nullptr) (void)hipEventDestroy(e1);
  (void)hipFree(stamps);
  return result;
}

void vrdxHipDescribePlan(VrdxSorter sorter, uint32_t elementCount, int keyValue, VrdxHipPlanInfo* info) {
  if (info == nullptr) return;
  std::memset(info, 0, sizeof(*info));
  if (sorter == nullptr || elementCount == 0) return;
  if (elementCount > VRDX_MAX_ELEMENTS) elementCount = VRDX_MAX_ELEMENTS;
  const bool kv = keyValue != 0;
  const uint32_t fourPasses = kv ? 68u : 36u;  // 4 (histogram) + 4 x (read + write)
  const uint32_t twoTrips = kv ? 36u : 20u;    // 4 (histogram) + scatter (read + write) + buckets (read + write)
  // the recorder's own planning function (storage address 0: the alignment at which the least fits)
  const SortPlan plan = PlanSort(sorter, kv, elementCount, 0);
  info->fallbackBytesPerElement = fourPasses;
  info->launches = plan.launches;
  if (plan.oneWorkgroup) {
    info->plan = VRDX_HIP_PLAN_ONE_WORKGROUP;
    info->bytesPerElement = info->fallbackBytesPerElement = kv ? 16u : 8u;
  } else if (plan.msdBits != 0) {
    info->plan = VRDX_HIP_PLAN_MSD;
    info->bits = plan.msdBits;
    info->bytesPerElement = twoTrips;
  } else if (plan.hybridCap != 0) {
    info->plan = VRDX_HIP_PLAN_HYBRID8;
    info->bits = 8;
    info->bytesPerElement = twoTrips;
  } else {
    info->plan = VRDX_HIP_PLAN_FOUR_PASSES;
    info->bytesPerElement = fourPasses;
  }
}

uint32_t vrdxHipReadPlanVerdict(VkCommandBuffer commandBuffer, VkBuffer storageBuffer, VkDeviceSize storageOffset) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(commandBuffer);
  uint32_t word = 0xFFFFFFFFu;
  if (hipMemcpyAsync(&word, BufferAddress(storageBuffer, storageOffset) + VRDX_OFF_PLAN, sizeof(word), hipMemcpyDeviceToHost,
                     stream) != hipSuccess)
    return 0xFFFFFFFFu;
  if (hipStreamSynchronize(stream) != hipSuccess) return 0xFFFFFFFFu;
  return word & vrdx::kMsdVerdictMask;  // (the MSD plan's scatter also notes its window's shift there, bits 8-13)
}

VkResult vrdxHipReadPlanCounters(VrdxSorter sorter, VkCommandBuffer commandBuffer, uint32_t* pRecorded, uint32_t* pDeclined) {
  if (sorter == nullptr || sorter->declinedPlans == nullptr) return VK_ERROR_INITIALIZATION_FAILED;
  hipStream_t stream = reinterpret_cast<hipStream_t>(commandBuffer);
  uint32_t declined = 0;
  if (hipMemcpyAsync(&declined, sorter->declinedPlans, sizeof(declined), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)  // (the copy targets this stack frame: never return while it may be in flight)
    return VK_ERROR_DEVICE_LOST;
  if (pRecorded != nullptr) *pRecorded = sorter->plansRecorded.load(std::memory_order_relaxed);
  if (pDeclined != nullptr) *pDeclined = declined;
  return VK_SUCCESS;
}

const char* vrdxHipVersionString(void) {
  // built once (thread-safe static initialisation), never rewritten: concurrent callers read one immutable buffer
  struct Text {
    char text[160];
    Text() {
      VrdxSorter_T nominal;  // an MI355X: 256 CUs, lane-ordered LDS atomics
      nominal.computeUnits = 256;
      const vrdx::TileConfig& k = vrdx::kTileConfigs[ConfigIndex(&nominal, false, 1u << 25, true, false)];
      const vrdx::TileConfig& kv = vrdx::kTileConfigs[ConfigIndex(&nominal, true, 1u << 25, true, false)];
      char kName[32], kvName[32];
      ConfigName(k, kName, sizeof(kName));
      ConfigName(kv, kvName, sizeof(kvName));
      std::snprintf(text, sizeof(text), "vrdx-hip %d.%d.%d gfx950 tiles at 2^25: keys=%s key-value=%s%s",
                    VRDX_VERSION_MAJOR, VRDX_VERSION_MINOR, VRDX_VERSION_PATCH, kName, kvName,
                    ForcedConfigIndex() >= 0 ? " (forced)" : " (size-adaptive)");
    }
  };
  static const Text once;
  return once.text;
}

}  // extern "C"

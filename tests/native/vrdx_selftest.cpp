// vrdx_selftest.cpp -- native GPU parity + timing driver for libvrdx_hip.so (no Python, no torch).
//
// Calls the product exclusively through the C-ABI of include/vk_radix_sort.h and checks every
// result bit for bit against the CPU oracle (oracle/liboracle.so: vrdx_oracle_sort).  Its checks
// are the reference's own correctness predicate (bench/bench.cc:41-64 under /root/reference:
// keys equal std::sort, keys+values equal std::stable_sort by key) extended to the edge cases
// the reference never tests.
//
//   vrdx_selftest parity            # parity battery, exit code != 0 on any mismatch
//   vrdx_selftest bench [log2n...]  # warm-up 1 + 10 timed runs per size, median (bench.cc:66-112)
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/vk_radix_sort.h"

extern "C" {
int vrdx_oracle_sort(uint32_t* keys, uint32_t* values, uint32_t elementCount, uint32_t* globalHistogramOut);
uint64_t vrdx_oracle_storage_size(uint32_t maxElementCount, uint32_t align, int keyValue);
}

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      std::exit(2);                                                                    \
    }                                                                                  \
  } while (0)

namespace {

// VRDX_SELFTEST_STORAGE_OFFSET=<bytes>: the timing modes (bench, sweep, adversarial) hand the storage over at this
// offset into its allocation (a multiple of 16, like minStorageBufferOffsetAlignment) -- how the alignment of the
// storage's scratch arrays was measured (profiles/r04_kv_pass_parity.txt).
VkDeviceSize StorageOffset() {
  static const VkDeviceSize offset = [] {
    const char* env = std::getenv("VRDX_SELFTEST_STORAGE_OFFSET");
    return env != nullptr ? (VkDeviceSize)std::strtoull(env, nullptr, 10) : (VkDeviceSize)0;
  }();
  return offset;
}

constexpr uint32_t kGuard = 64;          // untouched-tail guard elements
constexpr uint32_t kGuardWord = 0xDEADBEEFu;

struct Harness {
  VrdxSorter sorter = nullptr;
  hipStream_t stream = nullptr;
  uint8_t* dKeys = nullptr;     // keys | values | count, like bench/vulkan_benchmark.cc:386-388
  uint8_t* dStorage = nullptr;
  size_t keysCap = 0, storageCap = 0;
  VkQueryPool pool = nullptr;

  void init() {
    VrdxSorterCreateInfo info = {};
    VkResult r = vrdxCreateSorter(&info, &sorter);
    if (r != VK_SUCCESS) {
      std::fprintf(stderr, "vrdxCreateSorter failed: %d\n", (int)r);
      std::exit(2);
    }
    // VRDX_SELFTEST_CU_MASK=<hex word>: the sorts run on a stream restricted to the CUs whose bit is set in
    // that word, repeated over all CUs (55555555 = every other CU) -- a diagnostic for how much of a pass is
    // contention between CUs (tools: DESIGN.md section 5.3), not a mode of the library.
    if (const char* mask = std::getenv("VRDX_SELFTEST_CU_MASK")) {
      const uint32_t word = (uint32_t)std::strtoul(mask, nullptr, 16);
      uint32_t words[16];
      for (auto& w : words) w = word;
      HIP_OK(hipExtStreamCreateWithCUMask(&stream, 16, words));
    } else {
      HIP_OK(hipStreamCreate(&stream));
    }
    if (vrdxHipCreateQueryPool(15, &pool) != VK_SUCCESS) std::exit(2);
  }
  void reserve(size_t keysBytes, size_t storageBytes) {
    if (keysBytes > keysCap) {
      if (dKeys) HIP_OK(hipFree(dKeys));
      HIP_OK(hipMalloc((void**)&dKeys, keysBytes));
      keysCap = keysBytes;
    }
    if (storageBytes > storageCap) {
      if (dStorage) HIP_OK(hipFree(dStorage));
      HIP_OK(hipMalloc((void**)&dStorage, storageBytes));
      storageCap = storageBytes;
    }
  }
};

enum class Mode { Keys, KeysIndirect, KeyValue, KeyValueIndirect };
const char* ModeName(Mode m) {
  switch (m) {
    case Mode::Keys: return "keys";
    case Mode::KeysIndirect: return "keys-indirect";
    case Mode::KeyValue: return "kv";
    default: return "kv-indirect";
  }
}

uint32_t Align16(uint32_t x) { return (x + 15u) / 16u * 16u; }

// One sort through the C-ABI + bit-exact comparison with the oracle.  `maxCount` >= n is what the
// host passes in indirect mode (the device-side count is n).
bool RunCase(Harness& h, Mode mode, const std::vector<uint32_t>& keys, const std::vector<uint32_t>& values,
             uint32_t maxCount, const char* label, bool poisonStorage) {
  const uint32_t n = (uint32_t)keys.size();
  const bool kv = mode == Mode::KeyValue || mode == Mode::KeyValueIndirect;
  const bool indirect = mode == Mode::KeysIndirect || mode == Mode::KeyValueIndirect;
  if (!indirect) maxCount = n;

  const uint32_t inout = Align16((maxCount + kGuard) * 4u);
  VrdxSorterStorageRequirements req;
  if (kv)
    vrdxGetSorterKeyValueStorageRequirements(h.sorter, maxCount, &req);
  else
    vrdxGetSorterStorageRequirements(h.sorter, maxCount, &req);
  if (req.size != vrdx_oracle_storage_size(maxCount, 16, kv ? 1 : 0) || req.usage != 0x22u) {
    std::printf("FAIL %-34s %-12s n=%u storage requirement %llu (usage 0x%x) differs from the oracle\n", label,
                ModeName(mode), n, (unsigned long long)req.size, req.usage);
    return false;
  }
  const size_t guardBytes = 256;
  h.reserve((size_t)2 * inout + 16, (size_t)req.size + guardBytes);

  // host image: keys | values | count, tails poisoned
  std::vector<uint32_t> hk(inout / 4, kGuardWord), hv(inout / 4, kGuardWord);
  std::copy(keys.begin(), keys.end(), hk.begin());
  if (kv) std::copy(values.begin(), values.end(), hv.begin());
  HIP_OK(hipMemcpy(h.dKeys, hk.data(), inout, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h.dKeys + inout, hv.data(), inout, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h.dKeys + 2 * (size_t)inout, &n, 4, hipMemcpyHostToDevice));
  // storage arrives with arbitrary contents; a guard band after it must survive
  HIP_OK(hipMemset(h.dStorage, poisonStorage ? 0xA5 : 0x00, (size_t)req.size));
  HIP_OK(hipMemset(h.dStorage + req.size, 0x5A, guardBytes));

  VkCommandBuffer cmd = (VkCommandBuffer)h.stream;
  VkBuffer buf = (VkBuffer)h.dKeys;
  VkBuffer sto = (VkBuffer)h.dStorage;
  switch (mode) {
    case Mode::Keys: vrdxCmdSort(cmd, h.sorter, n, buf, 0, sto, 0, h.pool, 0); break;
    case Mode::KeysIndirect:
      vrdxCmdSortIndirect(cmd, h.sorter, maxCount, buf, 2 * (VkDeviceSize)inout, buf, 0, sto, 0, h.pool, 0);
      break;
    case Mode::KeyValue: vrdxCmdSortKeyValue(cmd, h.sorter, n, buf, 0, buf, inout, sto, 0, h.pool, 0); break;
    case Mode::KeyValueIndirect:
      vrdxCmdSortKeyValueIndirect(cmd, h.sorter, maxCount, buf, 2 * (VkDeviceSize)inout, buf, 0, buf, inout, sto,
                                  0, h.pool, 0);
      break;
  }
  HIP_OK(hipStreamSynchronize(h.stream));
  const uint32_t failure = n ? vrdxHipReadStatus(cmd, sto, 0) : 0;

  std::vector<uint32_t> gk(inout / 4), gv(inout / 4);
  std::vector<uint8_t> guard(guardBytes);
  HIP_OK(hipMemcpy(gk.data(), h.dKeys, inout, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(gv.data(), h.dKeys + inout, inout, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(guard.data(), h.dStorage + req.size, guardBytes, hipMemcpyDeviceToHost));

  std::vector<uint32_t> ek = keys, ev = values;
  vrdx_oracle_sort(ek.data(), kv ? ev.data() : nullptr, n, nullptr);

  bool ok = failure == 0;
  long firstBad = -1;
  for (uint32_t i = 0; i < n && firstBad < 0; ++i)
    if (gk[i] != ek[i] || (kv && gv[i] != ev[i])) firstBad = i;
  if (firstBad >= 0) ok = false;
  bool tailOk = true;
  for (uint32_t i = n; i < inout / 4; ++i)
    if (gk[i] != kGuardWord || gv[i] != kGuardWord) tailOk = false;
  for (uint8_t b : guard)
    if (b != 0x5A) tailOk = false;
  if (!tailOk) ok = false;

  uint64_t ts[15] = {0};
  const VkResult tr = vrdxHipGetQueryPoolResults(h.pool, 0, 15, ts);
  if (tr != VK_SUCCESS) ok = false;
  for (int i = 1; i < 15 && tr == VK_SUCCESS; ++i)
    if (ts[i] < ts[i - 1]) ok = false;

  if (!ok) {
    std::printf("FAIL %-34s %-12s n=%u max=%u failure=%u firstBad=%ld tailOk=%d ts=%d\n", label, ModeName(mode), n,
                maxCount, failure, firstBad, (int)tailOk, (int)tr);
    if (firstBad >= 0)
      std::printf("     got key %08x val %08x, want key %08x val %08x\n", gk[firstBad], gv[firstBad], ek[firstBad],
                  kv ? ev[firstBad] : 0u);
  }
  return ok;
}

std::vector<uint32_t> Mt(uint32_t n, int seed, uint32_t bits, std::vector<uint32_t>* values) {
  // DataGenerator(seed).Generate(n, bits): keys then values from one mt19937 stream
  std::mt19937 gen(seed);
  std::vector<uint32_t> k(n);
  for (auto& x : k) {
    const uint32_t r = gen();
    x = bits >= 32 ? r : (bits == 0 ? 0u : r >> (32 - bits));
  }
  if (values) {
    values->resize(n);
    for (auto& x : *values) x = gen();
  }
  return k;
}


// VkCommandBuffer == stream-ordered enqueue: vrdxCmdSort* must be capturable into a hipGraph
// (no query pool inside a capture) and the graph must be replayable.
bool GraphCase(Harness& h, bool kv, uint32_t n) {
  std::vector<uint32_t> v;
  auto k = Mt(n, 31, 32, &v);
  const uint32_t inout = Align16(n * 4u);
  VrdxSorterStorageRequirements req;
  if (kv)
    vrdxGetSorterKeyValueStorageRequirements(h.sorter, n, &req);
  else
    vrdxGetSorterStorageRequirements(h.sorter, n, &req);
  h.reserve((size_t)2 * inout + 16, (size_t)req.size + 256);
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  HIP_OK(hipStreamBeginCapture(h.stream, hipStreamCaptureModeGlobal));
  if (kv)
    vrdxCmdSortKeyValue((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dKeys, inout,
                        (VkBuffer)h.dStorage, 0, VK_NULL_HANDLE, 0);
  else
    vrdxCmdSort((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dStorage, 0, VK_NULL_HANDLE, 0);
  HIP_OK(hipStreamEndCapture(h.stream, &graph));
  size_t nodes = 0;
  HIP_OK(hipGraphGetNodes(graph, nullptr, &nodes));
  HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  std::vector<uint32_t> ek = k, ev = v;
  vrdx_oracle_sort(ek.data(), kv ? ev.data() : nullptr, n, nullptr);
  // general path: clear + histogram + four passes; sorts of <= 16384 elements are one kernel
  bool ok = nodes >= (n <= 16384u ? 1u : 6u);
  for (int replay = 0; replay < 3 && ok; ++replay) {
    HIP_OK(hipMemcpy(h.dKeys, k.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h.dKeys + inout, v.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipGraphLaunch(exec, h.stream));
    HIP_OK(hipStreamSynchronize(h.stream));
    std::vector<uint32_t> gk(n), gv(n);
    HIP_OK(hipMemcpy(gk.data(), h.dKeys, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gv.data(), h.dKeys + inout, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (gk != ek || (kv && gv != ev)) ok = false;
  }
  HIP_OK(hipGraphExecDestroy(exec));
  HIP_OK(hipGraphDestroy(graph));
  if (!ok) std::printf("FAIL hipGraph capture/replay %s n=%u nodes=%zu\n", kv ? "kv" : "keys", n, nodes);
  return ok;
}

int Parity(Harness& h, bool quick) {
  int failures = 0, cases = 0;
  auto run = [&](Mode m, const std::vector<uint32_t>& k, const std::vector<uint32_t>& v, uint32_t maxCount,
                 const char* label, bool poison = true) {
    ++cases;
    if (!RunCase(h, m, k, v, maxCount, label, poison)) ++failures;
  };
  const Mode allModes[] = {Mode::Keys, Mode::KeyValue, Mode::KeysIndirect, Mode::KeyValueIndirect};

  // sizes around every boundary of the reference (4096 partitions) and of our tiles (8192/16384)
  std::vector<uint32_t> sizes = {0,     1,     2,     63,    64,    65,    511,   512,   513,   4095,
                                 4096,  4097,  8191,  8192,  8193,  12411, 16383, 16384, 16385, 24577,
                                 32768, 65539, 100000, 262144, 262145, 1u << 20, (1u << 20) + 7};
  if (!quick) {
    sizes.push_back(5000011);
    sizes.push_back(1u << 24);
    // tail-split plans (PlanTiles): whole rounds of full tiles, then one round of small equal tiles -- a little and a
    // lot past one and two rounds of 256 x 32768 (the indirect cases plan for a larger bound than they sort)
    sizes.push_back((1u << 23) + 4097);
    sizes.push_back((1u << 23) + (1u << 21) + 77);
    sizes.push_back((1u << 24) + 70001);
    sizes.push_back(3u * (1u << 23) + 200003);
    // 8.1 M < N <= 18.1 M: the MSD plan with the half-size bucket kernel; the sizes above from 2^23 + 4097 to 2^24 take it
    sizes.push_back(16200000);
  }
  for (uint32_t n : sizes) {
    for (int seed : {1, 42}) {
      std::vector<uint32_t> v;
      auto k = Mt(n, seed, 32, &v);
      char label[64];
      std::snprintf(label, sizeof(label), "mt19937 seed=%d", seed);
      for (Mode m : allModes) {
        const bool indirect = m == Mode::KeysIndirect || m == Mode::KeyValueIndirect;
        if (indirect && seed != 1) continue;
        // indirect: host bound larger than the device-side count (grid sized from max)
        run(m, k, v, indirect ? n + n / 3 + 5000 : n, label);
      }
    }
  }
  // restricted-bit keys (Generate(n, bits), bench/data_generator.cc:15): heavy duplicates -> stability
  for (uint32_t bits : {0u, 1u, 2u, 4u, 8u, 12u, 16u, 24u}) {
    for (uint32_t n : {5000u, 70001u, 1u << 20}) {
      std::vector<uint32_t> v;
      auto k = Mt(n, 7, bits, &v);
      std::vector<uint32_t> iota(n);
      for (uint32_t i = 0; i < n; ++i) iota[i] = i;
      char label[64];
      std::snprintf(label, sizeof(label), "bits=%u values=iota", bits);
      run(Mode::KeyValue, k, iota, n, label);
      run(Mode::Keys, k, v, n, label);
      // same digits in the HIGH byte(s): exercises passes 1..3 with skew
      for (auto& x : k) x = (x << (32 - (bits ? bits : 1))) | (x & 0xFFu);
      std::snprintf(label, sizeof(label), "bits=%u high values=iota", bits);
      run(Mode::KeyValue, k, iota, n, label);
    }
  }
  // adversarial (BASELINE.json configs[3]): all-equal, all-0xFFFFFFFF (== the padding sentinel),
  // descending, ascending, few-distinct, per-pass digit skew
  {
    const uint32_t n = quick ? (1u << 20) + 12345 : (1u << 22) + 12345;
    std::vector<uint32_t> iota(n), k(n);
    for (uint32_t i = 0; i < n; ++i) iota[i] = i;
    std::fill(k.begin(), k.end(), 0x12345678u);
    run(Mode::KeyValue, k, iota, n, "all-equal 0x12345678");
    std::fill(k.begin(), k.end(), 0xFFFFFFFFu);
    run(Mode::KeyValue, k, iota, n, "all 0xFFFFFFFF (sentinel)");
    std::fill(k.begin(), k.end(), 0u);
    run(Mode::KeyValue, k, iota, n, "all zero");
    for (uint32_t i = 0; i < n; ++i) k[i] = n - 1 - i;
    run(Mode::KeyValue, k, iota, n, "descending");
    run(Mode::Keys, k, iota, n, "descending");
    for (uint32_t i = 0; i < n; ++i) k[i] = i;
    run(Mode::KeyValue, k, iota, n, "ascending");
    const uint32_t four[4] = {0xFFFFFFFFu, 0x00000000u, 0x80000001u, 0x7FFFFF00u};
    std::mt19937 g(3);
    for (uint32_t i = 0; i < n; ++i) k[i] = four[g() & 3];
    run(Mode::KeyValue, k, iota, n, "few-distinct (4 values)");
    for (int pass = 0; pass < 4; ++pass) {
      for (uint32_t i = 0; i < n; ++i) {
        uint32_t r = g();
        k[i] = r & ~(0xFFu << (8 * pass));  // digit `pass` constant 0, the others random
      }
      char label[64];
      std::snprintf(label, sizeof(label), "digit %d constant", pass);
      run(Mode::KeyValue, k, iota, n, label);
    }
    // mixed sentinel keys and a ragged tail
    for (uint32_t i = 0; i < n; ++i) k[i] = (g() & 7) ? g() : 0xFFFFFFFFu;
    run(Mode::KeyValue, k, iota, n, "1/8 sentinel keys");
  }
  // zero-initialised storage must work as well as poisoned storage; and storage reuse back to back
  {
    std::vector<uint32_t> v;
    auto k = Mt(300000, 5, 32, &v);
    run(Mode::KeyValue, k, v, 300000, "clean storage", false);
    run(Mode::KeyValue, k, v, 300000, "storage reuse");
    run(Mode::Keys, k, v, 300000, "storage reuse");
  }
  for (int kv = 0; kv < 2; ++kv)
    for (uint32_t n : {5000u, 200000u, 3000000u, 12000000u}) {  // one launch | eight-bit hybrid plan | ... | MSD plan, half-size buckets
      if (quick && n > 3000000u) continue;
      ++cases;
      if (!GraphCase(h, kv != 0, n)) ++failures;
    }
  std::printf("parity: %d cases, %d failures\n", cases, failures);
  return failures;
}

// The MSD plan (8.15 M elements and more): uniform keys at sizes on both sides of its ten- / eleven-bit ranges in every
// mode, and inputs the DEVICE must turn it down for (a bucket beyond the capacity: the four passes recorded behind it run).
// argv sizes override the list.
int MsdParity(Harness& h, const std::vector<uint32_t>& wanted) {
  int failures = 0, cases = 0;
  // expect: the verdict the device must reach (VRDX_HIP_VERDICT_MSD_RUNS / _MSD_SORTED), 0 = "the four passes ran", -1 = either
  auto run = [&](Mode m, const std::vector<uint32_t>& k, const std::vector<uint32_t>& v, uint32_t maxCount, const char* label,
                 int expect = -1) {
    ++cases;
    bool ok = RunCase(h, m, k, v, maxCount, label, true);
    const uint32_t verdict = vrdxHipReadPlanVerdict((VkCommandBuffer)h.stream, (VkBuffer)h.dStorage, 0);
    const bool taken = verdict == VRDX_HIP_VERDICT_MSD_RUNS || verdict == VRDX_HIP_VERDICT_MSD_SORTED;
    // (the expectation holds where the MSD plan is recorded: the first size of the list is the eight-bit plan's last)
    VrdxHipPlanInfo info;
    vrdxHipDescribePlan(h.sorter, maxCount, m == Mode::KeyValue || m == Mode::KeyValueIndirect, &info);
    if (info.plan == VRDX_HIP_PLAN_MSD && expect >= 0 && (expect == 0 ? taken : verdict != (uint32_t)expect)) {
      std::printf("FAIL %-34s %-12s n=%zu verdict %u, expected %d\n", label, ModeName(m), k.size(), verdict, expect);
      ok = false;
    }
    if (!ok) ++failures;
    std::fflush(stdout);
  };
  std::vector<uint32_t> sizes = wanted;
  if (sizes.empty()) sizes = {8144200u, 18149376u, 18149377u, 20000003u, 1u << 25, 36000001u, 45000000u};
  const int RUNS = VRDX_HIP_VERDICT_MSD_RUNS, SORTED = VRDX_HIP_VERDICT_MSD_SORTED;
  for (uint32_t n : sizes) {
    std::vector<uint32_t> v;
    auto k = Mt(n, 1, 32, &v);
    run(Mode::Keys, k, v, n, "msd uniform", RUNS);
    run(Mode::KeyValue, k, v, n, "msd uniform", RUNS);
    run(Mode::KeyValueIndirect, k, v, n + n / 7 + 5000, "msd uniform, larger bound");
    run(Mode::KeysIndirect, k, v, n + 70000, "msd uniform, larger bound");
    std::vector<uint32_t> iota(n);
    for (uint32_t i = 0; i < n; ++i) iota[i] = i;
    // 24-bit keys (round 6): the window the prologue chooses lies below their eight constant bits
    auto k24 = Mt(n, 3, 24, nullptr);
    run(Mode::KeyValue, k24, iota, n, "msd window: 24-bit keys", RUNS);
    run(Mode::Keys, k24, iota, n, "msd window: 24-bit keys", RUNS);
    // ... with ONE key outside that prefix, at the very end (no sample sees it: the count must): the four passes
    k24[n - 1] = 0x81234567u;
    run(Mode::KeyValue, k24, iota, n, "msd declines: prefix broken at n-1", 0);
    k24[n - 1] &= 0xFFFFFFu;
    k24[n / 3 + 7] |= 0x01000000u;  // (not one of the 64 sampled positions j (n - 1) / 63)
    run(Mode::Keys, k24, iota, n, "msd declines: prefix broken inside", 0);
    // 12-bit keys: ten bits of window, two bits for the local passes; 8-bit keys: at most 256 buckets can hold anything, which
    // at these sizes is more than they take -> turned down by the prologue
    auto k12 = Mt(n, 5, 12, nullptr);
    run(Mode::KeyValue, k12, iota, n, "msd window: 12-bit keys", n <= 37000000u ? RUNS : -1);  // (beyond: 1024 buckets cannot hold them)
    auto k8 = Mt(n, 6, 8, nullptr);
    run(Mode::KeyValue, k8, iota, n, "msd declines: 8-bit keys", 0);
    // duplicates inside buckets that fit: stability of the scatter and of both bucket passes (keys = 11 top bits | 3 low bits)
    std::mt19937 g(11);
    std::vector<uint32_t> dup(n);
    for (auto& x : dup) { const uint32_t r = g(); x = (r & 0xFFE00000u) | (r & 7u) | ((r >> 3 & 1u) << 12); }
    run(Mode::KeyValue, dup, iota, n, "msd duplicates values=iota", RUNS);
    // the same duplicates below a 9-bit prefix: window bits 12 ... 22 | one bit | three low bits, ONE local pass or two
    for (auto& x : dup) x = 0x5A800000u | (x >> 9);
    run(Mode::KeyValue, dup, iota, n, "msd window: duplicates under a prefix", RUNS);
    // one heavy bucket among uniform ones: a tenth of the keys share their top 11 bits
    for (uint32_t i = 0; i < n; ++i) dup[i] = (g() % 10u == 0) ? (0x5A400000u | (g() & 0x1FFFFFu)) : g();
    run(Mode::KeyValue, dup, iota, n, "msd declines: one heavy bucket", 0);
    // dense ids: descending, ascending (whether they fit depends on how much of the power of two above n they fill;
    // the verdict is checked where it is certain: n = 2^25)
    for (uint32_t i = 0; i < n; ++i) dup[i] = n - 1 - i;
    run(Mode::Keys, dup, iota, n, "msd window: descending", n == (1u << 25) ? RUNS : -1);
    run(Mode::KeyValue, dup, iota, n, "msd window: descending", n == (1u << 25) ? RUNS : -1);
    for (uint32_t i = 0; i < n; ++i) dup[i] = i;
    run(Mode::KeyValue, dup, iota, n, "msd window: ascending", n == (1u << 25) ? RUNS : -1);
    // keys that differ in their low ten bits only: the window does not go below bit 2, 256 buckets can hold something
    for (uint32_t i = 0; i < n; ++i) dup[i] = 0xABCDE000u | (g() & 0x3FFu);
    run(Mode::KeyValue, dup, iota, n, "msd: ten-bit keys");
    // all keys identical: nothing to do, and the plan says so
    std::fill(dup.begin(), dup.end(), 0x12345678u);
    run(Mode::KeyValue, dup, iota, n, "msd: all equal", SORTED);
    run(Mode::Keys, dup, iota, n, "msd: all equal", SORTED);
    dup[n - 2] = 0x12345679u;  // ... but one
    run(Mode::KeyValue, dup, iota, n, "msd declines: all equal but one", 0);
    // four distinct values
    const uint32_t four[4] = {3u, 0xFFFFFFFFu, 0x00010000u, 0x7F000000u};
    for (uint32_t i = 0; i < n; ++i) dup[i] = four[g() & 3];
    run(Mode::KeyValue, dup, iota, n, "msd declines: four values", 0);
  }
  std::printf("msd parity: %d cases, %d failures\n", cases, failures);
  return failures;
}

uint64_t Median(std::vector<uint64_t> v) {
  std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
  return v[v.size() / 2];
}

void Bench(Harness& h, const std::vector<int>& logs) {
  std::printf("%-10s %-6s %10s %10s %12s %10s %8s   [with 15 timestamps] stage ms (hist | scatter x4)\n", "n", "sort",
              "gpu_ms", "wall_ms", "GItems/s", "planGB/s", "%8TB/s");
  for (int lg : logs) {
    const uint32_t n = lg > 64 ? (uint32_t)lg : 1u << lg;  // an argument above 64 is the element count itself
    for (int kv = 0; kv < 2; ++kv) {
      VrdxSorterStorageRequirements req;
      if (kv)
        vrdxGetSorterKeyValueStorageRequirements(h.sorter, n, &req);
      else
        vrdxGetSorterStorageRequirements(h.sorter, n, &req);
      const uint32_t inout = Align16(n * 4u);
      h.reserve((size_t)2 * inout + 16, (size_t)req.size + StorageOffset());
      std::vector<uint64_t> gpu, wall, stage[5];
      VrdxHipPlanInfo plan;
      vrdxHipDescribePlan(h.sorter, n, kv, &plan);
      const bool msd = plan.plan == VRDX_HIP_PLAN_MSD;
      for (int runIdx = 0; runIdx < 11; ++runIdx) {  // 1 warm-up + 10 timed, fresh data each run
        std::vector<uint32_t> v;
        auto k = Mt(n, runIdx + 1, 32, &v);
        HIP_OK(hipMemcpy(h.dKeys, k.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(h.dKeys + inout, v.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_OK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        if (kv)
          vrdxCmdSortKeyValue((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dKeys, inout,
                              (VkBuffer)h.dStorage, StorageOffset(), h.pool, 0);
        else
          vrdxCmdSort((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dStorage, StorageOffset(), h.pool, 0);
        HIP_OK(hipStreamSynchronize(h.stream));
        const auto t1 = std::chrono::steady_clock::now();
        uint64_t ts[15];
        if (vrdxHipGetQueryPoolResults(h.pool, 0, 15, ts) != VK_SUCCESS) std::exit(3);
        if (runIdx == 0) continue;
        gpu.push_back(ts[14]);
        wall.push_back((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count());
        stage[0].push_back(ts[2] - ts[1]);
        if (msd) {  // the MSD plan's slots (include/vk_radix_sort.h): spine | scatter | buckets | the four returning passes
          stage[1].push_back(ts[3] - ts[2]);
          stage[2].push_back(ts[4] - ts[3]);
          stage[3].push_back(ts[5] - ts[4]);
          stage[4].push_back(ts[14] - ts[5]);
        } else {
          for (int p = 0; p < 4; ++p) stage[1 + p].push_back(ts[4 + 3 * p] - ts[3 + 3 * p]);
        }
      }
      // the same sort recorded WITHOUT a query pool (no per-stage events), bracketed by two events
      std::vector<uint64_t> bare;
      {
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        for (int runIdx = 0; runIdx < 11; ++runIdx) {
          std::vector<uint32_t> v;
          auto k = Mt(n, runIdx + 1, 32, &v);
          HIP_OK(hipMemcpy(h.dKeys, k.data(), (size_t)n * 4, hipMemcpyHostToDevice));
          HIP_OK(hipMemcpy(h.dKeys + inout, v.data(), (size_t)n * 4, hipMemcpyHostToDevice));
          HIP_OK(hipDeviceSynchronize());
          HIP_OK(hipEventRecord(e0, h.stream));
          if (kv)
            vrdxCmdSortKeyValue((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dKeys, inout,
                                (VkBuffer)h.dStorage, StorageOffset(), VK_NULL_HANDLE, 0);
          else
            vrdxCmdSort((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dStorage, StorageOffset(),
                        VK_NULL_HANDLE, 0);
          HIP_OK(hipEventRecord(e1, h.stream));
          HIP_OK(hipStreamSynchronize(h.stream));
          float msf = 0;
          HIP_OK(hipEventElapsedTime(&msf, e0, e1));
          if (runIdx > 0) bare.push_back((uint64_t)(msf * 1e6));
        }
        HIP_OK(hipEventDestroy(e0));
        HIP_OK(hipEventDestroy(e1));
      }
      const double ms = Median(bare) / 1e6;       // headline: no per-stage events inside the sort
      const double msStamped = Median(gpu) / 1e6; // with the 15-slot timestamp contract active
      const double bytes = (double)plan.bytesPerElement * n;  // the HBM bytes of the plan recorded for this size (vrdxHipDescribePlan; four passes: SURVEY.md section 8(d))
      const double gbps = bytes / (ms * 1e-3) / 1e9;
      std::printf("%-10u %-6s %10.4f %10.4f %12.3f %10.1f %7.1f%%   [stamped %.4f] %.4f %s %.4f %.4f %.4f %.4f\n", n,
                  kv ? "kv" : "keys", ms, Median(wall) / 1e6, n / (ms * 1e-3) / 1e9, gbps, 100.0 * gbps / 8000.0, msStamped,
                  Median(stage[0]) / 1e6, msd ? "| msd: spine scatter buckets fallback" : "|", Median(stage[1]) / 1e6,
                  Median(stage[2]) / 1e6, Median(stage[3]) / 1e6, Median(stage[4]) / 1e6);
      std::fflush(stdout);
    }
  }
}


// Every sample of the reference protocol at one size, not just the median: `rounds` times 1 warm-up + 10 sorts of fresh
// mt19937 data with the 15-slot timestamp contract, one line per sort (total and the stage intervals).  For telling a slow
// median apart: a slow size (every sample), a slow process (every sample of one round) or scattered slow sorts.
void Jitter(Harness& h, uint32_t n, bool kv, int rounds) {
  VrdxSorterStorageRequirements req;
  if (kv)
    vrdxGetSorterKeyValueStorageRequirements(h.sorter, n, &req);
  else
    vrdxGetSorterStorageRequirements(h.sorter, n, &req);
  const uint32_t inout = Align16(n * 4u);
  h.reserve((size_t)2 * inout + 16, (size_t)req.size + StorageOffset());
  VrdxHipPlanInfo plan;
  vrdxHipDescribePlan(h.sorter, n, kv, &plan);
  std::printf("# n=%u %s plan=%u bits=%u keys@%p storage@%p\n", n, kv ? "kv" : "keys", plan.plan, plan.bits, (void*)h.dKeys, (void*)h.dStorage);
  for (int round = 0; round < rounds; ++round)
    for (int runIdx = 0; runIdx < 11; ++runIdx) {
      std::vector<uint32_t> v;
      auto k = Mt(n, runIdx + 1, 32, &v);
      HIP_OK(hipMemcpy(h.dKeys, k.data(), (size_t)n * 4, hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(h.dKeys + inout, v.data(), (size_t)n * 4, hipMemcpyHostToDevice));
      HIP_OK(hipDeviceSynchronize());
      if (kv)
        vrdxCmdSortKeyValue((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dKeys, inout,
                            (VkBuffer)h.dStorage, StorageOffset(), h.pool, 0);
      else
        vrdxCmdSort((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dStorage, StorageOffset(), h.pool, 0);
      HIP_OK(hipStreamSynchronize(h.stream));
      uint64_t ts[15];
      if (vrdxHipGetQueryPoolResults(h.pool, 0, 15, ts) != VK_SUCCESS) std::exit(3);
      std::printf("%u %s round %d run %2d total %7.1f us | fill %5.1f hist %5.1f |", n, kv ? "kv" : "keys", round, runIdx, ts[14] / 1e3,
                  (ts[1] - ts[0]) / 1e3, (ts[2] - ts[1]) / 1e3);
      for (int i = 3; i < 15; ++i) std::printf(" %5.1f", (ts[i] - ts[i - 1]) / 1e3);
      std::printf("\n");
    }
  std::fflush(stdout);
}

// Timing-only size sweep for tile-geometry break points: N = 2^(lo + i*(hi-lo)/(points-1)), one
// random key/value set generated once (each run re-uploads the first N), median of 7 event-bracketed
// sorts without a query pool.
// linear = true: N = lo + i * (hi - lo) / (points - 1) element counts (the reference's sweep is linear: bench/bench.cc:17-20);
// otherwise lo and hi are log2 N.
void Sweep(Harness& h, double lo, double hi, int points, bool kv, bool linear = false) {
  const uint32_t nMax = linear ? (uint32_t)hi : (uint32_t)std::llround(std::pow(2.0, hi));
  std::vector<uint32_t> v;
  auto k = Mt(nMax, 7, 32, &v);
  VrdxSorterStorageRequirements req;
  vrdxGetSorterKeyValueStorageRequirements(h.sorter, nMax, &req);
  const uint32_t inoutMax = Align16(nMax * 4u);
  h.reserve((size_t)2 * inoutMax + 16, (size_t)req.size + StorageOffset());
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  std::printf("%-10s %-5s %10s %12s\n", "n", "sort", "gpu_ms", "GItems/s");
  for (int i = 0; i < points; ++i) {
    const double lg = points > 1 ? lo + (hi - lo) * i / (points - 1) : lo;
    const uint32_t n = linear ? (uint32_t)std::llround(lg) : (uint32_t)std::llround(std::pow(2.0, lg));
    const uint32_t inout = Align16(n * 4u);
    std::vector<uint64_t> t;
    for (int runIdx = 0; runIdx < 8; ++runIdx) {
      HIP_OK(hipMemcpy(h.dKeys, k.data(), (size_t)n * 4, hipMemcpyHostToDevice));
      if (kv) HIP_OK(hipMemcpy(h.dKeys + inout, v.data(), (size_t)n * 4, hipMemcpyHostToDevice));
      HIP_OK(hipDeviceSynchronize());
      HIP_OK(hipEventRecord(e0, h.stream));
      if (kv)
        vrdxCmdSortKeyValue((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dKeys, inout,
                            (VkBuffer)h.dStorage, StorageOffset(), VK_NULL_HANDLE, 0);
      else
        vrdxCmdSort((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dStorage, StorageOffset(), VK_NULL_HANDLE, 0);
      HIP_OK(hipEventRecord(e1, h.stream));
      HIP_OK(hipStreamSynchronize(h.stream));
      float msf = 0;
      HIP_OK(hipEventElapsedTime(&msf, e0, e1));
      if (runIdx > 0) t.push_back((uint64_t)(msf * 1e6));
    }
    const double ms = Median(t) / 1e6;
    std::printf("%-10u %-5s %10.4f %12.3f\n", n, kv ? "kv" : "keys", ms, n / (ms * 1e-3) / 1e9);
    std::fflush(stdout);
  }
  HIP_OK(hipEventDestroy(e0));
  HIP_OK(hipEventDestroy(e1));
}


// Per-pass times of `runs` sorts of 2^lg uniform keys (fresh mt19937 data for each), median / mean / min per stage from
// the 15-slot timestamp contract -- the per-pass parity table of profiles/r04_kv_pass_parity.txt.  Run under
// `rocprofv3 --kernel-trace` the same command gives the kernels' own durations (tools/pass_parity.py groups them).
void Passes(Harness& h, int lg, bool kv, int runs) {
  const uint32_t n = 1u << lg;
  const uint32_t inout = Align16(n * 4u);
  VrdxSorterStorageRequirements req;
  vrdxGetSorterKeyValueStorageRequirements(h.sorter, n, &req);
  h.reserve((size_t)2 * inout + 16, (size_t)req.size + StorageOffset());
  std::vector<uint64_t> stage[6];
  for (int run = 0; run <= runs; ++run) {
    std::vector<uint32_t> v;
    auto k = Mt(n, run + 1, 32, &v);
    HIP_OK(hipMemcpy(h.dKeys, k.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h.dKeys + inout, v.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipDeviceSynchronize());
    if (kv)
      vrdxCmdSortKeyValue((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dKeys, inout,
                          (VkBuffer)h.dStorage, StorageOffset(), h.pool, 0);
    else
      vrdxCmdSort((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dStorage, StorageOffset(), h.pool, 0);
    HIP_OK(hipStreamSynchronize(h.stream));
    uint64_t ts[15];
    if (vrdxHipGetQueryPoolResults(h.pool, 0, 15, ts) != VK_SUCCESS) std::exit(3);
    if (run == 0) continue;
    stage[0].push_back(ts[2] - ts[1]);
    for (int p = 0; p < 4; ++p) stage[1 + p].push_back(ts[4 + 3 * p] - ts[3 + 3 * p]);
    stage[5].push_back(ts[14]);
  }
  const char* names[6] = {"histogram", "pass 0", "pass 1", "pass 2", "pass 3", "whole sort"};
  std::printf("n=%u %s storage offset %llu, %d runs (us: median mean min)\n", n, kv ? "kv" : "keys",
              (unsigned long long)StorageOffset(), runs);
  for (int s = 0; s < 6; ++s) {
    double mean = 0;
    for (uint64_t t : stage[s]) mean += (double)t;
    mean /= (double)stage[s].size();
    std::printf("  %-10s %8.2f %8.2f %8.2f\n", names[s], Median(stage[s]) / 1e3, mean / 1e3,
                *std::min_element(stage[s].begin(), stage[s].end()) / 1e3);
  }
}

// `count` sorts of 2^lg uniform keys, each on its own resident array, enqueued BACK TO BACK with no host synchronisation in
// between (what bench.py's timed region does): prints the whole-batch rate.  Under `rocprofv3 --kernel-trace` the
// per-kernel durations show what a kernel costs BEHIND another sort (tools/pass_parity.py).
void BackToBack(Harness& h, int lg, bool kv, int count) {
  const uint32_t n = 1u << lg;
  const size_t inout = Align16(n * 4u);
  VrdxSorterStorageRequirements req;
  vrdxGetSorterKeyValueStorageRequirements(h.sorter, n, &req);
  h.reserve((size_t)count * 2 * inout + 16, (size_t)req.size);
  for (int i = 0; i < count; ++i) {
    std::vector<uint32_t> v;
    auto k = Mt(n, i + 1, 32, &v);
    HIP_OK(hipMemcpy(h.dKeys + (size_t)i * 2 * inout, k.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h.dKeys + (size_t)i * 2 * inout + inout, v.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  }
  HIP_OK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < count; ++i) {
    const VkDeviceSize at = (VkDeviceSize)i * 2 * inout;
    if (kv)
      vrdxCmdSortKeyValue((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, at, (VkBuffer)h.dKeys, at + inout,
                          (VkBuffer)h.dStorage, 0, VK_NULL_HANDLE, 0);
    else
      vrdxCmdSort((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, at, (VkBuffer)h.dStorage, 0, VK_NULL_HANDLE, 0);
  }
  HIP_OK(hipStreamSynchronize(h.stream));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("n=%u %s: %d sorts back to back in %.4f ms = %.4f ms per sort, %.2f GItems/s\n", n, kv ? "kv" : "keys", count, ms,
              ms / count, (double)n * count / (ms * 1e6));
}

// BASELINE.json configs[3]: N = 2^25 adversarial keys (all-equal, all-0xFFFFFFFF, descending,
// few-distinct) against uniform random; values = iota, parity checked with the oracle for the
// stable permutation, then timed (median of 5, data re-uploaded before every run).
int Adversarial(Harness& h, int lg) {
  const uint32_t n = 1u << lg;
  const uint32_t inout = Align16(n * 4u);
  VrdxSorterStorageRequirements req;
  vrdxGetSorterKeyValueStorageRequirements(h.sorter, n, &req);
  h.reserve((size_t)2 * inout + 16, (size_t)req.size + StorageOffset());
  std::vector<uint32_t> iota(n), k(n);
  for (uint32_t i = 0; i < n; ++i) iota[i] = i;
  std::mt19937 g(4);
  // the last two: every tile holds each digit of pass 0 exactly 128 times, so every run a tile writes in
  // pass 0 is 512 bytes -- starting on a 512-byte boundary ("balanced"), or 13 keys behind one
  // ("balanced+13": every run then begins and ends inside a 128-byte line).  Only pass 0 is comparable.
  // round 6: "24-bit" = DataGenerator::Generate(n, 24)-style keys (/root/reference/bench/data_generator.cc:15), "outlier" =
  // ascending keys with one key at the end that breaks their common prefix (the MSD plan's window is a guess from a sample:
  // the count must turn it down), "gaussian" = a sum of four uniform bytes in the top byte (mild skew: 1.5x the mean bucket)
  const char* names[] = {"uniform", "all-equal", "all-0xFFFFFFFF", "descending", "ascending", "few-distinct(4)",
                         "24-bit", "outlier", "gaussian", "balanced", "balanced+13"};
  constexpr int kPatterns = 11;
  double base[2] = {0, 0};
  int failures = 0;
  std::printf("%-18s %-5s %10s %12s %10s %-8s %-8s %9s\n", "keys", "sort", "gpu_ms", "GItems/s", "slowdown", "parity", "verdict", "unstamped");
  hipEvent_t bare0 = nullptr, bare1 = nullptr;
  HIP_OK(hipEventCreate(&bare0));
  HIP_OK(hipEventCreate(&bare1));
  const char* const patternsEnv = std::getenv("VRDX_SELFTEST_PATTERNS");  // the first k input patterns only
  const int patterns = patternsEnv != nullptr ? std::min(kPatterns, std::max(1, std::atoi(patternsEnv))) : kPatterns;
  for (int pattern = 0; pattern < patterns; ++pattern) {
    const uint32_t four[4] = {3u, 0xFFFFFFFFu, 0x00010000u, 0x7F000000u};
    for (uint32_t i = 0; i < n; ++i) {
      switch (pattern) {
        case 0: k[i] = g(); break;
        case 1: k[i] = 0x12345678u; break;
        case 2: k[i] = 0xFFFFFFFFu; break;
        case 3: k[i] = n - 1 - i; break;
        case 4: k[i] = i; break;
        case 5: k[i] = four[g() & 3]; break;
        case 6: k[i] = g() >> 8; break;
        case 7: k[i] = i + 1 < n ? i : 0x80000000u; break;
        case 8: {
          const uint32_t r = g(), r2 = g();
          k[i] = ((((r & 255u) + ((r >> 8) & 255u) + ((r >> 16) & 255u) + (r >> 24)) >> 2) << 24) | (r2 & 0x00FFFFFFu);
          break;
        }
        case 9: k[i] = (i & 0xFFu) * 0x01010101u; break;
        default: k[i] = i < 13 ? 0u : ((i - 13) & 0xFFu) * 0x01010101u; break;
      }
    }
    std::vector<uint32_t> ek = k, ev = iota;
    vrdx_oracle_sort(ek.data(), ev.data(), n, nullptr);
    for (int kv = 0; kv < 2; ++kv) {
      std::vector<uint64_t> times, bare;
      bool ok = true;
      uint32_t verdict = 0;
      double stageHist = 0, stagePass[4] = {0, 0, 0, 0};
      // runs 0-5: with the 15 timestamps (gpu_ms of run 0 is dropped; run 5 gives the stages); runs 6-10: WITHOUT them, two
      // events around the sort -- the figure that compares plans fairly (a plan with more stages carries more event records)
      for (int run = 0; run < 11; ++run) {
        HIP_OK(hipMemcpy(h.dKeys, k.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(h.dKeys + inout, iota.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_OK(hipDeviceSynchronize());
        const bool stamped = run < 6;
        if (!stamped) HIP_OK(hipEventRecord(bare0, h.stream));
        if (kv)
          vrdxCmdSortKeyValue((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dKeys, inout,
                              (VkBuffer)h.dStorage, StorageOffset(), stamped ? h.pool : VK_NULL_HANDLE, 0);
        else
          vrdxCmdSort((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dStorage, StorageOffset(),
                      stamped ? h.pool : VK_NULL_HANDLE, 0);
        if (!stamped) HIP_OK(hipEventRecord(bare1, h.stream));
        HIP_OK(hipStreamSynchronize(h.stream));
        if (!stamped) {
          float msBare = 0;
          HIP_OK(hipEventElapsedTime(&msBare, bare0, bare1));
          bare.push_back((uint64_t)(msBare * 1e6));
          continue;
        }
        uint64_t ts[15];
        if (vrdxHipGetQueryPoolResults(h.pool, 0, 15, ts) != VK_SUCCESS) return 3;
        if (run > 0) times.push_back(ts[14]);
        if (run == 5) {
          stageHist = (ts[2] - ts[1]) / 1e6;
          for (int p = 0; p < 4; ++p) stagePass[p] = (ts[4 + 3 * p] - ts[3 + 3 * p]) / 1e6;
        }
        if (run == 0) {
          std::vector<uint32_t> gk(n), gv(n);
          HIP_OK(hipMemcpy(gk.data(), h.dKeys, (size_t)n * 4, hipMemcpyDeviceToHost));
          HIP_OK(hipMemcpy(gv.data(), h.dKeys + inout, (size_t)n * 4, hipMemcpyDeviceToHost));
          ok = gk == ek && (!kv || gv == ev);
          if (vrdxHipReadStatus((VkCommandBuffer)h.stream, (VkBuffer)h.dStorage, StorageOffset()) != 0) ok = false;
          verdict = vrdxHipReadPlanVerdict((VkCommandBuffer)h.stream, (VkBuffer)h.dStorage, StorageOffset());
        }
      }
      const double ms = Median(times) / 1e6, msBare = Median(bare) / 1e6;
      if (pattern == 0) base[kv] = ms;
      if (!ok) ++failures;
      std::printf("%-18s %-5s %10.4f %12.3f %9.2fx %-8s %-8s %9.4f  hist %.4f | passes %.4f %.4f %.4f %.4f\n", names[pattern],
                  kv ? "kv" : "keys", ms, n / (ms * 1e-3) / 1e9, ms / base[kv], ok ? "ok" : "MISMATCH",
                  verdict == VRDX_HIP_VERDICT_MSD_RUNS ? "msd" : verdict == VRDX_HIP_VERDICT_MSD_SORTED ? "sorted" : "passes",
                  msBare, stageHist, stagePass[0], stagePass[1], stagePass[2], stagePass[3]);
      std::fflush(stdout);
    }
  }
  return failures;
}


// Soak: many sorts of random sizes and entropies back to back, every result checked bit for bit
// against the oracle -- hunts rare cross-workgroup races (status hand-off, ticket, LDS ranking)
// that a single pass over the parity battery could miss.  Two streams alternate so that sorts of
// different sizes overlap on the device.
// maxN: largest element count drawn (3 M by default; 40 M reaches the MSD plan with both bucket kernels, the tail split and block sums, at a
// few seconds of oracle time per sort).
int Soak(Harness& h, int seconds, uint32_t maxN = 3u << 20) {
  std::mt19937 g(12345);
  hipStream_t second;
  HIP_OK(hipStreamCreate(&second));
  const uint32_t inoutMax = Align16(maxN * 4u);
  VrdxSorterStorageRequirements req;
  vrdxGetSorterKeyValueStorageRequirements(h.sorter, maxN, &req);
  uint8_t* buf[2];
  uint8_t* sto[2];
  for (int i = 0; i < 2; ++i) {
    HIP_OK(hipMalloc((void**)&buf[i], (size_t)2 * inoutMax + 16));
    HIP_OK(hipMalloc((void**)&sto[i], (size_t)req.size));
  }
  hipStream_t streams[2] = {h.stream, second};
  const auto t0 = std::chrono::steady_clock::now();
  int sorts = 0, failures = 0;
  while (std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    std::vector<uint32_t> k[2], v[2];
    uint32_t n[2], inout[2];
    bool kv[2];
    for (int i = 0; i < 2; ++i) {
      const uint32_t r = g();
      n[i] = (r % 7 == 0) ? (g() % 20000) : (g() % maxN);
      const uint32_t bits = (g() % 4 == 0) ? (g() % 33) : 32;
      kv[i] = g() & 1;
      k[i].resize(n[i]);
      v[i].resize(n[i]);
      for (auto& x : k[i]) {
        const uint32_t y = g();
        x = bits >= 32 ? y : (bits == 0 ? 0u : y >> (32 - bits));
      }
      for (uint32_t j = 0; j < n[i]; ++j) v[i][j] = j;
      inout[i] = Align16(n[i] * 4u);
      HIP_OK(hipMemcpyAsync(buf[i], k[i].data(), (size_t)n[i] * 4, hipMemcpyHostToDevice, streams[i]));
      HIP_OK(hipMemcpyAsync(buf[i] + inout[i], v[i].data(), (size_t)n[i] * 4, hipMemcpyHostToDevice, streams[i]));
    }
    for (int i = 0; i < 2; ++i) {
      if (kv[i])
        vrdxCmdSortKeyValue((VkCommandBuffer)streams[i], h.sorter, n[i], (VkBuffer)buf[i], 0, (VkBuffer)buf[i], inout[i],
                            (VkBuffer)sto[i], 0, VK_NULL_HANDLE, 0);
      else
        vrdxCmdSort((VkCommandBuffer)streams[i], h.sorter, n[i], (VkBuffer)buf[i], 0, (VkBuffer)sto[i], 0, VK_NULL_HANDLE, 0);
    }
    for (int i = 0; i < 2; ++i) {
      HIP_OK(hipStreamSynchronize(streams[i]));
      std::vector<uint32_t> gk(n[i]), gv(n[i]);
      HIP_OK(hipMemcpy(gk.data(), buf[i], (size_t)n[i] * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(gv.data(), buf[i] + inout[i], (size_t)n[i] * 4, hipMemcpyDeviceToHost));
      const uint32_t status = n[i] ? vrdxHipReadStatus((VkCommandBuffer)streams[i], (VkBuffer)sto[i], 0) : 0;
      vrdx_oracle_sort(k[i].data(), kv[i] ? v[i].data() : nullptr, n[i], nullptr);
      const bool ok = status == 0 && gk == k[i] && (!kv[i] || gv == v[i]);
      if (!ok) {
        ++failures;
        std::printf("FAIL soak sort %d n=%u kv=%d status=%u\n", sorts, n[i], (int)kv[i], status);
      }
      ++sorts;
    }
  }
  std::printf("soak: %d sorts, %d failures\n", sorts, failures);
  return failures;
}

}  // namespace

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "parity";
  Harness h;
  h.init();
  std::printf("%s\n", vrdxHipVersionString());
  if (what == "parity" || what == "quick") return Parity(h, what == "quick") ? 1 : 0;
  if (what == "msd") {  // msd [n ...]
    std::vector<uint32_t> sizes;
    for (int i = 2; i < argc; ++i) sizes.push_back((uint32_t)std::strtoul(argv[i], nullptr, 10));
    return MsdParity(h, sizes) ? 1 : 0;
  }
  if (what == "trace") {  // one sort, for tools/trace.sh (stamps are dumped by vrdxDestroySorter)
    const uint32_t n = 1u << (argc > 2 ? std::atoi(argv[2]) : 25);
    const bool kv = argc > 3 && std::string(argv[3]) == "kv";
    VrdxSorterStorageRequirements req;
    vrdxGetSorterKeyValueStorageRequirements(h.sorter, n, &req);
    const uint32_t inout = Align16(n * 4u);
    h.reserve((size_t)2 * inout + 16, (size_t)req.size);
    for (int rep = 0; rep < 3; ++rep) {
      std::vector<uint32_t> v;
      auto k = Mt(n, rep + 1, 32, &v);
      const std::string pattern = argc > 4 ? argv[4] : "uniform";
      if (pattern == "ascending")
        for (uint32_t i = 0; i < n; ++i) k[i] = i;
      else if (pattern == "equal")
        std::fill(k.begin(), k.end(), 0x12345678u);
      HIP_OK(hipMemcpy(h.dKeys, k.data(), (size_t)n * 4, hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(h.dKeys + inout, v.data(), (size_t)n * 4, hipMemcpyHostToDevice));
      if (kv)
        vrdxCmdSortKeyValue((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dKeys, inout,
                            (VkBuffer)h.dStorage, 0, VK_NULL_HANDLE, 0);
      else
        vrdxCmdSort((VkCommandBuffer)h.stream, h.sorter, n, (VkBuffer)h.dKeys, 0, (VkBuffer)h.dStorage, 0, VK_NULL_HANDLE, 0);
      HIP_OK(hipStreamSynchronize(h.stream));
    }
    vrdxDestroySorter(h.sorter);
    return 0;
  }
  if (what == "sweep" || what == "lsweep") {  // sweep <lo log2> <hi log2> <points> [keys|kv]; lsweep <lo n> <hi n> <points> [keys|kv]
    Sweep(h, argc > 2 ? std::atof(argv[2]) : 22.0, argc > 3 ? std::atof(argv[3]) : 26.0, argc > 4 ? std::atoi(argv[4]) : 17,
          argc > 5 && std::string(argv[5]) == "kv", what == "lsweep");
    return 0;
  }
  if (what == "passes") {  // passes <log2n> [keys|kv] [runs]
    Passes(h, argc > 2 ? std::atoi(argv[2]) : 25, argc > 3 && std::string(argv[3]) == "kv", argc > 4 ? std::atoi(argv[4]) : 20);
    return 0;
  }
  if (what == "backtoback") {  // backtoback <log2n> [keys|kv] [sorts]
    BackToBack(h, argc > 2 ? std::atoi(argv[2]) : 25, argc > 3 && std::string(argv[3]) == "kv", argc > 4 ? std::atoi(argv[4]) : 10);
    return 0;
  }
  if (what == "soak")  // soak [seconds] [max elements]
    return Soak(h, argc > 2 ? std::atoi(argv[2]) : 30, argc > 3 ? (uint32_t)std::strtoul(argv[3], nullptr, 10) : 3u << 20) ? 1 : 0;
  if (what == "adversarial") return Adversarial(h, argc > 2 ? std::atoi(argv[2]) : 25) ? 1 : 0;
  if (what == "jitter") {  // jitter <n> [keys|kv] [rounds]
    Jitter(h, argc > 2 ? (uint32_t)std::strtoul(argv[2], nullptr, 10) : 1572864u, argc > 3 && std::string(argv[3]) == "kv",
           argc > 4 ? std::atoi(argv[4]) : 3);
    return 0;
  }
  if (what == "bench") {
    std::vector<int> logs;
    for (int i = 2; i < argc; ++i) logs.push_back(std::atoi(argv[i]));
    if (logs.empty()) logs = {18, 20, 22, 24, 25};
    Bench(h, logs);
    return 0;
  }
  std::fprintf(stderr, "usage: %s parity|quick|bench [log2n...]|adversarial [log2n]|soak [s]|sweep lo hi points [kv]|passes log2n [keys|kv] [runs]|trace\n", argv[0]);
  return 64;
}

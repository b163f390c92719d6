// Host recorder of the HIP backend: the eight vrdx* entry points of include/vk_radix_sort.h.
//
// Mirrors the reference's host side (src/vk_radix_sort.h.in:141-507) in behaviour:
//   * vrdxCreateSorter builds an immutable sorter and is the only call that can fail;
//   * the storage calculators are the reference's integer formulas, bit for bit;
//   * vrdxCmdSort* validate nothing, allocate nothing, never block the host: they append
//     stream-ordered work to the hipStream_t passed as VkCommandBuffer, the way gpuSort()
//     (:344-507) appends commands to a VkCommandBuffer.
// What differs is the recorded work: 1 clear + 1 fused histogram + 4 onesweep launches instead of
// 2 transfers + 12 dispatches + 12 barriers.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/vk_radix_sort.h"
#include "vrdx_kernels.h"
#include "vrdx_layout.h"

struct VrdxSorter_T {
  int device = 0;
  int computeUnits = 0;
  // (tile geometry is chosen per sort from the element count, see ConfigIndex)
  bool atomicRank = false;  // LDS returning atomics proven lane-ordered on this device
  // reference: VrdxSorter_T::minStorageBufferOffsetAlignment (src/vk_radix_sort.h.in:134)
  uint32_t minStorageBufferOffsetAlignment = VRDX_STORAGE_ALIGN;
};

struct VrdxHipQueryPool {
  uint32_t count = 0;
  hipEvent_t* events = nullptr;
  uint8_t* recorded = nullptr;
};

namespace {

int DeviceOrdinalFromHandle(const void* handle, int* ordinal) {
  if (handle == nullptr) return hipGetDevice(ordinal) == hipSuccess ? 0 : -1;
  *ordinal = (int)((uintptr_t)handle - 1);
  return 0;
}

// Tile geometry by problem size, measured on MI355X (profiles/r01_native_sweep_n.txt): small sorts
// want many small tiles (parallelism across 256 CUs); beyond that throughput grows monotonically
// with the tile (fewer look-backs per key, longer digit runs), up to the 32768 keys whose staging
// buffer still fits the CU's LDS.
//   index into vrdx::kTileConfigs: 0 = 512x16, 1 = 1024x16, 2 = 512x32, 3 = 1024x8, ... 7 = 1024x32
int ForcedConfigIndex() {
  static const int forced = [] {
    const char* env = std::getenv("VRDX_TILE_CONFIG");  // e.g. "512x16": one geometry for everything (tuning/testing)
    if (env == nullptr) return -1;
    for (int i = 0; i < vrdx::kNumTileConfigs; ++i) {
      char name[32];
      std::snprintf(name, sizeof(name), "%dx%d", vrdx::kTileConfigs[i].threads, vrdx::kTileConfigs[i].keysPerThread);
      if (std::strcmp(env, name) == 0) return i;
    }
    std::fprintf(stderr, "vrdx-hip: unknown VRDX_TILE_CONFIG '%s', using the defaults\n", env);
    return -1;
  }();
  return forced;
}

int ConfigIndex(bool keyValue, uint32_t elementCount) {
  const int forced = ForcedConfigIndex();
  if (forced >= 0) return forced;
  (void)keyValue;                             // the same break points serve keys-only and key+value
  if (elementCount <= (1u << 19)) return 3;   // 1024 x 8   (T = 8192: more tiles for 256 CUs)
  if (elementCount <= 6u << 20) return 1;     // 1024 x 16  (T = 16384, two workgroups per CU)
  return 7;                                   // 1024 x 32  (T = 32768, one 16-wave workgroup per CU:
                                              //             half the look-backs per key again)
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

inline uint8_t* BufferAddress(VkBuffer buffer, VkDeviceSize offset) {
  return reinterpret_cast<uint8_t*>(buffer) + offset;
}

// VRDX_DEBUG=1: report HIP errors of the enqueues on stderr (the entry points themselves return
// void and validate nothing, like the reference's vrdxCmd*).
bool DebugEnabled() {
  static const bool enabled = std::getenv("VRDX_DEBUG") != nullptr;
  return enabled;
}
void DebugCheck(const char* what) {
  if (!DebugEnabled()) return;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) std::fprintf(stderr, "vrdx-hip: %s -> %s\n", what, hipGetErrorString(e));
}

void Stamp(VrdxHipQueryPool* pool, uint32_t slot, hipStream_t stream) {
  if (pool == nullptr || slot >= pool->count) return;
  if (hipEventRecord(pool->events[slot], stream) == hipSuccess) pool->recorded[slot] = 1;
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
  if (elementCount > VRDX_MAX_ELEMENTS) elementCount = VRDX_MAX_ELEMENTS;

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

  const int configIndex = ConfigIndex(keyValue, elementCount);
  const uint32_t tileKeys = vrdx::kTileConfigs[configIndex].tileKeys();
  const vrdx::StorageLayout layout =
      vrdx::MakeLayout(elementCount, sorter->minStorageBufferOffsetAlignment, tileKeys);
  uint8_t* const storage = BufferAddress(storageBuffer, storageOffset);
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
    for (uint32_t s = 1; s < 15; ++s) Stamp(pool, query + s, stream);
    return;
  }

  // Clear count/tickets/failure word, the 4x256 global histogram (reference :382) and status
  // region 0 in one fill.  Indirect: also copy the device-side count to where the reference keeps
  // it (:368-379); the kernels themselves read it straight from the caller's buffer.  (Direct: the
  // count travels as a kernel argument, the slot stays 0 -- storage contents are scratch.)
  (void)hipMemsetAsync(storage, 0, layout.clearBytes, stream);
  DebugCheck("hipMemsetAsync(state)");
  if (countPtr != nullptr)
    (void)hipMemcpyAsync(storage + layout.countOffset, countPtr, sizeof(uint32_t),
                         hipMemcpyDeviceToDevice, stream);
  Stamp(pool, query + 1, stream);

  uint32_t* const globalHistogram = reinterpret_cast<uint32_t*>(storage + layout.histogramOffset);
  uint32_t* const status = reinterpret_cast<uint32_t*>(storage + layout.statusOffset);
  uint32_t* const tickets = reinterpret_cast<uint32_t*>(storage + layout.ticketOffset);
  uint32_t* const failure = reinterpret_cast<uint32_t*>(storage + layout.failureOffset);
  uint32_t* const keysScratch = reinterpret_cast<uint32_t*>(storage + layout.inoutOffset);
  uint32_t* const valuesScratch = reinterpret_cast<uint32_t*>(storage + layout.valuesOffset);
  const uint32_t statusRows = (uint32_t)layout.statusRows;

  // upsweep of all four passes at once
  {
    uint32_t grid = vrdx::RoundUp(elementCount, vrdx::kHistKeysPerTrip);
    const uint32_t cap = (uint32_t)sorter->computeUnits * vrdx::kHistWorkgroupsPerCu;
    if (grid > cap) grid = cap;
    if (grid == 0) grid = 1;
    vrdx::LaunchHistogram(stream, grid, keys, elementCount, countPtr, globalHistogram);
    DebugCheck("histogram_kernel");
  }

  const uint32_t tiles = vrdx::RoundUp(elementCount, tileKeys);
  for (uint32_t pass = 0; pass < VRDX_PASSES; ++pass) {
    Stamp(pool, query + 2 + 3 * pass + 0, stream);  // "upsweep" of this pass
    Stamp(pool, query + 2 + 3 * pass + 1, stream);  // "spine" (fused into the onesweep look-back)

    vrdx::OnesweepArgs args;
    // switch in->out to out->in for pass 1, pass 3 (reference :417-427)
    const bool odd = (pass & 1u) != 0;
    args.keysIn = odd ? keysScratch : keys;
    args.keysOut = odd ? keys : keysScratch;
    args.valuesIn = keyValue ? (odd ? valuesScratch : values) : nullptr;
    args.valuesOut = keyValue ? (odd ? values : valuesScratch) : nullptr;
    args.maxCount = elementCount;
    args.countPtr = countPtr;
    args.globalHistogram = globalHistogram + pass * VRDX_RADIX;
    args.statusCur = status + (size_t)(pass & 1u) * statusRows * VRDX_RADIX;
    args.statusNext =
        pass + 1 < VRDX_PASSES ? status + (size_t)((pass + 1) & 1u) * statusRows * VRDX_RADIX : nullptr;
    args.statusRows = statusRows;
    args.ticketCur = tickets + (pass & 1u);
    args.ticketNext = tickets + ((pass + 1) & 1u);
    args.failure = failure;
    args.shift = 8 * pass;
    args.trace = nullptr;
#ifdef VRDX_TRACE
    args.trace = TraceBuffer(pass, tiles);
#endif
    vrdx::LaunchOnesweep(stream, configIndex, tiles, keyValue, sorter->atomicRank, args);
    DebugCheck("onesweep_kernel");

    Stamp(pool, query + 2 + 3 * pass + 2, stream);  // "downsweep"
  }
  Stamp(pool, query + 14, stream);
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
      if (e == hipSuccess && !ordered && mode != nullptr && std::strcmp(mode, "atomic") == 0)
        std::fprintf(stderr, "vrdx-hip: VRDX_RANK=atomic refused, LDS atomics are not lane-ordered here\n");
    }
  }
  (void)hipSetDevice(previous);
  if (e != hipSuccess) {
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
  delete sorter;
}

void vrdxGetSorterStorageRequirements(VrdxSorter sorter, uint32_t maxElementCount,
                                      VrdxSorterStorageRequirements* requirements) {
  const vrdx::StorageLayout layout =
      vrdx::MakeLayout(maxElementCount, sorter->minStorageBufferOffsetAlignment,
                       vrdx::kTileConfigs[ConfigIndex(false, maxElementCount)].tileKeys());
  requirements->size = layout.keysOnlySize;
  requirements->usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT | VK_BUFFER_USAGE_TRANSFER_DST_BIT;
}

void vrdxGetSorterKeyValueStorageRequirements(VrdxSorter sorter, uint32_t maxElementCount,
                                              VrdxSorterStorageRequirements* requirements) {
  const vrdx::StorageLayout layout =
      vrdx::MakeLayout(maxElementCount, sorter->minStorageBufferOffsetAlignment,
                       vrdx::kTileConfigs[ConfigIndex(true, maxElementCount)].tileKeys());
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
  if (pool->events == nullptr || pool->recorded == nullptr) {
    delete[] pool->events;
    delete[] pool->recorded;
    delete pool;
    return VK_ERROR_OUT_OF_HOST_MEMORY;
  }
  std::memset(pool->recorded, 0, queryCount);
  for (uint32_t i = 0; i < queryCount; ++i) {
    if (hipEventCreate(&pool->events[i]) != hipSuccess) {
      for (uint32_t j = 0; j < i; ++j) (void)hipEventDestroy(pool->events[j]);
      delete[] pool->events;
      delete[] pool->recorded;
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
    const hipError_t e =
        hipEventElapsedTime(&ms, pool->events[firstQuery], pool->events[firstQuery + i]);
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

const char* vrdxHipVersionString(void) {
  static char text[128];
  const vrdx::TileConfig& k = vrdx::kTileConfigs[ConfigIndex(false, 1u << 25)];
  const vrdx::TileConfig& kv = vrdx::kTileConfigs[ConfigIndex(true, 1u << 25)];
  std::snprintf(text, sizeof(text), "vrdx-hip %d.%d.%d gfx950 tiles at 2^25: keys=%dx%d key-value=%dx%d%s",
                VRDX_VERSION_MAJOR, VRDX_VERSION_MINOR, VRDX_VERSION_PATCH, k.threads, k.keysPerThread, kv.threads,
                kv.keysPerThread, ForcedConfigIndex() >= 0 ? " (forced)" : " (size-adaptive)");
  return text;
}

}  // extern "C"

/*
 * vk_radix_sort.h -- MI355X (gfx950 / HIP) backend behind the VrdxSorter / vrdxCmdSort* surface.
 *
 * This is the drop-in boundary: a C-ABI shared library (libvrdx_hip.so) exporting the eight
 * entry points the reference declares at src/vk_radix_sort.h.in:24-81 (generated copy
 * include/vk_radix_sort.h:24-81), with the same names, parameter order and parameter types.
 * The reference is a header-only C++ library whose declarations have C++ linkage; here they are
 * `extern "C"` so that any FFI (ctypes, cgo, JNI, N-API) can bind them.
 *
 * Vulkan handle types keep their reference spelling.  When <vulkan/vulkan_core.h> is available it
 * is used; otherwise the minimal shim below supplies ABI-identical typedefs (every dispatchable
 * and non-dispatchable handle is an 8-byte pointer on 64-bit targets).
 *
 * HIP meaning of each handle:
 *   VkPhysicalDevice / VkDevice : HIP device ordinal, encoded with VRDX_HIP_DEVICE(ordinal);
 *                                 VK_NULL_HANDLE = the calling thread's current HIP device.
 *   VkPipelineCache             : ignored (kernels are precompiled for gfx950; no JIT).
 *   VkCommandBuffer             : hipStream_t.  "Recording" is a stream-ordered enqueue; nothing
 *                                 blocks the host, and the calls are legal inside
 *                                 hipStreamBeginCapture/EndCapture (hipGraph) when queryPool is NULL.
 *   VkBuffer + VkDeviceSize     : device pointer + byte offset.
 *   VkQueryPool                 : VrdxHipQueryPool (array of hipEvent_t), see vrdxHipCreateQueryPool.
 *
 * All arithmetic on this path is 32-bit unsigned integer; keys are sorted ascending, the sort is
 * stable, and results land back in keysBuffer / valuesBuffer.  How many trips through memory a sort
 * makes is the library's business and depends on its size (vrdxHipDescribePlan): one workgroup up to
 * 16384 elements; one scatter + one in-LDS sort per bucket (two trips) up to 67.1 M elements when the
 * device finds that every bucket fits; four ping-pong passes like the reference's otherwise.
 */
#ifndef VK_RADIX_SORT_H
#define VK_RADIX_SORT_H

#include <stdint.h>

#if defined(__has_include)
#if __has_include(<vulkan/vulkan_core.h>)
#include <vulkan/vulkan_core.h>
#define VRDX_HAVE_VULKAN_CORE 1
#endif
#endif

#ifndef VRDX_HAVE_VULKAN_CORE
/* ---- minimal Vulkan type shim (ABI-identical to vulkan_core.h on LP64) ---- */
#ifndef VK_DEFINE_HANDLE
#define VK_DEFINE_HANDLE(object) typedef struct object##_T* object;
#endif
#ifndef VK_NULL_HANDLE
#define VK_NULL_HANDLE 0
#endif
typedef struct VkPhysicalDevice_T* VkPhysicalDevice;
typedef struct VkDevice_T* VkDevice;
typedef struct VkPipelineCache_T* VkPipelineCache;
typedef struct VkCommandBuffer_T* VkCommandBuffer;
typedef struct VkBuffer_T* VkBuffer;
typedef struct VkQueryPool_T* VkQueryPool;
typedef uint64_t VkDeviceSize;
typedef uint32_t VkFlags;
typedef VkFlags VkBufferUsageFlags;
typedef enum VkResult {
  VK_SUCCESS = 0,
  VK_NOT_READY = 1,
  VK_ERROR_OUT_OF_HOST_MEMORY = -1,
  VK_ERROR_OUT_OF_DEVICE_MEMORY = -2,
  VK_ERROR_INITIALIZATION_FAILED = -3,
  VK_ERROR_DEVICE_LOST = -4,
  VK_ERROR_FEATURE_NOT_PRESENT = -8,
  VK_RESULT_MAX_ENUM = 0x7FFFFFFF
} VkResult;
#define VK_BUFFER_USAGE_TRANSFER_DST_BIT 0x00000002
#define VK_BUFFER_USAGE_STORAGE_BUFFER_BIT 0x00000020
#endif /* !VRDX_HAVE_VULKAN_CORE */

/* reference: src/vk_radix_sort.h.in:6-9 (v0.4.0, CMakeLists.txt:3) */
#define VRDX_VERSION_MAJOR 0
#define VRDX_VERSION_MINOR 4
#define VRDX_VERSION_PATCH 0
#define VRDX_VERSION ((VRDX_VERSION_MAJOR << 22) | (VRDX_VERSION_MINOR << 12) | VRDX_VERSION_PATCH)

/* HIP device ordinal <-> VkDevice / VkPhysicalDevice encoding (ordinal + 1 so that 0 stays NULL). */
#define VRDX_HIP_DEVICE(ordinal) ((VkDevice)(uintptr_t)((ordinal) + 1))
#define VRDX_HIP_PHYSICAL_DEVICE(ordinal) ((VkPhysicalDevice)(uintptr_t)((ordinal) + 1))

#ifdef __cplusplus
extern "C" {
#endif

struct VrdxSorter_T;

/* reference: src/vk_radix_sort.h.in:11-16 -- VrdxSorter owns the (precompiled) kernels' launch
 * configuration for one device; immutable after creation. */
VK_DEFINE_HANDLE(VrdxSorter)

/* reference: src/vk_radix_sort.h.in:18-22 */
typedef struct VrdxSorterCreateInfo {
  VkPhysicalDevice physicalDevice;
  VkDevice device;
  VkPipelineCache pipelineCache;
} VrdxSorterCreateInfo;

/* reference: src/vk_radix_sort.h.in:24,141-265.  Returns VK_SUCCESS, or
 * VK_ERROR_INITIALIZATION_FAILED (no usable HIP device / ordinal out of range),
 * VK_ERROR_FEATURE_NOT_PRESENT (device is not gfx950), VK_ERROR_OUT_OF_HOST_MEMORY.
 * On failure *pSorter is left untouched and nothing is leaked (reference cleanup(), :153-158). */
VkResult vrdxCreateSorter(const VrdxSorterCreateInfo* pCreateInfo, VrdxSorter* pSorter);

/* reference: src/vk_radix_sort.h.in:26,267-277.  NULL-safe. */
void vrdxDestroySorter(VrdxSorter sorter);

/* reference: src/vk_radix_sort.h.in:28-31 */
typedef struct VrdxSorterStorageRequirements {
  VkDeviceSize size;
  VkBufferUsageFlags usage;
} VrdxSorterStorageRequirements;

/* reference: src/vk_radix_sort.h.in:33,279-292.  Same formula, bit for bit:
 *   size = Align(4,A) + HistogramSize(N,A) + InoutSize(N,A), usage = STORAGE_BUFFER|TRANSFER_DST. */
void vrdxGetSorterStorageRequirements(VrdxSorter sorter, uint32_t maxElementCount,
                                      VrdxSorterStorageRequirements* requirements);

/* reference: src/vk_radix_sort.h.in:36,294-308.
 *   size = Align(4,A) + HistogramSize + Align(InoutSize,A) + InoutSize. */
void vrdxGetSorterKeyValueStorageRequirements(VrdxSorter sorter, uint32_t maxElementCount,
                                              VrdxSorterStorageRequirements* requirements);

/**
 * reference: src/vk_radix_sort.h.in:39-54,310-315.
 *
 * if queryPool is not VK_NULL_HANDLE, it records timestamps into N entries [query..query+N-1].
 *
 * N=15 (same slot contract as the reference):
 * query + 0: start
 * query + 1: after the state clear ("transfer")
 * query + 2 + (3 * i) + 0: upsweep of pass i
 * query + 2 + (3 * i) + 1: spine of pass i
 * query + 2 + (3 * i) + 2: downsweep of pass i
 * query + 14: sort end
 *
 * What lies between two slots here, by the plan vrdxHipDescribePlan reports (slots that have no stage of
 * their own coincide with the slot before them, so every difference is >= 0 and ts[14] - ts[0] is the sort):
 *   FOUR_PASSES   [1,2] the fused 4-digit histogram; [3 i + 3, 3 i + 4] pass i (rank + look-back + scatter in
 *                 one kernel: the "spine" slot 3 i + 3 coincides with the "upsweep" slot 3 i + 2, and that one
 *                 with the previous pass's "downsweep" slot for i > 0)
 *   HYBRID8       the same, plus [4,5] = one workgroup per bucket (pass 1's "upsweep"); passes 1-3 return at once
 *                 when the device takes the plan
 *   MSD           [1,2] histogram (it also chooses the window); [2,3] spine; [3,4] scatter by the
 *                 window bits; [4,5] one workgroup per bucket; the passes that are launches of their own follow
 *                 and return at once when the device takes the plan: [9,10] pass 2, [12,13] pass 3, and [6,7] pass 1
 *                 where it is a launch (sorts of up to 18.1 M elements: the half-size bucket kernel has no second
 *                 role).  Passes 0 and 1 are otherwise second roles of the scatter and bucket launches -- when the
 *                 device turns the plan down, [3,4] and [4,5] ARE passes 0 and 1 (with VRDX_MSD_FUSED=0 in the
 *                 environment they are launches of their own again, both inside [6,7])
 *   ONE_WORKGROUP [13,14] the one kernel
 */
void vrdxCmdSort(VkCommandBuffer commandBuffer, VrdxSorter sorter, uint32_t elementCount,
                 VkBuffer keysBuffer, VkDeviceSize keysOffset, VkBuffer storageBuffer,
                 VkDeviceSize storageOffset, VkQueryPool queryPool, uint32_t query);

/* reference: src/vk_radix_sort.h.in:55-58,317-323 */
void vrdxCmdSortIndirect(VkCommandBuffer commandBuffer, VrdxSorter sorter, uint32_t maxElementCount,
                         VkBuffer indirectBuffer, VkDeviceSize indirectOffset, VkBuffer keysBuffer,
                         VkDeviceSize keysOffset, VkBuffer storageBuffer,
                         VkDeviceSize storageOffset, VkQueryPool queryPool, uint32_t query);

/* reference: src/vk_radix_sort.h.in:60-63,325-331 */
void vrdxCmdSortKeyValue(VkCommandBuffer commandBuffer, VrdxSorter sorter, uint32_t elementCount,
                         VkBuffer keysBuffer, VkDeviceSize keysOffset, VkBuffer valuesBuffer,
                         VkDeviceSize valuesOffset, VkBuffer storageBuffer,
                         VkDeviceSize storageOffset, VkQueryPool queryPool, uint32_t query);

/**
 * reference: src/vk_radix_sort.h.in:65-81,333-342.
 *
 * indirectBuffer contains elementCount: a uint32_t read on the device from
 * indirectBuffer + indirectOffset when the sort executes.  It must not exceed maxElementCount
 * (values above it are clamped).  Keys, values and the count may live in one buffer at different
 * offsets (bench/vulkan_benchmark.cc:386-388).
 */
void vrdxCmdSortKeyValueIndirect(VkCommandBuffer commandBuffer, VrdxSorter sorter,
                                 uint32_t maxElementCount, VkBuffer indirectBuffer,
                                 VkDeviceSize indirectOffset, VkBuffer keysBuffer,
                                 VkDeviceSize keysOffset, VkBuffer valuesBuffer,
                                 VkDeviceSize valuesOffset, VkBuffer storageBuffer,
                                 VkDeviceSize storageOffset, VkQueryPool queryPool, uint32_t query);

/* ------------------------------------------------------------------------------------------
 * HIP-side companions of the Vulkan objects the reference's callers create themselves
 * (vkCreateQueryPool / vkGetQueryPoolResults, bench/vulkan_benchmark.cc:195-198,318-321).
 * They are not part of the reference API; they exist so a caller without a Vulkan device can
 * still use the 15-slot timestamp contract.
 * ------------------------------------------------------------------------------------------ */

/* Creates a pool of `queryCount` timestamp slots (hipEvent_t each) on the current device. */
VkResult vrdxHipCreateQueryPool(uint32_t queryCount, VkQueryPool* pQueryPool);
void vrdxHipDestroyQueryPool(VkQueryPool queryPool);
/* After the stream has completed: pData[i] = nanoseconds between slot firstQuery and slot
 * firstQuery+i (so pData[0] == 0; timestampPeriod == 1.0).  Returns VK_NOT_READY if a slot was
 * never recorded or has not completed. */
VkResult vrdxHipGetQueryPoolResults(VkQueryPool queryPool, uint32_t firstQuery, uint32_t queryCount,
                                    uint64_t* pData);

/* Device-side failure word of the LAST sort recorded with this storage (recording a sort clears the
 * word): 0 = ok.  A non-zero value means a bounded look-back spin expired (the GPU never hangs; the
 * output is then unspecified).  Synchronises the given stream.  Diagnostic only. */
uint32_t vrdxHipReadStatus(VkCommandBuffer commandBuffer, VkBuffer storageBuffer,
                           VkDeviceSize storageOffset);

/* The same diagnosis for EVERY sort recorded with this sorter since the previous call (or since
 * vrdxCreateSorter), whatever storage they used: the OR of their failure bits, kept in a device
 * word the sorter owns; reading it clears it.  This is what a caller that runs many sorts through
 * one storage buffer checks once at the end.  Bit 31 is the host side of it: set when the runtime
 * refused one of the sort's enqueues (fill, copy, kernel launch) -- the vrdxCmdSort* entry points return
 * void, so this is where such an error surfaces.  Synchronises the given stream (which must belong to
 * the sorter's device and be ordered after the sorts in question). */
uint32_t vrdxHipReadSorterStatus(VrdxSorter sorter, VkCommandBuffer commandBuffer);
/* Bits of that word (and of the one line vrdxDestroySorter prints on stderr when a sorter is destroyed with any of
 * them never read -- the vrdxCmdSort* entry points return void like the reference's, so nothing fails silently): */
#define VRDX_HIP_STATUS_LOOKBACK_GAVE_UP 0x00000001u /* a bounded look-back spin expired: that sort's result is unspecified */
#define VRDX_HIP_STATUS_RANK_ORDER       0x00000002u /* the periodic repeat of the LDS lane-order check failed: vrdxHipRecheck */
#define VRDX_HIP_STATUS_COUNT_CLAMPED    0x40000000u /* elementCount > 2^30 - 4 (where the reference's uint32 size math wraps, src/vk_radix_sort.h.in:105-115): the first 2^30 - 4 elements were sorted, the rest left alone */
#define VRDX_HIP_STATUS_ENQUEUE_REFUSED  0x80000000u /* the HIP runtime refused a fill, copy or launch of a sort */

/* Repeats, synchronously (~1 ms), the device check vrdxCreateSorter ran for the one-atomic ranking (LDS returning atomics
 * served in lane order: measured on every MI355X so far, not promised by the ISA manual) and switches the sorter to the
 * ballot ranking for the sorts recorded from then on if it fails (one line on stderr).  A small stream-ordered repeat
 * is recorded by the library itself behind every 65536th sort (VRDX_HIP_STATUS_RANK_ORDER); call this after a driver or
 * firmware update under a long-lived process, or whenever that bit shows up.  Not thread-safe against sorts being
 * recorded with the same sorter at the same time only in the sense that those may still use the old ranking. */
VkResult vrdxHipRecheck(VrdxSorter sorter);

/* What a pair of hipEventRecords adds to the kernel between them on this stream, in nanoseconds: the median over eight
 * runs of (event interval - the time a spinning kernel demonstrably ran by the device's own wall clock).  Slot
 * differences of the 15-slot timestamp contract include this much on top of the kernel's duration; bench.py subtracts
 * it to report kernel time (synchronises the stream; ~0.5 ms).  ~0 on failure is not assumed: returns UINT64_MAX then. */
uint64_t vrdxHipEventOverheadNs(VkCommandBuffer commandBuffer);

/* Which plan vrdxCmdSort* records for a sort of `elementCount` elements with this sorter (it depends on the count, the
 * device's CU count and the ranking mode only), and the HBM bytes per element that plan moves when the device lets it
 * run -- what a roofline figure for the whole sort has to be priced with.  Plans other than FOUR_PASSES and
 * ONE_WORKGROUP are taken or turned down ON THE DEVICE (a bucket that does not fit a workgroup's LDS: skewed keys), in
 * which case the four passes recorded behind them run and `fallbackBytesPerElement` applies. */
#define VRDX_HIP_PLAN_NONE 0u          /* elementCount == 0: nothing is recorded */
#define VRDX_HIP_PLAN_ONE_WORKGROUP 1u /* <= 16384 elements: one launch, one load and one store of the data */
#define VRDX_HIP_PLAN_FOUR_PASSES 2u   /* fused histogram + four onesweep passes: 36 B/key, 68 B/pair */
#define VRDX_HIP_PLAN_HYBRID8 3u       /* histogram + one scatter by the highest varying byte + one in-LDS sort per bucket */
#define VRDX_HIP_PLAN_HYBRID9 4u       /* round 4's nine-bit hybrid plan: no longer built (the MSD plan took its sizes); never reported, the value stays reserved */
#define VRDX_HIP_PLAN_MSD 5u           /* histogram + scatter by a 10 / 11-bit window (chosen on the device below the keys' common prefix) + up to two in-LDS passes per bucket */
typedef struct VrdxHipPlanInfo {
  uint32_t plan;                    /* VRDX_HIP_PLAN_* */
  uint32_t bits;                    /* digits of the plan's scatter through memory (8, 9, 10, 11; 0 otherwise) */
  uint32_t bytesPerElement;         /* algorithmic HBM bytes per element when the plan runs */
  uint32_t fallbackBytesPerElement; /* ... when the device turns it down (== bytesPerElement for plans decided on the host) */
  uint32_t launches;                /* kernel launches recorded (fills and copies not counted) */
} VrdxHipPlanInfo;
void vrdxHipDescribePlan(VrdxSorter sorter, uint32_t elementCount, int keyValue, VrdxHipPlanInfo* info);

/* What the DEVICE made of the plan of the last sort that used this storage (word 1 of the storage; synchronises the
 * stream): vrdxHipDescribePlan is the host's intention, this is the verdict.  Meaningful for the two plans that are
 * decided on the device; a sort that ran the four passes leaves one of the "turned down" values. */
#define VRDX_HIP_VERDICT_NONE 0u            /* no plan was taken: the four passes ran (also: FOUR_PASSES / ONE_WORKGROUP sorts) */
#define VRDX_HIP_VERDICT_HYBRID8_RUNS 1u    /* HYBRID8: launch 0 scattered by the highest varying byte, the buckets were sorted in LDS */
#define VRDX_HIP_VERDICT_HYBRID8_DECLINED 2u /* HYBRID8: a bucket did not fit, the four passes ran */
#define VRDX_HIP_VERDICT_MSD_RUNS 3u        /* MSD: scatter + bucket launches sorted the input (bytesPerElement applies) */
#define VRDX_HIP_VERDICT_MSD_SORTED 4u      /* MSD: all keys identical -- nothing was moved */
uint32_t vrdxHipReadPlanVerdict(VkCommandBuffer commandBuffer, VkBuffer storageBuffer, VkDeviceSize storageOffset);

/* How many sorts this sorter recorded with the MSD plan in front (*pRecorded, counted on the host when the sort is
 * recorded) and how many of those the device turned down (*pDeclined, counted on the device when the sort runs: a bucket
 * beyond the capacity, a key outside the sampled prefix, a sample that rules the plan out) -- such sorts run the four
 * passes and cost 1.3-1.5x the plan.  Uniform keys at the very top of a plan's size range are turned down with small
 * probability by design (3-4 % of headroom); a caller that sees the ratio rise knows its keys are skewed.  (A sort captured
 * into a hipGraph is recorded once and may run many times: every replay that is turned down counts.)  Synchronises the
 * stream; either pointer may be NULL. */
VkResult vrdxHipReadPlanCounters(VrdxSorter sorter, VkCommandBuffer commandBuffer, uint32_t* pRecorded, uint32_t* pDeclined);

/* Library build info: "vrdx-hip <version> gfx950 tiles at 2^25: keys=<threads>x<keys per thread>[x<sub-tiles>]
 * key-value=... (size-adaptive | forced)". */
const char* vrdxHipVersionString(void);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* VK_RADIX_SORT_H */

/* The reference is a single-header library activated with VRDX_IMPLEMENTATION
 * (src/vk_radix_sort.h.in:85-86, bench/vrdx_impl.cc:1-4).  The HIP backend lives in
 * libvrdx_hip.so, so the macro is accepted and ignored to keep such translation units compiling. */
#ifdef VRDX_IMPLEMENTATION
#undef VRDX_IMPLEMENTATION
#endif

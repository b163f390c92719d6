/*
 * vrdx_oracle.c -- CPU restatement of the reference's sort algorithm.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library (oracle/liboracle.so).  Nothing under vulkan_radix_sort_amd/ links, imports or calls it:
 * the product path is the HIP library and fails loudly when that is missing.
 *
 * What is restated (paths relative to /root/reference):
 *   - storage-size integer math ........ src/vk_radix_sort.h.in:105-115, 279-308
 *   - the recorded pass structure ...... src/vk_radix_sort.h.in:344-507 (4 passes, ping-pong,
 *                                        global histogram cleared once, partition count from N)
 *   - upsweep ........................... src/shader/upsweep.slang:10-45
 *   - spine ............................. src/shader/spine.slang:11-84
 *   - downsweep (+ KEY_VALUE build) ..... src/shader/downsweep.slang:41-224
 *
 * The GPU shaders cannot run in the authoring container (no Vulkan, no slangc), so this oracle is
 * pinned differently (see oracle/README.md): it is checked against oracle/_ref -- the reference's
 * OWN correctness predicate (bench/bench.cc:41-64), i.e. bench/cpu_benchmark.cc compiled from
 * /root/reference -- on every golden fixture and on randomized inputs (tests/test_oracle.py).
 *
 * Everything is uint32 arithmetic, exactly like the shaders.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RADIX 256u               /* src/shader/constants.slang:1 */
#define WORKGROUP_SIZE 512u      /* :2 */
#define PARTITION_DIVISION 8u    /* :3 */
#define PARTITION_SIZE (PARTITION_DIVISION * WORKGROUP_SIZE) /* :4 -> 4096 */

/* src/vk_radix_sort.h.in:105-106 */
static uint32_t RoundUp(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static uint32_t Align(uint32_t a, uint32_t b) { return (a + b - 1) / b * b; }

/* src/vk_radix_sort.h.in:108-111 */
static uint64_t HistogramSize(uint32_t elementCount, uint32_t align) {
  return Align((4 + 4 * RADIX + RoundUp(elementCount, PARTITION_SIZE) * RADIX) * (uint32_t)sizeof(uint32_t),
               align);
}

/* src/vk_radix_sort.h.in:113-115 */
static uint64_t InoutSize(uint32_t elementCount, uint32_t align) {
  return Align(elementCount * (uint32_t)sizeof(uint32_t), align);
}

/* src/vk_radix_sort.h.in:279-292 (keyValue == 0) and :294-308 (keyValue != 0) */
uint64_t vrdx_oracle_storage_size(uint32_t maxElementCount, uint32_t align, int keyValue) {
  uint64_t elementCountSize = Align((uint32_t)sizeof(uint32_t), align);
  uint64_t histogramSize = HistogramSize(maxElementCount, align);
  uint64_t inoutSize = InoutSize(maxElementCount, align);
  uint64_t histogramOffset = elementCountSize;
  uint64_t inoutOffset = histogramOffset + histogramSize;
  if (!keyValue) return inoutOffset + inoutSize;
  return inoutOffset + Align((uint32_t)inoutSize, align) + inoutSize;
}

/* VK_BUFFER_USAGE_STORAGE_BUFFER_BIT | VK_BUFFER_USAGE_TRANSFER_DST_BIT (:291, :307) */
uint32_t vrdx_oracle_storage_usage(void) { return 0x00000020u | 0x00000002u; }

/* Offsets gpuSort carves out of the storage buffer (src/vk_radix_sort.h.in:353-362, 405-415).
 * out[0]=elementCountOffset out[1]=histogramOffset out[2]=partitionHistogramOffset
 * out[3]=inoutOffset out[4]=valuesInoutOffset out[5]=partitionCount */
void vrdx_oracle_storage_offsets(uint32_t elementCount, uint32_t align, uint64_t out[6]) {
  uint64_t elementCountSize = Align((uint32_t)sizeof(uint32_t), align);
  uint64_t histogramSize = HistogramSize(elementCount, align);
  uint64_t inoutSize = Align(elementCount * (uint32_t)sizeof(uint32_t), align);
  out[0] = 0;
  out[1] = elementCountSize;
  out[2] = out[1] + sizeof(uint32_t) * 4 * RADIX;
  out[3] = out[1] + histogramSize;
  out[4] = out[3] + Align((uint32_t)inoutSize, align);
  out[5] = RoundUp(elementCount, PARTITION_SIZE);
}

/* upsweep.slang:10-45 for one pass: per-partition digit histogram (padding keys 0xffffffff ARE
 * counted, :32-34) and accumulation into globalHistogram[RADIX*pass + d] (:43). */
static void upsweep(const uint32_t* keys, uint32_t elementCount, uint32_t partitionCount, uint32_t pass,
                    uint32_t* globalHistogram, uint32_t* partitionHistogram) {
  for (uint32_t partitionIndex = 0; partitionIndex < partitionCount; ++partitionIndex) {
    uint32_t partitionStart = partitionIndex * PARTITION_SIZE;
    if (partitionStart >= elementCount) continue; /* :20 */
    uint32_t localHistogram[RADIX];
    memset(localHistogram, 0, sizeof(localHistogram));
    for (uint32_t k = 0; k < PARTITION_SIZE; ++k) {
      uint32_t keyIndex = partitionStart + k;
      uint32_t key = keyIndex < elementCount ? keys[keyIndex] : 0xffffffffu;
      uint32_t radix = (key >> (8 * pass)) & 0xffu;
      localHistogram[radix] += 1;
    }
    for (uint32_t d = 0; d < RADIX; ++d) {
      partitionHistogram[RADIX * partitionIndex + d] = localHistogram[d];
      globalHistogram[RADIX * pass + d] += localHistogram[d];
    }
  }
}

/* spine.slang:11-84: partitionHistogram[p][d] <- sum_{q<p} partitionHistogram[q][d] (in place),
 * and globalHistogram[pass][d] <- sum_{e<d} globalHistogram[pass][e] (workgroup 0, :62-83). */
static void spine(uint32_t elementCount, uint32_t pass, uint32_t* globalHistogram,
                  uint32_t* partitionHistogram) {
  uint32_t partitionCount = RoundUp(elementCount, PARTITION_SIZE); /* :25 */
  for (uint32_t d = 0; d < RADIX; ++d) {
    uint32_t reduction = 0;
    for (uint32_t p = 0; p < partitionCount; ++p) {
      uint32_t v = partitionHistogram[RADIX * p + d];
      partitionHistogram[RADIX * p + d] = reduction;
      reduction += v;
    }
  }
  uint32_t run = 0;
  for (uint32_t d = 0; d < RADIX; ++d) {
    uint32_t v = globalHistogram[RADIX * pass + d];
    globalHistogram[RADIX * pass + d] = run;
    run += v;
  }
}

/* downsweep.slang:41-224.  Inside a partition the shader's rank is
 *   exclusive-scan over (digit, wave) counters + counts of earlier slots + lower lanes (:92-176),
 * and the key of (wave w, slot i, lane l) sits at partitionStart + 8*laneCount*w + i*laneCount + l
 * (:79-80): (w, i, l) order IS memory order, so the partition-local sorted position of a key is
 *   #(keys of the partition with a smaller digit) + #(earlier keys of the partition, same digit).
 * dst = globalHistogram[pass][d] + partitionHistogram[p][d] - localExclusive[d] + localPos
 * (:179-183,198) and the store is suppressed when dst >= elementCount (:199,220). */
static void downsweep(const uint32_t* keysIn, uint32_t* keysOut, const uint32_t* valuesIn, uint32_t* valuesOut,
                      uint32_t elementCount, uint32_t partitionCount, uint32_t pass,
                      const uint32_t* globalHistogram, const uint32_t* partitionHistogram) {
  uint32_t* localKeys = (uint32_t*)malloc(sizeof(uint32_t) * PARTITION_SIZE);
  uint32_t* localValues = (uint32_t*)malloc(sizeof(uint32_t) * PARTITION_SIZE);
  for (uint32_t partitionIndex = 0; partitionIndex < partitionCount; ++partitionIndex) {
    uint32_t partitionStart = partitionIndex * PARTITION_SIZE;
    if (partitionStart >= elementCount) continue; /* :58 */

    uint32_t count[RADIX], localExclusive[RADIX], cursor[RADIX];
    memset(count, 0, sizeof(count));
    for (uint32_t k = 0; k < PARTITION_SIZE; ++k) {
      uint32_t keyIndex = partitionStart + k;
      uint32_t key = keyIndex < elementCount ? keysIn[keyIndex] : 0xffffffffu; /* :81 */
      count[(key >> (8 * pass)) & 0xffu] += 1;
    }
    uint32_t run = 0;
    for (uint32_t d = 0; d < RADIX; ++d) {
      localExclusive[d] = run;
      cursor[d] = run;
      run += count[d];
    }
    /* rearrange into the LDS-sorted order (:186-192) */
    for (uint32_t k = 0; k < PARTITION_SIZE; ++k) {
      uint32_t keyIndex = partitionStart + k;
      uint32_t key = keyIndex < elementCount ? keysIn[keyIndex] : 0xffffffffu;
      uint32_t value = (valuesIn && keyIndex < elementCount) ? valuesIn[keyIndex] : 0u; /* :85 */
      uint32_t pos = cursor[(key >> (8 * pass)) & 0xffu]++;
      localKeys[pos] = key;
      localValues[pos] = value;
    }
    /* binning (:195-206, :217-223) */
    for (uint32_t i = 0; i < PARTITION_SIZE; ++i) {
      uint32_t key = localKeys[i];
      uint32_t radix = (key >> (8 * pass)) & 0xffu;
      uint32_t dstOffset = globalHistogram[RADIX * pass + radix] +
                           partitionHistogram[RADIX * partitionIndex + radix] - localExclusive[radix] + i;
      if (dstOffset < elementCount) {
        keysOut[dstOffset] = key;
        if (valuesIn) valuesOut[dstOffset] = localValues[i];
      }
    }
  }
  free(localKeys);
  free(localValues);
}

/*
 * gpuSort (src/vk_radix_sort.h.in:344-507) on the CPU.  Sorts keys[0..elementCount) (and values,
 * when values != NULL) in place, ascending, stable.  Elements at index >= elementCount are not
 * touched.  Returns 0, or -1 when scratch memory cannot be allocated.
 *
 * If globalHistogramOut != NULL it receives the 4x256 globalHistogram table as the reference
 * leaves it after the sort (each row exclusive-scanned by spine).
 */
int vrdx_oracle_sort(uint32_t* keys, uint32_t* values, uint32_t elementCount, uint32_t* globalHistogramOut) {
  uint32_t partitionCount = RoundUp(elementCount, PARTITION_SIZE); /* :353 */
  uint32_t globalHistogram[4 * RADIX];
  memset(globalHistogram, 0, sizeof(globalHistogram)); /* vkCmdFillBuffer, :382 */
  if (globalHistogramOut) memset(globalHistogramOut, 0, sizeof(globalHistogram));
  if (elementCount == 0) return 0;

  uint32_t* partitionHistogram = (uint32_t*)malloc(sizeof(uint32_t) * RADIX * (size_t)partitionCount);
  uint32_t* keysScratch = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)elementCount);
  uint32_t* valuesScratch = values ? (uint32_t*)malloc(sizeof(uint32_t) * (size_t)elementCount) : NULL;
  if (!partitionHistogram || !keysScratch || (values && !valuesScratch)) {
    free(partitionHistogram);
    free(keysScratch);
    free(valuesScratch);
    return -1;
  }

  for (uint32_t pass = 0; pass < 4; ++pass) { /* :400 */
    /* switch in->out to out->in for pass 1, pass 3 (:417-427) */
    const uint32_t* keysIn = (pass % 2 == 1) ? keysScratch : keys;
    uint32_t* keysOut = (pass % 2 == 1) ? keys : keysScratch;
    const uint32_t* valuesIn = values ? ((pass % 2 == 1) ? valuesScratch : values) : NULL;
    uint32_t* valuesOut = values ? ((pass % 2 == 1) ? values : valuesScratch) : NULL;

    upsweep(keysIn, elementCount, partitionCount, pass, globalHistogram, partitionHistogram); /* :446-448 */
    spine(elementCount, pass, globalHistogram, partitionHistogram);                           /* :463-465 */
    downsweep(keysIn, keysOut, valuesIn, valuesOut, elementCount, partitionCount, pass, globalHistogram,
              partitionHistogram);                                                            /* :480-487 */
  }

  if (globalHistogramOut) memcpy(globalHistogramOut, globalHistogram, sizeof(globalHistogram));
  free(partitionHistogram);
  free(keysScratch);
  free(valuesScratch);
  return 0;
}

/* Raw (un-scanned) digit counts of the valid keys for all four passes: what a fused 4-digit
 * histogram produces.  Equals upsweep's globalHistogram contribution minus the padding keys. */
void vrdx_oracle_digit_counts(const uint32_t* keys, uint32_t elementCount, uint32_t out[4 * 256]) {
  memset(out, 0, sizeof(uint32_t) * 4 * RADIX);
  for (uint32_t i = 0; i < elementCount; ++i)
    for (uint32_t pass = 0; pass < 4; ++pass) out[RADIX * pass + ((keys[i] >> (8 * pass)) & 0xffu)] += 1;
}

/* 64-bit FNV-1a over little-endian uint32 words: fixture checksums for large outputs. */
uint64_t vrdx_oracle_hash(const uint32_t* data, uint64_t count) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint64_t i = 0; i < count; ++i) {
    uint32_t w = data[i];
    for (int b = 0; b < 4; ++b) {
      h ^= (uint8_t)(w >> (8 * b));
      h *= 0x100000001b3ull;
    }
  }
  return h;
}

// Storage layout and tile constants shared by the host recorder and the device kernels.
//
// The caller allocates exactly vrdxGetSorter[KeyValue]StorageRequirements().size bytes, and that
// number is the reference's formula bit for bit (src/vk_radix_sort.h.in:105-115,279-308), so our
// device state has to live inside the regions the reference carves (src/vk_radix_sort.h.in:353-362):
//
//   byte offset from storageOffset (A = 16)         reference use             our use
//   [0, 4)                                          element count             element count (direct)
//   [4, 12)                                         (padding)                 unused
//   [12, 16)                                        (padding)                 failure word
//   [A, A+4096)                                     globalHistogram[4][256]   globalHistogram[4][256] (raw counts)
//   [A+4096, A+4096+P*1024), P = ceil(N/4096)       partitionHistogram[P][256] tile status[2][rows][256], then
//                                                                              (>= 1 KiB is left) tile tickets[2]
//   16 B                                            (slack of the "4 +" term) unused
//   inoutOffset = A + HistogramSize(N)              keys scratch uint[N]      keys scratch uint[N]
//   inoutOffset + Align(InoutSize, A)               values scratch uint[N]    values scratch uint[N]
//
// Tile status: one 32-bit word {flag:2, value:30} per (tile, digit).  A tile of VRDX_TILE keys
// publishes its per-digit count (flag AGGREGATE) and later its inclusive prefix over all tiles
// up to itself (flag INCLUSIVE).  The last tile is never looked at, so rows = tiles - 1, and two
// regions (pass p uses region p & 1 and zeroes its own row of the other one for pass p + 1) need
// 2 * (ceil(N/TILE) - 1) KiB <= P KiB, which holds for every N when TILE >= 8192.
// N <= 2^30 - 4 (the reference's uint32 byte-size math wraps above that, :105-115) keeps every
// prefix inside 30 bits.
//
// The two tile tickets (one per pass parity) sit 512 bytes behind the second status region, in a
// cache line of their own: every workgroup of a pass hits its ticket with a device-scope atomic, and
// anything else in that 128-byte line becomes slow to READ meanwhile -- with the tickets in the
// padding at [4, 12) the line was shared with the first 28 counts of the global histogram, and every
// pass that read them took 3 us longer (measured: 30.0 vs 27.0 us per pass at N = 2^23).  They are
// outside the cleared prefix; the histogram kernel zeroes them.
#ifndef VRDX_LAYOUT_H
#define VRDX_LAYOUT_H

#include <stdint.h>

#define VRDX_RADIX 256u
#define VRDX_PASSES 4u
#define VRDX_REF_PARTITION_SIZE 4096u /* reference PARTITION_SIZE: only fixes the storage formula */
#define VRDX_STORAGE_ALIGN 16u        /* minStorageBufferOffsetAlignment the reference sees on desktop GPUs */

#define VRDX_FLAG_SHIFT 30u
#define VRDX_VALUE_MASK 0x3FFFFFFFu
#define VRDX_FLAG_EMPTY 0u
#define VRDX_FLAG_AGGREGATE 1u
#define VRDX_FLAG_INCLUSIVE 2u

#define VRDX_MAX_ELEMENTS 0x3FFFFFFCu /* 2^30 - 4: beyond it the reference's uint32 Align(4 * N, 16) wraps (:105-115) */

/* offsets inside the first 16 bytes */
#define VRDX_OFF_COUNT 0u
#define VRDX_OFF_FAILURE 12u
// word 1 of the storage (zeroed with the rest of the prefix before every sort): the hybrid plan's verdict, written by
// launch 0 -- 1 = the plan applies (launches 1-3 return at once), 2 = the four passes run (the bucket launch returns)
#define VRDX_OFF_PLAN 4u

#ifdef __cplusplus
namespace vrdx {

// reference: src/vk_radix_sort.h.in:105-106 (uint32 arithmetic on purpose)
static inline uint32_t RoundUp(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static inline uint32_t Align(uint32_t a, uint32_t b) { return (a + b - 1) / b * b; }

// reference: src/vk_radix_sort.h.in:108-111
static inline uint64_t HistogramSize(uint32_t elementCount, uint32_t align) {
  return Align((4 + 4 * VRDX_RADIX + RoundUp(elementCount, VRDX_REF_PARTITION_SIZE) * VRDX_RADIX) *
                   (uint32_t)sizeof(uint32_t),
               align);
}

// reference: src/vk_radix_sort.h.in:113-115
static inline uint64_t InoutSize(uint32_t elementCount, uint32_t align) {
  return Align(elementCount * (uint32_t)sizeof(uint32_t), align);
}

struct StorageLayout {
  uint64_t countOffset;      // element count word
  uint64_t ticketOffset;     // uint32[2], zeroed by the histogram kernel
  uint64_t failureOffset;    // uint32
  uint64_t histogramOffset;  // uint32[4][256]
  uint64_t statusOffset;     // uint32[2][rows][256]
  uint64_t statusRows;       // rows per region = max(tiles - 1, 0)
  uint64_t clearBytes;       // bytes from storageOffset zeroed before every sort
  uint64_t inoutOffset;      // keys scratch
  uint64_t valuesOffset;     // values scratch (KV only)
  uint64_t keysOnlySize;     // total storage, keys-only
  uint64_t keyValueSize;     // total storage, key-value
};

static inline StorageLayout MakeLayout(uint32_t maxElementCount, uint32_t align, uint32_t tileKeys) {
  StorageLayout l;
  const uint64_t elementCountSize = Align((uint32_t)sizeof(uint32_t), align);
  const uint64_t histogramSize = HistogramSize(maxElementCount, align);
  const uint64_t inoutSize = InoutSize(maxElementCount, align);
  l.countOffset = VRDX_OFF_COUNT;
  l.failureOffset = VRDX_OFF_FAILURE;
  l.histogramOffset = elementCountSize;
  l.statusOffset = l.histogramOffset + VRDX_PASSES * VRDX_RADIX * sizeof(uint32_t);
  const uint64_t tiles = ((uint64_t)maxElementCount + tileKeys - 1) / tileKeys;
  l.statusRows = tiles > 0 ? tiles - 1 : 0;
  // count + failure + global histogram + status region 0
  l.clearBytes = l.statusOffset + l.statusRows * VRDX_RADIX * sizeof(uint32_t);
  // 2 * rows KiB <= P - 1 KiB for every tile >= 8192 keys: at least 1 KiB is free behind region 1
  l.ticketOffset = l.statusOffset + 2 * l.statusRows * VRDX_RADIX * sizeof(uint32_t) + 512;
  l.inoutOffset = l.histogramOffset + histogramSize;
  l.valuesOffset = l.inoutOffset + Align((uint32_t)inoutSize, align);
  l.keysOnlySize = l.inoutOffset + inoutSize;
  l.keyValueSize = l.valuesOffset + inoutSize;
  return l;
}

}  // namespace vrdx
#endif

#endif  // VRDX_LAYOUT_H

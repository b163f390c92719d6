// Storage layout and tile constants shared by the host recorder and the device kernels.
//
// The caller allocates exactly vrdxGetSorter[KeyValue]StorageRequirements().size bytes, and that
// number is the reference's formula bit for bit (src/vk_radix_sort.h.in:105-115,279-308).  What the
// bytes hold is unspecified scratch from the caller's point of view, so only the SIZE is kept; the
// carving differs from the reference's (src/vk_radix_sort.h.in:353-362) in one respect since round 4:
// everything a kernel streams through is placed on a 128-byte boundary of the ABSOLUTE address
// (S = storage buffer + storageOffset, a multiple of 16 by the reference's contract, README.md:150).
//
//   byte offset from S (A = 16)                     reference use              our use
//   [0, 4)                                          element count              element count (indirect: copied)
//   [4, 8)                                          (padding)                  the plan's verdict (VRDX_OFF_PLAN)
//   [8, 12)                                         (padding)                  MSD plan: window, kind of input, "turned down" (VRDX_OFF_MSD_OVERFLOW)
//   [12, 16)                                        (padding)                  failure word
//   [A, A+4096)                                     globalHistogram[4][256]    the same table (raw counts)
//   first 128-byte boundary at or behind A+4096     partitionHistogram[P][256] MSD plan only: uint32 bucketCount[2^bits] (zeroed with
//                                                                              the prefix), bucketBase[2^bits], then uint16
//                                                                              tileCounts[msdTiles][2^bits]
//   statusOffset = behind them                                                 tile status[2][rows][256]
//   ticketOffset = end of the status regions                                   tile tickets[2], a 128-byte line of their own
//   inoutOffset  = ticketOffset + 128               keys scratch uint[N]       keys scratch uint[N]     (128-byte aligned)
//   valuesOffset = inoutOffset + Align(4 N, 128)    values scratch uint[N]     values scratch uint[N]   (128-byte aligned)
//
// Why: the reference's inoutOffset = 4128 + P * 1024 is 32 bytes past a 128-byte line for every N, so every
// wave-striped 256-byte load of a pass that reads the scratch arrays (passes 1 and 3) touched three lines instead of
// two: key+value passes 1 and 3 took 118-120 us instead of 113 at N = 2^25 (profiles/r04_kv_pass_parity.txt; keys-only
// sorts do not care).  It fits: the two status regions need 2 * rows KiB <= (P - 1) KiB (see below), so at least
// 1 KiB of the reference's partition-histogram area is free, of which the three pads (<= 124 + 112 bytes) and the
// ticket line (128) take at most 364 bytes.
//
// Tile status: one 32-bit word {flag:2, value:30} per (tile, digit).  A tile publishes its per-digit count
// (flag AGGREGATE) and later its inclusive prefix over all tiles up to itself (flag INCLUSIVE).  The last
// tile is never looked at, so rows = tiles - 1, and two regions (pass p uses region p & 1 and zeroes its own
// row of the other one for pass p + 1) need 2 * (tiles - 1) KiB <= (P - 1) KiB, P = ceil(N / 4096): true for
// every N when no tile is smaller than 8192 keys, and for the tail tiles of 4096 keys and more that a sort of
// one round and more may end with (PlanTiles in vrdx_api.cpp; checked by tests/native/layout_check.cpp).
// N <= 2^30 - 4 (the reference's uint32 byte-size math wraps above that, :105-115) keeps every
// prefix inside 30 bits.
//
// The two tile tickets (one per pass parity) have a 128-byte line to themselves: every workgroup of a pass hits
// its ticket with a device-scope atomic, and anything else in that line becomes slow to READ meanwhile -- with
// the tickets in the padding at [4, 12) the line was shared with the first 28 counts of the global histogram,
// and every pass that read them took 3 us longer (measured: 30.0 vs 27.0 us per pass at N = 2^23).  They are
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

/* block sums (sorts of one round): one row per VRDX_BLOCK_TILES tiles, word = arrivals << VRDX_BLOCK_COUNT_SHIFT | sum of counts */
#define VRDX_BLOCK_TILES 32u
#define VRDX_BLOCK_COUNT_SHIFT 26u
#define VRDX_BLOCK_SUM_MASK 0x03FFFFFFu
#define VRDX_BLOCK_PREFIX_MIN_TILES 64u

#define VRDX_MAX_ELEMENTS 0x3FFFFFFCu /* 2^30 - 4: beyond it the reference's uint32 Align(4 * N, 16) wraps (:105-115) */

/* offsets inside the first 16 bytes */
#define VRDX_OFF_COUNT 0u
#define VRDX_OFF_FAILURE 12u
// word 1 of the storage (zeroed with the rest of the prefix before every sort): the plan's verdict (vrdxHipReadPlanVerdict).
// Eight-bit hybrid plan, written by launch 0: 1 = the plan applies (launches 1-3 return at once), 2 = the four passes run (the
// bucket launch returns).  MSD plan, written by its scatter launch: 3 = the plan runs (| the window's shift << 8), 4 = all
// keys identical, nothing to do; the passes behind return on either.
#define VRDX_OFF_PLAN 4u
// word 2 (zeroed likewise): the MSD plan's word -- its low byte is raised by the spine when a bucket exceeds the capacity, by
// the histogram kernel for a key outside the sampled prefix or a sample that rules the plan out: the plan's scatter and
// bucket launches then return and the four passes run; above it the window's shift and the kind of input
// (vrdx_kernels.h, kMsdDecline* / kMsdShiftShift / kMsdModeShift)
#define VRDX_OFF_MSD_OVERFLOW 8u

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
  uint64_t msdBucketOffset;  // MSD plan only, on the first 128-byte line behind the table: uint32 bucketCount[2^msdBits] (inside the
                             //   prefix the fill zeroes: the histogram kernel adds into it), then bucketBase[2^msdBits]
  uint64_t msdCountsOffset;  // MSD plan only: uint16[msdTiles][2^msdBits] per-tile counts / prefixes of the keys' window bits,
  uint64_t msdCountsBytes;   //   behind them, in front of status region 0
  uint64_t statusOffset;     // uint32[2][rows + blockRows][256]: tile rows, then block rows, per region
  uint64_t statusRows;       // tile rows per region = max(tiles - 1, 0)
  uint64_t blockRows;        // block-sum rows per region (one per 32 tiles); 0 = classic look-back
  uint64_t regionBytes;      // (statusRows + blockRows) KiB
  uint64_t clearBytes;       // bytes from storageOffset zeroed by the fill in front of every sort: header + global histogram
  uint64_t statusClearOffset; // what the histogram kernel (MSD plan: the spine kernel) zeroes: status region 0
  uint64_t statusClearBytes;  // (pass 0 is its first reader)
  uint64_t inoutOffset;      // keys scratch
  uint64_t valuesOffset;     // values scratch (KV only)
  uint64_t keysOnlySize;     // total storage, keys-only
  uint64_t keyValueSize;     // total storage, key-value
};

// tiles: status rows are sized for this many tiles (PlanTiles); storageAddress: the absolute address of the storage
// (buffer + storageOffset) -- only its low 7 bits matter; the sizes do not depend on it.
// blockSums: the sort uses block sums (PlanTiles said so): one more row per 32 tiles in each status region.
// msdBits / msdTiles: the MSD plan is recorded in front of the passes (10 | 11 bits, tiles of 32768 keys; 0: it is not).
static inline StorageLayout MakeLayout(uint32_t maxElementCount, uint32_t align, uint64_t tiles,
                                       uint64_t storageAddress = 0, bool blockSums = false, uint32_t msdBits = 0,
                                       uint64_t msdTiles = 0) {
  StorageLayout l;
  const uint64_t elementCountSize = Align((uint32_t)sizeof(uint32_t), align);
  const uint64_t histogramSize = HistogramSize(maxElementCount, align);
  const uint64_t inoutSize = InoutSize(maxElementCount, align);
  l.countOffset = VRDX_OFF_COUNT;
  l.failureOffset = VRDX_OFF_FAILURE;
  l.histogramOffset = elementCountSize;
  const uint64_t tableEnd = l.histogramOffset + VRDX_PASSES * VRDX_RADIX * sizeof(uint32_t);
  l.msdBucketOffset = tableEnd + ((0 - (storageAddress + tableEnd)) & 127u);  // the first 128-byte line behind the table
  const uint64_t bucketWords = msdBits != 0 ? ((uint64_t)1 << msdBits) : 0;
  l.msdCountsOffset = l.msdBucketOffset + 8 * bucketWords;                      // behind the two bucket tables
  l.msdCountsBytes = msdBits != 0 ? msdTiles * ((uint64_t)2 << msdBits) : 0;  // 16 bits per (tile, bucket)
  l.statusOffset = l.msdCountsOffset + l.msdCountsBytes;  // (a multiple of 128 bytes like the rest)
  l.statusRows = tiles > 0 ? tiles - 1 : 0;
  l.blockRows = blockSums ? (tiles + VRDX_BLOCK_TILES - 1) / VRDX_BLOCK_TILES : 0;
  l.regionBytes = (l.statusRows + l.blockRows) * VRDX_RADIX * sizeof(uint32_t);
  // count + plan word + failure word + global histogram: what the histogram kernel's atomics and the passes' first
  // reads need zeroed BEFORE that kernel starts (status region 0 is zeroed by the histogram kernel itself, or by the MSD
  // plan's spine kernel); with the MSD plan also the bucket sizes, which its histogram kernel adds up with global atomics
  l.clearBytes = msdBits != 0 ? l.msdBucketOffset + 4 * bucketWords : tableEnd;
  l.statusClearOffset = l.statusOffset;
  l.statusClearBytes = l.regionBytes;
  l.ticketOffset = l.statusOffset + 2 * l.regionBytes;
  l.inoutOffset = l.ticketOffset + 128;
  l.valuesOffset = l.inoutOffset + (((uint64_t)maxElementCount * sizeof(uint32_t) + 127u) & ~(uint64_t)127u);
  // the reference's totals (what the caller allocates)
  const uint64_t refInoutOffset = l.histogramOffset + histogramSize;
  l.keysOnlySize = refInoutOffset + inoutSize;
  l.keyValueSize = refInoutOffset + Align((uint32_t)inoutSize, align) + inoutSize;
  return l;
}

// Everything the layout places lies inside what the caller allocated (the key+value size is the larger one; a keys-only
// sort never touches the values scratch).
static inline bool LayoutFits(const StorageLayout& l, uint32_t maxElementCount) {
  const uint64_t bytes = (uint64_t)maxElementCount * sizeof(uint32_t);
  return l.inoutOffset + bytes <= l.keysOnlySize && l.valuesOffset + bytes <= l.keyValueSize;
}

// The MSD plan's scatter (scatter_msd_kernel, one workgroup per CU and tile) cuts the sort into EQUAL tiles that fill whole
// rounds of `cus` tiles: keys per tile, a multiple of 4096 (four 64-key slots per wave of its 1024 threads), at most 32768.
// 520 tiles of 32768 keys would cost three rounds, the third for eight tiles; 768 tiles of 24576 cost three rounds of three
// quarters the length.  maxTiles: the spine kernel's reach; beyond it, tiles of full capacity.
// twoPerWorkgroup (round 6: keys-only sorts by ten bits, whose scatter takes two consecutive tiles per workgroup): the tiles
// fill whole rounds of `cus` PAIRS -- 2 x rounds x cus tiles -- so that a sort of 8.4 M keys is 256 workgroups of two tiles of
// 16384 and not 128 of two of 32768 on half the CUs.
static inline uint32_t MsdTileKeysFor(uint32_t elementCount, uint32_t cus, uint32_t maxTiles, bool twoPerWorkgroup = false) {
  if (elementCount == 0 || cus == 0) return 32768u;
  const uint32_t perWorkgroup = twoPerWorkgroup ? 2u : 1u;
  const uint32_t rounds = RoundUp(elementCount, cus * 32768u * perWorkgroup);
  uint32_t keys = 4096u * RoundUp(RoundUp(elementCount, rounds * cus * perWorkgroup), 4096u);
  if (keys > 32768u) keys = 32768u;
  return RoundUp(elementCount, keys) <= maxTiles ? keys : 32768u;
}

// Tile plan.  The 1024x32 kernels hold one workgroup per CU, so a sort of T full tiles takes ceil(T / CUs) rounds
// however full the last round is: 257 tiles cost two rounds, and so do 512.  Two remedies, both through the kernels'
// run-time slot counts (OnesweepArgs::slots / fullTiles / tailSlots, SpanOfTile in vrdx_kernels.hip):
//  * even split (round 3): a sort of no more than one round is cut into `cus` EQUAL tiles (a multiple of four 64-key
//    slots per wave) instead of tiles of the kernel's capacity;
//  * tail split (round 4): a longer sort keeps its whole rounds of full tiles, and what is left over -- less than one
//    round -- is cut into `cus` equal small tiles, so that the last round costs what its keys cost instead of a full
//    tile's life (profiles/r04_tail_split.txt).
// Pure integer math, shared with tests/native/layout_check.cpp (every plan's status rows must fit the storage).
struct TilePlan {
  uint32_t tiles;      // grid size = status rows + 1
  uint32_t slots;      // 0: every tile holds the kernel's capacity (the kernels without run-time slot counts)
  uint32_t fullTiles;  // tiles [0, fullTiles) hold `slots` slots per wave (and sub-tile), the rest tailSlots
  uint32_t tailSlots;
  bool blockSums;      // one round of 64 ... cus tiles of 32768 keys and more: block sums instead of the look-back chain
};

// Block sums pay when all tiles of a pass start together, i.e. in sorts of one round (BlockPrefix in vrdx_kernels.hip).
static inline bool BlockSumsApply(uint32_t tiles, uint32_t cus, uint32_t tileKeys) {
  // (BlockPrefix adds up at most 2 x 16 block rows: 32 blocks of VRDX_BLOCK_TILES tiles, whatever the CU count)
  return tileKeys >= 32768u && tiles >= VRDX_BLOCK_PREFIX_MIN_TILES && tiles <= cus && tiles <= 32u * VRDX_BLOCK_TILES;
}

// threads / keysPerThread / subTiles: the kernel's geometry; splitForms: its forms with run-time slot counts exist;
// evenSplit: allowed; tailPercent: tail split while the rest is at most this share of a round (0: never).
static inline TilePlan PlanTiles(uint32_t elementCount, uint32_t cus, uint32_t threads, uint32_t keysPerThread,
                                 uint32_t subTiles, bool splitForms, bool evenSplit, uint32_t tailPercent) {
  const uint32_t slotKeys = threads * subTiles;     // keys of a tile per slot
  const uint32_t capacity = slotKeys * keysPerThread;
  TilePlan plan;
  plan.tiles = RoundUp(elementCount, capacity);
  plan.slots = 0;
  plan.fullTiles = ~0u;
  plan.tailSlots = 0;
  plan.blockSums = BlockSumsApply(plan.tiles, cus, capacity);
  if (elementCount == 0 || !splitForms || cus == 0) return plan;
  const uint32_t granule = 4u * slotKeys;  // the kernels walk four slots at a time
  const uint64_t round = (uint64_t)cus * capacity;
  const uint32_t wholeRounds = (uint32_t)(elementCount / round);
  if (wholeRounds == 0) {
    if (!evenSplit) return plan;
    uint32_t slots = 4u * RoundUp(RoundUp(elementCount, cus), granule);
    if (slots < 8u) slots = 8u;  // 2 * (tiles - 1) <= P - 1 needs tiles of 8192 keys and more when all tiles are alike
    if (slots >= keysPerThread) return plan;
    plan.slots = plan.tailSlots = slots;
    plan.tiles = RoundUp(elementCount, slots * slotKeys);
    plan.blockSums = BlockSumsApply(plan.tiles, cus, slots * slotKeys);
    return plan;
  }
  if (tailPercent == 0) return plan;
  const uint32_t fullTiles = wholeRounds * cus;
  const uint32_t rest = elementCount - fullTiles * capacity;
  if (rest == 0 || (uint64_t)rest * 100u > (uint64_t)tailPercent * round) return plan;
  const uint32_t tailSlots = 4u * RoundUp(RoundUp(rest, cus), granule);
  if (tailSlots >= keysPerThread) return plan;
  plan.slots = keysPerThread;
  plan.fullTiles = fullTiles;
  plan.tailSlots = tailSlots;
  plan.tiles = fullTiles + RoundUp(rest, tailSlots * slotKeys);
  plan.blockSums = false;  // more than one round
  return plan;
}

}  // namespace vrdx
#endif

#endif  // VRDX_LAYOUT_H

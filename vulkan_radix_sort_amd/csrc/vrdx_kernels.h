// Launch interface between the host recorder (vrdx_api.cpp) and the device kernels
// (vrdx_kernels.hip).  Plain pointers and integers only.
#ifndef VRDX_KERNELS_H
#define VRDX_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace vrdx {

#ifndef VRDX_HIST_THREADS
#define VRDX_HIST_THREADS 1024
#endif
#ifndef VRDX_HIST_COPIES
#define VRDX_HIST_COPIES 8
#endif
#ifndef VRDX_HIST_WGS_PER_CU
#define VRDX_HIST_WGS_PER_CU 1
#endif
constexpr uint32_t kHistThreads = VRDX_HIST_THREADS;
constexpr uint32_t kHistCopies = VRDX_HIST_COPIES;
constexpr uint32_t kHistWorkgroupsPerCu = VRDX_HIST_WGS_PER_CU;
#ifndef VRDX_HIST_COPIES_LARGE
#define VRDX_HIST_COPIES_LARGE 32
#endif
constexpr uint32_t kHistCopiesLarge = VRDX_HIST_COPIES_LARGE;  // sorts of kHistManyCopiesFrom keys and more
constexpr uint32_t kHistManyCopiesFrom = 1u << 24;
// key+value sorts of more than kStreamingLoadsAbove and at most kStreamingLoadsUpTo elements read their
// tiles with non-temporal loads (vrdx_kernels.hip, StreamingLoads): 16 B per element = 1x ... 3x the 256 MiB
// Infinity Cache of an MI355X
constexpr uint32_t kHistStreamingLoadsAbove = 1u << 25;  // histogram: keys (4 B each) of more than half that cache
constexpr uint32_t kStreamingLoadsAbove = 1u << 24;
constexpr uint32_t kStreamingLoadsUpTo = 3u << 24;
constexpr uint32_t HistLdsBytes(uint32_t copies) { return 4u * 256u * copies * 4u; }  // [pass][digit][copy]
// keys one histogram workgroup counts per GROUP (kHistThreads lanes x four 16-byte loads x 4 keys); every wave keeps
// two groups of loads in flight (vrdx_kernels.hip)
constexpr uint32_t kHistGroupKeys = kHistThreads * 4 * 4;

struct TileConfig {
  int threads;
  int keysPerThread;
  int subTiles;  // 1: onesweep_kernel; 2: onesweep_pair_kernel (two sub-tiles per workgroup and status row)
  uint32_t tileKeys() const { return (uint32_t)threads * (uint32_t)keysPerThread * (uint32_t)subTiles; }
};
constexpr int kNumTileConfigs = 4;
extern const TileConfig kTileConfigs[kNumTileConfigs];

// Every spin is bounded: a look-back that makes no progress for this many trips sets the failure word and goes on
// (result unspecified) instead of hanging the GPU.
constexpr uint32_t kSpinLimit = 1u << 18;

struct OnesweepArgs {
  // The pass reads one pair of arrays and writes the other; which is which is decided ON THE DEVICE
  // (trivial passes are skipped, see PassPlan in vrdx_kernels.hip): with four ranking passes it is
  // caller -> scratch on passes 0, 2 and scratch -> caller on 1, 3 (reference :417-427).
  uint32_t* keysCaller;
  uint32_t* keysScratch;
  uint32_t* valuesCaller;     // KV only
  uint32_t* valuesScratch;    // KV only
  uint32_t maxCount;          // element count (direct) or upper bound (indirect)
  const uint32_t* countPtr;   // device-side element count (indirect) or nullptr
  const uint32_t* histogramTable;  // uint[4][256]: raw digit counts of all four passes
  uint32_t* statusCur;        // status region of this pass: [statusRows][256]
  uint32_t* statusNext;       // region to clear for the next pass, or nullptr on the last pass
  uint32_t statusRows;
  // Block sums (sorts of one round on the four-pass plan, BlockPrefix in vrdx_kernels.hip): one row per VRDX_BLOCK_TILES
  // tiles behind the tile rows of each status region; nullptr / 0: the classic look-back.
  uint32_t* blockCur;
  uint32_t* blockNext;
  uint32_t blockRows;
  uint32_t* ticketCur;
  uint32_t* ticketNext;
  uint32_t* failure;          // word in the caller's storage: this sort's (cleared when the next sort is recorded)
  uint32_t* stickyFailure;    // the sorter's own word: OR over every sort recorded with it (vrdxHipReadSorterStatus)
  uint32_t pass;              // 0..3: digit = (key >> 8 * pass) & 255 (the hybrid plan's launch 0 ranks by byte 3)
  uint32_t hybridCap;         // 0, or the bucket capacity of the hybrid plan recorded with this sort (PassPlan)
  uint32_t spinLimit;         // look-back trips without progress before the tile gives up (kSpinLimit)
  uint32_t earlyValues;       // KV: fetch the values right after the ranking instead of after the look-back
  uint32_t slots;             // 0: every tile holds the kernel's capacity; otherwise (PlanTiles) the first fullTiles tiles take
                              // `slots` slots of 64 keys per wave (and sub-tile), the tiles behind them tailSlots (multiples of 4)
  uint32_t fullTiles;
  uint32_t tailSlots;
  unsigned long long* trace;  // phase stamps, 8 per tile; nullptr outside tools/trace.sh builds
  uint32_t* planWord;         // hybridCap != 0: the verdict word in the storage (VRDX_OFF_PLAN), written by launch 0
                              // (last on purpose: the argument layout of the kernels that never read it stays as it was)
  // Non-zero: the MSD plan is recorded in front of the passes, which then return when the verdict word says 3 (the plan
  // has taken the sort).
  uint32_t planInFront;
};

// Raises the dynamic-LDS limit of both instantiations (keys-only, key-value) of one tile config.
hipError_t PrepareKernels(int configIndex);

// Also zeroes the two tile tickets and status region 0 (statusClearBytes from statusClear, whole 1 KiB rows): they live
// outside the prefix of the storage that the fill in front of this kernel clears.
hipError_t LaunchHistogram(hipStream_t stream, uint32_t grid, const uint32_t* keys, uint32_t maxCount,
                           const uint32_t* countPtr, uint32_t* globalHistogram, uint32_t* tickets, void* statusClear,
                           uint32_t statusClearBytes);

// atomicRank selects the one-LDS-atomic-per-key ranking; only legal when LdsOrderCheck() said so.
hipError_t LaunchOnesweep(hipStream_t stream, int configIndex, uint32_t grid, bool keyValue, bool atomicRank,
                          const OnesweepArgs& args);

// Small sorts (maxCount <= kSmallSortMaxElements): the whole sort in one workgroup and one launch,
// in place in keys / values (values == nullptr: keys-only); of the storage only *failure is written (0).
constexpr uint32_t kSmallSortMaxElements = 16384;
hipError_t PrepareSmallSort();
hipError_t LaunchSmallSort(hipStream_t stream, bool atomicRank, uint32_t* keys, uint32_t* values, uint32_t maxCount,
                           const uint32_t* countPtr, uint32_t* failure);

// Mid-size sorts, hybrid plan (PassPlan in vrdx_kernels.hip): launch 0 scatters by the keys' highest byte that varies,
// bucket_sort_kernel sorts each of the 256 buckets by the bytes below it inside one workgroup.  hybridCap = 4096, 8192, 16384
// or 32768 elements per bucket (1024 threads x 4 / 8 / 16 / 32); the device decides whether the plan applies.
struct BucketSortArgs {
  const uint32_t* keysScratch;
  uint32_t* keysCaller;
  const uint32_t* valuesScratch;  // KV only
  uint32_t* valuesCaller;         // KV only
  uint32_t maxCount;               // element count (direct) or upper bound (indirect)
  const uint32_t* countPtr;        // device-side element count (indirect) or nullptr
  const uint32_t* histogramTable;  // uint[4][256]
  uint32_t hybridCap;              // elements one workgroup can take (selects the instantiation)
  const uint32_t* planWord;        // the verdict word in the storage (VRDX_OFF_PLAN), written by launch 0
};
hipError_t PrepareBucketSort();
hipError_t LaunchBucketSort(hipStream_t stream, bool keyValue, bool atomicRank, const BucketSortArgs& args);

// The MSD plan of large sorts (round 5; vrdx_kernels.hip, "MSD plan"): THREE ranking steps of 10-11 bits instead of four
// of 8, and TWO trips of the data through memory instead of four --
//   histogram_msd_kernel  CHOOSES THE WINDOW (round 6): from 64 keys sampled evenly over the input (first and last key
//                         included; every workgroup takes the same sample and reaches the same answer) it takes the bits in
//                         which they differ and puts the 2^bits-wide window right below their common prefix -- keys of 24
//                         bits, dense sorted ids, anything narrow then spread over all the buckets like uniform 32-bit keys do
//                         over the top bits -- and predicts what the plan cannot take anyway (MsdMode below).  Then the byte
//                         histograms (for the fallback) and, per tile of tileKeys keys, the counts of the keys' window bits as
//                         16-bit numbers (tileCounts); the bucket sizes; a key that breaks the sampled prefix raises the
//                         overflow word -- the sample is the guess, the count stays the proof;
//   spine_msd_kernel      turns the counts, in place, into exclusive prefixes over the tiles and leaves every bucket's base;
//                         a bucket beyond `cap` elements sets *overflowWord;
//   scatter_msd_kernel    one stable scatter by the window bits, caller -> scratch: no ticket, no look-back, no status words --
//                         a tile's bases are bucketBase[d] + its row of prefixes; keys-only sorts take TWO consecutive tiles per
//                         workgroup (round 6: runs of 256 bytes instead of 128);
//   bucket_sort2_kernel   one workgroup per bucket sorts it by the bits below the window (none, one or two stable passes of
//                         up to 11 bits) inside its LDS, scratch -> caller.
// All of it with wave-private counters of 16 bits, two to a word.  The device decides (the overflow word): with a bucket
// beyond the capacity or a key outside the prefix the last two return at once and the four passes recorded behind them run.
constexpr uint32_t kMsdTileKeys = 32768;   // a scatter tile's capacity: 1024 threads x 32 keys
constexpr uint32_t kMsdMaxTiles = 2048;    // spine_msd_kernel: 64 chunks of at most 32 rows
constexpr uint32_t kMsdCapKeys = 36864;    // bucket capacity, keys-only: 1024 threads x 36 keys (144 KiB of staging)
constexpr uint32_t kMsdCapKeyValue = 36864;  // the same for pairs: keys and values take turns in the staging buffer
constexpr uint32_t kMsdHalfCap = 18432;    // buckets of the half-size bucket kernel: 512 threads x 36 elements (72 KiB of staging), two workgroups per CU
// What the histogram kernel decides travels in the OVERFLOW WORD itself (VRDX_OFF_MSD_OVERFLOW), so that every launch behind it learns
// "is the plan turned down, where is the window, what kind of input is it" from the one word it reads anyway -- a second
// word would be a second dependent memory round trip in front of the first key load of every workgroup:
//   bits 0-7   non-zero = the plan is turned down (kMsdDecline*: by the spine or by the histogram kernel)
//   bits 8-13  shift: the scatter ranks by (key >> shift) & (2^bits - 1); the bucket kernel sorts the `shift` bits below
//   bits 16-17 MsdMode
// The scatter passes the shift on in the verdict word (kMsdVerdict* | shift << 8) for the same reason.
constexpr uint32_t kMsdDeclineMask = 0xFFu;
constexpr uint32_t kMsdDeclineBucket = 1u;  // spine: a bucket holds more than the capacity
constexpr uint32_t kMsdDeclinePrefix = 2u;  // histogram: a key outside the sampled prefix / not the sampled key (all sampled keys identical)
constexpr uint32_t kMsdDeclineSample = 4u;  // histogram: the sample rules the plan out
constexpr uint32_t kMsdShiftShift = 8u, kMsdShiftMask = 63u, kMsdModeShift = 16u, kMsdModeMask = 3u;
enum MsdMode : uint32_t {
  kMsdModePlan = 0,       // counts per tile and window value; spine, scatter, buckets
  kMsdModeDeclined = 1,   // the sample shows a bucket the plan cannot hold (few distinct values, keys of fewer bits than the
                          // window over more elements than fit): the overflow word is raised at once, the four byte tables
                          // are all that is counted and the spine kernel returns
  kMsdModeIdentical = 2,  // every sampled key is the same: the histogram kernel checks that ALL are (or raises the overflow
                          // word); if so the input is sorted as it stands and every launch behind returns (verdict 4)
};
// The sample: one key per lane of wave 0 of every histogram workgroup, evenly spread, first and last key included.  (Round 6
// first took it in a one-workgroup kernel in place of the fill in front of the sort -- 4096 keys: an 11.5 us kernel, 1024 keys:
// 5.9 us, where the fill takes 4.3 and the histogram kernel cannot start before it has ended.  64 lines re-read by every
// workgroup are nothing next to the 512 KiB each of them streams.)
constexpr uint32_t kMsdSampleKeys = 64;
constexpr uint32_t kMsdSampleSkew = 8;      // sampled keys in one bucket (expected: 1 / 16 or less) from which the plan is turned down unseen
// verdict word (VRDX_OFF_PLAN), low byte, of the MSD plan; the passes behind it return on either
constexpr uint32_t kMsdVerdictMask = 0xFFu;
constexpr uint32_t kMsdVerdictRuns = 3;      // scatter and bucket launches do the sort (bits 8-13: the window's shift)
constexpr uint32_t kMsdVerdictSorted = 4;    // all keys identical: nothing to do
struct MsdArgs {
  uint32_t* keysCaller;
  uint32_t* keysScratch;
  uint32_t* valuesCaller;      // KV only
  uint32_t* valuesScratch;     // KV only
  uint32_t maxCount;           // element count (direct) or upper bound (indirect)
  const uint32_t* countPtr;    // device-side element count (indirect) or nullptr
  uint32_t* histogramTable;    // uint[4][256]
  uint32_t* tileCounts;        // [tiles][2^bits / 2] words = pairs of 16-bit numbers: counts, then prefixes over the tiles
  uint32_t* bucketBase;        // [2^bits]
  uint32_t* bucketCount;       // [2^bits]: zeroed by the fill in front of the sort; added up by the histogram kernel (windows below a prefix) or by the spine
  uint32_t* overflowWord;      // VRDX_OFF_MSD_OVERFLOW in the storage: non-zero = the plan is turned down
  uint32_t* planWord;          // VRDX_OFF_PLAN: the scatter writes kMsdVerdictRuns / kMsdVerdictSorted (the passes then return)
  uint32_t bits;               // 10 | 11
  uint32_t cap;                // elements a bucket may hold
  uint32_t tiles;              // ceil(maxCount / tileKeys) <= kMsdMaxTiles
  uint32_t tileKeys;           // keys per tile of the histogram's counts and of the scatter: a multiple of 4096, at most kMsdTileKeys
  // status region 0 of the fallback's passes (16-byte vectors): zeroed by the spine kernel, whose 32-64 workgroups have the
  // bandwidth to spare, instead of by the histogram kernel (which it cost 1.9 us at 2^25, round 4)
  void* statusClear;
  uint32_t statusVecs;
  uint32_t* tickets;           // zeroed by the histogram kernel
  uint32_t* declinedPlans;     // the sorter's counter of plans the device turned down (vrdxHipReadPlanCounters), or nullptr
};
hipError_t PrepareMsd();
hipError_t LaunchHistogramMsd(hipStream_t stream, uint32_t grid, const MsdArgs& args);
hipError_t LaunchSpineMsd(hipStream_t stream, const MsdArgs& args);
hipError_t LaunchScatterMsd(hipStream_t stream, bool keyValue, const MsdArgs& args);
hipError_t LaunchBucketSort2(hipStream_t stream, bool keyValue, const MsdArgs& args);
// The scatter (bucketLaunch = false) or bucket (true) launch of the plan with pass 0 / pass 1 of its fallback as a second
// role, chosen on the device by the plan's verdict: saves two of the four returning launches.  `pass` = the arguments and
// passGrid the grid LaunchOnesweep would have been given for that pass (the two-sub-tile kernel keys-only, 1024x32
// key+value; one-atomic ranking).
hipError_t LaunchMsdFused(hipStream_t stream, bool bucketLaunch, bool keyValue, const MsdArgs& args, const OnesweepArgs& pass,
                          uint32_t passGrid);

// Runs the device self-check of the LDS same-address atomic ordering on the current device
// (synchronous, ~1 ms).  *laneOrdered = true when returning atomics are served in lane order.
hipError_t LdsOrderCheck(bool* laneOrdered);
// The same check, small (8 workgroups, ~20 us) and stream-ordered: a mismatch sets bit 1 of *sticky.  Never blocks.
hipError_t LaunchLdsOrderRecheck(hipStream_t stream, uint32_t* sticky);
// One wave that runs for `ticks` ticks of the device's constant-rate wall clock and stores its first and last reading.
hipError_t LaunchSpin(hipStream_t stream, unsigned long long* out, uint32_t ticks);

}  // namespace vrdx

#endif  // VRDX_KERNELS_H

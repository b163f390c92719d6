// Device kernels of the MI355X (gfx950, wave64) 32-bit LSD radix sort behind vrdxCmdSort*.
//
// What the reference computes on this path (and what these kernels must reproduce bit for bit):
//   4 passes x 8-bit digits, each pass a stable counting sort of the keys (and values) by digit
//   `pass` -- src/shader/upsweep.slang:10-45 (per-partition digit histogram + global histogram),
//   src/shader/spine.slang:11-84 (exclusive scan over partitions and over digits) and
//   src/shader/downsweep.slang:41-224 (stable rank inside the partition + scatter).
//
// How it is done here (not a translation of those shaders):
//   * histogram_kernel  -- ONE coalesced 16 B/lane read of the keys builds all four 256-bin digit
//     histograms (the reference re-reads the keys once per pass) into the same uint[4][256] table
//     the reference calls globalHistogram (src/vk_radix_sort.h.in:405-406).
//   * onesweep_kernel   -- one launch per pass fuses upsweep + spine + downsweep: a tile of
//     THREADS*KPT keys is ranked stably inside each wave64 (one returning LDS atomic per key on a
//     wave-private counter, device-verified; 8-ballot match-any as the fallback), publishes its 256
//     digit counts as {flag,value} status words, resolves its global offsets with a decoupled
//     look-back over the preceding tiles (agent-scope relaxed atomics: the 8 XCD L2s are not
//     coherent), regroups keys by digit in LDS and writes them out four at a time with consecutive
//     lanes on consecutive addresses; values replay the permutation through the same LDS buffer.
//   * onesweep_pair_kernel -- the same pass with TWO 32768-key sub-tiles per workgroup, one ticket,
//     one status row and one look-back for both (every keys-only sort beyond one "round" of tiles per CU,
//     see ConfigIndex in vrdx_api.cpp).
//   * small_sort_kernel / bucket_sort_kernel -- sorts of up to 16384 elements in one workgroup; the hybrid plan of
//     mid-size sorts (one scatter by the highest varying byte, then every bucket in LDS).
//   * histogram_msd_kernel / spine_msd_kernel / scatter_msd_kernel / bucket_sort2[_half]_kernel -- the MSD plan of sorts
//     of 8.14 M ... 67 M elements: one chain-free scatter by a window of 10-11 bits (the top ones for uniform keys, chosen on
//     the device below the keys' common prefix otherwise), then every bucket in LDS in two passes.
//   Tile ids are handed out by an atomic ticket in ARRIVAL order, so a look-back only ever waits
//   on a tile that is already running; every spin is bounded (failure word, never a hang).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vrdx_kernels.h"
#include "vrdx_layout.h"

namespace vrdx {

// Cache-policy switches for measurements (tools/variants.sh, tools/nt_sweep.sh).  VRDX_STREAMING_LOADS:
// 0 never, 1 by size (the product, see StreamingLoads below), 2 always; VRDX_HIST_NT: the same three values
// for the histogram's key loads.  VRDX_NT_STORES: the `nt` bit on the scatter stores (off: measured, a loss).
#ifndef VRDX_STREAMING_LOADS
#define VRDX_STREAMING_LOADS 1
#endif
#ifndef VRDX_NT_STORES
#define VRDX_NT_STORES 0
#endif
#ifndef VRDX_HIST_NT
#define VRDX_HIST_NT 1
#endif
// Measurement only (profiles/r04_histogram_busy_stream.txt): the LAST pass of a sort scatters with non-temporal stores, so
// that its output does not sit dirty in the caches when the next sort's histogram starts.  Off in the product: a
// consumer of the sorted data wants it cached.
#ifndef VRDX_NT_LAST_PASS
#define VRDX_NT_LAST_PASS 0
#endif
// (The timing ablations round 5 measured the four-pass formulation's ceiling with -- VRDX_ABLATE: no look-back, no ticket,
// contiguous stores; results wrong by construction -- are gone from the source: profiles/r05_ceiling.txt has the numbers, commit
// 9caa359 the build.)
// Measurement switches of the MSD plan's kernels (tools/r05/ablate.sh builds the variants): VRDX_MSD_XCD = 0: tiles handed
// out round-robin instead of in consecutive chunks per XCD; VRDX_MSD_NT_LOADS / VRDX_MSD_BUCKET_NT: non-temporal loads in
// the scatter / the bucket kernel.
#ifndef VRDX_MSD_XCD
#define VRDX_MSD_XCD 1
#endif
#ifndef VRDX_MSD_NT_LOADS
#define VRDX_MSD_NT_LOADS 1
#endif
#ifndef VRDX_MSD_BUCKET_NT
#define VRDX_MSD_BUCKET_NT 0
#endif
#ifndef VRDX_MSD_SCATTER_DYN
#define VRDX_MSD_SCATTER_DYN 1
#endif
#ifndef VRDX_MSD_EVEN_WAVES
#define VRDX_MSD_EVEN_WAVES 1  // bucket_sort2_kernel deals a bucket's chunks out evenly over its waves
#endif
#ifndef VRDX_MSD_OUT_NT
#define VRDX_MSD_OUT_NT 1  // the bucket kernel's output stores non-temporal: 0 never, 1 by mode and size (the product), 2 always
#endif


// Timing-only phase trace for tools/trace.sh (never defined in the product build): thread 0 of
// every tile stores 100 MHz wall-clock stamps of its phase boundaries into OnesweepArgs::trace.
#ifdef VRDX_TRACE
#define VRDX_STAMP(slot)                                      \
  do {                                                        \
    if (a.trace != nullptr && tid == 0) stamps[slot] = wall_clock64(); \
  } while (0)
#else
#define VRDX_STAMP(slot) \
  do {                   \
  } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));  // 16-byte access, 4-byte aligned

// Tile status words cross CUs and XCDs inside one launch: every access is a relaxed agent-scope
// atomic (global_load/store ... sc1).  The word carries its own flag, so no fence is needed
// (the "data is the flag" granule form).
__device__ __forceinline__ uint32_t LoadStatus(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void StoreStatus(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One tile's count of one digit into its block's row: arrivals in the top bits, sum below (no return value).
__device__ __forceinline__ void AddToBlock(uint32_t* p, uint32_t count) {
  (void)__hip_atomic_fetch_add(p, count | (1u << VRDX_BLOCK_COUNT_SHIFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t ElementCount(uint32_t maxCount, const uint32_t* countPtr) {
  if (countPtr == nullptr) return maxCount;
  const uint32_t c = *countPtr;
  return c < maxCount ? c : maxCount;
}

// Lanes of this wave whose 8-bit digit equals mine (all 64 lanes must be active).
// 8 ballots; for each bit keep the lanes that agree with my bit.
__device__ __forceinline__ uint64_t MatchDigit(uint32_t digit) {
  uint64_t mask = ~0ull;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (digit >> b) & 1u;
    const uint64_t ballot = __ballot(bit);
    mask &= bit ? ballot : ~ballot;
  }
  return mask;
}

__device__ __forceinline__ uint32_t LanesBelow(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global
// load and store the wave has in flight (s_waitcnt vmcnt(0)): the scattered stores of the key phase,
// prefetched keys/values.  Nothing in these kernels hands GLOBAL data from one wave to another
// through a barrier (status words are agent-scope atomics with their own flag), so the onesweep
// kernels use this one throughout.
__device__ __forceinline__ void LdsBarrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Wave-striped load of KPT words per lane: out[i] = base[first + 64 * i] (pad where the index is
// >= n).  The byte offset is a 32-bit value (N < 2^30) added to the 64-bit base, which lets the
// compiler use ONE offset register plus immediates for all the loads (SGPR base + VGPR offset +
// imm) instead of a 64-bit address pair per load -- 60 registers that the two-sub-tile kernel
// does not have.
// DYN (even-split tiles, PlanTiles in vrdx_api.cpp): only the first `slots` (a multiple of 4, wave-uniform) of the
// KPT slots exist; the loops over the slots leave at the first chunk of four that does not, and nothing later reads
// a slot that was never loaded.
template <int KPT, bool NT = false, bool DYN = false>
__device__ __forceinline__ void LoadStriped(const uint32_t* base, uint32_t first, uint32_t n, bool full,
                                            uint32_t pad, uint32_t (&out)[KPT], uint32_t slots = KPT) {
  const char* const bytes = reinterpret_cast<const char*>(base);
  const uint32_t offset = first * 4u;
  // NT: the loads carry the non-temporal bit (StreamingLoads below).  Compile-time on purpose: given both
  // kinds of load on the two sides of a run-time branch, the compiler merges them into plain ones.
  auto word = [&](int i) {
    const uint32_t* const p = reinterpret_cast<const uint32_t*>(bytes + ((uint64_t)offset + (uint64_t)(i * 256)));
    return NT ? __builtin_nontemporal_load(p) : *p;
  };
  if (full) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
      out[i] = word(i);
    }
  } else {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
      out[i] = first + i * 64 < n ? word(i) : pad;
    }
  }
}

// Key+value sorts whose four buffers (16 bytes per element) are between one and three times the
// 256 MiB Infinity Cache read their tiles with NON-TEMPORAL loads: the reads then do not displace the
// lines the previous pass has just written, which is what this pass reads.  Measured on MI355X
// (tools/nt_sweep.sh, profiles/r02_streaming_loads.txt): +3 ... +11 % for 2^24 < N <= 3 * 2^24 pairs
// (66 instead of 59 GItems/s at 2^25), nothing below, -3 ... -4 % above; keys-only sorts gain nothing at
// any size.  The element count is the one the kernel sorts (indirect sorts included).
__device__ __forceinline__ bool StreamingLoads(bool keyValue, uint32_t n) {
  if (VRDX_STREAMING_LOADS == 0) return false;
  if (VRDX_STREAMING_LOADS == 2) return true;
  return keyValue && n > kStreamingLoadsAbove && n <= kStreamingLoadsUpTo;
}

// LoadStriped with the kind of load chosen at run time (uniform).  The empty asm statements keep the
// two arms distinguishable: identical loads at the head or the tail of both would be merged into one
// plain load.
template <int KPT, bool DYN = false>
__device__ __forceinline__ void LoadTile(const uint32_t* base, uint32_t first, uint32_t n, bool full, uint32_t pad,
                                         uint32_t (&out)[KPT], bool streaming, uint32_t slots = KPT) {
  if (streaming) {
    asm volatile("; non-temporal tile loads" ::: "memory");
    LoadStriped<KPT, true, DYN>(base, first, n, full, pad, out, slots);
    asm volatile("" ::: "memory");
  } else {
    LoadStriped<KPT, false, DYN>(base, first, n, full, pad, out, slots);
  }
}

// Wave-striped store, the inverse of LoadStriped: base[first + 64 * i] = in[i] where the index is < n.
template <int KPT, bool DYN = false>
__device__ __forceinline__ void StoreStriped(uint32_t* base, uint32_t first, uint32_t n, bool full,
                                             const uint32_t (&in)[KPT], uint32_t slots = KPT) {
  char* const bytes = reinterpret_cast<char*>(base);
  const uint32_t offset = first * 4u;
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
    if (full || first + i * 64 < n)
      *reinterpret_cast<uint32_t*>(bytes + ((uint64_t)offset + (uint64_t)(i * 256))) = in[i];
  }
}

// Trivial passes.  If ONE digit of a pass holds every key (16- or 24-bit keys, constant bytes,
// all-equal input) that pass's stable permutation is the identity.  Every workgroup of every pass
// reads the whole 4 x 256 table of global counts while its ticket is on its way (one load per thread)
// and derives the same plan:
//   * a trivial pass is SKIPPED altogether (the launch returns after its housekeeping) -- the data
//     stays where it is and the later passes read it from there;
//   * the result has to end in the caller's buffers, i.e. after an even number of buffer changes; if
//     the number of non-trivial passes is odd, the first trivial pass does change buffers: its tiles
//     copy their range (5 TB/s instead of a ranking pass's 3.5).
// So 16-bit keys cost two passes, all-equal input costs the histogram and four empty launches.
struct PassPlan {
  bool skip;         // nothing to do in this launch
  bool copy;         // identity permutation, but the data has to change buffers
  bool fromScratch;  // this pass reads the storage's scratch arrays and writes the caller's buffers
  uint32_t digit;    // the byte of the key this launch ranks by (= its pass index, except in the hybrid plan)
};

// The hybrid plan of mid-size sorts (recorded by the host when hybridCap != 0, chosen HERE, on the device, by every
// workgroup alike).  Let t be the highest byte of the keys that is not the same in all of them.  If no value of byte t
// occurs more than hybridCap times, launch 0 scatters by byte t (caller -> scratch), bucket_sort_kernel then sorts
// each of the 256 buckets by its bytes below t inside one workgroup's LDS (scratch -> caller), and launches 1..3 have
// nothing to do: two trips through memory instead of up to four, and -- what counts at these sizes -- two dependent
// kernels instead of four.  Inputs with a bucket that does not fit a workgroup keep the four-pass plan.

// The table loads are issued BEFORE the ticket atomic (LoadPassCounts) and consumed after it
// (PublishPassVotes): issued after it, wave 0 would wait for the ticket first and then for the loads
// -- two global round trips on every tile's critical path instead of one (3.3 us instead of 1.8 us
// before the first barrier at N = 2^23, measured).
// flags[g] = "some count of group g (64 consecutive counts of the table) equals n".
template <int THREADS>
struct PassCounts {
  uint32_t v[(VRDX_PASSES * VRDX_RADIX) / THREADS];
};

template <int THREADS>
__device__ __forceinline__ PassCounts<THREADS> LoadPassCounts(const uint32_t* histogramTable, int tid) {
  PassCounts<THREADS> c;
#pragma unroll
  for (int k = 0; k < (int)(VRDX_PASSES * VRDX_RADIX) / THREADS; ++k) c.v[k] = histogramTable[tid + k * THREADS];
  return c;
}

// flags[0..15]: see above.  flags[16 + g]: "some count of group g exceeds hybridCap" (32 words in all).
template <int THREADS>
__device__ __forceinline__ void PublishPassVotes(const PassCounts<THREADS>& c, uint32_t n, uint32_t hybridCap, int tid,
                                                 uint32_t* flags) {
#pragma unroll
  for (int k = 0; k < (int)(VRDX_PASSES * VRDX_RADIX) / THREADS; ++k) {
    const int group = (tid + k * THREADS) >> 6;
    const uint64_t any = __ballot(c.v[k] == n);
    const uint64_t over = __ballot(c.v[k] > hybridCap);
    if ((tid & 63) == 0) {
      flags[group] = any != 0ull ? 1u : 0u;
      flags[16 + group] = over != 0ull ? 1u : 0u;
    }
  }
}

__device__ __forceinline__ uint32_t TrivialPasses(const uint32_t* flags) {
  uint32_t trivial = 0;
#pragma unroll
  for (uint32_t q = 0; q < VRDX_PASSES; ++q)
    if ((flags[4 * q] | flags[4 * q + 1] | flags[4 * q + 2] | flags[4 * q + 3]) != 0) trivial |= 1u << q;
  return trivial;
}

// The byte the hybrid plan scatters by, or -1 when the plan does not apply (the same answer in every workgroup of
// every launch of the sort: it follows from the global histogram and the recorded capacity alone).
__device__ __forceinline__ int HybridByte(const uint32_t* flags, uint32_t hybridCap) {
  if (hybridCap == 0) return -1;
  const uint32_t ranked = ~TrivialPasses(flags) & 15u;
  if (ranked == 0) return -1;  // all keys equal: nothing to sort, the four (empty) passes say so already
  const int top = 31 - __clz((int)ranked);
  const bool fits = (flags[16 + 4 * top] | flags[17 + 4 * top] | flags[18 + 4 * top] | flags[19 + 4 * top]) == 0;
  return fits ? top : -1;
}

__device__ __forceinline__ PassPlan ReadPassPlan(const uint32_t* flags, uint32_t pass, uint32_t hybridCap) {
  PassPlan plan;
  const int hybridByte = HybridByte(flags, hybridCap);
  if (hybridByte >= 0) {
    plan.skip = pass != 0;
    plan.copy = false;
    plan.fromScratch = false;
    plan.digit = (uint32_t)hybridByte;
    return plan;
  }
  const uint32_t trivial = TrivialPasses(flags);
  const uint32_t ranked = VRDX_PASSES - (uint32_t)__popc(trivial);
  // odd number of ranking passes: the first trivial pass copies, so that the result ends at the caller
  const uint32_t copier = (ranked & 1u) ? (uint32_t)(__ffs(trivial) - 1) : VRDX_PASSES;
  const uint32_t changes = (~trivial & 15u) | (copier < VRDX_PASSES ? 1u << copier : 0u);
  plan.skip = ((changes >> pass) & 1u) == 0;
  plan.copy = !plan.skip && ((trivial >> pass) & 1u) != 0;
  plan.fromScratch = (__popc(changes & ((1u << pass) - 1u)) & 1) != 0;
  plan.digit = pass;
  return plan;
}

// Inclusive scan over the 64 lanes of a wave with DPP moves (all lanes must be active): four
// shifts inside each row of 16 lanes, then lane 15 of rows 0 and 2 is added to rows 1 and 3, then
// lane 31 to rows 2 and 3.  (__shfl_up compiles to ds_bpermute here: six dependent LDS round trips,
// ~0.35 us per scan on every tile's critical path.)
__device__ __forceinline__ uint32_t WaveInclusiveScan(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return (uint32_t)x;
}

// Exclusive scan of one value per thread over threads 0..255 (4 waves); other threads pass 0 and
// ignore the result.  Contains one barrier: every thread of the block must call it.
__device__ __forceinline__ uint32_t BlockExclusiveScan256(uint32_t v, uint32_t* scratch4, int tid) {
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t x = WaveInclusiveScan(v);
  if (wave < 4 && lane == 63) scratch4[wave] = x;
  LdsBarrier();
  uint32_t add = 0;
  if (wave < 4) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
      if (w < wave) add += scratch4[w];
  }
  return x - v + add;
}

// ---------------------------------------------------------------------------------------------
// histogram: all four digit histograms in one pass over the keys
// ---------------------------------------------------------------------------------------------
// LDS counters are replicated COPIES times (copy = lane % COPIES, copies of one bin in consecutive
// banks) so that all-equal / few-distinct keys do not serialise on one LDS address the way the
// reference's 512-way atomicAdd on localHistogram[radix] does (upsweep.slang:34).  With 32 copies
// every lane of a 32-lane access group has a bank of its own, so a ds_add_u32 takes its 4 LDS cycles
// whatever the keys are (measured, tools/probes/lds_op_rate.hip: 4.0 cycles with 32 copies, 4.25 with 16,
// 6.3 with 8): 4 atomics per key are 14 us per CU at N = 2^25 against 21 us for the read itself, so the
// kernel is bound by the read -- PROVIDED the loads of the next group are in flight while a group is
// counted.  Round 2's kernel loaded a group, waited, counted, and only then loaded again: a memory
// latency plus a counting phase per 64 KiB and CU, 35 us.  Here every wave keeps TWO groups of four
// 16-byte loads per lane in flight (128 KiB per CU) and counts the older one; the loads are
// unconditional (index clamped, the count masked) because a branch around a load makes the compiler
// wait for vmcnt(0) at the join.
//
// The grid-stride order matters as well: at any time the workgroups read ONE contiguous window of the input.  With
// one contiguous range per workgroup the same loop takes 37 instead of 30 us at N = 2^25
// (profiles/r03_chain_free_pass0.txt, "flatc").
constexpr uint32_t kHistGroupVecs = kHistGroupKeys / 4;     // 16-byte vectors per group (4 per lane)
static_assert(kHistGroupVecs == kHistThreads * 4, "four loads per lane and group");

template <bool NT>
__device__ __forceinline__ void HistFetch(const u32x4* keys4, uint32_t group, uint32_t tid, uint32_t nvec,
                                          u32x4 (&k)[4]) {
#pragma unroll
  for (uint32_t u = 0; u < 4; ++u) {
    uint32_t i = group * kHistGroupVecs + u * kHistThreads + tid;
    i = i < nvec ? i : (nvec != 0 ? nvec - 1 : 0u);  // out-of-range lanes re-read the last vector (never counted)
    k[u] = NT ? __builtin_nontemporal_load(keys4 + i) : keys4[i];
  }
}

// The kernel also zeroes status region 0 (statusVecs 16-byte vectors from statusClear, shared out over the workgroups)
// and the two tickets: nothing reads them before pass 0, which starts when this kernel has drained.  Before round 4
// the fill in front of the histogram cleared region 0 as well -- 1-2 MiB at N = 2^25, a 4.4 us fill kernel on the
// critical path of every sort where 4 KiB (header + table, which the atomics below need zeroed) take 2 us.
template <uint32_t COPIES>
__global__ __launch_bounds__(kHistThreads) void histogram_kernel(const uint32_t* __restrict__ keys,
                                                                  uint32_t maxCount,
                                                                  const uint32_t* countPtr,
                                                                  uint32_t* __restrict__ globalHistogram,
                                                                  uint32_t* __restrict__ tickets,
                                                                  u32x4* __restrict__ statusClear,
                                                                  uint32_t statusVecs) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* const bins = smem;  // [pass][digit][copy]
  constexpr uint32_t kByteTables = VRDX_PASSES;
  constexpr uint32_t kBinWords = kByteTables * VRDX_RADIX * COPIES;
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x == 0 && tid < 3) tickets[tid] = 0;  // outside the cleared prefix of the storage (vrdx_layout.h)
  const uint32_t n = ElementCount(maxCount, countPtr);

  const uint32_t copy = tid & (COPIES - 1);
  auto count = [&](uint32_t key) {
#pragma unroll
    for (uint32_t p = 0; p < kByteTables; ++p) {
      const uint32_t d = (key >> (8 * p)) & 0xFFu;
      atomicAdd(&bins[(p * VRDX_RADIX + d) * COPIES + copy], 1u);
    }
  };
  auto tally = [&](uint32_t group, const u32x4 (&k)[4], uint32_t nvec) {
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      if (group * kHistGroupVecs + u * kHistThreads + tid < nvec) {
        count(k[u][0]);
        count(k[u][1]);
        count(k[u][2]);
        count(k[u][3]);
      }
    }
  };

  const uint32_t nvec = n >> 2;
  const u32x4* keys4 = reinterpret_cast<const u32x4*>(keys);
  const uint32_t groups = (nvec + kHistGroupVecs - 1) / kHistGroupVecs;
  // Inputs of more than half the 256 MiB Infinity Cache are read with non-temporal loads (little of them
  // would still be there for pass 0): 51 instead of 55 us at N = 2^26, 90 instead of 104 us at 2^27, no
  // difference at 2^24 and 2^25 (profiles/r02_streaming_loads.txt).
  auto sweep = [&](auto streaming) {
    constexpr bool NT = decltype(streaming)::value;
    uint32_t g = blockIdx.x;
    const uint32_t end = groups;
    const uint32_t step = gridDim.x;
    u32x4 a[4], b[4];
    // the first two groups' loads fly while the counters are cleared (nvec >= 1 here)
    HistFetch<NT>(keys4, g, tid, nvec, a);
    HistFetch<NT>(keys4, g + step, tid, nvec, b);
    for (uint32_t i = tid; i < kBinWords; i += kHistThreads) bins[i] = 0;
    LdsBarrier();  // LDS only: the loads in flight are not waited for here
    for (; g < end; g += 2 * step) {
      tally(g, a, nvec);
      HistFetch<NT>(keys4, g + 2 * step, tid, nvec, a);
      if (g + step < end) tally(g + step, b, nvec);
      HistFetch<NT>(keys4, g + 3 * step, tid, nvec, b);
    }
  };
  const bool streamingInput = VRDX_HIST_NT == 2 || (VRDX_HIST_NT == 1 && n > kHistStreamingLoadsAbove);
  if (nvec == 0) {  // fewer than four keys
    for (uint32_t i = tid; i < kBinWords; i += kHistThreads) bins[i] = 0;
    LdsBarrier();
  } else if (streamingInput) {
    asm volatile("; non-temporal key loads" ::: "memory");  // keeps the two loops apart (see LoadTile)
    sweep(std::true_type{});
  } else {
    sweep(std::false_type{});
  }
  if (blockIdx.x == 0 && tid < (n & 3u)) count(keys[(nvec << 2) + tid]);
  // status region 0: issued behind the key loads, drained with them
  for (uint32_t i = blockIdx.x * kHistThreads + tid; i < statusVecs; i += gridDim.x * kHistThreads)
    statusClear[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  for (uint32_t b = tid; b < kByteTables * VRDX_RADIX; b += kHistThreads) {
    uint32_t sum = 0;
    // rotated by the lane so that the lanes of one read do not all hit copy c's bank
#pragma unroll
    for (uint32_t c = 0; c < COPIES; ++c) sum += bins[b * COPIES + ((c + tid) & (COPIES - 1))];
    if (sum != 0) atomicAdd(&globalHistogram[b], sum);
  }
}

// Sum over the four lanes of a quad (all lanes active), result in every lane.
__device__ __forceinline__ uint32_t QuadSum(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);  // quad_perm:[1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);  // quad_perm:[2,3,0,1]
  return (uint32_t)x;
}

// ---------------------------------------------------------------------------------------------
// onesweep: rank + decoupled look-back + scatter, one launch per pass
// ---------------------------------------------------------------------------------------------

// Which keys a tile covers.  Tiles are numbered by ticket; tile i < fullTiles holds slotsA slots of 64 keys per wave
// (times `lanesPerSlot` = threads, or two sub-tiles' worth), the tiles behind them slotsB -- PlanTiles in vrdx_api.cpp:
// whole rounds of full tiles, then one round of small equal tiles for the rest, so that the last, partial round of a
// sort costs what its keys cost.  fullTiles = ~0u: every tile alike.  All wave-uniform (scalar registers).
struct TileSpan {
  uint32_t start;   // first key
  uint32_t slots;   // slots per wave (and sub-tile) of this tile
  bool last;        // the tile that holds key n - 1: never looked at, publishes nothing
  bool beyond;      // starts at or behind n: nothing to do
};
__device__ __forceinline__ TileSpan SpanOfTile(uint32_t tile, uint32_t n, uint32_t slotsA, uint32_t slotsB,
                                               uint32_t fullTiles, uint32_t lanesPerSlot) {
  TileSpan s;
  const uint32_t frameA = slotsA * lanesPerSlot, frameB = slotsB * lanesPerSlot;
  const bool tail = tile >= fullTiles;
  const uint32_t keysA = tail ? fullTiles * frameA : 0u;
  const uint32_t frame = tail ? frameB : frameA;
  s.slots = tail ? slotsB : slotsA;
  s.start = keysA + (tail ? tile - fullTiles : tile) * frame;
  s.beyond = s.start >= n;
  s.last = !s.beyond && n - s.start <= frame;
  return s;
}

// Measured (tools/trace.sh, profiles/): agent-scope (sc1) status reads cost a CU roughly 13 GB/s,
// independent of the access width (16-byte row loads were SLOWER: 2.9 us per 4-row trip against
// 1.5 us per 8-row trip of 4-byte loads), a tile walks ~49 rows = 49 KiB before it meets an
// inclusive prefix, and that is the 5.6 us it spends here.  Wider windows only add bytes.
//
// Decoupled look-back by the WHOLE workgroup.  Thread (g, d) = (tid / 256, tid % 256) inspects
// kLookBackWindow consecutive predecessor tiles of digit d per trip, group g starting where group
// g - 1 ends, so one trip covers GROUPS * kLookBackWindow tiles with every status load in flight at
// once.  (At the start of a pass all resident tiles begin together and the inclusive prefixes
// spread tile by tile; the number of trips a tile needs falls with the square root of the
// window, which is why the window is wide.)  Group 0's thread of each digit then stitches the
// groups' partial sums together in order.  Returns the exclusive prefix in the threads tid < 256.
//
// lds: pos[256] | sum[GROUPS][256] | info[GROUPS][256]   (info = consumed | hitInclusive << 8)
#ifndef VRDX_LOOKBACK_WINDOW
#define VRDX_LOOKBACK_WINDOW 8
#endif
constexpr int kLookBackWindow = VRDX_LOOKBACK_WINDOW;
constexpr int32_t kLookBackDone = INT32_MIN;

// One 16-byte store of a sorted quad to out[index .. index + 3] (4-byte aligned).
__device__ __forceinline__ void StoreQuad(uint32_t* out, uint32_t index, u32x4 q, bool nt = false) {
  u32x4_a4* const p = reinterpret_cast<u32x4_a4*>(reinterpret_cast<char*>(out) + (uint64_t)(index * 4u));
  if (VRDX_NT_STORES) {
    __builtin_nontemporal_store(q, p);
  } else if (VRDX_NT_LAST_PASS && nt) {  // wave-uniform; the empty asm keeps the two kinds of store apart (see LoadTile)
    asm volatile("; non-temporal scatter" ::: "memory");
    __builtin_nontemporal_store(q, p);
    asm volatile("" ::: "memory");
  } else {
    *p = q;
  }
}

// RADIX: digits per status row (256 in every kernel built today; THREADS / RADIX groups of threads share the look-back).
template <int THREADS, int RADIX = 256, int W = kLookBackWindow>
__device__ __forceinline__ uint32_t LookBack(const uint32_t* status, uint32_t tile, int tid, uint32_t* lds,
                                             uint32_t* failure, uint32_t* stickyFailure, uint32_t spinLimit,
                                             uint32_t* traceTripsRows) {
  constexpr int GROUPS = THREADS / RADIX;
  int32_t* const pos = reinterpret_cast<int32_t*>(lds);
  uint32_t* const sum = lds + RADIX;
  uint32_t* const info = sum + GROUPS * RADIX;
  const int g = tid / RADIX;
  const int d = tid % RADIX;

  uint32_t exclusive = 0;
  uint32_t spins = 0;
  uint32_t traceTrips = 0, traceRows = 0;  // thread 0 only, reported to tools/trace.sh builds
  bool done = g != 0;  // only group 0 owns the per-digit state
  uint32_t* const vote = info + GROUPS * RADIX;  // [2]: "some digit is still walking", by trip parity
  if (g == 0) pos[d] = (int32_t)tile - 1;
  if (tid < 2) vote[tid] = 0;
  LdsBarrier();

  for (;;) {
    const int32_t j = pos[d];
    if (j != kLookBackDone) {
      const int32_t first = j - g * W;
      uint32_t v[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const int32_t row = first - k;
        v[k] = row >= 0 ? LoadStatus(&status[(uint32_t)row * (uint32_t)RADIX + d])
                        : (VRDX_FLAG_INCLUSIVE << VRDX_FLAG_SHIFT);
      }
      uint32_t partial = 0, consumed = 0, hit = 0;
      bool open = true;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const uint32_t flag = v[k] >> VRDX_FLAG_SHIFT;
        open = open && flag != VRDX_FLAG_EMPTY;
        if (open) {
          partial += v[k] & VRDX_VALUE_MASK;
          ++consumed;
          if (flag == VRDX_FLAG_INCLUSIVE) {
            hit = 1;
            open = false;
          }
        }
      }
      sum[g * RADIX + d] = partial;
      info[g * RADIX + d] = consumed | (hit << 8);
    }
    LdsBarrier();
    if (g == 0 && !done) {
      uint32_t advance = 0;
#pragma unroll
      for (int gg = 0; gg < GROUPS; ++gg) {
        const uint32_t in = info[gg * RADIX + d];
        exclusive += sum[gg * RADIX + d];
        advance += in & 0xFFu;
        if (in >> 8) {
          done = true;
          break;
        }
        if ((in & 0xFFu) < (uint32_t)W) break;  // blocked on a tile that has not published yet
      }
      if (!done && advance == 0) {
        if (++spins > spinLimit) {
          atomicOr(failure, 1u);  // bounded: give up (result unspecified) rather than hang the GPU
          if (stickyFailure != nullptr) atomicOr(stickyFailure, 1u);  // the sorter's word: survives the next sort's clear
          done = true;
        } else {
          __builtin_amdgcn_s_sleep(1);
        }
      }
      pos[d] = done ? kLookBackDone : j - (int32_t)advance;
      traceRows += advance;
    }
    // One barrier + vote: this trip's sum/info are consumed and pos is updated before the next
    // trip touches them.  (A hand-made vote: __syncthreads_and would add static LDS and push two
    // 80 KiB workgroups over the CU's 160 KiB.)
    if (!done) vote[traceTrips & 1u] = 1;
    LdsBarrier();
    const bool again = vote[traceTrips & 1u] != 0;
    ++traceTrips;
    if (!again) break;
    if (tid == 0) vote[traceTrips & 1u] = 0;
  }
  if (traceTripsRows != nullptr && tid == 0) *traceTripsRows = (traceTrips << 16) | (traceRows & 0xFFFFu);
  return exclusive;
}

// Block sums: the prefix of a tile WITHOUT the chain of inclusive prefixes -- sorts of ONE round of tiles.
//
// When all tiles of a pass start together (one per CU) every tile publishes its aggregate at about the same moment,
// and the classic look-back then waits for inclusive prefixes to spread tile by tile: tile i needs ~i/64 dependent
// trips (2.3-2.9 on average, 4-6 for the high tiles of the round, ~1.5 us each).  Here every tile also ADDS its 256
// counts to the row of its block of VRDX_BLOCK_TILES consecutive tiles (one no-return agent-scope atomic per digit; the
// word counts the arrivals in its top bits), and a tile's prefix is
//     the global digit base  +  the complete block rows before its block  +  the aggregates of the tiles before it in
//     its block:
// at most 31 + 31 words per digit, all of which exist as soon as the predecessors have PUBLISHED -- no tile waits for
// another tile's look-back.  Groups 0 and 1 (256 threads each, one per digit) add up the tile rows, groups 2 and 3 the
// block rows, 16 words per thread, each thread on its own (no barrier inside the loop).  Measured in round 2 on
// experiment/block-prefix (1.46 instead of 2.3 trips per tile; +2 ... +8 % at 2^22.5 ... 2^24 keys, -3 % at 2^25 and
// beyond, where later rounds find inclusive prefixes waiting for them), adopted in round 4 for sorts of one round
// (PlanTiles / MakeLayout decide: 64 ... CUs tiles of 32768 keys and more, four-pass plan).
// Returns the sum in the threads tid < 256; contains one barrier.
template <int THREADS>
__device__ __forceinline__ uint32_t BlockPrefix(const uint32_t* status, const uint32_t* blocks, uint32_t tile, int tid,
                                                uint32_t* lds, uint32_t* failure, uint32_t* stickyFailure,
                                                uint32_t spinLimit, uint32_t* traceTripsRows) {
  static_assert(THREADS == 1024, "two groups for the tile rows, two for the block rows");
  constexpr uint32_t W = VRDX_BLOCK_TILES / 2;
  const uint32_t g = (uint32_t)tid >> 8;
  const uint32_t d = (uint32_t)tid & 255u;
  const uint32_t block = tile / VRDX_BLOCK_TILES;
  const bool blockRows = g >= 2;
  const uint32_t span = blockRows ? block : tile % VRDX_BLOCK_TILES;  // rows of my kind to add up
  const uint32_t half = (g & 1u) * W;
  uint32_t lo = half < span ? half : span;
  const uint32_t hi = half + W < span ? half + W : span;
  const uint32_t* const rows = (blockRows ? blocks : status + (size_t)block * VRDX_BLOCK_TILES * VRDX_RADIX) + d;
  uint32_t sum = 0, spins = 0, trips = 0;
  while (lo < hi) {
    uint32_t v[W];
#pragma unroll
    for (uint32_t k = 0; k < W; ++k) v[k] = lo + k < hi ? LoadStatus(&rows[(size_t)(lo + k) * VRDX_RADIX]) : 0u;
    uint32_t consumed = 0;
    bool open = true;
#pragma unroll
    for (uint32_t k = 0; k < W; ++k) {
      const bool ready = blockRows ? (v[k] >> VRDX_BLOCK_COUNT_SHIFT) == VRDX_BLOCK_TILES
                                   : (v[k] >> VRDX_FLAG_SHIFT) != VRDX_FLAG_EMPTY;
      open = open && lo + k < hi && ready;
      if (open) {
        sum += v[k] & (blockRows ? VRDX_BLOCK_SUM_MASK : VRDX_VALUE_MASK);
        ++consumed;
      }
    }
    lo += consumed;
    ++trips;
    if (consumed == 0) {
      if (++spins > spinLimit) {
        atomicOr(failure, 1u);  // bounded, like the look-back: give up rather than hang the GPU
        if (stickyFailure != nullptr) atomicOr(stickyFailure, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  lds[g * 256u + d] = sum;
  if (traceTripsRows != nullptr && tid == 0) *traceTripsRows = (trips << 16) | (span & 0xFFFFu);
  LdsBarrier();
  return g == 0 ? lds[d] + lds[256u + d] + lds[512u + d] + lds[768u + d] : 0u;
}

// Test build only (-DVRDX_TESTING, tests/sticky_status_check.py): with a spin limit of 0 the first look-back trip that
// finds a predecessor unpublished gives up, and tile 0 holds its inclusive prefix back for ~0.3 ms, so that tile 1 is
// CERTAIN to find it unpublished -- the failure path is then exercised deterministically, not "in practice".
__device__ __forceinline__ void TestDelayFirstTile(uint32_t tile, uint32_t spinLimit) {
#ifdef VRDX_TESTING
  if (spinLimit == 0 && tile == 0)
    for (int i = 0; i < 100; ++i) __builtin_amdgcn_s_sleep(127);
#else
  (void)tile;
  (void)spinLimit;
#endif
}

// Stable rank of one key slot inside its wave, two interchangeable ways (same results):
//
//  * RankBallot  -- match-any with 8 ballots; leader lane bumps the wave-private digit counter.
//    Uses only architecturally defined behaviour.  ~90 VALU instructions per slot.
//  * RankAtomic  -- one returning LDS atomic per key: old = ds_add_rtn_u32(counter[digit], 1).
//    Several lanes of ONE instruction hitting one address are served in ascending lane order on
//    gfx950, which is exactly the stable order; the ISA manual does not promise that order, so
//    vrdxCreateSorter verifies it on the actual device (LdsOrderCheck below) and selects the
//    ballot form if the check ever fails.  A wave whose 64 digits are all equal (sorted or
//    constant inputs in the high passes) is handled by one lane adding 64, so the adversarial
//    inputs do not serialise 64-way on one LDS address.
//
// Counters are wave-private and LDS serves one wave's operations in program order, so the add of
// slot i is visible to slot i + 1 without any barrier.
// PACKED: ranks (< 64 * KPT <= 65536) are written two to a register, out[i / 2] bits 16*(i % 2).
template <int KPT, bool PACKED = false, bool DYN = false>
__device__ __forceinline__ void RankBallot(const uint32_t (&key)[KPT], uint32_t shift, uint32_t* myHist,
                                           int lane, uint32_t (&out)[PACKED ? KPT / 2 : KPT], uint32_t slots = KPT) {
  static_assert(!PACKED || KPT % 2 == 0, "whole pairs");
  uint32_t even = 0;
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
    const uint32_t d = (key[i] >> shift) & 0xFFu;
    const uint64_t same = MatchDigit(d);
    const uint32_t below = LanesBelow(same);
    const uint32_t prior = __hip_atomic_load(&myHist[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (below == 0)
      __hip_atomic_fetch_add(&myHist[d], (uint32_t)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t r = prior + below;
    if constexpr (PACKED) {
      if (i % 2 == 0) {
        even = r;
      } else {
        out[i / 2] = even | (r << 16);
        asm volatile("" : "+v"(out[i / 2]));  // pack now, not when first used
      }
    } else {
      out[i] = r;
    }
  }
}

// PACKED: ranks (< 64 * KPT <= 65536) are written two to a register, out[i / 2] bits 16*(i % 2).
// MASK: the digit is (key >> shift) & MASK -- 0xFF in every kernel built today.
template <int KPT, bool PACKED, bool DYN = false, uint32_t MASK = 0xFFu>
__device__ __forceinline__ void RankAtomic(const uint32_t (&key)[KPT], uint32_t shift, uint32_t* myHist,
                                           int lane, uint32_t (&out)[PACKED ? KPT / 2 : KPT], uint32_t slots = KPT) {
  // eight slots at a time: eight atomics in flight, eight wave-uniform flags live (more would
  // push the 64-bit flag masks out of the scalar register file); four with a run-time slot count
  // (the ranking chunk makes no measurable difference: 4 / 8 / 16 / 32 within 1 %, round 2)
  constexpr int CHUNK = (!DYN && KPT % 8 == 0) ? 8 : 4;
  static_assert(KPT % CHUNK == 0 && CHUNK % 2 == 0, "whole chunks of pairs");
  // The wave-uniform test costs two VALU instructions per key -- 5 % of a pass on random keys, where it never fires
  // (measured by compiling it out).  So a chunk is tested ("watched") only if its FIRST slot looks the part: a quarter
  // of the lanes or more share the first lane's digit -- one probe of five instructions per chunk, never true on random
  // digits (round 3; before, the first chunk of every tile was tested in full and later chunks while the test kept
  // firing: 0.6 us per keys-only pass at 2^25).  Sorted, constant, two-valued and long-run inputs take the watched
  // path chunk after chunk.  (Uniform slots in an unwatched chunk are still ranked correctly, by a 64-way same-address
  // atomic: ~60 LDS cycles instead of 2.)
#pragma unroll
  for (int base = 0; base < KPT; base += CHUNK) {
    if (DYN && (uint32_t)base >= slots) break;
    uint32_t r[CHUNK];
    const uint32_t probe = (key[base] >> shift) & MASK;
    const bool watch = __popcll(__ballot(probe != (uint32_t)__builtin_amdgcn_readfirstlane(probe))) <= 48;  // wave-uniform
    if (watch) {
      bool uniform[CHUNK];
#pragma unroll
      for (int c = 0; c < CHUNK; ++c) {
        const int i = base + c;
        const uint32_t d = (key[i] >> shift) & MASK;
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        const uint64_t others = __ballot(d != d0);
        uniform[c] = others == 0ull;  // wave-uniform
        uint32_t old = 0;
        bool ranked = false;  // wave-uniform
        if (!uniform[c] && __popcll(others) <= 48) {
          // (Round 5 tried the general form -- peel off up to FOUR groups with readlane / ballot, one adding lane per group
          // -- for keys of a handful of distinct values: ~50 instructions per slot on all sixteen waves cost what the 16-way
          // conflicts they avoid cost in the LDS, and the code they add made every watched input slower: four-valued keys at
          // 2^25, passes 0 + 1: 201 us instead of 174, profiles/r05_adversarial_2pow25.txt.  Not adopted.)
          // A quarter of the lanes and more share the first lane's digit (never on random digits).  If everybody else is
          // ONE other digit -- a byte that is 0x00 | 0xFF (small signed integers), the top pass of dense sorted keys
          // (k and k + 2^24 side by side) -- the slot would serialise 32-way on two counters: 64 LDS cycles instead of 9,
          // 13.6 instead of 2 us of ranking per tile (profiles/r03_few_digit_passes.txt).  Rank it with ballots: the first
          // lane of each group adds the group's size, a lane's rank is that counter plus the lanes of its group below it.
          const int leader1 = __builtin_ctzll(others);
          const uint64_t m1 = __ballot(d == (uint32_t)__builtin_amdgcn_readlane((int)d, leader1));
          if ((~others | m1) == ~0ull) {
            const bool first = d == d0;
            const uint64_t mine = first ? ~others : m1;
            const uint32_t below = LanesBelow(mine);
            if (below == 0)
              old = __hip_atomic_fetch_add(&myHist[d], (uint32_t)__popcll(mine), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t old0 = __builtin_amdgcn_readfirstlane(old);
            const uint32_t old1 = (uint32_t)__builtin_amdgcn_readlane((int)old, leader1);
            old = (first ? old0 : old1) + below;
            ranked = true;
          }
        }
        if (!ranked && (!uniform[c] || lane == 0))
          old = __hip_atomic_fetch_add(&myHist[d], uniform[c] ? 64u : 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        r[c] = old;
      }
#pragma unroll
      for (int c = 0; c < CHUNK; ++c)
        if (uniform[c]) r[c] = __builtin_amdgcn_readfirstlane(r[c]) + lane;
    } else {
#pragma unroll
      for (int c = 0; c < CHUNK; ++c)
        r[c] = __hip_atomic_fetch_add(&myHist[(key[base + c] >> shift) & MASK], 1u, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) {
      if (PACKED) {
        if (c % 2 == 1) {
          out[(base + c) / 2] = r[c - 1] | (r[c] << 16);
          asm volatile("" : "+v"(out[(base + c) / 2]));  // pack now, not when first used
        }
      } else {
        out[base + c] = r[c];
      }
    }
  }
}

// Staging-buffer swizzle.  Tile-local sorted position p lives at word StagingSlot<TILE>(p): bits
// 2..5 of p (the bank group of its 16-byte quad) are XORed with four higher bits.  On inputs whose
// tile histogram is flat and whose digits cycle from lane to lane (every pass of an LSD sort of
// ALREADY SORTED keys looks like that) the 64 lanes of one regroup store go to p = d * (TILE/256)
// + c, i.e. to ONE bank: 64-way conflicts, 14.4 us instead of 3.2 us per tile (measured, PMC
// SQ_LDS_BANK_CONFLICT 32.5 M cycles per pass).  With the swizzle they spread over 16 bank groups.
// Quads stay whole and 16-byte aligned, and the map is an involution (the source bits are above
// the target bits), so the scatter recovers p from the physical quad it reads.
template <uint32_t TILE>
__device__ __forceinline__ uint32_t StagingSlot(uint32_t p) {
  constexpr uint32_t kPerDigit = TILE / 256;  // keys per digit in a flat tile
  constexpr int kLog = kPerDigit >= 128 ? 7 : 6;
  static_assert(TILE >= (1u << (kLog + 4)), "source bits inside the tile");
#ifdef VRDX_NO_SWIZZLE
  return p;
#else
  return p ^ ((p >> (kLog - 2)) & 0x3Cu);
#endif
}

// Regroup: sorted[StagingSlot(rank + waveBase[digit])] = key, eight keys at a time -- the eight
// counter reads are issued together and only then the eight stores (LDS reads and writes of one
// array cannot be reordered by the compiler, so a read-store-read-store source order costs a full
// LDS round trip per key).  PACKED ranks come two to a register; with KEEP the physical slots are
// returned packed the same way (key+value stages the values through them).
template <int KPT, uint32_t STAGE, bool PACKED, bool KEEP, bool DYN = false, uint32_t MASK = 0xFFu>
__device__ __forceinline__ void RegroupKeys(const uint32_t (&key)[KPT], const uint32_t (&rank)[PACKED ? KPT / 2 : KPT],
                                            uint32_t shift, const uint32_t* waveBase, uint32_t* sorted,
                                            uint32_t (&slots)[KEEP ? KPT / 2 : 1], uint32_t slotCount = KPT) {
  constexpr int CHUNK = (!DYN && KPT % 8 == 0) ? 8 : 4;
#pragma unroll
  for (int base = 0; base < KPT; base += CHUNK) {
    if (DYN && (uint32_t)base >= slotCount) break;
    uint32_t p[CHUNK];
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) p[c] = waveBase[(key[base + c] >> shift) & MASK];
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) {
      const int i = base + c;
      uint32_t r;
      if constexpr (PACKED)
        r = (rank[i / 2] >> (16 * (i % 2))) & 0xFFFFu;
      else
        r = rank[i];
      p[c] = StagingSlot<STAGE>(p[c] + r);
      sorted[p[c]] = key[i];
      if constexpr (KEEP) {
        if (c % 2 == 1) {
          slots[i / 2] = p[c - 1] | (p[c] << 16);
          asm volatile("" : "+v"(slots[i / 2]));  // pack now, not when first used
        }
      }
    }
  }
}

// The compiler would otherwise keep every key's LDS counter address (&waveHist[digit], computed for
// the ranking) alive until the regroup: KPT more registers across the scan and the barriers.
template <int KPT>
__device__ __forceinline__ void ForgetDerivedValues(uint32_t (&key)[KPT]) {
#pragma unroll
  for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(key[i]));
}

// ---- scatter of a staged (sub-)tile: four consecutive sorted positions per lane ---------------
// The staging buffer is sorted by digit, so the four keys of a quad almost always share their digit
// (runs are 64-128 keys on uniform data) and go to four consecutive words: one 16-byte LDS read and
// one 16-byte store (4-byte aligned: gfx950 global stores need no natural alignment) instead of four
// of each; consecutive lanes still cover consecutive addresses.
//
// A quad that straddles a run boundary is rare per lane, but SOME lane of almost every wave has one
// (87 % of the wave-level quads at 128-key runs), and a wave walks through every branch any of its
// lanes takes: handled inside the main loop, word-by-word stores cost the pass 6-7 % (measured with
// the timing ablation 32).  So the main loop only stores whole single-digit quads, and the
// straddling quads are done afterwards by the threads that know where they are: thread d (< 256)
// holds the tile-local start of digit d's run from the scan; if that start is not a multiple of
// four, the quad around it straddles, and every straddling quad inside the valid range contains the
// start of some non-empty run.  Thread 0 (digit 0 starts at 0) takes the quad cut by a ragged end.
// Two runs shorter than four keys can put the same quad on two threads: the same words go to the same
// addresses twice.
__device__ __forceinline__ void StoreWord(uint32_t* base, uint32_t index, uint32_t value) {
  // 32-bit byte offset (N < 2^30) on a 64-bit base: one address register instead of a 64-bit pair
  *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + (uint64_t)(index * 4u)) = value;
}

// Tile-local position of the straddling quad this thread is responsible for, or ~0u.
__device__ __forceinline__ uint32_t BoundaryQuad(int tid, uint32_t myRunStart, uint32_t myRunLength, uint32_t valid) {
  uint32_t quad = ~0u;
  if (tid < 256 && myRunLength != 0 && (myRunStart & 3u) != 0 && myRunStart < valid) quad = myRunStart & ~3u;
  if (tid == 0 && (valid & 3u) != 0) quad = valid & ~3u;
  return quad;
}

template <uint32_t STAGE>
__device__ __forceinline__ void StoreBoundaryQuad(const uint32_t* sorted, const uint32_t* offset, uint32_t* out,
                                                  uint32_t quad, uint32_t valid, uint32_t packedDigits) {
  const u32x4 q = *reinterpret_cast<const u32x4*>(&sorted[StagingSlot<STAGE>(quad)]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t d = (packedDigits >> (8 * c)) & 0xFFu;
    if (quad + c < valid) StoreWord(out, offset[d] + quad + c, q[c]);
  }
}

// KEEP_DIGITS (key+value): digits[j] = first | last << 8 digit of main-loop quad j, and
// boundaryDigits = the four digits of this thread's boundary quad, for the value phase.
// Quads per batch: the 16-byte LDS reads of a batch are issued together, then its offset lookups,
// then its stores -- left alone the compiler does one quad at a time, two dependent LDS round trips
// each, with only four waves per SIMD to hide them.
template <int KPT, bool KV>
constexpr int ScatterBatch() {
  constexpr int quads = KPT / 4;
  // key+value: the values and their slots are live as well (and 1024x16 must stay within the 64
  // registers that let two workgroups share a CU)
  constexpr int want = KV ? (KPT >= 32 ? 4 : 2) : 8;
  return quads % want == 0 ? want : (quads % 4 == 0 ? 4 : (quads % 2 == 0 ? 2 : 1));
}

// DYN: the frame holds `valid` <= THREADS * slots keys; batches of quads beyond them are not read at all.
template <int THREADS, int KPT, bool KEEP_DIGITS, bool DYN = false>
__device__ __forceinline__ void ScatterStagedKeys(const uint32_t* sorted, const uint32_t* offset, uint32_t* out,
                                                  uint32_t valid, uint32_t shift, int tid, uint32_t boundaryQuad,
                                                  uint32_t (&digits)[KEEP_DIGITS ? KPT / 4 : 1],
                                                  uint32_t& boundaryDigits, bool nt = false) {
  constexpr uint32_t STAGE = THREADS * KPT;
  constexpr int B = DYN ? (KPT % 16 == 0 ? 4 : 1) : ScatterBatch<KPT, KEEP_DIGITS>();
#pragma unroll
  for (int j0 = 0; j0 < KPT / 4; j0 += B) {
    if (DYN && 4u * (uint32_t)j0 * THREADS >= valid) break;
    u32x4 k4[B];
    uint32_t o[B];
    bool whole[B];
#pragma unroll
    for (int b = 0; b < B; ++b) k4[b] = *reinterpret_cast<const u32x4*>(&sorted[4u * (tid + (j0 + b) * THREADS)]);
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const uint32_t p = StagingSlot<STAGE>(4u * (tid + (j0 + b) * THREADS));  // involution: the sorted position
      const uint32_t d0 = (k4[b][0] >> shift) & 0xFFu, d3 = (k4[b][3] >> shift) & 0xFFu;
      whole[b] = p + 3 < valid && d0 == d3;
      o[b] = offset[d0] + p;
      asm volatile("" : "+v"(o[b]));  // fetched here, for every quad: not sunk into the conditional store
      if (KEEP_DIGITS) digits[j0 + b] = d0 | (d3 << 8);
    }
#pragma unroll
    for (int b = 0; b < B; ++b)
      if (whole[b]) StoreQuad(out, o[b], k4[b], nt);
  }
  if (boundaryQuad != ~0u) {
    const u32x4 k4 = *reinterpret_cast<const u32x4*>(&sorted[StagingSlot<STAGE>(boundaryQuad)]);
    boundaryDigits = ((k4[0] >> shift) & 0xFFu) | (((k4[1] >> shift) & 0xFFu) << 8) |
                     (((k4[2] >> shift) & 0xFFu) << 16) | (((k4[3] >> shift) & 0xFFu) << 24);
    StoreBoundaryQuad<STAGE>(sorted, offset, out, boundaryQuad, valid, boundaryDigits);
  }
}

template <int THREADS, int KPT, bool DYN = false>
__device__ __forceinline__ void ScatterStagedValues(const uint32_t* sorted, const uint32_t* offset, uint32_t* out,
                                                    uint32_t valid, int tid, uint32_t boundaryQuad,
                                                    const uint32_t (&digits)[KPT / 4], uint32_t boundaryDigits, bool nt = false) {
  constexpr uint32_t STAGE = THREADS * KPT;
  constexpr int B = DYN ? (KPT % 16 == 0 ? 4 : 1) : ScatterBatch<KPT, false>();
#pragma unroll
  for (int j0 = 0; j0 < KPT / 4; j0 += B) {
    if (DYN && 4u * (uint32_t)j0 * THREADS >= valid) break;
    u32x4 v4[B];
    uint32_t o[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const uint32_t p = StagingSlot<STAGE>(4u * (tid + (j0 + b) * THREADS));
      o[b] = offset[digits[j0 + b] & 0xFFu] + p;
      v4[b] = *reinterpret_cast<const u32x4*>(&sorted[4u * (tid + (j0 + b) * THREADS)]);
      asm volatile("" : "+v"(o[b]));
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const uint32_t p = StagingSlot<STAGE>(4u * (tid + (j0 + b) * THREADS));
      const uint32_t d0 = digits[j0 + b] & 0xFFu, d3 = digits[j0 + b] >> 8;
      if (p + 3 < valid && d0 == d3)
        StoreQuad(out, o[b], v4[b], nt);
    }
  }
  if (boundaryQuad != ~0u)
    StoreBoundaryQuad<STAGE>(sorted, offset, out, boundaryQuad, valid, boundaryDigits);
}

// Key+value tiles replay the permutation for the values through the SAME staging buffer after the
// keys have left it (like the reference, downsweep.slang:208-224): the LDS footprint equals the
// keys-only one, so two workgroups fit per CU (keys and values staged together would need 128 KiB
// at T = 16384: one workgroup per CU and nothing to overlap its waits with).
template <int THREADS, int KPT, bool KV>
constexpr size_t OnesweepLdsWords() {
  // staging buffer (keys, then values) | per-wave digit counters.  Everything else lives inside
  // those two at times when they are idle: ticket + scan scratch at the front of the staging
  // buffer (before the regroup), look-back scratch at the bottom and the per-digit scatter offsets
  // in the top 256 words of the counters (after the regroup).  1024 x 16 is then exactly 80 KiB:
  // two workgroups per CU.
  return (size_t)THREADS * KPT + (size_t)(THREADS / 64) * 256;
}

// Waves per SIMD the register allocation must leave room for: two workgroups per CU whenever the
// LDS footprint allows two (every geometry except key+value tiles wider than 16384).
template <int THREADS, int KPT>
constexpr int MinWavesPerSimd() {
  // as many workgroups per CU as the LDS footprint allows (at most 8 waves per SIMD)
  constexpr size_t lds = OnesweepLdsWords<THREADS, KPT, false>() * 4;
  constexpr int workgroups = (int)((160 * 1024) / lds);
  constexpr int waves = workgroups * THREADS / 256;
  return waves > 8 ? 8 : (waves < 1 ? 1 : waves);
}

// DYN: even-split tiles (PlanTiles in vrdx_api.cpp).  A sort of at most one round of tiles is split EVENLY over all
// the CUs instead of into tiles of the kernel's capacity: every wave takes a.slots (a multiple of 4, < KPT) slots of
// 64 keys, a tile is a.slots * THREADS keys, and the loops over the slots stop there, so that a tile costs what its
// keys cost.  The waves still cover the tile in memory order, pads (the sort's last tile only) still sit at its end.
template <int THREADS, int KPT, bool KV, bool ATOMIC_RANK, bool DYN>
__device__ __forceinline__ void OnesweepBody(const OnesweepArgs a) {
  constexpr int WAVES = THREADS / 64;
  constexpr uint32_t TILE = THREADS * KPT;
  static_assert(THREADS >= 256 && THREADS % 256 == 0, "one thread per digit, whole look-back groups");
  static_assert(WAVES * 256 >= 256 * (1 + 2 * (THREADS / 256)) + 2 + 256,
                "look-back scratch and the scatter offsets alias the wave counters");

  static_assert(KPT % 4 == 0, "digits of the value phase are packed four to a register");

  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* const sorted = smem;                                    // TILE: keys (then values) regrouped by digit
  uint32_t* const waveHist = smem + TILE;                           // WAVES x 256, then look-back scratch
  uint32_t* const tileOffset = waveHist + (WAVES - 1) * 256;        // 256: global base - tile-local base (after the regroup)
  uint32_t* const scanScratch = smem;                               // 8   (before the regroup)
  uint32_t* const misc = smem + 8;                                  // [0] ticket (before the regroup)
  uint32_t* const planFlags = smem + 16;                            // 32 (before the regroup)
  uint32_t* const passDigitCounts = smem + 64;                      // 256 (before the regroup; block sums only)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
#ifdef VRDX_TRACE
  uint64_t stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  VRDX_STAMP(0);

  // Launches 1-3 of a sort whose hybrid plan applies have nothing to do, and launch 0 has said so in one word: they
  // return after one load instead of after the table, the ticket, the votes and a barrier (4 -> 2 us per empty launch).
  // (Not compiled into the key+value 1024x32 form with tiles of full capacity, the kernel of the 2^25 headline: the two
  // lines cost it two registers and 1.7 % there, measured.  Key+value sorts of 4.4 ... 8.1 M pairs do meet that form
  // with the hybrid plan recorded: their launches 1-3 take the long way to the same verdict, ~2 us each.  Running them on
  // the form with run-time slot counts instead -- which has the word, but fetches its values late -- measured the same
  // within 1 %, VRDX_EVEN_SPLIT=1, so they stay on this one.)
  constexpr bool kVerdictWord = !(KV && KPT == 32 && !DYN);
  if constexpr (kVerdictWord) {
    if (a.hybridCap != 0 && a.pass != 0 && *a.planWord == 1u) return;
  }
  // the MSD plan (recorded in front of launch 0) has taken the sort: nothing left for the passes
  if (a.planInFront != 0 && (*a.planWord & kMsdVerdictMask) >= kMsdVerdictRuns) return;

  const uint32_t n = ElementCount(a.maxCount, a.countPtr);
  uint32_t key[KPT];
  const PassCounts<THREADS> passCounts = LoadPassCounts<THREADS>(a.histogramTable, tid);
  if (tid == 0) misc[0] = atomicAdd(a.ticketCur, 1u);
  PublishPassVotes<THREADS>(passCounts, n, a.hybridCap, tid, planFlags);
  // block sums (sorts of one round, four-pass plan only: the digit is the pass): every tile needs this pass's 256 global
  // counts for the digit base -- they are in the registers of the threads [256 * pass, 256 * pass + 256) already
  // (built into the 32768-key geometry only, the one the plan ever selects it for: the smaller ones have no register to spare)
  constexpr bool kBlockSumsBuilt = THREADS == 1024 && KPT == 32;
  const bool blockSums = kBlockSumsBuilt && a.blockCur != nullptr;
  if constexpr (kBlockSumsBuilt) {
    if (blockSums && ((uint32_t)tid >> 8) == a.pass) passDigitCounts[tid & 255] = passCounts.v[0];
  }
  for (int i = tid; i < WAVES * 256; i += THREADS) waveHist[i] = 0;
  LdsBarrier();
  VRDX_STAMP(1);

  const uint32_t tile = misc[0];
  const PassPlan plan = ReadPassPlan(planFlags, a.pass, a.hybridCap);
  if constexpr (kVerdictWord) {
    if (a.hybridCap != 0 && a.pass == 0 && tile == 0 && tid == 0)  // the verdict, for the launches behind this one
      *a.planWord = HybridByte(planFlags, a.hybridCap) >= 0 ? 1u : 2u;
  }
  const uint32_t shift = 8u * plan.digit;
  const uint32_t* const keysIn = plan.fromScratch ? a.keysScratch : a.keysCaller;
  uint32_t* const keysOut = plan.fromScratch ? a.keysCaller : a.keysScratch;
  const uint32_t* const valuesIn = plan.fromScratch ? a.valuesScratch : a.valuesCaller;
  uint32_t* const valuesOut = plan.fromScratch ? a.valuesCaller : a.valuesScratch;
  if (plan.skip) {
    // Nothing moves in this pass; only the next pass's status rows and ticket are put in order
    // (workgroup b stands in for tile b: there is no look-back here that would care).
    if (a.statusNext != nullptr) {
      if (tid < 256 && blockIdx.x < a.statusRows) a.statusNext[blockIdx.x * VRDX_RADIX + tid] = 0;
      if (tid < 256 && blockIdx.x < a.blockRows) a.blockNext[blockIdx.x * VRDX_RADIX + tid] = 0;
      if (blockIdx.x == 0 && tid == 0) *a.ticketNext = 0;
    }
    return;
  }
  // DYN: the first a.fullTiles tiles take a.slots slots per wave, the tiles behind them a.tailSlots (PlanTiles)
  const TileSpan span = DYN ? SpanOfTile(tile, n, a.slots, a.tailSlots, a.fullTiles, THREADS)
                            : SpanOfTile(tile, n, (uint32_t)KPT, (uint32_t)KPT, ~0u, THREADS);
  if (span.beyond) return;  // uniform for the whole workgroup
  const uint32_t slots = DYN ? span.slots : (uint32_t)KPT;  // per wave
  const uint32_t frame = slots * THREADS;                   // keys this tile can hold (TILE unless DYN)
  const bool lastTile = span.last;
  const uint32_t tileStart = span.start;
  const uint32_t valid = (n - tileStart) < frame ? (n - tileStart) : frame;
  const uint32_t tileEnd = tileStart + valid;

  // Housekeeping for the NEXT pass (its kernel starts after this one has drained): clear my row
  // of the other status region and the other ticket.
  if (a.statusNext != nullptr) {
    if (tid < 256 && tile < a.statusRows) a.statusNext[tile * VRDX_RADIX + tid] = 0;
    if (tid < 256 && tile < a.blockRows) a.blockNext[tile * VRDX_RADIX + tid] = 0;
    if (tile == 0 && tid == 0) *a.ticketNext = 0;
  }

  // ---- load: wave-striped, so that (slot, lane) order == memory order inside a wave ----------
  // (the values are fetched once the keys have been staged)
  uint32_t val[KV ? KPT : 1];
  const uint32_t loadBase = tileStart + wave * (slots * 64) + lane;
  // Ragged last tile: pad with 0xFFFFFFFF like the reference (downsweep.slang:81).  Pads sit at the
  // highest memory positions of the tile and have digit 255 in every pass, so the stable ranking
  // puts them at tile-local positions >= valid, where nothing is written.
  const bool streaming = KV && StreamingLoads(KV, n);
  LoadTile<KPT, DYN>(keysIn, loadBase, tileEnd, valid == frame, 0xFFFFFFFFu, key, streaming, slots);
  if (plan.copy) {  // identity permutation that has to change buffers: copy the tile
    StoreStriped<KPT, DYN>(keysOut, loadBase, tileEnd, valid == frame, key, slots);
    if constexpr (KV) {
      LoadTile<KPT, DYN>(valuesIn, loadBase, tileEnd, valid == frame, 0u, val, streaming, slots);
      StoreStriped<KPT, DYN>(valuesOut, loadBase, tileEnd, valid == frame, val, slots);
    }
    return;
  }

  // ---- rank inside the wave (memory order) ---------------------------------------------------
  // key+value: ranks / positions live until the values are staged, so they are kept packed two to
  // a register (< TILE <= 65536) -- two workgroups per CU must fit the register file with no spill
  // at all (measured: 120 bytes of scratch per lane made a pass 30x slower).
  constexpr bool PACKED = KV;
  static_assert(!KV || TILE <= 65536, "packed 16-bit positions");
  uint32_t rank[PACKED ? KPT / 2 : KPT];
  if constexpr (ATOMIC_RANK)
    RankAtomic<KPT, PACKED, DYN>(key, shift, waveHist + wave * 256, lane, rank, slots);
  else
    RankBallot<KPT, PACKED, DYN>(key, shift, waveHist + wave * 256, lane, rank, slots);
  ForgetDerivedValues<KPT>(key);
  // key+value, early form: the values start their trip now and land during the scan and the regroup
  // (The form with run-time slot counts always fetches them late: its loops end in branches, the values would be live
  // across all of them and the kernel would need 140 registers -- 48 bytes of scratch per lane, 5-10 % slower, round 3.)
  constexpr bool kEarlyValuesBuilt = KV && !DYN;
  if constexpr (kEarlyValuesBuilt) {
    if (a.earlyValues) LoadTile<KPT, DYN>(valuesIn, loadBase, tileEnd, valid == frame, 0u, val, streaming, slots);  // pad: downsweep.slang:85
  }
  LdsBarrier();
  VRDX_STAMP(2);

  // ---- tile histogram, aggregate publish, tile-local digit offsets ---------------------------
  uint32_t count = 0;
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) count += waveHist[w * 256 + tid];
    // Tile 0 never publishes a bare aggregate: its inclusive value carries the global digit base.
    // (Block sums: every tile but the last publishes its aggregate and adds it to its block's row.)
    if ((tile != 0 || blockSums) && !lastTile)
      StoreStatus(&a.statusCur[tile * VRDX_RADIX + tid], (VRDX_FLAG_AGGREGATE << VRDX_FLAG_SHIFT) | count);
    if (blockSums && !lastTile) AddToBlock(&a.blockCur[(tile / VRDX_BLOCK_TILES) * VRDX_RADIX + tid], count);
  }
  const uint32_t tileExclusive = BlockExclusiveScan256(tid < 256 ? count : 0u, scanScratch, tid);
  uint32_t exclusive = 0;
  if (tile == 0 || blockSums) {
    // spine.slang:62-83 equivalent: exclusive scan of this pass's 256 global digit counts (block sums: by every tile,
    // the chain that would carry the base along from tile 0 does not exist)
    const uint32_t g = tid < 256 ? (blockSums ? passDigitCounts[tid] : a.histogramTable[plan.digit * VRDX_RADIX + tid]) : 0u;
    exclusive = BlockExclusiveScan256(g, scanScratch + 4, tid);
  }
  if (tid < 256) {
    uint32_t run = tileExclusive;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const uint32_t c = waveHist[w * 256 + tid];
      waveHist[w * 256 + tid] = run;
      run += c;
    }
  }
  LdsBarrier();
  VRDX_STAMP(3);

  // ---- regroup the keys by digit in LDS; key+value keeps the positions for the values ----------
  uint32_t packedPos[KV ? KPT / 2 : 1];
  RegroupKeys<KPT, TILE, PACKED, KV, DYN>(key, rank, shift, waveHist + wave * 256, sorted, packedPos, slots);
  LdsBarrier();  // waveHist is dead from here on: the look-back reuses it as scratch
  VRDX_STAMP(4);

  // ---- decoupled look-back over the preceding tiles, then publish the inclusive prefix --------
  uint32_t lookBackTrace = 0;
  if constexpr (kBlockSumsBuilt) {
    if (blockSums && tile != 0)
      exclusive += BlockPrefix<THREADS>(a.statusCur, a.blockCur, tile, tid, waveHist, a.failure, a.stickyFailure, a.spinLimit,
                                        &lookBackTrace);
  }
  if (!blockSums && tile != 0)
    exclusive = LookBack<THREADS>(a.statusCur, tile, tid, waveHist, a.failure, a.stickyFailure, a.spinLimit, &lookBackTrace);
  TestDelayFirstTile(tile, a.spinLimit);
  if (tid < 256) {
    if (!lastTile && !blockSums)
      StoreStatus(&a.statusCur[tile * VRDX_RADIX + tid],
                  (VRDX_FLAG_INCLUSIVE << VRDX_FLAG_SHIFT) | ((exclusive + count) & VRDX_VALUE_MASK));
    tileOffset[tid] = exclusive - tileExclusive;
  }
  LdsBarrier();
  VRDX_STAMP(5);

  // Key+value, late form (measurements only, VRDX_KV_EARLY_VALUES=0): the values are fetched now and
  // the key scatter covers their latency.  It used to win by 1-3 % for sorts of four and more rounds
  // of tiles; on the final kernels the early form is as fast or faster everywhere.  Issued right
  // before the look-back the loads queue in front of its agent-scope status reads (6 -> 9 us, measured).
  if constexpr (KV) {
    if (!kEarlyValuesBuilt || !a.earlyValues)
      LoadTile<KPT, DYN>(valuesIn, loadBase, tileEnd, valid == frame, 0u, val, streaming, slots);
  }

  // ---- scatter (ScatterStagedKeys above); key+value replays the permutation for the values ------
  uint32_t digits[KV ? KPT / 4 : 1];  // key+value: first and last digit of every quad, for the value phase
  uint32_t boundaryDigits = 0;
  const uint32_t boundaryQuad = BoundaryQuad(tid, tileExclusive, count, valid);
  const bool ntStores = VRDX_NT_LAST_PASS != 0 && a.pass == VRDX_PASSES - 1;
  ScatterStagedKeys<THREADS, KPT, KV, DYN>(sorted, tileOffset, keysOut, valid, shift, tid, boundaryQuad, digits,
                                           boundaryDigits, ntStores);
  if constexpr (KV) {
    LdsBarrier();  // every key has left the staging buffer
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
      sorted[(packedPos[i / 2] >> (16 * (i % 2))) & 0xFFFFu] = val[i];
    }
    LdsBarrier();
    ScatterStagedValues<THREADS, KPT, DYN>(sorted, tileOffset, valuesOut, valid, tid, boundaryQuad, digits,
                                           boundaryDigits, ntStores);
  }
#ifdef VRDX_TRACE
  VRDX_STAMP(6);
  if (a.trace != nullptr && tid == 0) {
    uint32_t xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    stamps[7] = ((uint64_t)(xcc & 0xF) << 60) | ((uint64_t)lookBackTrace << 24) | (blockIdx.x & 0xFFFFFFu);
    for (int i = 0; i < 8; ++i) a.trace[(size_t)tile * 8 + i] = stamps[i];
  }
#endif
}

template <int THREADS, int KPT, bool KV, bool ATOMIC_RANK, bool DYN>
__global__ __launch_bounds__(THREADS, (MinWavesPerSimd<THREADS, KPT>())) void onesweep_kernel(OnesweepArgs a) {
  OnesweepBody<THREADS, KPT, KV, ATOMIC_RANK, DYN>(a);
}

// ---------------------------------------------------------------------------------------------
// onesweep_pair_kernel: one workgroup = TWO consecutive sub-tiles of THREADS*KPT keys, ONE status
// row, ONE ticket and ONE look-back for both.
// ---------------------------------------------------------------------------------------------
// Why: the per-tile fixed costs (ticket 0.85 us, scan 0.9 us, look-back 3.1 us of a 16.5 us tile at
// 1024x32, profiles/r01_phase_trace_1024x32_keys.txt) shrink only with the tile, and the tile is
// capped by the LDS staging buffer (T words) and by the registers (KPT keys per lane).  Two
// sub-tiles A and B go through the SAME staging buffer one after the other and hold registers for
// one sub-tile's keys at a time, so the tile doubles at the same LDS and register footprint:
//
//   load A, rank A | issue the loads of B | scan A, regroup A into the staging buffer
//   rank B, scan B | publish {AGGREGATE, countA + countB} | look-back (once)
//   publish {INCLUSIVE, ...} | scatter A | regroup B | scatter B
//
// B's loads fly while A is scanned and regrouped.  Inside the tile A precedes B (B's digit base is
// the tile's base + countA), so the result is the same stable permutation.
//
// Keys-only sorts with the one-atomic ranking only (the key+value and the ballot forms spilled and were never
// selected).
//
// LDS: staging (THREADS*KPT) | wave counters (WAVES*256) | look-back scratch + scan scratch +
//      ticket | digit offsets of A and of B (2 x 256).
template <int THREADS, int KPT>
constexpr size_t PairLdsWords() {
  return (size_t)THREADS * KPT + (size_t)(THREADS / 64) * 256 + (size_t)256 * (2 + 2 * (THREADS / 256)) + 512;
}

template <int THREADS, int KPT>
constexpr int PairMinWavesPerSimd() {
  constexpr int workgroups = (int)((160 * 1024) / (PairLdsWords<THREADS, KPT>() * 4));
  constexpr int waves = workgroups * THREADS / 256;
  return waves > 8 ? 8 : (waves < 1 ? 1 : waves);
}

template <int THREADS, int KPT, bool DYN>  // DYN: even-split tiles, see onesweep_kernel
__device__ __forceinline__ void OnesweepPairBody(const OnesweepArgs a) {
  constexpr int WAVES = THREADS / 64;
  constexpr int GROUPS = THREADS / 256;
  constexpr uint32_t SUB = THREADS * KPT;  // keys per sub-tile == staging buffer words
  static_assert(THREADS >= 256 && THREADS % 256 == 0, "one thread per digit, whole look-back groups");
  static_assert(KPT % 4 == 0, "quads");
  static_assert(SUB <= 65536, "packed 16-bit positions");
  static_assert(PairLdsWords<THREADS, KPT>() * 4 <= 160 * 1024, "fits the CU's LDS");

  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* const sorted = smem;                                // SUB
  uint32_t* const waveHist = smem + SUB;                        // WAVES x 256
  uint32_t* const lookScratch = waveHist + WAVES * 256;         // 256 * (1 + 2 * GROUPS) + 2 used by LookBack
  uint32_t* const scanScratch = lookScratch + 256 * (1 + 2 * GROUPS) + 8;  // 8
  uint32_t* const misc = scanScratch + 8;                       // [0] ticket
  uint32_t* const offsetA = lookScratch + 256 * (2 + 2 * GROUPS);  // 256: global base - local base, sub-tile A
  uint32_t* const offsetB = offsetA + 256;                      // 256: same for sub-tile B

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
#ifdef VRDX_TRACE
  uint64_t stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  VRDX_STAMP(0);

  if (a.planInFront != 0 && (*a.planWord & kMsdVerdictMask) >= kMsdVerdictRuns) return;  // the MSD plan has taken the sort (see onesweep_kernel)
  const uint32_t n = ElementCount(a.maxCount, a.countPtr);
  const PassCounts<THREADS> passCounts = LoadPassCounts<THREADS>(a.histogramTable, tid);
  if (tid == 0) misc[0] = atomicAdd(a.ticketCur, 1u);
  PublishPassVotes<THREADS>(passCounts, n, a.hybridCap, tid, misc + 1);
  // block sums (see onesweep_kernel): this pass's 256 global counts, parked in the idle staging buffer
  const bool blockSums = THREADS == 1024 && a.blockCur != nullptr;
  uint32_t* const passDigitCounts = sorted + 64;
  if constexpr (THREADS == 1024) {
    if (blockSums && ((uint32_t)tid >> 8) == a.pass) passDigitCounts[tid & 255] = passCounts.v[0];
  }
  for (int i = tid; i < WAVES * 256; i += THREADS) waveHist[i] = 0;
  LdsBarrier();
  VRDX_STAMP(1);

  const uint32_t tile = misc[0];
  const PassPlan plan = ReadPassPlan(misc + 1, a.pass, a.hybridCap);
  const uint32_t shift = 8u * plan.digit;
  const uint32_t* const keysIn = plan.fromScratch ? a.keysScratch : a.keysCaller;
  uint32_t* const keysOut = plan.fromScratch ? a.keysCaller : a.keysScratch;
  if (plan.skip) {  // see onesweep_kernel
    if (a.statusNext != nullptr) {
      if (tid < 256 && blockIdx.x < a.statusRows) a.statusNext[blockIdx.x * VRDX_RADIX + tid] = 0;
      if (tid < 256 && blockIdx.x < a.blockRows) a.blockNext[blockIdx.x * VRDX_RADIX + tid] = 0;
      if (blockIdx.x == 0 && tid == 0) *a.ticketNext = 0;
    }
    return;
  }
  // DYN: the first a.fullTiles tiles take a.slots slots per wave and sub-tile, the tiles behind them a.tailSlots
  const TileSpan span = DYN ? SpanOfTile(tile, n, a.slots, a.tailSlots, a.fullTiles, 2 * THREADS)
                            : SpanOfTile(tile, n, (uint32_t)KPT, (uint32_t)KPT, ~0u, 2 * THREADS);
  if (span.beyond) return;  // uniform for the whole workgroup
  const uint32_t slots = DYN ? span.slots : (uint32_t)KPT;  // per wave and sub-tile
  const uint32_t sub = slots * THREADS;                     // keys per sub-tile (SUB unless DYN)
  const bool lastTile = span.last;
  const uint32_t tileStart = span.start;
  const uint32_t left = n - tileStart;
  const uint32_t validA = left < sub ? left : sub;
  const uint32_t validB = left > sub ? (left - sub < sub ? left - sub : sub) : 0u;
  const uint32_t endA = tileStart + validA, endB = tileStart + sub + validB;

  if (a.statusNext != nullptr) {
    if (tid < 256 && tile < a.statusRows) a.statusNext[tile * VRDX_RADIX + tid] = 0;
    if (tid < 256 && tile < a.blockRows) a.blockNext[tile * VRDX_RADIX + tid] = 0;
    if (tile == 0 && tid == 0) *a.ticketNext = 0;
  }

  constexpr bool PACKED = true;  // ranks and positions < SUB <= 65536, two to a register
  const uint32_t loadBaseA = tileStart + wave * (slots * 64) + lane;
  const uint32_t loadBaseB = loadBaseA + sub;
  uint32_t* const myHist = waveHist + wave * 256;

  // ---- sub-tile A: load, rank ------------------------------------------------------------------
  uint32_t keyA[KPT];
  LoadStriped<KPT, false, DYN>(keysIn, loadBaseA, endA, validA == sub, 0xFFFFFFFFu, keyA, slots);  // pad: downsweep.slang:81
  if (plan.copy) {  // identity permutation that has to change buffers: copy both sub-tiles
    StoreStriped<KPT, DYN>(keysOut, loadBaseA, endA, validA == sub, keyA, slots);
    LoadStriped<KPT, false, DYN>(keysIn, loadBaseB, endB, validB == sub, 0xFFFFFFFFu, keyA, slots);
    StoreStriped<KPT, DYN>(keysOut, loadBaseB, endB, validB == sub, keyA, slots);
    return;
  }
  uint32_t rankA[PACKED ? KPT / 2 : KPT];
  RankAtomic<KPT, PACKED, DYN>(keyA, shift, myHist, lane, rankA, slots);
  ForgetDerivedValues<KPT>(keyA);

  // ---- sub-tile B's keys start their trip now ---------------------------------------------------
  __builtin_amdgcn_sched_barrier(0);  // not earlier: A's keys and ranks are live
  uint32_t keyB[KPT];
  LoadStriped<KPT, false, DYN>(keysIn, loadBaseB, endB, validB == sub, 0xFFFFFFFFu, keyB, slots);
  LdsBarrier();
  VRDX_STAMP(2);

  // ---- A: tile histogram -> local digit offsets -------------------------------------------------
  uint32_t countA = 0;
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) countA += waveHist[w * 256 + tid];
  }
  const uint32_t localA = BlockExclusiveScan256(tid < 256 ? countA : 0u, scanScratch, tid);
  uint32_t exclusive = 0;
  if (tile == 0 || blockSums) {
    // spine.slang:62-83 equivalent: exclusive scan of this pass's 256 global digit counts (block sums: by every tile)
    const uint32_t g = tid < 256 ? (blockSums ? passDigitCounts[tid] : a.histogramTable[plan.digit * VRDX_RADIX + tid]) : 0u;
    exclusive = BlockExclusiveScan256(g, scanScratch + 4, tid);
  }
  if (tid < 256) {
    uint32_t run = localA;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const uint32_t c = waveHist[w * 256 + tid];
      waveHist[w * 256 + tid] = run;
      run += c;
    }
  }
  LdsBarrier();

  // ---- A: regroup into the staging buffer; from here to the next barrier a wave only touches ITS
  // row of the counters, so it can clear the row and rank B without waiting for the others -------
  uint32_t unusedSlots[1];
  RegroupKeys<KPT, SUB, PACKED, false, DYN>(keyA, rankA, shift, myHist, sorted, unusedSlots, slots);
#pragma unroll
  for (int i = 0; i < 4; ++i) myHist[lane + 64 * i] = 0;

  uint32_t rankB[PACKED ? KPT / 2 : KPT];
  RankAtomic<KPT, PACKED, DYN>(keyB, shift, myHist, lane, rankB, slots);
  ForgetDerivedValues<KPT>(keyB);
  LdsBarrier();
  VRDX_STAMP(3);

  // ---- B: tile histogram; publish the tile's aggregate; local digit offsets ---------------------
  uint32_t countB = 0;
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) countB += waveHist[w * 256 + tid];
    if ((tile != 0 || blockSums) && !lastTile)
      StoreStatus(&a.statusCur[tile * VRDX_RADIX + tid],
                  (VRDX_FLAG_AGGREGATE << VRDX_FLAG_SHIFT) | (countA + countB));
    if (blockSums && !lastTile) AddToBlock(&a.blockCur[(tile / VRDX_BLOCK_TILES) * VRDX_RADIX + tid], countA + countB);
  }
  const uint32_t localB = BlockExclusiveScan256(tid < 256 ? countB : 0u, scanScratch, tid);
  if (tid < 256) {
    uint32_t run = localB;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const uint32_t c = waveHist[w * 256 + tid];
      waveHist[w * 256 + tid] = run;
      run += c;
    }
  }
  LdsBarrier();
  VRDX_STAMP(4);

  // ---- one look-back for both sub-tiles ----------------------------------------------------------
  uint32_t lookBackTrace = 0;
  if constexpr (THREADS == 1024) {
    if (blockSums && tile != 0)
      exclusive += BlockPrefix<THREADS>(a.statusCur, a.blockCur, tile, tid, lookScratch, a.failure, a.stickyFailure,
                                        a.spinLimit, &lookBackTrace);
  }
  if (!blockSums && tile != 0)
    exclusive = LookBack<THREADS>(a.statusCur, tile, tid, lookScratch, a.failure, a.stickyFailure, a.spinLimit,
                                  &lookBackTrace);
  TestDelayFirstTile(tile, a.spinLimit);
  if (tid < 256) {
    if (!lastTile && !blockSums)
      StoreStatus(&a.statusCur[tile * VRDX_RADIX + tid],
                  (VRDX_FLAG_INCLUSIVE << VRDX_FLAG_SHIFT) | ((exclusive + countA + countB) & VRDX_VALUE_MASK));
    offsetA[tid] = exclusive - localA;
    offsetB[tid] = exclusive + countA - localB;
  }
  LdsBarrier();
  VRDX_STAMP(5);

  // ---- scatter A ---------------------------------------------------------------------------------
  uint32_t digits[1];
  uint32_t boundaryDigits = 0;
  const uint32_t boundaryQuadA = BoundaryQuad(tid, localA, countA, validA);
  const uint32_t boundaryQuadB = BoundaryQuad(tid, localB, countB, validB);
  const bool ntStores = VRDX_NT_LAST_PASS != 0 && a.pass == VRDX_PASSES - 1;
  ScatterStagedKeys<THREADS, KPT, false, DYN>(sorted, offsetA, keysOut, validA, shift, tid, boundaryQuadA, digits,
                                              boundaryDigits, ntStores);
  LdsBarrier();  // the staging buffer is free again

  // ---- B: regroup, scatter -----------------------------------------------------------------------
  RegroupKeys<KPT, SUB, PACKED, false, DYN>(keyB, rankB, shift, myHist, sorted, unusedSlots, slots);
  LdsBarrier();
  ScatterStagedKeys<THREADS, KPT, false, DYN>(sorted, offsetB, keysOut, validB, shift, tid, boundaryQuadB, digits,
                                              boundaryDigits, ntStores);
#ifdef VRDX_TRACE
  VRDX_STAMP(6);
  if (a.trace != nullptr && tid == 0) {
    uint32_t xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    stamps[7] = ((uint64_t)(xcc & 0xF) << 60) | ((uint64_t)lookBackTrace << 24) | (blockIdx.x & 0xFFFFFFu);
    for (int i = 0; i < 8; ++i) a.trace[(size_t)tile * 8 + i] = stamps[i];
  }
#endif
}

template <int THREADS, int KPT, bool DYN>
__global__ __launch_bounds__(THREADS, (PairMinWavesPerSimd<THREADS, KPT>())) void onesweep_pair_kernel(OnesweepArgs a) {
  OnesweepPairBody<THREADS, KPT, DYN>(a);
}

// ---------------------------------------------------------------------------------------------
// small_sort_kernel: the whole sort in ONE workgroup and ONE launch
// ---------------------------------------------------------------------------------------------
// Below ~16 K elements the six launches of the general path (clear, histogram, four passes) cost
// 30-45 us whatever N is: every launch is one tile's latency chain.  Here one workgroup keeps the
// keys (and values) in registers, and each of the four passes ranks them (the same wave-private
// counters and RankAtomic / RankBallot as the tile kernels), scans the 256 digit counts, moves the
// elements through an LDS staging buffer to their sorted positions and reads them back in
// wave-striped order.  No global histogram, no tickets, no status words; of the storage only the failure word is written.
// Positions >= n hold 0xFFFFFFFF pads (value 0): they are last in memory order and carry the
// largest key, so the stable sort leaves them behind the n real elements, which are what is stored.
// Key+value with THREADS * KPT = 32768 elements: two staging buffers of that size do not fit the CU's LDS, so keys and
// values take turns in ONE (like the pass kernels, and like the reference, downsweep.slang:208-224): two more barriers
// per pass, the same LDS traffic.
template <int THREADS, int KPT, bool KV>
constexpr bool SharedStage() {
  return KV && (size_t)THREADS * KPT * 2 * 4 + (size_t)(THREADS / 64) * 1024 + 64 > 160 * 1024;
}

template <int THREADS, int KPT, bool KV>
constexpr size_t SmallSortLdsWords() {
  return (size_t)THREADS * KPT * (KV && !SharedStage<THREADS, KPT, KV>() ? 2 : 1) + (size_t)(THREADS / 64) * 256 + 16;
}

// The sort of n <= THREADS * KPT elements by the key bytes [0, bytes) inside one workgroup: in[0..n) -> out[0..n)
// (in == out: in place).  Used by small_sort_kernel (the whole sort, bytes = 4) and by bucket_sort_kernel (one
// bucket of the hybrid plan, bytes = 3).
// A wave takes only as many slots of 64 elements as n needs (a multiple of four, like the even-split tiles of
// onesweep_kernel), so a bucket or small sort costs what its elements cost, not what the kernel could hold.
template <int THREADS, int KPT, bool KV, bool ATOMIC_RANK>
__device__ __forceinline__ void SortInWorkgroup(const uint32_t* keysIn, uint32_t* keysOut, const uint32_t* valuesIn,
                                                uint32_t* valuesOut, uint32_t n, uint32_t bytes, uint32_t* smem) {
  constexpr int WAVES = THREADS / 64;
  constexpr uint32_t TILE = THREADS * KPT;
  constexpr bool SHARED = SharedStage<THREADS, KPT, KV>();
  uint32_t* const stagedKeys = smem;                                  // TILE
  uint32_t* const stagedValues = SHARED ? smem : smem + TILE;         // TILE (key+value)
  uint32_t* const waveHist = smem + TILE * (KV && !SHARED ? 2 : 1);   // WAVES x 256
  uint32_t* const scanScratch = waveHist + WAVES * 256;               // 8

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  uint32_t* const myHist = waveHist + wave * 256;
  constexpr bool DYN = KPT >= 8;
  uint32_t slots = KPT;
  uint32_t first = wave * (KPT * 64) + lane;  // element i of this lane: first + 64 * i
  // The ceil(n / 256) chunks of four 64-element slots are dealt out EVENLY over the waves (round 5): wave w takes
  // base + (w < extra) chunks, consecutive in memory -- waves in order, (slot, lane) order inside a wave, so the ranking stays
  // stable; pads (the largest key, last in memory order) only ever sit in the last chunk.  Round 4 gave every wave the same
  // `slots` and let the waves behind n idle: the LDS work followed the bucket's size in steps of 256 elements
  // (profiles/r04_bucket_granule.txt), but a bucket of 16385 elements kept thirteen waves busy with twenty slots each and
  // every phase lasted as long as twenty slots take -- the 7-10 % steps of the size curve wherever the mean bucket crossed a
  // multiple of 4096 (4.2 / 4.5 / 6.0 / 6.3 M elements).  Dealt out evenly, one wave has twenty and the others sixteen.
  if constexpr (DYN) {
    const uint32_t chunks = (n + 255u) / 256u;  // <= WAVES * KPT / 4
    const uint32_t base = chunks / WAVES, extra = chunks % WAVES;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    slots = 4u * (base + (w < extra ? 1u : 0u));
    first = 256u * (w * base + (w < extra ? w : extra)) + lane;
  }

  uint32_t key[KPT];
  uint32_t val[KV ? KPT : 1];
  // (the "full" shortcut would skip the per-element bound check: only when no wave was shortened)
  LoadStriped<KPT, false, DYN>(keysIn, first, n, false, 0xFFFFFFFFu, key, slots);  // pad: downsweep.slang:81
  if constexpr (KV) LoadStriped<KPT, false, DYN>(valuesIn, first, n, false, 0u, val, slots);  // pad: downsweep.slang:85

  // SHARED: ranks, then staging slots, two to a register (< TILE <= 65536): keys, values and positions of 32 elements
  // per lane have to fit 128 registers
  constexpr bool PACKED = SHARED;
  static_assert(!PACKED || TILE <= 65536, "packed 16-bit positions");
#pragma unroll 1
  for (uint32_t shift = 0; shift < 8 * bytes; shift += 8) {
#pragma unroll
    for (int i = 0; i < 4; ++i) myHist[lane + 64 * i] = 0;  // my own row: no barrier needed before ranking
    uint32_t rank[PACKED ? KPT / 2 : KPT];
    if constexpr (ATOMIC_RANK)
      RankAtomic<KPT, PACKED, DYN>(key, shift, myHist, lane, rank, slots);
    else
      RankBallot<KPT, PACKED, DYN>(key, shift, myHist, lane, rank, slots);
    LdsBarrier();
    // The read-back addresses below do not depend on the pass: left alone the compiler computes all KPT of them once,
    // in front of the loop, and keeps them in registers through every pass (the 32768-element key+value form then spills).
    uint32_t firstNow = first;
    asm volatile("" : "+v"(firstNow));

    uint32_t count = 0;
    if (tid < 256) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) count += waveHist[w * 256 + tid];
    }
    const uint32_t exclusive = BlockExclusiveScan256(tid < 256 ? count : 0u, scanScratch, tid);
    if (tid < 256) {
      uint32_t run = exclusive;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = waveHist[w * 256 + tid];
        waveHist[w * 256 + tid] = run;
        run += c;
      }
    }
    LdsBarrier();

    if constexpr (SHARED) {
      // keys through the buffer, then the values through the same slots
#pragma unroll
      for (int base = 0; base < KPT; base += 4) {
        if (DYN && (uint32_t)base >= slots) break;
        uint32_t p[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = myHist[(key[base + c] >> shift) & 0xFFu];  // four reads in flight, then four stores
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = base + c;
          p[c] = StagingSlot<TILE>(p[c] + ((rank[i / 2] >> (16 * (i % 2))) & 0xFFFFu));
          stagedKeys[p[c]] = key[i];
          if (c % 2 == 1) {
            rank[i / 2] = p[c - 1] | (p[c] << 16);
            asm volatile("" : "+v"(rank[i / 2]));  // pack now, not when first used
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // one chunk at a time: hoisting every counter read costs registers this kernel lacks
      }
      LdsBarrier();
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
        key[i] = stagedKeys[StagingSlot<TILE>(firstNow + 64 * i)];
      }
      LdsBarrier();
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
        stagedValues[(rank[i / 2] >> (16 * (i % 2))) & 0xFFFFu] = val[i];
      }
      LdsBarrier();
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
        val[i] = stagedValues[StagingSlot<TILE>(firstNow + 64 * i)];
      }
      // the next pass writes the staging buffer only after two more barriers
    } else {
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
        const uint32_t slot = StagingSlot<TILE>(rank[i] + myHist[(key[i] >> shift) & 0xFFu]);
        stagedKeys[slot] = key[i];
        if constexpr (KV) stagedValues[slot] = val[i];
      }
      LdsBarrier();
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
        const uint32_t slot = StagingSlot<TILE>(firstNow + 64 * i);
        key[i] = stagedKeys[slot];
        if constexpr (KV) val[i] = stagedValues[slot];
      }
      // the next pass writes the staging buffers only after two more barriers
    }
  }

#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
    const uint32_t index = first + 64 * i;
    if (index < n) {
      keysOut[index] = key[i];
      if constexpr (KV) valuesOut[index] = val[i];
    }
  }
}

template <int THREADS, int KPT, bool KV, bool ATOMIC_RANK>
__global__ __launch_bounds__(THREADS) void small_sort_kernel(uint32_t* keys, uint32_t* values, uint32_t maxCount,
                                                              const uint32_t* countPtr, uint32_t* failure) {
  static_assert(THREADS >= 256 && THREADS % 256 == 0, "one thread per digit");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t n = ElementCount(maxCount, countPtr);
  if (threadIdx.x == 0) *failure = 0;  // the one word of storage vrdxHipReadStatus looks at: nothing here can spin
  SortInWorkgroup<THREADS, KPT, KV, ATOMIC_RANK>(keys, keys, values, values, n, VRDX_PASSES, smem);
}

// ---------------------------------------------------------------------------------------------
// bucket_sort_kernel: second half of the hybrid plan of mid-size sorts (see PassPlan)
// ---------------------------------------------------------------------------------------------
// Launch 0 has scattered the elements by byte t of the key (PassPlan) into the scratch arrays: bucket b is the range
// [base[b], base[b] + count[b]) with count = row t of the global histogram and base its exclusive scan.  Workgroup b
// sorts bucket b by the bytes below t with SortInWorkgroup (stable, so equal keys keep the order launch 0 left them
// in, which is their input order) and writes it to the same range of the caller's arrays.  Every workgroup first
// derives, from the same table as the pass kernels, whether the hybrid plan applies at all; otherwise this launch has
// nothing to do (the four-pass plan is running).

template <int THREADS, int KPT, bool KV, bool ATOMIC_RANK>
__global__ __launch_bounds__(THREADS) void bucket_sort_kernel(BucketSortArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int tid = threadIdx.x;
  uint32_t* const flags = smem + 16;  // 32
  const uint32_t verdict = *a.planWord;
  if (verdict == 2u || verdict == 3u) return;  // launch 0's verdict: the four passes are running
  // the same votes, from the same table, as in the pass kernels: every launch of the sort reaches the same verdict
  const uint32_t n = ElementCount(a.maxCount, a.countPtr);
  const PassCounts<THREADS> passCounts = LoadPassCounts<THREADS>(a.histogramTable, tid);
  PublishPassVotes<THREADS>(passCounts, n, a.hybridCap, tid, flags);
  LdsBarrier();
  const int byte = HybridByte(flags, a.hybridCap);
  if (byte < 0) return;  // uniform: the four-pass plan is running
  const uint32_t count = tid < 256 ? a.histogramTable[byte * VRDX_RADIX + tid] : 0u;
  const uint32_t base = BlockExclusiveScan256(count, smem, tid);
  if (tid == (int)blockIdx.x) {
    smem[8] = base;
    smem[9] = count;
  }
  LdsBarrier();
  // (read from LDS, so the compiler takes them for per-lane values: as scalars the four array bases stay out of the
  // vector registers, which the 32768-element key+value form has none to spare of)
  const uint32_t myBase = (uint32_t)__builtin_amdgcn_readfirstlane((int)smem[8]);
  const uint32_t myCount = (uint32_t)__builtin_amdgcn_readfirstlane((int)smem[9]);
  LdsBarrier();  // smem is the sort's from here on
  if (myCount == 0) return;  // uniform
  SortInWorkgroup<THREADS, KPT, KV, ATOMIC_RANK>(a.keysScratch + myBase, a.keysCaller + myBase,
                                                 KV ? a.valuesScratch + myBase : nullptr,
                                                 KV ? a.valuesCaller + myBase : nullptr, myCount, (uint32_t)byte, smem);
}

// ---------------------------------------------------------------------------------------------
// MSD plan (round 5): three ranking steps of 10-11 bits, two trips through memory
// ---------------------------------------------------------------------------------------------
// A four-pass LSD sort moves every key through HBM four times, and each of the four passes costs a CU the same LDS work
// per key (rank, regroup, read back) wherever the pass runs.  What LIMITS the digit width is the wave-private counter
// table: 16 waves x 2^bits counters.  With counters of 16 bits, two to a 32-bit word -- a wave holds at most 2304 keys
// and a workgroup at most 36864, so neither a count nor a position ever carries out of its half -- 2048 digits take the
// 64 KiB that 1024 took, and 32 bits are three steps (10-11 | 11 | 10-11) instead of four.  The first step is a stable
// scatter by the TOP bits through memory; the other two happen inside one workgroup per bucket:
//
//   histogram_msd_kernel   reads the keys once: byte histograms 0..2 like histogram_kernel, and the counts of the top
//                          BITS bits PER TILE of 32768 keys, written out as 16-bit numbers (byte 3's histogram is
//                          their sum) -- the reference's upsweep (upsweep.slang:10-45) for this one digit;
//   spine_msd_kernel       exclusive prefix over the tiles, in place; base and size of every bucket; a bucket beyond
//                          the capacity raises the overflow word (spine.slang:11-84);
//   scatter_msd_kernel     stable scatter by the top bits: base = bucketBase[d] + prefix[tile][d] + rank in the tile
//                          (downsweep.slang:41-224).  No ticket, no status words, no look-back, no spin: tiles are
//                          independent, which is what a row of 2048 status words per tile would have made expensive;
//   bucket_sort2_kernel    every bucket (<= 36864 elements) by its remaining 21-22 bits: two stable passes of <= 11 bits
//                          through the LDS staging buffer; bucket_sort2_half_kernel: the same in 512 threads for buckets
//                          of <= 18432 elements (sorts of up to 18.1 M), two workgroups to a CU.
//
// A stable scatter by the top bits followed by a stable sort of each bucket by the bits below is the permutation of
// four stable LSD passes.  If any bucket exceeds the capacity (skewed or few-distinct keys, keys below 2^21) the last two
// launches return and the four passes recorded behind them run as if nothing had happened.

// Exclusive scan of one value per thread over all THREADS threads; one barrier inside; scratch: THREADS / 64 words that
// nothing else touches until the next barrier.
template <int THREADS>
__device__ __forceinline__ uint32_t BlockExclusiveScanAll(uint32_t v, uint32_t* scratch, int tid) {
  constexpr int WAVES = THREADS / 64;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t x = WaveInclusiveScan(v);
  if (lane == 63) scratch[wave] = x;
  LdsBarrier();
  uint32_t add = 0;
#pragma unroll
  for (int w = 0; w < WAVES - 1; ++w)
    if (w < wave) add += scratch[w];
  return x - v + add;
}

// A loop that is unrolled by construction: body(integral_constant<int, I>) for I = FIRST, FIRST + STEP, ... < LAST, until
// it returns false.  (#pragma unroll gave up on the nine chunks of a 36-key lane -- multiple exits -- and a loop that is
// not unrolled indexes its register arrays through scratch memory.)
template <int FIRST, int LAST, int STEP, typename F>
__device__ __forceinline__ void StaticFor(F&& body) {
  if constexpr (FIRST < LAST) {
    if (!body(std::integral_constant<int, FIRST>{})) return;
    StaticFor<FIRST + STEP, LAST, STEP>(body);
  }
}

// Stable rank of a key slot inside its wave with PACKED counters: digit d counts in bits [16 (d & 1), +16) of word
// d >> 1 of the wave's row.  One returning LDS atomic per key, like RankAtomic: lanes of one instruction on one WORD are
// served in ascending lane order whatever they add (lds_order_check_packed_kernel verifies exactly this shape), and a
// wave's <= 2304 keys cannot carry out of a half.  Uniform slots (sorted or constant input) are ranked by one lane adding
// 64; the test runs only in chunks whose first slot looks the part (see RankAtomic).  Ranks come two to a register.
// The digit is the `width` bits of the key from bit `shift` up (both wave-uniform, width >= 1): its counter word and its
// half of the word are bit fields of the key themselves -- one v_bfe_u32 each, with the run-time window of round 6 as with
// constants (as `(key >> shift) & mask` a run-time mask cost the bucket kernel 11 us of 108 at 2^25).
__device__ __forceinline__ uint32_t DigitWord(uint32_t key, uint32_t shift, uint32_t width) {
  return __builtin_amdgcn_ubfe(key, shift + 1u, width - 1u);  // digit >> 1
}
__device__ __forceinline__ uint32_t DigitHalf(uint32_t key, uint32_t shift) {
  return __builtin_amdgcn_ubfe(key, shift, 1u) * 16u;  // (digit & 1) * 16
}
template <int KPT, bool DYN>
__device__ __forceinline__ void RankPacked16(const uint32_t (&key)[KPT], uint32_t shift, uint32_t width, uint32_t* myRow,
                                             int lane, uint32_t (&out)[KPT / 2], uint32_t slots = KPT) {
  constexpr int CHUNK = (!DYN && KPT % 8 == 0) ? 8 : 4;
  static_assert(KPT % CHUNK == 0 && CHUNK % 2 == 0, "whole chunks of pairs");
  StaticFor<0, KPT, CHUNK>([&](auto chunk) {
    constexpr int base = decltype(chunk)::value;
    if (DYN && (uint32_t)base >= slots) return false;
    uint32_t r[CHUNK];
    const uint32_t probe = __builtin_amdgcn_ubfe(key[base], shift, width);
    const bool watch = __popcll(__ballot(probe != (uint32_t)__builtin_amdgcn_readfirstlane(probe))) <= 48;  // wave-uniform
    if (watch) {
#pragma unroll
      for (int c = 0; c < CHUNK; ++c) {
        const uint32_t d = __builtin_amdgcn_ubfe(key[base + c], shift, width);
        const uint32_t sh = (d & 1u) * 16u;
        const bool uniform = __ballot(d != (uint32_t)__builtin_amdgcn_readfirstlane(d)) == 0ull;  // wave-uniform
        uint32_t old = 0;
        if (!uniform || lane == 0)
          old = __hip_atomic_fetch_add(&myRow[d >> 1], (uniform ? 64u : 1u) << sh, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        if (uniform) old = (uint32_t)__builtin_amdgcn_readfirstlane(old);
        r[c] = ((old >> sh) & 0xFFFFu) + (uniform ? (uint32_t)lane : 0u);
      }
    } else {
#pragma unroll
      for (int c = 0; c < CHUNK; ++c) {
        const uint32_t sh = DigitHalf(key[base + c], shift);
        const uint32_t old = __hip_atomic_fetch_add(&myRow[DigitWord(key[base + c], shift, width)], 1u << sh, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_WORKGROUP);
        r[c] = (old >> sh) & 0xFFFFu;
      }
    }
#pragma unroll
    for (int c = 1; c < CHUNK; c += 2) {
      out[(base + c) / 2] = r[c - 1] | (r[c] << 16);
      asm volatile("" : "+v"(out[(base + c) / 2]));  // pack now, not when first used
    }
    return true;
  });
}

// The scan over the packed counters [WAVES][ROW]: thread `col` owns word `col` (two digits) of every wave's row.
// ColumnTotals: both digits' counts over all waves (lo | hi << 16; <= 36864 each, no carry).  ColumnBases: replaces every
// wave's count by `add` + the counts of the waves before it -- the position of the wave's first key of that digit.
template <uint32_t ROW, int WAVES>
__device__ __forceinline__ uint32_t ColumnTotals(const uint32_t* counters, uint32_t col) {
  uint32_t total = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) total += counters[w * ROW + col];
  return total;
}
template <uint32_t ROW, int WAVES>
__device__ __forceinline__ void ColumnBases(uint32_t* counters, uint32_t col, uint32_t add) {
  static_assert(WAVES % 8 == 0, "halves of eight: eight reads in flight, then eight stores");
  uint32_t running = add;
#pragma unroll
  for (int w0 = 0; w0 < WAVES; w0 += 8) {
    uint32_t c[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) c[w] = counters[(w0 + w) * ROW + col];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      counters[(w0 + w) * ROW + col] = running;
      running += c[w];
    }
  }
}

// Both in one sweep where the registers allow it (keys-only kernels): the column is read ONCE into `column`, its total
// returned; ColumnBasesFrom then writes the bases from the registers -- sixteen LDS reads per thread and pass less.
template <uint32_t ROW, int WAVES>
__device__ __forceinline__ uint32_t ColumnRead(const uint32_t* counters, uint32_t col, uint32_t (&column)[WAVES]) {
  uint32_t total = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    column[w] = counters[w * ROW + col];
    total += column[w];
  }
  return total;
}
template <uint32_t ROW, int WAVES>
__device__ __forceinline__ void ColumnBasesFrom(uint32_t* counters, uint32_t col, uint32_t add, const uint32_t (&column)[WAVES]) {
  uint32_t running = add;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    counters[w * ROW + col] = running;
    running += column[w];
  }
}

// The same with TWO words (four digits) per thread, for workgroups of half as many threads as a row has words: thread
// `col` owns words 2 col and 2 col + 1 of every wave's row and reads them as one 8-byte quantity (a stride of two words
// in 4-byte reads would conflict two ways).  ColumnPairRead returns the totals of the two words; ColumnPairBasesFrom
// writes the bases, `add` holding the four digits' starting positions (lo | hi << 16 per word).
template <uint32_t ROW, int WAVES>
__device__ __forceinline__ u32x2 ColumnPairRead(const uint32_t* counters, uint32_t col, u32x2 (&column)[WAVES]) {
  u32x2 total = {0u, 0u};
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    column[w] = reinterpret_cast<const u32x2*>(counters + w * ROW)[col];
    total += column[w];
  }
  return total;
}
template <uint32_t ROW, int WAVES>
__device__ __forceinline__ void ColumnPairBasesFrom(uint32_t* counters, uint32_t col, u32x2 add, const u32x2 (&column)[WAVES]) {
  u32x2 running = add;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    reinterpret_cast<u32x2*>(counters + w * ROW)[col] = running;
    running += column[w];
  }
}

// ranks -> physical staging slots, in place: slot = StagingSlot(row[digit] + rank).  Reads only; the staging buffer may
// alias the counters once every wave has been through here (the caller's barrier).
template <int KPT, uint32_t STAGE, bool DYN, bool RAW = false>  // RAW: the position itself, not its swizzled staging slot
__device__ __forceinline__ void PositionsPacked16(const uint32_t (&key)[KPT], uint32_t shift, uint32_t width,
                                                  const uint32_t* myRow, uint32_t (&rankThenSlot)[KPT / 2],
                                                  uint32_t slots = KPT) {
  constexpr int CHUNK = (!DYN && KPT % 8 == 0) ? 8 : 4;
  StaticFor<0, KPT, CHUNK>([&](auto chunk) {
    constexpr int base = decltype(chunk)::value;
    if (DYN && (uint32_t)base >= slots) return false;
    uint32_t w[CHUNK];
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) w[c] = myRow[DigitWord(key[base + c], shift, width)];
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) {
      const int i = base + c;
      const uint32_t r = (rankThenSlot[i / 2] >> (16 * (i % 2))) & 0xFFFFu;
      const uint32_t position = ((w[c] >> DigitHalf(key[i], shift)) & 0xFFFFu) + r;
      w[c] = RAW ? position : StagingSlot<STAGE>(position);
    }
#pragma unroll
    for (int c = 1; c < CHUNK; c += 2) {
      rankThenSlot[(base + c) / 2] = w[c - 1] | (w[c] << 16);
      asm volatile("" : "+v"(rankThenSlot[(base + c) / 2]));
    }
    return true;
  });
}

// ---- the window of the MSD plan ------------------------------------------------------------------------
// The scatter digit used to be the keys' top BITS bits, whatever the keys: 24-bit keys, dense ascending or descending ids,
// anything whose top bits are constant then fell into a handful of buckets, the plan was turned down and the four passes ran
// (every pattern of BASELINE config 4 in round 5).  Since round 6 histogram_msd_kernel first looks at 64 keys, evenly spread
// over the input with the first and the last one among them (wave 0 of EVERY workgroup takes the same sample and reaches the
// same conclusion; workgroup 0 publishes it in the overflow word): the bits in which they all agree are the (guessed) common
// prefix, the window is the BITS bits right below it, and the bucket kernel sorts whatever is left below the window.  Uniform
// 32-bit keys have no common prefix and get the window they always had.  The sample is only the guess: every key is checked
// against the prefix and raises the overflow word if it breaks it, exactly like the spine does for a bucket that is too large.
// (For a monotone input the first and the last key are the extremes, so the guess is exact; for keys drawn at random a bit
// that varies in one key in ten is missed once in a thousand sorts -- and such keys do not spread over the window's buckets
// anyway.)
//
// The same sample predicts two cases the plan cannot take, so that they do not pay for the attempt (MsdMode): all sampled
// keys identical -> only the four byte tables are counted while every key is compared with the sampled one, and if all ARE
// identical nothing needs sorting (verdict 4); a bucket that is certainly too large -- fewer varying bits than the window over
// more elements than the buckets hold, or kMsdSampleSkew of the 64 sampled keys in one bucket -> the overflow word is raised
// at once and the four byte tables are all that is counted.  In both the spine kernel returns at once.
__device__ __forceinline__ uint32_t WaveOr(uint32_t v) {
  int x = (int)v;
  x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane(x, 63);               // lane 63 holds the OR over the wave
}

// ---- histogram_msd_kernel ---------------------------------------------------------------------------
// histogram_kernel (same pipeline: two groups of four 16-byte loads per lane in flight, all workgroups inside one window of
// the input) with a fifth table: the keys' WINDOW bits ((key >> shift) & (2^BITS - 1), the window chosen above),
// counted PER TILE of up to 32768 keys in 2^BITS bins x TC replicas of 32 KiB in all; after each tile the 2^BITS counts go
// out as 16-bit numbers (a tile holds at most 32768 keys: 0x8000 fits), are added to the workgroup's bucket sizes and
// cleared.  A wave whose 64 keys share their window bits would serialise 16-way on TC replicas: one lane adds 64 instead.
// Three forms of the loop, chosen by the window (uniform for the whole launch):
//   TOP       the window is the top BITS bits (uniform 32-bit keys: round 5's kernel): byte 3's table is not counted, it
//             follows from the bucket sizes;
//   PREFIXED  the window lies below a common prefix: byte 3 has a table of its own (the fallback needs all four if a key
//             breaks the prefix), and every key is XOR-ed against the reference key -- a difference above the window raises
//             the overflow word;
//   TABLES    the sample has turned the plan down, or every sampled key is identical: the four byte tables only, like
//             histogram_kernel; in the second case any key that differs from the reference raises the overflow word.
constexpr uint32_t kMsdTopBinWords = 8192;  // 32 KiB: 1024 bins x 8 replicas | 2048 x 4

constexpr uint32_t HistMsdByte3Copies(uint32_t copies) { return copies < 16u ? copies : 16u; }
constexpr uint32_t HistMsdLdsBytes(uint32_t copies, uint32_t bits) {
  return (3u * 256u * copies + kMsdTopBinWords + (1u << bits) + 256u * HistMsdByte3Copies(copies) + 4u) * 4u;
}

template <uint32_t COPIES, uint32_t BITS>
__global__ __launch_bounds__(kHistThreads) void histogram_msd_kernel(MsdArgs a) {
  // a tile is `rows` rows of 1024 sixteen-byte vectors = rows x 4096 keys (1 ... 8: the scatter's tiles of 4 ... 32 slots
  // of 64 keys per wave, MsdTileKeysFor in vrdx_layout.h)
  static_assert(kHistThreads == 1024, "a row of vectors per load instruction");
  constexpr uint32_t D = 1u << BITS;
  // The window bins are DOUBLE-BUFFERED (round 6): a tile counts into one half of the 32 KiB while the previous tile's half is
  // summed, written out and cleared by the first D / 2 threads -- one barrier per tile instead of two, and the other waves do
  // not wait for the flush.  (With one buffer the kernel stood still four times per workgroup at 2^25.)
  constexpr uint32_t kTopHalf = kMsdTopBinWords / 2;  // words per buffer
  constexpr uint32_t TC = kTopHalf / D;               // replicas of a window bin: 4 | 2
  constexpr uint32_t kByteWords = 3u * VRDX_RADIX * COPIES;
  constexpr uint32_t C3 = HistMsdByte3Copies(COPIES); // replicas of byte 3's own table (PREFIXED)
  constexpr uint32_t PER_BYTE = D / 256u;             // buckets per value of byte 3 when the window is at the top: 4 | 8
  static_assert(kMsdTopBinWords >= VRDX_RADIX * COPIES, "TABLES: byte 3's table takes the window bins' place");
  enum : uint32_t { TOP = 0, PREFIXED = 1, TABLES = 2 };
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* const bins = smem;                        // [3][256][COPIES]
  uint32_t* const top = smem + kByteWords;            // [2][D][TC]; TABLES: byte 3's [256][COPIES]
  uint32_t* window = top;                             // the buffer the current tile counts in
  uint32_t* const bucketSize = top + kMsdTopBinWords; // [D]: this workgroup's share of every bucket, word pairs owned by one thread
  uint32_t* const byte3 = bucketSize + D;             // PREFIXED: [256][C3]
  uint32_t* const verdict = byte3 + VRDX_RADIX * C3;  // [2]: wave 0's conclusion from the sample, the reference key
  const uint32_t* const keys = a.keysCaller;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  if (blockIdx.x == 0 && tid < 3) a.tickets[tid] = 0;
  const uint32_t n = ElementCount(a.maxCount, a.countPtr);
  const uint32_t rows = a.tileKeys / 4096u;
  const uint32_t tiles = a.tiles;
  // wave 0's sample (above): in flight in front of the first tile's loads
  static_assert(kMsdSampleKeys == 64, "one sampled key per lane of wave 0");
  uint32_t sampled = 0;
  if (tid < kMsdSampleKeys && n != 0u)
    sampled = keys[(uint32_t)(((uint64_t)tid * (uint64_t)(n - 1u)) / (kMsdSampleKeys - 1u))];  // keys 0 ... n - 1
  uint32_t ref = 0;  // key 0

  const uint32_t copy = tid & (COPIES - 1);
  uint32_t shift = 0, mode = 0, form = TOP;  // (from `decided`, in sweep)
  uint32_t differs = 0;  // OR of key ^ ref over this thread's keys (PREFIXED, TABLES)
  uint32_t likeRef = 0;  // PREFIXED: this thread's keys whose byte 3 is the reference key's (nearly all: it lies in the prefix)
  // One lane adds the whole wave's share where all its active lanes agree on the bin in ALL their NK keys (sorted, constant,
  // narrow inputs: 64 lanes on TC replicas of one bin would serialise 8- or 16-way); one test per NK keys.
  auto addShared = [&](auto nk, uint32_t* table, const uint32_t (&bin)[decltype(nk)::value], uint32_t replicas) {
    constexpr uint32_t NK = decltype(nk)::value;
    const uint64_t active = __ballot(true);  // (taken HERE: inside the one-lane branch below it would be that one lane)
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane(bin[0]);
    uint32_t other = 0;
#pragma unroll
    for (uint32_t i = 0; i < NK; ++i) other |= bin[i] ^ first;
    if (__ballot(other != 0u) != 0ull) {
#pragma unroll
      for (uint32_t i = 0; i < NK; ++i) atomicAdd(&table[bin[i] * replicas + (tid & (replicas - 1))], 1u);
    } else if (lane == (uint32_t)__builtin_ctzll(active)) {
      atomicAdd(&table[first * replicas], NK * (uint32_t)__popcll(active));
    }
  };
  auto count = [&](auto form, auto nk, const uint32_t (&key)[decltype(nk)::value]) {
    constexpr uint32_t FORM = decltype(form)::value;
    constexpr uint32_t NK = decltype(nk)::value;
#pragma unroll
    for (uint32_t i = 0; i < NK; ++i) {
#pragma unroll
      for (uint32_t p = 0; p < 3; ++p) {
        const uint32_t d = (key[i] >> (8 * p)) & 0xFFu;
        atomicAdd(&bins[(p * VRDX_RADIX + d) * COPIES + copy], 1u);
      }
      if constexpr (FORM == TABLES) atomicAdd(&top[(key[i] >> 24) * COPIES + copy], 1u);
      if constexpr (FORM != TOP) differs |= key[i] ^ ref;
    }
    if constexpr (FORM == PREFIXED) {
      // byte 3 (the fallback needs its table if a key breaks the prefix): counted in a register where it is the reference
      // key's, which it is for every key when the window ends at or below bit 24
      uint32_t unlike = 0;
#pragma unroll
      for (uint32_t i = 0; i < NK; ++i) {
        const bool like = ((key[i] ^ ref) >> 24) == 0u;
        likeRef += like ? 1u : 0u;
        unlike |= like ? 0u : 1u << i;
      }
      if (unlike != 0u) {
#pragma unroll
        for (uint32_t i = 0; i < NK; ++i)
          if ((unlike >> i) & 1u) {
            const uint32_t one[1] = {key[i] >> 24};
            addShared(std::integral_constant<uint32_t, 1>{}, byte3, one, C3);
          }
      }
    }
    if constexpr (FORM != TABLES) {
      uint32_t t[NK];
#pragma unroll
      for (uint32_t i = 0; i < NK; ++i) t[i] = FORM == TOP ? key[i] >> (32u - BITS) : (key[i] >> shift) & (D - 1u);
      addShared(nk, window, t, TC);
    }
  };
  constexpr std::integral_constant<uint32_t, 1> kOne{};
  constexpr std::integral_constant<uint32_t, 4> kFour{};
  // rows [first, first + 4) of tile `tile`: loads (index clamped, like HistFetch) and counts (masked by row and by nvec)
  const uint32_t tileVecs = rows * kHistThreads;
  auto fetch = [&](auto streaming, uint32_t tile, uint32_t first, uint32_t nvec, u32x4 (&k)[4]) {
    constexpr bool NT = decltype(streaming)::value;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      uint32_t i = tile * tileVecs + (first + u) * kHistThreads + tid;
      i = i < nvec ? i : (nvec != 0 ? nvec - 1 : 0u);
      k[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(keys) + i) : reinterpret_cast<const u32x4*>(keys)[i];
    }
  };
  auto tally = [&](auto form, uint32_t tile, uint32_t first, const u32x4 (&k)[4], uint32_t nvec) {
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      if (first + u < rows && tile * tileVecs + (first + u) * kHistThreads + tid < nvec) {
        const uint32_t four[4] = {k[u][0], k[u][1], k[u][2], k[u][3]};
        count(form, kFour, four);
      }
    }
  };
  // the tile's counts out, added to the workgroup's bucket sizes, its buffer cleared -- by the first D / 2 threads, behind ONE
  // barrier (every wave has counted the tile); the buffer is counted in again two tiles on, i.e. behind the next tile's barrier
  auto flush = [&](uint32_t tile, uint32_t* counted) {
    LdsBarrier();
    for (uint32_t w = tid; w < D / 2; w += kHistThreads) {  // whole waves: D / 2 is a multiple of 64
      uint32_t* const mine = counted + (size_t)w * 2u * TC;  // bins 2 w and 2 w + 1, TC replicas each
      uint32_t lo = 0, hi = 0;
      if constexpr (TC == 4) {
        const u32x4 x = reinterpret_cast<const u32x4*>(mine)[0], y = reinterpret_cast<const u32x4*>(mine)[1];
        lo = x[0] + x[1] + x[2] + x[3];
        hi = y[0] + y[1] + y[2] + y[3];
        reinterpret_cast<u32x4*>(mine)[0] = u32x4{0u, 0u, 0u, 0u};
        reinterpret_cast<u32x4*>(mine)[1] = u32x4{0u, 0u, 0u, 0u};
      } else {
        static_assert(TC == 2 || TC == 4, "two bins of a thread are one or two 16-byte vectors");
        const u32x4 x = reinterpret_cast<const u32x4*>(mine)[0];
        lo = x[0] + x[1];
        hi = x[2] + x[3];
        reinterpret_cast<u32x4*>(mine)[0] = u32x4{0u, 0u, 0u, 0u};
      }
      a.tileCounts[(size_t)tile * (D / 2) + w] = lo | (hi << 16);
      reinterpret_cast<u32x2*>(bucketSize)[w] += u32x2{lo, hi};
    }
  };

  const uint32_t nvec = n >> 2;
  auto sweep = [&](auto streaming) {
    const uint32_t step = gridDim.x;
    u32x4 a4[4], b4[4];
    uint32_t tile = blockIdx.x;
    const bool two = rows > 4;  // (uniform) tiles of more than four rows keep a second set of loads in flight
    // the first loads (and wave 0's sample in front of them) fly while the counters are cleared
    fetch(streaming, tile, 0, nvec, a4);
    if (two)
      fetch(streaming, tile, 4, nvec, b4);
    else
      fetch(streaming, tile + step, 0, nvec, b4);  // (tiles of at most four rows: the second set is the workgroup's NEXT tile)
    static_assert((kByteWords + kMsdTopBinWords + D) % 4u == 0, "cleared sixteen bytes at a time");
    for (uint32_t i = tid; i < (kByteWords + kMsdTopBinWords + D) / 4u; i += kHistThreads)
      reinterpret_cast<u32x4*>(bins)[i] = u32x4{0u, 0u, 0u, 0u};
    LdsBarrier();
    if (tid < 64u) {  // wave 0: what the sample says
      const uint32_t differ = WaveOr(sampled) & WaveOr(~sampled);  // bits that are 1 in some sampled key and 0 in another
      const uint32_t varying = differ != 0u ? 32u - (uint32_t)__clz((int)differ) : 0u;  // they all lie below this bit
      // (never lower than bit 2: each of the bucket kernel's two passes then has a bit to rank by, and keys of twelve bits
      // and fewer, for which that costs buckets, are rare at these sizes and quick in the four passes, two of which are trivial)
      const uint32_t lowest = (varying > BITS + 2u ? varying : BITS + 2u) - BITS;  // the window's lowest bit
      const uint32_t spread = varying > lowest ? varying - lowest : 0u;  // bits of the window that vary: <= BITS
      uint32_t kind = kMsdModePlan;
      if (n != 0u && varying == 0u) {
        kind = kMsdModeIdentical;
      } else if (spread < BITS && (uint64_t)n > ((uint64_t)a.cap << spread)) {
        kind = kMsdModeDeclined;  // at most 2^spread buckets can hold anything
      } else {
        uint32_t* const bin = &top[__builtin_amdgcn_ubfe(sampled, lowest, BITS) * TC];
        const uint32_t seen = atomicAdd(bin, 1u) + 1u;
        // (the estimate as well: a short indirect count samples every key many times)
        if (__ballot(seen >= kMsdSampleSkew && (uint64_t)seen * n / kMsdSampleKeys > a.cap) != 0ull) kind = kMsdModeDeclined;
        *bin = 0u;  // (behind every lane's atomic: LDS operations of one wave are carried out in order)
      }
      if (tid == 0) {
        verdict[0] = (lowest << kMsdShiftShift) | (kind << kMsdModeShift) | (kind == kMsdModeDeclined ? kMsdDeclineSample : 0u);
        verdict[1] = sampled;
      }
    }
    LdsBarrier();
    const uint32_t decided = verdict[0];
    ref = verdict[1];
    // for the launches behind this one (the fill in front has zeroed the word; keys outside the prefix are OR-ed into it too)
    if (blockIdx.x == 0 && tid == 0) atomicOr(a.overflowWord, decided);
    shift = (decided >> kMsdShiftShift) & kMsdShiftMask;
    mode = (decided >> kMsdModeShift) & kMsdModeMask;
    form = mode != kMsdModePlan ? (uint32_t)TABLES : (shift == 32u - BITS ? (uint32_t)TOP : (uint32_t)PREFIXED);
    auto loop = [&](auto formTag) {
      constexpr uint32_t FORM = decltype(formTag)::value;
      if constexpr (FORM == PREFIXED) {  // (byte 3's own table: only this form counts in it)
        for (uint32_t i = tid; i < VRDX_RADIX * C3; i += kHistThreads) byte3[i] = 0;
        LdsBarrier();
      }
      // the end of a tile: the last one to three keys belong to the tile that holds key n - 1; its counts out
      auto finish = [&](uint32_t done) {
        if (done == nvec / tileVecs && tid < (n & 3u)) {
          const uint32_t one[1] = {keys[(nvec << 2) + tid]};
          count(formTag, kOne, one);
        }
        if constexpr (FORM != TABLES) {
          flush(done, window);
          window = window == top ? top + kTopHalf : top;
        }
      };
      if (two) {  // tiles of five to eight rows: both sets of loads belong to one tile
        for (; tile < tiles; tile += step) {
          tally(formTag, tile, 0, a4, nvec);  // (counts into `window`)
          fetch(streaming, tile + step, 0, nvec, a4);
          tally(formTag, tile, 4, b4, nvec);
          fetch(streaming, tile + step, 4, nvec, b4);
          finish(tile);
        }
      } else {    // tiles of at most four rows: a set of loads per tile, two tiles in flight
        for (; tile < tiles; tile += 2 * step) {
          tally(formTag, tile, 0, a4, nvec);
          fetch(streaming, tile + 2 * step, 0, nvec, a4);
          finish(tile);
          tally(formTag, tile + step, 0, b4, nvec);  // (beyond the last tile: every lane masked)
          fetch(streaming, tile + 3 * step, 0, nvec, b4);
          if (tile + step < tiles) finish(tile + step);
        }
      }
    };
    if (form == TOP)
      loop(std::integral_constant<uint32_t, TOP>{});
    else if (form == PREFIXED)
      loop(std::integral_constant<uint32_t, PREFIXED>{});
    else
      loop(std::integral_constant<uint32_t, TABLES>{});
  };
  const bool streamingInput = VRDX_HIST_NT == 2 || (VRDX_HIST_NT == 1 && n > kHistStreamingLoadsAbove);
  if (streamingInput) {
    asm volatile("; non-temporal key loads" ::: "memory");
    sweep(std::true_type{});
  } else {
    sweep(std::false_type{});
  }
  if (form == PREFIXED && likeRef != 0u) atomicAdd(&byte3[(ref >> 24) * C3 + (tid & (C3 - 1))], likeRef);
  __syncthreads();

  if (tid < 3 * VRDX_RADIX) {
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t c = 0; c < COPIES; ++c) sum += bins[tid * COPIES + ((c + tid) & (COPIES - 1))];
    if (sum != 0) atomicAdd(&a.histogramTable[tid], sum);
  } else {
    const uint32_t v = tid - 3 * VRDX_RADIX;  // a value of byte 3
    uint32_t sum = 0;
    if (form == TOP) {
#pragma unroll
      for (uint32_t q = 0; q < PER_BYTE; ++q) sum += bucketSize[v * PER_BYTE + q];
    } else if (form == PREFIXED) {
#pragma unroll
      for (uint32_t c = 0; c < C3; ++c) sum += byte3[v * C3 + ((c + tid) & (C3 - 1))];
    } else {
#pragma unroll
      for (uint32_t c = 0; c < COPIES; ++c) sum += top[v * COPIES + ((c + tid) & (COPIES - 1))];
    }
    if (sum != 0) atomicAdd(&a.histogramTable[tid], sum);
  }
  // the bucket sizes, for the spine's bucket bases (under the top window they follow from byte 3's table, like in round 5)
  if (form == PREFIXED) {
    for (uint32_t d = tid; d < D; d += kHistThreads) {
      const uint32_t mine = bucketSize[d];
      if (mine != 0) atomicAdd(&a.bucketCount[d], mine);
    }
  }
  // a key outside the prefix (PREFIXED), a key that is not the reference key (all sampled keys identical)
  const uint32_t above = shift + BITS;  // < 32 in the PREFIXED form
  const bool broken = form == PREFIXED ? (differs >> (above & 31u)) != 0u : (mode == kMsdModeIdentical && differs != 0u);
  if (__ballot(broken) != 0ull && lane == 0) atomicOr(a.overflowWord, kMsdDeclinePrefix);
}

// ---- spine_msd_kernel -------------------------------------------------------------------------------
// tileCounts[tile][w] (two 16-bit counts per word) -> exclusive prefixes over the tiles, in place; bucketBase.
// One workgroup per 16 words (32 buckets): thread (chunk, word) adds up its chunk of the rows -- 64 chunks, at most 32 rows
// each, every load in flight at once, 64 bytes per row and workgroup -- the 64 chunk sums of a word are scanned by one
// wave, and the thread walks its rows again from registers.  Totals are kept in 32 bits per digit: a bucket beyond 65535
// must not go unnoticed because its half wrapped (the prefixes it leaves are garbage then, and nobody reads them).
// The first bucket's base is the number of keys in the buckets below it: the histogram kernel has added up every bucket's
// size (bucketCount).  A plan that is already turned down (the sample's prediction, a key outside the sampled prefix) or
// has nothing to scatter (all keys identical) leaves only the fallback's status region to clear.
template <uint32_t BITS>
__global__ __launch_bounds__(1024) void spine_msd_kernel(MsdArgs a) {
  constexpr uint32_t D = 1u << BITS, ROW = D / 2;
  constexpr uint32_t WORDS = 16, CHUNKS = 64, MAXROWS = kMsdMaxTiles / CHUNKS;
  __shared__ uint32_t sumLo[WORDS][CHUNKS], sumHi[WORDS][CHUNKS], total[2 * WORDS], below[16];
  const uint32_t tid = threadIdx.x;
  const uint32_t j = tid & (WORDS - 1), c = tid / WORDS;
  const uint32_t w = blockIdx.x * WORDS + j;
  const uint32_t rows = (a.tiles + CHUNKS - 1) / CHUNKS;  // per chunk, <= MAXROWS (the host sees to it)
  const uint32_t r0 = c * rows;
  uint32_t* const column = a.tileCounts + w;
  const uint32_t decided = *a.overflowWord;

  // status region 0 of the passes recorded behind the plan (nothing reads it before they start)
  {
    u32x4* const clear = reinterpret_cast<u32x4*>(a.statusClear);
    for (uint32_t i = blockIdx.x * 1024u + tid; i < a.statusVecs; i += gridDim.x * 1024u) clear[i] = u32x4{0u, 0u, 0u, 0u};
  }
  // keys in the buckets below this workgroup's first one: under the top window a prefix of byte 3's table, else of the bucket
  // sizes the histogram kernel has added up
  const uint32_t firstBucket = blockIdx.x * 2u * WORDS;
  // (both asked for before `decided` has arrived: three loads per thread, not one more round trip per workgroup)
  uint32_t underTop = tid < 256u && tid < (firstBucket >> (BITS - 8u)) ? a.histogramTable[3u * VRDX_RADIX + tid] : 0u;
  uint32_t underSizes = 0;
#pragma unroll
  for (uint32_t d = tid; d < D; d += 1024u)
    if (d < firstBucket) underSizes += a.bucketCount[d];

  // (and so is the column: a plan that is turned down reads 2-4 MiB for nothing, 2 us; one that runs does not wait for the
  // verdict first)
  uint32_t v[MAXROWS];
#pragma unroll
  for (uint32_t k = 0; k < MAXROWS; ++k) v[k] = (k < rows && r0 + k < a.tiles) ? column[(size_t)(r0 + k) * ROW] : 0u;
  if ((decided & kMsdDeclineMask) != 0u || ((decided >> kMsdModeShift) & kMsdModeMask) != kMsdModePlan) return;  // nothing to scan
  uint32_t under = ((decided >> kMsdShiftShift) & kMsdShiftMask) == 32u - BITS ? underTop : underSizes;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (uint32_t k = 0; k < MAXROWS; ++k) {
    lo += v[k] & 0xFFFFu;
    hi += v[k] >> 16;
  }
  sumLo[j][c] = lo;
  sumHi[j][c] = hi;
  under = WaveInclusiveScan(under);
  if ((tid & 63u) == 63u) below[tid >> 6] = under;
  __syncthreads();
  {
    const uint32_t word = tid >> 6, chunk = tid & 63u;  // one wave per word
    const uint32_t x = sumLo[word][chunk], y = sumHi[word][chunk];
    const uint32_t ix = WaveInclusiveScan(x), iy = WaveInclusiveScan(y);
    sumLo[word][chunk] = ix - x;
    sumHi[word][chunk] = iy - y;
    if (chunk == 63u) {
      total[2 * word] = ix;
      total[2 * word + 1] = iy;
    }
  }
  __syncthreads();
  lo = sumLo[j][c];
  hi = sumHi[j][c];
#pragma unroll
  for (uint32_t k = 0; k < MAXROWS; ++k) {
    if (k < rows && r0 + k < a.tiles) column[(size_t)(r0 + k) * ROW] = (lo & 0xFFFFu) | (hi << 16);
    lo += v[k] & 0xFFFFu;
    hi += v[k] >> 16;
  }
  if (tid < 64u) {
    const uint32_t mine = tid < 2 * WORDS ? total[tid] : 0u;
    const uint32_t inclusive = WaveInclusiveScan(mine);
    if (tid < 2 * WORDS) {
      uint32_t base = inclusive - mine;
#pragma unroll
      for (uint32_t q = 0; q < 16; ++q) base += below[q];
      a.bucketBase[firstBucket + tid] = base;
      a.bucketCount[firstBucket + tid] = mine;  // (what the histogram kernel has added up below a prefix, the same number)
      if (mine > a.cap) atomicOr(a.overflowWord, kMsdDeclineBucket);
    }
  }
}

// ---- scatter_msd_kernel -----------------------------------------------------------------------------
// One tile of 32768 keys: load (wave-striped), rank by the top BITS bits with packed counters, scan, positions, regroup
// through the staging buffer -- which takes the counters' place once every wave knows its positions: 128 KiB of staging
// and 64 KiB of counters would not fit side by side -- and out in quads like the pass kernels, boundary quads by the thread
// that owns the run (two runs per thread here).  The tile's 2^BITS bases arrive with the keys: one row of prefixes and the
// bucket bases, loaded first.
//
// XCD-AWARE tile order.  A tile's run of bucket d is followed in memory by the NEXT tile's run of bucket d, and with runs of
// 128 bytes (32768 keys over 1024 buckets) nearly every 128-byte line is written by two tiles.  The eight XCDs have an L2
// each: written from two of them, a line leaves both as a partial line.  Measured at 2^25 keys (tools/r05/ablate.sh,
// profiles/r05_msd_scatter.txt): no stores 48.6 us, contiguous stores 64.0, the real runs 98.6-106.9 with tiles handed out
// round-robin -- and 91.3 with this order: XCD x (the workgroups b with b % 8 == x: observed dispatch order, for speed only,
// nothing depends on it) takes the CONSECUTIVE tiles [x C, (x + 1) C), C = ceil(tiles / 8), so that the lines of a
// bucket's range are completed inside one L2.  The grid is 8 C workgroups; one without a tile returns at once.
// NON-TEMPORAL loads of the keys and values (read once): they then do not push the lines the scatter is still completing
// out of that L2, nor the scattered data out of the caches before the bucket kernel reads them: 79.4 instead of 91.3 us,
// and the bucket kernel behind it 112 instead of 116.
// (A persistent form -- one workgroup per CU striding through the tiles, the next tile's keys prefetched into the registers
// the staged keys had left, every store a counted buffer store so that the prefetch could be waited for without draining the
// scatter -- was built and measured: 91.3 against 86.0 us for one tile per workgroup.  Tiles of 16384 keys, two workgroups
// per CU: 97.6 against 93.3.  Neither kept.)
template <uint32_t BITS>
constexpr size_t ScatterMsdLdsWords() {
  return (size_t)kMsdTileKeys + (1u << BITS) + 32;
}

template <uint32_t BITS, bool KV>
__device__ __forceinline__ void ScatterMsdBody(const MsdArgs a) {
  constexpr int THREADS = 1024, KPT = 32, WAVES = THREADS / 64;
  constexpr uint32_t TILE = kMsdTileKeys, D = 1u << BITS, ROW = D / 2u;
  static_assert(THREADS * KPT == TILE && ROW <= (uint32_t)THREADS && WAVES * ROW <= TILE, "geometry");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* const sorted = smem;               // TILE: keys (then values) regrouped by digit
  uint32_t* const counters = smem;             // WAVES x ROW packed counters; dead before the first key is staged
  uint32_t* const tileOffset = smem + TILE;    // D: global base - tile-local base
  uint32_t* const scanScratch = tileOffset + D;  // 32

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t n = ElementCount(a.maxCount, a.countPtr);
  // the grid is 8 C workgroups, C = ceil(tiles / 8): workgroup b takes tile (b % 8) C + b / 8 of its XCD's chunk
  const uint32_t perXcd = gridDim.x / 8u;
  const uint32_t tile = VRDX_MSD_XCD ? (blockIdx.x % 8u) * perXcd + blockIdx.x / 8u : blockIdx.x;
  // EVEN-SPLIT tiles (MsdTilePlan in vrdx_api.cpp): every wave takes `slots` of its KPT slots of 64 keys (a multiple of four),
  // a tile is slots x 1024 keys, chosen so that the tiles fill whole rounds of one workgroup per CU -- 520 tiles of 32768 keys
  // cost three rounds, the third for eight tiles; 768 tiles of 24576 cost three rounds of three quarters the length.
  constexpr bool DYN = VRDX_MSD_SCATTER_DYN != 0;  // (0: tiles of full capacity only, with VRDX_MSD_EVEN=0 -- measurements)
  const uint32_t slots = DYN ? a.tileKeys / (uint32_t)THREADS : (uint32_t)KPT;
  const uint32_t frame = DYN ? a.tileKeys : TILE;
  const uint32_t tileStart = tile * frame;
  // Verdict first, loads second: a wave cannot end with loads in flight, so a launch that is turned down (a bucket beyond the
  // capacity: the four passes behind it run) would still read every key -- 22 us at 2^25 for nothing, measured.
  const uint32_t decided = *a.overflowWord;
  if ((decided & kMsdDeclineMask) != 0u) {
    if (blockIdx.x == 0 && tid == 0 && a.declinedPlans != nullptr) atomicAdd(a.declinedPlans, 1u);  // (vrdxHipReadPlanCounters)
    return;
  }
  // the window the histogram kernel chose (uniform 32-bit keys: the top BITS bits); all keys identical: nothing to scatter, and
  // the launches behind return on the verdict
  const uint32_t SHIFT = (decided >> kMsdShiftShift) & kMsdShiftMask;
  if (((decided >> kMsdModeShift) & kMsdModeMask) == kMsdModeIdentical) {
    if (blockIdx.x == 0 && tid == 0) *a.planWord = kMsdVerdictSorted;
    return;
  }
  const uint32_t valid = tile < a.tiles && tileStart < n ? ((n - tileStart) < frame ? (n - tileStart) : frame) : 0u;
  const uint32_t tileEnd = tileStart + valid;
  const uint32_t loadBase = tileStart + wave * (slots * 64) + lane;
  uint32_t key[KPT];
  LoadStriped<KPT, VRDX_MSD_NT_LOADS != 0, DYN>(a.keysCaller, loadBase, tileEnd, valid == frame, 0xFFFFFFFFu, key, slots);  // pad: downsweep.slang:81
  uint32_t prefixWord = 0, base0 = 0, base1 = 0;
  if ((uint32_t)tid < ROW && valid != 0) {
    prefixWord = a.tileCounts[(size_t)tile * ROW + tid];
    base0 = a.bucketBase[2 * tid];
    base1 = a.bucketBase[2 * tid + 1];
  }
  if (blockIdx.x == 0 && tid == 0) *a.planWord = kMsdVerdictRuns | (SHIFT << kMsdShiftShift);  // for the launches behind this one
  if (valid == 0) return;

  uint32_t* const myRow = counters + wave * ROW;
#pragma unroll
  for (uint32_t i = 0; i < ROW / 256u; ++i) reinterpret_cast<u32x4*>(myRow)[lane + 64 * i] = u32x4{0u, 0u, 0u, 0u};
  uint32_t rank[KPT / 2];  // ranks, then physical staging slots, two to a register
  RankPacked16<KPT, DYN>(key, SHIFT, BITS, myRow, lane, rank, slots);
  ForgetDerivedValues<KPT>(key);
  LdsBarrier();

  // tile histogram, tile-local bases, every wave's first position per digit
  uint32_t column[WAVES];
  const uint32_t totals = (uint32_t)tid < ROW ? ColumnRead<ROW, WAVES>(counters, tid, column) : 0u;
  const uint32_t count0 = totals & 0xFFFFu, count1 = totals >> 16;
  const uint32_t local0 = BlockExclusiveScanAll<THREADS>(count0 + count1, scanScratch, tid);
  const uint32_t local1 = local0 + count0;
  if ((uint32_t)tid < ROW) {
    ColumnBasesFrom<ROW, WAVES>(counters, tid, local0 | (local1 << 16), column);
    tileOffset[2 * tid] = base0 + (prefixWord & 0xFFFFu) - local0;
    tileOffset[2 * tid + 1] = base1 + (prefixWord >> 16) - local1;
  }
  LdsBarrier();
  PositionsPacked16<KPT, TILE, DYN>(key, SHIFT, BITS, myRow, rank, slots);
  LdsBarrier();  // the counters are dead: the staging buffer takes their place
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
    sorted[(rank[i / 2] >> (16 * (i % 2))) & 0xFFFFu] = key[i];
  }
  LdsBarrier();
  // key+value: the values are fetched now and land while the keys are scattered (like the pass kernels' early value fetch)
  uint32_t val[KV ? KPT : 1];
  if constexpr (KV)
    LoadStriped<KPT, VRDX_MSD_NT_LOADS != 0, DYN>(a.valuesCaller, loadBase, tileEnd, valid == frame, 0u, val, slots);  // pad: downsweep.slang:85

  // scatter: whole single-digit quads in the main loop, the quads around a run's start by the thread that owns the run
  constexpr int QUADS = KPT / 4;
  constexpr int B = 4;
  uint32_t quadDigits[KV ? QUADS : 1];  // key+value: first | last << 16 digit of every main-loop quad, for the value phase
  uint32_t boundaryQuad[2] = {~0u, ~0u};
  uint32_t boundaryDigits[2][2] = {{0, 0}, {0, 0}};
  if ((uint32_t)tid < ROW) {
    if (count0 != 0 && (local0 & 3u) != 0 && local0 < valid) boundaryQuad[0] = local0 & ~3u;
    if (count1 != 0 && (local1 & 3u) != 0 && local1 < valid) boundaryQuad[1] = local1 & ~3u;
  }
  if (tid == 0 && (valid & 3u) != 0) boundaryQuad[0] = valid & ~3u;  // (digit 0 starts at 0: thread 0's first slot is free)
  auto scatterQuads = [&](uint32_t* out, bool keysPhase) {
#pragma unroll
    for (int j0 = 0; j0 < QUADS; j0 += B) {
      if (4u * (uint32_t)j0 * THREADS >= valid) break;
      u32x4 w4[B];
      uint32_t o[B];
      bool whole[B];
#pragma unroll
      for (int b = 0; b < B; ++b) w4[b] = *reinterpret_cast<const u32x4*>(&sorted[4u * (tid + (j0 + b) * THREADS)]);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int j = j0 + b;
        const uint32_t p = StagingSlot<TILE>(4u * (tid + j * THREADS));  // involution: the sorted position of the quad
        uint32_t d0, d3;
        if (keysPhase) {
          d0 = __builtin_amdgcn_ubfe(w4[b][0], SHIFT, BITS);
          d3 = __builtin_amdgcn_ubfe(w4[b][3], SHIFT, BITS);
          if constexpr (KV) quadDigits[j] = d0 | (d3 << 16);
        } else {
          d0 = quadDigits[KV ? j : 0] & 0xFFFFu;
          d3 = quadDigits[KV ? j : 0] >> 16;
        }
        whole[b] = p + 3 < valid && d0 == d3;
        o[b] = tileOffset[d0] + p;
        asm volatile("" : "+v"(o[b]));  // fetched here, for every quad: not sunk into the conditional store
      }
#pragma unroll
      for (int b = 0; b < B; ++b)
        if (whole[b]) StoreQuad(out, o[b], w4[b]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (boundaryQuad[s] == ~0u) continue;
      const u32x4 q = *reinterpret_cast<const u32x4*>(&sorted[StagingSlot<TILE>(boundaryQuad[s])]);
      if (keysPhase) {
        boundaryDigits[s][0] = __builtin_amdgcn_ubfe(q[0], SHIFT, BITS) | (__builtin_amdgcn_ubfe(q[1], SHIFT, BITS) << 16);
        boundaryDigits[s][1] = __builtin_amdgcn_ubfe(q[2], SHIFT, BITS) | (__builtin_amdgcn_ubfe(q[3], SHIFT, BITS) << 16);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t d = (boundaryDigits[s][c / 2] >> (16 * (c % 2))) & 0xFFFFu;
        if (boundaryQuad[s] + c < valid) StoreWord(out, tileOffset[d] + boundaryQuad[s] + c, q[c]);
      }
    }
  };
  scatterQuads(a.keysScratch, true);
  if constexpr (KV) {
    LdsBarrier();  // every key has left the staging buffer
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
      sorted[(rank[i / 2] >> (16 * (i % 2))) & 0xFFFFu] = val[i];
    }
    LdsBarrier();
    scatterQuads(a.valuesScratch, false);
  }
}

// ---- the keys-only scatter (round 6): two tiles per workgroup -----------------------------------------------
// TWO consecutive tiles A, B per workgroup, keys-only (key+value sorts keep one tile per workgroup, above: two tiles' keys
// AND values do not fit the register file).  What the scatter pays for is its write pattern: a tile of 32768 keys
// over 1024 buckets writes runs of 128 bytes at 4-byte alignment, 1.45 x the bytes (profiles/r05_pmc_traffic.json).  In memory a
// tile's run of bucket d is followed by the NEXT tile's run of bucket d (prefix[B][d] = prefix[A][d] + count[A][d]), so a
// workgroup that holds both tiles writes ONE run of twice the length: both tiles' keys in registers (the two-sub-tile pass
// kernel's trick), ranked with a counter row per (sub-tile, wave) -- the column scan walks A's sixteen rows, then B's: inside a
// bucket all of A precedes all of B, the stable order -- and staged through the 128 KiB buffer in two halves BY POSITION
// ([0, H) then [H, 2H), H = a tile's keys: whatever the keys are, a half fits), each half written out in quads like
// scatter_msd_kernel does.  A run that straddles H is cut in two, like any run at a tile's end.
// Measured against the one-tile form (profiles/r06_scatter_pair.txt, rocprofv3, ten sorts back to back): 70.4 instead of 83.0 us
// at 2^25, 36.8 instead of 43.1 at 2^24; WRITE_SIZE 157.9 MB per launch instead of 193.2 (1.18 x instead of 1.44 x of what must
// be written), reads unchanged.
template <uint32_t BITS>
__device__ __forceinline__ void ScatterMsdPairBody(const MsdArgs a) {
  constexpr int THREADS = 1024, KPT = 32, WAVES = THREADS / 64;
  constexpr uint32_t TILE = kMsdTileKeys, D = 1u << BITS, ROW = D / 2u;
  static_assert(2 * WAVES * ROW <= TILE, "both sub-tiles' counter rows inside the staging buffer");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* const sorted = smem;               // TILE: half of the super-tile's keys at a time, regrouped by digit
  uint32_t* const counters = smem;             // 2 x WAVES x ROW packed counters; dead before the first key is staged
  uint32_t* const tileOffset = smem + TILE;    // D: global base - local base of the super-tile
  uint32_t* const scanScratch = tileOffset + D;  // 32

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t n = ElementCount(a.maxCount, a.countPtr);
  const uint32_t pairs = (a.tiles + 1u) / 2u;
  // XCD x takes the consecutive super-tiles [x C, (x + 1) C), C = ceil(pairs / 8) (scatter_msd_kernel)
  const uint32_t perXcd = (pairs + 7u) / 8u;
  const uint32_t pair = (blockIdx.x % 8u) * perXcd + blockIdx.x / 8u;
  const uint32_t slots = a.tileKeys / (uint32_t)THREADS;
  const uint32_t frame = a.tileKeys;          // H
  const uint32_t decided = *a.overflowWord;
  if ((decided & kMsdDeclineMask) != 0u) {
    if (blockIdx.x == 0 && tid == 0 && a.declinedPlans != nullptr) atomicAdd(a.declinedPlans, 1u);
    return;
  }
  const uint32_t SHIFT = (decided >> kMsdShiftShift) & kMsdShiftMask;
  if (((decided >> kMsdModeShift) & kMsdModeMask) == kMsdModeIdentical) {
    if (blockIdx.x == 0 && tid == 0) *a.planWord = kMsdVerdictSorted;
    return;
  }
  const uint32_t tileA = 2u * pair;
  const uint32_t startA = tileA * frame, startB = startA + frame;
  const bool exists = blockIdx.x < 8u * perXcd && pair < pairs && startA < n;
  const uint32_t left = exists ? n - startA : 0u;
  const uint32_t validA = left < frame ? left : frame;
  const uint32_t validB = left > frame ? (left - frame < frame ? left - frame : frame) : 0u;
  const uint32_t valid = validA + validB;
  const uint32_t baseA = startA + wave * (slots * 64) + lane, baseB = baseA + frame;
  uint32_t keyA[KPT], keyB[KPT];
  LoadStriped<KPT, VRDX_MSD_NT_LOADS != 0, true>(a.keysCaller, baseA, startA + validA, validA == frame, 0xFFFFFFFFu, keyA, slots);
  LoadStriped<KPT, VRDX_MSD_NT_LOADS != 0, true>(a.keysCaller, baseB, startB + validB, validB == frame, 0xFFFFFFFFu, keyB, slots);
  uint32_t prefixWord = 0, base0 = 0, base1 = 0;
  if ((uint32_t)tid < ROW && valid != 0) {
    prefixWord = a.tileCounts[(size_t)tileA * ROW + tid];
    base0 = a.bucketBase[2 * tid];
    base1 = a.bucketBase[2 * tid + 1];
  }
  if (blockIdx.x == 0 && tid == 0) *a.planWord = kMsdVerdictRuns | (SHIFT << kMsdShiftShift);
  if (valid == 0) return;

  uint32_t* const rowA = counters + wave * ROW;
  uint32_t* const rowB = counters + (WAVES + wave) * ROW;
#pragma unroll
  for (uint32_t i = 0; i < ROW / 256u; ++i) {
    reinterpret_cast<u32x4*>(rowA)[lane + 64 * i] = u32x4{0u, 0u, 0u, 0u};
    reinterpret_cast<u32x4*>(rowB)[lane + 64 * i] = u32x4{0u, 0u, 0u, 0u};
  }
  uint32_t posA[KPT / 2], posB[KPT / 2];  // ranks, then positions inside the super-tile (< 65536), two to a register
  RankPacked16<KPT, true>(keyA, SHIFT, BITS, rowA, lane, posA, slots);
  ForgetDerivedValues<KPT>(keyA);
  RankPacked16<KPT, true>(keyB, SHIFT, BITS, rowB, lane, posB, slots);
  ForgetDerivedValues<KPT>(keyB);
  LdsBarrier();

  // the super-tile's histogram over all 32 rows (A's waves, then B's), its local bases, every row's first position per digit
  const uint32_t totals = (uint32_t)tid < ROW ? ColumnTotals<ROW, 2 * WAVES>(counters, tid) : 0u;
  const uint32_t count0 = totals & 0xFFFFu, count1 = totals >> 16;
  const uint32_t local0 = BlockExclusiveScanAll<THREADS>(count0 + count1, scanScratch, tid);
  const uint32_t local1 = local0 + count0;
  if ((uint32_t)tid < ROW) {
    ColumnBases<ROW, 2 * WAVES>(counters, tid, local0 | (local1 << 16));
    tileOffset[2 * tid] = base0 + (prefixWord & 0xFFFFu) - local0;
    tileOffset[2 * tid + 1] = base1 + (prefixWord >> 16) - local1;
  }
  LdsBarrier();
  PositionsPacked16<KPT, TILE, true, true>(keyA, SHIFT, BITS, rowA, posA, slots);
  PositionsPacked16<KPT, TILE, true, true>(keyB, SHIFT, BITS, rowB, posB, slots);
  LdsBarrier();  // the counters are dead: the staging buffer takes their place

  constexpr int QUADS = KPT / 4;
  constexpr int B = 2;  // (four at a time: 128 registers and 48 bytes of scratch)
  // the quads around a run's start, by the thread that owns the run (two runs per thread, scatter_msd_kernel)
  uint32_t boundaryQuad[2] = {~0u, ~0u};
  if ((uint32_t)tid < ROW) {
    if (count0 != 0 && (local0 & 3u) != 0 && local0 < valid) boundaryQuad[0] = local0 & ~3u;
    if (count1 != 0 && (local1 & 3u) != 0 && local1 < valid) boundaryQuad[1] = local1 & ~3u;
  }
  if (tid == 0 && (valid & 3u) != 0) boundaryQuad[0] = valid & ~3u;  // (digit 0 starts at 0: thread 0's first slot is free)
#pragma unroll 1
  for (uint32_t half = 0; half < 2; ++half) {
    const uint32_t from = half * frame;  // positions [from, from + frame) go through the buffer now
    if (from >= valid) break;
    if (half != 0) LdsBarrier();         // every quad of the first half has been read
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (i % 4 == 0 && (uint32_t)i >= slots) break;
      const uint32_t pa = ((posA[i / 2] >> (16 * (i % 2))) & 0xFFFFu) - from;
      const uint32_t pb = ((posB[i / 2] >> (16 * (i % 2))) & 0xFFFFu) - from;
      if (pa < frame) sorted[StagingSlot<TILE>(pa)] = keyA[i];
      if (pb < frame) sorted[StagingSlot<TILE>(pb)] = keyB[i];
    }
    LdsBarrier();
    const uint32_t here = valid - from < frame ? valid - from : frame;  // keys in the buffer
#pragma unroll
    for (int j0 = 0; j0 < QUADS; j0 += B) {
      if (4u * (uint32_t)j0 * THREADS >= here) break;
      u32x4 w4[B];
      uint32_t o[B];
      bool whole[B];
#pragma unroll
      for (int b = 0; b < B; ++b) w4[b] = *reinterpret_cast<const u32x4*>(&sorted[4u * (tid + (j0 + b) * THREADS)]);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const uint32_t p = StagingSlot<TILE>(4u * (tid + (j0 + b) * THREADS));  // involution: the quad's position in this half
        const uint32_t d0 = __builtin_amdgcn_ubfe(w4[b][0], SHIFT, BITS), d3 = __builtin_amdgcn_ubfe(w4[b][3], SHIFT, BITS);
        whole[b] = p + 3 < here && d0 == d3;
        o[b] = tileOffset[d0] + from + p;
        asm volatile("" : "+v"(o[b]));
      }
#pragma unroll
      for (int b = 0; b < B; ++b)
        if (whole[b]) StoreQuad(a.keysScratch, o[b], w4[b]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t q0 = boundaryQuad[s] - from;  // (~0u - from: beyond `frame`)
      if (boundaryQuad[s] == ~0u || q0 >= frame) continue;
      const u32x4 q = *reinterpret_cast<const u32x4*>(&sorted[StagingSlot<TILE>(q0)]);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (q0 + c < here) StoreWord(a.keysScratch, tileOffset[__builtin_amdgcn_ubfe(q[c], SHIFT, BITS)] + boundaryQuad[s] + c, q[c]);
    }
  }
}
// The plan's scatter by mode: two tiles per workgroup for keys-only sorts by ten bits, one tile per workgroup otherwise
// (key+value; eleven bits: with 1024-word counter rows the two-tile body needs 116 bytes of scratch per lane).
template <uint32_t BITS, bool KV>
constexpr bool MsdScatterTakesPairs() { return !KV && BITS == 10; }
template <uint32_t BITS, bool KV>
__device__ __forceinline__ void ScatterMsdRole(const MsdArgs a) {
  if constexpr (MsdScatterTakesPairs<BITS, KV>())
    ScatterMsdPairBody<BITS>(a);
  else
    ScatterMsdBody<BITS, KV>(a);
}

// ---- bucket_sort2_kernel ----------------------------------------------------------------------------
// Workgroup b sorts bucket b = [bucketBase[b], + bucketCount[b]) of the scratch arrays by the 32 - BITS bits below the
// scatter's, in two stable passes (11 bits, then the rest) with 2048 packed counters per wave, and writes it to the same
// range of the caller's arrays.  Keys (and values) stay in registers between the passes; the staging buffer -- up to
// 144 KiB -- takes the counters' place inside each pass like in scatter_msd_kernel; key+value stages the values through
// the same slots after the keys.  A wave takes only as many slots as the bucket needs (like SortInWorkgroup).
template <int KPT, int THREADS = 1024>
constexpr size_t BucketSort2LdsWords() {
  return (size_t)THREADS * KPT + 32;
}

template <uint32_t BITS, int KPT, bool KV, int THREADS = 1024>
__device__ __forceinline__ void BucketSort2Bucket(const MsdArgs a, const uint32_t bucket, const uint32_t below) {
  constexpr int WAVES = THREADS / 64;
  static_assert(THREADS == 1024 || THREADS == 512, "one or two words of a counter row per thread");
  constexpr uint32_t TILE = THREADS * KPT;
  constexpr uint32_t ROW = 1024;  // 2048 packed counters per wave
  static_assert(32u - BITS <= 22u, "two passes of at most eleven bits");
  static_assert(TILE <= 65536 && WAVES * ROW <= TILE && KPT % 4 == 0, "packed positions; the staging buffer covers the counters");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* const staged = smem;             // TILE
  uint32_t* const counters = smem;           // WAVES x ROW, inside each pass only
  uint32_t* const scanScratch = smem + TILE; // 32

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // The bits below the scatter's window, two to twenty-two (the window never lies lower than bit 2), in TWO passes of half
  // the bits each: 11 | 11 under the top window of uniform keys.  (11 | 4 for the fifteen bits of dense 25-bit ids measured
  // 146 us against 120: sixteen digits are eight counter words for 64 lanes, an eight-way conflict in every atomic of the
  // second pass; 8 | 7 has none.  And always two passes, also where one would do: with a run-time number of passes the same
  // loop ran 115 us instead of 108 for uniform keys.)
  const uint32_t W0 = (below + 1u) / 2u, W1 = below / 2u;
  constexpr uint32_t passes = 2u;
  const uint32_t myBase = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.bucketBase[bucket]);
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.bucketCount[bucket]);
  if (n == 0) return;
  const uint32_t* const keysIn = a.keysScratch + myBase;
  uint32_t* const keysOut = a.keysCaller + myBase;
  const uint32_t* const valuesIn = KV ? a.valuesScratch + myBase : nullptr;
  uint32_t* const valuesOut = KV ? a.valuesCaller + myBase : nullptr;

  constexpr bool DYN = true;
  // The bucket's ceil(n / 256) chunks of four 64-element slots are dealt out EVENLY over the sixteen waves: wave w takes
  // base + (w < extra) chunks, consecutive in memory (waves in order, (slot, lane) order inside a wave: the ranking stays
  // stable).  With `slots` the same for every wave (SortInWorkgroup) a bucket of 16385 elements keeps thirteen waves busy
  // with twenty slots each, and every phase lasts as long as twenty slots take; dealt out evenly one wave has twenty and
  // fifteen have sixteen (VRDX_MSD_EVEN_WAVES = 0: the old rule, measurements).
  uint32_t slots, first;
  if (VRDX_MSD_EVEN_WAVES) {
    const uint32_t chunks = (n + 255u) / 256u;  // <= 144: n <= 36864
    const uint32_t base = chunks / WAVES, extra = chunks % WAVES;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    slots = 4u * (base + (w < extra ? 1u : 0u));
    first = 256u * (w * base + (w < extra ? w : extra)) + lane;  // element i of this lane: first + 64 * i
  } else {
    slots = 4u * ((n + 4u * THREADS - 1u) / (4u * THREADS));
    slots = slots < (uint32_t)KPT ? slots : (uint32_t)KPT;
    first = wave * (slots * 64) + lane;
    // a wave whose range starts at or behind n holds nothing but pads and skips every loop over its slots (SortInWorkgroup)
    const uint32_t waveStart = (uint32_t)wave * (slots * 64);
    const uint32_t mine = n > waveStart ? n - waveStart : 0u;
    const uint32_t waveSlots = 4u * ((mine + 255u) / 256u);
    slots = waveSlots < slots ? waveSlots : slots;
  }
  uint32_t key[KPT];
  uint32_t val[KV ? KPT : 1];
  LoadStriped<KPT, VRDX_MSD_BUCKET_NT != 0, DYN>(keysIn, first, n, false, 0xFFFFFFFFu, key, slots);
  if constexpr (KV) LoadStriped<KPT, VRDX_MSD_BUCKET_NT != 0, DYN>(valuesIn, first, n, false, 0u, val, slots);

  uint32_t* const myRow = counters + wave * ROW;
#pragma unroll 1
  for (uint32_t pass = 0; pass < passes; ++pass) {
    const uint32_t shift = pass == 0 ? 0u : W0;
    const uint32_t width = pass == 0 ? W0 : W1;
    const uint32_t words = width > 1u ? 1u << (width - 1u) : 1u;  // of every wave's row that this pass counts in (two digits to a word)
#pragma unroll
    for (uint32_t i = 0; i < ROW / 256u; ++i)
      if (256u * i < words) reinterpret_cast<u32x4*>(myRow)[lane + 64 * i] = u32x4{0u, 0u, 0u, 0u};
    uint32_t rank[KPT / 2];
    RankPacked16<KPT, DYN>(key, shift, width, myRow, lane, rank, slots);
    ForgetDerivedValues<KPT>(key);
    LdsBarrier();
    uint32_t firstNow = first;  // (see SortInWorkgroup: keeps the read-back addresses out of registers across the passes)
    asm volatile("" : "+v"(firstNow));
    if constexpr (THREADS == 512) {  // two words of the row per thread
      // (threads whose words no digit of this pass counts in only take part in the scan's barrier)
      // (not in the key+value form: 121 of its 128 registers are taken, the test costs it seven and 12 bytes of scratch)
      const bool mine = KV || 2u * (uint32_t)tid < words;
      u32x2 column[WAVES];
      const u32x2 totals = mine ? ColumnPairRead<ROW, WAVES>(counters, tid, column) : u32x2{0u, 0u};
      const uint32_t count0 = totals[0] & 0xFFFFu, count1 = totals[0] >> 16, count2 = totals[1] & 0xFFFFu;
      const uint32_t local0 = BlockExclusiveScanAll<THREADS>(count0 + count1 + count2 + (totals[1] >> 16), scanScratch + 16 * pass, tid);
      const uint32_t local2 = local0 + count0 + count1;
      if (mine)
        ColumnPairBasesFrom<ROW, WAVES>(counters, tid, u32x2{local0 | ((local0 + count0) << 16), local2 | ((local2 + count2) << 16)}, column);
    } else if constexpr (!KV) {  // (key+value: the values are live as well; the column is read twice instead of kept)
      const bool mine = (uint32_t)tid < words;
      uint32_t column[WAVES];
      const uint32_t totals = mine ? ColumnRead<ROW, WAVES>(counters, tid, column) : 0u;
      const uint32_t count0 = totals & 0xFFFFu;
      const uint32_t local0 = BlockExclusiveScanAll<THREADS>(count0 + (totals >> 16), scanScratch + 16 * pass, tid);
      if (mine) ColumnBasesFrom<ROW, WAVES>(counters, tid, local0 | ((local0 + count0) << 16), column);
    } else {
      const bool mine = (uint32_t)tid < words;
      const uint32_t totals = mine ? ColumnTotals<ROW, WAVES>(counters, tid) : 0u;
      const uint32_t count0 = totals & 0xFFFFu;
      const uint32_t local0 = BlockExclusiveScanAll<THREADS>(count0 + (totals >> 16), scanScratch + 16 * pass, tid);
      if (mine) ColumnBases<ROW, WAVES>(counters, tid, local0 | ((local0 + count0) << 16));
    }
    LdsBarrier();
    PositionsPacked16<KPT, TILE, DYN>(key, shift, width, myRow, rank, slots);
    LdsBarrier();  // the counters are dead: the staging buffer takes their place
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
      staged[(rank[i / 2] >> (16 * (i % 2))) & 0xFFFFu] = key[i];
    }
    LdsBarrier();
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
      key[i] = staged[StagingSlot<TILE>(firstNow + 64 * i)];
    }
    if constexpr (KV) {
      LdsBarrier();
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
        staged[(rank[i / 2] >> (16 * (i % 2))) & 0xFFFFu] = val[i];
      }
      LdsBarrier();
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
        val[i] = staged[StagingSlot<TILE>(firstNow + 64 * i)];
      }
    }
    LdsBarrier();  // the next pass clears its counters where the staging buffer is
  }
  // Key+value sorts of more than 2^24 pairs write their result with NON-TEMPORAL stores: 256 MiB and more of output is the
  // size of the Infinity Cache, and left dirty in the caches it is written back under the next kernel's reads -- the next
  // sort's histogram in a batch of sorts.  Measured (tools/r05/out_nt2.sh, bench.py's loop of 20 sorts): key+value 82 ->
  // 86.3 GItems/s at 2^25, a single sort on its own unchanged (0.4225 / 0.4239 ms).  Keys-only sorts keep plain stores:
  // their loop gains the same 4 % but a single sort LOSES 4 % (its last kernel then waits for its own write-back), and a
  // consumer of 128 MiB of sorted keys finds a good part of them in the cache.  (The empty asm statements keep the two
  // kinds of store apart, see LoadTile.)
  const bool streamOut = VRDX_MSD_OUT_NT == 2 || (VRDX_MSD_OUT_NT == 1 && KV && a.maxCount > kStreamingLoadsAbove);
  if (streamOut) {
    asm volatile("; non-temporal output" ::: "memory");
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
      const uint32_t index = first + 64 * i;
      if (index < n) {
        __builtin_nontemporal_store(key[i], &keysOut[index]);
        if constexpr (KV) __builtin_nontemporal_store(val[i], &valuesOut[index]);
      }
    }
    asm volatile("" ::: "memory");
  } else {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (DYN && i % 4 == 0 && (uint32_t)i >= slots) break;
      const uint32_t index = first + 64 * i;
      if (index < n) {
        keysOut[index] = key[i];
        if constexpr (KV) valuesOut[index] = val[i];
      }
    }
  }
}

// The bucket launch: workgroup b sorts bucket b, b + workgroups, ... -- one bucket per workgroup when there are 2^BITS of them
// (the half-size kernel), two when half that: the keys-only launch that is also pass 1 of the fallback then carries no
// workgroups the pass has no tile for (1024 workgroups of 144 KiB of LDS for 512 tiles cost a turned-down sort 3-6 us).
// (Every pass of a bucket ends with a barrier: the next bucket may clear its counters where the staging buffer was.)
template <uint32_t BITS, int KPT, bool KV, int THREADS = 1024>
__device__ __forceinline__ void BucketSort2Body(const MsdArgs a, const uint32_t workgroups) {
  const uint32_t verdict = *a.planWord;
  if ((verdict & kMsdVerdictMask) != kMsdVerdictRuns) return;  // the plan does not apply (the four passes are running), or nothing needs sorting
  const uint32_t below = (verdict >> kMsdShiftShift) & kMsdShiftMask;
#pragma unroll 1
  for (uint32_t bucket = blockIdx.x; bucket < (1u << BITS); bucket += workgroups) BucketSort2Bucket<BITS, KPT, KV, THREADS>(a, bucket, below);
}

template <uint32_t BITS, bool KV>
__global__ __launch_bounds__(1024) void scatter_msd_kernel(MsdArgs a) {
  ScatterMsdRole<BITS, KV>(a);
}
template <uint32_t BITS, int KPT, bool KV>
__global__ __launch_bounds__(1024) void bucket_sort2_kernel(MsdArgs a) {
  BucketSort2Body<BITS, KPT, KV>(a, gridDim.x);
}
// Buckets of no more than 18432 elements (sorts of up to 18.1 M elements by ten bits): workgroups of 512 threads and
// 72 KiB of LDS, TWO to a CU -- one loads or scans while the other ranks.  A bucket has a fixed cost of 5.7 us in the
// kernel above (load latency, two column scans, eight barriers), half the time of a bucket of 8192 keys, and with one
// workgroup per CU nothing runs beside it.
// (__launch_bounds__' second argument is WAVES PER SIMD here: two workgroups of eight waves on four SIMDs = 4, i.e. at most
// 128 registers; the kernels take 85 keys-only and 121 key+value.)
template <uint32_t BITS, bool KV>
__global__ __launch_bounds__(512, 4) void bucket_sort2_half_kernel(MsdArgs a) {
  BucketSort2Body<BITS, kMsdHalfCap / 512, KV, 512>(a, gridDim.x);
}

// ---- the plan's launches double as the first two launches of its fallback ------------------------------
// Behind the MSD plan the four passes are recorded as the fallback for keys the device turns the plan down for; when the plan
// runs they return on the verdict word -- 4.4 us each at 2^25 (512-1024 workgroups of 144-156 KiB of LDS and one load), 18 us
// of a 400 us key+value sort.  Two of the four are saved by giving the plan's own launches a second ROLE: the scatter launch is
// pass 0 of the fallback when the plan is turned down, the bucket launch is pass 1 when the scatter has not run -- one branch
// on a word every workgroup reads anyway, the grid the larger of the two roles' (a workgroup beyond its role's range returns:
// both bodies check).  The pass is the one the recorder would have launched at these sizes: the two-sub-tile kernel keys-only,
// onesweep_kernel<1024, 32, kv> key+value, the one-atomic ranking (the plan is recorded with it only), DYN by the tile plan.
// Passes 2 and 3 stay launches of their own.  Keys-only since round 5 (0.2520 instead of 0.2610 ms at 2^25); the key+value
// form measured 16 us SLOWER in its scatter role then (a private segment of 36 bytes that no instruction touches, and with
// it a scratch set-up per wave) and waited for round 6, which found that segment to be a matter of the kernels' shape (below).
template <uint32_t BITS, bool KV, bool DYN>
__device__ __forceinline__ void FallbackPassBody(const OnesweepArgs p) {
  if constexpr (KV)
    OnesweepBody<1024, 32, true, true, DYN>(p);
  else
    OnesweepPairBody<1024, 32, DYN>(p);
}
template <bool KV>
constexpr size_t MsdFusedLdsWords(uint32_t bits, bool bucketLaunch) {
  const size_t pass = KV ? OnesweepLdsWords<1024, 32, true>() : PairLdsWords<1024, 32>();
  const size_t plan = bucketLaunch ? BucketSort2LdsWords<(KV ? kMsdCapKeyValue : kMsdCapKeys) / 1024>()
                                   : (size_t)kMsdTileKeys + ((size_t)1 << bits) + 32;
  return pass > plan ? pass : plan;
}

template <uint32_t BITS, bool KV, bool DYN>
__global__ __launch_bounds__(1024) void msd_scatter_or_pass0_kernel(MsdArgs m, OnesweepArgs p) {
  // (The SHAPE of these two kernels is not free: with the key+value pass body inside -- 106 SGPRs, 70 of them spilled to VGPR
  // lanes -- an if / else of the two roles, or the counter in front of the pass, leaves a 36-byte private segment that no
  // instruction touches and a scratch set-up per wave with it (round 5: 16 us in the scatter role); "the plan's role and
  // return, then the pass, then the counter" and, below, "the pass first" compile without.  tests/test_abi.py watches it.)
  if ((*m.overflowWord & kMsdDeclineMask) == 0u) {
    ScatterMsdRole<BITS, KV>(m);
    return;
  }
  // (the grid is the larger of the two roles': a workgroup beyond the pass's tiles has no ticket to take)
  if (blockIdx.x > p.statusRows) return;
  FallbackPassBody<BITS, KV, DYN>(p);
  if (blockIdx.x == 0 && threadIdx.x == 0 && m.declinedPlans != nullptr) atomicAdd(m.declinedPlans, 1u);  // (vrdxHipReadPlanCounters)
}
template <uint32_t BITS, bool KV, bool DYN>
__global__ __launch_bounds__(1024) void msd_buckets_or_pass1_kernel(MsdArgs m, OnesweepArgs p) {
  const uint32_t verdict = *m.planWord & kMsdVerdictMask;
  if (verdict < kMsdVerdictRuns) {  // the plan was turned down: pass 1 of the four
    if (blockIdx.x > p.statusRows) return;
    FallbackPassBody<BITS, KV, DYN>(p);
    return;
  }
  // (two buckets per workgroup whatever the pass's grid is: with key+value tiles of 32768 the pass has as many tiles as the plan
  // has buckets, and 1032 workgroups of one bucket each measured 182 us where 512 of two take 173)
  const uint32_t workgroups = min(gridDim.x, (1u << BITS) / 2u);
  if (verdict == kMsdVerdictRuns && blockIdx.x < workgroups)
    BucketSort2Body<BITS, (KV ? kMsdCapKeyValue : kMsdCapKeys) / 1024, KV>(m, workgroups);
}

// ---------------------------------------------------------------------------------------------
// device self-check for RankAtomic's precondition
// ---------------------------------------------------------------------------------------------
// Every wave ranks pseudo-random digit vectors of several entropies (constant, 2, 4, 16, 256
// distinct values, sparse collisions) both ways and counts disagreements.  16 waves per
// workgroup hammer the LDS at the same time, in the shapes the sort kernels use: 8 slots per lane
// unpacked (1024x8 tiles), then 32 slots per lane unpacked and packed (1024x32 tiles, keys-only and
// key+value / two-sub-tile form), 32 atomics in flight per lane.
__device__ __forceinline__ uint32_t OrderCheckKey(uint32_t tid, uint32_t wave, uint32_t slot) {
  uint32_t x = (blockIdx.x * 1024u + tid) * 0x9E3779B9u + slot * 0x85EBCA6Bu;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  const uint32_t mode = (blockIdx.x + wave + slot) % 6u;
  const uint32_t mask = mode == 0 ? 0u : mode == 1 ? 1u : mode == 2 ? 3u : mode == 3 ? 15u : mode == 4 ? 0x21u : 255u;
  return x & mask;
}

// sticky (the periodic re-check recorded behind every 65536th sort, vrdx_api.cpp): a mismatch sets bit 1 of the sorter's
// status word instead of being counted.
__global__ __launch_bounds__(1024) void lds_order_check_kernel(uint32_t* mismatches, uint32_t* sticky) {
  __shared__ uint32_t counters[2 * 16 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 16 * 256; i += 1024) counters[i] = 0;
  __syncthreads();
  uint32_t* const histA = counters + wave * 256;
  uint32_t* const histB = counters + (16 + wave) * 256;
  uint32_t bad = 0;
  for (uint32_t round = 0; round < 8; ++round) {
    uint32_t key[8], ra[8], rb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) key[i] = OrderCheckKey(tid, wave, round * 8 + i);
    RankAtomic<8, false>(key, 0, histA, lane, ra);
    RankBallot<8>(key, 0, histB, lane, rb);
#pragma unroll
    for (int i = 0; i < 8; ++i) bad += ra[i] != rb[i];
  }
  // the counters keep counting (both sets alike), so the ranks below stay under 2^16: 64 lanes x (64 + 64) slots.
  // The reference ranks come packed two to a register (register budget).
#pragma unroll 1
  for (uint32_t form = 0; form < 2; ++form) {
    uint32_t key[32], want[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) key[i] = OrderCheckKey(tid, wave, 64 + 32 * form + i);
    RankBallot<32, true>(key, 0, histB, lane, want);
    if (form == 0) {
      uint32_t ra[32];
      RankAtomic<32, false>(key, 0, histA, lane, ra);
#pragma unroll
      for (int i = 0; i < 16; ++i) bad += (ra[2 * i] | (ra[2 * i + 1] << 16)) != want[i];
    } else {
      uint32_t rp[16];
      RankAtomic<32, true>(key, 0, histA, lane, rp);
#pragma unroll
      for (int i = 0; i < 16; ++i) bad += rp[i] != want[i];
    }
  }
  if (bad != 0) {
    if (mismatches != nullptr) atomicAdd(mismatches, bad);
    if (sticky != nullptr) atomicOr(sticky, 2u);
  }
}

// The same for RankPacked16 (MSD plan): 2048 digits in packed 16-bit counters, two digits to a word, so lanes of one
// instruction meet on one WORD with DIFFERENT addends -- the shape the check above does not cover.  The reference ranks
// come from eleven ballots and a non-returning add by each group's first lane (sums do not depend on any order).
// Dynamic LDS: [16][1024] packed counters for the atomics | the same for the ballots = 128 KiB.
__device__ __forceinline__ uint32_t OrderCheckDigit11(uint32_t tid, uint32_t wave, uint32_t slot) {
  uint32_t x = (blockIdx.x * 1024u + tid) * 0x9E3779B9u + slot * 0x85EBCA6Bu + 0x7F4A7C15u;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  const uint32_t mode = (blockIdx.x + wave + slot) % 7u;
  // constant | the two halves of one word | two words | 16 values | sparse collisions | 64 neighbours | all eleven bits
  const uint32_t mask = mode == 0 ? 0u : mode == 1 ? 1u : mode == 2 ? 3u : mode == 3 ? 15u : mode == 4 ? 0x421u : mode == 5 ? 63u : 2047u;
  return ((x & mask) + (mode == 5 ? 0x300u : 0u)) & 2047u;
}

__global__ __launch_bounds__(1024) void lds_order_check_packed_kernel(uint32_t* mismatches, uint32_t* sticky) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* const rowA = smem + wave * 1024;
  uint32_t* const rowB = smem + (16 + wave) * 1024;
  for (int i = lane; i < 1024; i += 64) rowA[i] = rowB[i] = 0;  // wave-private rows: no barrier needed
  uint32_t bad = 0;
#pragma unroll 1
  for (uint32_t round = 0; round < 4; ++round) {
    uint32_t key[16], got[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) key[i] = OrderCheckDigit11(tid, wave, round * 16 + i);
    RankPacked16<16, false>(key, 0, 11u, rowA, lane, got);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t d = key[i];
      uint64_t same = ~0ull;
#pragma unroll
      for (int b = 0; b < 11; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t ballot = __ballot(bit);
        same &= bit ? ballot : ~ballot;
      }
      const uint32_t below = LanesBelow(same);
      const uint32_t sh = (d & 1u) * 16u;
      const uint32_t prior = (__hip_atomic_load(&rowB[d >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> sh) & 0xFFFFu;
      if (below == 0)
        (void)__hip_atomic_fetch_add(&rowB[d >> 1], (uint32_t)__popcll(same) << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      bad += ((got[i / 2] >> (16 * (i % 2))) & 0xFFFFu) != prior + below;
    }
  }
  if (bad != 0) {
    if (mismatches != nullptr) atomicAdd(mismatches, bad);
    if (sticky != nullptr) atomicOr(sticky, 2u);
  }
}

// ---------------------------------------------------------------------------------------------
// a kernel of KNOWN duration, for calibrating what a pair of HIP events adds to the kernel between them
// ---------------------------------------------------------------------------------------------
// One wave spins on the constant-rate wall clock (100 MHz on gfx950) for `ticks` ticks and leaves its first and last
// reading: out[1] - out[0] is the time the kernel demonstrably ran.  vrdxHipEventOverheadNs brackets it with two events.
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long* out, uint32_t ticks) {
  const unsigned long long t0 = wall_clock64();
  unsigned long long t = t0;
  while (t - t0 < ticks) {
    __builtin_amdgcn_s_sleep(1);
    t = wall_clock64();
  }
  if (threadIdx.x == 0) {
    out[0] = t0;
    out[1] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// Every launcher returns the hipError_t of ITS launch (hipLaunchKernel), so that the recorder never has to consult
// the calling thread's sticky last-error state, which an unrelated earlier failure may have set.
constexpr size_t kOrderCheckPackedLdsBytes = 2 * 16 * 1024 * sizeof(uint32_t);

template <typename... Args>
static hipError_t Launch(const void* kernel, uint32_t grid, uint32_t block, size_t ldsBytes, hipStream_t stream,
                         Args... args) {
  void* argv[] = {static_cast<void*>(&args)...};
  return hipLaunchKernel(kernel, dim3(grid), dim3(block), argv, ldsBytes, stream);
}

template <int THREADS, int KPT, bool KV, bool ATOMIC_RANK, bool DYN = false>
static const void* OnesweepKernel() {
  return reinterpret_cast<const void*>(&onesweep_kernel<THREADS, KPT, KV, ATOMIC_RANK, DYN>);
}
template <int THREADS, int KPT, bool DYN = false>
static const void* PairKernel() {
  return reinterpret_cast<const void*>(&onesweep_pair_kernel<THREADS, KPT, DYN>);
}

// EVEN: the forms with run-time slot counts (args.slots != 0: even-split and tail-split tiles, PlanTiles in
// vrdx_api.cpp) are built for this geometry as well.
template <int THREADS, int KPT, bool EVEN = false>
static hipError_t PrepareConfig() {
  const int keysBytes = (int)(OnesweepLdsWords<THREADS, KPT, false>() * sizeof(uint32_t));
  const int kvBytes = (int)(OnesweepLdsWords<THREADS, KPT, true>() * sizeof(uint32_t));
  const struct {
    const void* fn;
    int bytes;
  } kernels[8] = {
      {OnesweepKernel<THREADS, KPT, false, false>(), keysBytes},
      {OnesweepKernel<THREADS, KPT, false, true>(), keysBytes},
      {OnesweepKernel<THREADS, KPT, true, false>(), kvBytes},
      {OnesweepKernel<THREADS, KPT, true, true>(), kvBytes},
      {OnesweepKernel<THREADS, KPT, false, false, EVEN>(), keysBytes},
      {OnesweepKernel<THREADS, KPT, false, true, EVEN>(), keysBytes},
      {OnesweepKernel<THREADS, KPT, true, false, EVEN>(), kvBytes},
      {OnesweepKernel<THREADS, KPT, true, true, EVEN>(), kvBytes},
  };
  for (int i = 0; i < (EVEN ? 8 : 4); ++i) {
    const hipError_t e = hipFuncSetAttribute(kernels[i].fn, hipFuncAttributeMaxDynamicSharedMemorySize, kernels[i].bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <int THREADS, int KPT, bool EVEN = false>
static hipError_t LaunchConfig(hipStream_t stream, uint32_t grid, bool keyValue, bool atomicRank,
                               const OnesweepArgs& args) {
  const size_t lds = (keyValue ? OnesweepLdsWords<THREADS, KPT, true>() : OnesweepLdsWords<THREADS, KPT, false>()) *
                     sizeof(uint32_t);
  const void* kernel =
      keyValue ? (atomicRank ? OnesweepKernel<THREADS, KPT, true, true>() : OnesweepKernel<THREADS, KPT, true, false>())
               : (atomicRank ? OnesweepKernel<THREADS, KPT, false, true>() : OnesweepKernel<THREADS, KPT, false, false>());
  if (args.slots != 0) {  // run-time slot counts
    if (!EVEN || args.slots % 4 != 0 || args.slots > (uint32_t)KPT || args.tailSlots % 4 != 0 || args.tailSlots == 0 ||
        args.tailSlots > (uint32_t)KPT)
      return hipErrorInvalidValue;
    kernel = keyValue ? (atomicRank ? OnesweepKernel<THREADS, KPT, true, true, EVEN>()
                                    : OnesweepKernel<THREADS, KPT, true, false, EVEN>())
                      : (atomicRank ? OnesweepKernel<THREADS, KPT, false, true, EVEN>()
                                    : OnesweepKernel<THREADS, KPT, false, false, EVEN>());
  }
  return Launch(kernel, grid, THREADS, lds, stream, args);
}

// The two-sub-tile kernel exists for keys-only sorts with the one-atomic ranking (the ballot form of it spills
// 152 bytes per lane and ConfigIndex never selected it): its key+value form would hold sub-tile B's keys and
// ranks, A's staging slots and A's values at once and spills (measured 40 GItems/s in round 1; a 768-thread
// form with 168 registers and no spill measured 54.5 GItems/s against 67.2 for onesweep_kernel<1024, 32>,
// profiles/r03_geometry.txt), so it is not built.
template <int THREADS, int KPT>
static hipError_t PreparePairConfig() {
  const int bytes = (int)(PairLdsWords<THREADS, KPT>() * sizeof(uint32_t));
  const hipError_t e = hipFuncSetAttribute(PairKernel<THREADS, KPT, true>(), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(PairKernel<THREADS, KPT>(), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int THREADS, int KPT>
static hipError_t LaunchPairConfig(hipStream_t stream, uint32_t grid, bool keyValue, bool atomicRank,
                                   const OnesweepArgs& args) {
  if (keyValue || !atomicRank) return hipErrorInvalidValue;  // never selected (ConfigIndex)
  const size_t lds = PairLdsWords<THREADS, KPT>() * sizeof(uint32_t);
  if (args.slots != 0) {  // run-time slot counts
    if (args.slots % 4 != 0 || args.slots > (uint32_t)KPT || args.tailSlots % 4 != 0 || args.tailSlots == 0 ||
        args.tailSlots > (uint32_t)KPT)
      return hipErrorInvalidValue;
    return Launch(PairKernel<THREADS, KPT, true>(), grid, THREADS, lds, stream, args);
  }
  return Launch(PairKernel<THREADS, KPT>(), grid, THREADS, lds, stream, args);
}

// Every geometry here is selected by ConfigIndex (vrdx_api.cpp) for some size range; nothing else is built.
const TileConfig kTileConfigs[kNumTileConfigs] = {
    {1024, 8, 1}, {1024, 16, 1}, {1024, 32, 1}, {1024, 32, 2},
};

hipError_t PrepareKernels(int configIndex) {
  if (configIndex == 0) {  // once per sorter: the histogram kernels' dynamic LDS
    const struct {
      const void* fn;
      uint32_t bytes;
    } kernels[2] = {
        {reinterpret_cast<const void*>(&histogram_kernel<kHistCopies>), HistLdsBytes(kHistCopies)},
        {reinterpret_cast<const void*>(&histogram_kernel<kHistCopiesLarge>), HistLdsBytes(kHistCopiesLarge)},
    };
    for (const auto& k : kernels) {
      const hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.bytes);
      if (e != hipSuccess) return e;
    }
  }
  switch (configIndex) {
    case 0: return PrepareConfig<1024, 8>();
    case 1: return PrepareConfig<1024, 16>();
    case 2: return PrepareConfig<1024, 32, true>();
    case 3: return PreparePairConfig<1024, 32>();
    default: return hipErrorInvalidValue;
  }
}

hipError_t LdsOrderCheck(bool* laneOrdered) {
  uint32_t* d = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), sizeof(uint32_t));
  if (e != hipSuccess) return e;
  uint32_t h = 0xFFFFFFFFu;
  e = hipMemset(d, 0, sizeof(uint32_t));
  uint32_t* const noSticky = nullptr;
  if (e == hipSuccess) e = Launch(reinterpret_cast<const void*>(&lds_order_check_kernel), 512, 1024, 0, nullptr, d, noSticky);
  // the packed-counter shape of the MSD plan: two digits to a word (PrepareMsd has raised this kernel's LDS limit)
  if (e == hipSuccess)
    e = Launch(reinterpret_cast<const void*>(&lds_order_check_packed_kernel), 256, 1024, kOrderCheckPackedLdsBytes, nullptr, d, noSticky);
  if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(uint32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e == hipSuccess) *laneOrdered = h == 0;
  return e;
}

hipError_t LaunchLdsOrderRecheck(hipStream_t stream, uint32_t* sticky) {
  uint32_t* const noCount = nullptr;
  const hipError_t e = Launch(reinterpret_cast<const void*>(&lds_order_check_kernel), 8, 1024, 0, stream, noCount, sticky);
  if (e != hipSuccess) return e;
  return Launch(reinterpret_cast<const void*>(&lds_order_check_packed_kernel), 8, 1024, kOrderCheckPackedLdsBytes, stream, noCount, sticky);
}

hipError_t LaunchSpin(hipStream_t stream, unsigned long long* out, uint32_t ticks) {
  return Launch(reinterpret_cast<const void*>(&spin_kernel), 1, 64, 0, stream, out, ticks);
}

// ---- single-launch path for small sorts ---------------------------------------------------------
template <int THREADS, int KPT, bool KV, bool ATOMIC_RANK>
static const void* SmallKernel() {
  return reinterpret_cast<const void*>(&small_sort_kernel<THREADS, KPT, KV, ATOMIC_RANK>);
}

template <int THREADS, int KPT>
static hipError_t PrepareSmall() {
  const struct {
    const void* fn;
    size_t words;
  } kernels[4] = {
      {SmallKernel<THREADS, KPT, false, false>(), SmallSortLdsWords<THREADS, KPT, false>()},
      {SmallKernel<THREADS, KPT, false, true>(), SmallSortLdsWords<THREADS, KPT, false>()},
      {SmallKernel<THREADS, KPT, true, false>(), SmallSortLdsWords<THREADS, KPT, true>()},
      {SmallKernel<THREADS, KPT, true, true>(), SmallSortLdsWords<THREADS, KPT, true>()},
  };
  for (const auto& k : kernels) {
    const hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(k.words * 4));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <int THREADS, int KPT>
static hipError_t LaunchSmall(hipStream_t stream, bool atomicRank, uint32_t* keys, uint32_t* values, uint32_t maxCount,
                              const uint32_t* countPtr, uint32_t* failure) {
  const bool keyValue = values != nullptr;
  const size_t lds = (keyValue ? SmallSortLdsWords<THREADS, KPT, true>() : SmallSortLdsWords<THREADS, KPT, false>()) * 4;
  const void* const kernel =
      keyValue ? (atomicRank ? SmallKernel<THREADS, KPT, true, true>() : SmallKernel<THREADS, KPT, true, false>())
               : (atomicRank ? SmallKernel<THREADS, KPT, false, true>() : SmallKernel<THREADS, KPT, false, false>());
  return Launch(kernel, 1, THREADS, lds, stream, keys, values, maxCount, countPtr, failure);
}

hipError_t PrepareSmallSort() {
  hipError_t e = PrepareSmall<256, 16>();
  if (e == hipSuccess) e = PrepareSmall<1024, 16>();
  return e;
}

hipError_t LaunchSmallSort(hipStream_t stream, bool atomicRank, uint32_t* keys, uint32_t* values, uint32_t maxCount,
                           const uint32_t* countPtr, uint32_t* failure) {
  if (maxCount <= 256u * 16u) return LaunchSmall<256, 16>(stream, atomicRank, keys, values, maxCount, countPtr, failure);
  return LaunchSmall<1024, 16>(stream, atomicRank, keys, values, maxCount, countPtr, failure);
}

// ---- second half of the hybrid plan ---------------------------------------------------------------
template <int KPT, bool KV, bool ATOMIC_RANK>
static const void* BucketKernel() {
  return reinterpret_cast<const void*>(&bucket_sort_kernel<1024, KPT, KV, ATOMIC_RANK>);
}

template <int KPT>
static hipError_t PrepareBucket() {
  const struct {
    const void* fn;
    size_t words;
  } kernels[4] = {
      {BucketKernel<KPT, false, false>(), SmallSortLdsWords<1024, KPT, false>()},
      {BucketKernel<KPT, false, true>(), SmallSortLdsWords<1024, KPT, false>()},
      {BucketKernel<KPT, true, false>(), SmallSortLdsWords<1024, KPT, true>()},
      {BucketKernel<KPT, true, true>(), SmallSortLdsWords<1024, KPT, true>()},
  };
  for (const auto& k : kernels) {
    const hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(k.words * 4));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <int KPT>
static hipError_t LaunchBucket(hipStream_t stream, bool keyValue, bool atomicRank, const BucketSortArgs& args) {
  const size_t lds = (keyValue ? SmallSortLdsWords<1024, KPT, true>() : SmallSortLdsWords<1024, KPT, false>()) * 4;
  const void* const kernel = keyValue ? (atomicRank ? BucketKernel<KPT, true, true>() : BucketKernel<KPT, true, false>())
                                      : (atomicRank ? BucketKernel<KPT, false, true>() : BucketKernel<KPT, false, false>());
  return Launch(kernel, VRDX_RADIX, 1024, lds, stream, args);
}

hipError_t PrepareBucketSort() {
  hipError_t e = PrepareBucket<4>();
  if (e == hipSuccess) e = PrepareBucket<8>();
  if (e == hipSuccess) e = PrepareBucket<16>();
  // 32768-element buckets: with the one-atomic ranking only (the ballot forms would spill); key+value stages keys and
  // values through ONE buffer (SharedStage)
  if (e == hipSuccess)
    e = hipFuncSetAttribute(BucketKernel<32, false, true>(), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(SmallSortLdsWords<1024, 32, false>() * 4));
  if (e == hipSuccess)
    e = hipFuncSetAttribute(BucketKernel<32, true, true>(), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(SmallSortLdsWords<1024, 32, true>() * 4));
  return e;
}

hipError_t LaunchBucketSort(hipStream_t stream, bool keyValue, bool atomicRank, const BucketSortArgs& args) {
  switch (args.hybridCap) {
    case 1024u * 4u: return LaunchBucket<4>(stream, keyValue, atomicRank, args);
    case 1024u * 8u: return LaunchBucket<8>(stream, keyValue, atomicRank, args);
    case 1024u * 16u: return LaunchBucket<16>(stream, keyValue, atomicRank, args);
    case 1024u * 32u:
      if (!atomicRank) return hipErrorInvalidValue;  // never recorded (HybridCapacity)
      if (keyValue)
        return Launch(BucketKernel<32, true, true>(), VRDX_RADIX, 1024, SmallSortLdsWords<1024, 32, true>() * 4, stream, args);
      return Launch(BucketKernel<32, false, true>(), VRDX_RADIX, 1024, SmallSortLdsWords<1024, 32, false>() * 4, stream, args);
    default: return hipErrorInvalidValue;
  }
}

hipError_t LaunchHistogram(hipStream_t stream, uint32_t grid, const uint32_t* keys, uint32_t maxCount,
                           const uint32_t* countPtr, uint32_t* globalHistogram, uint32_t* tickets, void* statusClear,
                           uint32_t statusClearBytes) {
  u32x4* const clear = reinterpret_cast<u32x4*>(statusClear);
  const uint32_t vecs = statusClearBytes / 16u;  // whole status rows: a multiple of 1 KiB, 128-byte aligned
  const bool many = maxCount >= kHistManyCopiesFrom;
  const void* const kernel = many ? reinterpret_cast<const void*>(&histogram_kernel<kHistCopiesLarge>)
                                  : reinterpret_cast<const void*>(&histogram_kernel<kHistCopies>);
  return Launch(kernel, grid, kHistThreads, HistLdsBytes(many ? kHistCopiesLarge : kHistCopies), stream, keys, maxCount, countPtr,
                globalHistogram, tickets, clear, vecs);
}

// ---- MSD plan -----------------------------------------------------------------------------------------
template <uint32_t BITS>
static hipError_t PrepareMsdBits() {
  const struct {
    const void* fn;
    size_t bytes;
  } kernels[] = {
      {reinterpret_cast<const void*>(&histogram_msd_kernel<kHistCopies, BITS>), HistMsdLdsBytes(kHistCopies, BITS)},
      {reinterpret_cast<const void*>(&histogram_msd_kernel<kHistCopiesLarge, BITS>), HistMsdLdsBytes(kHistCopiesLarge, BITS)},
      {reinterpret_cast<const void*>(&scatter_msd_kernel<BITS, false>), ScatterMsdLdsWords<BITS>() * 4},
      {reinterpret_cast<const void*>(&scatter_msd_kernel<BITS, true>), ScatterMsdLdsWords<BITS>() * 4},
      {reinterpret_cast<const void*>(&bucket_sort2_kernel<BITS, kMsdCapKeys / 1024, false>), BucketSort2LdsWords<kMsdCapKeys / 1024>() * 4},
      {reinterpret_cast<const void*>(&bucket_sort2_kernel<BITS, kMsdCapKeyValue / 1024, true>), BucketSort2LdsWords<kMsdCapKeyValue / 1024>() * 4},
  };
  for (const auto& k : kernels) {
    const hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.bytes);
    if (e != hipSuccess) return e;
  }
  const struct {
    const void* fn;
    size_t bytes;
  } fused[] = {
      {reinterpret_cast<const void*>(&msd_scatter_or_pass0_kernel<BITS, false, false>), MsdFusedLdsWords<false>(BITS, false) * 4},
      {reinterpret_cast<const void*>(&msd_scatter_or_pass0_kernel<BITS, false, true>), MsdFusedLdsWords<false>(BITS, false) * 4},
      {reinterpret_cast<const void*>(&msd_buckets_or_pass1_kernel<BITS, false, false>), MsdFusedLdsWords<false>(BITS, true) * 4},
      {reinterpret_cast<const void*>(&msd_buckets_or_pass1_kernel<BITS, false, true>), MsdFusedLdsWords<false>(BITS, true) * 4},
      {reinterpret_cast<const void*>(&msd_scatter_or_pass0_kernel<BITS, true, false>), MsdFusedLdsWords<true>(BITS, false) * 4},
      {reinterpret_cast<const void*>(&msd_scatter_or_pass0_kernel<BITS, true, true>), MsdFusedLdsWords<true>(BITS, false) * 4},
      {reinterpret_cast<const void*>(&msd_buckets_or_pass1_kernel<BITS, true, false>), MsdFusedLdsWords<true>(BITS, true) * 4},
      {reinterpret_cast<const void*>(&msd_buckets_or_pass1_kernel<BITS, true, true>), MsdFusedLdsWords<true>(BITS, true) * 4},
  };
  for (const auto& k : fused) {
    const hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t PrepareMsd() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_order_check_packed_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOrderCheckPackedLdsBytes);
  if (e == hipSuccess) e = PrepareMsdBits<10>();
  if (e == hipSuccess) e = PrepareMsdBits<11>();
  for (int kv = 0; kv < 2 && e == hipSuccess; ++kv)
    e = hipFuncSetAttribute(kv ? reinterpret_cast<const void*>(&bucket_sort2_half_kernel<10, true>)
                               : reinterpret_cast<const void*>(&bucket_sort2_half_kernel<10, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BucketSort2LdsWords<kMsdHalfCap / 512, 512>() * 4));
  return e;
}

hipError_t LaunchHistogramMsd(hipStream_t stream, uint32_t grid, const MsdArgs& args) {
  const bool many = args.maxCount >= kHistManyCopiesFrom;
  const void* kernel;
  // tiles of rows x 4096 keys (the scatter's even-split tiles); a tile of full capacity is eight rows
  if (args.tileKeys == 0 || args.tileKeys % 4096u != 0 || args.tileKeys > kMsdTileKeys) return hipErrorInvalidValue;
  if (args.bits == 10)
    kernel = many ? reinterpret_cast<const void*>(&histogram_msd_kernel<kHistCopiesLarge, 10>)
                  : reinterpret_cast<const void*>(&histogram_msd_kernel<kHistCopies, 10>);
  else if (args.bits == 11)
    kernel = many ? reinterpret_cast<const void*>(&histogram_msd_kernel<kHistCopiesLarge, 11>)
                  : reinterpret_cast<const void*>(&histogram_msd_kernel<kHistCopies, 11>);
  else
    return hipErrorInvalidValue;
  return Launch(kernel, grid, kHistThreads, HistMsdLdsBytes(many ? kHistCopiesLarge : kHistCopies, args.bits), stream, args);
}

hipError_t LaunchSpineMsd(hipStream_t stream, const MsdArgs& args) {
  if (args.tiles > kMsdMaxTiles) return hipErrorInvalidValue;
  if (args.bits == 10) return Launch(reinterpret_cast<const void*>(&spine_msd_kernel<10>), 512 / 16, 1024, 0, stream, args);
  if (args.bits == 11) return Launch(reinterpret_cast<const void*>(&spine_msd_kernel<11>), 1024 / 16, 1024, 0, stream, args);
  return hipErrorInvalidValue;
}

// workgroups of the plan's scatter: one per tile, or per two tiles (keys-only, ten bits), rounded up to a multiple of 8
static uint32_t MsdScatterGrid(uint32_t tiles, bool keyValue, uint32_t bits) {
  const uint32_t units = !keyValue && bits == 10 ? (tiles + 1u) / 2u : tiles;
  return 8u * ((units + 7u) / 8u);
}

hipError_t LaunchScatterMsd(hipStream_t stream, bool keyValue, const MsdArgs& args) {
  const void* kernel;
  size_t lds;
  if (args.tileKeys == 0 || args.tileKeys % 4096u != 0 || args.tileKeys > kMsdTileKeys) return hipErrorInvalidValue;
  if (args.bits == 10) {
    kernel = keyValue ? reinterpret_cast<const void*>(&scatter_msd_kernel<10, true>)
                      : reinterpret_cast<const void*>(&scatter_msd_kernel<10, false>);
    lds = ScatterMsdLdsWords<10>() * 4;
  } else if (args.bits == 11) {
    kernel = keyValue ? reinterpret_cast<const void*>(&scatter_msd_kernel<11, true>)
                      : reinterpret_cast<const void*>(&scatter_msd_kernel<11, false>);
    lds = ScatterMsdLdsWords<11>() * 4;
  } else {
    return hipErrorInvalidValue;
  }
  // a multiple of 8 workgroups: eight chunks of consecutive tiles (keys-only: pairs of tiles), one per XCD (see the kernels)
  return Launch(kernel, MsdScatterGrid(args.tiles, keyValue, args.bits), 1024, lds, stream, args);
}

// Workgroups of the plan's bucket launch.  The full-size kernel (one workgroup per CU) takes TWO buckets per workgroup, one
// after the other (BucketSort2Body): half as many workgroups to start and to drain -- 107.4 instead of 110.9 us keys-only at
// 2^25, 175-178 instead of 179-180 key+value (tools/r06/bucket_grid.sh, bucket_grid2.sh).  The half-size kernel (two
// workgroups per CU) keeps one: 47.7 against 47.0 us keys-only at 2^24 with two.
static uint32_t MsdBucketGrid(uint32_t bits, bool halfSizeKernel) { return halfSizeKernel ? 1u << bits : (1u << bits) / 2u; }

hipError_t LaunchBucketSort2(hipStream_t stream, bool keyValue, const MsdArgs& args) {
  constexpr int kKeys = kMsdCapKeys / 1024, kPairs = kMsdCapKeyValue / 1024;
  if (args.cap == kMsdHalfCap && args.bits == 10) {
    const void* half = keyValue ? reinterpret_cast<const void*>(&bucket_sort2_half_kernel<10, true>)
                                : reinterpret_cast<const void*>(&bucket_sort2_half_kernel<10, false>);
    return Launch(half, MsdBucketGrid(args.bits, true), 512, BucketSort2LdsWords<kMsdHalfCap / 512, 512>() * 4, stream, args);
  }
  if (args.cap != (keyValue ? kMsdCapKeyValue : kMsdCapKeys)) return hipErrorInvalidValue;
  const void* kernel;
  if (args.bits == 10)
    kernel = keyValue ? reinterpret_cast<const void*>(&bucket_sort2_kernel<10, kPairs, true>)
                      : reinterpret_cast<const void*>(&bucket_sort2_kernel<10, kKeys, false>);
  else if (args.bits == 11)
    kernel = keyValue ? reinterpret_cast<const void*>(&bucket_sort2_kernel<11, kPairs, true>)
                      : reinterpret_cast<const void*>(&bucket_sort2_kernel<11, kKeys, false>);
  else
    return hipErrorInvalidValue;
  const size_t lds = (keyValue ? BucketSort2LdsWords<kPairs>() : BucketSort2LdsWords<kKeys>()) * 4;
  return Launch(kernel, MsdBucketGrid(args.bits, false), 1024, lds, stream, args);
}

// The plan's scatter / bucket launch with the fallback's pass 0 / pass 1 as its second role (bucketLaunch selects which).
// passGrid: the grid LaunchOnesweep would have used for that pass.
template <uint32_t BITS, bool KV>
static hipError_t LaunchMsdFusedBits(hipStream_t stream, bool bucketLaunch, const MsdArgs& m, const OnesweepArgs& p, uint32_t passGrid) {
  const bool dyn = p.slots != 0;
  const void* kernel =
      bucketLaunch ? (dyn ? reinterpret_cast<const void*>(&msd_buckets_or_pass1_kernel<BITS, KV, true>)
                          : reinterpret_cast<const void*>(&msd_buckets_or_pass1_kernel<BITS, KV, false>))
                   : (dyn ? reinterpret_cast<const void*>(&msd_scatter_or_pass0_kernel<BITS, KV, true>)
                          : reinterpret_cast<const void*>(&msd_scatter_or_pass0_kernel<BITS, KV, false>));
  // the larger of the two roles' grids, a multiple of 8 (the scatter derives its tile from the grid: eight chunks of tiles, one
  // per XCD; a workgroup beyond its role's range returns)
  // (the bucket launch: two buckets per workgroup, MsdBucketGrid -- a pass of up to 2^BITS / 2 tiles then has no idle workgroups)
  const uint32_t planGrid = bucketLaunch ? MsdBucketGrid(BITS, false) : MsdScatterGrid(m.tiles, KV, BITS);
  const uint32_t grid = 8u * (((planGrid > passGrid ? planGrid : passGrid) + 7u) / 8u);
  // the pass's run-time slot counts, checked like LaunchPairConfig / LaunchConfig do
  if (dyn && (p.slots % 4 != 0 || p.slots > 32u || p.tailSlots % 4 != 0 || p.tailSlots == 0 || p.tailSlots > 32u))
    return hipErrorInvalidValue;
  return Launch(kernel, grid, 1024, MsdFusedLdsWords<KV>(BITS, bucketLaunch) * 4, stream, m, p);
}

hipError_t LaunchMsdFused(hipStream_t stream, bool bucketLaunch, bool keyValue, const MsdArgs& m, const OnesweepArgs& p,
                          uint32_t passGrid) {
  if (m.tileKeys == 0 || m.tileKeys % 4096u != 0 || m.tileKeys > kMsdTileKeys ||
      (bucketLaunch && m.cap != (keyValue ? kMsdCapKeyValue : kMsdCapKeys)))
    return hipErrorInvalidValue;
  if (m.bits == 10)
    return keyValue ? LaunchMsdFusedBits<10, true>(stream, bucketLaunch, m, p, passGrid)
                    : LaunchMsdFusedBits<10, false>(stream, bucketLaunch, m, p, passGrid);
  if (m.bits == 11)
    return keyValue ? LaunchMsdFusedBits<11, true>(stream, bucketLaunch, m, p, passGrid)
                    : LaunchMsdFusedBits<11, false>(stream, bucketLaunch, m, p, passGrid);
  return hipErrorInvalidValue;
}

hipError_t LaunchOnesweep(hipStream_t stream, int configIndex, uint32_t grid, bool keyValue, bool atomicRank,
                          const OnesweepArgs& args) {
  switch (configIndex) {
    case 0: return LaunchConfig<1024, 8>(stream, grid, keyValue, atomicRank, args);
    case 1: return LaunchConfig<1024, 16>(stream, grid, keyValue, atomicRank, args);
    case 2: return LaunchConfig<1024, 32, true>(stream, grid, keyValue, atomicRank, args);
    case 3: return LaunchPairConfig<1024, 32>(stream, grid, keyValue, atomicRank, args);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace vrdx

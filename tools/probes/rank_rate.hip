// Probe (GPU box): cycles RankAtomic / RankBallot spend per 32-slot ranking of one tile's keys (16 waves per CU, one
// workgroup per CU), by key pattern.  Includes the product's kernel source to call the very same functions.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/rank_rate.hip -lamdhip64 -o /tmp/rank_rate && /tmp/rank_rate
#include "../../vulkan_radix_sort_amd/csrc/vrdx_kernels.hip"

#include <stdio.h>
#include <stdlib.h>

namespace {
constexpr int kIters = 64;

// pattern 0: random digits; 1: one digit for the whole tile (sorted input, top pass); 2: digit = lane (sorted, low pass);
// 3: four digits at random; 4: one digit per wave-slot, changing from slot to slot; 5: keys = element index, run-time
// shift 24 (exactly the top pass of ascending input)
template <int PATTERN, bool BALLOT, int ROWS_AT = 0>
__global__ __launch_bounds__(1024) void rank_probe(uint32_t* out, unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t key[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    uint32_t r = (blockIdx.x * 1024u + tid) * 32u + i + 1u;
    r ^= r >> 15; r *= 0x2C1B3C6Du; r ^= r >> 12; r *= 0x297A2D39u; r ^= r >> 15;
    // (pattern 1 reads its digit from memory, per lane: a value the compiler cannot prove wave-uniform)
    if (PATTERN == 5) { key[i] = blockIdx.x * 32768u + wave * 2048u + 64u * i + lane; continue; }
    key[i] = PATTERN == 0 ? (r & 255u) : PATTERN == 1 ? out[1 + ((tid + i) & 1)] : PATTERN == 2 ? (uint32_t)lane * 4u + (i & 3) : PATTERN == 3 ? ((r >> 8) & 3u) * 37u
                                                                                                              : (uint32_t)((i * 7 + wave) & 255);
  }
  uint32_t* const myHist = lds + ROWS_AT + wave * 256;  // ROWS_AT: word offset of the counter rows (the kernels': 32768)
  const uint32_t shift = PATTERN == 5 ? out[3] : 0u;  // run-time 24
  uint32_t acc = 0;
  __syncthreads();
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < kIters; ++it) {
    for (int i = 0; i < 4; ++i) myHist[lane + 64 * i] = 0;
    uint32_t rank[32];
    if (BALLOT)
      vrdx::RankBallot<32, false>(key, shift, myHist, lane, rank);
    else
      vrdx::RankAtomic<32, false>(key, shift, myHist, lane, rank);
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += rank[i];
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(key[i]));
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == 0x12345u) out[0] = acc;
}

template <int PATTERN, bool BALLOT, int ROWS_AT = 0>
void Run(const char* what) {
  uint32_t* out;
  unsigned long long* cyc;
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  (void)hipMalloc((void**)&out, 16);
  { const uint32_t init[4] = {0, 7, 7, 24}; (void)hipMemcpy(out, init, 16, hipMemcpyHostToDevice); }
  (void)hipMalloc((void**)&cyc, 8 * cus);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rank_probe<PATTERN, BALLOT, ROWS_AT>), hipFuncAttributeMaxDynamicSharedMemorySize, (ROWS_AT + 16 * 256) * 4);
    hipLaunchKernelGGL((rank_probe<PATTERN, BALLOT, ROWS_AT>), dim3(cus), dim3(1024), (ROWS_AT + 16 * 256) * 4, 0, out, cyc);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); exit(1); }
  unsigned long long* h = (unsigned long long*)malloc(8 * cus);
  (void)hipMemcpy(h, cyc, 8 * cus, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < cus; ++i) sum += (double)h[i];
  printf("%-8s %-44s %8.0f ticks per tile ranking (32 slots x 16 waves)\n", BALLOT ? "ballot" : "atomic", what, sum / cus / kIters);
  free(h);
  (void)hipFree(out);
  (void)hipFree(cyc);
}
}  // namespace

int main() {
  Run<0, false>("random digits");
  Run<1, false>("one digit (sorted input, top pass)");
  Run<2, false>("digit = lane (sorted input, low pass)");
  Run<3, false>("four digits at random");
  Run<4, false>("one digit per slot, changing");
  Run<5, false>("keys = index, shift 24 (ascending input, top pass)");
  Run<5, false, 32768>("keys = index, shift 24, counter rows at 128 KiB");
  Run<0, false, 32768>("random digits, counter rows at 128 KiB");
  Run<0, true>("random digits");
  Run<1, true>("one digit (sorted input, top pass)");
  Run<3, true>("four digits at random");
  return 0;
}

// Probe: when several lanes of ONE wave64 instruction do ds_add_rtn_u32 on the same LDS address,
// are the returned old values ascending in lane order?  (The ISA does not promise it.)
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 lds_atomic_order.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

__global__ void probe(const uint32_t* digits, uint32_t* ranks, int slots, int waves_busy) {
  __shared__ uint32_t cnt[16 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * 256; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  uint32_t* mine = cnt + wave * 256;
  const size_t base = ((size_t)blockIdx.x * (blockDim.x / 64) + wave) * slots * 64;
  for (int s = 0; s < slots; ++s) {
    const uint32_t d = digits[base + s * 64 + lane];
    ranks[base + s * 64 + lane] = atomicAdd(&mine[d], 1u);
  }
}

int main() {
  const int blocks = 512, threads = 1024, slots = 64;
  const size_t total = (size_t)blocks * (threads / 64) * slots * 64;
  std::vector<uint32_t> h(total), r(total);
  std::mt19937 g(1);
  // mixture of entropy levels per wave-slot
  for (size_t w = 0; w < total / 64; ++w) {
    const int mode = w % 6;
    const uint32_t mask = mode == 0 ? 0u : mode == 1 ? 1u : mode == 2 ? 3u : mode == 3 ? 15u : mode == 4 ? 0x21u : 255u;
    for (int l = 0; l < 64; ++l) h[w * 64 + l] = g() & mask;
  }
  uint32_t *dd, *dr;
  hipMalloc(&dd, total * 4); hipMalloc(&dr, total * 4);
  hipMemcpy(dd, h.data(), total * 4, hipMemcpyHostToDevice);
  long bad = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, dd, dr, slots, 0);
    hipMemcpy(r.data(), dr, total * 4, hipMemcpyDeviceToHost);
    // expected: stable rank = running count per (wave, digit) in (slot, lane) order
    for (size_t wv = 0; wv < total / (64 * slots); ++wv) {
      uint32_t c[256] = {0};
      for (int s = 0; s < slots; ++s)
        for (int l = 0; l < 64; ++l) {
          const size_t i = (wv * slots + s) * 64 + l;
          if (r[i] != c[h[i]]++) ++bad;
        }
    }
  }
  printf("lds_atomic_order: %zu lane-ops x5, mismatches vs lane-ascending order: %ld\n", total, bad);
  return bad != 0;
}

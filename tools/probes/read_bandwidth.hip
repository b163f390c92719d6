// Probe (GPU box): what a one-pass streaming READ of N uint32 keys can reach on this device, and what
// each ingredient of histogram_kernel adds to it.  Not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/read_bandwidth.hip -o /tmp/read_bandwidth
//   /tmp/read_bandwidth [log2n]
//
// MODE 0: xor-reduce only (one global atomic per workgroup)     1: one LDS atomic per key (digit 0)
//      2: four LDS atomics per key                              3: mode 2 + the reduction into 1024 global bins
// Loads are 16 B per lane, U of them in flight per lane, grid-stride over chunks of THREADS * U uint4.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <int THREADS, int U, int MODE, int COPIES, bool NT>
__global__ __launch_bounds__(THREADS) void read_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                       uint32_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t bins[];  // [place][digit][copy]
  const uint32_t tid = threadIdx.x;
  if (MODE >= 1) {
    for (uint32_t i = tid; i < 4 * 256 * COPIES; i += THREADS) bins[i] = 0;
    __syncthreads();
  }
  const uint32_t copy = tid & (COPIES - 1);
  uint32_t acc = 0;
  auto count = [&](uint32_t key) {
    if (MODE == 0) {
      acc ^= key;
    } else if (MODE == 1) {
      atomicAdd(&bins[(key & 0xFFu) * COPIES + copy], 1u);
    } else {
#pragma unroll
      for (uint32_t p = 0; p < 4; ++p) atomicAdd(&bins[(p * 256 + ((key >> (8 * p)) & 0xFFu)) * COPIES + copy], 1u);
    }
  };
  const uint32_t nvec = n >> 2;
  const uint4* keys4 = reinterpret_cast<const uint4*>(keys);
  const uint32_t chunk = THREADS * U;
  for (uint32_t base = blockIdx.x * chunk; base < nvec; base += gridDim.x * chunk) {
    uint4 k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + u * THREADS + tid;
      if (i < nvec) {
        if (NT) {
          typedef uint32_t native4 __attribute__((ext_vector_type(4)));
          const native4 v = __builtin_nontemporal_load(reinterpret_cast<const native4*>(keys4) + i);
          k[u] = make_uint4(v.x, v.y, v.z, v.w);
        } else {
          k[u] = keys4[i];
        }
      } else
        k[u] = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + u * THREADS + tid;
      if (i < nvec) {
        count(k[u].x);
        count(k[u].y);
        count(k[u].z);
        count(k[u].w);
      }
    }
  }
  if (MODE == 0) {
    if (acc == 0x12345678u) atomicAdd(out, 1u);  // keeps the loads alive, practically never taken
  } else {
    __syncthreads();
    for (uint32_t b = tid; b < 1024; b += THREADS) {
      uint32_t sum = 0;
#pragma unroll
      for (uint32_t c = 0; c < COPIES; ++c) sum += bins[b * COPIES + ((c + tid) & (COPIES - 1))];
      if (MODE == 3) {
        if (sum != 0) atomicAdd(&out[b], sum);
      } else if (sum == 0x12345678u) {
        atomicAdd(out, 1u);
      }
    }
  }
}

// The sort kernels' own access pattern: a workgroup reads one tile of THREADS * KPT keys, every wave KPT
// rows of 64 consecutive dwords (4 bytes per lane per load, KPT loads in flight per lane); VEC = 4: the
// same bytes as 16-byte loads (KPT / 4 of them per lane).
template <int THREADS, int KPT, int VEC>
__global__ __launch_bounds__(THREADS) void tile_read_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                            uint32_t* __restrict__ out) {
  const uint32_t tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t base = tile * (THREADS * KPT) + wave * (KPT * 64);
  uint32_t acc = 0;
  if (VEC == 1) {
    uint32_t k[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) k[i] = keys[base + i * 64 + lane];
#pragma unroll
    for (int i = 0; i < KPT; ++i) acc ^= k[i];
  } else {
    const uint4* keys4 = reinterpret_cast<const uint4*>(keys + base);
    uint4 k[KPT / 4];
#pragma unroll
    for (int i = 0; i < KPT / 4; ++i) k[i] = keys4[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < KPT / 4; ++i) acc ^= k[i].x ^ k[i].y ^ k[i].z ^ k[i].w;
  }
  if (acc == 0x12345678u) atomicAdd(out, 1u);
}

template <int THREADS, int KPT, int VEC>
static void RunTile(const uint32_t* keys, uint32_t n, uint32_t* out) {
  auto fn = tile_read_kernel<THREADS, KPT, VEC>;
  const int grid = (int)(n / (THREADS * KPT));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(THREADS), 0, 0, keys, n, out);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(THREADS), 0, 0, keys, n, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps;
  printf("tile read: threads=%4d keys/thread=%2d bytes/lane/load=%2d tiles=%5d  %7.2f us  %5.2f TB/s\n", THREADS, KPT,
         4 * VEC, grid, us, 4.0 * n / us * 1e-6);
  fflush(stdout);
}

template <int THREADS, int U, int MODE, int COPIES, bool NT>
static void Run(const uint32_t* keys, uint32_t n, uint32_t* out, int grid, int cus) {
  const size_t lds = MODE >= 1 ? (size_t)4 * 256 * COPIES * 4 : 0;
  auto fn = read_kernel<THREADS, U, MODE, COPIES, NT>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(THREADS), lds, 0, keys, n, out);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(THREADS), lds, 0, keys, n, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps;
  printf("threads=%4d unroll=%d mode=%d copies=%2d nt=%d grid=%5d (%.1f/CU)  %7.2f us  %5.2f TB/s\n", THREADS, U, MODE,
         COPIES, (int)NT, grid, (double)grid / cus, us, 4.0 * n / us * 1e-6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int log2n = argc > 1 ? atoi(argv[1]) : 25;
  const uint32_t n = 1u << log2n;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<uint32_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (auto& x : h) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    x = (uint32_t)(s >> 16);
  }
  uint32_t *keys, *out;
  CHECK(hipMalloc(&keys, (size_t)n * 4));
  CHECK(hipMalloc(&out, 4096));
  CHECK(hipMemcpy(keys, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(out, 0, 4096));
  printf("n = 2^%d, %d CUs\n", log2n, cus);

  puts("-- pure read: threads x unroll x grid");
  for (int g : {1, 2, 4, 8, 16}) Run<256, 4, 0, 8, false>(keys, n, out, cus * g, cus);
  for (int g : {1, 2, 4, 8}) Run<256, 8, 0, 8, false>(keys, n, out, cus * g, cus);
  for (int g : {1, 2, 4}) Run<512, 4, 0, 8, false>(keys, n, out, cus * g, cus);
  for (int g : {1, 2}) Run<1024, 4, 0, 8, false>(keys, n, out, cus * g, cus);
  for (int g : {1, 2}) Run<1024, 8, 0, 8, false>(keys, n, out, cus * g, cus);
  for (int g : {1, 2}) Run<1024, 2, 0, 8, false>(keys, n, out, cus * g, cus);
  puts("-- the sort kernels' tile read: one tile per workgroup, wave-striped");
  RunTile<1024, 32, 1>(keys, n, out);
  RunTile<1024, 32, 4>(keys, n, out);
  RunTile<1024, 16, 1>(keys, n, out);
  RunTile<1024, 16, 4>(keys, n, out);
  RunTile<1024, 8, 1>(keys, n, out);
  RunTile<512, 32, 1>(keys, n, out);
  RunTile<256, 32, 1>(keys, n, out);
  RunTile<256, 32, 4>(keys, n, out);
  puts("-- pure read, non-temporal loads");
  for (int g : {1, 2}) Run<1024, 4, 0, 8, true>(keys, n, out, cus * g, cus);
  for (int g : {4, 8}) Run<256, 4, 0, 8, true>(keys, n, out, cus * g, cus);
  puts("-- one LDS atomic per key");
  Run<1024, 4, 1, 32, false>(keys, n, out, cus, cus);
  Run<1024, 4, 1, 8, false>(keys, n, out, cus, cus);
  Run<1024, 4, 1, 8, false>(keys, n, out, cus * 2, cus);
  Run<512, 4, 1, 8, false>(keys, n, out, cus * 4, cus);
  Run<256, 4, 1, 8, false>(keys, n, out, cus * 8, cus);
  puts("-- four LDS atomics per key");
  Run<1024, 4, 2, 32, false>(keys, n, out, cus, cus);
  Run<1024, 8, 2, 32, false>(keys, n, out, cus, cus);
  Run<1024, 4, 2, 8, false>(keys, n, out, cus, cus);
  Run<1024, 4, 2, 8, false>(keys, n, out, cus * 2, cus);
  Run<1024, 4, 2, 16, false>(keys, n, out, cus * 2, cus);
  Run<512, 4, 2, 8, false>(keys, n, out, cus * 4, cus);
  Run<512, 4, 2, 16, false>(keys, n, out, cus * 4, cus);
  Run<256, 4, 2, 8, false>(keys, n, out, cus * 8, cus);
  Run<256, 8, 2, 8, false>(keys, n, out, cus * 8, cus);
  Run<256, 4, 2, 4, false>(keys, n, out, cus * 8, cus);
  puts("-- four LDS atomics per key + reduction into the 1024 global bins (= histogram_kernel)");
  Run<1024, 4, 3, 32, false>(keys, n, out, cus, cus);
  Run<1024, 4, 3, 8, false>(keys, n, out, cus, cus);
  Run<1024, 4, 3, 8, false>(keys, n, out, cus * 2, cus);
  Run<512, 4, 3, 8, false>(keys, n, out, cus * 4, cus);
  Run<256, 4, 3, 8, false>(keys, n, out, cus * 8, cus);
  return 0;
}

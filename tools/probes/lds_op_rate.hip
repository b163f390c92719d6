// Probe (GPU box): LDS-array cycles per wave-instruction of the operations the sort kernels are made of, with
// 16 waves per CU issuing them back to back (the shape of the rank / regroup / histogram phases).
// Not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/lds_op_rate.hip -o /tmp/lds_op_rate && /tmp/lds_op_rate
//
// mode 0  ds_add_u32      histogram bins, 32 replicas (every lane its own bank)
//      1  ds_add_u32      histogram bins, 16 replicas
//      2  ds_add_u32      histogram bins, 8 replicas
//      3  ds_add_rtn_u32  random digit of a wave-private 256-counter row (RankAtomic)
//      4  ds_add_u32      the same addresses, nothing returned
//      5  ds_read_b32     random word of a wave-private 256-word row (RegroupKeys: counter lookup)
//      6  ds_write_b32    random word of a 32768-word buffer (RegroupKeys: staging store)
//      7  ds_read_b128    consecutive quads (scatter)
//      8  ds_add_rtn_u32  wave-private row, digits drawn from 4 values (few-distinct)
//      9  ds_add_rtn_u32  lane 0 only, the SAME digit in every wave's row (sorted input, top pass: RankAtomic's uniform path)
//     10  ds_add_rtn_u32  all 64 lanes on that one counter
//     11  ds_add_rtn_u32  lane 0 only, the waves' counters in 16 different banks
//     12  ds_add_u32      lane 0 only, the same digit in every wave's row, nothing returned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

constexpr int kThreads = 1024;
constexpr int kOps = 16;     // per iteration and lane
constexpr int kIters = 256;

__device__ __forceinline__ uint32_t Mix(uint32_t x) {
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void probe(uint32_t* out, unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint32_t i = tid; i < 36864; i += kThreads) lds[i] = 0;
  __syncthreads();
  uint32_t addr[kOps];
#pragma unroll
  for (int i = 0; i < kOps; ++i) {
    const uint32_t r = Mix((blockIdx.x * kThreads + tid) * 16u + i + 1u);
    const uint32_t d = r & 255u;
    if (MODE == 0) addr[i] = ((i & 3) * 256 + d) * 32 + (lane & 31);
    else if (MODE == 1) addr[i] = ((i & 3) * 256 + d) * 16 + (lane & 15);
    else if (MODE == 2) addr[i] = ((i & 3) * 256 + d) * 8 + (lane & 7);
    else if (MODE == 3 || MODE == 4 || MODE == 5) addr[i] = 32768 + wave * 256 + d;
    else if (MODE == 6) addr[i] = (r >> 8) & 32767u;
    else if (MODE == 7) addr[i] = 4u * ((tid + i * kThreads) & 8191u);
    else if (MODE == 8) addr[i] = 32768 + wave * 256 + ((r >> 8) & 3u) * 37u;
    else if (MODE == 11) addr[i] = 32768 + wave * 256 + 5 + wave * 4;
    else addr[i] = 32768 + wave * 256 + 5;
  }
  uint32_t acc = 0;
  __syncthreads();
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kOps; ++i) {
      if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 4) {
        __hip_atomic_fetch_add(&lds[addr[i]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 3 || MODE == 8 || MODE == 10) {
        acc += __hip_atomic_fetch_add(&lds[addr[i]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 9 || MODE == 11) {
        if (lane == 0) acc += __hip_atomic_fetch_add(&lds[addr[i]], 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 12) {
        if (lane == 0) __hip_atomic_fetch_add(&lds[addr[i]], 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 5) {
        acc += __hip_atomic_load(&lds[addr[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 6) {
        __hip_atomic_store(&lds[addr[i]], acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q = *reinterpret_cast<volatile u32x4*>(&lds[addr[i]]);
        acc += q[0] ^ q[1] ^ q[2] ^ q[3];
      }
    }
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == 0x12345u) out[0] = acc;
}

template <int MODE>
static void Run(const char* what) {
  uint32_t* out;
  unsigned long long* cyc;
  int cus = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  CHECK(hipMalloc((void**)&out, 4));
  CHECK(hipMalloc((void**)&cyc, 8 * cus));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(kThreads), 36864 * 4, 0, out, cyc);
  CHECK(hipDeviceSynchronize());
  unsigned long long* h = (unsigned long long*)malloc(8 * cus);
  CHECK(hipMemcpy(h, cyc, 8 * cus, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int i = 0; i < cus; ++i) sum += (double)h[i];
  const double perInstr = sum / cus / ((double)kIters * kOps * (kThreads / 64));
  printf("mode %d  %-58s %6.2f clock64 ticks per wave-instruction per CU\n", MODE, what, perInstr);
  free(h);
  CHECK(hipFree(out));
  CHECK(hipFree(cyc));
}

int main() {
  // clock64() ticks: calibrate against wall time once
  Run<0>("ds_add_u32, 32 replicas (own bank)");
  Run<1>("ds_add_u32, 16 replicas");
  Run<2>("ds_add_u32, 8 replicas");
  Run<3>("ds_add_rtn_u32, random digit, wave-private row");
  Run<4>("ds_add_u32, random digit, wave-private row");
  Run<5>("ds_read_b32, random word of a wave-private row");
  Run<6>("ds_write_b32, random word of 32768");
  Run<7>("ds_read_b128, consecutive quads");
  Run<8>("ds_add_rtn_u32, 4 distinct digits, wave-private row");
  Run<9>("ds_add_rtn_u32, lane 0 only, one digit in every wave's row");
  Run<10>("ds_add_rtn_u32, 64 lanes on one counter per wave");
  Run<11>("ds_add_rtn_u32, lane 0 only, 16 banks");
  Run<12>("ds_add_u32, lane 0 only, one digit in every wave's row");
  return 0;
}

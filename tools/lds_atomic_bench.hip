// Micro-benchmark: LDS atomic-add throughput on MI355X for the access patterns of the Gram kernels.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/lds_atomic_bench && /tmp/lds_atomic_bench
// Prints atomics per clock per CU for: random u32 adds into a 100 KB tile, random byte-packed adds, random f32 adds,
// and conflict-free (lane-linear) u32 adds, with 16 waves per CU (one 1024-thread workgroup, as k_itemknn_fit).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ void __launch_bounds__(1024) k_bench(int iters, int words, unsigned long long* cycles, uint32_t* sink) {
  extern __shared__ uint32_t acc[];
  for (int w = threadIdx.x; w < words; w += 1024) acc[w] = 0;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      x = x * 1664525u + 1013904223u;
      const uint32_t r = (x >> 8) % (uint32_t)(MODE == 1 ? words * 4 : words);
      if (MODE == 0) atomicAdd(&acc[r], 1u);
      if (MODE == 1) atomicAdd(&acc[r >> 2], 1u << (8 * (r & 3)));
      if (MODE == 2) atomicAdd(reinterpret_cast<float*>(acc) + r, 1.0f);
      if (MODE == 3) atomicAdd(&acc[(threadIdx.x + e * 1024 + i * 64) % words], 1u);
      if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (r >> 1), (unsigned long long)x);
      if (MODE == 5) { uint32_t old = atomicAdd(&acc[r], 1u); x += old; }
      if (MODE == 6) atomicAdd(reinterpret_cast<double*>(acc) + (r >> 1), 1.0 + (double)(x & 1023u) * 1e-3);
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) atomicAdd(cycles, t1 - t0);
  uint32_t s = 0;
  for (int w = threadIdx.x; w < words; w += 1024) s += acc[w];
  if (s == 0xdeadbeef) sink[0] = s;
}

template <int MODE>
void run(const char* name, int words) {
  unsigned long long* d_cycles;
  uint32_t* sink;
  hipMalloc(&d_cycles, 8);
  hipMalloc(&sink, 8);
  hipMemset(d_cycles, 0, 8);
  const int iters = 2000, grid = 256;
  hipFuncSetAttribute((const void*)k_bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, words * 4);
  hipLaunchKernelGGL(k_bench<MODE>, dim3(grid), dim3(1024), words * 4, 0, iters, words, d_cycles, sink);
  hipDeviceSynchronize();
  hipMemset(d_cycles, 0, 8);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(k_bench<MODE>, dim3(grid), dim3(1024), words * 4, 0, iters, words, d_cycles, sink);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  unsigned long long c = 0;
  hipMemcpy(&c, d_cycles, 8, hipMemcpyDeviceToHost);
  const double atomics_per_cu = (double)iters * 8 * 1024;
  printf("%-34s %7.2f atomics/clk/CU (s_memtime)   %8.3f ms  => %.3e atomics/s chip-wide\n", name, atomics_per_cu / ((double)c / grid), ms,
         atomics_per_cu * grid / (ms * 1e-3));
}

int main() {
  run<0>("u32 add, random, 25600 words", 25600);
  run<1>("byte-packed add, random, 6400 words", 6400);
  run<2>("f32 add, random, 25600 words", 25600);
  run<3>("u32 add, lane-linear", 25600);
  run<4>("u64 add, random, 12800 qwords", 25600);
  run<5>("u32 add returning, random", 25600);
  run<6>("f64 add, random, 12288 doubles", 24576);
  return 0;
}

// hipMalloc / hipFree cost by size (the block cache of rp_api.hip exists because of it): hipcc -O2 tools/alloc_bench.hip -o /tmp/alloc_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double ms(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
  return std::chrono::duration<double, std::milli>(b - a).count();
}
int main() {
  (void)hipFree(nullptr);
  for (size_t gb : {1, 20, 24, 28, 31, 32, 33, 36, 40, 40}) {
    void* p = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, gb << 30);
    auto t1 = std::chrono::steady_clock::now();
    (void)hipMemsetAsync(p, 0, gb << 30, nullptr);
    (void)hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    (void)hipFree(p);
    auto t3 = std::chrono::steady_clock::now();
    printf("%zu GiB: hipMalloc %.1f ms (%d), first memset %.1f ms, hipFree %.1f ms\n", gb, ms(t0, t1), (int)e, ms(t1, t2), ms(t2, t3));
  }
  // several live blocks of 20 GiB
  std::vector<void*> live;
  for (int i = 0; i < 8; ++i) {
    void* p = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, (size_t)20 << 30);
    auto t1 = std::chrono::steady_clock::now();
    printf("live block %d of 20 GiB: hipMalloc %.1f ms (%d)\n", i, ms(t0, t1), (int)e);
    if (e == hipSuccess) live.push_back(p);
  }
  for (void* p : live) (void)hipFree(p);
  for (int rep = 0; rep < 2; ++rep) {
    void* p = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, (size_t)40 << 30);
    auto t1 = std::chrono::steady_clock::now();
    printf("after those: 40 GiB hipMalloc %.1f ms (%d)\n", ms(t0, t1), (int)e);
    (void)hipFree(p);
  }
  return 0;
}

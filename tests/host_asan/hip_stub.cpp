// Test infrastructure: a HOST stand-in for the HIP runtime entry points the library's host code calls, so that its host logic - context,
// block cache, grow-only scratch, timing bookkeeping, option and argument validation (rp_api.hip) and, round 6, the whole host
// orchestration of the fits and predicts (rp_csr / rp_itemknn* / rp_predict* / rp_ease / rp_ease_dist: plans, streams, events, plane and
// block caches, collectives on a loop-back communicator, agreed allocations) - can run under AddressSanitizer and
// UndefinedBehaviorSanitizer on a machine without a GPU (sanitizers belong on the CPU build).  "Device" memory is
// malloc'ed host memory; streams and events are counters; kernels are never launched.
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>

#include <set>

static std::set<void*> g_live;
static int g_fail_next_malloc = 0;  // hipMalloc fails this many times (the cache's trim-and-retry paths)
extern "C" void stub_fail_mallocs(int n) { g_fail_next_malloc = n; }
extern "C" int stub_live_blocks() { return (int)g_live.size(); }

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
  p->multiProcessorCount = 256;
  p->totalGlobalMem = (size_t)288 << 30;
  return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) {
  if (g_fail_next_malloc > 0) {
    --g_fail_next_malloc;
    *p = nullptr;
    return hipErrorOutOfMemory;
  }
  *p = calloc(n ? n : 1, 1);  // (zeroed: the host reads counts and flags back that no kernel ever wrote)
  if (!*p) return hipErrorOutOfMemory;
  g_live.insert(*p);
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  if (!g_live.erase(p)) abort();  // a pointer the "device" never handed out, or a double free
  free(p);
  return hipSuccess;
}
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  *total_b = (size_t)288 << 30;
  *free_b = (size_t)200 << 30;
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }
// what a host-only compile of a file with kernels still references
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3*, dim3*, size_t*, hipStream_t*) { return hipSuccess; }
static long g_launches = 0;
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }
long stub_launches() { return g_launches; }
}

// Test infrastructure (run by tests/test_abi_cpu.py::test_host_code_under_asan): the host side of rp_api.hip under AddressSanitizer
// with the stub runtime of hip_stub.cpp - argument validation, options, the block cache (reuse, limit, trim, retry after a failed
// allocation), the grow-only scratch, timing bookkeeping, scan launch plumbing.  Exit code 0 = every check held (and ASan was silent).
#include <stdio.h>
#include <string.h>

#include "../../recpack_amd/csrc/rp_common.h"

extern "C" void stub_fail_mallocs(int n);
extern "C" int stub_live_blocks();

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      fprintf(stderr, "check failed at line %d: %s\n", __LINE__, #cond);   \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main() {
  CHECK(rphip_version() == RPHIP_VERSION);
  int n = 0;
  CHECK(rphip_device_count(&n) == RPHIP_OK && n == 1);
  CHECK(rphip_device_count(nullptr) == RPHIP_EINVAL && strlen(rphip_last_error()) > 0);
  rphip_ctx* ctx = nullptr;
  CHECK(rphip_ctx_create(0, nullptr, nullptr) == RPHIP_EINVAL);
  CHECK(rphip_ctx_create(3, nullptr, &ctx) == RPHIP_EINVAL && ctx == nullptr);
  CHECK(rphip_ctx_create(0, nullptr, &ctx) == RPHIP_OK && ctx != nullptr);
  int ncu = 0;
  int64_t hbm = 0;
  char arch[8];
  CHECK(rphip_device_info(ctx, &ncu, &hbm, arch, sizeof(arch)) == RPHIP_OK && ncu == 256 && !strncmp(arch, "gfx950", 6) && strlen(arch) == 7);
  // options: every documented name, bad values, an unknown name
  const char* names[] = {"predict_exact64", "predict_half", "gemm_mode", "gemm_wide", "gemm_presplit", "gemm_supertile"};
  for (const char* nm : names) CHECK(rphip_ctx_set_option(ctx, nm, 0) == RPHIP_OK && rphip_ctx_set_option(ctx, nm, strcmp(nm, "gemm_mode") ? 1 : 2) == RPHIP_OK);
  CHECK(rphip_ctx_set_option(ctx, "gemm_mode", 2) == RPHIP_OK && rphip_ctx_set_option(ctx, "gemm_mode", 1) == RPHIP_EINVAL);
  CHECK(rphip_ctx_set_option(ctx, "predict_tile32", 1000) == RPHIP_EINVAL && rphip_ctx_set_option(ctx, "predict_tile32", 2048) == RPHIP_OK);
  CHECK(rphip_ctx_set_option(ctx, "gemm_flush_k", 100) == RPHIP_EINVAL && rphip_ctx_set_option(ctx, "gemm_flush_k", 512) == RPHIP_OK);
  CHECK(rphip_ctx_set_option(ctx, "gemm_workspace_bytes", 1) == RPHIP_EINVAL);
  CHECK(rphip_ctx_set_option(ctx, "no_such_option", 1) == RPHIP_EINVAL && rphip_ctx_set_option(nullptr, "gemm_mode", 2) == RPHIP_EINVAL);
  // block cache: a freed block comes back for a request of the same (rounded) size; other sizes allocate
  const int base = stub_live_blocks();
  void *a = nullptr, *b = nullptr, *c = nullptr;
  CHECK(rp_dev_alloc(ctx, 1000, &a) == RPHIP_OK && rp_dev_alloc(ctx, 5000, &b) == RPHIP_OK);
  memset(a, 1, 1000);
  memset(b, 2, 5000);
  rp_dev_free(ctx, a);
  CHECK(ctx->cached_bytes == 1024 && ctx->free_blocks.size() == 1);
  CHECK(rp_dev_alloc(ctx, 1024, &c) == RPHIP_OK && c == a && ctx->cached_bytes == 0);
  rp_dev_free(ctx, c);
  rp_dev_free(ctx, b);
  rp_dev_free(ctx, nullptr);
  CHECK(ctx->free_blocks.size() == 2 && stub_live_blocks() == base + 2);
  // best fit with splitting: a request of at least 1/16 of a cached block is cut from it, the rest stays cached and the parts
  // are joined again when freed; a request below 1/16 of every cached block goes to the runtime
  {
    rp_dev_trim(ctx);
    void *big = nullptr, *p1 = nullptr, *p2 = nullptr, *small = nullptr;
    const int b0 = stub_live_blocks();
    CHECK(rp_dev_alloc(ctx, 1 << 16, &big) == RPHIP_OK);
    rp_dev_free(ctx, big);
    CHECK(rp_dev_cache_serves(ctx, 1 << 14, 8) == 4 && rp_dev_cache_serves(ctx, 1 << 17, 1) == 0);
    CHECK(rp_dev_alloc(ctx, 1 << 14, &p1) == RPHIP_OK && p1 == big && ctx->cached_bytes == (3u << 14) && stub_live_blocks() == b0 + 1);
    CHECK(rp_dev_alloc(ctx, 1 << 14, &p2) == RPHIP_OK && p2 == (char*)big + (1 << 14) && ctx->free_blocks.size() == 1);
    memset(p1, 3, 1 << 14);
    memset(p2, 4, 1 << 14);
    CHECK(rp_dev_alloc(ctx, 256, &small) == RPHIP_OK && stub_live_blocks() == b0 + 2);  // (256 < 32 KB / 16: not carved from it)
    rp_dev_trim(ctx);  // (a split region with live parts cannot go back)
    CHECK(ctx->free_blocks.size() == 1 && stub_live_blocks() == b0 + 2);
    rp_dev_free(ctx, p1);
    CHECK(ctx->free_blocks.size() == 2);  // (p1 and the tail do not touch)
    rp_dev_free(ctx, p2);
    CHECK(ctx->free_blocks.size() == 1 && ctx->free_blocks[0].bytes == (1u << 16) && ctx->cached_bytes == (1u << 16));
    rp_dev_free(ctx, small);
    rp_dev_trim(ctx);
    CHECK(ctx->free_blocks.empty() && ctx->cached_bytes == 0 && stub_live_blocks() == b0);
    CHECK(rp_dev_alloc(ctx, 1000, &a) == RPHIP_OK && rp_dev_alloc(ctx, 5000, &b) == RPHIP_OK);
    rp_dev_free(ctx, a);
    rp_dev_free(ctx, b);
  }
  // a failing allocation gives the cache back and retries
  stub_fail_mallocs(1);
  CHECK(rp_dev_alloc(ctx, 1 << 20, &a) == RPHIP_OK && ctx->free_blocks.empty() && stub_live_blocks() == base + 1);
  stub_fail_mallocs(1);  // (nothing cached now: one failure is final)
  CHECK(rp_dev_alloc(ctx, 1 << 21, &b) == RPHIP_ENOMEM && b == nullptr);
  rp_dev_free(ctx, a);
  // the limit: a block that would push the cache over it is released at once
  CHECK(rphip_ctx_set_option(ctx, "cache_limit_bytes", 4096) == RPHIP_OK && ctx->free_blocks.empty());
  CHECK(rp_dev_alloc(ctx, 8192, &a) == RPHIP_OK);
  rp_dev_free(ctx, a);
  CHECK(ctx->free_blocks.empty() && stub_live_blocks() == base);
  CHECK(rphip_ctx_set_option(ctx, "cache_limit_bytes", -1) == RPHIP_EINVAL);
  // a block the cache never saw goes straight to the runtime
  void* raw = nullptr;
  CHECK(hipMalloc(&raw, 64) == hipSuccess);
  rp_dev_free(ctx, raw);
  CHECK(stub_live_blocks() == base);
  // scratch: grows, keeps its buffer for smaller requests, retries after trimming the cache
  void *s1 = nullptr, *s2 = nullptr;
  CHECK(rp_scratch(ctx, 3, 100, &s1) == RPHIP_OK && rp_scratch(ctx, 3, 50, &s2) == RPHIP_OK && s1 == s2);
  memset(s1, 3, 100);
  CHECK(rp_scratch(ctx, 3, 100000, &s2) == RPHIP_OK);
  memset(s2, 4, 100000);
  CHECK(rphip_ctx_set_option(ctx, "cache_limit_bytes", 1 << 30) == RPHIP_OK);
  CHECK(rp_dev_alloc(ctx, 4096, &a) == RPHIP_OK);
  rp_dev_free(ctx, a);
  stub_fail_mallocs(1);
  CHECK(rp_scratch(ctx, 4, 1 << 20, &s1) == RPHIP_OK && ctx->free_blocks.empty());
  stub_fail_mallocs(1);
  CHECK(rp_scratch(ctx, 6, 1 << 20, &s1) == RPHIP_ENOMEM);
  // timing bookkeeping: pairs are pooled and resolved lazily
  hipEvent_t ev;
  for (int i = 0; i < 5000; ++i) {  // (past the 4096 pending pairs that force a resolve)
    rp_time_begin(ctx, &ev);
    rp_time_end(ctx, ev, RPHIP_T_PREDICT, 10.0);
  }
  rphip_timings tm;
  CHECK(rphip_get_timings(ctx, &tm) == RPHIP_OK && tm.launches[RPHIP_T_PREDICT] == 5000 && tm.work[RPHIP_T_PREDICT] == 50000.0);
  CHECK(tm.ms[RPHIP_T_PREDICT] == 5000.0 && rphip_reset_timings(ctx) == RPHIP_OK);
  CHECK(rphip_get_timings(ctx, &tm) == RPHIP_OK && tm.launches[RPHIP_T_PREDICT] == 0);
  CHECK(rphip_get_timings(nullptr, &tm) == RPHIP_EINVAL && rphip_get_timings(ctx, nullptr) == RPHIP_EINVAL);
  // the scan's host plumbing (kernels are not run here): block sums in scratch
  int32_t* in = nullptr;
  CHECK(rp_dev_alloc(ctx, 4 * 10001, (void**)&in) == RPHIP_OK);
  CHECK(rp_exclusive_scan_i32(ctx, in, in, 10000) == RPHIP_OK && rp_exclusive_scan_i32(ctx, in, in, 0) == RPHIP_OK);
  rp_dev_free(ctx, in);
  int64_t counters[8];
  CHECK(rphip_get_counters(ctx, counters) == RPHIP_OK && rphip_get_counters(ctx, nullptr) == RPHIP_EINVAL);
  CHECK(rphip_ctx_trim(ctx) == RPHIP_OK && rphip_ctx_sync(ctx) == RPHIP_OK && rphip_ctx_sync(nullptr) == RPHIP_EINVAL);
  CHECK(rphip_ctx_destroy(ctx) == RPHIP_OK && rphip_ctx_destroy(nullptr) == RPHIP_OK);
  CHECK(stub_live_blocks() == 0);  // nothing of the "device" left behind
  printf("host asan driver: ok\n");
  return 0;
}

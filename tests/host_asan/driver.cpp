// Test infrastructure (run by tests/test_abi_cpu.py::test_host_code_under_asan): the host side of rp_api.hip under AddressSanitizer
// with the stub runtime of hip_stub.cpp - argument validation, options, the block cache (reuse, limit, trim, retry after a failed
// allocation), the grow-only scratch, timing bookkeeping, scan launch plumbing.  Exit code 0 = every check held (and ASan was silent).
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/recpack_hip_diag.h"
#include "../../recpack_amd/csrc/rp_common.h"

extern "C" void stub_fail_mallocs(int n);
extern "C" int stub_live_blocks();
extern "C" long stub_launches();

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      fprintf(stderr, "check failed at line %d: %s [%s]\n", __LINE__, #cond, rphip_last_error()); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

// ---- (round 6) the host ORCHESTRATION of the fits and predicts with the kernels as no-ops: every launch plan, allocation, stream / event
// dance, read-back and release of rp_csr, rp_itemknn*, rp_predict*, rp_ease (rp_spd_inverse: three streams, plane cache, block cache)
// and rp_ease_dist (four streams, the collectives of a loop-back communicator, agreed allocations) runs under the sanitizers.  The
// "device" memory holds whatever malloc and the memsets left, so the numbers mean nothing; what is checked is that no call reads or
// writes outside its blocks, leaks one, or trips UBSan - and that the calls that must refuse do.
static int host_drivers(rphip_ctx* ctx) {
  // a small power-law-ish matrix on the host: user u has items (u * 7 + k * k) % I, k < 3 + u % 5
  auto make = [](int64_t U, int64_t I, std::vector<int32_t>& ip, std::vector<int32_t>& ix) {
    ip.assign(1, 0);
    ix.clear();
    for (int64_t u = 0; u < U; ++u) {
      std::vector<int32_t> row;
      for (int k = 0; k < 3 + (int)(u % 5); ++k) row.push_back((int32_t)((u * 7 + (int64_t)k * k * 13) % I));
      std::sort(row.begin(), row.end());
      row.erase(std::unique(row.begin(), row.end()), row.end());
      ix.insert(ix.end(), row.begin(), row.end());
      ip.push_back((int32_t)ix.size());
    }
  };
  for (int64_t I : {(int64_t)300, (int64_t)3000, (int64_t)30000}) {  // one column tile / one / two; EASE: one 128-block panel / blocked / (skipped)
    const int64_t U = 500;
    std::vector<int32_t> ip, ix;
    make(U, I, ip, ix);
    rphip_csr* X = nullptr;
    CHECK(rphip_csr_from_host(ctx, U, I, (int64_t)ix.size(), ip.data(), 0, ix.data(), &X) == RPHIP_OK && X);
    int64_t nu = 0, ni = 0, nz = 0;
    CHECK(rphip_csr_shape(X, &nu, &ni, &nz) == RPHIP_OK && nu == U && ni == I && nz == (int64_t)ix.size());
    std::vector<int32_t> deg((size_t)I);
    CHECK(rphip_csr_item_degrees(ctx, X, deg.data()) == RPHIP_OK);
    // ItemKNN: small K, K beyond the numbered merges (two tiles: the large-K route), every variant of the fit
    for (int32_t K : {5, 200, 20000}) {
      rphip_model* m = nullptr;
      CHECK(rphip_itemknn_fit(ctx, X, 0, I, K, &m) == RPHIP_OK && m);
      const int32_t Ke = (int32_t)std::min<int64_t>(K, I - 1);
      if ((int64_t)I * Ke < (1 << 22)) {
        std::vector<int32_t> idx((size_t)I * Ke), nnz((size_t)I);
        std::vector<double> val((size_t)I * Ke);
        CHECK(rphip_model_topk_download(ctx, m, idx.data(), val.data(), nnz.data()) == RPHIP_OK);
      }
      std::vector<int32_t> pi((size_t)U * 10), pn((size_t)U);
      std::vector<float> pv((size_t)U * 10);
      CHECK(rphip_predict_topk(ctx, m, X, 0, U, 10, 1, pi.data(), pv.data(), pn.data()) == RPHIP_OK);
      {  // Kp beyond the fused kernels: full scores per user chunk + the large-K selection
        std::vector<int32_t> bi((size_t)U * 4000), bn((size_t)U);  // (the caller's arrays hold n x Kp entries whatever the item count)
        std::vector<float> bv((size_t)U * 4000);
        CHECK(rphip_predict_topk(ctx, m, X, 0, U, 4000, 1, bi.data(), bv.data(), bn.data()) == RPHIP_OK);
      }
      std::vector<int64_t> rn((size_t)U);
      int64_t total = -1;
      CHECK(rphip_predict_full_count(ctx, m, X, 0, U, rn.data(), &total) == RPHIP_OK);
      CHECK(rphip_model_destroy(m) == RPHIP_OK);
    }
    {
      rphip_model* m = nullptr;
      CHECK(rphip_itemknn_fit_ex(ctx, X, 0, I, 50, RPHIP_SIM_CONDITIONAL_PROBABILITY, 0.5, 0, &m) == RPHIP_OK);
      CHECK(rphip_model_destroy(m) == RPHIP_OK);
      CHECK(rphip_itemknn_fit_ex(ctx, X, 0, I, 50, RPHIP_SIM_COSINE, 0.0, 1, &m) == RPHIP_OK);
      CHECK(rphip_model_destroy(m) == RPHIP_OK);
      CHECK(rphip_itemknn_fit_ex(ctx, X, 0, I, 20000, RPHIP_SIM_COSINE, 0.0, 1, &m) == RPHIP_OK);  // (30 000 items: three tiles, the large-K merge in row chunks)
      CHECK(rphip_model_destroy(m) == RPHIP_OK);
      std::vector<double> vals(ix.size(), 0.5);
      CHECK(rphip_csr_set_values(ctx, X, vals.data()) == RPHIP_OK);
      for (int32_t K : {50, 20000}) {
        CHECK(rphip_itemknn_fit_values(ctx, X, 0, I, K, 0, nullptr, &m) == RPHIP_OK);
        CHECK(rphip_model_destroy(m) == RPHIP_OK);
      }
      CHECK(rphip_csr_set_values(ctx, X, nullptr) == RPHIP_OK);
      CHECK(rphip_itemknn_fit_ex(ctx, X, 0, I, 50, RPHIP_SIM_CONDITIONAL_PROBABILITY, 2.0, 0, &m) == RPHIP_EINVAL);
    }
    if (I <= 3000) {
      // EASE: the blocked solve's host driver (rp_spd_inverse), K7, the dense predict's two passes, pruning, the cached-Gram sweep
      rphip_model* e = nullptr;
      CHECK(rphip_ease_fit(ctx, X, 20.0, 0.5, &e) == RPHIP_OK && e);
      std::vector<int32_t> pi((size_t)U * 10), pn((size_t)U);
      std::vector<float> pv((size_t)U * 10);
      CHECK(rphip_predict_topk(ctx, e, X, 0, U, 10, 1, pi.data(), pv.data(), pn.data()) == RPHIP_OK);
      std::vector<uint8_t> keep((size_t)I, 1);
      CHECK(rphip_predict_topk_filtered(ctx, e, X, 0, U, 10, 1, keep.data(), pi.data(), pv.data(), pn.data()) == RPHIP_OK);
      int64_t before = 0, after = 0;
      CHECK(rphip_model_dense_prune(ctx, e, I * 2, &before, &after) == RPHIP_OK);
      int64_t empty = -1;
      CHECK(rphip_model_empty_rows(ctx, e, &empty) == RPHIP_OK);
      CHECK(rphip_model_destroy(e) == RPHIP_OK);
      rphip_model* g = nullptr;
      CHECK(rphip_ease_gram_create(ctx, X, &g) == RPHIP_OK && g);
      for (double l2 : {10.0, 1000.0}) {
        CHECK(rphip_ease_fit_gram(ctx, X, g, l2, 0.0, &e) == RPHIP_OK);
        CHECK(rphip_model_destroy(e) == RPHIP_OK);
      }
      CHECK(rphip_model_destroy(g) == RPHIP_OK);
      CHECK(rphip_ease_fit(ctx, X, 0.0, 0.0, &e) != RPHIP_OK);  // l2 = 0 is refused
      // the distributed solve and its predict routes on a loop-back communicator: one rank of two, one of three
      for (int W : {2, 3}) {
        rphip_comm* comm = nullptr;
        CHECK(rphip_comm_loopback_create(ctx, W, W - 1, 100.0, 1.0, &comm) == RPHIP_OK && comm);
        int64_t mism = -1;
        double ms = 0.0;
        for (int op = 0; op < 5; ++op) (void)rphip_comm_selftest(ctx, comm, op, 1 << 12, &mism, &ms);
        rphip_model* local = nullptr;
        CHECK(rphip_ease_fit_dist(ctx, comm, X, 20.0, 0.0, I <= 300 ? 1 : 4, &local) == RPHIP_OK);
        int64_t st8[8];
        CHECK(rphip_ease_fit_dist_stats(ctx, st8) == RPHIP_OK);
        const int64_t panel = st8[5];
        rphip_model *full = nullptr, *rows8 = nullptr;
        CHECK(rphip_ease_model_allgather(ctx, comm, local, I, panel, &full) == RPHIP_OK && full);
        std::vector<int64_t> bounds((size_t)W + 1);
        for (int r = 0; r <= W; ++r) bounds[(size_t)r] = U * r / W;
        const int64_t n_me = bounds[(size_t)W] - bounds[(size_t)W - 1];
        void *oi = nullptr, *ov = nullptr, *on = nullptr;
        CHECK(rp_dev_alloc(ctx, (size_t)U * 10 * 4, &oi) == RPHIP_OK && rp_dev_alloc(ctx, (size_t)U * 10 * 4, &ov) == RPHIP_OK && rp_dev_alloc(ctx, (size_t)U * 4, &on) == RPHIP_OK);
        CHECK(rphip_predict_topk_dev(ctx, full, X, bounds[(size_t)W - 1], bounds[(size_t)W], 10, 1, (int32_t*)oi, (float*)ov, (int32_t*)on) == RPHIP_OK);
        const int rc8 = rphip_ease_model_allgather_i8(ctx, comm, local, I, panel, &rows8);
        if (rc8 == RPHIP_OK) {
          const int rco = rphip_predict_topk_owner(ctx, comm, local, rows8, X, bounds.data(), 10, 0, (int32_t*)oi, (float*)ov, (int32_t*)on);
          CHECK(rco == RPHIP_OK || rco == RPHIP_ELIMIT);  // (ELIMIT: too few items for the first pass - refused before the first collective)
          CHECK(rphip_model_destroy(rows8) == RPHIP_OK);
        } else {
          CHECK(rc8 == RPHIP_ESTATE);
        }
        (void)n_me;
        int64_t ub_o = -1, nu_o = -1;
        CHECK(rphip_predict_topk_dist(ctx, comm, local, X, 10, 0, &ub_o, &nu_o, (int32_t*)oi, (float*)ov, (int32_t*)on) == RPHIP_OK && nu_o >= 0);
        rp_dev_free(ctx, oi);
        rp_dev_free(ctx, ov);
        rp_dev_free(ctx, on);
        CHECK(rphip_model_destroy(full) == RPHIP_OK);
        if (local) CHECK(rphip_model_destroy(local) == RPHIP_OK);
        CHECK(rphip_comm_destroy(comm) == RPHIP_OK);
      }
    }
    CHECK(rphip_csr_destroy(X) == RPHIP_OK);
  }
  // COO ingestion (the own radix sort's pass plumbing) and an out-of-range pair
  {
    std::vector<int64_t> u = {0, 3, 3, 1, 0}, i = {2, 1, 1, 0, 2};
    rphip_csr* X = nullptr;
    CHECK(rphip_csr_from_coo_host(ctx, 5, 4, 5, u.data(), i.data(), &X) == RPHIP_OK);
    CHECK(rphip_csr_destroy(X) == RPHIP_OK);
  }
  return 0;
}

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
  if (host_drivers(ctx)) return 1;
  CHECK(rphip_ctx_trim(ctx) == RPHIP_OK);
  CHECK(rphip_ctx_destroy(ctx) == RPHIP_OK && rphip_ctx_destroy(nullptr) == RPHIP_OK);
  CHECK(stub_live_blocks() == 0);  // nothing of the "device" left behind
  printf("host asan driver: ok (%ld kernel launches planned by the host code, none run)\n", stub_launches());
  return 0;
}

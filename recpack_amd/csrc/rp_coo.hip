// CSR ingestion from raw (user id, item id) pairs on the device.
//
// Replaces the construction behind InteractionMatrix.values (recpack/matrix/interaction_matrix.py:202-217:
// csr_matrix((ones, (uid, iid)), shape) - duplicates summed into counts) followed by to_csr_matrix(binary=True)
// (recpack/matrix/util.py:27-48): the hot path only needs the binary pattern, i.e. the DISTINCT pairs, rows sorted by
// item.  Pairs may come in any order and may repeat.
//
//   k_coo_keys      key = uid * n_items + iid (64 bit), range check
//   radix sort      rocPRIM (a plain library sort; not part of the hot path), only the bits the keys can have
//   k_coo_heads     flag the first occurrence of every key; exclusive scan of the flags = position in the CSR
//   k_coo_emit      indices[pos] = key % n_items, row pointer entries of the users that start (gaps = empty users)
// then K0 (rp_csr_prepare) as for any other input.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "rp_common.h"

__global__ void k_coo_keys(const int64_t* __restrict__ uid, const int64_t* __restrict__ iid, int64_t n, int64_t n_users, int64_t n_items,
                           unsigned long long* __restrict__ key, int32_t* __restrict__ err) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = uid[k], i = iid[k];
    if (u < 0 || u >= n_users || i < 0 || i >= n_items) {
      *err = 1;
      key[k] = 0ull;
    } else {
      key[k] = (unsigned long long)u * (unsigned long long)n_items + (unsigned long long)i;
    }
  }
}

__global__ void k_coo_heads(const unsigned long long* __restrict__ key, int64_t n, int32_t* __restrict__ head) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
    head[k] = (k == 0 || key[k] != key[k - 1]) ? 1 : 0;
}

// pos[k] = exclusive scan of head; pos[n] = number of distinct pairs
__global__ void k_coo_emit(const unsigned long long* __restrict__ key, const int32_t* __restrict__ head, const int32_t* __restrict__ pos,
                           int64_t n, int64_t n_users, int64_t n_items, int32_t* __restrict__ indptr, int32_t* __restrict__ indices) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    if (!head[k]) continue;
    const int p = pos[k];
    const long long u = (long long)(key[k] / (unsigned long long)n_items);
    indices[p] = (int32_t)(key[k] % (unsigned long long)n_items);
    // this entry opens the rows of every user after the previous distinct pair's user up to u
    long long prev_u = -1;
    if (k > 0) prev_u = (long long)(key[k - 1] / (unsigned long long)n_items);  // key[k-1] is the previous distinct key or its copy
    for (long long v = prev_u + 1; v <= u; ++v) indptr[v] = p;
  }
  // users after the last pair, and indptr[n_users]
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long last_u = n > 0 ? (long long)(key[n - 1] / (unsigned long long)n_items) : -1;
    const int total = pos[n];
    for (long long v = last_u + 1; v <= n_users; ++v) indptr[v] = total;
  }
}

static int rp_coo_build(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t n_pairs, const int64_t* uid_dev, const int64_t* iid_dev,
                        rphip_csr** out) {
  if (n_users < 0 || n_items < 0 || n_pairs < 0 || n_users >= (1ll << 31) - 1 || n_items >= (1ll << 31) - 1 || n_pairs >= (1ll << 31) - 1)
    RP_FAIL(RPHIP_ELIMIT, "shape (%lld, %lld) with %lld pairs exceeds the int32 index range", (long long)n_users, (long long)n_items,
            (long long)n_pairs);
  hipStream_t st = ctx->stream;
  const int64_t n = n_pairs;
  int32_t* indptr = nullptr;
  int32_t* indices = nullptr;
  RP_HIP(hipMalloc((void**)&indptr, (size_t)(n_users + 1) * 4));
  if (hipMalloc((void**)&indices, (size_t)std::max<int64_t>(n, 1) * 4) != hipSuccess) {
    hipFree(indptr);
    RP_FAIL(RPHIP_ENOMEM, "hipMalloc of the CSR indices failed");
  }
  auto fail = [&](int rc) {
    hipFree(indptr);
    hipFree(indices);
    return rc;
  };
  int32_t total = 0;
  if (n == 0) {
    if (hipMemsetAsync(indptr, 0, (size_t)(n_users + 1) * 4, st) != hipSuccess) return fail(RPHIP_EHIP);
  } else {
    // bits the keys can have
    int bits = 1;
    while (bits < 64 && ((unsigned long long)n_users * (unsigned long long)n_items - 1ull) >> bits) ++bits;
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_keys(nullptr, tmp_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)n, 0, bits, st) != hipSuccess)
      return fail(RPHIP_EHIP);
    char* sc = nullptr;
    const size_t kb = ((size_t)n * 8 + 255) & ~(size_t)255, hb = ((size_t)(n + 1) * 4 + 255) & ~(size_t)255;
    int rc = rp_scratch(ctx, 0, 2 * kb + 2 * hb + tmp_bytes + 512, (void**)&sc);
    if (rc != RPHIP_OK) return fail(rc);
    unsigned long long* key = (unsigned long long*)sc;
    unsigned long long* sorted = (unsigned long long*)(sc + kb);
    int32_t* head = (int32_t*)(sc + 2 * kb);
    int32_t* pos = (int32_t*)(sc + 2 * kb + hb);
    int32_t* err = (int32_t*)(sc + 2 * kb + 2 * hb);
    void* tmp = sc + 2 * kb + 2 * hb + 256;
    hipEvent_t ev;
    rp_time_begin(ctx, &ev);
    hipMemsetAsync(err, 0, 4, st);
    const int grid = (int)std::min<int64_t>((int64_t)ctx->n_cu * 16, rp_cdiv(n, 256));
    hipLaunchKernelGGL(k_coo_keys, dim3(grid), dim3(256), 0, st, uid_dev, iid_dev, n, n_users, n_items, key, err);
    if (rocprim::radix_sort_keys(tmp, tmp_bytes, key, sorted, (size_t)n, 0, bits, st) != hipSuccess) return fail(RPHIP_EHIP);
    hipLaunchKernelGGL(k_coo_heads, dim3(grid), dim3(256), 0, st, sorted, n, head);
    rc = rp_exclusive_scan_i32(ctx, head, pos, n);
    if (rc != RPHIP_OK) return fail(rc);
    hipLaunchKernelGGL(k_coo_emit, dim3(grid), dim3(256), 0, st, sorted, head, pos, n, n_users, n_items, indptr, indices);
    int32_t h[2] = {0, 0};
    if (hipMemcpyAsync(&h[0], err, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&h[1], pos + n, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return fail(RPHIP_EHIP);
    rp_time_end(ctx, ev, RPHIP_T_CSR_TRANSPOSE, 16.0 * (double)n);
    if (h[0]) {
      rp_set_error("a user or item id lies outside the shape (%lld, %lld)", (long long)n_users, (long long)n_items);
      return fail(RPHIP_EINVAL);
    }
    total = h[1];
  }
  rphip_csr* m = new rphip_csr();
  m->ctx = ctx;
  m->n_users = n_users;
  m->n_items = n_items;
  m->nnz = total;
  m->owns_x = true;
  m->x_ptr = indptr;
  m->x_idx = indices;
  int r = rp_csr_prepare(ctx, m);
  if (r != RPHIP_OK) {
    rphip_csr_destroy(m);
    return r;
  }
  *out = m;
  return RPHIP_OK;
}

extern "C" int rphip_csr_from_coo_device(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t n_pairs, const int64_t* uid_dev,
                                         const int64_t* iid_dev, rphip_csr** out) {
  RP_ENTER(ctx);
  if (!ctx || !out || (n_pairs > 0 && (!uid_dev || !iid_dev))) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  RP_TRY(rp_set_device(ctx));
  return rp_coo_build(ctx, n_users, n_items, n_pairs, uid_dev, iid_dev, out);
}

extern "C" int rphip_csr_from_coo_host(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t n_pairs, const int64_t* uid_host,
                                       const int64_t* iid_host, rphip_csr** out) {
  RP_ENTER(ctx);
  if (!ctx || !out || (n_pairs > 0 && (!uid_host || !iid_host))) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  RP_TRY(rp_set_device(ctx));
  int64_t* d = nullptr;
  const size_t nb = (size_t)std::max<int64_t>(n_pairs, 1) * 8;
  RP_HIP(hipMalloc((void**)&d, 2 * nb));
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  hipError_t e = hipSuccess;
  if (n_pairs > 0) {
    e = hipMemcpyAsync(d, uid_host, (size_t)n_pairs * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync((char*)d + nb, iid_host, (size_t)n_pairs * 8, hipMemcpyHostToDevice, ctx->stream);
  }
  rp_time_end(ctx, ev, RPHIP_T_H2D, 16.0 * (double)n_pairs);
  int rc = (e == hipSuccess) ? rp_coo_build(ctx, n_users, n_items, n_pairs, d, (const int64_t*)((char*)d + nb), out) : RPHIP_EHIP;
  hipStreamSynchronize(ctx->stream);
  hipFree(d);
  return rc;
}

// the CSR arrays of a device matrix (int32 indptr[n_users + 1], int32 indices[nnz]) to the host
extern "C" int rphip_csr_download(rphip_ctx* ctx, const rphip_csr* m, int32_t* indptr_host, int32_t* indices_host) {
  RP_ENTER(ctx);
  if (!ctx || !m || !indptr_host || (m->nnz > 0 && !indices_host)) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_TRY(rp_set_device(ctx));
  RP_HIP(hipMemcpyAsync(indptr_host, m->x_ptr, (size_t)(m->n_users + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (m->nnz > 0) RP_HIP(hipMemcpyAsync(indices_host, m->x_idx, (size_t)m->nnz * 4, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

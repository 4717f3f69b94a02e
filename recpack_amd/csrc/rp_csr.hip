// K0: the device-resident interaction matrix: X (CSR by user), X^T (CSR by item), item degrees, column tiles.
//
// Replaces InteractionMatrix.values + to_csr_matrix(binary=True) (recpack/matrix/interaction_matrix.py:202-217,
// recpack/matrix/util.py:27-48) as far as the hot path needs them: only the pattern (indptr, indices) is used.
// X^T is what scipy builds implicitly for `X.T @ X` (nearest_neighbour.py:80 via sklearn, ease.py:81).
//
//   k_rank_hist_ranges   degrees AND the position of every entry inside its item row, one pass over nnz per item
//                 range (LDS counters per block, one global atomic per (block, item))
//   scan          xt_ptr = exclusive_scan(deg)
//   k_scatter     xt_idx[xt_ptr[j] + rank[k]] = u          8 lanes per user row, coalesced 32-B segments; the same
//                 pass checks the column order and (n_items > one LDS tile) writes per user the first position
//                 with column >= t*W
//   k_inv_deg     inv_deg[j] = fl(1/c_j)
#include <algorithm>

#include <cmath>

#include "rp_common.h"

// Degrees and in-row ranks without hot global atomics: the item axis is cut into ranges that fit LDS, a block
// (chunk of nnz, range) counts its entries with returning LDS atomics (= rank inside the block), reserves a
// span per item with ONE global atomic per (block, item), then adds the span's base to the stored ranks.
// A popular item (84 % of the users at 1M x 100K) costs n_chunks global atomics instead of one per interaction.
// Only items in [item_lo, item_hi) are counted (a shard's K0: the degrees of the other items come from an all-reduce).
__global__ void __launch_bounds__(1024) k_rank_hist_ranges(const int32_t* __restrict__ idx, long long nnz, long long chunk, int n_items,
                                                          int range_w, int32_t* __restrict__ deg, int32_t* __restrict__ rank,
                                                          int32_t* __restrict__ err, int item_lo, int item_hi) {
  extern __shared__ __attribute__((aligned(16))) char smem_h[];
  int* cnt = reinterpret_cast<int*>(smem_h);
  const int lo = item_lo + blockIdx.y * range_w, hi = min(item_hi, lo + range_w);
  const long long k0 = (long long)blockIdx.x * chunk, k1 = min(nnz, k0 + chunk);
  for (int w = threadIdx.x; w < range_w; w += 1024) cnt[w] = 0;
  __syncthreads();
  for (long long k = k0 + threadIdx.x; k < k1; k += 1024) {
    const int j = idx[k];
    if (j >= lo && j < hi) rank[k] = atomicAdd(&cnt[j - lo], 1);
    if (blockIdx.y == 0 && (unsigned)j >= (unsigned)n_items) *err = 1;
  }
  __syncthreads();
  for (int w = threadIdx.x; w < hi - lo; w += 1024) {
    const int c = cnt[w];
    cnt[w] = c ? atomicAdd(&deg[lo + w], c) : 0;
  }
  __syncthreads();
  for (long long k = k0 + threadIdx.x; k < k1; k += 1024) {
    const int j = idx[k];
    if (j >= lo && j < hi) rank[k] += cnt[j - lo];
  }
}

// One pass over X with 8 lanes per user row (coalesced 32-byte segments):
//   * X^T: xt_idx[xt_ptr[j] + rank[k]] = u
//   * are the column indices of every row strictly increasing?  (err[1] = 1 if not; with several column tiles err[0] = 3)
//   * n_tiles > 1: tile_ptr[u][t] = first position of row u with column >= t * tile_w - an entry whose tile differs from
//     its predecessor's opens the tiles in between; the row's first entry opens tiles 0.., its last closes the rest
//   * P = sum_u d_u^2
__global__ void k_scatter(const int32_t* __restrict__ x_ptr, const int32_t* __restrict__ x_idx, int n_users,
                          const int32_t* __restrict__ xt_ptr, const int32_t* __restrict__ rank, int32_t* __restrict__ xt_idx,
                          unsigned long long* __restrict__ p_sum, int n_items, int n_tiles, int tile_w, int32_t* __restrict__ tile_ptr,
                          int32_t* __restrict__ err, int xt_lo, int xt_hi, uint16_t* __restrict__ x_rel) {
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, sub = threadIdx.x & 7;
  const int ng = (gridDim.x * blockDim.x) >> 3;
  const float inv_w = 1.0f / (float)tile_w;
  auto tile_of = [&](int j) {
    int t = (int)((float)j * inv_w);
    while ((long long)(t + 1) * tile_w <= j) ++t;
    while ((long long)t * tile_w > j) --t;
    return t;
  };
  unsigned long long local = 0;
  for (int u = g; u < n_users; u += ng) {
    const int s = x_ptr[u], e = x_ptr[u + 1];
    int32_t* tp = (n_tiles > 1) ? tile_ptr + (int64_t)u * (n_tiles + 1) : nullptr;
    if (sub == 0) {
      local += (unsigned long long)(e - s) * (unsigned long long)(e - s);
      if (tp && e == s)
        for (int t = 0; t <= n_tiles; ++t) tp[t] = s;
    }
    for (int k = s + sub; k < e; k += 8) {
      const int j = x_idx[k];
      const int prev = (k > s) ? x_idx[k - 1] : -1;
      if (j >= xt_lo && j < xt_hi) xt_idx[xt_ptr[j] + rank[k]] = u;  // (a shard only fills the item rows it will fit)
      if (j <= prev) {
        err[1] = 1;
        if (tp) err[0] = 3;
      }
      const bool in_range = (unsigned)j < (unsigned)n_items;
      const int tj0 = (in_range && n_tiles > 1) ? tile_of(j) : 0;
      x_rel[k] = (uint16_t)(in_range ? j - tj0 * tile_w : 0);  // (the column inside its tile)
      if (tp && in_range) {
        const int tj = tj0;
        const int tprev = (k > s && (unsigned)prev < (unsigned)n_items) ? tile_of(prev) : -1;
        for (int t = tprev + 1; t <= tj; ++t) tp[t] = k;
        if (k == e - 1)
          for (int t = tj + 1; t <= n_tiles; ++t) tp[t] = e;
      }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) local += __shfl_xor(local, d);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(p_sum, local);
}

__global__ void k_inv_deg(const int32_t* __restrict__ deg, double* __restrict__ inv, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int d = deg[i];
    inv[i] = d > 0 ? 1.0 / (double)d : 0.0;
  }
}

__global__ void k_check_ptr(const int32_t* __restrict__ x_ptr, int n_users, int64_t nnz, int32_t* __restrict__ err) {
  int dmax = 0;
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u <= n_users; u += gridDim.x * blockDim.x) {
    const int a = x_ptr[u];
    if (u == 0 && a != 0) *err = 2;
    if (u == n_users && a != (int)nnz) *err = 2;
    if (u < n_users) {
      const int d = x_ptr[u + 1] - a;
      if (d < 0) *err = 2;
      dmax = max(dmax, d);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, __shfl_xor(dmax, o));
  if ((threadIdx.x & 63) == 0 && dmax > 0) atomicMax(&err[4], dmax);  // longest history
}

int rp_csr_prepare(rphip_ctx* ctx, rphip_csr* m) {
  hipStream_t st = ctx->stream;
  const int64_t U = m->n_users, I = m->n_items, nnz = m->nnz;
  if (nnz < 8) {
    // the fit kernels read the column indices in 8-entry pieces (the last piece of the matrix is the one that ENDS at nnz): a
    // matrix with fewer entries gets a padded copy of its indices
    int32_t* pad = nullptr;
    RP_TRY(rp_dev_alloc(ctx, 8 * sizeof(int32_t), (void**)&pad));
    RP_HIP(hipMemsetAsync(pad, 0, 8 * sizeof(int32_t), st));
    if (nnz > 0) RP_HIP(hipMemcpyAsync(pad, m->x_idx, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    RP_HIP(hipStreamSynchronize(st));
    if (m->owns_x && m->x_idx) hipFree((void*)m->x_idx);
    m->x_idx = pad;
    m->x_idx_padded = true;
  }
  m->n_tiles = (int)std::max<int64_t>(1, rp_cdiv(I, RP_TILE_MAX));
  m->tile_w = (int)(rp_cdiv(rp_cdiv(std::max<int64_t>(I, 1), m->n_tiles), 64) * 64);
  RP_TRY(rp_dev_alloc(ctx, (size_t)(I + 1) * sizeof(int32_t), (void**)&m->xt_ptr));
  RP_TRY(rp_dev_alloc(ctx, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int32_t), (void**)&m->xt_idx));
  RP_TRY(rp_dev_alloc(ctx, (size_t)std::max<int64_t>(I, 1) * sizeof(int32_t), (void**)&m->deg));
  RP_TRY(rp_dev_alloc(ctx, (size_t)std::max<int64_t>(I, 1) * sizeof(double), (void**)&m->inv_deg));
  char* sc = nullptr;
  const size_t rank_bytes = ((size_t)std::max<int64_t>(nnz, 1) * 4 + 15) & ~(size_t)15;  // keeps p_sum 8-B aligned
  RP_TRY(rp_scratch(ctx, 0, rank_bytes + 64, (void**)&sc));
  int32_t* rank = (int32_t*)sc;
  int32_t* err = (int32_t*)(sc + rank_bytes);
  unsigned long long* p_sum = (unsigned long long*)(err + 2);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  RP_HIP(hipMemsetAsync(m->deg, 0, (size_t)std::max<int64_t>(I, 1) * sizeof(int32_t), st));
  RP_HIP(hipMemsetAsync(err, 0, 64, st));
  const int bt = 256;
  const int grid_u = (int)std::min<int64_t>((int64_t)ctx->n_cu * 16, std::max<int64_t>(1, rp_cdiv(U + 1, bt)));
  hipLaunchKernelGGL(k_check_ptr, dim3(grid_u), dim3(bt), 0, st, m->x_ptr, (int)U, nnz, err);
  int32_t h_chk[5] = {0, 0, 0, 0, 0};
  RP_HIP(hipMemcpyAsync(h_chk, err, sizeof(h_chk), hipMemcpyDeviceToHost, st));
  RP_HIP(hipStreamSynchronize(st));
  const int32_t h_err = h_chk[0];
  m->max_user_deg = h_chk[4];
  if (h_err) RP_FAIL(RPHIP_EINVAL, "indptr is not a valid row pointer array (not monotone, or indptr[0] != 0, or indptr[-1] != nnz)");
  // a shard with deg_partial counts (and ranks) only the item rows it will fit: 1 / world of the LDS and global atomics
  const int64_t i_lo = m->deg_partial ? std::max<int64_t>(0, m->xt_row_begin) : 0;
  const int64_t i_hi = m->deg_partial ? (m->xt_row_end < 0 ? I : std::min<int64_t>(I, m->xt_row_end)) : I;
  if (nnz > 0 && I > 0) {
    const int64_t span = std::max<int64_t>(1, i_hi - i_lo);
    const int n_ranges = (int)rp_cdiv(span, 32768);
    const int range_w = (int)rp_cdiv(span, n_ranges);
    const long long target_blocks = std::max<long long>(1, (long long)ctx->n_cu * 3 / n_ranges);
    const long long chunk = std::max<long long>(4096, rp_cdiv(nnz, target_blocks));
    const size_t hl = (size_t)range_w * 4;
    RP_HIP(hipFuncSetAttribute((const void*)k_rank_hist_ranges, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
    hipLaunchKernelGGL(k_rank_hist_ranges, dim3((unsigned)rp_cdiv(nnz, chunk), n_ranges), dim3(1024), hl, st, m->x_idx, (long long)nnz, chunk,
                       (int)I, range_w, m->deg, rank, err, (int)i_lo, (int)i_hi);
  }
  RP_TRY(rp_exclusive_scan_i32(ctx, m->deg, m->xt_ptr, I));
  if (m->n_tiles > 1) RP_TRY(rp_dev_alloc(ctx, (size_t)std::max<int64_t>(U, 1) * (m->n_tiles + 1) * sizeof(int32_t), (void**)&m->tile_ptr));
  {  // tile-relative columns, padded so that a chunk of eight read at any entry stays inside
    const size_t n_rel = (((size_t)nnz + 15) & ~(size_t)15) + 16;
    RP_TRY(rp_dev_alloc(ctx, n_rel * sizeof(uint16_t), (void**)&m->x_rel));
    RP_HIP(hipMemsetAsync(m->x_rel, 0, n_rel * sizeof(uint16_t), st));
  }
  if (U > 0) {  // (rows without entries still need their tile pointers)
    const int grid_s = (int)std::min<int64_t>((int64_t)ctx->n_cu * 16, std::max<int64_t>(1, rp_cdiv(U * 8, bt)));
    hipLaunchKernelGGL(k_scatter, dim3(grid_s), dim3(bt), 0, st, m->x_ptr, m->x_idx, (int)U, m->xt_ptr, rank, m->xt_idx, p_sum,
                       (int)I, m->n_tiles, m->tile_w, m->tile_ptr, err, (int)std::max<int64_t>(0, m->xt_row_begin),
                       (int)(m->xt_row_end < 0 ? I : std::min<int64_t>(I, m->xt_row_end)), m->x_rel);
  }
  if (!m->deg_partial)  // (a shard's degrees are completed by rphip_csr_finish_degrees after the all-reduce)
    hipLaunchKernelGGL(k_inv_deg, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, rp_cdiv(I, bt)))), dim3(bt), 0, st,
                       m->deg, m->inv_deg, (int)I);
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, RPHIP_T_CSR_TRANSPOSE, 12.0 * (double)nnz + 4.0 * (double)(U + I + 2) + 4.0 * (double)(I + 1));
  struct {
    int32_t err[2];
    unsigned long long p;
  } h;
  RP_HIP(hipMemcpyAsync(&h, err, sizeof(h), hipMemcpyDeviceToHost, st));
  RP_HIP(hipStreamSynchronize(st));
  if (h.err[0] == 1) RP_FAIL(RPHIP_EINVAL, "a column index is outside [0, n_items)");
  if (h.err[0] == 3)
    RP_FAIL(RPHIP_EUNSORTED, "n_items=%lld needs %d column tiles: column indices of every row must be strictly increasing",
            (long long)I, m->n_tiles);
  m->n_products = (int64_t)h.p;
  m->sorted_rows = (m->n_tiles > 1) || (h.err[1] == 0);  // several tiles: enforced above; one tile: checked
  return RPHIP_OK;
}

static int rp_csr_check_shape(int64_t U, int64_t I, int64_t nnz) {
  if (U < 0 || I < 0 || nnz < 0) RP_FAIL(RPHIP_EINVAL, "negative shape");
  if (U >= (1ll << 31) - 1 || I >= (1ll << 31) - 1 || nnz >= (1ll << 31) - 1)
    RP_FAIL(RPHIP_ELIMIT, "n_users, n_items and nnz must be below 2^31-1 (got %lld, %lld, %lld)", (long long)U, (long long)I,
            (long long)nnz);
  return RPHIP_OK;
}

extern "C" int rphip_csr_from_host(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t nnz, const void* indptr_host,
                                   int indptr_is_int64, const int32_t* indices_host, rphip_csr** out) {
  if (!ctx || !out || !indptr_host || (nnz > 0 && !indices_host)) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  RP_TRY(rp_csr_check_shape(n_users, n_items, nnz));
  RP_TRY(rp_set_device(ctx));
  rphip_csr* m = new rphip_csr();
  m->ctx = ctx;
  m->n_users = n_users;
  m->n_items = n_items;
  m->nnz = nnz;
  m->owns_x = true;
  int32_t* dp = nullptr;
  int32_t* di = nullptr;
  std::vector<int32_t> tmp;
  const int32_t* hp = (const int32_t*)indptr_host;
  if (indptr_is_int64) {
    const int64_t* p64 = (const int64_t*)indptr_host;
    tmp.resize((size_t)n_users + 1);
    for (int64_t i = 0; i <= n_users; ++i) {
      if (p64[i] < 0 || p64[i] > nnz) {
        delete m;
        RP_FAIL(RPHIP_EINVAL, "indptr[%lld]=%lld outside [0, nnz]", (long long)i, (long long)p64[i]);
      }
      tmp[(size_t)i] = (int32_t)p64[i];
    }
    hp = tmp.data();
  }
  hipError_t e1 = hipMalloc((void**)&dp, (size_t)(n_users + 1) * sizeof(int32_t));
  hipError_t e2 = hipMalloc((void**)&di, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int32_t));
  m->x_ptr = dp;
  m->x_idx = di;
  if (e1 != hipSuccess || e2 != hipSuccess) {
    rphip_csr_destroy(m);
    RP_FAIL(RPHIP_ENOMEM, "hipMalloc of the CSR arrays failed");
  }
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  e1 = hipMemcpyAsync(dp, hp, (size_t)(n_users + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
  e2 = nnz > 0 ? hipMemcpyAsync(di, indices_host, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
  rp_time_end(ctx, ev, RPHIP_T_H2D, 4.0 * (double)(n_users + 1 + nnz));
  hipError_t e3 = hipStreamSynchronize(ctx->stream);  // tmp may go out of scope
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    rphip_csr_destroy(m);
    RP_FAIL(RPHIP_EHIP, "host to device copy of the CSR arrays failed");
  }
  int r = rp_csr_prepare(ctx, m);
  if (r != RPHIP_OK) {
    rphip_csr_destroy(m);
    return r;
  }
  *out = m;
  return RPHIP_OK;
}

extern "C" int rphip_csr_from_device(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t nnz, const int32_t* indptr_dev,
                                     const int32_t* indices_dev, rphip_csr** out) {
  return rphip_csr_from_device_rows(ctx, n_users, n_items, nnz, indptr_dev, indices_dev, 0, -1, out);
}

extern "C" int rphip_csr_from_device_rows(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t nnz, const int32_t* indptr_dev,
                                          const int32_t* indices_dev, int64_t fit_row_begin, int64_t fit_row_end, rphip_csr** out) {
  if (!ctx || !out || !indptr_dev || (nnz > 0 && !indices_dev)) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  RP_TRY(rp_csr_check_shape(n_users, n_items, nnz));
  if (fit_row_end >= 0 && (fit_row_begin < 0 || fit_row_begin > fit_row_end || fit_row_end > n_items))
    RP_FAIL(RPHIP_EINVAL, "fit row range [%lld,%lld) outside 0..%lld", (long long)fit_row_begin, (long long)fit_row_end, (long long)n_items);
  RP_TRY(rp_set_device(ctx));
  rphip_csr* m = new rphip_csr();
  m->xt_row_begin = fit_row_end < 0 ? 0 : fit_row_begin;
  m->xt_row_end = fit_row_end;
  m->ctx = ctx;
  m->n_users = n_users;
  m->n_items = n_items;
  m->nnz = nnz;
  m->owns_x = false;
  m->x_ptr = indptr_dev;
  m->x_idx = indices_dev;
  int r = rp_csr_prepare(ctx, m);
  if (r != RPHIP_OK) {
    rphip_csr_destroy(m);
    return r;
  }
  *out = m;
  return RPHIP_OK;
}

// A shard's matrix for the multi-GPU fit (north_star: "all-reduce over xGMI"): like
// rphip_csr_from_device_rows, but the degree histogram and the in-row ranks are computed for the item rows [fit_row_begin,
// fit_row_end) only - the shard's share of K0's atomics.  The handle is NOT usable for a fit until the caller has summed the
// degree arrays of all shards (rphip_csr_degrees_device -> one all-reduce of n_items int32) and called rphip_csr_finish_degrees.
extern "C" int rphip_csr_from_device_shard(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t nnz, const int32_t* indptr_dev,
                                           const int32_t* indices_dev, int64_t fit_row_begin, int64_t fit_row_end, rphip_csr** out) {
  if (!ctx || !out || !indptr_dev || (nnz > 0 && !indices_dev)) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  RP_TRY(rp_csr_check_shape(n_users, n_items, nnz));
  if (fit_row_begin < 0 || fit_row_begin > fit_row_end || fit_row_end > n_items)
    RP_FAIL(RPHIP_EINVAL, "fit row range [%lld,%lld) outside 0..%lld", (long long)fit_row_begin, (long long)fit_row_end, (long long)n_items);
  RP_TRY(rp_set_device(ctx));
  rphip_csr* m = new rphip_csr();
  m->xt_row_begin = fit_row_begin;
  m->xt_row_end = fit_row_end;
  m->deg_partial = true;
  m->ctx = ctx;
  m->n_users = n_users;
  m->n_items = n_items;
  m->nnz = nnz;
  m->owns_x = false;
  m->x_ptr = indptr_dev;
  m->x_idx = indices_dev;
  int r = rp_csr_prepare(ctx, m);
  if (r != RPHIP_OK) {
    rphip_csr_destroy(m);
    return r;
  }
  *out = m;
  return RPHIP_OK;
}

extern "C" int rphip_csr_degrees_device(rphip_csr* m, int32_t** deg_dev) {
  if (!m || !deg_dev) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *deg_dev = m->deg;
  return RPHIP_OK;
}

extern "C" int rphip_csr_finish_degrees(rphip_ctx* ctx, rphip_csr* m) {
  if (!ctx || !m) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_TRY(rp_set_device(ctx));
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  hipLaunchKernelGGL(k_inv_deg, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, rp_cdiv(m->n_items, 256)))), dim3(256), 0, ctx->stream,
                     m->deg, m->inv_deg, (int)m->n_items);
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, RPHIP_T_CSR_TRANSPOSE, 12.0 * (double)m->n_items);
  m->deg_partial = false;
  return RPHIP_OK;
}

extern "C" int rphip_csr_destroy(rphip_csr* m) {
  if (!m) return RPHIP_OK;
  if (m->ctx) hipSetDevice(m->ctx->device);
  // x_ptr / x_idx of a *_from_device matrix belong to the caller, who may release them the moment this returns: asynchronous
  // launches (rphip_predict_topk_dev ...) may still be reading them on the context's stream
  if (m->ctx && !m->owns_x) hipStreamSynchronize(m->ctx->stream);
  if (m->x_idx_padded) {
    rp_dev_free(m->ctx, (void*)m->x_idx);
    m->x_idx = nullptr;
  }
  if (m->owns_x) {  // (hipFree waits for the device; cached blocks carry an event instead)
    if (m->x_ptr) hipFree((void*)m->x_ptr);
    if (m->x_idx) hipFree((void*)m->x_idx);
  }
  rp_dev_free(m->ctx, m->xt_ptr);
  rp_dev_free(m->ctx, m->xt_idx);
  rp_dev_free(m->ctx, m->deg);
  rp_dev_free(m->ctx, m->inv_deg);
  rp_dev_free(m->ctx, m->tile_ptr);
  rp_dev_free(m->ctx, m->x_rel);
  rp_dev_free(m->ctx, m->row_cost);
  rp_dev_free(m->ctx, m->x_val);
  delete m;
  return RPHIP_OK;
}

// Values for the stored entries (NULL: back to binary).  Scanned on the host for the largest magnitude (predict's
// fixed-point scale) and for negative entries (the signed selection rules of the weighted fit).
extern "C" int rphip_csr_set_values(rphip_ctx* ctx, rphip_csr* m, const double* values_host) {
  if (!ctx || !m) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_TRY(rp_set_device(ctx));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  rp_dev_free(ctx, m->x_val);
  m->x_val = nullptr;
  m->val_max_abs = 0.0;
  m->val_neg = false;
  if (!values_host) return RPHIP_OK;
  double mx = 0.0;
  bool neg = false;
  for (int64_t k = 0; k < m->nnz; ++k) {
    const double v = values_host[k];
    if (!std::isfinite(v)) RP_FAIL(RPHIP_EINVAL, "value %lld is not finite", (long long)k);
    mx = std::max(mx, std::fabs(v));
    neg |= v < 0.0;
  }
  const size_t bytes = (size_t)std::max<int64_t>(1, m->nnz) * sizeof(double);
  RP_TRY(rp_dev_alloc(ctx, bytes, (void**)&m->x_val));
  if (m->nnz > 0) RP_HIP(hipMemcpyAsync(m->x_val, values_host, (size_t)m->nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  m->val_max_abs = mx;
  m->val_neg = neg;
  return RPHIP_OK;
}

extern "C" int rphip_csr_shape(const rphip_csr* m, int64_t* n_users, int64_t* n_items, int64_t* nnz) {
  if (!m) RP_FAIL(RPHIP_EINVAL, "matrix is NULL");
  if (n_users) *n_users = m->n_users;
  if (n_items) *n_items = m->n_items;
  if (nnz) *nnz = m->nnz;
  return RPHIP_OK;
}

extern "C" int rphip_csr_item_degrees(rphip_ctx* ctx, const rphip_csr* m, int32_t* deg_host) {
  if (!ctx || !m || !deg_host) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_TRY(rp_set_device(ctx));
  if (m->n_items > 0) RP_HIP(hipMemcpyAsync(deg_host, m->deg, (size_t)m->n_items * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

extern "C" int rphip_csr_num_products(rphip_ctx* ctx, const rphip_csr* m, int64_t* p_out) {
  if (!ctx || !m || !p_out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *p_out = m->n_products;
  return RPHIP_OK;
}

// CSR ingestion from raw (user id, item id) pairs on the device.
//
// Replaces the construction behind InteractionMatrix.values (recpack/matrix/interaction_matrix.py:202-217:
// csr_matrix((ones, (uid, iid)), shape) - duplicates summed into counts) followed by to_csr_matrix(binary=True)
// (recpack/matrix/util.py:27-48): the hot path only needs the binary pattern, i.e. the DISTINCT pairs, rows sorted by
// item.  Pairs may come in any order and may repeat.
//
//   k_coo_keys      key = uid * n_items + iid (64 bit), range check
//   radix sort      own LSD radix sort (round 6: the last library kernel of the tree is gone), 4-bit digits over the bits the keys can
//                   have: per pass k_coo_radix<false> (per-block digit counts, digit-major), one exclusive scan, k_coo_radix<true>
//                   (stable rank of every key inside its block from per-thread counter columns in LDS, the block's keys laid out
//                   digit by digit in LDS and copied out in runs)
//   k_coo_heads     flag the first occurrence of every key; exclusive scan of the flags = position in the CSR
//   k_coo_emit      indices[pos] = key % n_items, row pointer entries of the users that start (gaps = empty users)
// then K0 (rp_csr_prepare) as for any other input.
#include <cstring>

#include <hip/hip_runtime.h>

#include <algorithm>

#include "rp_common.h"
#include "rp_wave.h"

// ---- LSD radix sort of 64-bit keys, one 4-bit digit per pass ---------------------------------------------------------------------
// A block owns RS_TILE consecutive keys; thread t owns the RS_E consecutive keys [t * RS_E, (t + 1) * RS_E) of them, so "block,
// thread, position in the thread's run" is the input order and a rank built in that order is STABLE (an LSD pass must be).
//   SCATTER = false   counts[d * n_blocks + b] = keys of block b with digit d  (digit-major: ONE exclusive scan of this array gives
//                     every (digit, block) group its place in the output)
//   SCATTER = true    the keys come into LDS with coalesced loads; every thread walks its run and numbers each key among the keys of
//                     the same digit it has seen (its own column of 16 counters: no atomics); a scan over the columns - digit-major,
//                     thread-minor - turns the numbers into positions inside the block's sorted tile; the tile is laid out in LDS
//                     and copied out: consecutive threads write consecutive keys of a digit's run
constexpr int RS_T = 256, RS_E = 16, RS_TILE = RS_T * RS_E, RS_BITS = 4, RS_BINS = 1 << RS_BITS;

template <bool SCATTER>
__global__ void __launch_bounds__(RS_T) k_coo_radix(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out, long long n, int shift,
                                                    int32_t* __restrict__ counts, const int32_t* __restrict__ offsets, int n_blocks) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const long long k0 = (long long)b * RS_TILE;
  const int n_here = (int)min((long long)RS_TILE, n - k0);
  if constexpr (!SCATTER) {
    __shared__ int hist[RS_BINS];
    if (tid < RS_BINS) hist[tid] = 0;
    __syncthreads();
    int c[RS_BINS];
#pragma unroll
    for (int d = 0; d < RS_BINS; ++d) c[d] = 0;
    for (int i = tid; i < n_here; i += RS_T) {
      const int dg = (int)(in[k0 + i] >> shift) & (RS_BINS - 1);
#pragma unroll
      for (int d = 0; d < RS_BINS; ++d) c[d] += (dg == d) ? 1 : 0;
    }
#pragma unroll
    for (int d = 0; d < RS_BINS; ++d) {
      int v = c[d];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && v) atomicAdd(&hist[d], v);
    }
    __syncthreads();
    if (tid < RS_BINS) counts[(long long)tid * n_blocks + b] = hist[tid];
  } else {
    constexpr int PAD = RS_TILE + RS_TILE / RS_E;  // (a thread's run starts 17 keys after its neighbour's: its LDS reads spread over the banks)
    __shared__ unsigned long long stage[PAD];
    __shared__ uint16_t cnt[RS_BINS * RS_T];       // [digit][thread]
    __shared__ int dstart[RS_BINS + 1], gbase[RS_BINS], wsum[RS_T / 64];
    for (int i = tid; i < n_here; i += RS_T) stage[i + i / RS_E] = in[k0 + i];
#pragma unroll
    for (int d = 0; d < RS_BINS; ++d) cnt[d * RS_T + tid] = 0;
    if (tid < RS_BINS) gbase[tid] = offsets[(long long)tid * n_blocks + b];
    __syncthreads();
    unsigned long long key[RS_E];
    int rk[RS_E];  // digit | number among the thread's earlier keys of that digit << 8
#pragma unroll
    for (int e = 0; e < RS_E; ++e) {
      const int i = tid * RS_E + e;
      rk[e] = -1;
      key[e] = 0ull;
      if (i < n_here) {
        key[e] = stage[i + tid];
        const int dg = (int)(key[e] >> shift) & (RS_BINS - 1);
        const int c = cnt[dg * RS_T + tid];
        cnt[dg * RS_T + tid] = (uint16_t)(c + 1);
        rk[e] = dg | (c << 8);
      }
    }
    __syncthreads();  // (every run has left the stage: it is reused for the sorted tile)
    // exclusive scan of cnt in memory order: thread q takes the 16 consecutive entries [16 q, 16 q + 16)
    {
      int v[RS_E], sum = 0;
#pragma unroll
      for (int e = 0; e < RS_E; ++e) {
        v[e] = cnt[tid * RS_E + e];
        sum += v[e];
      }
      const int inc = rp_wave_incl_scan(sum, lane);
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      int run = inc - sum;
      for (int w = 0; w < wave; ++w) run += wsum[w];
      if ((tid & (RS_T / RS_BINS - 1)) == 0) dstart[tid / (RS_T / RS_BINS)] = run;  // (entry 16 q is (digit q / 16, thread 0) when q % 16 == 0)
      if (tid == RS_T - 1) dstart[RS_BINS] = run + sum;
#pragma unroll
      for (int e = 0; e < RS_E; ++e) {
        cnt[tid * RS_E + e] = (uint16_t)run;
        run += v[e];
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < RS_E; ++e)
      if (rk[e] >= 0) stage[cnt[(rk[e] & 255) * RS_T + tid] + (rk[e] >> 8)] = key[e];
    __syncthreads();
    const int total = dstart[RS_BINS];
    for (int i = tid; i < total; i += RS_T) {
      const unsigned long long k = stage[i];
      const int dg = (int)(k >> shift) & (RS_BINS - 1);
      out[(long long)gbase[dg] + (i - dstart[dg])] = k;
    }
  }
}

// sorts `n` keys held in `a` (scratch `b`, both n keys); returns the buffer that holds the result.  counts / offsets: 16 * n_blocks + 1 ints each
static int rp_coo_sort(rphip_ctx* ctx, unsigned long long* a, unsigned long long* b, int64_t n, int bits, int32_t* counts, int32_t* offsets,
                       unsigned long long** sorted) {
  hipStream_t st = ctx->stream;
  const int n_blocks = (int)rp_cdiv(n, RS_TILE);
  unsigned long long *src = a, *dst = b;
  for (int shift = 0; shift < bits; shift += RS_BITS) {
    hipLaunchKernelGGL(k_coo_radix<false>, dim3(n_blocks), dim3(RS_T), 0, st, src, dst, (long long)n, shift, counts, offsets, n_blocks);
    RP_TRY(rp_exclusive_scan_i32(ctx, counts, offsets, (int64_t)RS_BINS * n_blocks));
    hipLaunchKernelGGL(k_coo_radix<true>, dim3(n_blocks), dim3(RS_T), 0, st, src, dst, (long long)n, shift, counts, offsets, n_blocks);
    RP_HIP(hipGetLastError());
    std::swap(src, dst);
  }
  *sorted = src;
  return RPHIP_OK;
}

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
    char* sc = nullptr;
    const size_t kb = ((size_t)n * 8 + 255) & ~(size_t)255, hb = ((size_t)(n + 1) * 4 + 255) & ~(size_t)255;
    const size_t cb = (((size_t)RS_BINS * (size_t)rp_cdiv(n, RS_TILE) + 1) * 4 + 255) & ~(size_t)255;  // digit counts per block, and their scan
    int rc = rp_scratch(ctx, 0, 2 * kb + 2 * hb + 2 * cb + 512, (void**)&sc);
    if (rc != RPHIP_OK) return fail(rc);
    unsigned long long* key = (unsigned long long*)sc;
    unsigned long long* other = (unsigned long long*)(sc + kb);
    unsigned long long* sorted = nullptr;
    int32_t* head = (int32_t*)(sc + 2 * kb);
    int32_t* pos = (int32_t*)(sc + 2 * kb + hb);
    int32_t* err = (int32_t*)(sc + 2 * kb + 2 * hb);
    int32_t* counts = (int32_t*)(sc + 2 * kb + 2 * hb + 256);
    int32_t* offsets = (int32_t*)(sc + 2 * kb + 2 * hb + 256 + cb);
    hipEvent_t ev;
    rp_time_begin(ctx, &ev);
    hipMemsetAsync(err, 0, 4, st);
    const int grid = (int)std::min<int64_t>((int64_t)ctx->n_cu * 16, rp_cdiv(n, 256));
    hipLaunchKernelGGL(k_coo_keys, dim3(grid), dim3(256), 0, st, uid_dev, iid_dev, n, n_users, n_items, key, err);
    rc = rp_coo_sort(ctx, key, other, n, bits, counts, offsets, &sorted);
    if (rc != RPHIP_OK) return fail(rc);
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

// K0: the device-resident interaction matrix: X (CSR by user), X^T (CSR by item), item degrees, column tiles.
//
// Replaces InteractionMatrix.values + to_csr_matrix(binary=True) (recpack/matrix/interaction_matrix.py:202-217,
// recpack/matrix/util.py:27-48) as far as the hot path needs them: only the pattern (indptr, indices) is used.
// X^T is what scipy builds implicitly for `X.T @ X` (nearest_neighbour.py:80 via sklearn, ease.py:81).
//
// Round 4: the transposition is a two-level counting sort whose scattered writes stay inside L2 (round 3: every entry's rank by
// returning LDS atomics with x_idx read once per 32 768-item range, then 5 * 10^7 isolated 4-byte stores over 200 MB: 3.4 ms):
//   k_check_ptr   per user: the row pointers are one (monotone, first 0, last nnz), longest history, P = sum d_u^2, the tile pointers
//                 of rows without entries
//   k_rows        per ENTRY (round 5; 8 lanes per user row before: 0.86 ms at 1M x 100K, bound by its per-row chains): persistent
//                 blocks take 4096 consecutive entries, four per thread in one 16-byte load, an entry's row by a search in the
//                 block's row pointers (LDS); column order check, tile pointers, tile-relative columns (one 8-byte store per
//                 four), and the number of entries per BUCKET of 512 consecutive items (LDS histogram per block)
//   scan          bucket offsets
//   k_partition   8192 consecutive entries per block: (item inside its bucket, user) packed into one word, reordered in LDS by
//                 bucket and written out in runs - every bucket's entries end up contiguous (coalesced writes)
//   k_bucket<0>   per 8192 entries of ONE bucket: LDS counters of its 512 items -> deg (one global atomic per (block, item))
//   scan          xt_ptr = exclusive_scan(deg)
//   k_bucket<1>   the same blocks: rank inside the block by returning LDS atomics, a span per (block, item) from a global cursor,
//                 xt_idx[xt_ptr[j] + rank] = u - the stores of a bucket land in the ~1 MB its 512 item rows occupy
//   k_inv_deg     inv_deg[j] = fl(1/c_j)
#include <algorithm>

#include <cmath>

#include "rp_common.h"
#include "rp_wave.h"

constexpr int RP_K0_BUCKET_MAX = 4096;  // buckets the partition pass can keep in LDS
constexpr int RP_K0_E = 8192;           // entries per block of the partition and bucket passes
constexpr int RP_K0_T = 256;            // threads of those blocks

constexpr int RP_K0_RE = 4096;          // entries per chunk of the row pass

// One pass over the entries of X, flat (see the header):
//   * are the column indices of every row strictly increasing?  (err[1] = 1 if not; with several column tiles err[0] = 3)
//   * a column outside [0, n_items): err[0] = 1
//   * n_tiles > 1: tile_ptr[u][t] = first position of row u with column >= t * tile_w - an entry whose tile differs from
//     its predecessor's opens the tiles in between; the row's first entry opens tiles 0.., its last closes the rest (rows without
//     entries: k_check_ptr)
//   * x_rel = the column inside its tile
//   * bucket_cnt[b] += entries with item in bucket b (items [xt_lo, xt_hi) only: a shard counts the rows it will fit)
__global__ void __launch_bounds__(256) k_rows(const int32_t* __restrict__ x_ptr, const int32_t* __restrict__ x_idx, int n_users, long long nnz,
                                             int n_items, int n_tiles, int tile_w, int32_t* __restrict__ tile_ptr, int32_t* __restrict__ err,
                                             int xt_lo, int xt_hi, uint16_t* __restrict__ x_rel, int bshift, int n_buckets,
                                             int32_t* __restrict__ bucket_cnt, int vec_ok, const int32_t* __restrict__ chunk_u0) {
  extern __shared__ int bh[];      // [n_buckets], then the chunk's row pointers [RP_K0_RE + 2]
  int* rows = bh + n_buckets;
  const int tid = threadIdx.x;
  for (int b = tid; b < n_buckets; b += 256) bh[b] = 0;
  const float inv_w = 1.0f / (float)tile_w;
  auto tile_of = [&](int j) {
    int t = (int)((float)j * inv_w);
    while ((long long)(t + 1) * tile_w <= j) ++t;
    while ((long long)t * tile_w > j) --t;
    return t;
  };
  const long long n_chunks = (nnz + RP_K0_RE - 1) / RP_K0_RE;
  for (long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const long long k0 = c * RP_K0_RE, k1 = min(nnz, k0 + RP_K0_RE);
    __syncthreads();  // (the previous chunk's row pointers are dead; the histogram is cleared)
    // users whose rows meet [k0, k1): from the user of entry k0 to the user of entry k1 (k_check_ptr left both; two binary searches
    // over x_ptr by one thread per chunk were ~20 dependent loads with the other 255 waiting)
    const int u0 = chunk_u0[c], nu = chunk_u0[c + 1] + 1 - u0;  // (x_ptr[u0 + nu] >= k1)
    const bool cached = nu <= RP_K0_RE + 1;  // (rows without entries inside the range count too: longer ranges are searched in x_ptr itself)
    if (cached)
      for (int i = tid; i <= nu; i += 256) rows[i] = x_ptr[u0 + i];
    __syncthreads();
    const int32_t* __restrict__ grow = x_ptr + u0;
    auto R = [&](int i) { return cached ? rows[i] : grow[i]; };
#pragma unroll 2
    for (int g = tid; g < RP_K0_RE / 4; g += 256) {
      const long long k = k0 + 4 * (long long)g;
      if (k >= k1) break;
      const int n = (int)min((long long)4, k1 - k);
      int j4[4] = {0, 0, 0, 0};
      if (vec_ok && n == 4) {
        const int4 v = *reinterpret_cast<const int4*>(x_idx + k);
        j4[0] = v.x, j4[1] = v.y, j4[2] = v.z, j4[3] = v.w;
      } else {
        for (int e = 0; e < n; ++e) j4[e] = x_idx[k + e];
      }
      int prev = (k > 0) ? x_idx[k - 1] : -1;
      int lo = 0, hi = nu;  // the row u0 + lo with R(lo) <= k < R(lo + 1)
      while (hi - lo > 1) {
        const int m = (lo + hi) >> 1;
        if (R(m) <= (int)k) lo = m; else hi = m;
      }
      int rs = R(lo), re = R(lo + 1);
      uint32_t rel[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (e < n) {
          const int kk = (int)k + e;
          while (kk >= re) {  // the next row with entries
            ++lo;
            rs = re;
            re = R(lo + 1);
          }
          const bool first = kk == rs;
          const int j = j4[e];
          if (!first && j <= prev) {
            err[1] = 1;
            if (tile_ptr) err[0] = 3;
          }
          const bool in_range = (unsigned)j < (unsigned)n_items;
          if (!in_range) err[0] = 1;
          if (in_range && j >= xt_lo && j < xt_hi) atomicAdd(&bh[j >> bshift], 1);
          const int tj = (in_range && n_tiles > 1) ? tile_of(j) : 0;
          rel[e] = in_range ? (uint32_t)(j - tj * tile_w) : 0u;
          if (tile_ptr && in_range) {
            int32_t* tp = tile_ptr + (int64_t)(u0 + lo) * (n_tiles + 1);
            const int tprev = (!first && (unsigned)prev < (unsigned)n_items) ? tile_of(prev) : -1;
            for (int t = tprev + 1; t <= tj; ++t) tp[t] = kk;
            if (kk == re - 1)
              for (int t = tj + 1; t <= n_tiles; ++t) tp[t] = re;
          }
          prev = j;
        }
      }
      if (n == 4) {
        *reinterpret_cast<uint2*>(x_rel + k) = make_uint2(rel[0] | (rel[1] << 16), rel[2] | (rel[3] << 16));
      } else {
        for (int e = 0; e < n; ++e) x_rel[k + e] = (uint16_t)rel[e];
      }
    }
  }
  __syncthreads();
  for (int b = tid; b < n_buckets; b += 256)
    if (bh[b]) atomicAdd(&bucket_cnt[b], bh[b]);
}

// Entries [k0, k0 + E) of X (flat) into their buckets: word = (item - bucket * 2^bshift) << ubits | user.  A thread takes groups of
// four consecutive entries (one 16-byte load); the block's row pointers sit in LDS relative to k0 (16 bits each), the first entry of
// a group finds its row by a search there and the others walk on.  Per block: LDS histogram over the buckets (the returning add is
// the entry's rank inside the block's share of its bucket), one global atomic per (block, bucket) reserves the share's place, the
// entries are laid out bucket by bucket in LDS and copied out: consecutive threads write consecutive words of a bucket (the bucket
// of a stage position from a table of every 16th position's bucket + a short walk).
template <typename PT>
__global__ void __launch_bounds__(RP_K0_T) k_partition(const int32_t* __restrict__ x_ptr, const int32_t* __restrict__ x_idx, int n_users, long long nnz,
                                                       int n_items, int xt_lo, int xt_hi, int bshift, int ubits, int n_buckets,
                                                       const int32_t* __restrict__ bucket_off, int32_t* __restrict__ bucket_fill, PT* __restrict__ out,
                                                       const int32_t* __restrict__ chunk_u0, int n_chunks_re, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) char sm_p[];
  static_assert(RP_K0_E % RP_K0_RE == 0, "a partition block is a whole number of row-pass chunks");
  static_assert(RP_K0_E <= 65535, "row pointers relative to the block's first entry are kept in 16 bits");
  constexpr int G = RP_K0_E / 4 / RP_K0_T;  // groups of four entries per thread
  constexpr int R = 4 * G;
  PT* stage = reinterpret_cast<PT*>(sm_p);                                       // [E]
  int* hist = reinterpret_cast<int*>(sm_p + (size_t)RP_K0_E * sizeof(PT));       // [n_buckets] count, then global base - local start
  int* lstart = hist + n_buckets;                                                // [n_buckets + 1] start inside the stage
  uint16_t* ptab = reinterpret_cast<uint16_t*>(lstart + n_buckets + 1);          // [E / 16 + 2] bucket of every 16th stage position
  uint16_t* rows = ptab + (RP_K0_E / 16 + 2);                                    // [E + 2] row pointers - k0, clamped to [0, E]
  __shared__ int wsum[RP_K0_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long k0 = (long long)blockIdx.x * RP_K0_E, k1 = min(nnz, k0 + RP_K0_E);
  for (int b = tid; b < n_buckets; b += RP_K0_T) hist[b] = 0;
  // users whose rows meet [k0, k1): the user of entry k0 .. the user of entry k1 (chunk_u0 of k_check_ptr, see k_rows)
  const int cc = (int)blockIdx.x * (RP_K0_E / RP_K0_RE);
  const int u0 = chunk_u0[cc], nu = chunk_u0[min(cc + RP_K0_E / RP_K0_RE, n_chunks_re)] + 1 - u0;  // (rows without entries inside the range are harmless: the search skips them)
  // nu counts the EMPTY rows inside the range too, so it is unbounded (50 000 history-less users between two entries): the LDS copy
  // holds E + 2 pointers - longer ranges are searched in x_ptr itself
  const bool cached = nu <= RP_K0_E + 1;
  const int32_t* __restrict__ grow = x_ptr + u0;
  auto rel = [&](int p) { return (int)min((long long)RP_K0_E, max((long long)0, (long long)p - k0)); };
  if (cached)
    for (int i = tid; i <= nu; i += RP_K0_T) rows[i] = (uint16_t)rel(grow[i]);
  __syncthreads();
  auto RW = [&](int i) { return cached ? (int)rows[i] : rel(grow[i]); };
  int bk[R];
  PT pay[R];
  int rk[R];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int kr = 4 * (tid + g * RP_K0_T);  // first entry of the group, relative to k0
    const long long k = k0 + kr;
    const int n = (int)max((long long)0, min((long long)4, k1 - k));
    int j4[4] = {-1, -1, -1, -1};
    if (vec_ok && n == 4) {
      const int4 v = *reinterpret_cast<const int4*>(x_idx + k);
      j4[0] = v.x, j4[1] = v.y, j4[2] = v.z, j4[3] = v.w;
    } else {
      for (int e = 0; e < n; ++e) j4[e] = x_idx[k + e];
    }
    int lo = 0, hi = nu;  // the row u0 + lo with RW(lo) <= kr < RW(lo + 1)
    if (n > 0) {
      while (hi - lo > 1) {
        const int m = (lo + hi) >> 1;
        if (RW(m) <= kr) lo = m; else hi = m;
      }
    }
    int re = (n > 0) ? RW(lo + 1) : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * g + e;
      bk[r] = -1;
      rk[r] = 0;
      pay[r] = 0;
      if (e < n) {
        while (kr + e >= re) {  // the next row with entries
          ++lo;
          re = RW(lo + 1);
        }
        const int j = j4[e];
        if ((unsigned)j < (unsigned)n_items && j >= xt_lo && j < xt_hi) {
          const int b = j >> bshift;
          bk[r] = b;
          pay[r] = ((PT)(j - (b << bshift)) << ubits) | (PT)(u0 + lo);
          rk[r] = atomicAdd(&hist[b], 1);
        }
      }
    }
  }
  __syncthreads();
  // start of every bucket inside the stage: exclusive scan over the buckets (a serial run per thread + a scan of the threads' sums)
  {
    const int per = (n_buckets + RP_K0_T - 1) / RP_K0_T;
    const int b0 = tid * per, b1 = min(n_buckets, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += hist[b];
    const int inc = rp_wave_incl_scan(sum, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (tid == RP_K0_T - 1) lstart[n_buckets] = run + sum;
    for (int b = b0; b < b1; ++b) {
      const int c = hist[b];
      lstart[b] = run;
      // the share's place in the bucket: ONE global atomic per (block, bucket); hist[b] becomes (global position of the share) - (its start in the stage)
      hist[b] = c ? bucket_off[b] + atomicAdd(&bucket_fill[b], c) - run : 0;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (bk[r] >= 0) stage[lstart[bk[r]] + rk[r]] = pay[r];
  for (int sl = tid; sl <= RP_K0_E / 16; sl += RP_K0_T) {
    const int i = 16 * sl;
    int lo = 0, hi = n_buckets;  // the bucket with lstart[lo] <= i < lstart[lo + 1] (empty buckets share a start: take the last)
    while (hi - lo > 1) {
      const int m = (lo + hi) >> 1;
      if (lstart[m] <= i) lo = m; else hi = m;
    }
    ptab[sl] = (uint16_t)lo;
  }
  __syncthreads();
  const int total = lstart[n_buckets];
  for (int i = tid; i < total; i += RP_K0_T) {
    int b = ptab[i >> 4];
    while (lstart[b + 1] <= i) ++b;
    out[hist[b] + i] = stage[i];
  }
}

// Blocks of the bucket passes: bucket b has ceil(size_b / E) of them; blk_off = exclusive scan of that.  One thread per bucket.
__global__ void k_bucket_blocks(const int32_t* __restrict__ bucket_off, int n_buckets, int32_t* __restrict__ n_blk) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n_buckets) n_blk[b] = (bucket_off[b + 1] - bucket_off[b] + RP_K0_E - 1) / RP_K0_E;
}

// FILL = false: deg[item] += the block's count.  FILL = true: xt_idx[xt_ptr[item] + place] = user, place = the block's span of the
// item's row (one returning global atomic per (block, item) on `cursor`) + the entry's rank inside the block (returning LDS add).
// A block works on at most E entries of ONE bucket: the item counters are 2^bshift words of LDS.  Blocks are dealt so that the eight
// XCDs each take a contiguous eighth of them (blockIdx mod 8 names the XCD): the stores of a bucket meet in one L2.
template <typename PT, bool FILL>
__global__ void __launch_bounds__(RP_K0_T) k_bucket(const PT* __restrict__ part, const int32_t* __restrict__ bucket_off, const int32_t* __restrict__ blk_off,
                                                    int n_buckets, int bshift, int ubits, int32_t* __restrict__ deg, const int32_t* __restrict__ xt_ptr,
                                                    int32_t* __restrict__ cursor, int32_t* __restrict__ xt_idx, int f_lo, int f_hi) {
  extern __shared__ int cnt[];
  constexpr int R = RP_K0_E / RP_K0_T;
  const int tid = threadIdx.x;
  const int nblk = blk_off[n_buckets];
  const int per_x = ((int)gridDim.x + 7) >> 3;
  const int vb = ((int)blockIdx.x & 7) * per_x + ((int)blockIdx.x >> 3);
  if (vb >= nblk) return;
  __shared__ int s_b;
  if (tid == 0) {
    int lo = 0, hi = n_buckets;  // the bucket with blk_off[lo] <= vb < blk_off[lo + 1] (buckets without blocks share a start: the last)
    while (hi - lo > 1) {
      const int m = (lo + hi) >> 1;
      if (blk_off[m] <= vb) lo = m; else hi = m;
    }
    s_b = lo;
  }
  const int bw = 1 << bshift;
  for (int w = tid; w < bw; w += RP_K0_T) cnt[w] = 0;
  __syncthreads();
  const int b = s_b;
  const int e0 = bucket_off[b] + (vb - blk_off[b]) * RP_K0_E, e1 = min(bucket_off[b + 1], e0 + RP_K0_E);
  const PT umask = ((PT)1 << ubits) - 1;
  PT pay[R];
  int rk[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = e0 + tid + r * RP_K0_T;
    pay[r] = 0;
    rk[r] = -1;
    if (e < e1) {
      pay[r] = part[e];
      rk[r] = atomicAdd(&cnt[(int)(pay[r] >> ubits)], 1);
    }
  }
  __syncthreads();
  const int item0 = b << bshift;
  for (int w = tid; w < bw; w += RP_K0_T) {
    const int c = cnt[w];
    if (c) {
      if (FILL)  // (rows outside [f_lo, f_hi) are not wanted - a matrix built for a row range: marked, their entries skipped)
        cnt[w] = (item0 + w >= f_lo && item0 + w < f_hi) ? xt_ptr[item0 + w] + atomicAdd(&cursor[item0 + w], c) : -1;
      else
        atomicAdd(&deg[item0 + w], c);
    } else if (FILL) {
      cnt[w] = -1;
    }
  }
  if (!FILL) return;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (rk[r] >= 0) {
      const int base = cnt[(int)(pay[r] >> ubits)];
      if (base >= 0) xt_idx[base + rk[r]] = (int32_t)(pay[r] & umask);
    }
}

__global__ void k_inv_deg(const int32_t* __restrict__ deg, double* __restrict__ inv, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int d = deg[i];
    inv[i] = d > 0 ? 1.0 / (double)d : 0.0;
  }
}

// per user: the row pointers are valid, the longest history (err[4]), P = sum d_u^2 (*p_sum), and - several column tiles - the
// tile pointers of the rows WITHOUT entries (k_rows visits entries)
__global__ void k_check_ptr(const int32_t* __restrict__ x_ptr, int n_users, int64_t nnz, int32_t* __restrict__ err,
                            unsigned long long* __restrict__ p_sum, int32_t* __restrict__ tile_ptr, int n_tiles, int32_t* __restrict__ chunk_u0) {
  int dmax = 0;
  unsigned long long local = 0;
  const int n_chunks = (int)((nnz + RP_K0_RE - 1) / RP_K0_RE);
  if (blockIdx.x == 0 && threadIdx.x == 0) chunk_u0[n_chunks] = max(n_users - 1, 0);
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u <= n_users; u += gridDim.x * blockDim.x) {
    const int a = x_ptr[u];
    if (u == 0 && a != 0) *err = 2;
    if (u == n_users && a != (int)nnz) *err = 2;
    if (u < n_users) {
      const int d = x_ptr[u + 1] - a;
      if (d < 0) *err = 2;
      dmax = max(dmax, d);
      if (d > 0) local += (unsigned long long)d * (unsigned long long)d;
      // chunk_u0[c] = the user whose row holds entry c * RP_K0_RE (the entry passes start their row search there; with invalid
      // pointers - a or a + d outside [0, nnz] - nothing is written: the passes do not run then)
      if (d > 0 && a >= 0 && (int64_t)a + d <= nnz)
        for (int c = (a + RP_K0_RE - 1) / RP_K0_RE; (int64_t)c * RP_K0_RE < (int64_t)a + d; ++c) chunk_u0[c] = u;
      if (d == 0 && tile_ptr) {
        int32_t* tp = tile_ptr + (int64_t)u * (n_tiles + 1);
        for (int t = 0; t <= n_tiles; ++t) tp[t] = a;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dmax = max(dmax, __shfl_xor(dmax, o));
    local += __shfl_xor(local, o);
  }
  // (one atomic per BLOCK: 16 000 waves queueing for one word were 0.16 of this kernel's 0.18 ms)
  __shared__ int s_max[16];
  __shared__ unsigned long long s_sum[16];
  if ((threadIdx.x & 63) == 0) {
    s_max[threadIdx.x >> 6] = dmax;
    s_sum[threadIdx.x >> 6] = local;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
      dmax = max(dmax, s_max[w]);
      local += s_sum[w];
    }
    if (dmax > 0) atomicMax(&err[4], dmax);  // longest history
    if (local) atomicAdd(p_sum, local);
  }
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
  // buckets of 2^bshift consecutive items (512, more when that would give over RP_K0_BUCKET_MAX buckets); a partitioned entry packs
  // (item inside its bucket, user) into 32 bits when they fit (8M users at 512 items per bucket), into 64 otherwise
  int bshift = 9;
  while (((I + ((int64_t)1 << bshift) - 1) >> bshift) > RP_K0_BUCKET_MAX) ++bshift;
  const int n_buckets = (int)std::max<int64_t>(1, (I + ((int64_t)1 << bshift) - 1) >> bshift);
  int ubits = 1;
  while (((int64_t)1 << ubits) < U) ++ubits;
  const bool wide = bshift + ubits > 32;
  if (((size_t)4 << bshift) > (size_t)128 * 1024) RP_FAIL(RPHIP_ELIMIT, "n_items=%lld is beyond what K0's bucket pass holds in LDS", (long long)I);
  char* sc = nullptr;
  const size_t part_bytes = ((size_t)std::max<int64_t>(nnz, 1) * (wide ? 8 : 4) + 15) & ~(size_t)15;  // keeps p_sum 8-B aligned
  const size_t aux_ints = (size_t)5 * n_buckets + 8 + (size_t)std::max<int64_t>(I, 1) + (size_t)rp_cdiv(nnz, RP_K0_RE) + 2;
  RP_TRY(rp_scratch(ctx, 0, part_bytes + 64 + aux_ints * 4, (void**)&sc));
  int32_t* err = (int32_t*)(sc + part_bytes);
  unsigned long long* p_sum = (unsigned long long*)(err + 2);
  int32_t* bucket_cnt = (int32_t*)(sc + part_bytes + 64);  // [nb]
  int32_t* bucket_fill = bucket_cnt + n_buckets;             // [nb]
  int32_t* n_blk = bucket_fill + n_buckets;                  // [nb]
  int32_t* bucket_off = n_blk + n_buckets;                   // [nb + 1]
  int32_t* blk_off = bucket_off + n_buckets + 1;             // [nb + 1]
  int32_t* cursor = blk_off + n_buckets + 1;                 // [I]
  const int n_chunks_re = (int)rp_cdiv(nnz, RP_K0_RE);
  int32_t* chunk_u0 = cursor + std::max<int64_t>(I, 1);      // [n_chunks_re + 1]: the user of every RP_K0_RE-th entry
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  RP_HIP(hipMemsetAsync(m->deg, 0, (size_t)std::max<int64_t>(I, 1) * sizeof(int32_t), st));
  RP_HIP(hipMemsetAsync(err, 0, 64 + aux_ints * 4, st));  // (flags, P, bucket counters, cursor)
  const int bt = 256;
  const int grid_u = (int)std::min<int64_t>((int64_t)ctx->n_cu * 4, std::max<int64_t>(1, rp_cdiv(U + 1, bt)));
  if (m->n_tiles > 1) RP_TRY(rp_dev_alloc(ctx, (size_t)std::max<int64_t>(U, 1) * (m->n_tiles + 1) * sizeof(int32_t), (void**)&m->tile_ptr));
  hipLaunchKernelGGL(k_check_ptr, dim3(grid_u), dim3(bt), 0, st, m->x_ptr, (int)U, nnz, err, p_sum, m->tile_ptr, m->n_tiles, chunk_u0);
  int32_t h_chk[5] = {0, 0, 0, 0, 0};
  RP_HIP(hipMemcpyAsync(h_chk, err, sizeof(h_chk), hipMemcpyDeviceToHost, st));
  RP_HIP(hipStreamSynchronize(st));
  const int32_t h_err = h_chk[0];
  m->max_user_deg = h_chk[4];
  if (h_err) RP_FAIL(RPHIP_EINVAL, "indptr is not a valid row pointer array (not monotone, or indptr[0] != 0, or indptr[-1] != nnz)");
  // a shard with deg_partial counts (and fills) only the item rows it will fit
  const int64_t i_lo = m->deg_partial ? std::max<int64_t>(0, m->xt_row_begin) : 0;
  const int64_t i_hi = m->deg_partial ? (m->xt_row_end < 0 ? I : std::min<int64_t>(I, m->xt_row_end)) : I;
  // (the rows of X^T that are filled: a matrix built for a row range fills those rows only, its degrees cover all items)
  const int64_t f_lo = std::max<int64_t>(0, m->xt_row_begin), f_hi = m->xt_row_end < 0 ? I : std::min<int64_t>(I, m->xt_row_end);
  {  // tile-relative columns, padded so that a chunk of eight read at any entry stays inside
    const size_t n_rel = (((size_t)nnz + 15) & ~(size_t)15) + 16;
    RP_TRY(rp_dev_alloc(ctx, n_rel * sizeof(uint16_t), (void**)&m->x_rel));
    // (k_rows writes every entry; only the padding behind them is cleared)
    const size_t done = (size_t)nnz & ~(size_t)7;
    RP_HIP(hipMemsetAsync(m->x_rel + done, 0, (n_rel - done) * sizeof(uint16_t), st));
  }
  if (nnz > 0) {
    const int grid_s = (int)std::min<int64_t>((int64_t)ctx->n_cu * 6, rp_cdiv(nnz, RP_K0_RE));
    const int vec_ok = ((reinterpret_cast<uintptr_t>(m->x_idx) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_rows, dim3(grid_s), dim3(bt), ((size_t)n_buckets + RP_K0_RE + 2) * 4, st, m->x_ptr, m->x_idx, (int)U, (long long)nnz, (int)I,
                       m->n_tiles, m->tile_w, m->tile_ptr, err, (int)i_lo, (int)i_hi, m->x_rel, bshift, n_buckets, bucket_cnt, vec_ok, chunk_u0);
  }
  RP_TRY(rp_exclusive_scan_i32(ctx, bucket_cnt, bucket_off, n_buckets));
  if (nnz > 0 && I > 0) {
    const unsigned pgrid = (unsigned)rp_cdiv(nnz, RP_K0_E);
    const size_t plds = (size_t)RP_K0_E * (wide ? 8 : 4) + ((size_t)2 * n_buckets + 1) * 4 + ((size_t)RP_K0_E / 16 + 2) * 2 + ((size_t)RP_K0_E + 2) * 2 + 16;
    const int pvec_ok = ((reinterpret_cast<uintptr_t>(m->x_idx) & 15) == 0) ? 1 : 0;
    if (wide) {
      RP_HIP(hipFuncSetAttribute((const void*)k_partition<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
      hipLaunchKernelGGL(k_partition<uint64_t>, dim3(pgrid), dim3(RP_K0_T), plds, st, m->x_ptr, m->x_idx, (int)U, (long long)nnz, (int)I, (int)i_lo,
                         (int)i_hi, bshift, ubits, n_buckets, bucket_off, bucket_fill, (uint64_t*)sc, chunk_u0, n_chunks_re, pvec_ok);
    } else {
      RP_HIP(hipFuncSetAttribute((const void*)k_partition<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
      hipLaunchKernelGGL(k_partition<uint32_t>, dim3(pgrid), dim3(RP_K0_T), plds, st, m->x_ptr, m->x_idx, (int)U, (long long)nnz, (int)I, (int)i_lo,
                         (int)i_hi, bshift, ubits, n_buckets, bucket_off, bucket_fill, (uint32_t*)sc, chunk_u0, n_chunks_re, pvec_ok);
    }
    hipLaunchKernelGGL(k_bucket_blocks, dim3((unsigned)rp_cdiv(n_buckets, 256)), dim3(256), 0, st, bucket_off, n_buckets, n_blk);
    RP_TRY(rp_exclusive_scan_i32(ctx, n_blk, blk_off, n_buckets));
    const unsigned bgrid = (unsigned)(rp_cdiv(rp_cdiv(nnz, RP_K0_E) + n_buckets, 8) * 8);
    const size_t blds = (size_t)4 << bshift;
    if (wide) {
      RP_HIP(hipFuncSetAttribute((const void*)k_bucket<uint64_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
      RP_HIP(hipFuncSetAttribute((const void*)k_bucket<uint64_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
      hipLaunchKernelGGL((k_bucket<uint64_t, false>), dim3(bgrid), dim3(RP_K0_T), blds, st, (const uint64_t*)sc, bucket_off, blk_off, n_buckets, bshift, ubits,
                         m->deg, nullptr, nullptr, nullptr, 0, 0);
    } else {
      RP_HIP(hipFuncSetAttribute((const void*)k_bucket<uint32_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
      RP_HIP(hipFuncSetAttribute((const void*)k_bucket<uint32_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
      hipLaunchKernelGGL((k_bucket<uint32_t, false>), dim3(bgrid), dim3(RP_K0_T), blds, st, (const uint32_t*)sc, bucket_off, blk_off, n_buckets, bshift, ubits,
                         m->deg, nullptr, nullptr, nullptr, 0, 0);
    }
    RP_TRY(rp_exclusive_scan_i32(ctx, m->deg, m->xt_ptr, I));
    // (a shard's degrees - and with them xt_ptr - cover its own rows [i_lo, i_hi): exactly the rows it fills)
    if (wide)
      hipLaunchKernelGGL((k_bucket<uint64_t, true>), dim3(bgrid), dim3(RP_K0_T), blds, st, (const uint64_t*)sc, bucket_off, blk_off, n_buckets, bshift, ubits,
                         m->deg, m->xt_ptr, cursor, m->xt_idx, (int)f_lo, (int)f_hi);
    else
      hipLaunchKernelGGL((k_bucket<uint32_t, true>), dim3(bgrid), dim3(RP_K0_T), blds, st, (const uint32_t*)sc, bucket_off, blk_off, n_buckets, bshift, ubits,
                         m->deg, m->xt_ptr, cursor, m->xt_idx, (int)f_lo, (int)f_hi);
  } else {
    RP_TRY(rp_exclusive_scan_i32(ctx, m->deg, m->xt_ptr, I));
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
  RP_ENTER(ctx);
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
  RP_ENTER(ctx);
  return rphip_csr_from_device_rows(ctx, n_users, n_items, nnz, indptr_dev, indices_dev, 0, -1, out);
}

extern "C" int rphip_csr_from_device_rows(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t nnz, const int32_t* indptr_dev,
                                          const int32_t* indices_dev, int64_t fit_row_begin, int64_t fit_row_end, rphip_csr** out) {
  RP_ENTER(ctx);
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
  RP_ENTER(ctx);
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
  RP_ENTER(ctx);
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
  RP_ENTER(m->ctx);  // (the block cache and the stream belong to the context: a destructor on another thread waits its turn)
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
  RP_ENTER(ctx);
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
  RP_ENTER(ctx);
  if (!ctx || !m || !deg_host) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_TRY(rp_set_device(ctx));
  if (m->n_items > 0) RP_HIP(hipMemcpyAsync(deg_host, m->deg, (size_t)m->n_items * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

extern "C" int rphip_csr_num_products(rphip_ctx* ctx, const rphip_csr* m, int64_t* p_out) {
  RP_ENTER(ctx);
  if (!ctx || !m || !p_out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *p_out = m->n_products;
  return RPHIP_OK;
}

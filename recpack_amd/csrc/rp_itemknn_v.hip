// ItemKNN on a VALUED interaction matrix: the Gram of per-entry float64 weights, K1-K3 for the time-aware family.
//
// Replaces TARSItemKNN._fit (recpack/algorithms/time_aware_item_knn/base.py:166-179): the decayed matrix W
// (:181-201, built by the host class) goes through compute_cosine_similarity / compute_conditional_probability /
// compute_pearson_similarity (recpack/algorithms/nearest_neighbour.py:22-118) and get_top_K_values (recpack/util.py:80-96).
// All three are one kernel:
//     G_ij = sum over the users u of item i that also hold j of  L_ui * R_uj,      S_ij = G_ij * rowscale_i
//   cosine   L = R = W with every column scaled to unit length by the host (what sklearn's cosine_similarity does
//            first: normalize(X.T), then the sparse product), rowscale = 1
//   pearson  the same on the centred matrix (host), entries that centre to exactly 0 dropped as scipy's subtraction drops them
//   conditional probability   L = 1 (to_binary(X).T @ X), R = W, rowscale_i = fl(1 / c_i)   (A @ co_mat, pop_discount = 0)
//
// Stored entries and zeros, as the reference has them: scipy's sparse product stores the (i, j) whose sum is not
// exactly 0, setdiag(0) stores the diagonal as an explicit 0, get_top_K_values takes the min(K, stored) largest stored
// values and the result drops the zero.  With negative entries (pearson) the diagonal's zero outranks them, so per row
//     K_eff = min(K, n_pos) + max(0, min(K - n_pos - 1, n_neg))
// non-zero entries are selected.
// Float64 sums by LDS atomics: the order of the adds shows in the last bits, as scipy's own order does; parity is the
// tie-aware rule with eps = 1e-9 and values at 1e-11 (tests/test_gpu_tars.py).  A coverage kernel like the normalize_X
// variant (rp_itemknn_w.hip): no row slicing, no counter packing.
#include <algorithm>

#include "rp_common.h"
#include "rp_merge_big.h"
#include "rp_select.h"
#include "rp_wave.h"

constexpr int RP_VMERGE_THREADS = 256;
constexpr int RP_VTILE_MAX = 12288;
constexpr int RP_VTILE_HERMANN = 8192;  // 12 bytes per column: the sum and the number of users behind it
constexpr int RPHIP_PAIR_HERMANN = 100;
constexpr unsigned long long RP_SIGN64 = 0x8000000000000000ull;

struct rp_vfit_params {
  const int32_t* x_ptr;
  const int32_t* x_idx;
  const double* x_val;
  const int32_t* xt_ptr;
  const int32_t* xt_idx;
  const int32_t* order;   // rows by descending cost
  const int32_t* vt_ptr;  // T > 1: [U * (T + 1)] first position of row u with column >= t * W
  const double* xt_val;   // !left_binary: [nnz] the value of every X^T entry (order of xt_idx)
  const double* roww;     // [I] or NULL
  int left_binary, signed_vals;
  int pair_kind;          // 0: products L_ui * X_uj; > 0: a decay (RPHIP_DECAY_*) of |X_uj - X_ui|, equal values add nothing
  double pair_decay, pair_max;
  int n_items, n_rows, n_tiles, tile_w, K, row_begin;
  int cand_stride;  // T > 1: entries between two tiles' lists of a row (K; min(K, tile) on the large-K route)
  const int32_t* n_rows_dev;  // when set: the number of rows in `order` (a list built on the device)
  int32_t* ticket;
  int32_t* out_idx;       // T == 1: model rows; T > 1: per-tile candidates
  double* out_val;
  int32_t* out_nnz;
  int32_t* out_cnt;       // T > 1 and signed: n_pos per (row, tile)
};

// any double -> u64 in the same order
__device__ __forceinline__ uint64_t rp_ordered_f64(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b & RP_SIGN64) ? ~b : (b | RP_SIGN64);
}

struct rp_vdense_src {
  const double* acc;
  const uint32_t* cnt;  // Hermann: the number of users behind every sum (the similarity is their average)
  int t0, row;
  int sh;               // leading key bits shared by all candidates (rp_key_prefix): dropped, the radix digits start below them
  __device__ __forceinline__ double value(int w) const { return cnt ? acc[w] * (1.0 / (double)cnt[w]) : acc[w]; }
  __device__ __forceinline__ bool valid(int w) const { return acc[w] != 0.0 && (t0 + w) != row; }
  __device__ __forceinline__ uint64_t key(int w) const { return rp_ordered_f64(value(w)) << sh; }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)(t0 + w); }
};

// one term of the Gram: products, a decay of the time distance, or Hermann's 1 / (distance + age) with its user count
__device__ __forceinline__ void rp_vadd(double* acc, uint32_t* cntw, int j, double q, double x, int pair_kind, double pair_decay,
                                        double pair_max) {
  if (pair_kind == RPHIP_PAIR_HERMANN) {
    // hermann_2010.py:64-103: S_u,i,j = 1 / (|t_uj - t_ui| + now - min(t_ui, t_uj)) for t_uj != t_ui, averaged over ALL
    // users that hold both items (the equal-timestamp pairs count in the denominator only)
    const double d = fabs(x - q);
    atomicAdd(&cntw[j], 1u);
    if (d != 0.0) atomicAdd(&acc[j], 1.0 / (d + (pair_max - fmin(x, q))));
  } else if (pair_kind) {
    const double d = fabs(x - q);
    if (d != 0.0) atomicAdd(&acc[j], rp_pair_decay(pair_kind, pair_decay, pair_max, d));
  } else {
    atomicAdd(&acc[j], q * x);
  }
}

__device__ __forceinline__ int rp_keff(int K, int n_pos, int n_neg) {
  int k = min(K, n_pos);
  const int rest = K - n_pos - 1;
  if (rest > 0) k += min(rest, n_neg);
  return k;
}

__global__ void __launch_bounds__(RP_FIT_THREADS) k_itemknn_fit_values(rp_vfit_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTHR = RP_FIT_THREADS, NW = NTHR / 64;
  double* acc = reinterpret_cast<double*>(smem);
  const int W = P.tile_w, T = P.n_tiles;
  const bool hermann = P.pair_kind == RPHIP_PAIR_HERMANN;
  uint32_t* cntw = hermann ? reinterpret_cast<uint32_t*>(smem + (size_t)W * 8) : nullptr;
  rp_select_smem sel = rp_select_carve(smem + (size_t)W * (hermann ? 12 : 8), W, NW);
  int* s_task = &sel.misc[RPS_USER];
  int* s_cnt = &sel.misc[RPS_USER + 1];  // [2]
  uint32_t* s_red = reinterpret_cast<uint32_t*>(&sel.misc[RPS_USER + 4]);  // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave * 4 + (lane >> 4), gl = lane & 15;  // 64 groups of 16 lanes: one user per group and round
  for (int w = tid; w < W; w += NTHR) {
    acc[w] = 0.0;
    if (hermann) cntw[w] = 0u;
  }
  const long long n_tasks = (long long)(P.n_rows_dev ? *P.n_rows_dev : P.n_rows) * T;
  while (true) {
    __syncthreads();
    if (tid == 0) {
      s_task[0] = atomicAdd(P.ticket, 1);
      s_cnt[0] = s_cnt[1] = 0;
      s_red[0] = s_red[1] = 0xffffffffu;
      s_red[2] = s_red[3] = 0u;
    }
    __syncthreads();
    const long long task = s_task[0];
    if (task >= n_tasks) break;
    const int row = P.order[task / T], tile = (int)(task % T);
    const int t0 = tile * W, n_cols = min(W, P.n_items - t0);
    // ---- accumulate
    // Four users per 16-lane group and round, the loads of one kind issued together: user id and left value, then the
    // slice bounds, then the first 16 entries of every slice.
    const int pe = P.xt_ptr[row + 1];
    constexpr int UNR = 4;
    auto add = [&](int j, double q, double x) __attribute__((always_inline)) {
      rp_vadd(acc, cntw, j, q, x, P.pair_kind, P.pair_decay, P.pair_max);
    };
    for (int p0 = P.xt_ptr[row] + grp; p0 < pe; p0 += 64 * UNR) {
      int u[UNR], s[UNR], e[UNR], c[UNR];
      double q[UNR], x[UNR];
#pragma unroll
      for (int r = 0; r < UNR; ++r) {
        const int p = p0 + 64 * r;
        u[r] = (p < pe) ? P.xt_idx[p] : -1;
        q[r] = (p < pe && !P.left_binary) ? P.xt_val[p] : 1.0;  // the user's value for this row's item
      }
#pragma unroll
      for (int r = 0; r < UNR; ++r) {
        s[r] = e[r] = 0;
        if (u[r] >= 0) {
          if (T > 1) {  // the slice of the user's row that falls into this tile
            const int32_t* tp = P.vt_ptr + (int64_t)u[r] * (T + 1) + tile;
            s[r] = tp[0];
            e[r] = tp[1];
          } else {
            s[r] = P.x_ptr[u[r]];
            e[r] = P.x_ptr[u[r] + 1];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < UNR; ++r) {
        const int k = s[r] + gl;
        c[r] = (k < e[r]) ? P.x_idx[k] : -1;
        x[r] = (k < e[r]) ? P.x_val[k] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < UNR; ++r)
        if (c[r] >= 0) add(c[r] - t0, q[r], x[r]);
#pragma unroll
      for (int r = 0; r < UNR; ++r)  // slices longer than 16 entries
        for (int k = s[r] + gl + 16; k < e[r]; k += 16) add(P.x_idx[k] - t0, q[r], P.x_val[k]);
    }
    __syncthreads();
    // ---- with negative entries the diagonal's stored zero decides how many of them the reference keeps
    int kk = P.K;
    if (P.signed_vals) {
      int np = 0, nn = 0;
      for (int w = tid; w < n_cols; w += NTHR) {
        if ((t0 + w) == row) continue;
        const double v = acc[w];
        np += v > 0.0;
        nn += v < 0.0;
      }
      np = rp_wave_sum(np);
      nn = rp_wave_sum(nn);
      if (lane == 0) {
        atomicAdd(&s_cnt[0], np);
        atomicAdd(&s_cnt[1], nn);
      }
      __syncthreads();
      if (T == 1) kk = rp_keff(P.K, s_cnt[0], s_cnt[1]);
    }
    // ---- select
    rp_vdense_src src{acc, cntw, t0, row, 0};
    src.sh = rp_key_prefix<NTHR>(src, n_cols, s_red);
    const int rl = row - P.row_begin;
    int kp = 0;
    if (kk > 0) kp = rp_select_topk<NTHR>(src, n_cols, kk, sel);
    if (T == 1) {
      const double ri = P.roww ? P.roww[row] : 1.0;
      if (kk > 0)
        rp_select_emit<NTHR>(n_cols, sel, [&](int pos, int w) {
          P.out_idx[(int64_t)rl * P.K + pos] = t0 + w;
          P.out_val[(int64_t)rl * P.K + pos] = src.value(w) * ri;
        });
      if (tid == 0) P.out_nnz[rl] = kp;
    } else {
      const int64_t o = ((int64_t)rl * T + tile) * P.cand_stride;
      rp_select_emit<NTHR>(n_cols, sel, [&](int pos, int w) {
        P.out_idx[o + pos] = t0 + w;
        P.out_val[o + pos] = src.value(w);
      });
      if (tid == 0) {
        P.out_nnz[(int64_t)rl * T + tile] = kp;
        if (P.signed_vals) P.out_cnt[(int64_t)rl * T + tile] = s_cnt[0];
      }
    }
    __syncthreads();
    for (int w = tid; w < W; w += NTHR) {  // (the diagonal is in no list)
      acc[w] = 0.0;
      if (hermann) cntw[w] = 0u;
    }
    __syncthreads();
  }
}

// ---- light rows: all columns at once in a hash table ------------------------------------------------------------------
// With more items than one tile holds, every (row, tile) task walks the row's users again; for the many rows with a
// handful of users that is all fixed cost (C4: 87 % of the rows have at most 64 users).  A row whose products number at
// most RP_VHASH_MAX touches at most that many columns: they fit an open-addressing table in LDS (column -> float64 sum),
// one task per row, no tiles, no merge.  Selection runs over the table's slots; the selected entries are put in column
// order through a small staging area.
constexpr int RP_VHASH_SLOTS = 8192;  // (a power of two)
constexpr int RP_VHASH_MAX = 4800;    // products per row: load factor <= 0.59
constexpr int RP_VHASH_KMAX = 512;    // staging area over the dead histogram: 4 + 8 bytes per selected entry
constexpr uint32_t RP_VHASH_EMPTY = 0xffffffffu;

__device__ __forceinline__ int rp_vhash_slot(uint32_t* keys, uint32_t col) {
  uint32_t h = (col * 2654435761u) >> 19;
  while (true) {
    const uint32_t prev = atomicCAS(&keys[h], RP_VHASH_EMPTY, col);
    if (prev == RP_VHASH_EMPTY || prev == col) return (int)h;
    h = (h + 1) & (RP_VHASH_SLOTS - 1);
  }
}

struct rp_vhash_src {
  const double* acc;
  const uint32_t* cnt;
  const uint32_t* keys;
  int row;
  int sh;
  __device__ __forceinline__ double value(int w) const { return cnt ? acc[w] * (1.0 / (double)cnt[w]) : acc[w]; }
  __device__ __forceinline__ bool valid(int w) const { return keys[w] != RP_VHASH_EMPTY && acc[w] != 0.0 && (int)keys[w] != row; }
  __device__ __forceinline__ uint64_t key(int w) const { return rp_ordered_f64(value(w)) << sh; }
  __device__ __forceinline__ uint32_t col(int w) const { return keys[w]; }
};

__global__ void __launch_bounds__(RP_FIT_THREADS) k_itemknn_fit_values_hash(rp_vfit_params P, const int32_t* __restrict__ rows,
                                                                           const int32_t* __restrict__ n_rows_dev) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTHR = RP_FIT_THREADS, NW = NTHR / 64, W = RP_VHASH_SLOTS;
  double* acc = reinterpret_cast<double*>(smem);
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem + (size_t)W * 8);
  const bool hermann = P.pair_kind == RPHIP_PAIR_HERMANN;
  uint32_t* cntw = hermann ? reinterpret_cast<uint32_t*>(smem + (size_t)W * 12) : nullptr;
  rp_select_smem sel = rp_select_carve(smem + (size_t)W * (hermann ? 16 : 12), W, NW);
  int* s_task = &sel.misc[RPS_USER];
  int* s_cnt = &sel.misc[RPS_USER + 1];  // [2]
  uint32_t* s_red = reinterpret_cast<uint32_t*>(&sel.misc[RPS_USER + 4]);  // [4]
  uint32_t* ecol = sel.hist;                                    // staging (the histogram is dead once the selection is done)
  double* eval = reinterpret_cast<double*>(sel.hist + 1024);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave * 4 + (lane >> 4), gl = lane & 15;
  for (int w = tid; w < W; w += NTHR) {
    acc[w] = 0.0;
    keys[w] = RP_VHASH_EMPTY;
    if (hermann) cntw[w] = 0u;
  }
  const int n_tasks = *n_rows_dev;
  while (true) {
    __syncthreads();
    if (tid == 0) {
      s_task[0] = atomicAdd(P.ticket + 1, 1);
      s_cnt[0] = s_cnt[1] = 0;
      s_red[0] = s_red[1] = 0xffffffffu;
      s_red[2] = s_red[3] = 0u;
    }
    __syncthreads();
    const int task = s_task[0];
    if (task >= n_tasks) break;
    const int row = rows[task];
    // ---- accumulate: one user per 16-lane group and round (these rows have few users)
    const int pe = P.xt_ptr[row + 1];
    for (int p = P.xt_ptr[row] + grp; p < pe; p += 64) {
      const int u = P.xt_idx[p];
      const double q = P.left_binary ? 1.0 : P.xt_val[p];
      const int e = P.x_ptr[u + 1];
      for (int k = P.x_ptr[u] + gl; k < e; k += 16) {
        const int slot = rp_vhash_slot(keys, (uint32_t)P.x_idx[k]);
        rp_vadd(acc, cntw, slot, q, P.x_val[k], P.pair_kind, P.pair_decay, P.pair_max);
      }
    }
    __syncthreads();
    rp_vhash_src src{acc, cntw, keys, row, 0};
    src.sh = rp_key_prefix<NTHR>(src, W, s_red);
    int kk = P.K;
    if (P.signed_vals) {
      int np = 0, nn = 0;
      for (int w = tid; w < W; w += NTHR) {
        if (keys[w] == RP_VHASH_EMPTY || (int)keys[w] == row) continue;
        const double v = acc[w];
        np += v > 0.0;
        nn += v < 0.0;
      }
      np = rp_wave_sum(np);
      nn = rp_wave_sum(nn);
      if (lane == 0) {
        atomicAdd(&s_cnt[0], np);
        atomicAdd(&s_cnt[1], nn);
      }
      __syncthreads();
      kk = rp_keff(P.K, s_cnt[0], s_cnt[1]);
    }
    const int rl = row - P.row_begin;
    int kp = 0;
    if (kk > 0) {
      kp = rp_select_topk<NTHR>(src, W, kk, sel);
      const double ri = P.roww ? P.roww[row] : 1.0;
      rp_select_emit<NTHR>(W, sel, [&](int pos, int w) {  // slot order: staged, then ranked by column
        ecol[pos] = keys[w];
        eval[pos] = src.value(w) * ri;
      });
      __syncthreads();
      for (int e = tid; e < kp; e += NTHR) {
        const uint32_t c = ecol[e];
        int rank = 0;
        for (int o = 0; o < kp; ++o) rank += ecol[o] < c;
        P.out_idx[(int64_t)rl * P.K + rank] = (int)c;
        P.out_val[(int64_t)rl * P.K + rank] = eval[e];
      }
    }
    if (tid == 0) P.out_nnz[rl] = kp;
    __syncthreads();
    for (int w = tid; w < W; w += NTHR) {
      acc[w] = 0.0;
      keys[w] = RP_VHASH_EMPTY;
      if (hermann) cntw[w] = 0u;
    }
  }
}

// light[rl] = the row's products (sum of its users' degrees) number at most RP_VHASH_MAX; one wave per row
__global__ void k_vrow_light(const int32_t* __restrict__ xt_ptr, const int32_t* __restrict__ xt_idx, const int32_t* __restrict__ x_ptr,
                             int row_begin, int n_rows, int32_t* __restrict__ light) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  for (int rl = blockIdx.x * wpb + (threadIdx.x >> 6); rl < n_rows; rl += gridDim.x * wpb) {
    const int i = row_begin + rl;
    const int b = xt_ptr[i], e = xt_ptr[i + 1];
    int ok = 0;
    if (e - b <= RP_VHASH_MAX) {
      int sum = 0;
      for (int p = b + lane; p < e; p += 64) {
        const int u = xt_idx[p];
        sum += x_ptr[u + 1] - x_ptr[u];
      }
      sum = rp_wave_sum(sum);
      ok = sum <= RP_VHASH_MAX;
    }
    if (lane == 0) light[rl] = ok;
  }
}

// stable partition of `order` (descending cost) into the tiled kernel's list and the hash kernel's list; one workgroup
__global__ void __launch_bounds__(1024) k_vrow_split(const int32_t* __restrict__ order, const int32_t* __restrict__ light, int row_begin,
                                                     int n_rows, int32_t* __restrict__ heavy_list, int32_t* __restrict__ light_list,
                                                     int32_t* __restrict__ counts) {
  __shared__ int s_h[16], s_l[16], s_base[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base[0] = s_base[1] = 0;
  __syncthreads();
  for (int k0 = 0; k0 < n_rows; k0 += 1024) {
    const int k = k0 + tid;
    const int row = (k < n_rows) ? order[k] : -1;
    const bool is_l = row >= 0 && light[row - row_begin] != 0, is_h = row >= 0 && !is_l;
    const uint64_t mh = __ballot(is_h), ml = __ballot(is_l);
    if (lane == 0) {
      s_h[wave] = __popcll(mh);
      s_l[wave] = __popcll(ml);
    }
    __syncthreads();
    int oh = s_base[0], ol = s_base[1];
    for (int w = 0; w < wave; ++w) {
      oh += s_h[w];
      ol += s_l[w];
    }
    const uint64_t lt = (1ull << lane) - 1ull;
    if (is_h) heavy_list[oh + __popcll(mh & lt)] = row;
    if (is_l) light_list[ol + __popcll(ml & lt)] = row;
    __syncthreads();
    if (tid == 0) {
      int th = 0, tl = 0;
      for (int w = 0; w < 16; ++w) {
        th += s_h[w];
        tl += s_l[w];
      }
      s_base[0] += th;
      s_base[1] += tl;
    }
    __syncthreads();
  }
  if (tid == 0) {
    counts[0] = s_base[0];
    counts[1] = s_base[1];
  }
}

// The two lookups the accumulate loop would otherwise repeat per (row, user, tile), done once per fit:
// tile pointers of every user row (rows are column-sorted) ...
__global__ void k_user_tile_ptr(const int32_t* __restrict__ x_ptr, const int32_t* __restrict__ x_idx, int64_t n_users, int T, int W,
                                int32_t* __restrict__ out) {
  const int64_t n = n_users * (T + 1);
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = g / (T + 1);
    const int t = (int)(g % (T + 1));
    int lo = x_ptr[u], hi = x_ptr[u + 1];
    const int64_t c0 = (int64_t)t * W;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (x_idx[mid] < c0) lo = mid + 1; else hi = mid;
    }
    out[g] = lo;
  }
}

// ... and the value of every X^T entry: entry p of item row i holds user u, its value is X[u, i] (found in the sorted row u)
__global__ void k_xt_values(const int32_t* __restrict__ xt_ptr, const int32_t* __restrict__ xt_idx, const int32_t* __restrict__ x_ptr,
                            const int32_t* __restrict__ x_idx, const double* __restrict__ x_val, int n_items, int64_t nnz,
                            double* __restrict__ out) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = n_items;  // the row i with xt_ptr[i] <= p < xt_ptr[i + 1]
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)xt_ptr[mid + 1] <= p) lo = mid + 1; else hi = mid;
    }
    const int i = lo;
    const int u = xt_idx[p];
    lo = x_ptr[u];
    hi = x_ptr[u + 1];
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (x_idx[mid] < i) lo = mid + 1; else hi = mid;
    }
    out[p] = x_val[lo];
  }
}

int rp_user_tile_ptr(rphip_ctx* ctx, const rphip_csr* X, int T, int W, int32_t* out) {
  const int64_t n = X->n_users * (int64_t)(T + 1);
  if (n <= 0) return RPHIP_OK;
  hipLaunchKernelGGL(k_user_tile_ptr, dim3((unsigned)std::min<int64_t>(65535, rp_cdiv(n, 256))), dim3(256), 0, ctx->stream, X->x_ptr, X->x_idx,
                     X->n_users, T, W, out);
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}

struct rp_vlist_src {
  const double* g_;
  const int32_t* col_;
  int n;
  __device__ __forceinline__ bool valid(int w) const { return w < n; }
  __device__ __forceinline__ uint64_t key(int w) const { return rp_ordered_f64(g_[w]); }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)col_[w]; }
};

// The per-tile lists of one row where they lie in memory (large K, round 5; rp_glist_src of the binary fit)
struct rp_vglist_src {
  const int32_t* idx_;   // the row's lists: tile t at idx_ + t * K
  const double* g_;
  const int* off;        // [T + 1] in LDS
  int T, K, n;
  __device__ __forceinline__ int at(int w) const {
    int t = 0;
    while (t + 1 < T && w >= off[t + 1]) ++t;
    return t * K + (w - off[t]);
  }
  __device__ __forceinline__ bool valid(int w) const { return w < n; }
  __device__ __forceinline__ uint64_t key(int w) const { return rp_ordered_f64(g_[at(w)]); }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)idx_[at(w)]; }
};

// top-K_eff of the T per-tile candidate lists of a row (each holds the tile's K largest non-zero entries); GLOBAL: the lists are
// not staged in LDS
template <bool GLOBAL>
__global__ void __launch_bounds__(RP_VMERGE_THREADS) k_itemknn_merge_values(const int32_t* __restrict__ cand_idx, const double* __restrict__ cand_g,
                                                                           const int32_t* __restrict__ cand_nnz, const int32_t* __restrict__ cand_cnt,
                                                                           const double* __restrict__ roww, int signed_vals, int n_tiles, int K,
                                                                           int n_rows, int row_begin, const int32_t* __restrict__ light,
                                                                           int32_t* __restrict__ out_idx, double* __restrict__ out_val,
                                                                           int32_t* __restrict__ out_nnz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = n_tiles, cap = T * K;
  double* lg = reinterpret_cast<double*>(smem);
  int32_t* lcol = reinterpret_cast<int32_t*>(smem + (size_t)cap * 8);
  rp_select_smem sel = rp_select_carve(smem + (GLOBAL ? 0 : (size_t)cap * 12 + ((cap & 1) ? 4 : 0)), cap, RP_VMERGE_THREADS / 64);
  int* s_off = &sel.misc[RPS_USER];  // [T + 1], then K_eff
  const int tid = threadIdx.x;
  for (int rl = blockIdx.x; rl < n_rows; rl += gridDim.x) {
    if (light && light[rl]) continue;  // done by the hash kernel
    __syncthreads();
    if (tid == 0) {
      int o = 0, np = 0;
      for (int t = 0; t < T; ++t) {
        s_off[t] = o;
        o += cand_nnz[(int64_t)rl * T + t];
        if (signed_vals) np += cand_cnt[(int64_t)rl * T + t];
      }
      s_off[T] = o;
      // negatives beyond a tile's K never matter: at most K - n_pos of them are kept in total
      s_off[T + 1] = signed_vals ? rp_keff(K, np, 0x3fffffff) : K;
    }
    __syncthreads();
    const int n = s_off[T];
    const int kk = min(s_off[T + 1], n);
    const double ri = roww ? roww[row_begin + rl] : 1.0;
    int kp = 0;
    if constexpr (GLOBAL) {
      const int32_t* gi = cand_idx + (int64_t)rl * T * K;
      const double* gg = cand_g + (int64_t)rl * T * K;
      rp_vglist_src src{gi, gg, s_off, T, K, n};
      if (kk > 0) {
        kp = rp_select_topk<RP_VMERGE_THREADS>(src, n, kk, sel);
        rp_select_emit<RP_VMERGE_THREADS>(n, sel, [&](int pos, int w) {
          const int a = src.at(w);
          out_idx[(int64_t)rl * K + pos] = gi[a];
          out_val[(int64_t)rl * K + pos] = gg[a] * ri;
        });
      }
    } else {
      for (int t = 0; t < T; ++t) {
        const int cn = s_off[t + 1] - s_off[t];
        const int64_t so = ((int64_t)rl * T + t) * K;
        for (int k = tid; k < cn; k += RP_VMERGE_THREADS) {
          lcol[s_off[t] + k] = cand_idx[so + k];
          lg[s_off[t] + k] = cand_g[so + k];
        }
      }
      __syncthreads();
      rp_vlist_src src{lg, lcol, n};
      if (kk > 0) {
        kp = rp_select_topk<RP_VMERGE_THREADS>(src, n, kk, sel);
        rp_select_emit<RP_VMERGE_THREADS>(n, sel, [&](int pos, int w) {
          out_idx[(int64_t)rl * K + pos] = lcol[w];
          out_val[(int64_t)rl * K + pos] = lg[w] * ri;
        });
      }
    }
    if (tid == 0) out_nnz[rl] = kp;
  }
}

// the same merge for K x tiles >= 32 768 (round 6: rp_merge_big.h; lists of `stride` entries per tile)
__global__ void __launch_bounds__(RP_MBIG_THREADS) k_itemknn_merge_values_big(const int32_t* __restrict__ cand_idx, const double* __restrict__ cand_g,
                                                                             const int32_t* __restrict__ cand_nnz, const int32_t* __restrict__ cand_cnt,
                                                                             const double* __restrict__ roww, int signed_vals, int n_tiles, int stride, int K,
                                                                             int n_rows, int row_begin, const int32_t* __restrict__ light,
                                                                             int32_t* __restrict__ out_idx, double* __restrict__ out_val,
                                                                             int32_t* __restrict__ out_nnz) {
  __shared__ rp_mbig_smem S;
  const int T = n_tiles;
  for (int rl = blockIdx.x; rl < n_rows; rl += gridDim.x) {
    if (light && light[rl]) continue;  // done by the hash kernel
    const int32_t* gi = cand_idx + (int64_t)rl * T * stride;
    const double* gg = cand_g + (int64_t)rl * T * stride;
    const int32_t* cn = cand_nnz + (int64_t)rl * T;
    int kk = K;
    if (signed_vals) {  // negatives beyond a tile's K never matter: at most K - n_pos - 1 of them are kept in total (rp_keff)
      int np = 0;
      for (int t = 0; t < T; ++t) np += cand_cnt[(int64_t)rl * T + t];
      kk = rp_keff(K, np, 0x3fffffff);
    }
    const double ri = roww ? roww[row_begin + rl] : 1.0;
    const int n_out = rp_merge_big_row(
        S, T, stride, kk, [&](int t) { return cn[t]; }, [&](int a) { return rp_ordered_f64(gg[a]); },
        [&](int pos, int a) {
          out_idx[(int64_t)rl * K + pos] = gi[a];
          out_val[(int64_t)rl * K + pos] = gg[a] * ri;
        });
    if (threadIdx.x == 0) out_nnz[rl] = n_out;
  }
}

static int rp_fit_values_impl(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K, int32_t left_binary,
                              const double* row_scale_host, int pair_kind, double pair_decay, double pair_max, rphip_model** out) {
  if (!ctx || !X || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (!X->x_val) RP_FAIL(RPHIP_ESTATE, "the matrix carries no values (rphip_csr_set_values)");
  RP_NEED_DEGREES(X);
  if (row_begin < 0 || row_end > X->n_items || row_begin > row_end)
    RP_FAIL(RPHIP_EINVAL, "row range [%lld,%lld) outside 0..%lld", (long long)row_begin, (long long)row_end, (long long)X->n_items);
  if (K < 1) RP_FAIL(RPHIP_EINVAL, "K must be >= 1 (got %d)", K);
  if (X->xt_row_end >= 0) RP_FAIL(RPHIP_ESTATE, "the valued fit needs a matrix with all item rows of X^T (not rphip_csr_from_device_rows)");
  RP_TRY(rp_set_device(ctx));
  const int32_t Keff = (int32_t)std::min<int64_t>(K, std::max<int64_t>(1, X->n_items - 1));
  K = Keff;
  const int n_rows = (int)(row_end - row_begin);
  const int64_t I = X->n_items;
  const int tile_max = (pair_kind == RPHIP_PAIR_HERMANN) ? RP_VTILE_HERMANN : RP_VTILE_MAX;
  const int T = (int)std::max<int64_t>(1, rp_cdiv(I, tile_max));
  // (round 5) any K on one column tile; over several tiles the merge keeps a 15-bit slot number per candidate (rp_select.h) ...
  // (round 6) ... and K x tiles >= 32 768 takes per-tile lists of min(K, tile) entries merged by rp_merge_big.h, the rows in chunks
  const bool big_k = T > 1 && (int64_t)T * Keff >= RPS_FLAG;
  const int W = (int)(rp_cdiv(rp_cdiv(std::max<int64_t>(I, 1), T), 64) * 64);
  const int Ks = big_k ? std::min<int>(Keff, W) : Keff;  // entries of a tile's candidate list
  if (T > 15) RP_FAIL(RPHIP_ELIMIT, "%d valued column tiles (n_items=%lld) exceed the supported 15", T, (long long)I);
  if ((T > 1 || !left_binary) && !X->sorted_rows) RP_FAIL(RPHIP_EINVAL, "the valued fit needs column-sorted rows");
  hipStream_t st = ctx->stream;
  rphip_model* m = nullptr;
  RP_TRY(rp_model_alloc_topk(ctx, I, row_begin, row_end, K, &m));
  const size_t nk = (size_t)std::max<int64_t>(1, (int64_t)n_rows * K);
  hipMemsetAsync(m->idx, 0xFF, nk * sizeof(int32_t), st);
  hipMemsetAsync(m->val, 0, nk * sizeof(double), st);
  hipMemsetAsync(m->nnz, 0, (size_t)std::max(1, n_rows) * sizeof(int32_t), st);
  auto fail = [&](int rc) {
    rphip_model_destroy(m);
    return rc;
  };
  // the rows [rb, rb + nr): plan, the tile (and hash) kernels, the merge of the per-tile lists; the lists go to oi / ov / on
  auto run_rows = [&](int64_t rb, int nr, int32_t* oi, double* ov, int32_t* on) -> int {
    char* sc = nullptr;
    const size_t cand_n = (T > 1) ? (size_t)nr * T * Ks : 0;
    const bool use_hash = T > 1 && K <= RP_VHASH_KMAX;  // light rows: one task per row in a hash table instead of T tiles
    const size_t vt_n = (T > 1) ? (size_t)X->n_users * (T + 1) : 0;
    const size_t xtv_n = left_binary ? 0 : (size_t)X->nnz;
    const size_t bytes = (size_t)I * 8 + RP_PLAN_BUCKETS * 4 + (size_t)nr * 4 + 64 + cand_n * 12 + (size_t)nr * T * 8 + vt_n * 4 +
                         xtv_n * 8 + (size_t)nr * 12 + 2048;
    int rc = rp_scratch(ctx, 0, bytes, (void**)&sc);
    if (rc != RPHIP_OK) return fail(rc);
    size_t o = 0;
    auto carve = [&](size_t b) {
      char* p = sc + o;
      o += (b + 15) & ~(size_t)15;
      return p;
    };
    double* roww = (double*)carve((size_t)I * 8);
    int32_t* bucket = (int32_t*)carve(RP_PLAN_BUCKETS * 4);
    int32_t* order = (int32_t*)carve((size_t)nr * 4);
    int32_t* ticket = (int32_t*)carve(64);
    double* cand_g = (double*)carve(cand_n * 8);
    int32_t* cand_idx = (int32_t*)carve(cand_n * 4);
    int32_t* cand_nnz = (int32_t*)carve((size_t)nr * T * 4);
    int32_t* cand_cnt = (int32_t*)carve((size_t)nr * T * 4);
    int32_t* light_flag = (int32_t*)carve((size_t)nr * 4);
    int32_t* heavy_list = (int32_t*)carve((size_t)nr * 4);
    int32_t* light_list = (int32_t*)carve((size_t)nr * 4);
    int32_t* list_counts = (int32_t*)carve(64);
    int32_t* vt_ptr = (int32_t*)carve(vt_n * 4);
    double* xt_val = (double*)carve(xtv_n * 8);
    hipEvent_t ev;
    rp_time_begin(ctx, &ev);
    hipMemsetAsync(ticket, 0, 64, st);
    if (T > 1) {
      hipMemsetAsync(cand_nnz, 0, (size_t)nr * T * 4, st);
      hipMemsetAsync(cand_cnt, 0, (size_t)nr * T * 4, st);
    }
    rc = rp_plan_row_order(ctx, X, rb, nr, bucket, order);
    if (rc != RPHIP_OK) return fail(rc);
    if (T > 1) {
      rc = rp_user_tile_ptr(ctx, X, T, W, vt_ptr);
      if (rc != RPHIP_OK) return fail(rc);
    }
    if (use_hash) {
      hipLaunchKernelGGL(k_vrow_light, dim3((unsigned)std::min<int64_t>(4096, rp_cdiv(nr, 4))), dim3(256), 0, st, X->xt_ptr, X->xt_idx,
                         X->x_ptr, (int)rb, nr, light_flag);
      hipLaunchKernelGGL(k_vrow_split, dim3(1), dim3(1024), 0, st, order, light_flag, (int)rb, nr, heavy_list, light_list,
                         list_counts);
    }
    if (!left_binary)
      hipLaunchKernelGGL(k_xt_values, dim3((unsigned)std::min<int64_t>(65535, rp_cdiv(X->nnz, 256))), dim3(256), 0, st, X->xt_ptr, X->xt_idx,
                         X->x_ptr, X->x_idx, X->x_val, (int)I, X->nnz, xt_val);
    if (row_scale_host && hipMemcpyAsync(roww, row_scale_host, (size_t)I * 8, hipMemcpyHostToDevice, st) != hipSuccess) return fail(RPHIP_EHIP);
    rp_time_end(ctx, ev, RPHIP_T_FIT_PLAN, 0.0);
    rp_vfit_params P;
    P.x_ptr = X->x_ptr;
    P.x_idx = X->x_idx;
    P.x_val = X->x_val;
    P.xt_ptr = X->xt_ptr;
    P.xt_idx = X->xt_idx;
    P.order = use_hash ? heavy_list : order;
    P.n_rows_dev = use_hash ? list_counts : nullptr;
    P.vt_ptr = vt_ptr;
    P.xt_val = xt_val;
    P.roww = row_scale_host ? roww : nullptr;
    P.left_binary = left_binary ? 1 : 0;
    P.signed_vals = (X->val_neg && !pair_kind) ? 1 : 0;  // decayed distances are never negative
    P.pair_kind = pair_kind;
    P.pair_decay = pair_decay;
    P.pair_max = pair_max;
    P.n_items = (int)I;
    P.n_rows = nr;
    P.n_tiles = T;
    P.tile_w = W;
    P.K = K;
    P.cand_stride = Ks;
    P.row_begin = (int)rb;
    P.ticket = ticket;
    P.out_idx = (T == 1) ? oi : cand_idx;
    P.out_val = (T == 1) ? ov : cand_g;
    P.out_nnz = (T == 1) ? on : cand_nnz;
    P.out_cnt = cand_cnt;
    const size_t lds = (size_t)W * (pair_kind == RPHIP_PAIR_HERMANN ? 12 : 8) + rp_select_smem_bytes(W);
    if (hipFuncSetAttribute((const void*)k_itemknn_fit_values, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(RPHIP_EHIP);
    rp_time_begin(ctx, &ev);
    hipLaunchKernelGGL(k_itemknn_fit_values, dim3(ctx->n_cu), dim3(RP_FIT_THREADS), lds, st, P);
    rp_time_end(ctx, ev, RPHIP_T_GRAM_TOPK, 0.0);
    if (use_hash) {
      rp_vfit_params H = P;  // writes the model rows directly
      H.out_idx = oi;
      H.out_val = ov;
      H.out_nnz = on;
      const size_t hlds = (size_t)RP_VHASH_SLOTS * (pair_kind == RPHIP_PAIR_HERMANN ? 16 : 12) + rp_select_smem_bytes(RP_VHASH_SLOTS);
      if (hipFuncSetAttribute((const void*)k_itemknn_fit_values_hash, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hlds) != hipSuccess)
        return fail(RPHIP_EHIP);
      rp_time_begin(ctx, &ev);
      hipLaunchKernelGGL(k_itemknn_fit_values_hash, dim3(ctx->n_cu), dim3(RP_FIT_THREADS), hlds, st, H, light_list, list_counts + 1);
      rp_time_end(ctx, ev, RPHIP_T_GRAM_TOPK_WAVE, 0.0);
    }
    if (T > 1 && big_k) {
      rp_time_begin(ctx, &ev);
      hipLaunchKernelGGL(k_itemknn_merge_values_big, dim3((unsigned)std::min<int64_t>(nr, (int64_t)ctx->n_cu * 8)), dim3(RP_MBIG_THREADS), 0, st, cand_idx, cand_g,
                         cand_nnz, cand_cnt, P.roww, P.signed_vals, T, Ks, K, nr, (int)rb, use_hash ? light_flag : nullptr, oi, ov, on);
      rp_time_end(ctx, ev, RPHIP_T_TOPK_MERGE, 0.0);
    } else if (T > 1) {
      size_t mlds = (size_t)T * K * 12 + 8 + rp_select_smem_bytes(T * K, RP_VMERGE_THREADS / 64);
      const bool glob = mlds > 160 * 1024;  // the lists of a row do not fit the LDS: they are read where they lie (slot numbers only in LDS)
      if (glob) mlds = rp_select_smem_bytes(T * K, RP_VMERGE_THREADS / 64);
      if (mlds > 160 * 1024 || T * K >= RPS_FLAG) {
        rp_set_error("K=%d with %d column tiles needs %zu bytes of LDS for the merge", K, T, mlds);
        return fail(RPHIP_ELIMIT);
      }
      const void* mk = glob ? (const void*)k_itemknn_merge_values<true> : (const void*)k_itemknn_merge_values<false>;
      if (hipFuncSetAttribute(mk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds) != hipSuccess) return fail(RPHIP_EHIP);
      const int mper = std::max<int>(1, std::min<int>(8, (int)((160 * 1024) / mlds)));
      const dim3 mgrid((unsigned)std::min<int64_t>(nr, (int64_t)ctx->n_cu * mper));
      rp_time_begin(ctx, &ev);
      if (glob)
        hipLaunchKernelGGL(k_itemknn_merge_values<true>, mgrid, dim3(RP_VMERGE_THREADS), mlds, st, cand_idx, cand_g, cand_nnz, cand_cnt, P.roww, P.signed_vals, T, K,
                           nr, (int)rb, use_hash ? light_flag : nullptr, oi, ov, on);
      else
        hipLaunchKernelGGL(k_itemknn_merge_values<false>, mgrid, dim3(RP_VMERGE_THREADS), mlds, st, cand_idx, cand_g, cand_nnz, cand_cnt, P.roww, P.signed_vals, T, K,
                           nr, (int)rb, use_hash ? light_flag : nullptr, oi, ov, on);
      rp_time_end(ctx, ev, RPHIP_T_TOPK_MERGE, 0.0);
    }
    if (hipGetLastError() != hipSuccess) return fail(RPHIP_EHIP);
      return RPHIP_OK;
  };
  if (n_rows > 0 && X->nnz > 0) {
    // in chunks where the per-tile lists of all rows at once would be too much memory (T x min(K, tile) x 12 bytes per row)
    const int64_t per_row = T > 1 ? (int64_t)T * Ks * 12 + (int64_t)T * 8 : 0;
    const int64_t budget = (int64_t)6 << 30;
    const int64_t R = (per_row > 0 && per_row * n_rows > budget) ? std::max<int64_t>(1, budget / per_row) : n_rows;
    for (int64_t c0 = 0; c0 < n_rows; c0 += R) {
      const int rc = run_rows(row_begin + c0, (int)std::min<int64_t>(R, n_rows - c0), m->idx + c0 * K, m->val + c0 * K, m->nnz + c0);
      if (rc != RPHIP_OK) return rc;  // (run_rows has released the model)
    }
  }
  int r = rp_model_finalize_topk(ctx, m);
  if (r != RPHIP_OK) return fail(r);
  *out = m;
  return RPHIP_OK;
}

extern "C" int rphip_itemknn_fit_values(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K, int32_t left_binary,
                                        const double* row_scale_host, rphip_model** out) {
  RP_ENTER(ctx);
  return rp_fit_values_impl(ctx, X, row_begin, row_end, K, left_binary, row_scale_host, 0, 0.0, 0.0, out);
}

extern "C" int rphip_itemknn_fit_timedist(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K, int32_t decay_kind,
                                          double decay, double max_distance, const double* row_scale_host, rphip_model** out) {
  RP_ENTER(ctx);
  if (decay_kind < RPHIP_DECAY_NONE || decay_kind > RPHIP_DECAY_INVERSE) RP_FAIL(RPHIP_EINVAL, "unknown decay function %d", decay_kind);
  if ((decay_kind == RPHIP_DECAY_CONCAVE || decay_kind == RPHIP_DECAY_LOG || decay_kind == RPHIP_DECAY_LINEAR) && !(max_distance > 0.0))
    RP_FAIL(RPHIP_EINVAL, "the %s decay needs max_distance > 0 (got %g)", decay_kind == RPHIP_DECAY_LOG ? "log" : decay_kind == RPHIP_DECAY_LINEAR ? "linear" : "concave",
            max_distance);
  return rp_fit_values_impl(ctx, X, row_begin, row_end, K, 0, row_scale_host, decay_kind, decay, max_distance, out);
}

extern "C" int rphip_itemknn_fit_hermann(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K, double now,
                                         rphip_model** out) {
  RP_ENTER(ctx);
  if (X && X->x_val && !(now > X->val_max_abs)) RP_FAIL(RPHIP_EINVAL, "now = %g must lie after every timestamp (max %g)", now, X->val_max_abs);
  return rp_fit_values_impl(ctx, X, row_begin, row_end, K, 0, nullptr, RPHIP_PAIR_HERMANN, 0.0, now, out);
}

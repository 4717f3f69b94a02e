// ItemKNN with normalize_X = True: the weighted (float64) Gram variant of K1-K3.
#include <algorithm>

#include "rp_common.h"
#include "rp_merge_big.h"
#include "rp_select.h"
#include "rp_wave.h"

constexpr int RP_MERGE_THREADS = 256;

// ---- normalize_X: weighted Gram (nearest_neighbour.py:207-210) ------------------------------------------------------
// With normalize_X the binarised rows are l1-normalised first, X[u, i] = w_u = 1/d_u, so the Gram entry of (i, j) is a
// float64 sum over the common users: G_ij = sum w_u^2 (cosine, :69-84: S_ij = G_ij / sqrt(G_ii G_jj)) or sum w_u
// (conditional probability, :22-66: to_binary(X).T @ X, S_ij = G_ij / (c_i c_j^alpha) with the binary frequencies c).
// The tile holds doubles (LDS f64 atomic adds, half the columns of the integer tile), there is no byte-counter or
// slicing path: this variant is about coverage, not the roofline.  The float sums depend on the order of the adds in the
// last bits (as they do in scipy); parity is the tie-aware rule with eps = 1e-9 and values at 1e-11.
constexpr int RP_WTILE_MAX = 12288;

struct rp_wfit_params {
  const int32_t* x_ptr;
  const int32_t* x_idx;
  const int32_t* xt_ptr;
  const int32_t* xt_idx;
  const int32_t* order;    // rows by descending degree
  const int32_t* vt_ptr;   // T > 1: [U * (T + 1)] tile pointers of the user rows
  const double* user_q;    // [U] what one common user adds
  const double* colw;      // [I] key = G_ij * colw_j (monotone in S_ij for a fixed row)
  const double* roww;      // [I] S_ij = key * roww_i
  int n_items, n_rows, n_tiles, tile_w, K, row_begin;
  int cand_stride;         // T > 1: entries between two tiles' lists of a row (K; min(K, tile) on the large-K route)
  int32_t* ticket;
  int32_t* out_idx;        // T == 1: model rows; T > 1: per-tile candidates
  double* out_val;         //         values        ;        G_ij of the candidates
  int32_t* out_nnz;
};

struct rp_wdense_src {
  const double* acc;
  const double* colw;
  int t0, row;
  int sh;  // leading key bits shared by all candidates (rp_key_prefix, rp_wave.h)
  __device__ __forceinline__ bool valid(int w) const { return acc[w] != 0.0 && (t0 + w) != row; }
  __device__ __forceinline__ uint64_t key(int w) const { return (uint64_t)__double_as_longlong(acc[w] * colw[t0 + w]) << sh; }  // > 0
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)(t0 + w); }
};

__global__ void __launch_bounds__(RP_FIT_THREADS) k_itemknn_fit_weighted(rp_wfit_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTHR = RP_FIT_THREADS, NW = NTHR / 64;
  double* acc = reinterpret_cast<double*>(smem);
  const int W = P.tile_w, T = P.n_tiles;
  rp_select_smem sel = rp_select_carve(smem + (size_t)W * 8, W, NW);
  int* s_task = &sel.misc[RPS_USER];
  uint32_t* s_red = reinterpret_cast<uint32_t*>(&sel.misc[RPS_USER + 4]);  // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave * 4 + (lane >> 4), gl = lane & 15;  // 64 groups of 16 lanes: one user per group and round
  for (int w = tid; w < W; w += NTHR) acc[w] = 0.0;
  const long long n_tasks = (long long)P.n_rows * T;
  while (true) {
    __syncthreads();
    if (tid == 0) {
      s_task[0] = atomicAdd(P.ticket, 1);
      s_red[0] = s_red[1] = 0xffffffffu;
      s_red[2] = s_red[3] = 0u;
    }
    __syncthreads();
    const long long task = s_task[0];
    if (task >= n_tasks) break;
    const int row = P.order[task / T], tile = (int)(task % T);
    const int t0 = tile * W, n_cols = min(W, P.n_items - t0);
    // ---- accumulate
    // four users per 16-lane group and round, the loads of one kind issued together (see rp_itemknn_v.hip)
    const int pe = P.xt_ptr[row + 1];
    constexpr int UNR = 4;
    for (int p0 = P.xt_ptr[row] + grp; p0 < pe; p0 += 64 * UNR) {
      int u[UNR], s[UNR], e[UNR], c[UNR];
      double q[UNR];
#pragma unroll
      for (int r = 0; r < UNR; ++r) {
        const int p = p0 + 64 * r;
        u[r] = (p < pe) ? P.xt_idx[p] : -1;
      }
#pragma unroll
      for (int r = 0; r < UNR; ++r) {
        s[r] = e[r] = 0;
        q[r] = 0.0;
        if (u[r] >= 0) {
          q[r] = P.user_q[u[r]];
          if (T > 1) {  // the slice of the (column-sorted) row that falls into this tile
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
      }
#pragma unroll
      for (int r = 0; r < UNR; ++r)
        if (c[r] >= 0) atomicAdd(&acc[c[r] - t0], q[r]);
#pragma unroll
      for (int r = 0; r < UNR; ++r)
        for (int k = s[r] + gl + 16; k < e[r]; k += 16) atomicAdd(&acc[P.x_idx[k] - t0], q[r]);
    }
    __syncthreads();
    // ---- select
    rp_wdense_src src{acc, P.colw, t0, row, 0};
    src.sh = rp_key_prefix<NTHR>(src, n_cols, s_red);
    const int kp = rp_select_topk<NTHR>(src, n_cols, P.K, sel);
    const int rl = row - P.row_begin;
    if (T == 1) {
      const double ri = P.roww[row];
      rp_select_emit<NTHR>(n_cols, sel, [&](int pos, int w) {
        P.out_idx[(int64_t)rl * P.K + pos] = t0 + w;
        P.out_val[(int64_t)rl * P.K + pos] = (acc[w] * P.colw[t0 + w]) * ri;
      });
      if (tid == 0) P.out_nnz[rl] = kp;
    } else {
      const int64_t o = ((int64_t)rl * T + tile) * P.cand_stride;
      rp_select_emit<NTHR>(n_cols, sel, [&](int pos, int w) {
        P.out_idx[o + pos] = t0 + w;
        P.out_val[o + pos] = acc[w];
      });
      if (tid == 0) P.out_nnz[(int64_t)rl * T + tile] = kp;
    }
    rp_select_foreach<NTHR>(n_cols, sel, [&](int w) { acc[w] = 0.0; });
    if (tid == 0 && (unsigned)(row - t0) < (unsigned)n_cols) acc[row - t0] = 0.0;
  }
}

struct rp_wlist_src {
  const int32_t* col_;
  const double* g_;
  const double* colw;
  int n;
  __device__ __forceinline__ bool valid(int w) const { return w < n; }
  __device__ __forceinline__ uint64_t key(int w) const { return (uint64_t)__double_as_longlong(g_[w] * colw[col_[w]]); }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)col_[w]; }
};

// The per-tile lists of one row where they lie in memory (large K, round 5; rp_glist_src of the binary fit): slot w of the row's
// concatenated lists is entry w - off[t] of tile t's list; only the selection's own list of slot numbers is in LDS.
struct rp_wglist_src {
  const int32_t* idx_;   // the row's lists: tile t at idx_ + t * K
  const double* g_;
  const int* off;        // [T + 1] in LDS
  const double* colw;
  int T, K, n;
  __device__ __forceinline__ int at(int w) const {
    int t = 0;
    while (t + 1 < T && w >= off[t + 1]) ++t;
    return t * K + (w - off[t]);
  }
  __device__ __forceinline__ bool valid(int w) const { return w < n; }
  __device__ __forceinline__ uint64_t key(int w) const {
    const int a = at(w);
    return (uint64_t)__double_as_longlong(g_[a] * colw[idx_[a]]);
  }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)idx_[at(w)]; }
};

// top-K of the T per-tile candidate lists of a row (weighted variant of k_itemknn_merge; GLOBAL: the lists are not staged in LDS)
template <bool GLOBAL>
__global__ void __launch_bounds__(RP_MERGE_THREADS) k_itemknn_merge_weighted(const int32_t* __restrict__ cand_idx, const double* __restrict__ cand_g,
                                                                            const int32_t* __restrict__ cand_nnz, const double* __restrict__ colw,
                                                                            const double* __restrict__ roww, int n_tiles, int K, int n_rows,
                                                                            int row_begin, int32_t* __restrict__ out_idx,
                                                                            double* __restrict__ out_val, int32_t* __restrict__ out_nnz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = n_tiles, cap = T * K;
  double* lg = reinterpret_cast<double*>(smem);
  int32_t* lcol = reinterpret_cast<int32_t*>(smem + (size_t)cap * 8);
  rp_select_smem sel = rp_select_carve(smem + (GLOBAL ? 0 : (size_t)cap * 12 + ((cap & 1) ? 4 : 0)), cap, RP_MERGE_THREADS / 64);
  int* s_off = &sel.misc[RPS_USER];
  const int tid = threadIdx.x;
  for (int rl = blockIdx.x; rl < n_rows; rl += gridDim.x) {
    __syncthreads();
    if (tid == 0) {
      int o = 0;
      for (int t = 0; t < T; ++t) {
        s_off[t] = o;
        o += cand_nnz[(int64_t)rl * T + t];
      }
      s_off[T] = o;
    }
    __syncthreads();
    const int n = s_off[T];
    const double ri = roww[row_begin + rl];
    if constexpr (GLOBAL) {
      const int32_t* gi = cand_idx + (int64_t)rl * T * K;
      const double* gg = cand_g + (int64_t)rl * T * K;
      rp_wglist_src src{gi, gg, s_off, colw, T, K, n};
      const int kp = rp_select_topk<RP_MERGE_THREADS>(src, n, K, sel);
      rp_select_emit<RP_MERGE_THREADS>(n, sel, [&](int pos, int w) {
        const int a = src.at(w);
        const int j = gi[a];
        out_idx[(int64_t)rl * K + pos] = j;
        out_val[(int64_t)rl * K + pos] = (gg[a] * colw[j]) * ri;
      });
      if (tid == 0) out_nnz[rl] = kp;
    } else {
      for (int t = 0; t < T; ++t) {
        const int cn = s_off[t + 1] - s_off[t];
        const int64_t so = ((int64_t)rl * T + t) * K;
        for (int k = tid; k < cn; k += RP_MERGE_THREADS) {
          lcol[s_off[t] + k] = cand_idx[so + k];
          lg[s_off[t] + k] = cand_g[so + k];
        }
      }
      __syncthreads();
      rp_wlist_src src{lcol, lg, colw, n};
      const int kp = rp_select_topk<RP_MERGE_THREADS>(src, n, K, sel);
      rp_select_emit<RP_MERGE_THREADS>(n, sel, [&](int pos, int w) {
        const int j = lcol[w];
        out_idx[(int64_t)rl * K + pos] = j;
        out_val[(int64_t)rl * K + pos] = (lg[w] * colw[j]) * ri;
      });
      if (tid == 0) out_nnz[rl] = kp;
    }
  }
}

// the same merge for K x tiles >= 32 768 (round 6: rp_merge_big.h; lists of `stride` entries per tile)
__global__ void __launch_bounds__(RP_MBIG_THREADS) k_itemknn_merge_weighted_big(const int32_t* __restrict__ cand_idx, const double* __restrict__ cand_g,
                                                                               const int32_t* __restrict__ cand_nnz, const double* __restrict__ colw,
                                                                               const double* __restrict__ roww, int n_tiles, int stride, int K, int n_rows,
                                                                               int row_begin, int32_t* __restrict__ out_idx, double* __restrict__ out_val,
                                                                               int32_t* __restrict__ out_nnz) {
  __shared__ rp_mbig_smem S;
  const int T = n_tiles;
  for (int rl = blockIdx.x; rl < n_rows; rl += gridDim.x) {
    const int32_t* gi = cand_idx + (int64_t)rl * T * stride;
    const double* gg = cand_g + (int64_t)rl * T * stride;
    const int32_t* cn = cand_nnz + (int64_t)rl * T;
    const double ri = roww[row_begin + rl];
    const int n_out = rp_merge_big_row(
        S, T, stride, K, [&](int t) { return cn[t]; }, [&](int a) { return (uint64_t)__double_as_longlong(gg[a] * colw[gi[a]]); },
        [&](int pos, int a) {
          const int j = gi[a];
          out_idx[(int64_t)rl * K + pos] = j;
          out_val[(int64_t)rl * K + pos] = (gg[a] * colw[j]) * ri;
        });
    if (threadIdx.x == 0) out_nnz[rl] = n_out;
  }
}

// q_u = w_u^2 (cosine) or w_u (conditional probability), w_u = fl(1 / d_u)
__global__ void k_user_weights(const int32_t* __restrict__ x_ptr, int n_users, int squared, double* __restrict__ q) {
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n_users; u += gridDim.x * blockDim.x) {
    const int d = x_ptr[u + 1] - x_ptr[u];
    const double w = d > 0 ? 1.0 / (double)d : 0.0;
    q[u] = squared ? w * w : w;
  }
}

// cosine: G_ii = sum of q_u over the item's users (one wave per item, fixed order), colw_i = 1 / sqrt(G_ii)
__global__ void k_weighted_norms(const int32_t* __restrict__ xt_ptr, const int32_t* __restrict__ xt_idx, const double* __restrict__ q,
                                 int n_items, double* __restrict__ colw) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int i = blockIdx.x * wpb + (threadIdx.x >> 6); i < n_items; i += gridDim.x * wpb) {
    double s = 0.0;
    for (int p = xt_ptr[i] + lane; p < xt_ptr[i + 1]; p += 64) s += q[xt_idx[p]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) colw[i] = s > 0.0 ? 1.0 / sqrt(s) : 0.0;
  }
}

int rp_gram_weighted(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K, int32_t similarity,
                            const double* cp_colw_host, rphip_model** out) {
  if (row_begin < 0 || row_end > X->n_items || row_begin > row_end)
    RP_FAIL(RPHIP_EINVAL, "row range [%lld,%lld) outside 0..%lld", (long long)row_begin, (long long)row_end, (long long)X->n_items);
  if (K < 1) RP_FAIL(RPHIP_EINVAL, "K must be >= 1 (got %d)", K);
  RP_NEED_DEGREES(X);
  if (X->xt_row_end >= 0) RP_FAIL(RPHIP_ESTATE, "normalize_X needs a matrix with all item rows of X^T (not rphip_csr_from_device_rows)");
  const int32_t Keff = (int32_t)std::min<int64_t>(K, std::max<int64_t>(1, X->n_items - 1));
  K = Keff;
  const int n_rows = (int)(row_end - row_begin);
  const int64_t I = X->n_items, U = X->n_users;
  const int T = (int)std::max<int64_t>(1, rp_cdiv(I, RP_WTILE_MAX));
  // (round 5) any K on one column tile; over several tiles the merge keeps a 15-bit slot number per candidate (rp_select.h) ...
  // (round 6) ... and K x tiles >= 32 768 takes per-tile lists of min(K, tile) entries merged by rp_merge_big.h, the rows in chunks
  const bool big_k = T > 1 && (int64_t)T * Keff >= RPS_FLAG;
  const int W = (int)(rp_cdiv(rp_cdiv(std::max<int64_t>(I, 1), T), 64) * 64);
  const int Ks = big_k ? std::min<int>(Keff, W) : Keff;  // entries of a tile's candidate list
  if (T > 15) RP_FAIL(RPHIP_ELIMIT, "%d weighted column tiles (n_items=%lld) exceed the supported 15", T, (long long)I);
  if (T > 1 && !X->sorted_rows) RP_FAIL(RPHIP_EINVAL, "normalize_X with more than %d items needs column-sorted rows", RP_WTILE_MAX);
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
  // the rows [rb, rb + nr): plan, the tile kernel, the merge of the per-tile lists; the lists go to oi / ov / on (the first row's place)
  auto run_rows = [&](int64_t rb, int nr, int32_t* oi, double* ov, int32_t* on) -> int {
    char* sc = nullptr;
    const size_t cand_n = (T > 1) ? (size_t)nr * T * Ks : 0;
    const size_t vt_n = (T > 1) ? (size_t)U * (T + 1) : 0;
    const size_t bytes = (size_t)U * 8 + (size_t)I * 8 + RP_PLAN_BUCKETS * 4 + (size_t)nr * 4 + 64 + cand_n * 12 + (size_t)nr * T * 4 +
                         vt_n * 4 + 512;
    int rc = rp_scratch(ctx, 0, bytes, (void**)&sc);
    if (rc != RPHIP_OK) return fail(rc);
    size_t o = 0;
    auto carve = [&](size_t b) {
      char* p = sc + o;
      o += (b + 15) & ~(size_t)15;
      return p;
    };
    double* user_q = (double*)carve((size_t)U * 8);
    double* colw = (double*)carve((size_t)I * 8);
    int32_t* bucket = (int32_t*)carve(RP_PLAN_BUCKETS * 4);
    int32_t* order = (int32_t*)carve((size_t)nr * 4);
    int32_t* ticket = (int32_t*)carve(64);
    double* cand_g = (double*)carve(cand_n * 8);
    int32_t* cand_idx = (int32_t*)carve(cand_n * 4);
    int32_t* cand_nnz = (int32_t*)carve((size_t)nr * T * 4);
    int32_t* vt_ptr = (int32_t*)carve(vt_n * 4);
    hipEvent_t ev;
    rp_time_begin(ctx, &ev);
    hipMemsetAsync(ticket, 0, 64, st);
    if (T > 1) hipMemsetAsync(cand_nnz, 0, (size_t)nr * T * 4, st);
    rc = rp_plan_row_order(ctx, X, rb, nr, bucket, order);
    if (rc != RPHIP_OK) return fail(rc);
    if (T > 1) {
      rc = rp_user_tile_ptr(ctx, X, T, W, vt_ptr);
      if (rc != RPHIP_OK) return fail(rc);
    }
    hipLaunchKernelGGL(k_user_weights, dim3((unsigned)std::min<int64_t>(2048, rp_cdiv(U, 256))), dim3(256), 0, st, X->x_ptr, (int)U,
                       similarity == RPHIP_SIM_COSINE ? 1 : 0, user_q);
    const double* roww = nullptr;
    if (similarity == RPHIP_SIM_COSINE) {
      hipLaunchKernelGGL(k_weighted_norms, dim3((unsigned)std::min<int64_t>(4096, rp_cdiv(I, 4))), dim3(256), 0, st, X->xt_ptr, X->xt_idx, user_q,
                         (int)I, colw);
      roww = colw;
    } else {
      if (hipMemcpyAsync(colw, cp_colw_host, (size_t)I * 8, hipMemcpyHostToDevice, st) != hipSuccess) return fail(RPHIP_EHIP);
      roww = X->inv_deg;
    }
    rp_time_end(ctx, ev, RPHIP_T_FIT_PLAN, 0.0);
    rp_wfit_params P;
    P.x_ptr = X->x_ptr;
    P.x_idx = X->x_idx;
    P.xt_ptr = X->xt_ptr;
    P.xt_idx = X->xt_idx;
    P.order = order;
    P.vt_ptr = vt_ptr;
    P.user_q = user_q;
    P.colw = colw;
    P.roww = roww;
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
    const size_t lds = (size_t)W * 8 + rp_select_smem_bytes(W);
    if (hipFuncSetAttribute((const void*)k_itemknn_fit_weighted, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(RPHIP_EHIP);
    rp_time_begin(ctx, &ev);
    hipLaunchKernelGGL(k_itemknn_fit_weighted, dim3(ctx->n_cu), dim3(RP_FIT_THREADS), lds, st, P);
    rp_time_end(ctx, ev, RPHIP_T_GRAM_TOPK, 0.0);
    if (T > 1 && big_k) {
      rp_time_begin(ctx, &ev);
      hipLaunchKernelGGL(k_itemknn_merge_weighted_big, dim3((unsigned)std::min<int64_t>(nr, (int64_t)ctx->n_cu * 8)), dim3(RP_MBIG_THREADS), 0, st, cand_idx, cand_g,
                         cand_nnz, colw, roww, T, Ks, K, nr, (int)rb, oi, ov, on);
      rp_time_end(ctx, ev, RPHIP_T_TOPK_MERGE, 0.0);
    } else if (T > 1) {
      size_t mlds = (size_t)T * K * 12 + 8 + rp_select_smem_bytes(T * K, RP_MERGE_THREADS / 64);
      const bool glob = mlds > 160 * 1024;  // the lists of a row do not fit the LDS: they are read where they lie (slot numbers only in LDS)
      if (glob) mlds = rp_select_smem_bytes(T * K, RP_MERGE_THREADS / 64);
      if (mlds > 160 * 1024 || T * K >= RPS_FLAG) {
        rp_set_error("K=%d with %d column tiles needs %zu bytes of LDS for the merge", K, T, mlds);
        return fail(RPHIP_ELIMIT);
      }
      const void* mk = glob ? (const void*)k_itemknn_merge_weighted<true> : (const void*)k_itemknn_merge_weighted<false>;
      if (hipFuncSetAttribute(mk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds) != hipSuccess) return fail(RPHIP_EHIP);
      const int mper = std::max<int>(1, std::min<int>(8, (int)((160 * 1024) / mlds)));
      rp_time_begin(ctx, &ev);
      const dim3 mgrid((unsigned)std::min<int64_t>(nr, (int64_t)ctx->n_cu * mper));
      if (glob)
        hipLaunchKernelGGL(k_itemknn_merge_weighted<true>, mgrid, dim3(RP_MERGE_THREADS), mlds, st, cand_idx, cand_g, cand_nnz, colw, roww, T, K, nr,
                           (int)rb, oi, ov, on);
      else
        hipLaunchKernelGGL(k_itemknn_merge_weighted<false>, mgrid, dim3(RP_MERGE_THREADS), mlds, st, cand_idx, cand_g, cand_nnz, colw, roww, T, K, nr,
                           (int)rb, oi, ov, on);
      rp_time_end(ctx, ev, RPHIP_T_TOPK_MERGE, 0.0);
    }
    if (hipGetLastError() != hipSuccess) return fail(RPHIP_EHIP);
      return RPHIP_OK;
  };
  if (n_rows > 0 && X->nnz > 0) {
    // in chunks where the per-tile lists of all rows at once would be too much memory (T x min(K, tile) x 12 bytes per row)
    const int64_t per_row = T > 1 ? (int64_t)T * Ks * 12 + (int64_t)T * 4 : 0;
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


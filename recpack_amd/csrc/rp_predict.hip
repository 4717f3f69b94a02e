// K8: predict for neighbour-list models: scores = X @ S with a fused per-user top-Kp.
//
// Replaces ItemSimilarityMatrixAlgorithm._predict (recpack/algorithms/base.py:237-255, `X @ similarity_matrix_`,
// scipy csr_matmat on one thread).  With Kp the kernel also applies what the reference does to the scores
// afterwards: optional history removal (recpack/pipelines/pipeline.py:174-175) and the per-user top-K of
// MetricTopK (recpack/metrics/base.py:189 -> recpack/util.py:50-77).  The U x I score matrix is never stored.
//
//   task = (user u, column tile t)
//   accumulate  for every history item i of u: the <= K (j, s_ij) pairs of S row i (contiguous, coalesced)
//               are added into a dense f32 tile in LDS with LDS float atomics; one wave per history item
//   select      exact top-Kp by (score desc, column asc)  (rp_select.h), emitted in column order
//
// The "full" variants return every non-zero score (the reference's CSR result) in two phases: count, fill.
#include <algorithm>

#include "rp_common.h"
#include "rp_select.h"
#include "rp_wave.h"

constexpr int RP_PRED_THREADS = 1024;

struct rp_pred_params {
  const int32_t* x_ptr;   // prediction input (binary CSR)
  const int32_t* x_idx;
  const int32_t* s_idx;   // model: [I*K]
  const float* s_val;
  const int32_t* s_nnz;
  const float* dense_B;   // K9: dense model (EASE), row-major with leading dimension dense_ld; NULL for neighbour lists
  long long dense_ld;
  int n_items, n_tiles, tile_w, K, Kp, exclude_history;
  int user_begin, n_users;
  const int32_t* user_list;  // optional: local user ids to process (mode 0); count in *n_list
  const int32_t* n_list;
  int mode;               // 0 = top-Kp, 1 = count, 2 = fill
  int32_t* out_idx;       // mode 0, T==1: [n*Kp]; T>1: candidates [n*T*Kp]
  float* out_val;
  int32_t* out_nnz;       // [n] or [n*T]
  long long* row_nnz;     // mode 1: [n]
  const long long* indptr;  // mode 2: [n+1]
  int32_t* full_idx;      // mode 2
  double* full_val;
};

struct rp_score_src {
  static constexpr bool PACK4 = false;
  const float* acc;
  int t0;
  __device__ __forceinline__ bool valid(int w) const { return acc[w] != 0.0f; }
  __device__ __forceinline__ uint64_t key(int w) const {
    // order-preserving map of a float (any sign) to u32, placed in the high half
    uint32_t b = __float_as_uint(acc[w]);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return (uint64_t)b << 32;
  }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)(t0 + w); }
};

__global__ void __launch_bounds__(RP_PRED_THREADS) k_predict_sparse(rp_pred_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* acc = reinterpret_cast<float*>(smem);
  rp_select_smem sel = rp_select_carve(smem + (size_t)P.tile_w * 4, P.tile_w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = RP_PRED_THREADS / 64;
  const int W = P.tile_w, T = P.n_tiles, K = P.K;

  for (int w = tid; w < W; w += RP_PRED_THREADS) acc[w] = 0.0f;
  __syncthreads();

  // modes 1/2 walk the tiles of a user inside the workgroup (output offsets are sequential);
  // mode 0 treats (user, tile) as independent tasks.
  const int n_u = P.user_list ? *P.n_list : P.n_users;
  const long long n_tasks = (P.mode == 0) ? (long long)n_u * T : (long long)n_u;
  for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
    const int uk = (P.mode == 0) ? (int)(task / T) : (int)task;
    const int ul = P.user_list ? P.user_list[uk] : uk;
    const int u = P.user_begin + ul;
    const int hs = P.x_ptr[u], he = P.x_ptr[u + 1];
    long long off = (P.mode == 2) ? P.indptr[ul] : 0;
    long long total = 0;
    const int t_first = (P.mode == 0) ? (int)(task % T) : 0;
    const int t_last = (P.mode == 0) ? t_first + 1 : T;
    for (int tile = t_first; tile < t_last; ++tile) {
      const int t0 = tile * W;
      const int n_cols = min(W, P.n_items - t0);
      // ---- accumulate
      if (P.dense_B) {
        // K9: score[u, t0:t0+n_cols] = sum of the history items' rows of B; every thread owns CPT columns
        // (registers, no atomics), rows are streamed coalesced
        constexpr int CPT = (RP_TILE_MAX + RP_PRED_THREADS - 1) / RP_PRED_THREADS;
        float r[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) r[c] = 0.0f;
        for (int h = hs; h < he; ++h) {
          const float* Bi = P.dense_B + (long long)P.x_idx[h] * P.dense_ld + t0;
#pragma unroll
          for (int c = 0; c < CPT; ++c) {
            const int w = tid + c * RP_PRED_THREADS;
            if (w < n_cols) r[c] += Bi[w];
          }
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const int w = tid + c * RP_PRED_THREADS;
          if (w < n_cols) acc[w] = r[c];
        }
      } else {
        for (int h = hs + wave; h < he; h += NW) {
          const int i = P.x_idx[h];
          const int nn = P.s_nnz[i];
          const int32_t* si = P.s_idx + (int64_t)i * K;
          const float* sv = P.s_val + (int64_t)i * K;
          for (int k = lane; k < nn; k += 64) {
            const int j = si[k] - t0;
            if ((unsigned)j < (unsigned)n_cols) atomicAdd(&acc[j], sv[k]);
          }
        }
      }
      __syncthreads();
      if (P.exclude_history) {
        for (int h = hs + tid; h < he; h += RP_PRED_THREADS) {
          const int j = P.x_idx[h] - t0;
          if ((unsigned)j < (unsigned)n_cols) acc[j] = 0.0f;
        }
        __syncthreads();
      }
      // ---- select
      rp_score_src src{acc, t0};
      const int kk = (P.mode == 0) ? P.Kp : 0x7fffffff;
      const int kp = rp_select_topk<RP_PRED_THREADS>(src, n_cols, kk, sel);
      if (P.mode == 0) {
        const int64_t o = (T == 1) ? (int64_t)ul * P.Kp : ((int64_t)ul * T + tile) * P.Kp;
        rp_select_emit<RP_PRED_THREADS>(n_cols, sel, [&](int pos, int w) {
          P.out_idx[o + pos] = t0 + w;
          P.out_val[o + pos] = acc[w];
        });
        if (tid == 0) P.out_nnz[(T == 1) ? ul : ul * T + tile] = kp;
      } else if (P.mode == 2) {
        rp_select_emit<RP_PRED_THREADS>(n_cols, sel, [&](int pos, int w) {
          P.full_idx[off + pos] = t0 + w;
          P.full_val[off + pos] = (double)acc[w];
        });
        off += kp;
      }
      total += kp;
      rp_select_foreach<RP_PRED_THREADS>(n_cols, sel, [&](int w) { acc[w] = 0.0f; });  // clear what was touched
      __syncthreads();
    }
    if (P.mode == 1 && tid == 0) P.row_nnz[ul] = total;
  }
}

// ---- merge of per-tile top-Kp candidates -----------------------------------------------------------
struct rp_flist_src {
  static constexpr bool PACK4 = false;
  const int32_t* col_;
  const float* val_;
  int n;
  __device__ __forceinline__ bool valid(int w) const { return w < n; }
  __device__ __forceinline__ uint64_t key(int w) const {
    uint32_t b = __float_as_uint(val_[w]);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return (uint64_t)b << 32;
  }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)col_[w]; }
};

__global__ void __launch_bounds__(RP_PRED_THREADS) k_predict_merge(const int32_t* __restrict__ cand_idx, const float* __restrict__ cand_val,
                                                                  const int32_t* __restrict__ cand_nnz, int n_tiles, int Kp, int n_users,
                                                                  const int32_t* __restrict__ user_list, const int32_t* __restrict__ n_list,
                                                                  int32_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                                  int32_t* __restrict__ out_nnz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = n_tiles, cap = T * Kp;
  int32_t* lcol = reinterpret_cast<int32_t*>(smem);
  float* lval = reinterpret_cast<float*>(smem + (size_t)cap * 4);
  rp_select_smem sel = rp_select_carve(smem + (size_t)cap * 8, cap);
  int* s_off = &sel.misc[RPS_USER];
  const int tid = threadIdx.x;
  const int n_u = user_list ? *n_list : n_users;
  for (int uk = blockIdx.x; uk < n_u; uk += gridDim.x) {
    const int ul = user_list ? user_list[uk] : uk;
    __syncthreads();
    if (tid == 0) {
      int o = 0;
      for (int t = 0; t < T; ++t) {
        s_off[t] = o;
        o += cand_nnz[(int64_t)ul * T + t];
      }
      s_off[T] = o;
    }
    __syncthreads();
    const int n = s_off[T];
    for (int t = 0; t < T; ++t) {
      const int cn = s_off[t + 1] - s_off[t];
      const int64_t so = ((int64_t)ul * T + t) * Kp;
      for (int k = tid; k < cn; k += RP_PRED_THREADS) {
        lcol[s_off[t] + k] = cand_idx[so + k];
        lval[s_off[t] + k] = cand_val[so + k];
      }
    }
    __syncthreads();
    rp_flist_src src{lcol, lval, n};
    const int kp = rp_select_topk<RP_PRED_THREADS>(src, n, Kp, sel);
    rp_select_emit<RP_PRED_THREADS>(n, sel, [&](int pos, int w) {
      out_idx[(int64_t)ul * Kp + pos] = lcol[w];
      out_val[(int64_t)ul * Kp + pos] = lval[w];
    });
    if (tid == 0) out_nnz[ul] = kp;
  }
}


// ---- K8, wave-per-user form ------------------------------------------------------------------------
// One WAVE owns one user: no workgroup barriers, PW_WAVES independent users per workgroup and ~12 per CU.
//   * lane l holds a cursor into the (column-sorted) neighbour list of the user's l-th history item
//     (PW_NB register sets => up to 64*PW_NB history items; heavier users take the workgroup kernel);
//   * the item axis is walked in tiles of PW_WT columns; each lane adds its row's entries that fall into the
//     tile to the wave's private f32 LDS tile (LDS float atomics; entries are prefetched two ahead);
//   * the tile is then scanned once (16 B per lane) and cleared; a lane first compares the maximum of its four
//     scores with the running threshold, so after warm-up almost every step is a read, a clear and one compare;
//     scores above the threshold enter the wave's pool in column order (rp_wave.h), which cuts itself to the
//     Kp best with an exact radix select when it is full;
//   * at the end the pool is cut to Kp, ranked by column and written out.
constexpr int PW_WT = 2048;
constexpr int PW_NB = 4;        // 64*PW_NB history items per user at most (cursors live in registers)
constexpr int PW_WAVES = 4;
constexpr int PW_UCH = 4;       // users per ticket
constexpr int PW_KP_MAX = 384;  // pool capacity = Kp rounded up + 128 <= 512

__host__ __device__ constexpr int rp_pw_cap(int Kp) { return ((Kp + 63) / 64) * 64 + 128; }
__host__ __device__ constexpr size_t rp_pw_wave_bytes(int Kp) {
  return (size_t)PW_WT * 4 + (size_t)rp_pw_cap(Kp) * 8 + 1024 + 128;
}

struct rp_pw_params {
  const int32_t* x_ptr;
  const int32_t* x_idx;
  const int2* pairs;
  const int32_t* s_nnz;
  int n_items, K, Kp, exclude_history, user_begin, n_users;
  int32_t* ticket;
  int32_t* out_idx;
  float* out_val;
  int32_t* out_nnz;
  unsigned long long* stamps;  // -DRP_STAMPS only
};

#ifdef RP_STAMPS
#define PW_STAMP(ph)                                                   \
  do {                                                                 \
    if (lane == 0) {                                                   \
      unsigned long long _t = __builtin_amdgcn_s_memtime();            \
      atomicAdd(&P.stamps[ph], _t - t_last);                           \
      t_last = _t;                                                     \
    }                                                                  \
  } while (0)
#else
#define PW_STAMP(ph) do { } while (0)
#endif

__global__ void __launch_bounds__(PW_WAVES * 64) k_predict_wave(rp_pw_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = P.K, Kp = P.Kp;
  const int cap = rp_pw_cap(Kp);
  char* base = smem + (size_t)wave * rp_pw_wave_bytes(Kp);
  float* acc = reinterpret_cast<float*>(base);
  float4* acc4 = reinterpret_cast<float4*>(acc);
  rp_wpool<uint32_t> pool;
  pool.key = reinterpret_cast<uint32_t*>(base + (size_t)PW_WT * 4);
  pool.pay = pool.key + cap;
  pool.hist = pool.pay + cap;
  pool.gidx = reinterpret_cast<uint16_t*>(pool.hist + 256);
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const int2 SENT = make_int2(0x7fffffff, 0);

  for (int i = lane; i < PW_WT / 4; i += 64) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef RP_STAMPS
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif

  while (true) {
    int ubase = 0;
    if (lane == 0) ubase = atomicAdd(P.ticket, PW_UCH);
    ubase = __shfl(ubase, 0);
    if (ubase >= P.n_users) break;
    const int uend = min(ubase + PW_UCH, P.n_users);
    for (int ul = ubase; ul < uend; ++ul) {
      const int u = P.user_begin + ul;
      const int hs = P.x_ptr[u], he = P.x_ptr[u + 1];
      const int d = he - hs;
      if (d <= 0 || d > 64 * PW_NB) continue;  // heavy users: workgroup kernel
      // per-lane cursors into the neighbour lists of history items lane, lane+64, ... ; entries prefetched two ahead
      int cur[PW_NB], end[PW_NB];
      int2 e0[PW_NB], e1[PW_NB], e2[PW_NB];
#pragma unroll
      for (int b = 0; b < PW_NB; ++b) {
        const int h = hs + b * 64 + lane;
        cur[b] = 0;
        end[b] = 0;
        if (h < he) {
          const int it = P.x_idx[h];
          cur[b] = it * K;
          end[b] = cur[b] + P.s_nnz[it];
        }
        e0[b] = (cur[b] < end[b]) ? P.pairs[cur[b]] : SENT;
        e1[b] = (cur[b] + 1 < end[b]) ? P.pairs[cur[b] + 1] : SENT;
        e2[b] = (cur[b] + 2 < end[b]) ? P.pairs[cur[b] + 2] : SENT;
      }
      pool.n = 0;
      pool.tau = 0;  // scores are positive: bits of a positive float order like the float
      PW_STAMP(0);

      for (int t0 = 0; t0 < P.n_items; t0 += PW_WT) {
        const int t_end = min(t0 + PW_WT, P.n_items);
        bool touched = false;
#pragma unroll
        for (int b = 0; b < PW_NB; ++b) {
          if (b * 64 >= d) break;
          while (__any(e0[b].x < t_end)) {
            if (e0[b].x < t_end) {
              atomicAdd(&acc[e0[b].x - t0], __int_as_float(e0[b].y));
              e0[b] = e1[b];
              e1[b] = e2[b];
              cur[b]++;
              e2[b] = (cur[b] + 2 < end[b]) ? P.pairs[cur[b] + 2] : SENT;
            }
            touched = true;
          }
        }
        PW_STAMP(1);
        if (!touched) continue;
        if (P.exclude_history) {
          for (int h = hs + lane; h < he; h += 64) {
            const int it = P.x_idx[h];
            if (it >= t0 && it < t_end) acc[it - t0] = 0.0f;
          }
        }
        __builtin_amdgcn_wave_barrier();
        for (int it = 0; it < PW_WT / 256; ++it) {
          const int q = it * 64 + lane;
          typedef float f32x4_t __attribute__((ext_vector_type(4)));
          volatile f32x4_t* accv = reinterpret_cast<volatile f32x4_t*>(acc);
          const f32x4_t v = accv[q];
          const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
          if (mx != 0.0f) accv[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          const float tau_f = __uint_as_float(pool.tau);
          if (!__any(mx > tau_f)) continue;
          // column order inside the step is lane-major, component-minor: push the four components of all lanes
          // as ONE ordered batch (positions from four ballots) so that the pool's strict threshold stays exact
          const float vv[4] = {v.x, v.y, v.z, v.w};
          bool keep[4];
          uint64_t m[4];
          int add = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            keep[c] = vv[c] > tau_f;
            m[c] = __ballot(keep[c]);
            add += __popcll(m[c]);
          }
          if (pool.n + add > cap && pool.n > Kp) {
            rp_wpool_cut<uint32_t, 0>(pool, Kp, lane);
            const float t2 = __uint_as_float(pool.tau);
            add = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              keep[c] = vv[c] > t2;
              m[c] = __ballot(keep[c]);
              add += __popcll(m[c]);
            }
          }
          if (pool.n + add > cap) {
            // more admissible scores in one step than the pool has room for: push the four components as separate
            // batches.  Their order is not the column order, so admission is non-strict here and ties are
            // resolved exactly by the next cut.
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const bool k2 = vv[c] >= __uint_as_float(pool.tau) && vv[c] > 0.0f;
              rp_wpool_push<uint32_t, 0>(pool, Kp, cap, k2, __float_as_uint(vv[c]), (uint32_t)(t0 + q * 4 + c), lane);
            }
            continue;
          }
          int below = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) below += __popcll(m[c] & lt_mask);
          int own = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (keep[c]) {
              const int pos = pool.n + below + own;
              ((volatile uint32_t*)pool.key)[pos] = __float_as_uint(vv[c]);
              ((volatile uint32_t*)pool.pay)[pos] = (uint32_t)(t0 + q * 4 + c);
            }
            own += keep[c] ? 1 : 0;
          }
          pool.n += add;
        }
        PW_STAMP(2);
      }
      // ---- final: best Kp by (score desc, column asc), then column order for the output row
      __builtin_amdgcn_wave_barrier();
      if (pool.n > Kp) rp_wpool_cut<uint32_t, 0>(pool, Kp, lane);
      const int n_out = pool.n;
      volatile uint32_t* pk = pool.key;
      volatile uint32_t* pp = pool.pay;
      for (int e = lane; e < n_out; e += 64) {
        const uint32_t col = pp[e];
        int rank = 0;
        for (int o = 0; o < n_out; ++o) rank += (pp[o] < col) ? 1 : 0;
        P.out_idx[(int64_t)ul * Kp + rank] = (int)col;
        P.out_val[(int64_t)ul * Kp + rank] = __uint_as_float(pk[e]);
      }
      if (lane == 0) P.out_nnz[ul] = n_out;
      __builtin_amdgcn_wave_barrier();
      PW_STAMP(3);
    }
  }
}

// users whose history exceeds what k_predict_wave holds in registers
__global__ void k_pred_split(const int32_t* __restrict__ x_ptr, int user_begin, int n_users, int thr, int32_t* __restrict__ list,
                             int32_t* __restrict__ cnt) {
  for (int ul = blockIdx.x * blockDim.x + threadIdx.x; ul < n_users; ul += gridDim.x * blockDim.x) {
    const int d = x_ptr[user_begin + ul + 1] - x_ptr[user_begin + ul];
    if (d > thr) list[atomicAdd(cnt, 1)] = ul;
  }
}

// model finalisation: f64 values -> f32, {column, value} pairs, column-order check
__global__ void k_model_finalize(const int32_t* __restrict__ idx, const double* __restrict__ val, const int32_t* __restrict__ nnz,
                                 int K, int64_t rows, float* __restrict__ val32, int2* __restrict__ pairs, int32_t* __restrict__ unsorted) {
  const int64_t n = rows * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = (float)val[i];
    val32[i] = v;
    pairs[i] = make_int2(idx[i], __float_as_int(v));
    const int k = (int)(i % K);
    if (k > 0 && k < nnz[i / K] && idx[i] <= idx[i - 1]) *unsorted = 1;
  }
}

int rp_model_finalize_topk(rphip_ctx* ctx, rphip_model* m) {
  const int64_t rows = m->row_end - m->row_begin;
  const int64_t n = rows * (int64_t)m->K;
  m->cols_sorted = true;
  if (n <= 0) return RPHIP_OK;
  int32_t* flag = nullptr;
  RP_TRY(rp_scratch(ctx, 5, 64, (void**)&flag));
  RP_HIP(hipMemsetAsync(flag, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_model_finalize, dim3((unsigned)std::min<int64_t>(4096, rp_cdiv(n, 256))), dim3(256), 0, ctx->stream, m->idx, m->val,
                     m->nnz, m->K, rows, m->val32, m->pairs, flag);
  RP_HIP(hipGetLastError());
  int32_t h = 0;
  RP_HIP(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  m->cols_sorted = (h == 0);
  return RPHIP_OK;
}

// ---- host ------------------------------------------------------------------------------------------
static int rp_pred_check(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* X, int64_t ub, int64_t ue) {
  if (!ctx || !model || !X) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (model->kind != RPHIP_MODEL_TOPK && model->kind != RPHIP_MODEL_DENSE) RP_FAIL(RPHIP_ESTATE, "unknown model kind");
  if (model->row_begin != 0 || model->row_end != model->n_items)
    RP_FAIL(RPHIP_ESTATE, "the model holds rows [%lld,%lld) of %lld: assemble all rows before predict", (long long)model->row_begin,
            (long long)model->row_end, (long long)model->n_items);
  if (X->n_items != model->n_items)
    RP_FAIL(RPHIP_EINVAL, "prediction matrix has %lld items, the model %lld", (long long)X->n_items, (long long)model->n_items);
  if (ub < 0 || ue > X->n_users || ub > ue) RP_FAIL(RPHIP_EINVAL, "user range [%lld,%lld) outside 0..%lld", (long long)ub, (long long)ue, (long long)X->n_users);
  return RPHIP_OK;
}

static void rp_pred_fill(rp_pred_params& P, const rphip_model* m, const rphip_csr* X, int64_t ub, int64_t ue) {
  P.x_ptr = X->x_ptr;
  P.x_idx = X->x_idx;
  P.s_idx = m->idx;
  P.s_val = m->val32;
  P.s_nnz = m->nnz;
  P.dense_B = (m->kind == RPHIP_MODEL_DENSE) ? m->B : nullptr;
  P.dense_ld = m->ld;
  P.n_items = (int)X->n_items;
  P.n_tiles = X->n_tiles;
  P.tile_w = X->tile_w;
  P.K = m->K;
  P.Kp = 0;
  P.exclude_history = 0;
  P.user_begin = (int)ub;
  P.n_users = (int)(ue - ub);
  P.user_list = nullptr;
  P.n_list = nullptr;
  P.mode = 0;
  P.out_idx = nullptr;
  P.out_val = nullptr;
  P.out_nnz = nullptr;
  P.row_nnz = nullptr;
  P.indptr = nullptr;
  P.full_idx = nullptr;
  P.full_val = nullptr;
}

static size_t rp_pred_lds(int tile_w) { return (size_t)tile_w * 4 + rp_select_smem_bytes(tile_w); }

// algorithmic bytes of K8: 8*K*nnz_pred + 4*(nnz_pred+U) + 8*U*Kp
static double rp_pred_work(const rphip_csr* X, const rphip_model* m, int64_t n_users, int Kp) {
  const double share = X->n_users > 0 ? (double)n_users / (double)X->n_users : 0.0;
  if (m->kind == RPHIP_MODEL_DENSE)  // K9: 4*I*nnz_pred (rows of B streamed) + 8*U*Kp
    return 4.0 * (double)m->n_items * (double)X->nnz * share + 8.0 * (double)n_users * Kp;
  return 8.0 * m->K * (double)X->nnz * share + 4.0 * ((double)X->nnz * share + (double)n_users) + 8.0 * (double)n_users * Kp;
}

extern "C" int rphip_predict_topk_dev(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                      int64_t user_end, int32_t Kp, int32_t exclude_history, int32_t* idx_dev, float* val_dev,
                                      int32_t* nnz_dev) {
  RP_TRY(rp_pred_check(ctx, model, Xpred, user_begin, user_end));
  if (!idx_dev || !val_dev || !nnz_dev) RP_FAIL(RPHIP_EINVAL, "NULL output");
  if (Kp < 1 || Kp > RP_K_MAX) RP_FAIL(RPHIP_ELIMIT, "Kp=%d outside 1..%d", Kp, RP_K_MAX);
  RP_TRY(rp_set_device(ctx));
  const int64_t n = user_end - user_begin;
  if (n == 0) return RPHIP_OK;
  hipStream_t st = ctx->stream;
  const int T = Xpred->n_tiles;
  hipMemsetAsync(idx_dev, 0xFF, (size_t)n * Kp * 4, st);
  hipMemsetAsync(val_dev, 0, (size_t)n * Kp * 4, st);
  hipMemsetAsync(nnz_dev, 0, (size_t)n * 4, st);
  rp_pred_params P;
  rp_pred_fill(P, model, Xpred, user_begin, user_end);
  P.Kp = Kp;
  P.exclude_history = exclude_history;
  P.mode = 0;
  // light users (history <= 64*PW_NB items) go to the wave-per-user kernel, the rest to the workgroup kernel
  // (measured: with one column tile the workgroup kernel is still faster - 8.0 vs 11.8 ms at 100K x 20K -, with
  // several tiles the wave kernel wins - 187 vs 347 ms at 1M x 100K)
  const bool use_wave = model->kind == RPHIP_MODEL_TOPK && model->cols_sorted && Kp <= PW_KP_MAX && (int64_t)model->n_items * model->K < (1ll << 31);
  int32_t* heavy_list = nullptr;
  int32_t* heavy_cnt = nullptr;
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  if (use_wave) {
    char* hb = nullptr;
    RP_TRY(rp_scratch(ctx, 4, (size_t)n * 4 + 64, (void**)&hb));
    heavy_cnt = (int32_t*)hb;          // [0] = count, [1] = ticket
    heavy_list = (int32_t*)(hb + 64);
    hipMemsetAsync(hb, 0, 64, st);
    hipLaunchKernelGGL(k_pred_split, dim3((unsigned)std::min<int64_t>(1024, rp_cdiv(n, 256))), dim3(256), 0, st, Xpred->x_ptr,
                       (int)user_begin, (int)n, 64 * PW_NB, heavy_list, heavy_cnt);
    rp_pw_params W;
    W.x_ptr = Xpred->x_ptr;
    W.x_idx = Xpred->x_idx;
    W.pairs = model->pairs;
    W.s_nnz = model->nnz;
    W.n_items = (int)Xpred->n_items;
    W.K = model->K;
    W.Kp = Kp;
    W.exclude_history = exclude_history;
    W.user_begin = (int)user_begin;
    W.n_users = (int)n;
    W.ticket = heavy_cnt + 1;
    W.out_idx = idx_dev;
    W.out_val = val_dev;
    W.out_nnz = nnz_dev;
    W.stamps = nullptr;
#ifdef RP_STAMPS
    {
      void* sp = nullptr;
      RP_TRY(rp_scratch(ctx, 6, 64 * 8, &sp));
      W.stamps = (unsigned long long*)sp;
      hipMemsetAsync(sp, 0, 64 * 8, st);
    }
#endif
    const size_t wl = rp_pw_wave_bytes(Kp) * PW_WAVES;
    RP_HIP(hipFuncSetAttribute((const void*)k_predict_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl));
    const int wpc = std::max<int>(1, (int)((160 * 1024) / wl));
    const int wg = (int)std::min<int64_t>(rp_cdiv(n, PW_UCH * PW_WAVES), (int64_t)ctx->n_cu * wpc);
    hipLaunchKernelGGL(k_predict_wave, dim3(wg), dim3(PW_WAVES * 64), wl, st, W);
#ifdef RP_STAMPS
    {
      unsigned long long h[8];
      hipMemcpyAsync(h, W.stamps, sizeof(h), hipMemcpyDeviceToHost, st);
      hipStreamSynchronize(st);
      fprintf(stderr, "[stamps predict_wave] cycles/1e6: setup %.1f accumulate %.1f scan+pool %.1f final %.1f\n", h[0] / 1e6, h[1] / 1e6,
              h[2] / 1e6, h[3] / 1e6);
    }
#endif
    P.user_list = heavy_list;
    P.n_list = heavy_cnt;
  }
  int32_t* cand_idx = nullptr;
  float* cand_val = nullptr;
  int32_t* cand_nnz = nullptr;
  if (T == 1) {
    P.out_idx = idx_dev;
    P.out_val = val_dev;
    P.out_nnz = nnz_dev;
  } else {
    if (T > 15) RP_FAIL(RPHIP_ELIMIT, "%d column tiles exceed the supported 15", T);
    char* cb = nullptr;
    const size_t nc = (size_t)n * T * Kp;
    RP_TRY(rp_scratch(ctx, 2, nc * 8 + (size_t)n * T * 4 + 64, (void**)&cb));
    cand_idx = (int32_t*)cb;
    cand_val = (float*)(cb + nc * 4);
    cand_nnz = (int32_t*)(cb + nc * 8);
    hipMemsetAsync(cand_nnz, 0, (size_t)n * T * 4, st);
    P.out_idx = cand_idx;
    P.out_val = cand_val;
    P.out_nnz = cand_nnz;
  }
  const size_t lds = rp_pred_lds(Xpred->tile_w);
  RP_HIP(hipFuncSetAttribute((const void*)k_predict_sparse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (int)std::min<int64_t>(n * T, (int64_t)ctx->n_cu * (use_wave ? 1 : 8));
  hipLaunchKernelGGL(k_predict_sparse, dim3(grid), dim3(RP_PRED_THREADS), lds, st, P);
  if (T > 1) {
    const size_t mlds = (size_t)T * Kp * 8 + rp_select_smem_bytes(T * Kp);
    if (mlds > 160 * 1024 || T * Kp >= RPS_FLAG)
      RP_FAIL(RPHIP_ELIMIT, "Kp=%d with %d column tiles needs %zu bytes of LDS for the merge", Kp, T, mlds);
    RP_HIP(hipFuncSetAttribute((const void*)k_predict_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
    const int mg = (int)std::min<int64_t>(n, (int64_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(k_predict_merge, dim3(mg), dim3(RP_PRED_THREADS), mlds, st, cand_idx, cand_val, cand_nnz, T, Kp, (int)n,
                       use_wave ? heavy_list : nullptr, heavy_cnt, idx_dev, val_dev, nnz_dev);
  }
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, RPHIP_T_PREDICT, rp_pred_work(Xpred, model, n, Kp));
  return RPHIP_OK;
}

extern "C" int rphip_predict_topk(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                  int64_t user_end, int32_t Kp, int32_t exclude_history, int32_t* idx_host, float* val_host,
                                  int32_t* nnz_host) {
  RP_TRY(rp_pred_check(ctx, model, Xpred, user_begin, user_end));
  if (!idx_host || !val_host || !nnz_host) RP_FAIL(RPHIP_EINVAL, "NULL output");
  if (Kp < 1 || Kp > RP_K_MAX) RP_FAIL(RPHIP_ELIMIT, "Kp=%d outside 1..%d", Kp, RP_K_MAX);
  RP_TRY(rp_set_device(ctx));
  const int64_t n = user_end - user_begin;
  if (n == 0) return RPHIP_OK;
  char* ob = nullptr;
  RP_TRY(rp_scratch(ctx, 3, (size_t)n * Kp * 8 + (size_t)n * 4 + 64, (void**)&ob));
  int32_t* di = (int32_t*)ob;
  float* dv = (float*)(ob + (size_t)n * Kp * 4);
  int32_t* dn = (int32_t*)(ob + (size_t)n * Kp * 8);
  RP_TRY(rphip_predict_topk_dev(ctx, model, Xpred, user_begin, user_end, Kp, exclude_history, di, dv, dn));
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  RP_HIP(hipMemcpyAsync(idx_host, di, (size_t)n * Kp * 4, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipMemcpyAsync(val_host, dv, (size_t)n * Kp * 4, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipMemcpyAsync(nnz_host, dn, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  rp_time_end(ctx, ev, RPHIP_T_D2H, (double)n * Kp * 8.0);
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

extern "C" int rphip_predict_full_count(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                        int64_t user_end, int64_t* row_nnz_host, int64_t* total_out) {
  RP_TRY(rp_pred_check(ctx, model, Xpred, user_begin, user_end));
  if (!row_nnz_host || !total_out) RP_FAIL(RPHIP_EINVAL, "NULL output");
  RP_TRY(rp_set_device(ctx));
  const int64_t n = user_end - user_begin;
  *total_out = 0;
  if (n == 0) return RPHIP_OK;
  long long* dn = nullptr;
  RP_TRY(rp_scratch(ctx, 3, (size_t)n * 8, (void**)&dn));
  rp_pred_params P;
  rp_pred_fill(P, model, Xpred, user_begin, user_end);
  P.mode = 1;
  P.row_nnz = dn;
  const size_t lds = rp_pred_lds(Xpred->tile_w);
  RP_HIP(hipFuncSetAttribute((const void*)k_predict_sparse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (int)std::min<int64_t>(n, (int64_t)ctx->n_cu * 8);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  hipLaunchKernelGGL(k_predict_sparse, dim3(grid), dim3(RP_PRED_THREADS), lds, ctx->stream, P);
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, RPHIP_T_PREDICT, rp_pred_work(Xpred, model, n, 0));
  RP_HIP(hipMemcpyAsync(row_nnz_host, dn, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  int64_t tot = 0;
  for (int64_t i = 0; i < n; ++i) tot += row_nnz_host[i];
  *total_out = tot;
  return RPHIP_OK;
}

extern "C" int rphip_predict_full_fill(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                       int64_t user_end, const int64_t* indptr_host, int32_t* indices_host, double* data_host) {
  RP_TRY(rp_pred_check(ctx, model, Xpred, user_begin, user_end));
  if (!indptr_host) RP_FAIL(RPHIP_EINVAL, "indptr is NULL");
  RP_TRY(rp_set_device(ctx));
  const int64_t n = user_end - user_begin;
  if (n == 0) return RPHIP_OK;
  const int64_t total = indptr_host[n];
  if (total < 0) RP_FAIL(RPHIP_EINVAL, "indptr[n] is negative");
  if (total == 0) return RPHIP_OK;
  if (!indices_host || !data_host) RP_FAIL(RPHIP_EINVAL, "NULL output");
  char* ob = nullptr;
  RP_TRY(rp_scratch(ctx, 3, (size_t)total * 12 + (size_t)(n + 1) * 8 + 64, (void**)&ob));
  double* dv = (double*)ob;
  long long* dp = (long long*)(ob + (size_t)total * 8);
  int32_t* di = (int32_t*)(ob + (size_t)total * 8 + (size_t)(n + 1) * 8);
  RP_HIP(hipMemcpyAsync(dp, indptr_host, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  rp_pred_params P;
  rp_pred_fill(P, model, Xpred, user_begin, user_end);
  P.mode = 2;
  P.indptr = dp;
  P.full_idx = di;
  P.full_val = dv;
  const size_t lds = rp_pred_lds(Xpred->tile_w);
  RP_HIP(hipFuncSetAttribute((const void*)k_predict_sparse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (int)std::min<int64_t>(n, (int64_t)ctx->n_cu * 8);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  hipLaunchKernelGGL(k_predict_sparse, dim3(grid), dim3(RP_PRED_THREADS), lds, ctx->stream, P);
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, RPHIP_T_PREDICT, rp_pred_work(Xpred, model, n, 0) + 12.0 * (double)total);
  RP_HIP(hipMemcpyAsync(indices_host, di, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipMemcpyAsync(data_host, dv, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

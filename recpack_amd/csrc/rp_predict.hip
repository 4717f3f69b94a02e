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
// One WAVE owns one user: no workgroup barriers, one user per 64-thread workgroup and ~7 in flight per CU.
//   * the item axis is walked in tiles of PW_WT columns; toff[i][t] (built once per model) is the offset of the
//     first entry of neighbour list i with column >= t*PW_WT, so for a tile lane l knows the slice of the l-th
//     history item's (column-sorted) list that falls into it;
//   * the slices of the 64 history items of a batch are cut into chunks of four entries and chunk c goes to lane
//     c % 64 (two 16-byte loads per chunk): the loads are coalesced per list and every lane is busy whatever the
//     history length; histories longer than 64 take several batches per tile;
//   * scores are accumulated in 64-bit FIXED POINT (f32 similarity * 2^50, exact for every similarity >= 2^-27) with
//     u64 LDS atomics: measured on MI355X an LDS f32 atomic add sustains 0.33 per clock per CU, u32/u64 integer adds
//     4.4-4.8 (tools/lds_atomic_bench.hip).  The integer sum is exact and independent of the order of the adds, so
//     predict is bit-reproducible;
//   * a 2048-bit dirty map (one word per lane, LDS atomic OR) records the touched columns; after the tile's adds
//     each lane walks the set bits of its word, reads and clears the sums and offers them to the wave's running
//     top-Kp pool (rp_wave.h: exact radix-select cuts; admission is exact for any arrival order);
//   * at the end the pool is cut to Kp, ranked by column and written out (score = (float)(sum * 2^-50)).
constexpr int PW_WT = 2048;
constexpr int PW_CH = 4;        // entries per chunk (32 bytes)
constexpr int PW_UCH = 4;       // users per ticket
constexpr int PW_KP_MAX = 384;  // pool capacity = Kp rounded up + 128 <= 512
constexpr int PW_FIX_SHIFT = 50;

__host__ __device__ constexpr int rp_pw_cap(int Kp) { return ((Kp + 63) / 64) * 64 + 128; }
__host__ __device__ constexpr size_t rp_pw_wave_bytes(int Kp) { return (size_t)PW_WT * 8 + 256 + (size_t)rp_pw_cap(Kp) * 12 + 1024 + 128; }

// f32 bits (positive, normal) -> value * 2^50 as u64, by integer arithmetic (exact when the value is >= 2^-27)
__device__ __forceinline__ unsigned long long rp_f32_to_fix(int bits) {
  const unsigned long long m = (unsigned long long)((bits & 0x7fffff) | 0x800000);
  const int sh = ((bits >> 23) & 0xff) - 127 - 23 + PW_FIX_SHIFT;
  return sh >= 0 ? (m << sh) : (m >> (-sh));
}

struct rp_pw_params {
  const int32_t* x_ptr;
  const int32_t* x_idx;
  const int2* pairs;
  const uint16_t* toff;  // [I * (n_wt + 1)]
  int n_items, n_wt, K, Kp, exclude_history, user_begin, n_users;
  int32_t* ticket;
  int32_t* out_idx;
  float* out_val;
  int32_t* out_nnz;
  unsigned long long* stamps;  // -DRP_STAMPS only
};

#ifdef RP_STAMPS
#define PW_STAMP(ph) do { unsigned long long _t = __builtin_amdgcn_s_memtime(); t_ph[ph] += _t - t_last; t_last = _t; } while (0)
#define PW_FLUSH() do { if (lane == 0) { for (int _p = 0; _p < 4; ++_p) atomicAdd(&P.stamps[_p], t_ph[_p]); } t_ph[0] = t_ph[1] = t_ph[2] = t_ph[3] = 0; } while (0)
#else
#define PW_STAMP(ph) do { } while (0)
#define PW_FLUSH() do { } while (0)
#endif

__global__ void __launch_bounds__(64) k_predict_wave(rp_pw_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int K = P.K, Kp = P.Kp, NT = P.n_wt;
  const int cap = rp_pw_cap(Kp);
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);
  volatile uint32_t* dirty = reinterpret_cast<volatile uint32_t*>(smem + (size_t)PW_WT * 8);
  rp_wpool<uint64_t> pool;
  pool.key = reinterpret_cast<uint64_t*>(smem + (size_t)PW_WT * 8 + 256);
  pool.pay = reinterpret_cast<uint32_t*>(pool.key + cap);
  pool.hist = pool.pay + cap;
  pool.gidx = reinterpret_cast<uint16_t*>(pool.hist + 256);
  typedef int v4i_a4 __attribute__((ext_vector_type(4), aligned(4)));
  const int* pairs_i = reinterpret_cast<const int*>(P.pairs);

  for (int i = lane; i < PW_WT; i += 64) acc[i] = 0ull;
  dirty[lane] = 0u;
#ifdef RP_STAMPS
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
  unsigned long long t_ph[4] = {0, 0, 0, 0};
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
      if (he <= hs) continue;
      pool.n = 0;
      pool.tau = 0;
      pool.tau_col = 0;
      PW_STAMP(0);

      // The first batch of 64 history items is software-pipelined over the tiles: its tile offsets are loaded two
      // tiles ahead and the entries of tile t+1 are in flight while tile t is added, scanned and pooled.  Without
      // this the three dependent loads (item id -> offsets -> entries) are paid once per tile.
      const int it0 = (hs + lane < he) ? P.x_idx[hs + lane] : -1;
      const uint16_t* to0 = P.toff + (long long)max(it0, 0) * (NT + 1);
      int o_a = (it0 >= 0) ? (int)to0[0] : 0;              // offsets of tiles t+1, t+2, t+3 relative to the loop below
      int o_b = (it0 >= 0) ? (int)to0[1] : 0;
      int o_c = (it0 >= 0 && NT >= 2) ? (int)to0[2] : o_b;
      // state of the prepared (next) tile's first chunk step
      int p_s = 0, p_len = 0, p_choff = 0, p_nchunks = 0, p_n = 0;
      v4i_a4 p_a = (v4i_a4){0, 0, 0, 0}, p_b = p_a;
      auto prepare = [&](int oa, int ob) {
        p_s = (it0 >= 0) ? it0 * K + oa : 0;
        p_len = (it0 >= 0) ? ob - oa : 0;
        const int nch = (p_len + PW_CH - 1) / PW_CH;
        const int inc = rp_wave_incl_scan(nch, lane);
        p_choff = inc - nch;
        p_nchunks = __shfl(inc, 63);
        int lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
          const int v = __shfl(p_choff, lo + step);
          if (v <= lane) lo += step;
        }
        const int us = __shfl(p_s, lo), ulen = __shfl(p_len, lo), co = __shfl(p_choff, lo);
        const int k0 = (lane - co) * PW_CH;
        p_n = (lane < p_nchunks) ? min(PW_CH, ulen - k0) : 0;
        if (p_n > 0) {
          const int* src = pairs_i + 2 * (long long)(us + k0);
          p_a = *reinterpret_cast<const v4i_a4*>(src);
          p_b = *reinterpret_cast<const v4i_a4*>(src + 4);
        }
      };
      auto add_chunk = [&](int n, const v4i_a4& a, const v4i_a4& b, int t0) {
        const int ec[4] = {a.x, a.z, b.x, b.z};
        const int ev[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
        for (int e = 0; e < PW_CH; ++e) {
          if (e < n) {
            const int j = ec[e] - t0;
            atomicAdd(&acc[j], rp_f32_to_fix(ev[e]));
            atomicOr((uint32_t*)&dirty[j >> 5], 1u << (j & 31));
          }
        }
      };
      prepare(o_a, o_b);  // tile 0

      for (int t = 0; t < NT; ++t) {
        const int t0 = t * PW_WT;
        // current tile = what was prepared; then prepare tile t+1 and fetch the offset of tile t+3
        const int c_s = p_s, c_len = p_len, c_choff = p_choff, c_nchunks = p_nchunks, c_n = p_n;
        const v4i_a4 c_a = p_a, c_b = p_b;
        o_a = o_b;
        o_b = o_c;
        if (t + 3 <= NT && it0 >= 0) o_c = (int)to0[t + 3];
        if (t + 1 < NT) prepare(o_a, o_b);
        bool touched = c_nchunks > 0;
        // ---- first batch: the prefetched first step, then any further steps
        add_chunk(c_n, c_a, c_b, t0);
        for (int c0 = 64; c0 < c_nchunks; c0 += 64) {
          const int c = c0 + lane;
          int lo = 0;
#pragma unroll
          for (int step = 32; step > 0; step >>= 1) {
            const int v = __shfl(c_choff, lo + step);
            if (v <= c) lo += step;
          }
          const int us = __shfl(c_s, lo), ulen = __shfl(c_len, lo), co = __shfl(c_choff, lo);
          const int k0 = (c - co) * PW_CH;
          const int n = (c < c_nchunks) ? min(PW_CH, ulen - k0) : 0;
          v4i_a4 a = (v4i_a4){0, 0, 0, 0}, b2 = a;
          if (n > 0) {
            const int* src = pairs_i + 2 * (long long)(us + k0);
            a = *reinterpret_cast<const v4i_a4*>(src);
            b2 = *reinterpret_cast<const v4i_a4*>(src + 4);
          }
          add_chunk(n, a, b2, t0);
        }
        // ---- histories longer than 64 items: the remaining batches, not pipelined
        for (int hb = hs + 64; hb < he; hb += 64) {
          const int h = hb + lane;
          int s = 0, len = 0;
          if (h < he) {
            const int it = P.x_idx[h];
            const uint16_t* to = P.toff + (long long)it * (NT + 1) + t;
            const int o0 = to[0], o1 = to[1];
            s = it * K + o0;
            len = o1 - o0;
          }
          const int nch = (len + PW_CH - 1) / PW_CH;
          const int inc = rp_wave_incl_scan(nch, lane);
          const int choff = inc - nch;
          const int n_chunks = __shfl(inc, 63);
          if (n_chunks > 0) touched = true;
          for (int c0 = 0; c0 < n_chunks; c0 += 64) {
            const int c = c0 + lane;
            int lo = 0;  // owner = largest lane l with choff[l] <= c
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
              const int v = __shfl(choff, lo + step);
              if (v <= c) lo += step;
            }
            const int us = __shfl(s, lo), ulen = __shfl(len, lo), co = __shfl(choff, lo);
            const int k0 = (c - co) * PW_CH;
            const int n = (c < n_chunks) ? min(PW_CH, ulen - k0) : 0;
            v4i_a4 a = (v4i_a4){0, 0, 0, 0}, b2 = a;
            if (n > 0) {
              const int* src = pairs_i + 2 * (long long)(us + k0);
              a = *reinterpret_cast<const v4i_a4*>(src);      // the pairs array is padded: reading past
              b2 = *reinterpret_cast<const v4i_a4*>(src + 4);  // the slice's end is harmless
            }
            add_chunk(n, a, b2, t0);
          }
        }
        PW_STAMP(1);
        if (!touched) continue;
        if (P.exclude_history) {
          for (int h = hs + lane; h < he; h += 64) {
            const int j = P.x_idx[h] - t0;
            if ((unsigned)j < (unsigned)PW_WT) acc[j] = 0ull;  // the dirty bit stays: a zero sum is skipped below
          }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- offer the touched columns to the pool: lane l walks the set bits of dirty word l
        uint32_t w = dirty[lane];
        dirty[lane] = 0u;
        while (__any(w != 0u)) {
          const bool has = w != 0u;
          const int bpos = has ? (__ffs((int)w) - 1) : 0;
          const int j = lane * 32 + bpos;
          unsigned long long v = 0ull;
          if (has) {
            v = ((volatile unsigned long long*)acc)[j];
            ((volatile unsigned long long*)acc)[j] = 0ull;
            w &= w - 1u;
          }
          const uint32_t col = (uint32_t)(t0 + j);
          const bool keep = has && v != 0ull && rp_wpool_admits<uint64_t, 0>(pool, v, col);
          rp_wpool_push<uint64_t, 0>(pool, Kp, cap, keep, v, col, lane);
        }
        PW_STAMP(2);
      }
      // ---- final: best Kp by (score desc, column asc), then column order for the output row
      __builtin_amdgcn_wave_barrier();
      if (pool.n > Kp) rp_wpool_cut<uint64_t, 0>(pool, Kp, lane);
      const int n_out = pool.n;
      volatile uint64_t* pk = pool.key;
      volatile uint32_t* pp = pool.pay;
      for (int e = lane; e < n_out; e += 64) {
        const uint32_t col = pp[e];
        int rank = 0;
        for (int o = 0; o < n_out; ++o) rank += (pp[o] < col) ? 1 : 0;
        P.out_idx[(int64_t)ul * Kp + rank] = (int)col;
        P.out_val[(int64_t)ul * Kp + rank] = (float)((double)pk[e] * 0x1p-50);
      }
      if (lane == 0) P.out_nnz[ul] = n_out;
      __builtin_amdgcn_wave_barrier();
      PW_STAMP(3);
      PW_FLUSH();
    }
  }
}

// toff[i][t] = number of entries of neighbour list i with column < t*PW_WT (lists are column-sorted)
__global__ void k_build_toff(const int32_t* __restrict__ idx, const int32_t* __restrict__ nnz, int K, int n_items, int n_wt,
                             uint16_t* __restrict__ toff) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) {
    uint16_t* o = toff + (long long)i * (n_wt + 1);
    const int32_t* row = idx + (long long)i * K;
    const int n = nnz[i];
    int t = 0;
    o[0] = 0;
    for (int k = 0; k < n; ++k) {
      const int tj = row[k] / PW_WT;
      while (t < tj) o[++t] = (uint16_t)k;
    }
    while (t < n_wt) o[++t] = (uint16_t)n;
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
  // neighbour-list models with column-sorted rows: wave-per-user kernel (any history length); otherwise (dense
  // models, unsorted lists, very large Kp) the workgroup kernel below
  const bool use_wave = model->kind == RPHIP_MODEL_TOPK && model->cols_sorted && Kp <= PW_KP_MAX && model->K < 65536 &&
                        (int64_t)model->n_items * model->K < (1ll << 30);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  if (use_wave) {
    rphip_model* mm = const_cast<rphip_model*>(model);  // lazily built, cached acceleration structure
    const int n_wt = (int)rp_cdiv(model->n_items, PW_WT);
    if (!mm->toff) {
      RP_HIP(hipMalloc((void**)&mm->toff, (size_t)model->n_items * (n_wt + 1) * sizeof(uint16_t)));
      hipLaunchKernelGGL(k_build_toff, dim3((unsigned)std::min<int64_t>(4096, rp_cdiv(model->n_items, 256))), dim3(256), 0, st, model->idx,
                         model->nnz, model->K, (int)model->n_items, n_wt, mm->toff);
    }
    int32_t* tk = nullptr;
    RP_TRY(rp_scratch(ctx, 4, 64, (void**)&tk));
    hipMemsetAsync(tk, 0, 64, st);
    rp_pw_params W;
    W.x_ptr = Xpred->x_ptr;
    W.x_idx = Xpred->x_idx;
    W.pairs = model->pairs;
    W.toff = mm->toff;
    W.n_items = (int)Xpred->n_items;
    W.n_wt = n_wt;
    W.K = model->K;
    W.Kp = Kp;
    W.exclude_history = exclude_history;
    W.user_begin = (int)user_begin;
    W.n_users = (int)n;
    W.ticket = tk;
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
    const size_t wl = rp_pw_wave_bytes(Kp);
    RP_HIP(hipFuncSetAttribute((const void*)k_predict_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl));
    const int wpc = std::max<int>(1, std::min<int>(16, (int)((160 * 1024) / wl)));
    const int wg = (int)std::min<int64_t>(rp_cdiv(n, PW_UCH), (int64_t)ctx->n_cu * wpc);
    hipLaunchKernelGGL(k_predict_wave, dim3(wg), dim3(64), wl, st, W);
    RP_HIP(hipGetLastError());
#ifdef RP_STAMPS
    {
      unsigned long long h[8];
      hipMemcpyAsync(h, W.stamps, sizeof(h), hipMemcpyDeviceToHost, st);
      hipStreamSynchronize(st);
      fprintf(stderr, "[stamps predict_wave] cycles/1e6: setup %.1f accumulate %.1f scan+pool %.1f final %.1f (grid %d)\n", h[0] / 1e6, h[1] / 1e6,
              h[2] / 1e6, h[3] / 1e6, wg);
    }
#endif
    rp_time_end(ctx, ev, RPHIP_T_PREDICT, rp_pred_work(Xpred, model, n, Kp));
    return RPHIP_OK;
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
  const int grid = (int)std::min<int64_t>(n * T, (int64_t)ctx->n_cu * 8);
  hipLaunchKernelGGL(k_predict_sparse, dim3(grid), dim3(RP_PRED_THREADS), lds, st, P);
  if (T > 1) {
    const size_t mlds = (size_t)T * Kp * 8 + rp_select_smem_bytes(T * Kp);
    if (mlds > 160 * 1024 || T * Kp >= RPS_FLAG)
      RP_FAIL(RPHIP_ELIMIT, "Kp=%d with %d column tiles needs %zu bytes of LDS for the merge", Kp, T, mlds);
    RP_HIP(hipFuncSetAttribute((const void*)k_predict_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
    const int mg = (int)std::min<int64_t>(n, (int64_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(k_predict_merge, dim3(mg), dim3(RP_PRED_THREADS), mlds, st, cand_idx, cand_val, cand_nnz, T, Kp, (int)n,
                       nullptr, nullptr, idx_dev, val_dev, nnz_dev);
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

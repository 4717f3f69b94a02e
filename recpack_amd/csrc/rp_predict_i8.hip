// K9, first pass of the two-pass predict on the int8 copy of B - ONE TASK PER USER (round 5).
//
// Replaces, with rp_predict.hip's rescoring, ItemSimilarityMatrixAlgorithm._predict for dense models (recpack/algorithms/base.py:248,
// `X @ similarity_matrix_`) followed by the per-user top-K of MetricTopK (recpack/metrics/base.py:189).
//
// Round 4's first pass made (user, column tile) the task of a workgroup: every task wrote its sums to an LDS tile, found a lower bound
// of the tile's Kc-th best score, handed every column above it to a per-(user, tile) list in HBM, and a merge kernel then cut the
// user's T lists to Kc candidates for the rescoring.  Kernel time at 1M x 100K: 0.89 s for a stream that alone runs in 0.66 s -
// 0.23 s of per-(user, tile) selection and start-up - plus 0.07 s of merges.
//
// Here a workgroup owns a USER and walks his column tiles one after the other:
//   * the sums of a tile stay in the registers they were accumulated in (16 columns per thread, no LDS tile: tiles are 16 384
//     columns wide whatever the LDS allows, 7 instead of 9 at 100 000 items);
//   * a running threshold tau - a lower bound of the user's Kc-th best approximate score - decides from the registers which columns
//     are candidates; they are appended to ONE list per user in LDS (a rare branch: ~2 % of the threads of a tile have one).  The
//     first tile finds tau as round 4 did (every thread's best score, of those the R-th best of every wave: R x waves >= Kc columns
//     reach it); whenever the list passes half its capacity it is cut to its Kc best entries and tau rises to the smallest of them;
//   * after the last tile the list is cut to Kc entries and written as the rescoring's input: no per-tile lists, no merge kernel,
//     no barrier per tile but one.
// Every column the list drops lies below Kc columns that stay, so the list ends as the user's Kc best columns by approximate
// score - what the merge produced - and the rescoring's proof (k_predict_rescore) is unchanged.  A user whose list overflows (a
// model of few distinct values: everything ties) goes to the fallback list that the one-pass kernel serves, like an unproven one.
#include <algorithm>

#include "rp_common.h"
#include "rp_predict_i8.h"
#include "rp_select.h"
#include "rp_wave.h"

constexpr int RPU_THREADS = 1024;
constexpr int RPU_CAP = 2048;    // entries of a user's candidate list in LDS
constexpr int RPU_PRUNE = 1024;  // the list is cut to Kc entries when it holds more than this after a tile
constexpr int RPU_KC_MAX = 256;
constexpr int RPU_HIST = 2048;   // history items kept in LDS for the membership test of history removal (longer histories: searched in memory)

struct rpu_list_src {
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

__device__ __forceinline__ int rpu_l2g(const rp_i8_params& P, int l) {
  return P.cyc_nb ? rp_cyc_panel(l / P.cyc_nb, P.cyc_r, P.cyc_w) * P.cyc_nb + l % P.cyc_nb : P.col_begin + l;
}

// is item g in the (ascending) history h[0, n) ?
template <class Ptr>
__device__ __forceinline__ bool rpu_in_history(Ptr h, int n, int g) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int m = (lo + hi) >> 1;
    const int v = h[m];
    if (v == g) return true;
    if (v < g) lo = m + 1; else hi = m;
  }
  return false;
}

__global__ void __launch_bounds__(RPU_THREADS, 8) k_predict_user_i8(rp_i8_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (explicit LDS pointers: generic ones kept in structs compile to FLAT accesses whose waits drain the global loads in flight)
  RP_LDS int32_t* lcol = (RP_LDS int32_t*)smem;                             // [CAP]
  RP_LDS float* lval = (RP_LDS float*)(smem + (size_t)RPU_CAP * 4);         // [CAP]
  RP_LDS int32_t* tcol = (RP_LDS int32_t*)(smem + (size_t)RPU_CAP * 8);     // [KC_MAX]
  RP_LDS float* tval = (RP_LDS float*)(smem + (size_t)RPU_CAP * 8 + (size_t)RPU_KC_MAX * 4);
  RP_LDS int32_t* s_hist = (RP_LDS int32_t*)(smem + (size_t)RPU_CAP * 8 + (size_t)RPU_KC_MAX * 8);  // [HIST]
  char* sel_base = smem + (size_t)RPU_CAP * 8 + (size_t)RPU_KC_MAX * 8 + (size_t)RPU_HIST * 4;
  rp_select_smem sel = rp_select_carve(sel_base, RPU_CAP);
  RP_LDS int* s_cnt = (RP_LDS int*)&sel.misc[RPS_USER];                    // [0] list entries (the append position), [1] history items inside the tile, [2] next user,
                                                                           // [4], [5] entries appended by the even / odd tiles (see the count below)
  RP_LDS float* s_tau = (RP_LDS float*)&sel.misc[RPS_USER + 8];            // [16]
  RP_LDS int* s_cw = (RP_LDS int*)&sel.misc[RPS_LCNT];                     // [16] (the selection's own between its calls: here only between barriers of the first tile)
  const int32_t* lcol_g = reinterpret_cast<const int32_t*>(smem);          // (the selection's source takes generic pointers)
  const float* lval_g = reinterpret_cast<const float*>(smem + (size_t)RPU_CAP * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = RPU_THREADS / 64;
  const int W = P.tile_w, T = P.n_tiles, Kc = P.Kc;

  typedef int rp_i4 __attribute__((ext_vector_type(4)));
  // One 32-bit word at a time - four sign-extending byte conversions and four fmas - and the words strictly in order: the empty asm
  // statements tie each word to the sums the previous one wrote (left to itself the compiler converts all 64 bytes of the four rows
  // first and spills: two workgroups per CU leave a thread 64 registers).
  auto add16 = [&](float(&acc)[16], const rp_i4& v, float sc) __attribute__((always_inline)) {
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
      unsigned w = (unsigned)v[wd];
      if (wd == 0)
        asm volatile("" : "+v"(w), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
      else
        asm volatile("" : "+v"(w), "+v"(acc[4 * wd - 4]), "+v"(acc[4 * wd - 3]), "+v"(acc[4 * wd - 2]), "+v"(acc[4 * wd - 1]));
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[4 * wd + b] = fmaf((float)(int)(int8_t)(w >> (8 * b)), sc, acc[4 * wd + b]);
    }
  };

  while (true) {
    // ---- next user (a ticket: histories differ by two orders of magnitude)
    __syncthreads();
    if (tid == 0) {
      s_cnt[2] = atomicAdd(P.ticket, 1);
      s_cnt[0] = 0;
      s_cnt[4] = s_cnt[5] = 0;
    }
    __syncthreads();
    const int slot = s_cnt[2];  // the user's place in this pass (the row of its outputs)
    if (slot >= P.n_users) break;
    const int ul = P.user_list ? P.user_list[slot] : slot;
    const int u = P.user_begin + ul;
    const int hs = P.x_ptr[u], he = P.x_ptr[u + 1];
    if (he <= hs) {  // no history: no score
      if (tid == 0) P.out_nnz[slot] = 0;
      continue;
    }
    if (P.max_hist > 0 && he - hs > P.max_hist) {  // (one workgroup walks all tiles of a user: among few users a long one is the whole launch)
      if (tid == 0) {
        P.out_nnz[slot] = -1;
        P.fail_list[atomicAdd(P.n_fail, 1)] = P.fail_base + ul;
      }
      continue;
    }
    float tau = -INFINITY;
    bool have_tau = false, failed = false;
    int n_other = 0;  // entries appended by the tiles of the other parity, as of the previous tile's barrier
    const bool hist_lds = P.exclude_history && he - hs <= RPU_HIST;
    if (hist_lds) {
      for (int k = tid; k < he - hs; k += RPU_THREADS) s_hist[k] = P.x_idx[hs + k];
      __syncthreads();  // (every tile looks its history columns up before anything else reads its sums)
    }

    for (int tile = 0; tile < T; ++tile) {
      const int t0 = tile * W;
      const int n_cols = min(W, P.n_items - t0);
      const int w16 = (n_cols + 15) >> 4;  // (reads past n_cols stay inside the padded row: its length is a multiple of 64)
      const bool active = tid < w16;
      // ---- accumulate: sixteen consecutive columns per 16-byte load, four rows in flight
      float r[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) r[e] = 0.0f;
      constexpr int RIF = 4;
      typedef const __attribute__((address_space(1))) int8_t* rp_gptr;  // (a GLOBAL pointer: rebuilt from an integer it would be a flat one)
      int h = hs;
      for (; h + RIF - 1 < he; h += RIF) {
        rp_gptr R[RIF];
        float sc[RIF];
#pragma unroll
        for (int q = 0; q < RIF; ++q) {
          const int it = __builtin_amdgcn_readfirstlane(P.x_idx[h + q]);  // (uniform: the row's base and its scale in scalar registers)
          const unsigned long long rb = reinterpret_cast<unsigned long long>(P.B8 + (long long)it * P.ld + t0);
          R[q] = (rp_gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rb >> 32)) << 32) |
                           (unsigned)__builtin_amdgcn_readfirstlane((int)rb));
          sc[q] = P.b8_scale[it];
        }
        rp_i4 v[RIF];
        if (active) {
#pragma unroll
          for (int q = 0; q < RIF; ++q)  // (scalar base + 32-bit offset: no address registers)
            v[q] = *reinterpret_cast<const __attribute__((address_space(1))) rp_i4*>(R[q] + (unsigned)(16 * tid));
        } else {
#pragma unroll
          for (int q = 0; q < RIF; ++q) v[q] = (rp_i4){0, 0, 0, 0};
        }
#pragma unroll
        for (int q = 0; q < RIF; ++q) add16(r, v[q], sc[q]);
      }
      for (; h < he; ++h) {
        const int it = __builtin_amdgcn_readfirstlane(P.x_idx[h]);
        const int8_t* R0 = P.B8 + (long long)it * P.ld + t0;
        const float sc0 = P.b8_scale[it];
        if (active) add16(r, *reinterpret_cast<const rp_i4*>(R0 + 16 * tid), sc0);
      }
      // (columns past the tile's end and zero sums are no candidates)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (!active || 16 * tid + e >= n_cols) r[e] = 0.0f;

      // ---- history removal (round 6): the history's columns lose their sums HERE, in the registers - a thread's sixteen columns are
      // consecutive item ids (a cyclic slab: inside one panel, whose width is a multiple of 16), the history is sorted: one lower bound per
      // thread and tile.  Until round 6 they were dropped when appended, and the first threshold counted `history length` columns
      // more: a user of 700 items found none in a tile of 896 threads, overflowed, and took the exact pass with all his 700 f32 rows
      // (1M x 100K: 14 490 instead of 9 820 users there, 96 ms instead of 18).
      const bool zero_hist = P.exclude_history && (P.cyc_nb == 0 || (P.cyc_nb & 15) == 0);
      if (zero_hist && active) {
        const int g0 = rpu_l2g(P, t0 + 16 * tid), nh = he - hs;
        int lo = 0, hi = nh;
        unsigned mask = 0;
        if (hist_lds) {
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_hist[mid] < g0) lo = mid + 1; else hi = mid;
          }
          for (int k = lo; k < nh; ++k) {
            const int d = s_hist[k] - g0;
            if (d >= 16) break;
            mask |= 1u << d;
          }
        } else {
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (P.x_idx[hs + mid] < g0) lo = mid + 1; else hi = mid;
          }
          for (int k = lo; k < nh; ++k) {
            const int d = P.x_idx[hs + k] - g0;
            if (d >= 16) break;
            mask |= 1u << d;
          }
        }
        if (mask) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if ((mask >> e) & 1u) r[e] = 0.0f;
        }
      }

      // ---- an item filter (ExcludeItems / SelectItems of the reference's pipelines): the same way - sixteen bytes of the mask per thread
      if (P.keep && active) {
        const bool run16 = P.cyc_nb == 0 || (P.cyc_nb & 15) == 0;
        const int g0 = rpu_l2g(P, t0 + 16 * tid);
        if (run16 && (g0 & 3) == 0 && 16 * tid + 16 <= n_cols) {
          const uint32_t* kw = reinterpret_cast<const uint32_t*>(P.keep + g0);
          uint32_t k4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) k4[q] = kw[q];
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (((k4[e >> 2] >> (8 * (e & 3))) & 0xFFu) == 0) r[e] = 0.0f;
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (16 * tid + e < n_cols && P.keep[run16 ? g0 + e : rpu_l2g(P, t0 + 16 * tid + e)] == 0) r[e] = 0.0f;
        }
      }

      if (!have_tau) {
        // ---- a lower bound of the Kc-th best score: every thread's best, of those the R-th best of every wave that owns columns.
        // Where the history's columns could not be cleared above (a cyclic slab of odd panels) the best columns may be history items
        // (dropped when they are appended): as many more are counted.
        int n_hist = P.exclude_history && !zero_hist ? he - hs : 0;  // (short histories: all of them counted - no pass, no barrier)
        if (P.exclude_history && !zero_hist && he - hs > 128) {
          if (tid == 0) s_cnt[1] = 0;
          __syncthreads();
          int c = 0;
          for (int k = hs + tid; k < he; k += RPU_THREADS) {
            int g = P.x_idx[k], l;
            if (P.cyc_nb) {
              const int p = g / P.cyc_nb;
              l = (rp_cyc_owner(p, P.cyc_w) == P.cyc_r) ? (p / P.cyc_w) * P.cyc_nb + g % P.cyc_nb : -1;
            } else {
              l = g - P.col_begin;
            }
            c += ((unsigned)(l - t0) < (unsigned)n_cols) ? 1 : 0;
          }
          c = rp_wave_sum(c);
          if (lane == 0 && c) atomicAdd((int*)&s_cnt[1], c);
          __syncthreads();
          n_hist = s_cnt[1];
        }
        // (round 6) which wave counts down how many of its threads' best scores?  cw = the wave's threads with a score; every wave
        // takes its share of `need` in proportion to it, R = ceil(need x cw / sum of cw): the same quantile everywhere, the shares add
        // up to `need` or more, and no wave is asked for more than it has.  (Until round 5 all waves counted the same R.  A tile whose
        // last wave owns four threads - 200 000 items: 13 tiles of 15 424 columns, 964 threads - never got a threshold, handed on every
        // column and sent ALL users to the exact pass; at 50 000 items - 784 threads, sixteen in the last wave - that wave counted down
        // to its 15th of 16 and the threshold fell to the floor: a quarter of the users overflowed their list.)
        const int need = Kc + n_hist;
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (r[e] != 0.0f) m = fmaxf(m, r[e]);
        const int cw = __popcll(__ballot(m > -INFINITY));
        if (lane == 0) s_cw[wave] = cw;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) tot += s_cw[q];
        const bool level = tot >= need;  // (else: fewer than `need` threads of the tile have a score)
        const int R = level ? (int)(((long long)need * cw + tot - 1) / tot) : 0;
        float rth = INFINITY;
        for (int it = 0; it < R; ++it) {
          float mx = m;
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
          rth = mx;
          const unsigned long long hit = __ballot(m == mx);
          if (lane == __ffsll((long long)hit) - 1) m = -INFINITY;
        }
        if (lane == 0) s_tau[wave] = level ? rth : -INFINITY;  // (a wave that counts nothing leaves the minimum alone: +inf)
        __syncthreads();
        float t = s_tau[0];
#pragma unroll
        for (int q = 1; q < NW; ++q) t = fminf(t, s_tau[q]);
        if (t > -INFINITY && t < INFINITY) {  // (fewer than `need` scores in the tile: it hands on everything, the next one tries again)
          tau = t;
          have_tau = true;
        }
      }
      // ---- the candidates of this tile, from the registers
      {
        bool any = false;
#pragma unroll
        for (int e = 0; e < 16; ++e) any |= (r[e] != 0.0f && r[e] >= tau);
        if (any) {
          int mine = 0;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float v = r[e];
            if (v != 0.0f && v >= tau) {
              const int g = rpu_l2g(P, t0 + 16 * tid + e);
              if (!P.exclude_history || zero_hist || !(hist_lds ? rpu_in_history(s_hist, he - hs, g) : rpu_in_history(P.x_idx + hs, he - hs, g))) {
                const int pos = atomicAdd((int*)&s_cnt[0], 1);
                ++mine;
                if (pos < RPU_CAP) {
                  lcol[pos] = g;
                  lval[pos] = v;
                }
              }
            }
          }
          if (mine) atomicAdd((int*)&s_cnt[4 + (tile & 1)], mine);
        }
      }
      __syncthreads();
      // The list's length, the SAME number in every thread: s_cnt[0] may already be moving - a wave that needs no
      // threshold pass goes straight on to the next tile's appends - but the counter of this tile's parity stands still until the
      // tile after the next, which nobody reaches before the next barrier; the other parity's was read behind the previous barrier.
      const int n_mine = s_cnt[4 + (tile & 1)];
      const int n = n_mine + n_other;
      n_other = n_mine;
      if (n > RPU_CAP) {  // (more candidates than the list holds - scores that tie by the thousand: the exact pass serves this user)
        failed = true;
        break;
      }
      const bool last = tile == T - 1;
      if (n > RPU_PRUNE || last) {
        // ---- the list's Kc best entries; they become the list (and its smallest entry the threshold), or - after the last tile - the
        // rescoring's candidates
        rpu_list_src src{lcol_g, lval_g, n};
        const int kp = rp_select_topk<RPU_THREADS>(src, n, Kc, sel);
        if (last) {
          const long long o = (long long)slot * Kc;
          rp_select_emit<RPU_THREADS>(n, sel, [&](int pos, int w) {
            P.out_idx[o + pos] = lcol[w];
            P.out_val[o + pos] = lval[w];
          });
          if (tid == 0) P.out_nnz[slot] = kp;
        } else {
          rp_select_emit<RPU_THREADS>(n, sel, [&](int pos, int w) {
            tcol[pos] = lcol[w];
            tval[pos] = lval[w];
          });
          __syncthreads();
          float mn = INFINITY;
          if (tid < kp) {
            lcol[tid] = tcol[tid];
            mn = lval[tid] = tval[tid];
          }
          if (tid < RPU_KC_MAX) {  // (the waves that hold entries)
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mn = fminf(mn, __shfl_xor(mn, d));
            if (lane == 0) s_tau[wave] = mn;
          }
          if (tid == 0) {
            s_cnt[0] = kp;
            s_cnt[4 + (tile & 1)] = kp;
            s_cnt[5 - (tile & 1)] = 0;
          }
          n_other = kp;
          __syncthreads();
          if (kp == Kc) {
            float t = s_tau[0];
#pragma unroll
            for (int q = 1; q < RPU_KC_MAX / 64; ++q) t = fminf(t, s_tau[q]);
            tau = fmaxf(tau, t);
            have_tau = true;
          }
        }
      }
    }
    if (failed && tid == 0) {
      P.out_nnz[slot] = -1;  // (the rescoring skips this user)
      P.fail_list[atomicAdd(P.n_fail, 1)] = P.fail_base + ul;
    }
  }
}

size_t rp_predict_user_i8_lds() { return (size_t)RPU_CAP * 8 + (size_t)RPU_KC_MAX * 8 + (size_t)RPU_HIST * 4 + rp_select_smem_bytes(RPU_CAP); }

int rp_predict_user_i8_geometry(int64_t n_cols, int* n_tiles, int* tile_w) {
  // sixteen columns per thread: at most 16 384 per tile; tiles of equal width, a multiple of 64 (the rows' padding)
  const int64_t T = std::max<int64_t>(1, rp_cdiv(n_cols, 16 * RPU_THREADS));
  *n_tiles = (int)T;
  *tile_w = (int)(rp_cdiv(rp_cdiv(n_cols, T), 64) * 64);
  return RPHIP_OK;
}

int rp_predict_user_i8(rphip_ctx* ctx, const rp_i8_params& P0, int32_t* ticket_dev, bool leave_room) {
  if (P0.Kc < 1 || P0.Kc > RPU_KC_MAX) RP_FAIL(RPHIP_ELIMIT, "first pass per user: Kc = %d outside 1..%d", P0.Kc, RPU_KC_MAX);
  rp_i8_params P = P0;
  P.ticket = ticket_dev;
  const size_t lds = rp_predict_user_i8_lds();
  static bool attr_set = false;
  if (!attr_set) {
    RP_HIP(hipFuncSetAttribute((const void*)k_predict_user_i8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  RP_HIP(hipMemsetAsync(ticket_dev, 0, 4, ctx->stream));
  // two workgroups per CU fill every thread slot of the chip; leave_room: a quarter of the CUs get one - the strips of a panel gather
  // that arrive beside this pass (rphip_ease_model_allgather) are unpacked by kernels that need somewhere to run (with the chip full
  // they started when this pass ended: no overlap at all - measured with the loop-back communicator)
  const int grid = (int)std::min<int64_t>(P.n_users, (int64_t)ctx->n_cu * 2 - (leave_room ? ctx->n_cu / 4 : 0));
  hipLaunchKernelGGL(k_predict_user_i8, dim3(grid), dim3(RPU_THREADS), lds, ctx->stream, P);
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}

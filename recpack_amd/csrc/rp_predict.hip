// K8: predict for neighbour-list models: scores = X @ S with a fused per-user top-Kp.
//
// Replaces ItemSimilarityMatrixAlgorithm._predict (recpack/algorithms/base.py:237-255, `X @ similarity_matrix_`,
// scipy csr_matmat on one thread).  With Kp the kernel also applies what the reference does to the scores
// afterwards: optional history removal (recpack/pipelines/pipeline.py:174-175) and the per-user top-K of
// MetricTopK (recpack/metrics/base.py:189 -> recpack/util.py:50-77).  The U x I score matrix is never stored.
//
//   task = (user u, column tile t)
//   accumulate  neighbour lists: for every history item i of u the <= K (j, s_ij) pairs of S row i (contiguous,
//               coalesced) are added into a dense tile of 64-bit FIXED-POINT sums in LDS with integer LDS atomics,
//               one wave per history item; dense models (K9): rows of B streamed into per-thread f32 registers
//   select      exact top-Kp by (score desc, column asc)  (rp_select.h), emitted in column order
//
// Fixed point: an LDS f32 atomic add sustains 0.33 per clock per CU on MI355X, u32/u64 integer adds 4.4-4.8
// (tools/lds_atomic_bench.hip).  Scores of neighbour-list models are therefore summed as signed 64-bit integers,
// value * 2^shift with shift chosen per call from the model's largest |similarity| and the longest history so that
// no sum can overflow (rp_fix_shift).  The integer sum is exact and independent of the order of the adds: predict
// is bit-reproducible, and the f32 similarities themselves are the only rounding before the final conversion.
//
// The "full" variants return every non-zero score (the reference's CSR result) in two phases: count, fill.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "rp_comm.h"
#include "rp_common.h"
#include "rp_predict_i8.h"
#include "rp_merge_big.h"
#include "rp_select.h"
#include "rp_wave.h"

constexpr int RP_PRED_THREADS = 1024;
constexpr int RP_PRED_FIX_TILE = 15360;  // columns of an 8-byte tile + selection buffers in 160 KB of LDS
constexpr unsigned long long RP_KEY_FLIP = 0x8000000000000000ull;  // two's complement sum -> order-preserving u64 key

// Scale of the sums.  |score| <= max_deg * vmax_abs < 2^(e + dl): shift <= 62 - e - dl keeps every sum inside 63 bits;
// shift <= 50 - e keeps every single term below 2^50 units so that the conversion below is one f64 add.
static int rp_fix_shift(double vmax_abs, int max_deg) {
  int e = 0;
  if (vmax_abs > 0.0 && std::isfinite(vmax_abs)) (void)std::frexp(vmax_abs, &e);  // vmax_abs < 2^e
  int dl = 0;
  while ((1ll << dl) < (long long)std::max(max_deg, 1)) ++dl;
  return std::min(200, std::max(-100, std::min(62 - e - dl, 50 - e)));
}

// f32 -> round(value * 2^shift) as a two's complement 64-bit integer: adding 1.5 * 2^(52-shift) in f64 leaves the
// integer in the low mantissa bits (|value| * 2^shift < 2^50 by the choice of shift; round to nearest even).
struct rp_fix_scale {
  double magic;
  unsigned long long magic_bits;
  int shift;
};
static rp_fix_scale rp_fix_make(int shift) {
  rp_fix_scale f;
  f.shift = shift;
  f.magic = std::ldexp(1.5, 52 - shift);
  std::memcpy(&f.magic_bits, &f.magic, 8);
  return f;
}
__device__ __forceinline__ unsigned long long rp_f32_to_fix(int bits, const rp_fix_scale& f) {
  const double r = (double)__int_as_float(bits) + f.magic;
  return (unsigned long long)__double_as_longlong(r) - f.magic_bits;
}
// a weighted term (similarity * the entry's value, valued prediction matrices): the float64 product, same conversion
__device__ __forceinline__ unsigned long long rp_f64_to_fix(double v, const rp_fix_scale& f) {
  const double r = v + f.magic;
  return (unsigned long long)__double_as_longlong(r) - f.magic_bits;
}
__device__ __forceinline__ double rp_fix_to_f64(unsigned long long sum, int shift) { return ldexp((double)(long long)sum, -shift); }

struct rp_pred_params {
  const int32_t* x_ptr;   // prediction input (CSR; binary unless x_val)
  const int32_t* x_idx;
  const double* x_val;    // optional values of the entries (time-decayed histories, time_aware_item_knn/base.py:139-150)
  const int32_t* s_idx;   // model: [I*K]
  const float* s_val;
  const int32_t* s_nnz;
  const float* dense_B;   // K9: dense model (EASE), row-major with leading dimension dense_ld; NULL for neighbour lists
  long long dense_ld;
  const _Float16* dense_B16;  // HALF instances: fp16(B * b16_scale), same leading dimension (the first pass of the two-pass K9)
  float b16_inv;              // 1 / b16_scale (a power of two)
  const int8_t* dense_B8;     // I8 instances: int8 rows (same leading dimension, in bytes), entry = q * b8_scale[row]
  const float* b8_scale;
  int tile_cap;               // HALF instances, mode 0: > 0 = a tile hands on EVERY column at or above its lower bound of the Kp-th best
                              // score (at most tile_cap of them; lists of that stride) instead of selecting exactly Kp
  int col_begin;          // dense column slab: column 0 of dense_B is item col_begin (n_items = the slab's width then)
  int cyc_nb, cyc_w, cyc_r;  // cyc_nb > 0: the slab's columns are block-cyclic panels (rphip_ease_fit_dist): local column l is item
                             // rp_cyc_panel(l / cyc_nb, cyc_r, cyc_w) * cyc_nb + l % cyc_nb (rp_common.h)
  const uint8_t* keep;    // optional item filter (mode 0): only items with keep[j] != 0 are candidates
  int n_items, n_tiles, tile_w, K, Kp, exclude_history;
  rp_fix_scale fix;       // neighbour lists: scale of the fixed-point sums
  int user_begin, n_users;
  const int32_t* user_list;  // optional: local user ids to process (mode 0); count in *n_list (n_list NULL: n_users of them)
  int list_compact;          // with user_list and several tiles: the per-tile candidate lists are indexed by the POSITION in the list
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

// item of a slab's local column / local column of an item (-1: not in this slab)
__device__ __forceinline__ int rp_pred_l2g(const rp_pred_params& P, int l) {
  return P.cyc_nb ? rp_cyc_panel(l / P.cyc_nb, P.cyc_r, P.cyc_w) * P.cyc_nb + l % P.cyc_nb : P.col_begin + l;
}
__device__ __forceinline__ int rp_pred_g2l(const rp_pred_params& P, int g) {
  if (!P.cyc_nb) return g - P.col_begin;
  const int p = g / P.cyc_nb;
  return (rp_cyc_owner(p, P.cyc_w) == P.cyc_r) ? (p / P.cyc_w) * P.cyc_nb + g % P.cyc_nb : -0x40000000;
}

// tile of f32 scores (dense models) or of fixed-point sums (neighbour lists)
template <bool DENSE>
struct rp_score_src {
  const void* acc_;
  int t0;  // first LOCAL column of the tile (ties go to the smaller column: local order is item order)
  const uint8_t* keep;
  int sh;  // leading key bits shared by all candidates (rp_key_prefix, rp_wave.h): the radix digits start below them
  const rp_pred_params* P;
  float tau;  // DENSE: scores below it are no candidates (a lower bound of the K-th best score, see k_predict_tile; -inf = off)
  __device__ __forceinline__ bool valid(int w) const {
    if (DENSE) {
      const float v = ((const float*)acc_)[w];
      return v != 0.0f && v >= tau && (!keep || keep[rp_pred_l2g(*P, t0 + w)] != 0);
    }
    return ((const unsigned long long*)acc_)[w] != 0ull && (!keep || keep[rp_pred_l2g(*P, t0 + w)] != 0);
  }
  __device__ __forceinline__ uint64_t key(int w) const {
    if (DENSE) {  // order-preserving map of a float (any sign) to u32, placed in the high half
      uint32_t b = __float_as_uint(((const float*)acc_)[w]);
      b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
      return ((uint64_t)b << 32) << sh;
    }
    return (((const unsigned long long*)acc_)[w] ^ RP_KEY_FLIP) << sh;
  }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)(t0 + w); }
};

typedef _Float16 rp_h8 __attribute__((ext_vector_type(8)));

// HALF (dense models only): the rows are read from the model's fp16 copy - half the bytes of the HBM-bound K9 - and summed in
// f32: APPROXIMATE scores, good for a candidate list that rp_predict_rescore then scores exactly (see rp_predict_topk_impl).
// HC8 (HALF only): 16-byte column groups per thread.  The default covers RP_TILE_MAX columns with one workgroup per CU (its LDS
// tile); HC8 = 2 (tiles of at most 16384 columns, half the LDS, at most 64 registers) lets TWO workgroups share a CU: one streams
// rows while the other selects - with fp16 rows the selection and the per-task start-up are a third of the first pass.
// I8 (HALF only, round 4): the rows come from the model's int8 copy - q = rint(b / s_row), s_row = rowmax / 127 - a quarter of B's bytes:
// sixteen columns per 16-byte load, score += float(q) * s_row.  The candidate list has to be longer for the same proof (the bound
// is the sum of the rows' half-steps), tools/k9_fp8_gate.py.
template <bool DENSE, bool HALF = false, int HC8 = 0, bool I8 = false>
__global__ void __launch_bounds__(RP_PRED_THREADS, HC8 == 2 ? 8 : 4) k_predict_tile(rp_pred_params P) {
  static_assert(DENSE || !HALF, "the fp16 first pass is for dense models");
  static_assert(HALF || !I8, "the int8 rows are a first pass");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ACC_B = DENSE ? 4 : 8;
  float* accf = reinterpret_cast<float*>(smem);
  unsigned long long* accq = reinterpret_cast<unsigned long long*>(smem);
  rp_select_smem sel = rp_select_carve(smem + (size_t)P.tile_w * ACC_B, P.tile_w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = RP_PRED_THREADS / 64;
  const int W = P.tile_w, T = P.n_tiles, K = P.K, shift = P.fix.shift;
  auto clear = [&](int w) {
    if (DENSE)
      accf[w] = 0.0f;
    else
      accq[w] = 0ull;
  };
  auto score = [&](int w) -> double { return DENSE ? (double)accf[w] : rp_fix_to_f64(accq[w], shift); };

  for (int w = tid; w < W; w += RP_PRED_THREADS) clear(w);
  __syncthreads();

  // modes 1/2 walk the tiles of a user inside the workgroup (output offsets are sequential);
  // mode 0 treats (user, tile) as independent tasks.
  const int n_u = (P.user_list && P.n_list) ? *P.n_list : P.n_users;
  const long long n_tasks = (P.mode == 0) ? (long long)n_u * T : (long long)n_u;
  for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
    const int uk = (P.mode == 0) ? (int)(task / T) : (int)task;
    const int ul = P.user_list ? P.user_list[uk] : uk;
    const int uo = (P.user_list && P.list_compact) ? uk : ul;  // index of the user's candidate lists
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
      if (HALF && I8) {
        // sixteen consecutive columns per 16-byte load, four rows in flight
        constexpr int C16 = HC8 ? (HC8 + 1) / 2 : (RP_TILE_MAX + 16 * RP_PRED_THREADS - 1) / (16 * RP_PRED_THREADS);
        float r[C16][16];
#pragma unroll
        for (int c = 0; c < C16; ++c)
#pragma unroll
          for (int e = 0; e < 16; ++e) r[c][e] = 0.0f;
        const int w16 = (n_cols + 15) >> 4;  // (reads past n_cols stay inside the padded row: its length is a multiple of 64)
#ifndef RP_K9_I8_RIF
#define RP_K9_I8_RIF 4  // (8 rows in flight: C4 predict 1.43 s against 1.33)
#endif
        constexpr int RIF = RP_K9_I8_RIF;  // rows in flight
        typedef int rp_i4 __attribute__((ext_vector_type(4)));
        // One 32-bit word at a time - four sign-extending byte conversions (v_cvt_f32_i32 sext(BYTE_n): one instruction each) and four
        // fmas - and the words strictly in order: the empty asm statements tie each word to the sums the previous one wrote.
        // Left to itself the compiler converted all 64 bytes of the four rows first and spilled 230 registers (two workgroups
        // per CU leave a thread 64).
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
        // NR rows in flight: their sixteen-byte pieces are requested together, then added one row at a time
        auto rows_block = [&](int h0, auto nr) __attribute__((always_inline)) {
          constexpr int NR = decltype(nr)::value;
          typedef const __attribute__((address_space(1))) int8_t* rp_gptr;  // (a GLOBAL pointer: rebuilt from an integer it would be a flat one)
          rp_gptr R[NR];
          float sc[NR];
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            const int it = __builtin_amdgcn_readfirstlane(P.x_idx[h0 + q]);  // (uniform: the row's base and its scale in scalar registers)
            const unsigned long long rb = reinterpret_cast<unsigned long long>(P.dense_B8 + (long long)it * P.dense_ld + t0);
            R[q] = (rp_gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rb >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)rb));
            sc[q] = P.b8_scale[it];
          }
          rp_i4 v[NR][C16];
#pragma unroll
          for (int c = 0; c < C16; ++c) {
            const int gq = tid + c * RP_PRED_THREADS;
            if (gq < w16) {
#pragma unroll
              for (int q = 0; q < NR; ++q)  // (scalar base + 32-bit offset: no address registers)
                v[q][c] = *reinterpret_cast<const __attribute__((address_space(1))) rp_i4*>(R[q] + (unsigned)(16 * gq));
            } else {
#pragma unroll
              for (int q = 0; q < NR; ++q) v[q][c] = (rp_i4){0, 0, 0, 0};
            }
          }
#pragma unroll
          for (int c = 0; c < C16; ++c)
#pragma unroll
            for (int q = 0; q < NR; ++q) add16(r[c], v[q][c], sc[q]);
        };
        int h = hs;
        for (; h + RIF - 1 < he; h += RIF) rows_block(h, std::integral_constant<int, RIF>{});
        if (RIF > 4 && h + 3 < he) {
          rows_block(h, std::integral_constant<int, 4>{});
          h += 4;
        }
        for (; h < he; ++h) {
          const int it = __builtin_amdgcn_readfirstlane(P.x_idx[h]);
          const int8_t* R0 = P.dense_B8 + (long long)it * P.dense_ld + t0;
          const float sc0 = P.b8_scale[it];
#pragma unroll
          for (int c = 0; c < C16; ++c) {
            const int gq = tid + c * RP_PRED_THREADS;
            if (gq < w16) add16(r[c], *reinterpret_cast<const rp_i4*>(R0 + 16 * gq), sc0);
          }
        }
#pragma unroll
        for (int c = 0; c < C16; ++c) {
          const int gq = tid + c * RP_PRED_THREADS;
          if (gq < w16) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (16 * gq + e < n_cols) accf[16 * gq + e] = r[c][e];
          }
        }
      } else if (HALF) {
        // eight consecutive columns per 16-byte load, four rows in flight
        constexpr int C8 = HC8 ? HC8 : (RP_TILE_MAX + 8 * RP_PRED_THREADS - 1) / (8 * RP_PRED_THREADS);
        float r[C8][8];
#pragma unroll
        for (int c = 0; c < C8; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) r[c][e] = 0.0f;
        const int w8 = (n_cols + 7) >> 3;  // (reads past n_cols stay inside the padded row)
        constexpr int RIF = HC8 == 2 ? 2 : 4;  // rows in flight (two workgroups per CU: half as many each)
        int h = hs;
        for (; h + RIF - 1 < he; h += RIF) {
          const _Float16* R[RIF];
#pragma unroll
          for (int q = 0; q < RIF; ++q) R[q] = P.dense_B16 + (long long)P.x_idx[h + q] * P.dense_ld + t0;
          rp_h8 v[RIF][C8];
#pragma unroll
          for (int c = 0; c < C8; ++c) {
            const int gq = tid + c * RP_PRED_THREADS;
            if (gq < w8) {
#pragma unroll
              for (int q = 0; q < RIF; ++q) v[q][c] = *reinterpret_cast<const rp_h8*>(R[q] + 8 * gq);
            } else {
#pragma unroll
              for (int q = 0; q < RIF; ++q) v[q][c] = (rp_h8)(_Float16)0.0f;
            }
          }
#pragma unroll
          for (int c = 0; c < C8; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if (RIF == 4)
                r[c][e] += ((float)v[0][c][e] + (float)v[1][c][e]) + ((float)v[2 % RIF][c][e] + (float)v[3 % RIF][c][e]);
              else
                r[c][e] += (float)v[0][c][e] + (float)v[1][c][e];
            }
        }
        for (; h < he; ++h) {
          const _Float16* R0 = P.dense_B16 + (long long)P.x_idx[h] * P.dense_ld + t0;
#pragma unroll
          for (int c = 0; c < C8; ++c) {
            const int gq = tid + c * RP_PRED_THREADS;
            if (gq < w8) {
              const rp_h8 v = *reinterpret_cast<const rp_h8*>(R0 + 8 * gq);
#pragma unroll
              for (int e = 0; e < 8; ++e) r[c][e] += (float)v[e];
            }
          }
        }
#pragma unroll
        for (int c = 0; c < C8; ++c) {
          const int gq = tid + c * RP_PRED_THREADS;
          if (gq < w8) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (8 * gq + e < n_cols) accf[8 * gq + e] = r[c][e] * P.b16_inv;
          }
        }
      } else if (DENSE) {
        // K9: score[u, t0:t0+n_cols] = sum of the history items' rows of B.  Every thread owns groups of four
        // consecutive columns (registers, no atomics); a row is streamed with one 16-byte load per group, two rows in
        // flight (row stride and tile offsets are multiples of 64 floats, the rows are padded to 128)
        constexpr int C4 = (RP_TILE_MAX + 4 * RP_PRED_THREADS - 1) / (4 * RP_PRED_THREADS);
        float4 r[C4];
#pragma unroll
        for (int c = 0; c < C4; ++c) r[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int w4 = (n_cols + 3) >> 2;  // groups in this tile (reads past n_cols stay inside the padded row)
        auto add_row = [&](const float* Bi) {
#pragma unroll
          for (int c = 0; c < C4; ++c) {
            const int gq = tid + c * RP_PRED_THREADS;
            if (gq < w4) {
              const float4 v = *reinterpret_cast<const float4*>(Bi + 4 * gq);
              r[c].x += v.x;
              r[c].y += v.y;
              r[c].z += v.z;
              r[c].w += v.w;
            }
          }
        };
        int h = hs;
        for (; h + 1 < he; h += 2) {
          const float* B0 = P.dense_B + (long long)P.x_idx[h] * P.dense_ld + t0;
          const float* B1 = P.dense_B + (long long)P.x_idx[h + 1] * P.dense_ld + t0;
          float4 v0[C4], v1[C4];
#pragma unroll
          for (int c = 0; c < C4; ++c) {
            const int gq = tid + c * RP_PRED_THREADS;
            v0[c] = v1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gq < w4) {
              v0[c] = *reinterpret_cast<const float4*>(B0 + 4 * gq);
              v1[c] = *reinterpret_cast<const float4*>(B1 + 4 * gq);
            }
          }
#pragma unroll
          for (int c = 0; c < C4; ++c) {  // same order as the one-row form: row h, then row h+1
            r[c].x = (r[c].x + v0[c].x) + v1[c].x;
            r[c].y = (r[c].y + v0[c].y) + v1[c].y;
            r[c].z = (r[c].z + v0[c].z) + v1[c].z;
            r[c].w = (r[c].w + v0[c].w) + v1[c].w;
          }
        }
        if (h < he) add_row(P.dense_B + (long long)P.x_idx[h] * P.dense_ld + t0);
#pragma unroll
        for (int c = 0; c < C4; ++c) {
          const int gq = tid + c * RP_PRED_THREADS;
          if (gq < w4) {
            const int w = 4 * gq;
            accf[w] = r[c].x;
            if (w + 1 < n_cols) accf[w + 1] = r[c].y;
            if (w + 2 < n_cols) accf[w + 2] = r[c].z;
            if (w + 3 < n_cols) accf[w + 3] = r[c].w;
          }
        }
      } else {
        for (int h = hs + wave; h < he; h += NW) {
          const int i = P.x_idx[h];
          const int nn = P.s_nnz[i];
          const int32_t* si = P.s_idx + (int64_t)i * K;
          const float* sv = P.s_val + (int64_t)i * K;
          if (P.x_val) {
            const double xw = P.x_val[h];
            for (int k = lane; k < nn; k += 64) {
              const int j = si[k] - t0;
              if ((unsigned)j < (unsigned)n_cols) atomicAdd(&accq[j], rp_f64_to_fix((double)sv[k] * xw, P.fix));
            }
            continue;
          }
          for (int k = lane; k < nn; k += 64) {
            const int j = si[k] - t0;
            if ((unsigned)j < (unsigned)n_cols) atomicAdd(&accq[j], rp_f32_to_fix(__float_as_int(sv[k]), P.fix));
          }
        }
      }
      __syncthreads();
      if (P.exclude_history) {
        for (int h = hs + tid; h < he; h += RP_PRED_THREADS) {
          const int j = rp_pred_g2l(P, P.x_idx[h]) - t0;
          if ((unsigned)j < (unsigned)n_cols) clear(j);
        }
        __syncthreads();
      }
      // ---- select (top-Kp: the scores of one user share their sign and most exponent bits - dropped first)
      rp_score_src<DENSE> src{smem, t0, P.mode == 0 ? P.keep : nullptr, 0, &P, -INFINITY};
      if (HALF && P.mode == 0 && !P.keep && P.Kp <= 16 * NW) {
        // A dense tile makes every column a candidate and the selection walks the whole tile once per radix level.  A lower bound
        // of the Kp-th best score first: every thread's best score, of those the R-th best of each wave (R = ceil(Kp / waves)): the
        // waves' R best threads are R * waves >= Kp different columns, so the Kp-th best score of the tile is at least the smallest
        // of these.  Only scores at or above it (a per cent of the tile) enter the candidate list; the selection's result is the same.
        const int R = (P.Kp + NW - 1) / NW;
        float m = -INFINITY;
        for (int w = 8 * tid; w < n_cols; w += 8 * RP_PRED_THREADS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = (w + e < n_cols) ? accf[w + e] : 0.0f;
            if (v != 0.0f) m = fmaxf(m, v);
          }
        }
        float rth = -INFINITY;
        for (int it = 0; it < R; ++it) {
          float mx = m;
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
          rth = mx;
          const unsigned long long hit = __ballot(m == mx);
          if (lane == __ffsll((long long)hit) - 1) m = -INFINITY;
        }
        float* s_tau = reinterpret_cast<float*>(&sel.misc[RPS_USER + 8]);
        if (lane == 0) s_tau[wave] = rth;
        __syncthreads();
        float t = s_tau[0];
#pragma unroll
        for (int q = 1; q < NW; ++q) t = fminf(t, s_tau[q]);
        src.tau = t;
        __syncthreads();
        if (P.tile_cap > 0 && T > 1) {
          // (round 4) ... and that is all the first pass needs: at least Kp columns of the tile reach tau, so the Kp-th best of the
          // user's merged lists does too, and a column below tau is below everything the merge keeps - the proof of the second pass
          // (k_predict_rescore: nothing outside the list beats its smallest entry) holds for the union of such lists as it does for
          // exact per-tile selections.  One pass over the tile instead of the key prefix, the compaction, the radix levels, the
          // classification and the emit: the selection was a third of the first pass's time once the rows were int8.
          int* s_cnt = &sel.misc[RPS_USER + 8 + NW];
          if (tid == 0) *s_cnt = 0;
          __syncthreads();
          const int64_t o = ((int64_t)uo * T + tile) * P.tile_cap;
          for (int w0 = 0; w0 < n_cols; w0 += RP_PRED_THREADS) {
            const int w = w0 + tid;
            const float v = (w < n_cols) ? accf[w] : 0.0f;
            const bool take = v != 0.0f && v >= t;
            const unsigned long long m = __ballot(take);
            if (m) {
              int base = 0;
              if (lane == 0) base = atomicAdd(s_cnt, __popcll(m));
              base = __shfl(base, 0);
              const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
              if (take && pos < P.tile_cap) {
                P.out_idx[o + pos] = rp_pred_l2g(P, t0 + w);
                P.out_val[o + pos] = v;
              }
            }
          }
          __syncthreads();
          const int n_take = *s_cnt;
          if (n_take <= P.tile_cap) {  // (no clearing: the next tile's sums overwrite every column, and the barrier above is behind the last read)
            if (tid == 0) P.out_nnz[uo * T + tile] = n_take;
            total += n_take;
            continue;
          }
          // (more than the list holds: the exact selection below writes the tile's best Kp over the list's head)
        }
      }
      if (P.mode == 0) {
        uint32_t* s_red = reinterpret_cast<uint32_t*>(&sel.misc[RPS_USER]);
        if (tid == 0) {
          s_red[0] = s_red[1] = 0xffffffffu;
          s_red[2] = s_red[3] = 0u;
        }
        __syncthreads();
        src.sh = rp_key_prefix<RP_PRED_THREADS>(src, n_cols, s_red);
      }
      const int kk = (P.mode == 0) ? P.Kp : 0x7fffffff;
      const int kp = rp_select_topk<RP_PRED_THREADS>(src, n_cols, kk, sel);
      if (P.mode == 0) {
        const int urow = (P.list_compact == 2) ? uo : ul;  // (2: one tile and a user list - the output rows go by list position too)
        const int64_t o = (T == 1) ? (int64_t)urow * P.Kp : ((int64_t)uo * T + tile) * (P.tile_cap > 0 ? P.tile_cap : P.Kp);
        rp_select_emit<RP_PRED_THREADS>(n_cols, sel, [&](int pos, int w) {
          P.out_idx[o + pos] = rp_pred_l2g(P, t0 + w);
          P.out_val[o + pos] = (float)score(w);
        });
        if (tid == 0) P.out_nnz[(T == 1) ? urow : uo * T + tile] = kp;
      } else if (P.mode == 2) {
        rp_select_emit<RP_PRED_THREADS>(n_cols, sel, [&](int pos, int w) {
          P.full_idx[off + pos] = rp_pred_l2g(P, t0 + w);
          P.full_val[off + pos] = score(w);
        });
        off += kp;
      }
      total += kp;
      if (P.keep && P.mode == 0) {  // filtered-out columns are not in the candidate list: clear the whole tile
        __syncthreads();
        for (int w = tid; w < n_cols; w += RP_PRED_THREADS) clear(w);
      } else {
        rp_select_foreach<RP_PRED_THREADS>(n_cols, sel, [&](int w) { clear(w); });  // clear what was touched
      }
      __syncthreads();
    }
    if (P.mode == 1 && tid == 0) P.row_nnz[ul] = total;
  }
}

// ---- the two-pass K9: fp16 copy of B, exact rescoring of the candidates ---------------------------------------------------------
// K9 streams 4 * I * nnz bytes of B from HBM (PMC: at the achievable rate), so the bytes are what is left to cut.  First pass: the
// same kernel on an fp16 copy of B (half the bytes) keeps the Kc > Kp best APPROXIMATE scores of every user.  Second pass
// (k_predict_rescore): the candidates' exact f32 scores - the same sums in the same order as the one-pass kernel's, bit for bit -
// their exact top-Kp, and a PROOF that no other column could have been among them: a column outside the list has an approximate
// score <= a_Kc (the Kc-th), hence an exact one <= a_Kc + E_u, where E_u = sum over the history of the row's largest rounding error
// (+ the f32 summation slack) bounds |approximate - exact| for this user; if the Kp-th exact score of the list is above that, the
// list's top-Kp IS the exact top-Kp.  A user whose list fails the test (measured: < 0.1 % at Kc = Kp + 64, tools/k9_half_gate.py)
// goes to a list that the one-pass kernel then serves.  The results are those of the one-pass kernel, bit for bit.
__global__ void __launch_bounds__(256) k_dense_rowmax(const float* __restrict__ B, long long ld, int n_rows, int n_cols, float* __restrict__ rowmax,
                                                      unsigned* __restrict__ gmax) {
  __shared__ float red[4];
  for (int i = blockIdx.x; i < n_rows; i += gridDim.x) {
    const float* row = B + (long long)i * ld;
    float m = 0.0f;
    for (int j = threadIdx.x * 4; j < n_cols; j += 1024) {  // (rows are padded to a multiple of 4 floats; the padding holds zeros)
      const float4 v = *reinterpret_cast<const float4*>(row + j);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      rowmax[i] = m;
      if (m > 0.0f) atomicMax(gmax, __float_as_uint(m));
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_dense_to_half(const float* __restrict__ B, long long ld, int n_rows, int n_cols8, float scale,
                                                       _Float16* __restrict__ H, float* __restrict__ delta) {
  __shared__ float red[4];
  const float inv = 1.0f / scale;
  for (int i = blockIdx.x; i < n_rows; i += gridDim.x) {
    const float* row = B + (long long)i * ld;
    _Float16* out = H + (long long)i * ld;
    float m = 0.0f;
    for (int j = threadIdx.x * 8; j < n_cols8; j += 2048) {
      const float4 a = *reinterpret_cast<const float4*>(row + j), b = *reinterpret_cast<const float4*>(row + j + 4);
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      rp_h8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        h[e] = (_Float16)(x[e] * scale);
        m = fmaxf(m, fabsf((float)h[e] * inv - x[e]));
      }
      *reinterpret_cast<rp_h8*>(out + j) = h;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) delta[i] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
  }
}

// the int8 copy: q = rint(b / s), s = rowmax / 127, one workgroup per row in two sweeps (the second one finds the row in L2);
// delta = the row's largest |q s - b| as measured
__global__ void __launch_bounds__(256) k_dense_to_i8(const float* __restrict__ B, long long ld, int n_rows, int n_cols, int8_t* __restrict__ Q,
                                                     float* __restrict__ scale, float* __restrict__ delta, float* __restrict__ rowmax) {
  __shared__ float red[4];
  auto wg_max = [&](float m) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    __syncthreads();  // (red is free again)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  };
  const int n16 = n_cols;  // (a multiple of 16 inside the padded row; the padding holds zeros)
  for (int i = blockIdx.x; i < n_rows; i += gridDim.x) {
    const float* row = B + (long long)i * ld;
    int8_t* out = Q + (long long)i * ld;
    float m = 0.0f;
    {  // (round 6) eight 16-byte loads in flight per thread: with one per iteration the sweep ran at 2.5 TB/s
      int j = threadIdx.x * 4;
      for (; j + 7 * 1024 < n16; j += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(row + j + q * 1024);
#pragma unroll
        for (int q = 0; q < 8; ++q) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[q].x), fabsf(v[q].y)), fmaxf(fabsf(v[q].z), fabsf(v[q].w))));
      }
      for (; j < n16; j += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(row + j);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
    }
    m = wg_max(m);
    const bool ok = m > 0.0f && m < 3.0e38f;
    const float s = ok ? m / 127.0f : 0.0f, inv = ok ? 127.0f / m : 0.0f;
    float err = 0.0f;
    for (int j = threadIdx.x * 16; j < n16; j += 4096) {
      uint32_t w[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(row + j + 4 * g);
        const float x[4] = {v.x, v.y, v.z, v.w};
        uint32_t pk = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float qf = fminf(fmaxf(rintf(x[e] * inv), -127.0f), 127.0f);
          err = fmaxf(err, fabsf(qf * s - x[e]));
          pk |= ((uint32_t)(int)qf & 0xffu) << (8 * e);
        }
        w[g] = pk;
      }
      *reinterpret_cast<uint4*>(out + j) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    err = wg_max(err);
    if (threadIdx.x == 0) {
      scale[i] = s;
      delta[i] = ok ? err : 3.0e38f;  // (a row with an infinite entry proves nothing: its users take the exact pass)
      rowmax[i] = m;
    }
  }
}

struct rp_rescore_params {
  const int32_t* x_ptr;
  const int32_t* x_idx;
  const float* B;
  long long ld;
  int col_begin, cyc_nb, cyc_w, cyc_r;
  const float* delta;
  const float* rowmax;
  const int32_t* cand_idx;  // [n][Kc] approximate top-Kc (any order), global column ids
  const float* cand_val;
  const int32_t* cand_nnz;
  int user_begin, n_users, Kp, Kc;
  int32_t* out_idx;
  float* out_val;
  int32_t* out_nnz;
  int32_t* fail_list;  // users whose list does not prove its top-Kp (ul + fail_base); count in *n_fail
  int fail_base;
  const int32_t* user_list;  // optional (mode 0): row k of the candidate arrays belongs to user user_begin + user_list[k]; the outputs are the user's
  int32_t* n_fail;
  // (round 5) the rescoring split between the owners of the candidates' columns (rphip_predict_topk_owner), MODE of the kernel:
  //   1  B holds THIS rank's columns only (col_begin / n_own_cols or cyc_*): the exact sums of the candidates it owns go to
  //      part_out[ul][Kc] (0 for the others and for what the skip rule leaves out), nothing is ranked;
  //   2  nothing is gathered: a candidate's exact sum is the sum over the n_parts ranks' parts part_in[s][n_users][Kc] - one of them
  //      holds it, the others 0 - then the ranking and the proof as in mode 0.  Both modes take the skip decisions from the same
  //      code on the same numbers, so every rank leaves out the same candidates.
  float* part_out;
  const float* part_in;
  int n_parts;
  int n_own_cols;  // mode 1 without a block-cyclic layout: the rank owns columns [col_begin, col_begin + n_own_cols)
};

// one wave per user; NQ = candidates per lane
template <int NQ, int MODE = 0>
__global__ void __launch_bounds__(256) k_predict_rescore(rp_rescore_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* se = reinterpret_cast<float*>(smem) + (size_t)wv * 2 * 64 * NQ;  // exact scores, then columns
  int* sc = reinterpret_cast<int*>(se + 64 * NQ);
  for (int row = blockIdx.x * 4 + wv; row < P.n_users; row += gridDim.x * 4) {
    const int ul = (MODE == 0 && P.user_list) ? P.user_list[row] : row;
    const int u = P.user_begin + ul;
    const int hs = P.x_ptr[u], he = P.x_ptr[u + 1];
    const int nc = P.cand_nnz[row];
    if (nc < 0) continue;  // (the first pass per user gave this one to the fallback list itself: his candidate list overflowed)
    int col[NQ], loc[NQ];
    bool mine[NQ];  // (mode 1: the candidate's column is this rank's)
    float e[NQ], av[NQ];
    float amin = 3.0e38f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int k = lane + 64 * q;
      const bool on = k < nc;
      col[q] = on ? P.cand_idx[(long long)row * P.Kc + k] : -1;
      av[q] = on ? P.cand_val[(long long)row * P.Kc + k] : -3.0e38f;
      if (on) amin = fminf(amin, av[q]);
      int l = 0;
      bool own = true;
      if (on) {
        if (P.cyc_nb) {
          const int p = col[q] / P.cyc_nb;
          l = (p / P.cyc_w) * P.cyc_nb + col[q] % P.cyc_nb;
          if (MODE == 1) own = rp_cyc_owner(p, P.cyc_w) == P.cyc_r;  // (block-cyclic with reflected odd rounds, rp_common.h)
        } else {
          l = col[q] - P.col_begin;
          if (MODE == 1) own = (unsigned)l < (unsigned)P.n_own_cols;
        }
      }
      loc[q] = (MODE == 1 && !own) ? 0 : l;
      mine[q] = on && own;
      e[q] = 0.0f;
    }
    // the bound of |approximate - exact| for this user
    float eu = 0.0f, rm = 0.0f;
    for (int h = hs + lane; h < he; h += 64) {
      const int i = P.x_idx[h];
      eu += P.delta[i];
      rm += P.rowmax[i];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      eu += __shfl_xor(eu, d);
      rm += __shfl_xor(rm, d);
      amin = fminf(amin, __shfl_xor(amin, d));
    }
    // (f32 summation slack of both passes: d terms of magnitude <= rowmax each; and a margin on the bound itself)
    const float E = 1.0009765625f * (eu + 2.0f * (float)(he - hs) * 5.9604645e-8f * rm);
    // a column outside the list: approximate score <= a_Kc, or exactly zero (never a candidate) - and a list shorter than Kc holds
    // every column whose approximate score is not zero
    const float a_bound = (nc == P.Kc) ? fmaxf(amin, 0.0f) : 0.0f;
    // (round 4) Candidates that cannot be among the best Kp are not scored: with a_Kp the Kp-th largest APPROXIMATE score of the
    // list, Kp candidates have exact scores >= a_Kp - E, and one with a_j + E < a_Kp - E lies below all of them.  The gathers -
    // a 64-byte line per candidate and history item - are the cost of this kernel, and the list is as long as the hardest users
    // need (the median user of the int8 first pass needs 133 of its 192).  What was skipped is treated like a column outside the
    // list: the proof below also asks the Kp-th exact score to beat the best skipped bound.
    bool skip[NQ];
    float a_skip = -3.0e38f;
    {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        se[lane + 64 * q] = av[q];
        skip[q] = false;
      }
      RP_WAVE_FENCE();
      if (nc > P.Kp) {
        int gt[NQ], ge[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) gt[q] = ge[q] = 0;
        for (int k = 0; k < 64 * NQ; ++k) {
          const float ak = se[k];
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            gt[q] += ak > av[q] ? 1 : 0;
            ge[q] += ak >= av[q] ? 1 : 0;
          }
        }
        float a_kp = -3.0e38f;  // the value with fewer than Kp above it and at least Kp at or above it
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (gt[q] < P.Kp && P.Kp <= ge[q]) a_kp = fmaxf(a_kp, av[q]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) a_kp = fmaxf(a_kp, __shfl_xor(a_kp, d));
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          skip[q] = col[q] >= 0 && av[q] + E < a_kp - E;
          if (skip[q]) a_skip = fmaxf(a_skip, av[q]);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) a_skip = fmaxf(a_skip, __shfl_xor(a_skip, d));
      }
      RP_WAVE_FENCE();
    }
    if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) skip[q] = skip[q] || !mine[q];  // (a_skip keeps the value every rank computes: all candidates)
    }
    // exact scores: the one-pass kernel's sums, same order (history order, one add per row)
    int h = (MODE == 2) ? he : hs;
    if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int k = lane + 64 * q;
        if (k < nc && !skip[q]) {
          float sum = 0.0f;  // (one part holds the score, the others +0: the sum is that score, bit for bit)
          for (int sp = 0; sp < P.n_parts; ++sp) sum += P.part_in[((long long)sp * P.n_users + ul) * P.Kc + k];
          e[q] = sum;
        }
      }
    }
    for (; h + 3 < he; h += 4) {
      const float* R0 = P.B + (long long)P.x_idx[h] * P.ld;
      const float* R1 = P.B + (long long)P.x_idx[h + 1] * P.ld;
      const float* R2 = P.B + (long long)P.x_idx[h + 2] * P.ld;
      const float* R3 = P.B + (long long)P.x_idx[h + 3] * P.ld;
      float v0[NQ], v1[NQ], v2[NQ], v3[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        v0[q] = v1[q] = v2[q] = v3[q] = 0.0f;
        if (!skip[q]) {
          v0[q] = R0[loc[q]];
          v1[q] = R1[loc[q]];
          v2[q] = R2[loc[q]];
          v3[q] = R3[loc[q]];
        }
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) e[q] = (((e[q] + v0[q]) + v1[q]) + v2[q]) + v3[q];
    }
    for (; h < he; ++h) {
      const float* R0 = P.B + (long long)P.x_idx[h] * P.ld;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (!skip[q]) e[q] += R0[loc[q]];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (skip[q]) e[q] = 0.0f;  // (not a candidate any more: a zero score is none)
    if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int k = lane + 64 * q;
        if (k < P.Kc) P.part_out[(long long)ul * P.Kc + k] = e[q];
      }
      RP_WAVE_FENCE();
      continue;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool ok = col[q] >= 0 && e[q] != 0.0f;  // (a score that cancels to exactly zero is not stored)
      se[lane + 64 * q] = ok ? e[q] : -3.0e38f;
      sc[lane + 64 * q] = ok ? col[q] : 0x7fffffff;
    }
    RP_WAVE_FENCE();
    // rank by (score desc, column asc) among the list
    const int n_all = 64 * NQ;
    int rank[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) rank[q] = 0;
    for (int k = 0; k < n_all; ++k) {
      const float ek = se[k];
      const int ck = sc[k];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float mine_e = (col[q] >= 0 && e[q] != 0.0f) ? e[q] : -3.0e38f;
        const int mine_c = (col[q] >= 0 && e[q] != 0.0f) ? col[q] : 0x7fffffff;
        rank[q] += (ek > mine_e || (ek == mine_e && ck < mine_c)) ? 1 : 0;
      }
    }
    int n_valid = 0;
    float e_kp = -3.0e38f;  // the Kp-th exact score
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool ok = col[q] >= 0 && e[q] != 0.0f;
      n_valid += __popcll(__ballot(ok));
      if (ok && rank[q] == P.Kp - 1) e_kp = e[q];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) e_kp = fmaxf(e_kp, __shfl_xor(e_kp, d));
    const bool proven = n_valid >= P.Kp && e_kp > fmaxf(a_bound, a_skip) + E;
    if (!proven) {
      if (lane == 0) P.fail_list[atomicAdd(P.n_fail, 1)] = P.fail_base + ul;
      RP_WAVE_FENCE();
      continue;
    }
    // the selected ones in column order
    RP_WAVE_FENCE();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool selq = col[q] >= 0 && e[q] != 0.0f && rank[q] < P.Kp;
      sc[lane + 64 * q] = selq ? col[q] : 0x7fffffff;
    }
    RP_WAVE_FENCE();
    int pos[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) pos[q] = 0;
    for (int k = 0; k < n_all; ++k) {
      const int ck = sc[k];
#pragma unroll
      for (int q = 0; q < NQ; ++q) pos[q] += (ck < col[q]) ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (col[q] >= 0 && e[q] != 0.0f && rank[q] < P.Kp) {
        P.out_idx[(long long)ul * P.Kp + pos[q]] = col[q];
        P.out_val[(long long)ul * P.Kp + pos[q]] = e[q];
      }
    }
    if (lane == 0) P.out_nnz[ul] = P.Kp;
    RP_WAVE_FENCE();
  }
}

// ---- merge of per-tile top-Kp candidates -----------------------------------------------------------
struct rp_flist_src {
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
                                                                  int32_t* __restrict__ out_nnz, int sort_cols, int stride = 0, int compact = 0) {
  // compact (with user_list): the candidate lists of the uk-th user of the list are the uk-th ones (n_list NULL: n_users entries)
  // stride: entries between two lists of a user (0: Kp; larger when the lists may hold more than Kp entries each)
  if (stride <= 0) stride = Kp;
  // sort_cols: the lists are not consecutive column ranges (the ranks' block-cyclic panels): the selected entries are ranked by
  // column before they are written (Kp * 8 more bytes of LDS in front); otherwise list order is column order
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  char* smem = smem_all + (sort_cols ? (size_t)Kp * 8 : 0);
  int32_t* scol = reinterpret_cast<int32_t*>(smem_all);
  float* sval = reinterpret_cast<float*>(smem_all + (size_t)Kp * 4);
  const int T = n_tiles, cap = T * stride;
  int32_t* lcol = reinterpret_cast<int32_t*>(smem);
  float* lval = reinterpret_cast<float*>(smem + (size_t)cap * 4);
  rp_select_smem sel = rp_select_carve(smem + (size_t)cap * 8, cap);
  int* s_off = &sel.misc[RPS_USER];
  const int tid = threadIdx.x;
  const int n_u = (user_list && n_list) ? *n_list : n_users;
  for (int uk = blockIdx.x; uk < n_u; uk += gridDim.x) {
    const int ul = user_list ? user_list[uk] : uk;
    const int uc = (user_list && compact) ? uk : ul;
    __syncthreads();
    if (tid == 0) {
      int o = 0;
      for (int t = 0; t < T; ++t) {
        s_off[t] = o;
        o += cand_nnz[(int64_t)uc * T + t];
      }
      s_off[T] = o;
    }
    __syncthreads();
    const int n = s_off[T];
    for (int t = 0; t < T; ++t) {
      const int cn = s_off[t + 1] - s_off[t];
      const int64_t so = ((int64_t)uc * T + t) * stride;
      for (int k = tid; k < cn; k += RP_PRED_THREADS) {
        lcol[s_off[t] + k] = cand_idx[so + k];
        lval[s_off[t] + k] = cand_val[so + k];
      }
    }
    __syncthreads();
    rp_flist_src src{lcol, lval, n};
    const int kp = rp_select_topk<RP_PRED_THREADS>(src, n, Kp, sel);
    if (sort_cols) {
      rp_select_emit<RP_PRED_THREADS>(n, sel, [&](int pos, int w) {
        scol[pos] = lcol[w];
        sval[pos] = lval[w];
      });
      __syncthreads();
      for (int p = tid; p < kp; p += RP_PRED_THREADS) {  // (columns are distinct: every item is in one list)
        const int c = scol[p];
        int r = 0;
        for (int q = 0; q < kp; ++q) r += scol[q] < c;
        out_idx[(int64_t)ul * Kp + r] = c;
        out_val[(int64_t)ul * Kp + r] = sval[p];
      }
    } else {
      rp_select_emit<RP_PRED_THREADS>(n, sel, [&](int pos, int w) {
        out_idx[(int64_t)ul * Kp + pos] = lcol[w];
        out_val[(int64_t)ul * Kp + pos] = lval[w];
      });
    }
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
//   * scores are accumulated as 64-bit fixed-point sums with u64 LDS atomics (see the top of this file);
//   * a 2048-bit dirty map (one word per lane, LDS atomic OR) records the touched columns; after the tile's adds
//     each lane walks the set bits of its word, reads and clears the sums and offers them to the wave's running
//     top-Kp pool (rp_wave.h: exact radix-select cuts; admission is exact for any arrival order);
//   * at the end the pool is cut to Kp, ranked by column and written out (score = (float)(sum * 2^-shift)).
//
// Two flavours of the sums (template parameter AccT), both integer, both independent of the order of the adds:
//   uint64_t  one scale per call (rp_fix_shift): exact - the f32 storage of the similarities is the only rounding.  2048-column
//             tiles (16 KB of LDS per wave).  Valued prediction matrices (WGT) and RPHIP_PREDICT_EXACT64 run here.
//   uint32_t  one scale PER USER: with M_u = sum over the history of the list's largest |similarity| (rowmax, an upper bound of
//             any |score| of that user) and 2^(e-1) <= M_u (1 + 2^-12) < 2^e the terms are rint(s_ij * 2^(30-e)).  No sum can
//             leave 31 bits (|sum| <= M_u 2^(30-e) + d_u / 2 < 2^30 + 2^29); a score is off by at most d_u * 2^(e-31)
//             <= d_u * M_u * 2^-30 - for a 64-item history 6e-8 of the user's score BOUND, the size of the f32 rounding of the
//             similarities themselves and of the f32 output value.  Half the LDS per column buys 4096-column tiles at the
//             same occupancy: half the (user, tile) pairs, whose fixed cost is what the kernel spends its time on.
constexpr int PW_WT = 2048;     // columns per tile per dirty word of a lane (a tile is PW_WT * DW columns)
constexpr int PW_CH = 4;        // entries per chunk (32 bytes)
constexpr int PW_UCH = 4;       // users per ticket (16 measured: 45.3 against 44.8 ms - 250 000 ticket atomics in 45 ms do not contend)
constexpr int PW_HEAVY = 192;   // users with longer histories are handed out first (a 1700-item history is 40 typical users of work:
                                // picked up at the end of the queue it would run on alone)
constexpr int PW_KP_MAX = 384;  // pool capacity = Kp rounded up + 128 <= 512

// LDS of one wave without the pool: sums, dirty map + history map, cut histogram (also the chunk-owner scratch), gather list
__host__ __device__ constexpr size_t rp_pw_fixed_bytes(int acc_bytes, int wt) {
  return (size_t)wt * acc_bytes + 512 * (wt / PW_WT) + 1024 + 128;
}
// Pool capacity: every admitted candidate sits in the pool until the next exact cut back to Kp entries, and a cut costs about
// as much as ten tiles - so the pool takes all the LDS that `waves` waves per CU leave (at most 512 entries, the cut's limit;
// at least Kp rounded up + 64, what a push needs).  A stream of N scores admits about Kp (1 + ln(N / Kp)) of them.
static int rp_pw_cap_for(int Kp, int acc_bytes, int wt, int waves) {
  const int base = ((Kp + 63) / 64) * 64;
  const long long budget = (160 * 1024) / waves - (long long)rp_pw_fixed_bytes(acc_bytes, wt);
  const int fit = (int)(budget / (acc_bytes + 4)) / 16 * 16;
  return std::max(base + 64, std::min(512, fit));
}
template <typename T>
__device__ __forceinline__ constexpr T rp_pw_flip() { return (T)1 << (8 * sizeof(T) - 1); }  // two's complement sum -> ordered key

struct rp_pw_params {
  const int32_t* x_ptr;
  const int32_t* x_idx;
  const int2* pairs;
  const uint16_t* toff;  // [I * (n_wt + 1)]
  const float* rowmax;   // [I] largest |similarity| of each neighbour list (the per-user scale of the 32-bit sums)
  const uint8_t* keep;   // optional item filter: only items with keep[j] != 0 are candidates
  const double* x_val;   // optional values of the prediction matrix's entries (WGT instances)
  int n_items, n_wt, K, Kp, exclude_history, user_begin, n_users;
  int cap;               // pool capacity (rp_pw_cap_for)
  const int32_t* heavy;  // local ids of the users with more than PW_HEAVY history items (they go first), count in *n_heavy
  const int32_t* n_heavy;
  rp_fix_scale fix;
  int32_t* ticket;
  int32_t* out_idx;
  float* out_val;
  int32_t* out_nnz;
};


typedef int v4i_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <typename AccT>
struct rp_pw_state {
  RP_LDS AccT* acc;
  RP_LDS uint32_t* dirty;
  rp_wpool<AccT> pool;
  const int* pairs_i;
  int lane, cap;
  rp_fix_scale fix;  // 64-bit sums: the call's scale
  float fmul;        // 32-bit sums: 2^ushift, this user's scale
  int ushift;
};


template <typename AccT, bool WGT, bool MAP = true>
__device__ __forceinline__ void rp_pw_add_chunk(rp_pw_state<AccT>& S, int n, const v4i_a4& a, const v4i_a4& b, int t0, double xw) {
  const int ec[4] = {a.x, a.z, b.x, b.z};
  const int ev[4] = {a.y, a.w, b.y, b.w};
  // Straight-line code, no per-entry branch (with one the compiler computed each entry's operands behind an
  // s_waitcnt lgkmcnt(0), i.e. one full LDS round trip per entry).  An entry past the chunk's end becomes (column `lane` of
  // the tile, similarity 0, no map bit): it adds 0 to a word of the lane's own and ORs 0 into the lane's own map word - three
  // selects per entry (column, value, bit) instead of four, and no two lanes meet on one address (all idle entries on ONE
  // column serialise in the LDS: measured +38 % on the whole kernel).
#pragma unroll
  for (int e = 0; e < PW_CH; ++e) {
    const bool on = e < n;
    const int j = on ? ec[e] - t0 : S.lane;
    const int bits = on ? ev[e] : 0;  // (+0.0f)
    AccT v;
    if constexpr (sizeof(AccT) == 4)  // this user's scale: rint(s * 2^ushift), exact integer adds from here on
      v = (AccT)__float2int_rn(__int_as_float(bits) * S.fmul);
    else if (WGT)  // the owner's value times the similarity, in float64
      v = rp_f64_to_fix((double)__int_as_float(bits) * xw, S.fix);
    else
      v = rp_f32_to_fix(bits, S.fix);
    atomicAdd((AccT*)&S.acc[j], v);
    if constexpr (MAP) {
      const uint32_t bit = on ? (1u << ((j >> 6) & 31)) : 0u;
      atomicOr((uint32_t*)&S.dirty[(j & 63) + 64 * (j >> 11)], bit);  // word (j mod 64) of the tile's (j / 2048)-th map
    }
  }
}

// take a sum out of the accumulator: returns it and leaves 0 (one returning LDS operation; of several lanes that hold the same
// column exactly one gets the sum)
__device__ __forceinline__ uint32_t rp_lds_exch(RP_LDS uint32_t* p) { return atomicExch((unsigned int*)p, 0u); }
__device__ __forceinline__ uint64_t rp_lds_exch(RP_LDS uint64_t* p) { return (uint64_t)atomicExch((unsigned long long*)p, 0ull); }

// chunk c of a batch whose lanes hold (start, length, first chunk) of their slices: owner lane by binary search
// over the exclusive chunk offsets, then the chunk's two 16-byte loads
template <typename AccT, bool WGT>
__device__ __forceinline__ int rp_pw_fetch_chunk(const rp_pw_state<AccT>& S, int c, int n_chunks, int s, int len, int choff, v4i_a4* a,
                                                 v4i_a4* b, double wt, double* wo) {
  int lo = 0;
#pragma unroll
  for (int step = 32; step > 0; step >>= 1) {
    const int v = __shfl(choff, lo + step);
    if (v <= c) lo += step;
  }
  const int us = __shfl(s, lo), ulen = __shfl(len, lo), co = __shfl(choff, lo);
  if (WGT) *wo = __shfl(wt, lo);
  const int k0 = (c - co) * PW_CH;
  const int n = (c < n_chunks) ? min(PW_CH, ulen - k0) : 0;
  if (n > 0) {
    const int* src = S.pairs_i + 2 * (long long)(us + k0);  // the pairs array is padded: reading past the
    *a = *reinterpret_cast<const v4i_a4*>(src);               // slice's end is harmless
    *b = *reinterpret_cast<const v4i_a4*>(src + 4);
  }
  return n;
}

// ---- the rare paths of a tile, kept OUT OF LINE on purpose.  Both load and immediately consume global data; inlined,
// their temporaries made the compiler put conservative s_waitcnt vmcnt(..) into the common path, which then waited for
// the prefetched entries of the next tile in the middle of the current one.  As calls they cost nothing unless taken.
//
// Further chunk steps of one pipelined batch (more than 64 chunks of the batch fall into the tile).
template <typename AccT, bool WGT>
__device__ __attribute__((noinline)) void rp_pw_extra_steps(RP_LDS AccT* acc, RP_LDS uint32_t* dirty, RP_LDS uint32_t* hist,
                                                            const int* pairs_i, rp_fix_scale fix, float fmul, int lane, int s, int len,
                                                            int choff, int nchunks, int t0, double wt, int c_from) {
  rp_pw_state<AccT> S;
  S.acc = acc;
  S.dirty = dirty;
  S.pool.hist = hist;
  S.pairs_i = pairs_i;
  S.fix = fix;
  S.fmul = fmul;
  S.lane = lane;
  for (int c0 = c_from; c0 < nchunks; c0 += 64) {
    v4i_a4 a = (v4i_a4){0, 0, 0, 0}, b2 = a;
    double xw = 1.0;
    const int n = rp_pw_fetch_chunk<AccT, WGT>(S, c0 + lane, nchunks, s, len, choff, &a, &b2, wt, &xw);
    rp_pw_add_chunk<AccT, WGT>(S, n, a, b2, t0, xw);
  }
}

// the item filter for the four slots of a walk round (only lanes whose sum would enter the pool look the item up)
__device__ __attribute__((noinline)) uint32_t rp_pw_keep_lookup(const uint8_t* keep, int c0, int c1, int c2, int c3, uint32_t want) {
  uint32_t ok = 0;
  if ((want & 1u) && keep[c0] != 0) ok |= 1u;
  if ((want & 2u) && keep[c1] != 0) ok |= 2u;
  if ((want & 4u) && keep[c2] != 0) ok |= 4u;
  if ((want & 8u) && keep[c3] != 0) ok |= 8u;
  return ok;
}

// One user.  The work is a sequence of UNITS (column tile t, batch g of 64 history items), tile-major; after the last batch
// of a tile its touched columns are offered to the pool.  The sequence is software-pipelined three units deep, the same
// way for every history length:
//   iteration k   requests  the item ids of unit k+3                      (one coalesced load)
//                           the tile offsets of unit k+2 (toff[item][t], [t+1]; needs the ids of k+2)
//                           the entries of unit k+1: scan of the slice lengths, chunk -> lane assignment, ST steps of two
//                           16-byte loads per lane (needs the offsets of k+1)
//                 then adds the entries of unit k (requested one iteration earlier) and, at a tile's end, walks the tile.
// Everything an iteration requests is consumed at the top of the next one at the earliest, i.e. after the adds and the walk
// - the long part of an iteration - so the loads have that long to arrive.  All of them are UNCONDITIONAL (clamped
// addresses; dead units and idle lanes read something harmless): with loads under branches the compiler loses count of
// the loads in flight and falls back to s_waitcnt vmcnt(0) in front of the adds, which waits for the entries just requested.
// Two entry sets alternate (the loop body handles two units), so no register is copied while its load is in flight.
// ST steps of 64 chunks of 4 entries are prefetched per unit (3 for 4096-column tiles: a full batch has 64 x 8.2 entries in
// about 160 chunks); chunks beyond them (rare) are fetched on the spot by rp_pw_extra_steps.
// History removal: a lane whose item lies in the current tile sets its bit in `excl` (same layout as the dirty map); the walk
// skips and clears those columns.
template <int ST>
struct rp_pw_uset {
  int s, len, choff, nchunks;
  int n[ST];
  v4i_a4 a[ST], b[ST];
  double w[ST];  // WGT: the value of the history item the lane's chunk belongs to
};

// ONE: the history fits one batch (at most 64 items), i.e. every unit is a whole tile and the unit's entries - still in
// registers after the adds - ARE the list of touched columns: the walk exchanges each entry's sum against 0 (the first lane to
// arrive at a column takes the sum, later ones read 0), so there is no dirty map to maintain in the adds, no bit scanning and no
// imbalance between lanes in the walk (a lane whose map word holds nine bits made the whole wave run three rounds).
template <typename AccT, int WT, bool WGT, int ST, bool ONE>
__device__ __forceinline__ void rp_pw_user_units(const rp_pw_params& P, rp_pw_state<AccT>& S, int hs, int he) {
  constexpr int DW = WT / PW_WT;
  constexpr AccT FLIP = rp_pw_flip<AccT>();
  const int lane = S.lane, K = P.K, Kp = P.Kp, NT = P.n_wt, cap = S.cap;
  const int G = (he - hs + 63) >> 6;
  const int NU = NT * G;
  RP_LDS uint32_t* excl = S.dirty + 64 * DW;
  rp_pw_uset<ST> set0, set1;

  // ---- stage state.  (t, g) of units k .. k+3; ids / weights of k .. k+2; raw (just loaded) values of the youngest stages
  int t0c = 0, g0c = 0, t1c = 0, g1c = 0, t2c = 0, g2c = 0, t3c = 0, g3c = 0;
  auto next_unit = [&](int& t, int& g) __attribute__((always_inline)) {
    ++g;
    if (g == G) {
      g = 0;
      ++t;
    }
  };
  int it0 = -1, it1 = -1, it2 = -1, raw_it = 0, raw_oa = 0, raw_ob = 0;
  bool valid3 = false;
  double wt0 = 1.0, wt1 = 1.0, wt2 = 1.0, raw_wt = 1.0;
  auto load_ids = [&](int t, int g) __attribute__((always_inline)) {
    const int h = hs + 64 * g + lane;
    raw_it = P.x_idx[min(h, he - 1)];  // (he > hs: the clamped read stays inside the user's row)
    if (WGT) raw_wt = P.x_val[min(h, he - 1)];
    valid3 = (h < he) && (t < NT);
  };
  auto load_offs = [&](int it, int t) __attribute__((always_inline)) {
    const uint16_t* to = P.toff + (long long)max(it, 0) * (NT + 1) + min(t, NT - 1);
    raw_oa = (int)to[0];
    raw_ob = (int)to[1];
  };
  // Entries of one unit.  Chunk c of the batch goes to lane c % 64 of step c / 64.  Owner of chunk c = the largest lane whose
  // first chunk is <= c: every owner writes its lane id at its first chunk's slot of a 64 ST-word LDS scratch (the cut's
  // histogram, idle here), a running maximum over the slots (DPP per step, carried from step to step) fills the gaps.  The
  // scratch reads of all steps go out together, then the owners' (start, length, first chunk) broadcasts of all steps: two LDS
  // round trips per unit whatever ST is.  Steps beyond the unit's chunks issue nothing.
  auto fetch = [&](rp_pw_uset<ST>& D, int it, int oa, int ob, double wt) __attribute__((always_inline)) {
    RP_LDS uint32_t* sc = S.pool.hist;
    D.s = (it >= 0) ? it * K + oa : 0;
    D.len = (it >= 0) ? ob - oa : 0;
    const int nch = (D.len + PW_CH - 1) / PW_CH;
    const int inc = rp_wave_incl_scan(nch, lane);
    D.choff = inc - nch;
    D.nchunks = __builtin_amdgcn_readlane(inc, 63);
#pragma unroll
    for (int st = 0; st < ST; ++st) sc[64 * st + lane] = 0u;
    RP_WAVE_FENCE();
    if (nch > 0 && D.choff < 64 * ST) sc[D.choff] = (uint32_t)lane;
    RP_WAVE_FENCE();
    int own[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) own[st] = (int)sc[64 * st + lane];
    int carry = 0;
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      own[st] = rp_wave_incl_max(max(own[st], (lane == 0) ? carry : 0));
      carry = __builtin_amdgcn_readlane(own[st], 63);
    }
    int us[ST], ulen[ST], co[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      us[st] = __shfl(D.s, own[st]);
      ulen[st] = __shfl(D.len, own[st]);
      co[st] = __shfl(D.choff, own[st]);
      if (WGT) D.w[st] = __shfl(wt, own[st]);
    }
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      const int c = 64 * st + lane;
      const int k0 = (c - co[st]) * PW_CH;
      const int n = (c < D.nchunks) ? min(PW_CH, ulen[st] - k0) : 0;
      D.n[st] = n;
      if (64 * st < D.nchunks) {  // (uniform; whatever is requested here is used at the top of the next iteration at the earliest)
        const int* src = S.pairs_i + 2 * (long long)(n > 0 ? us[st] + k0 : 0);  // (padded array: reading past a slice's end is harmless)
        D.a[st] = *reinterpret_cast<const v4i_a4*>(src);
        D.b[st] = *reinterpret_cast<const v4i_a4*>(src + 4);
      }
    }
  };
#pragma unroll
  for (int st = 0; st < ST; ++st) {
    set0.w[st] = set1.w[st] = 1.0;
    set0.n[st] = set1.n[st] = 0;
    set0.a[st] = set0.b[st] = set1.a[st] = set1.b[st] = (v4i_a4){0, 0, 0, 0};
  }
  set0.s = set0.len = set0.choff = set0.nchunks = 0;
  set1.s = set1.len = set1.choff = set1.nchunks = 0;

  // ---- prologue (dependent loads, once per user): ids of units 0..2, offsets of units 0..1, entries of unit 0
  {
    load_ids(0, 0);
    const int i0 = valid3 ? raw_it : -1;
    const double w0 = (WGT && valid3) ? raw_wt : 1.0;
    t1c = 0, g1c = 0;
    next_unit(t1c, g1c);  // unit 1
    load_ids(t1c, g1c);
    const int i1 = valid3 ? raw_it : -1;
    const double w1 = (WGT && valid3) ? raw_wt : 1.0;
    t2c = t1c, g2c = g1c;
    next_unit(t2c, g2c);  // unit 2
    load_offs(i0, 0);
    const int oa0 = (i0 >= 0) ? raw_oa : 0, ob0 = (i0 >= 0) ? raw_ob : 0;
    fetch(set0, i0, oa0, ob0, w0);  // entries of unit 0
    load_offs(i1, t1c);             // offsets of unit 1: consumed at the top of iteration 0
    load_ids(t2c, g2c);             // ids of unit 2: likewise
    // the loop's top rotates: (it0, it1, it2) <- (it1, it2, raw); so unit 0 sits in it1, unit 1 in it2
    it1 = i0;
    wt1 = w0;
    it2 = i1;
    wt2 = w1;
    // counters are rotated the same way: stage 1 holds unit 0, stage 2 unit 1, stage 3 unit 2
    t3c = t2c, g3c = g2c;
    t2c = t1c, g2c = g1c;
    t1c = 0, g1c = 0;
  }
  bool touched = false;

  // one iteration: C = entries of unit k (requested last iteration), N = set to fill for unit k+1
  auto unit = [&](rp_pw_uset<ST>& C, rp_pw_uset<ST>& N) __attribute__((always_inline)) {
    // ---- top: rotate the stages; what the previous iteration requested is used from here on.  The empty asm makes the
    // youngest of those loads (ids, offsets) "used" HERE: the compiler's wait for them then sits in front of this
    // iteration's requests - placed at their first arithmetic use it would come after the entry loads below and wait for
    // those too (loads return in order)
    asm volatile("" : "+v"(raw_it), "+v"(raw_oa), "+v"(raw_ob));
    if (WGT) asm volatile("" : "+v"(raw_wt));
    it0 = it1;
    wt0 = wt1;
    t0c = t1c, g0c = g1c;
    it1 = it2;
    wt1 = wt2;
    t1c = t2c, g1c = g2c;
    const int oa1 = (it1 >= 0) ? raw_oa : 0, ob1 = (it1 >= 0) ? raw_ob : 0;
    it2 = valid3 ? raw_it : -1;
    if (WGT) wt2 = valid3 ? raw_wt : 1.0;
    t2c = t3c, g2c = g3c;
    next_unit(t3c, g3c);
    // ---- requests for the units ahead
    fetch(N, it1, oa1, ob1, wt1);  // entries of unit k+1
    load_offs(it2, t2c);           // offsets of unit k+2
    load_ids(t3c, g3c);            // ids of unit k+3
    // ---- unit k: adds
    const int t0 = t0c * WT;
    touched |= C.nchunks > 0;
#pragma unroll
    for (int st = 0; st < ST; ++st)
      if (64 * st < C.nchunks) rp_pw_add_chunk<AccT, WGT, !ONE>(S, C.n[st], C.a[st], C.b[st], t0, C.w[st]);
    if (C.nchunks > 64 * ST)  // (rare; sets map bits, which the ONE form never reads: it scans such a tile densely)
      rp_pw_extra_steps<AccT, WGT>(S.acc, S.dirty, S.pool.hist, S.pairs_i, S.fix, S.fmul, lane, C.s, C.len, C.choff, C.nchunks, t0, wt0,
                                   64 * ST);
    if (!ONE && P.exclude_history) {
      const int j = it0 - t0;
      if ((unsigned)j < (unsigned)WT) atomicOr((uint32_t*)&excl[(j & 63) + 64 * (j >> 11)], 1u << ((j >> 6) & 31));
    }
    if (g0c != G - 1) return;  // more batches of this tile to come
    // ---- end of tile t0c: offer the touched columns to the pool.  Lane l walks the set bits of dirty word l (word l
    // covers columns l, l+64, ..: the lanes' reads of one round fall into different LDS banks), four set bits per lane and
    // round so that the sums of a round are read with one wait
    // one round of the walk: four slots per lane (has[q]: the slot holds a column, j[q] its index in the tile, v[q] its sum,
    // already taken out of the accumulator) are offered to the pool
    auto offer4 = [&](bool(&has)[4], int(&j)[4], AccT(&v)[4]) __attribute__((always_inline)) {
      bool kq[4];
      uint32_t live = 0;  // bit q: some lane keeps its slot q
      // most sums lose against the pool's threshold once it is warm: the slots of the round are voted on first and only
      // slots with a surviving lane are pushed (rolled loop: one inlined copy of the pool cut; values rotate through 0)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] ^= FLIP;  // signed sum -> unsigned order; a zero sum is no candidate
        kq[q] = has[q] && v[q] != FLIP && rp_wpool_admits<AccT, 0>(S.pool, v[q], (uint32_t)(t0 + j[q]));
      }
      if (P.keep) {  // item filter: only sums that would enter the pool pay the lookup
        const uint32_t want = (kq[0] ? 1u : 0u) | (kq[1] ? 2u : 0u) | (kq[2] ? 4u : 0u) | (kq[3] ? 8u : 0u);
        if (__any(want != 0u)) {
          const uint32_t ok = rp_pw_keep_lookup(P.keep, t0 + j[0], t0 + j[1], t0 + j[2], t0 + j[3], want);
#pragma unroll
          for (int q = 0; q < 4; ++q) kq[q] = (ok >> q) & 1u;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) live |= __any(kq[q]) ? (1u << q) : 0u;
      if (!live) return;
      // the round's survivors in one go when they fit the pool as it is (the common case: a few of them); slot by slot
      // with cuts in between otherwise
      const uint64_t m0 = __ballot(kq[0]), m1 = __ballot(kq[1]), m2 = __ballot(kq[2]), m3 = __ballot(kq[3]);
      const int c0n = __popcll(m0), c1n = __popcll(m1), c2n = __popcll(m2), c3n = __popcll(m3);
      if (S.pool.n + c0n + c1n + c2n + c3n <= cap) {
        const uint64_t below = (1ull << lane) - 1ull;
        int base = S.pool.n;
        if (kq[0]) {
          const int pos = base + __popcll(m0 & below);
          S.pool.key[pos] = v[0];
          S.pool.pay[pos] = (uint32_t)(t0 + j[0]);
        }
        base += c0n;
        if (kq[1]) {
          const int pos = base + __popcll(m1 & below);
          S.pool.key[pos] = v[1];
          S.pool.pay[pos] = (uint32_t)(t0 + j[1]);
        }
        base += c1n;
        if (kq[2]) {
          const int pos = base + __popcll(m2 & below);
          S.pool.key[pos] = v[2];
          S.pool.pay[pos] = (uint32_t)(t0 + j[2]);
        }
        base += c2n;
        if (kq[3]) {
          const int pos = base + __popcll(m3 & below);
          S.pool.key[pos] = v[3];
          S.pool.pay[pos] = (uint32_t)(t0 + j[3]);
        }
        S.pool.n = base + c3n;
        return;
      }
#pragma unroll 1
      for (int q = 0; q < 4 && live; ++q, live >>= 1) {
        if (live & 1u) {
          const uint32_t col = (uint32_t)(t0 + j[0]);
          // the threshold may have moved since the vote (a cut inside an earlier push): ask again
          const bool keep = kq[0] && rp_wpool_admits<AccT, 0>(S.pool, v[0], col);
          rp_wpool_push<AccT, 0>(S.pool, Kp, cap, keep, v[0], col, lane);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          kq[r] = kq[r + 1];
          j[r] = j[r + 1];
          v[r] = v[r + 1];
        }
      }
    };
    if constexpr (ONE) {
      RP_WAVE_FENCE();
      if (P.exclude_history) {  // the user's own items: their sums are dropped before anybody takes them
        const int jx = it0 - t0;
        if ((unsigned)jx < (unsigned)WT) S.acc[jx] = 0;
        RP_WAVE_FENCE();
      }
      if (C.nchunks > 64 * ST) {
        // more chunks than the prefetched steps hold (rp_pw_extra_steps added the rest): every column of the tile is looked at
#pragma unroll 1
        for (int r = 0; r < WT / 256; ++r) {
          bool has[4];
          int j[4];
          AccT v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            has[q] = true;
            j[q] = lane + 64 * (4 * r + q);
            v[q] = S.acc[j[q]];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) S.acc[j[q]] = 0;
          offer4(has, j, v);
        }
#pragma unroll
        for (int d = 0; d < DW; ++d) S.dirty[lane + 64 * d] = 0u;  // (the bits rp_pw_extra_steps set)
      } else {
#pragma unroll
        for (int st = 0; st < ST; ++st) {
          if (64 * st < C.nchunks) {  // (uniform)
            const int ec[4] = {C.a[st].x, C.a[st].z, C.b[st].x, C.b[st].z};
            bool has[4];
            int j[4];
            AccT v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              has[q] = q < C.n[st];
              j[q] = has[q] ? ec[q] - t0 : lane;  // (an idle slot points at a word of the lane's own: no two lanes meet there)
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = has[q] ? (AccT)rp_lds_exch(&S.acc[j[q]]) : (AccT)0;
            offer4(has, j, v);
          }
        }
      }
      return;
    }
    const bool was_touched = touched;
    touched = false;
    RP_WAVE_FENCE();
    // the lane's map words as ONE 64-bit word (DW <= 2): bit b is column lane + 64 b of the tile.  Walking the two words one
    // after the other ran as many rounds as the fullest lane of EACH needed; together the lanes' loads even out
    static_assert(DW <= 2, "one 64-bit map word per lane");
    {
      uint64_t ex = 0ull;
      if (P.exclude_history) {
        ex = excl[lane];
        excl[lane] = 0u;
        if (DW > 1) {
          ex |= (uint64_t)excl[lane + 64] << 32;
          excl[lane + 64] = 0u;
        }
      }
      if (was_touched) {
        uint64_t w = S.dirty[lane];
        S.dirty[lane] = 0u;
        if (DW > 1) {
          w |= (uint64_t)S.dirty[lane + 64] << 32;
          S.dirty[lane + 64] = 0u;
        }
        ex &= w;  // history items that were touched: their sums are dropped
        w &= ~ex;
        while (__any(ex != 0ull)) {
          if (ex != 0ull) S.acc[lane + 64 * (__ffsll((unsigned long long)ex) - 1)] = 0;
          ex &= ex - 1ull;
        }
        while (__any(w != 0ull)) {
          bool has[4];
          int j[4];
          AccT v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            has[q] = w != 0ull;
            j[q] = lane + 64 * (has[q] ? (__ffsll((unsigned long long)w) - 1) : 0);
            w &= w - 1ull;  // 0 stays 0
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = has[q] ? S.acc[j[q]] : (AccT)0;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (has[q]) S.acc[j[q]] = 0;
          offer4(has, j, v);
        }
      }
    }
  };
  for (int k = 0; k < NU; k += 2) {
    unit(set0, set1);
    if (k + 1 < NU) unit(set1, set0);
  }
}

// sum of a non-negative float over the wave (DPP row shifts + row broadcasts), returned in every lane
#define RP_DPP_F32(x, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ float rp_wave_sum_f32(float v) {
  v += RP_DPP_F32(v, RP_DPP_ROW_SHR(1), 0xf);
  v += RP_DPP_F32(v, RP_DPP_ROW_SHR(2), 0xf);
  v += RP_DPP_F32(v, RP_DPP_ROW_SHR(4), 0xf);
  v += RP_DPP_F32(v, RP_DPP_ROW_SHR(8), 0xf);
  v += RP_DPP_F32(v, RP_DPP_ROW_BCAST15, 0xa);
  v += RP_DPP_F32(v, RP_DPP_ROW_BCAST31, 0xc);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <typename AccT, int WT, bool WGT>
__global__ void __launch_bounds__(64) k_predict_wave(rp_pw_params P) {
  static_assert(sizeof(AccT) == 8 || !WGT, "valued prediction matrices use the 64-bit sums");
  constexpr int DW = WT / PW_WT;
  constexpr AccT FLIP = rp_pw_flip<AccT>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int Kp = P.Kp;
  rp_pw_state<AccT> S;
  S.lane = lane;
  S.cap = P.cap;
  S.fix = P.fix;
  S.fmul = 1.0f;
  S.ushift = P.fix.shift;
  S.acc = (RP_LDS AccT*)smem;
  S.dirty = (RP_LDS uint32_t*)(smem + (size_t)WT * sizeof(AccT));
  S.pool.key = (RP_LDS AccT*)(smem + (size_t)WT * sizeof(AccT) + 512 * DW);  // dirty map, then the history map (256 B per 2048 columns each)
  S.pool.pay = (RP_LDS uint32_t*)(S.pool.key + S.cap);
  S.pool.hist = S.pool.pay + S.cap;
  S.pool.gidx = (RP_LDS uint16_t*)(S.pool.hist + 256);
  S.pairs_i = reinterpret_cast<const int*>(P.pairs);

  for (int i = lane; i < WT; i += 64) S.acc[i] = 0;
#pragma unroll
  for (int d = 0; d < 2 * DW; ++d) S.dirty[lane + 64 * d] = 0u;

  while (true) {
    int ubase = 0;
    if (lane == 0) ubase = atomicAdd(P.ticket, PW_UCH);
    ubase = __shfl(ubase, 0);
    const int n_heavy = *P.n_heavy;
    if (ubase >= P.n_users + n_heavy) break;
    const int uend = min(ubase + PW_UCH, P.n_users + n_heavy);
    for (int q = ubase; q < uend; ++q) {
      // positions [0, n_heavy): the long histories; then every user in order, the long ones skipped
      const int ul = (q < n_heavy) ? P.heavy[q] : q - n_heavy;
      const int u = P.user_begin + ul;
      const int hs = P.x_ptr[u], he = P.x_ptr[u + 1];
      if (he <= hs) continue;
      if (q >= n_heavy && he - hs > PW_HEAVY) continue;  // done in the first part
      if constexpr (sizeof(AccT) == 4) {
        // this user's scale: M_u = sum of the history lists' largest |similarity| bounds every |score|;
        // 2^ushift * M_u (1 + 2^-12) < 2^30 (the margin covers the f32 rounding of the sum itself)
        float m = 0.0f;
        for (int h = hs + lane; h < he; h += 64) m += P.rowmax[P.x_idx[h]];
        m = rp_wave_sum_f32(m) * 1.000244140625f;
        if (!(m > 0.0f)) continue;  // every list empty: no score
        const int e = (int)((__float_as_uint(m) >> 23) & 255u) - 126;  // m < 2^e (m is normal: similarities are f32)
        S.ushift = max(-96, min(96, 30 - e));
        S.fmul = __uint_as_float((uint32_t)(S.ushift + 127) << 23);
      }
      S.pool.n = 0;
      S.pool.tau = 0;
      S.pool.tau_col = 0;
      if (he - hs <= 64)
        rp_pw_user_units<AccT, WT, WGT, (WT > PW_WT ? 3 : 2), true>(P, S, hs, he);
      else
        rp_pw_user_units<AccT, WT, WGT, (WT > PW_WT ? 3 : 2), false>(P, S, hs, he);
      // ---- final: best Kp by (score desc, column asc), then column order for the output row
      RP_WAVE_FENCE();
      if (S.pool.n > Kp) rp_wpool_cut<AccT, 0>(S.pool, Kp, lane);
      const int n_out = S.pool.n;
      RP_LDS AccT* pk = S.pool.key;
      RP_LDS uint32_t* pp = S.pool.pay;
      if (lane < 4) pp[n_out + lane] = 0xffffffffu;  // pads the last group of four (n_out + 4 <= cap)
      typedef uint32_t v4u __attribute__((ext_vector_type(4)));
      for (int e = lane; e < n_out; e += 64) {
        const uint32_t col = pp[e];
        int rank = 0;
        for (int o = 0; o < n_out; o += 4) {
          const v4u q = *reinterpret_cast<RP_LDS v4u*>(&pp[o]);
          rank += (q.x < col ? 1 : 0) + (q.y < col ? 1 : 0) + (q.z < col ? 1 : 0) + (q.w < col ? 1 : 0);
        }
        P.out_idx[(int64_t)ul * Kp + rank] = (int)col;
        if constexpr (sizeof(AccT) == 4)
          P.out_val[(int64_t)ul * Kp + rank] = (float)ldexp((double)(int32_t)(pk[e] ^ FLIP), -S.ushift);
        else
          P.out_val[(int64_t)ul * Kp + rank] = (float)rp_fix_to_f64(pk[e] ^ FLIP, S.fix.shift);
      }
      if (lane == 0) P.out_nnz[ul] = n_out;
      RP_WAVE_FENCE();
    }
  }
}

// local ids of the users whose history is longer than PW_HEAVY (any order), count in *n
__global__ void k_pw_heavy_users(const int32_t* __restrict__ x_ptr, int user_begin, int n_users, int32_t* __restrict__ list,
                                 int32_t* __restrict__ n) {
  for (int ul = blockIdx.x * blockDim.x + threadIdx.x; ul < n_users; ul += gridDim.x * blockDim.x) {
    const int u = user_begin + ul;
    if (x_ptr[u + 1] - x_ptr[u] > PW_HEAVY) list[atomicAdd(n, 1)] = ul;
  }
}

// toff[i][t] = number of entries of neighbour list i with column < t*wt (lists are column-sorted); rowmax[i] = the list's
// largest |similarity| (f32)
__global__ void k_build_toff(const int32_t* __restrict__ idx, const float* __restrict__ val32, const int32_t* __restrict__ nnz, int K,
                             int n_items, int n_wt, int wt, uint16_t* __restrict__ toff, float* __restrict__ rowmax) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) {
    uint16_t* o = toff + (long long)i * (n_wt + 1);
    const int32_t* row = idx + (long long)i * K;
    const float* rv = val32 + (long long)i * K;
    const int n = nnz[i];
    int t = 0;
    float mx = 0.0f;
    o[0] = 0;
    for (int k = 0; k < n; ++k) {
      const int tj = row[k] / wt;
      while (t < tj) o[++t] = (uint16_t)k;
      mx = fmaxf(mx, fabsf(rv[k]));
    }
    while (t < n_wt) o[++t] = (uint16_t)n;
    rowmax[i] = mx;
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
    const int ab = __float_as_int(v) & 0x7fffffff;  // |v| as ordered bits
    if (k < nnz[i / K] && ab > unsorted[1]) atomicMax(&unsorted[1], ab);
  }
}

int rp_model_finalize_topk(rphip_ctx* ctx, rphip_model* m) {
  const int64_t rows = m->row_end - m->row_begin;
  const int64_t n = rows * (int64_t)m->K;
  m->cols_sorted = true;
  if (n <= 0) return RPHIP_OK;
  int32_t* flag = nullptr;
  RP_TRY(rp_scratch(ctx, 5, 64, (void**)&flag));
  RP_HIP(hipMemsetAsync(flag, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_model_finalize, dim3((unsigned)std::min<int64_t>(4096, rp_cdiv(n, 256))), dim3(256), 0, ctx->stream, m->idx, m->val,
                     m->nnz, m->K, rows, m->val32, m->pairs, flag);
  RP_HIP(hipGetLastError());
  int32_t h[2] = {0, 0};
  RP_HIP(hipMemcpyAsync(h, flag, 8, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  m->cols_sorted = (h[0] == 0);
  std::memcpy(&m->vmax_abs, &h[1], 4);
  return RPHIP_OK;
}

// ---- host ------------------------------------------------------------------------------------------
static int rp_pred_check(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* X, int64_t ub, int64_t ue) {
  if (!ctx || !model || !X) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (model->kind != RPHIP_MODEL_TOPK && model->kind != RPHIP_MODEL_DENSE) RP_FAIL(RPHIP_ESTATE, "unknown model kind");
  if (model->is_gram) RP_FAIL(RPHIP_ESTATE, "this handle holds a Gram matrix (rphip_ease_gram_create), not a fitted model");
  if (model->kind == RPHIP_MODEL_DENSE && !model->B)
    RP_FAIL(RPHIP_ESTATE, "this model holds the int8 rows only (rphip_ease_model_allgather_i8): predict with rphip_predict_topk_owner");
  if (model->row_begin != 0 || model->row_end != model->n_items)
    RP_FAIL(RPHIP_ESTATE, "the model holds rows [%lld,%lld) of %lld: assemble all rows before predict", (long long)model->row_begin,
            (long long)model->row_end, (long long)model->n_items);
  if (X->n_items != model->n_items)
    RP_FAIL(RPHIP_EINVAL, "prediction matrix has %lld items, the model %lld", (long long)X->n_items, (long long)model->n_items);
  if (X->x_val && model->kind == RPHIP_MODEL_DENSE)
    RP_FAIL(RPHIP_EINVAL, "a valued prediction matrix with a dense model: the reference's EASE binarises its input (base.py:141-151)");
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
  P.dense_B16 = nullptr;
  P.b16_inv = 1.0f;
  P.dense_B8 = nullptr;
  P.b8_scale = nullptr;
  P.tile_cap = 0;
  P.keep = nullptr;
  P.n_items = (int)X->n_items;
  P.col_begin = 0;
  P.cyc_nb = P.cyc_w = P.cyc_r = 0;
  if (m->kind == RPHIP_MODEL_DENSE && m->n_cols > 0) {  // a column slab: the kernel walks the slab's own columns
    P.n_items = (int)m->n_cols;
    P.col_begin = (int)m->col_begin;
    if (m->cyc_nb > 0) {
      P.cyc_nb = m->cyc_nb;
      P.cyc_w = m->cyc_world;
      P.cyc_r = m->cyc_rank;
      P.col_begin = 0;
    }
    P.n_tiles = (int)std::max<int64_t>(1, rp_cdiv(m->n_cols, RP_TILE_MAX));
    P.tile_w = (int)(rp_cdiv(rp_cdiv(m->n_cols, P.n_tiles), 64) * 64);
  } else if (m->kind == RPHIP_MODEL_DENSE) {
    P.n_tiles = X->n_tiles;
    P.tile_w = X->tile_w;
  } else {  // 8-byte sums: narrower tiles than K0's plan
    P.n_tiles = (int)std::max<int64_t>(1, rp_cdiv(X->n_items, RP_PRED_FIX_TILE));
    P.tile_w = (int)(rp_cdiv(rp_cdiv(std::max<int64_t>(X->n_items, 1), P.n_tiles), 64) * 64);
  }
  P.fix = rp_fix_make(rp_fix_shift((double)m->vmax_abs * (X->x_val ? X->val_max_abs : 1.0), X->max_user_deg));
  P.x_val = (m->kind == RPHIP_MODEL_DENSE) ? nullptr : X->x_val;
  P.K = m->K;
  P.Kp = 0;
  P.exclude_history = 0;
  P.user_begin = (int)ub;
  P.n_users = (int)(ue - ub);
  P.user_list = nullptr;
  P.list_compact = 0;
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

static size_t rp_pred_lds(const rp_pred_params& P) { return (size_t)P.tile_w * (P.dense_B ? 4 : 8) + rp_select_smem_bytes(P.tile_w); }

static int rp_pred_launch(rphip_ctx* ctx, const rp_pred_params& P, int grid) {
  const size_t lds = rp_pred_lds(P);
  if (P.dense_B) {
    RP_HIP(hipFuncSetAttribute((const void*)k_predict_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_predict_tile<true>, dim3(grid), dim3(RP_PRED_THREADS), lds, ctx->stream, P);
  } else {
    RP_HIP(hipFuncSetAttribute((const void*)k_predict_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_predict_tile<false>, dim3(grid), dim3(RP_PRED_THREADS), lds, ctx->stream, P);
  }
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}

// algorithmic bytes of K8: 8*K*nnz_pred + 4*(nnz_pred+U) + 8*U*Kp
static double rp_pred_work(const rphip_csr* X, const rphip_model* m, int64_t n_users, int Kp) {
  const double share = X->n_users > 0 ? (double)n_users / (double)X->n_users : 0.0;
  if (m->kind == RPHIP_MODEL_DENSE)  // K9: 4*I*nnz_pred (rows of B streamed) + 8*U*Kp
    return 4.0 * (double)(m->n_cols > 0 ? m->n_cols : m->n_items) * (double)X->nnz * share + 8.0 * (double)n_users * Kp;
  return 8.0 * m->K * (double)X->nnz * share + 4.0 * ((double)X->nnz * share + (double)n_users) + 8.0 * (double)n_users * Kp;
}

// the model's fp16 (kind 1) or int8 (kind 2) copy (first use; a model is immutable once fitted - pruning rebuilds it; a context
// that asks for the other kind gets it rebuilt)
static int rp_model_half(rphip_ctx* ctx, const rphip_model* m, int kind) {
  if (m->b16_failed) return RPHIP_OK;
  if (m->B16 && m->b16_kind == kind) return RPHIP_OK;
  hipStream_t st = ctx->stream;
  if (m->B16) {
    RP_HIP(hipStreamSynchronize(st));
    rp_dev_free(ctx, m->B16);
    rp_dev_free(ctx, m->b16_delta);
    rp_dev_free(ctx, m->b16_rowmax);
    rp_dev_free(ctx, m->b8_scale);
    m->B16 = nullptr;
    m->b16_delta = m->b16_rowmax = m->b8_scale = nullptr;
    m->b16_kind = 0;
  }
  const int64_t rows = m->n_items;
  const int64_t cols = m->n_cols > 0 ? m->n_cols : m->n_items;
  if (m->ld % (kind == 2 ? 16 : 8)) {  // (16-byte loads of eight fp16 / sixteen int8)
    m->b16_failed = true;
    return RPHIP_OK;
  }
  const size_t esz = kind == 2 ? 1 : sizeof(_Float16);
  unsigned* gmax = nullptr;
  RP_TRY(rp_scratch(ctx, 5, 64, (void**)&gmax));
  {  // the copy is an optimisation: it must leave room for everything else (16 GiB or a tenth of the device, whichever is less)
    size_t free_b = 0, total_b = 0;
    hipMemGetInfo(&free_b, &total_b);
    const size_t need = (size_t)rows * m->ld * esz + std::min<size_t>((size_t)16 << 30, total_b / 10);
    if (free_b + ctx->cached_bytes < need) {
      // (round 6) after a fit at 200 000 items the solve's plane workspaces hold most of what the model leaves: a predict needs none
      // of them, the next fit allocates them again (the one-pass kernel it would take otherwise costs three times the first pass)
      RP_HIP(hipStreamSynchronize(st));
      RP_TRY(rp_scratch_release_planes(ctx));
      hipMemGetInfo(&free_b, &total_b);
    }
    if (free_b + ctx->cached_bytes < need) {
      m->b16_failed = true;
      return RPHIP_OK;
    }
  }
  if (rp_dev_alloc(ctx, (size_t)rows * m->ld * esz, (void**)&m->B16) != RPHIP_OK ||
      rp_dev_alloc(ctx, (size_t)rows * sizeof(float), (void**)&m->b16_delta) != RPHIP_OK ||
      rp_dev_alloc(ctx, (size_t)rows * sizeof(float), (void**)&m->b16_rowmax) != RPHIP_OK ||
      (kind == 2 && rp_dev_alloc(ctx, (size_t)rows * sizeof(float), (void**)&m->b8_scale) != RPHIP_OK)) {
    rp_dev_free(ctx, m->B16);
    rp_dev_free(ctx, m->b16_delta);
    rp_dev_free(ctx, m->b16_rowmax);
    rp_dev_free(ctx, m->b8_scale);
    m->B16 = nullptr;
    m->b16_delta = m->b16_rowmax = m->b8_scale = nullptr;
    m->b16_failed = true;  // not an error: the one-pass kernel gives the same result
    rp_set_error("%s", "");
    return RPHIP_OK;
  }
  m->b16_kind = kind;
  if (kind == 2) {
    const int grid8 = (int)std::min<int64_t>(rows, (int64_t)ctx->n_cu * 16);
    hipLaunchKernelGGL(k_dense_to_i8, dim3(grid8), dim3(256), 0, st, m->B, (long long)m->ld, (int)rows, (int)std::min<int64_t>(m->ld, rp_cdiv(cols, 16) * 16),
                       reinterpret_cast<int8_t*>(m->B16), m->b8_scale, m->b16_delta, m->b16_rowmax);
    RP_HIP(hipGetLastError());
    m->b16_scale = 1.0f;
    return RPHIP_OK;
  }
  RP_HIP(hipMemsetAsync(gmax, 0, 4, st));
  const int grid = (int)std::min<int64_t>(rows, (int64_t)ctx->n_cu * 16);
  const int cols4 = (int)(rp_cdiv(cols, 4) * 4), cols8 = (int)std::min<int64_t>(m->ld, rp_cdiv(cols, 8) * 8);
  hipLaunchKernelGGL(k_dense_rowmax, dim3(grid), dim3(256), 0, st, m->B, (long long)m->ld, (int)rows, cols4, m->b16_rowmax, gmax);
  unsigned h = 0;
  RP_HIP(hipMemcpyAsync(&h, gmax, 4, hipMemcpyDeviceToHost, st));
  RP_HIP(hipStreamSynchronize(st));
  float mx;
  memcpy(&mx, &h, 4);
  // the largest magnitude goes to [2^14, 2^15): fp16's normal range then reaches down to 2^-29 of it
  int e = 0;
  float scale = 1.0f;
  if (mx > 0.0f && std::isfinite(mx)) {
    (void)std::frexp(mx, &e);  // mx in [2^(e-1), 2^e)
    scale = std::ldexp(1.0f, std::max(-100, std::min(100, 15 - e)));
  }
  m->b16_scale = scale;
  hipLaunchKernelGGL(k_dense_to_half, dim3(grid), dim3(256), 0, st, m->B, (long long)m->ld, (int)rows, cols8, scale, m->B16, m->b16_delta);
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}

constexpr int RPHIP_KC2 = 256;  // candidates per user of the second list (the per-user pass holds no longer one)

template <int NQ, int MODE = 0>
static void rp_launch_rescore(rphip_ctx* ctx, const rp_rescore_params& R) {
  const int grid = (int)std::min<int64_t>(rp_cdiv(R.n_users, 4), (int64_t)ctx->n_cu * 8);
  hipLaunchKernelGGL((k_predict_rescore<NQ, MODE>), dim3(grid), dim3(256), (size_t)4 * 2 * 64 * NQ * 4, ctx->stream, R);
}
template <int MODE>
static void rp_launch_rescore_kc(rphip_ctx* ctx, const rp_rescore_params& R) {  // Kc <= 256
  switch (R.Kc / 64) {
    case 1: rp_launch_rescore<1, MODE>(ctx, R); break;
    case 2: rp_launch_rescore<2, MODE>(ctx, R); break;
    case 3: rp_launch_rescore<3, MODE>(ctx, R); break;
    default: rp_launch_rescore<4, MODE>(ctx, R); break;
  }
}

// The first pass with one task per user (rp_predict_i8.hip) on the int8 rows of `mm`: approximate top-Kc of users
// [user_begin, user_begin + n) into a_idx / a_val [n][Kc] and a_nnz [n] (-1: the user's list overflowed - he is appended to `fail`
// by his local id).  n_fail: [0] the fail list's count, [1] the pass's ticket (both zeroed by the caller).  user_list: the pass serves
// the n users user_begin + user_list[k] instead, row k of the outputs is the k-th of them (the second list of the unproven users).
static int rp_first_pass_i8(rphip_ctx* ctx, const rphip_model* mm, const rphip_csr* Xpred, const rp_pred_params& P, int64_t user_begin, int64_t n,
                            int Kc, int exclude_history, int32_t* a_idx, float* a_val, int32_t* a_nnz, int32_t* fail, int32_t* n_fail,
                            bool leave_room, const uint8_t* keep_dev = nullptr, const int32_t* user_list = nullptr, int max_hist = 0) {
  rp_i8_params U8;
  U8.x_ptr = Xpred->x_ptr;
  U8.x_idx = Xpred->x_idx;
  U8.B8 = reinterpret_cast<const int8_t*>(mm->B16);
  U8.ld = mm->ld;
  U8.b8_scale = mm->b8_scale;
  U8.n_items = P.n_items;
  RP_TRY(rp_predict_user_i8_geometry(P.n_items, &U8.n_tiles, &U8.tile_w));
  U8.col_begin = P.col_begin;
  U8.cyc_nb = P.cyc_nb;
  U8.cyc_w = P.cyc_w;
  U8.cyc_r = P.cyc_r;
  U8.exclude_history = exclude_history;
  U8.keep = keep_dev;
  U8.user_begin = (int)user_begin;
  U8.n_users = (int)n;
  U8.user_list = user_list;
  U8.max_hist = max_hist;
  U8.Kc = Kc;
  U8.out_idx = a_idx;
  U8.out_val = a_val;
  U8.out_nnz = a_nnz;
  U8.fail_list = fail;
  U8.n_fail = n_fail;
  U8.fail_base = 0;
  U8.ticket = nullptr;
  return rp_predict_user_i8(ctx, U8, n_fail + 1, leave_room);
}

// ---- (round 6) the fused top-Kp for ANY Kp ("lift Kp > 2048 the same way") ---------------------------------
// Beyond RP_K_MAX the selection's LDS buffers do not hold a user's list.  Then the scores of a chunk of users are formed in full - the
// kernels of predict() itself (modes 1 and 2 of k_predict_tile: count, then fill a CSR of every non-zero score, columns ascending,
// float64 sums) - and every user's row is cut to its Kp best on the spot: a workgroup per user drops the history items and the filtered
// columns in place, then runs the selection of rp_merge_big.h over the row (one list; ties to the smaller column; output in column
// order).  Chunks of users bound the CSR to ~4 GB.  The values are predict()'s float64 sums rounded to float32 - not the fixed-point sums
// of the fused kernels: the two routes may differ in the last bit of a score, and in which of two columns wins a tie that only the
// rounding creates.  Dense models on all columns and neighbour-list models; a column slab keeps RPHIP_ELIMIT.
__device__ __forceinline__ uint64_t rp_pred_ord64(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ void k_ll_to_i32(const long long* __restrict__ in, int32_t* __restrict__ out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (int32_t)in[i];
}
__global__ void k_i32_to_ll(const int32_t* __restrict__ in, long long* __restrict__ out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (long long)in[i];
}

__global__ void __launch_bounds__(RP_MBIG_THREADS) k_predict_topk_big(const int32_t* __restrict__ x_ptr, const int32_t* __restrict__ x_idx, int sorted_rows,
                                                                     int user_begin, int n_users, const long long* __restrict__ indptr,
                                                                     int32_t* __restrict__ full_idx, double* __restrict__ full_val, int Kp,
                                                                     int exclude_history, const uint8_t* __restrict__ keep, int32_t* __restrict__ out_idx,
                                                                     float* __restrict__ out_val, int32_t* __restrict__ out_nnz) {
  __shared__ rp_mbig_smem S;
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (int ul = blockIdx.x; ul < n_users; ul += gridDim.x) {
    const int u = user_begin + ul;
    const long long r0 = indptr[ul];
    const int nr = (int)(indptr[ul + 1] - r0);
    int32_t* ri = full_idx + r0;
    double* rv = full_val + r0;
    const int hs = x_ptr[u], he = x_ptr[u + 1];
    // ---- in place: keep the entries that may be recommended (the write position never passes the read position)
    int kept = 0;
    for (int base = 0; base < nr; base += RP_MBIG_THREADS) {
      const int e = base + tid;
      int c = -1;
      double v = 0.0;
      bool ok = false;
      if (e < nr) {
        c = ri[e];
        v = rv[e];
        ok = v != 0.0 && (!keep || keep[c]);
        if (ok && exclude_history) {
          if (sorted_rows) {
            int lo = hs, hi = he;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (x_idx[mid] < c) lo = mid + 1; else hi = mid;
            }
            ok = !(lo < he && x_idx[lo] == c);
          } else {
            for (int k = hs; k < he && ok; ++k) ok = x_idx[k] != c;
          }
        }
      }
      const uint64_t m = __ballot(ok);
      if (lane == 0) S.w[0][wave] = __popcll(m);
      __syncthreads();  // (everybody has read its entry of this round)
      int pos = kept + __popcll(m & lt_mask), tot = 0;
      for (int q = 0; q < RP_MBIG_THREADS / 64; ++q) {
        if (q < wave) pos += S.w[0][q];
        tot += S.w[0][q];
      }
      if (ok) {
        ri[pos] = c;
        rv[pos] = v;
      }
      kept += tot;
      __syncthreads();
    }
    if (tid == 0) s_n = kept;
    __syncthreads();
    const int n_valid = s_n;
    const int n_out = rp_merge_big_row(
        S, 1, 0, Kp, [&](int) { return n_valid; }, [&](int a) { return rp_pred_ord64(rv[a]); },
        [&](int pos, int a) {
          out_idx[(long long)ul * Kp + pos] = ri[a];
          out_val[(long long)ul * Kp + pos] = (float)rv[a];
        });
    if (tid == 0) out_nnz[ul] = n_out;
    __syncthreads();
  }
}

static int rp_predict_topk_big(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin, int64_t user_end, int32_t Kp,
                               int32_t exclude_history, const uint8_t* keep_dev, int32_t* idx_dev, float* val_dev, int32_t* nnz_dev) {
  if (model->kind == RPHIP_MODEL_DENSE && (model->n_cols > 0 || model->cyc_nb))
    RP_FAIL(RPHIP_ELIMIT, "Kp=%d beyond %d is served on models that hold all columns (not a column slab)", Kp, RP_K_MAX);
  hipStream_t st = ctx->stream;
  const int64_t n = user_end - user_begin, I = Xpred->n_items;
  RP_TRY(rp_model_gather_wait(ctx, model, true));
  // users per chunk: a row of the full result has at most I entries of 12 bytes
  const int64_t UCH = std::max<int64_t>(1, std::min<int64_t>(n, ((int64_t)4 << 30) / (I * 12)));
  char* sb = nullptr;
  RP_TRY(rp_scratch(ctx, 14, (size_t)UCH * I * 12 + (size_t)(UCH + 1) * 24 + 256, (void**)&sb));
  double* fv = (double*)sb;
  long long* cnt64 = (long long*)(sb + (size_t)UCH * I * 8);
  long long* ptr64 = cnt64 + (UCH + 1);
  int32_t* cnt32 = (int32_t*)(ptr64 + (UCH + 1));
  int32_t* ptr32 = cnt32 + (UCH + 1);
  int32_t* fi = (int32_t*)(sb + (size_t)UCH * I * 8 + (size_t)(UCH + 1) * 24);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  for (int64_t c0 = 0; c0 < n; c0 += UCH) {
    const int64_t nch = std::min<int64_t>(UCH, n - c0);
    rp_pred_params P;
    rp_pred_fill(P, model, Xpred, user_begin + c0, user_begin + c0 + nch);
    const int grid = (int)std::min<int64_t>(nch, (int64_t)ctx->n_cu * 8);
    P.mode = 1;
    P.row_nnz = cnt64;
    RP_TRY(rp_pred_launch(ctx, P, grid));
    const unsigned cg = (unsigned)std::min<int64_t>(1024, rp_cdiv(nch + 1, 256));
    hipLaunchKernelGGL(k_ll_to_i32, dim3(cg), dim3(256), 0, st, cnt64, cnt32, (int)nch);
    RP_TRY(rp_exclusive_scan_i32(ctx, cnt32, ptr32, nch));
    hipLaunchKernelGGL(k_i32_to_ll, dim3(cg), dim3(256), 0, st, ptr32, ptr64, (int)nch + 1);
    P.mode = 2;
    P.row_nnz = nullptr;
    P.indptr = ptr64;
    P.full_idx = fi;
    P.full_val = fv;
    RP_TRY(rp_pred_launch(ctx, P, grid));
    hipLaunchKernelGGL(k_predict_topk_big, dim3((unsigned)std::min<int64_t>(nch, (int64_t)ctx->n_cu * 8)), dim3(RP_MBIG_THREADS), 0, st, Xpred->x_ptr,
                       Xpred->x_idx, Xpred->sorted_rows ? 1 : 0, (int)(user_begin + c0), (int)nch, ptr64, fi, fv, Kp, exclude_history, keep_dev,
                       idx_dev + c0 * Kp, val_dev + c0 * Kp, nnz_dev + c0);
    RP_HIP(hipGetLastError());
  }
  rp_time_end(ctx, ev, model->kind == RPHIP_MODEL_DENSE ? RPHIP_T_PREDICT_DENSE : RPHIP_T_PREDICT, rp_pred_work(Xpred, model, n, Kp));
  return RPHIP_OK;
}

static int rp_predict_topk_impl(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                int64_t user_end, int32_t Kp, int32_t exclude_history, const uint8_t* keep_dev, int32_t* idx_dev,
                                float* val_dev, int32_t* nnz_dev) {
  RP_TRY(rp_pred_check(ctx, model, Xpred, user_begin, user_end));
  if (!idx_dev || !val_dev || !nnz_dev) RP_FAIL(RPHIP_EINVAL, "NULL output");
  if (Kp < 1) RP_FAIL(RPHIP_EINVAL, "Kp=%d must be at least 1", Kp);
  RP_TRY(rp_set_device(ctx));
  const int64_t n = user_end - user_begin;
  if (n == 0) return RPHIP_OK;
  hipStream_t st = ctx->stream;
  ctx->predict_stats[0] = ctx->predict_stats[1] = -1;
  ctx->predict_stats[2] = ctx->predict_stats[3] = ctx->predict_stats[4] = ctx->predict_stats[5] = 0;
  hipMemsetAsync(idx_dev, 0xFF, (size_t)n * Kp * 4, st);
  hipMemsetAsync(val_dev, 0, (size_t)n * Kp * 4, st);
  hipMemsetAsync(nnz_dev, 0, (size_t)n * 4, st);
  if (Kp > RP_K_MAX)  // (round 6) any Kp: full scores per chunk of users, cut to Kp on the spot
    return rp_predict_topk_big(ctx, model, Xpred, user_begin, user_end, Kp, exclude_history, keep_dev, idx_dev, val_dev, nnz_dev);
  rp_pred_params P;
  rp_pred_fill(P, model, Xpred, user_begin, user_end);
  P.Kp = Kp;
  P.exclude_history = exclude_history;
  P.mode = 0;
  P.keep = keep_dev;
  const int T = P.n_tiles;
  // neighbour-list models with column-sorted rows: wave-per-user kernel (any history length); otherwise (dense
  // models, unsorted lists, very large Kp) the workgroup kernel below
  const bool use_wave = model->kind == RPHIP_MODEL_TOPK && model->cols_sorted && Kp <= PW_KP_MAX && model->K < 65536 &&
                        (int64_t)model->n_items * model->K < (1ll << 30);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  if (use_wave) {
    RP_TRY(rp_model_gather_wait(ctx, model, true));
    const rphip_model* mm = model;  // (its tile offsets are built on first use: mutable members, under the context's lock)
    // 32-bit per-user-scaled sums on 4096-column tiles unless the caller asked for the exact 64-bit sums or the prediction
    // matrix carries values (their products are formed in float64)
    const bool x32 = !Xpred->x_val && ctx->predict_exact64 == 0;
    const int wt = x32 ? ctx->pw_wt32 : PW_WT;
    const int n_wt = (int)rp_cdiv(model->n_items, wt);
    if (!mm->toff || mm->toff_wt != wt) {
      if (mm->toff) {
        RP_HIP(hipStreamSynchronize(st));
        rp_dev_free(ctx, mm->toff);
        rp_dev_free(ctx, mm->rowmax);
        mm->toff = nullptr;
        mm->rowmax = nullptr;
      }
      RP_TRY(rp_dev_alloc(ctx, (size_t)model->n_items * (n_wt + 1) * sizeof(uint16_t), (void**)&mm->toff));
      RP_TRY(rp_dev_alloc(ctx, (size_t)model->n_items * sizeof(float), (void**)&mm->rowmax));
      mm->toff_wt = wt;
      hipLaunchKernelGGL(k_build_toff, dim3((unsigned)std::min<int64_t>(4096, rp_cdiv(model->n_items, 256))), dim3(256), 0, st, model->idx,
                         model->val32, model->nnz, model->K, (int)model->n_items, n_wt, wt, mm->toff, mm->rowmax);
    }
    int32_t* tk = nullptr;  // [0] ticket, [1] number of long-history users, [16..] their ids
    RP_TRY(rp_scratch(ctx, 4, 64 + (size_t)n * 4, (void**)&tk));
    hipMemsetAsync(tk, 0, 64, st);
    hipLaunchKernelGGL(k_pw_heavy_users, dim3((unsigned)std::min<int64_t>(2048, rp_cdiv(n, 256))), dim3(256), 0, st, Xpred->x_ptr,
                       (int)user_begin, (int)n, tk + 16, tk + 1);
    rp_pw_params W;
    W.heavy = tk + 16;
    W.n_heavy = tk + 1;
    W.x_ptr = Xpred->x_ptr;
    W.x_idx = Xpred->x_idx;
    W.pairs = model->pairs;
    W.toff = mm->toff;
    W.rowmax = mm->rowmax;
    W.keep = keep_dev;
    W.n_items = (int)Xpred->n_items;
    W.n_wt = n_wt;
    W.K = model->K;
    W.Kp = Kp;
    W.exclude_history = exclude_history;
    W.user_begin = (int)user_begin;
    W.n_users = (int)n;
    W.fix = P.fix;
    W.ticket = tk;
    W.out_idx = idx_dev;
    W.out_val = val_dev;
    W.out_nnz = nnz_dev;
    // waves per CU: 12 for the 2048-column tiles of the 32-bit sums (the kernel's registers allow three waves per SIMD), 8 else
    const int acc_b = x32 ? 4 : 8;
    int wpc = (x32 && wt == PW_WT) ? 12 : 8;
    while (wpc > 1 && (160 * 1024) / wpc < (long long)(rp_pw_fixed_bytes(acc_b, wt) + (size_t)(((Kp + 63) / 64) * 64 + 64) * (acc_b + 4))) --wpc;
    W.cap = rp_pw_cap_for(Kp, acc_b, wt, wpc);
    size_t wl = rp_pw_fixed_bytes(acc_b, wt) + (size_t)W.cap * (acc_b + 4);
    if (const char* e = rp_diag_env("RP_PW_WPC")) {  // (diagnostic build, tools/k8_wpc.py: fewer waves per CU by asking for more LDS than a wave needs)
      wpc = std::max(1, std::min(wpc, atoi(e)));
      wl = std::max<size_t>(wl, (size_t)(160 * 1024) / wpc - 512);
    }
    W.x_val = Xpred->x_val;
    const int wg = (int)std::min<int64_t>(rp_cdiv(n, PW_UCH), (int64_t)ctx->n_cu * wpc);
    auto launch = [&](auto kern) {
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl);
      hipLaunchKernelGGL(kern, dim3(wg), dim3(64), wl, st, W);
    };
    if (W.x_val)
      launch(k_predict_wave<uint64_t, PW_WT, true>);
    else if (!x32)
      launch(k_predict_wave<uint64_t, PW_WT, false>);
    else if (wt == 2 * PW_WT)
      launch(k_predict_wave<uint32_t, 2 * PW_WT, false>);
    else
      launch(k_predict_wave<uint32_t, PW_WT, false>);
    RP_HIP(hipGetLastError());
    rp_time_end(ctx, ev, model->kind == RPHIP_MODEL_DENSE ? RPHIP_T_PREDICT_DENSE : RPHIP_T_PREDICT, rp_pred_work(Xpred, model, n, Kp));
    return RPHIP_OK;
  }
  // ---- dense models: the two-pass K9 (see k_predict_rescore) unless switched off, the fp16 copy does not fit, or the candidate
  // list would be most of the slab anyway
  // Candidates per user.  fp16 rows: Kp + 24 .. (Kp = 100: 128 - measured to prove 99.9 % of the users at C3 / C4, tools/k9_half_gate.py;
  // 192: the same share, +8 % time).  int8 rows (round 4): the bound is the sum of the rows' half-steps, 2^-8 of the rows' largest
  // entries instead of 2^-12: at Kp = 100 a list of 192 proves 98.5 % of the users, 256 99.8 % (tools/k9_fp8_gate.py, C3 and C4; the
  // median user needs 133)
  // the first pass on narrower tiles, two workgroups per CU (k_predict_tile's HC8 = 2; the int8 rows know no other form)
  int Th = T, Wh = P.tile_w;
  bool fits2 = false;
  if (ctx->predict_half_w > 0 && model->kind == RPHIP_MODEL_DENSE) {
    const int th = (int)rp_cdiv(P.n_items, ctx->predict_half_w);
    const int wh = (int)(rp_cdiv(rp_cdiv(P.n_items, th), 64) * 64);
    if (th <= 15 && wh <= 16384 && (size_t)wh * 4 + rp_select_smem_bytes(wh) <= (size_t)78 * 1024) {
      Th = th;
      Wh = wh;
      fits2 = true;
    }
  }
  // (a borrowed slab whose leading dimension is a multiple of 8 but not of 16 cannot be read as int8 rows: it keeps the fp16 copy)
  int kc_i8 = ((Kp + 92 + 63) / 64) * 64;
  if (const char* e = rp_diag_env("RP_K9_I8_KC")) kc_i8 = std::max(64, atoi(e) / 64 * 64);
  // (round 6) the per-user first pass (rp_predict_i8.hip) has its own tiling: it does not need the narrow tiles of the (user, tile) pass
  // to exist - beyond 15 x predict_half_w items (168 000) `fits2` is false, and a 200 000-item model took the one-pass f32 kernel for
  // every user (27 s at 5M x 200K where the int8 pass takes a third)
  const bool user_pass_certain = model->kind == RPHIP_MODEL_DENSE && ctx->predict_half && ctx->predict_user_tasks &&
                                 (!exclude_history || Xpred->sorted_rows) && kc_i8 <= 256 && (int64_t)4 * kc_i8 <= (int64_t)P.n_items;
  const bool want_i8 = ctx->predict_int8 && (fits2 || user_pass_certain) && (model->ld % 16) == 0;
  const int Kc = want_i8 && kc_i8 <= 512 && (int64_t)4 * kc_i8 <= (int64_t)P.n_items ? kc_i8 : ((Kp + 24 + 63) / 64) * 64;
  const bool two_pass = model->kind == RPHIP_MODEL_DENSE && ctx->predict_half && Kc <= 512 && (int64_t)4 * Kc <= (int64_t)P.n_items && T <= 15;
  // ONE predicate decides both who enters the per-user first pass and who may leave a gathered model's f32 rows in flight
  // (every other route - the (user, tile) first pass, its rescoring, the one-pass kernel, a rebuild of the
  // quantised copy from B - reads the f32 rows at once and waits for the gather here)
  const bool per_user_first_pass = two_pass && want_i8 && Kc == kc_i8 && model->B16 && model->b16_kind == 2 && !model->b16_failed &&
                                   ctx->predict_user_tasks && Kc <= 256 && (!exclude_history || Xpred->sorted_rows);
  if (!per_user_first_pass) RP_TRY(rp_model_gather_wait(ctx, model, true));
  if (two_pass) {
    const rphip_model* mm = model;  // (the quantised copy is built on first use: mutable members, under the context's lock)
    RP_TRY(rp_model_half(ctx, mm, want_i8 && Kc == kc_i8 ? 2 : 1));
    if (mm->B16) {
      const bool i8 = mm->b16_kind == 2;
      if (i8 && ctx->predict_user_tasks && Kc <= 256 && (!exclude_history || Xpred->sorted_rows)) {
        // ---- (round 5) first pass with ONE TASK PER USER (rp_predict_i8.hip): the sums of a column tile stay in registers, a running
        // threshold hands candidates to one list per user in LDS, the workgroup cuts it to Kc entries itself - no per-tile lists in
        // HBM, no merge kernel, one launch for all users of the call - then the rescoring and the fallback as before
        const size_t szA = (size_t)n * Kc;
        const int64_t FCH = T == 1 ? n : std::min<int64_t>(n, 16384);  // users per launch of the fallback: bounds its per-tile lists
        const size_t szF = T > 1 ? (size_t)FCH * T * Kp : 0;
        char* sb = nullptr;
        RP_TRY(rp_scratch(ctx, 14, (szA + szF) * 8 + (size_t)FCH * T * 4 + (size_t)n * 12 + 256, (void**)&sb));
        int32_t* a_idx = (int32_t*)sb;
        float* a_val = (float*)(sb + szA * 4);
        int32_t* t_idx = (int32_t*)(sb + szA * 8);
        float* t_val = (float*)(sb + szA * 8 + szF * 4);
        int32_t* t_nnz = (int32_t*)(sb + (szA + szF) * 8);
        int32_t* a_nnz = t_nnz + FCH * T;
        int32_t* fail = a_nnz + n;
        int32_t* n_fail = fail + n;  // [0] count, [1] the first pass's ticket, [2] the count after the first pass (the overflowed lists);
        int32_t* n_fail2 = n_fail + 4;  // the same two words of the second list's passes
        int32_t* fail2 = n_fail + 16;   // [n] the users the exact pass serves when a second list was tried
        hipMemsetAsync(n_fail, 0, 64, st);
        RP_TRY(rp_first_pass_i8(ctx, mm, Xpred, P, user_begin, n, Kc, exclude_history, a_idx, a_val, a_nnz, fail, n_fail, model->gather_ev != nullptr, keep_dev));
        RP_HIP(hipMemcpyAsync(n_fail + 2, n_fail, 4, hipMemcpyDeviceToDevice, st));
        RP_TRY(rp_model_gather_wait(ctx, model, true));  // (a gathered model: the f32 rows the rescoring reads arrive beside the first pass)
        rp_rescore_params R{};
        R.x_ptr = Xpred->x_ptr;
        R.x_idx = Xpred->x_idx;
        R.B = model->B;
        R.ld = model->ld;
        R.col_begin = P.col_begin;
        R.cyc_nb = P.cyc_nb;
        R.cyc_w = P.cyc_w;
        R.cyc_r = P.cyc_r;
        R.delta = mm->b16_delta;
        R.rowmax = mm->b16_rowmax;
        R.cand_idx = a_idx;
        R.cand_val = a_val;
        R.cand_nnz = a_nnz;
        R.user_begin = (int)user_begin;
        R.n_users = (int)n;
        R.Kp = Kp;
        R.Kc = Kc;
        R.out_idx = idx_dev;
        R.out_val = val_dev;
        R.out_nnz = nnz_dev;
        R.fail_list = fail;
        R.n_fail = n_fail;
        R.fail_base = 0;
        switch (Kc / 64) {
          case 2: rp_launch_rescore<2>(ctx, R); break;
          case 3: rp_launch_rescore<3>(ctx, R); break;
          default: rp_launch_rescore<4>(ctx, R); break;
        }
        RP_HIP(hipGetLastError());
        // the users without a proof (and those whose list overflowed): the one-pass kernel, as few launches as their lists allow
        const size_t lds1 = rp_pred_lds(P);
        RP_HIP(hipFuncSetAttribute((const void*)k_predict_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        const size_t mlds_p = (size_t)T * Kp * 8 + rp_select_smem_bytes(T * Kp);
        if (T > 1) {
          if (mlds_p > 160 * 1024 || T * Kp >= RPS_FLAG) RP_FAIL(RPHIP_ELIMIT, "Kp=%d with %d column tiles needs %zu bytes of LDS for the merge", Kp, T, mlds_p);
          RP_HIP(hipFuncSetAttribute((const void*)k_predict_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds_p));
        }
        int h3[3] = {0, 0, 0};
        RP_HIP(hipMemcpyAsync(h3, n_fail, 12, hipMemcpyDeviceToHost, st));
        RP_HIP(hipStreamSynchronize(st));
        int h_fail = h3[0];
        const int n_over = h3[2];
        ctx->predict_stats[1] = n;
        ctx->predict_stats[2] = Kc;
        ctx->predict_stats[3] = 1;
        ctx->predict_stats[4] = ctx->predict_stats[5] = 0;
        // ---- (round 6) a SECOND LIST for the users without a proof: the same two passes on them alone with the longest list the pass
        // holds (256).  A proof fails when the Kp-th exact score does not clear the list's end by the user's error bound - long
        // histories, history removal and item filters (the best columns gone) make that 1.5 - 3 % of the users, and the exact pass
        // reads four times the bytes per row for them.  The users whose list overflowed gain nothing from it: they go straight on.
        const int Kc2 = RPHIP_KC2;
        const int64_t n2 = (int64_t)h_fail - n_over;
        if (ctx->predict_second_list && Kc < Kc2 && n2 > 0 && (size_t)n2 * Kc2 <= szA && (int64_t)4 * Kc2 <= (int64_t)P.n_items) {
          int32_t* b_idx = (int32_t*)sb;  // (the first lists are done with: their place)
          float* b_val = (float*)(sb + (size_t)n2 * Kc2 * 4);
          const int32_t* list2 = fail + n_over;
          if (n_over > 0) RP_HIP(hipMemcpyAsync(fail2, fail, (size_t)n_over * 4, hipMemcpyDeviceToDevice, st));
          RP_HIP(hipMemcpyAsync(n_fail2, n_fail + 2, 4, hipMemcpyDeviceToDevice, st));
          RP_TRY(rp_first_pass_i8(ctx, mm, Xpred, P, user_begin, n2, Kc2, exclude_history, b_idx, b_val, a_nnz, fail2, n_fail2, false, keep_dev, list2, ctx->predict_second_list_hist));
          rp_rescore_params R2 = R;
          R2.cand_idx = b_idx;
          R2.cand_val = b_val;
          R2.n_users = (int)n2;
          R2.Kc = Kc2;
          R2.user_list = list2;
          R2.fail_list = fail2;
          R2.n_fail = n_fail2;
          rp_launch_rescore<RPHIP_KC2 / 64>(ctx, R2);
          RP_HIP(hipGetLastError());
          int h2 = 0;
          RP_HIP(hipMemcpyAsync(&h2, n_fail2, 4, hipMemcpyDeviceToHost, st));
          RP_HIP(hipStreamSynchronize(st));
          ctx->predict_stats[4] = n2;
          ctx->predict_stats[5] = (int64_t)h_fail - h2;  // (proven by the second list)
          h_fail = h2;
          fail = fail2;
        }
        ctx->predict_stats[0] = h_fail;
        for (int64_t f0 = 0; f0 < h_fail; f0 += FCH) {
          const int64_t nf = std::min<int64_t>(FCH, h_fail - f0);
          const int grid = (int)std::min<int64_t>(nf * T, (int64_t)ctx->n_cu * 8), mgrid = (int)std::min<int64_t>(nf, (int64_t)ctx->n_cu * 8);
          rp_pred_params F = P;
          F.user_begin = (int)user_begin;
          F.n_users = (int)nf;
          F.user_list = fail + f0;
          F.n_list = nullptr;
          F.list_compact = 1;
          if (T > 1) hipMemsetAsync(t_nnz, 0, (size_t)nf * T * 4, st);
          F.out_idx = T > 1 ? t_idx : idx_dev;
          F.out_val = T > 1 ? t_val : val_dev;
          F.out_nnz = T > 1 ? t_nnz : nnz_dev;
          hipLaunchKernelGGL(k_predict_tile<true>, dim3(grid), dim3(RP_PRED_THREADS), lds1, st, F);
          if (T > 1)
            hipLaunchKernelGGL(k_predict_merge, dim3(mgrid), dim3(RP_PRED_THREADS), mlds_p, st, t_idx, t_val, t_nnz, T, Kp, (int)nf, fail + f0, nullptr,
                               idx_dev, val_dev, nnz_dev, 0, 0, 1);
          RP_HIP(hipGetLastError());
        }
        rp_time_end(ctx, ev, RPHIP_T_PREDICT_DENSE, rp_pred_work(Xpred, model, n, Kp));
        return RPHIP_OK;
      }
      const bool narrow = fits2 && (i8 || Th != T || Wh != P.tile_w);
      if (!fits2) {
        Th = T;
        Wh = P.tile_w;
      }
      const int Tm = std::max(T, Th);
      // first-pass tiles hand on every column above their bound (k_predict_tile, tile_cap): lists of up to 4 Kc entries
      int cap_t = (narrow && Th > 1 && !keep_dev && Kc <= 256) ? 4 * Kc : 0;
      //  the threshold lists are four times as long as the exact ones: where their merge would not fit the LDS
      // (15 tiles x 4 x 256 entries = 162 KB) or their scratch would pass 6 GB the tiles select exactly, as in round 3
      if (cap_t > 0 && ((size_t)Th * cap_t * 8 + rp_select_smem_bytes(Th * cap_t) > (size_t)160 * 1024 || Th * cap_t >= RPS_FLAG ||
                        (size_t)std::min<int64_t>(n, 1 << 16) * std::max(T, Th) * cap_t * 8 > ((size_t)6 << 30)))
        cap_t = 0;
      const int stride_t = cap_t > 0 ? cap_t : Kc;
      const int64_t UCH = cap_t > 0 ? 1 << 16 : 1 << 17;  // users per chunk: bounds the candidate scratch
      const int64_t nchm = std::min<int64_t>(n, UCH);
      const size_t szT = (Tm > 1) ? (size_t)nchm * Tm * stride_t : 0, szA = (size_t)nchm * Kc;
      char* sb = nullptr;
      RP_TRY(rp_scratch(ctx, 14, (szT + szA) * 8 + (size_t)nchm * (Tm + 1) * 4 + (size_t)n * 4 + 256, (void**)&sb));
      int32_t* a_idx = (int32_t*)sb;
      float* a_val = (float*)(sb + szA * 4);
      int32_t* t_idx = (int32_t*)(sb + szA * 8);
      float* t_val = (float*)(sb + szA * 8 + szT * 4);
      int32_t* a_nnz = (int32_t*)(sb + (szA + szT) * 8);
      int32_t* t_nnz = a_nnz + nchm;
      int32_t* fail = t_nnz + nchm * Tm;  // [n]: the users of ALL chunks whose list proves nothing (ids relative to user_begin)
      int32_t* n_fail = fail + n;      // (one word behind the list)
      const size_t lds1 = rp_pred_lds(P);
      rp_pred_params Hh = P;
      Hh.n_tiles = Th;
      Hh.tile_w = Wh;
      const size_t ldsh = rp_pred_lds(Hh);
      if (i8)
        RP_HIP(hipFuncSetAttribute((const void*)k_predict_tile<true, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh));
      else if (narrow)
        RP_HIP(hipFuncSetAttribute((const void*)k_predict_tile<true, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh));
      else
        RP_HIP(hipFuncSetAttribute((const void*)k_predict_tile<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
      RP_HIP(hipFuncSetAttribute((const void*)k_predict_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
      const size_t mlds_c = (size_t)Th * stride_t * 8 + rp_select_smem_bytes(Th * stride_t), mlds_p = (size_t)T * Kp * 8 + rp_select_smem_bytes(T * Kp);
      if (Tm > 1) {
        if (std::max(mlds_c, mlds_p) > 160 * 1024 || Tm * stride_t >= RPS_FLAG)
          RP_FAIL(RPHIP_ELIMIT, "Kp=%d with %d column tiles needs %zu bytes of LDS for the merge", Kp, Tm, std::max(mlds_c, mlds_p));
        RP_HIP(hipFuncSetAttribute((const void*)k_predict_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(mlds_c, mlds_p)));
      }
      hipMemsetAsync(n_fail, 0, 4, st);
      for (int64_t c0 = 0; c0 < n; c0 += UCH) {
        const int64_t nch = std::min<int64_t>(UCH, n - c0);
        const int grid = (int)std::min<int64_t>(nch * T, (int64_t)ctx->n_cu * 8), mgrid = (int)std::min<int64_t>(nch, (int64_t)ctx->n_cu * 8);
        const int gridh = (int)std::min<int64_t>(nch * Th, (int64_t)ctx->n_cu * 16);
        // first pass: approximate top-Kc from the fp16 rows
        rp_pred_params H = Hh;
        H.user_begin = (int)(user_begin + c0);
        H.n_users = (int)nch;
        H.Kp = Kc;
        H.dense_B16 = mm->B16;
        H.b16_inv = 1.0f / mm->b16_scale;
        H.dense_B8 = reinterpret_cast<const int8_t*>(mm->B16);
        H.b8_scale = mm->b8_scale;
        H.tile_cap = cap_t;
        hipMemsetAsync(a_nnz, 0, (size_t)nch * 4, st);
        if (Th > 1) hipMemsetAsync(t_nnz, 0, (size_t)nch * Th * 4, st);
        H.out_idx = Th > 1 ? t_idx : a_idx;
        H.out_val = Th > 1 ? t_val : a_val;
        H.out_nnz = Th > 1 ? t_nnz : a_nnz;
        if (i8)
          hipLaunchKernelGGL((k_predict_tile<true, true, 2, true>), dim3(gridh), dim3(RP_PRED_THREADS), ldsh, st, H);
        else if (narrow)
          hipLaunchKernelGGL((k_predict_tile<true, true, 2>), dim3(gridh), dim3(RP_PRED_THREADS), ldsh, st, H);
        else
          hipLaunchKernelGGL((k_predict_tile<true, true>), dim3(grid), dim3(RP_PRED_THREADS), lds1, st, H);
        if (Th > 1)
          hipLaunchKernelGGL(k_predict_merge, dim3(mgrid), dim3(RP_PRED_THREADS), mlds_c, st, t_idx, t_val, t_nnz, Th, Kc, (int)nch, nullptr, nullptr, a_idx,
                             a_val, a_nnz, 0, stride_t);
        // second pass: exact scores of the candidates, their top-Kp, the proof; users without one go to `fail`
        rp_rescore_params R{};
        R.x_ptr = Xpred->x_ptr;
        R.x_idx = Xpred->x_idx;
        R.B = model->B;
        R.ld = model->ld;
        R.col_begin = P.col_begin;
        R.cyc_nb = P.cyc_nb;
        R.cyc_w = P.cyc_w;
        R.cyc_r = P.cyc_r;
        R.delta = mm->b16_delta;
        R.rowmax = mm->b16_rowmax;
        R.cand_idx = a_idx;
        R.cand_val = a_val;
        R.cand_nnz = a_nnz;
        R.user_begin = (int)(user_begin + c0);
        R.n_users = (int)nch;
        R.Kp = Kp;
        R.Kc = Kc;
        R.out_idx = idx_dev + c0 * Kp;
        R.out_val = val_dev + c0 * Kp;
        R.out_nnz = nnz_dev + c0;
        R.fail_list = fail;
        R.n_fail = n_fail;
        R.fail_base = (int)c0;
        switch (Kc / 64) {
          case 2: rp_launch_rescore<2>(ctx, R); break;
          case 3: rp_launch_rescore<3>(ctx, R); break;
          case 4: rp_launch_rescore<4>(ctx, R); break;
          case 5: rp_launch_rescore<5>(ctx, R); break;
          case 6: rp_launch_rescore<6>(ctx, R); break;
          case 7: rp_launch_rescore<7>(ctx, R); break;
          default: rp_launch_rescore<8>(ctx, R); break;
        }
        RP_HIP(hipGetLastError());
      }
      // The one-pass kernel for the users without a proof - of ALL chunks, in as few launches as possible (round 4: per chunk the
      // ~1 000 users of a 65 536-user chunk left most of the chip idle: 146 of 1 290 ms at C4).  Their count comes to the host.
      int h_fail = 0;
      RP_HIP(hipMemcpyAsync(&h_fail, n_fail, 4, hipMemcpyDeviceToHost, st));
      RP_HIP(hipStreamSynchronize(st));
      ctx->predict_stats[0] = h_fail;
      ctx->predict_stats[1] = n;
      ctx->predict_stats[2] = Kc;
      // list entries per launch: what the tile lists hold (one tile: no lists, the kernel writes the output rows themselves)
      const int64_t FCH = T == 1 ? std::max<int64_t>(1, h_fail) : std::max<int64_t>(1, std::min<int64_t>(nchm, (int64_t)(szT / ((size_t)T * Kp))));
      for (int64_t f0 = 0; f0 < h_fail; f0 += FCH) {
        const int64_t nf = std::min<int64_t>(FCH, h_fail - f0);
        const int grid = (int)std::min<int64_t>(nf * T, (int64_t)ctx->n_cu * 8), mgrid = (int)std::min<int64_t>(nf, (int64_t)ctx->n_cu * 8);
        rp_pred_params F = P;
        F.user_begin = (int)user_begin;
        F.n_users = (int)nf;
        F.user_list = fail + f0;
        F.n_list = nullptr;
        F.list_compact = 1;
        if (T > 1) hipMemsetAsync(t_nnz, 0, (size_t)nf * T * 4, st);
        F.out_idx = T > 1 ? t_idx : idx_dev;
        F.out_val = T > 1 ? t_val : val_dev;
        F.out_nnz = T > 1 ? t_nnz : nnz_dev;
        hipLaunchKernelGGL(k_predict_tile<true>, dim3(grid), dim3(RP_PRED_THREADS), lds1, st, F);
        if (T > 1)
          hipLaunchKernelGGL(k_predict_merge, dim3(mgrid), dim3(RP_PRED_THREADS), mlds_p, st, t_idx, t_val, t_nnz, T, Kp, (int)nf, fail + f0, nullptr,
                             idx_dev, val_dev, nnz_dev, 0, 0, 1);
        RP_HIP(hipGetLastError());
      }
      rp_time_end(ctx, ev, RPHIP_T_PREDICT_DENSE, rp_pred_work(Xpred, model, n, Kp));
      return RPHIP_OK;
    }
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
  RP_TRY(rp_pred_launch(ctx, P, (int)std::min<int64_t>(n * T, (int64_t)ctx->n_cu * 8)));
  if (T > 1) {
    const size_t mlds = (size_t)T * Kp * 8 + rp_select_smem_bytes(T * Kp);
    if (mlds > 160 * 1024 || T * Kp >= RPS_FLAG)
      RP_FAIL(RPHIP_ELIMIT, "Kp=%d with %d column tiles needs %zu bytes of LDS for the merge", Kp, T, mlds);
    RP_HIP(hipFuncSetAttribute((const void*)k_predict_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
    const int mg = (int)std::min<int64_t>(n, (int64_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(k_predict_merge, dim3(mg), dim3(RP_PRED_THREADS), mlds, st, cand_idx, cand_val, cand_nnz, T, Kp, (int)n,
                       nullptr, nullptr, idx_dev, val_dev, nnz_dev, 0);
  }
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, model->kind == RPHIP_MODEL_DENSE ? RPHIP_T_PREDICT_DENSE : RPHIP_T_PREDICT, rp_pred_work(Xpred, model, n, Kp));
  return RPHIP_OK;
}

// Merge of per-user candidate lists (e.g. the local top-Kp of every rank's column slab, gathered): cand_idx / cand_val
// [n_users][n_lists][Kp], cand_nnz [n_users][n_lists] (device) -> the best Kp of each user's union by (score desc, column asc),
// columns ascending in the output row - the order of the single-GPU result.
extern "C" int rphip_predict_merge_topk(rphip_ctx* ctx, int32_t n_lists, int32_t Kp, int64_t n_users, const int32_t* cand_idx_dev,
                                        const float* cand_val_dev, const int32_t* cand_nnz_dev, int32_t* idx_dev, float* val_dev,
                                        int32_t* nnz_dev) {
  RP_ENTER(ctx);
  if (!ctx || !cand_idx_dev || !cand_val_dev || !cand_nnz_dev || !idx_dev || !val_dev || !nnz_dev) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (n_lists < 1 || Kp < 1 || Kp > RP_K_MAX || n_users < 0) RP_FAIL(RPHIP_EINVAL, "n_lists %d, Kp %d, n_users %lld", n_lists, Kp, (long long)n_users);
  // the merge keeps the lists' offsets in the selection's 32 spare LDS words (RPS_MISC_INTS - RPS_USER)
  if (n_lists > RPS_MISC_INTS - RPS_USER - 1) RP_FAIL(RPHIP_ELIMIT, "at most %d candidate lists per user (got %d)", RPS_MISC_INTS - RPS_USER - 1, n_lists);
  RP_TRY(rp_set_device(ctx));
  if (n_users == 0) return RPHIP_OK;
  const size_t mlds = (size_t)n_lists * Kp * 8 + rp_select_smem_bytes(n_lists * Kp) + (size_t)Kp * 8;
  if (mlds > 160 * 1024 || n_lists * Kp >= RPS_FLAG)
    RP_FAIL(RPHIP_ELIMIT, "Kp=%d with %d lists needs %zu bytes of LDS for the merge", Kp, n_lists, mlds);
  hipStream_t st = ctx->stream;
  hipMemsetAsync(idx_dev, 0xFF, (size_t)n_users * Kp * 4, st);
  hipMemsetAsync(val_dev, 0, (size_t)n_users * Kp * 4, st);
  RP_HIP(hipFuncSetAttribute((const void*)k_predict_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
  const int mg = (int)std::min<int64_t>(n_users, (int64_t)ctx->n_cu * 8);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  hipLaunchKernelGGL(k_predict_merge, dim3(mg), dim3(RP_PRED_THREADS), mlds, st, cand_idx_dev, cand_val_dev, cand_nnz_dev, (int)n_lists, (int)Kp,
                     (int)n_users, nullptr, nullptr, idx_dev, val_dev, nnz_dev, 1);
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, RPHIP_T_PREDICT, 16.0 * (double)n_users * n_lists * Kp);
  return RPHIP_OK;
}

// ---- predict with the model's columns spread over the ranks of a communicator --------------------------
// Every rank scores the users against ITS columns (a local top-Kp per user), the lists travel to the rank that owns the user - one
// personalised exchange per chunk, each rank receives only its own users' lists: U * Kp * 8 bytes per rank over the whole call -
// and are merged there by (score desc, column asc), the single-GPU order.  Rank r ends with the rows of users
// [r * per, (r + 1) * per), per = ceil(U / world).
// received slots [list s][user][Kp] -> the merge's [user][list][Kp]
__global__ void k_lists_to_users(const char* __restrict__ recv, long long slot_bytes, int W, int cs, int n, int Kp, int32_t* __restrict__ ci,
                                 float* __restrict__ cv, int32_t* __restrict__ cn) {
  const long long total = (long long)n * W * Kp;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(e % Kp);
    const long long uw = e / Kp;
    const int s = (int)(uw % W);
    const long long u = uw / W;
    const char* slot = recv + (long long)s * slot_bytes;
    ci[e] = reinterpret_cast<const int32_t*>(slot)[u * Kp + k];
    cv[e] = reinterpret_cast<const float*>(slot + (long long)cs * Kp * 4)[u * Kp + k];
    if (k == 0) cn[uw] = reinterpret_cast<const int32_t*>(slot + (long long)cs * Kp * 8)[u];
  }
}

extern "C" int rphip_predict_topk_dist(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* model, const rphip_csr* X, int32_t Kp,
                                       int32_t exclude_history, int64_t* user_begin_out, int64_t* n_users_out, int32_t* idx_dev, float* val_dev,
                                       int32_t* nnz_dev) {
  RP_ENTER(ctx);
  if (!ctx || !comm || !X || !idx_dev || !val_dev || !nnz_dev) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (Kp < 1 || Kp > RP_K_MAX) RP_FAIL(RPHIP_ELIMIT, "Kp=%d outside 1..%d", Kp, RP_K_MAX);
  RP_TRY(rp_set_device(ctx));
  const int W = rp_comm_world(comm), me = rp_comm_rank(comm);
  const int64_t U = X->n_users;
  const int64_t per = rp_cdiv(std::max<int64_t>(U, 1), W);
  auto share = [&](int r) { return std::max<int64_t>(0, std::min<int64_t>(U, (int64_t)(r + 1) * per) - (int64_t)r * per); };
  const int64_t mine = share(me);
  if (user_begin_out) *user_begin_out = std::min<int64_t>(U, (int64_t)me * per);
  if (n_users_out) *n_users_out = mine;
  if (W > RPS_MISC_INTS - RPS_USER - 1) RP_FAIL(RPHIP_ELIMIT, "at most %d ranks", RPS_MISC_INTS - RPS_USER - 1);
  hipStream_t st = ctx->stream;
  const int cs = (int)std::min<int64_t>(per, std::max<int64_t>(1024, ((int64_t)1 << 18) / W));  // users per destination and chunk
  const int64_t slot = (int64_t)cs * ((int64_t)Kp * 8 + 4);
  char *sendb = nullptr, *recvb = nullptr, *cand = nullptr;
  struct hold {  // released on every way out
    rphip_ctx* c;
    void* p[3] = {nullptr, nullptr, nullptr};
    ~hold() {
      hipStreamSynchronize(c->stream);
      for (void* q : p) rp_dev_free(c, q);
    }
  } H{ctx};
  RP_TRY(rp_dev_alloc(ctx, (size_t)slot * W, (void**)&sendb));
  H.p[0] = sendb;
  RP_TRY(rp_dev_alloc(ctx, (size_t)slot * W, (void**)&recvb));
  H.p[1] = recvb;
  RP_TRY(rp_dev_alloc(ctx, (size_t)slot * W, (void**)&cand));
  H.p[2] = cand;
  int32_t* ci = (int32_t*)cand;
  float* cv = (float*)(cand + (int64_t)cs * W * Kp * 4);
  int32_t* cn = (int32_t*)(cand + (int64_t)cs * W * Kp * 8);
  const size_t mlds = (size_t)W * Kp * 8 + rp_select_smem_bytes(W * Kp) + (size_t)Kp * 8;
  if (mlds > 160 * 1024 || W * Kp >= RPS_FLAG) RP_FAIL(RPHIP_ELIMIT, "Kp=%d with %d ranks needs %zu bytes of LDS for the merge", Kp, W, mlds);
  std::vector<int64_t> soff(W), sby(W), roff(W), rby(W);
  for (int64_t o = 0; o < per; o += cs) {
    const int64_t len_me = std::max<int64_t>(0, std::min<int64_t>(cs, mine - o));
    for (int d = 0; d < W; ++d) {
      const int64_t len_d = std::max<int64_t>(0, std::min<int64_t>(cs, share(d) - o));
      char* sl = sendb + (int64_t)d * slot;
      soff[d] = (int64_t)d * slot;
      sby[d] = len_d > 0 ? slot : 0;
      roff[d] = (int64_t)d * slot;
      rby[d] = len_me > 0 ? slot : 0;
      if (len_d == 0) continue;
      int32_t* li = (int32_t*)sl;
      float* lv = (float*)(sl + (int64_t)cs * Kp * 4);
      int32_t* ln = (int32_t*)(sl + (int64_t)cs * Kp * 8);
      if (model) {
        const int64_t ub = (int64_t)d * per + o;
        RP_TRY(rp_predict_topk_impl(ctx, model, X, ub, ub + len_d, Kp, exclude_history, nullptr, li, lv, ln));
      } else {  // this rank holds no columns: empty lists
        RP_HIP(hipMemsetAsync(ln, 0, (size_t)len_d * 4, st));
      }
    }
    RP_TRY(rp_comm_exchange(comm, sendb, soff.data(), sby.data(), recvb, roff.data(), rby.data(), st));
    if (len_me > 0) {
      const long long total = (long long)len_me * W * Kp;
      hipLaunchKernelGGL(k_lists_to_users, dim3((unsigned)std::min<long long>(8192, rp_cdiv(total, 256))), dim3(256), 0, st, recvb, (long long)slot, W, cs,
                         (int)len_me, (int)Kp, ci, cv, cn);
      RP_HIP(hipMemsetAsync(idx_dev + o * Kp, 0xFF, (size_t)len_me * Kp * 4, st));
      RP_HIP(hipMemsetAsync(val_dev + o * Kp, 0, (size_t)len_me * Kp * 4, st));
      RP_HIP(hipFuncSetAttribute((const void*)k_predict_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
      hipLaunchKernelGGL(k_predict_merge, dim3((unsigned)std::min<int64_t>(len_me, (int64_t)ctx->n_cu * 8)), dim3(RP_PRED_THREADS), mlds, st, ci, cv, cn, W,
                         (int)Kp, (int)len_me, nullptr, nullptr, idx_dev + o * Kp, val_dev + o * Kp, nnz_dev + o, 1);
      RP_HIP(hipGetLastError());
    }
  }
  return RPHIP_OK;
}

// ---- (round 5) predict with the exact scores formed WHERE THE COLUMNS ARE -----------------------------------------------------------
// The gathered-model route (rphip_ease_model_allgather + the single-GPU predict per rank) moves 4 n^2 bytes to every rank for a
// rescoring that reads 4 Kc nnz of them; in the 8-rank emulation at 1M x 100K the f32 strips and their unpacking took the first pass
// a third of its HBM (predict 167 ms where the pass and the rescoring alone take 103).  Here only the int8 rows travel
// (rphip_ease_model_allgather_i8: n^2 bytes):
//   1. every rank runs the first pass (rp_predict_i8.hip) for ITS users on the gathered int8 rows: Kc candidates per user;
//   2. the candidate lists are all-gathered (U Kc 8 bytes per rank: 1.5 GB at C4, 2 ms);
//   3. every rank forms the exact f32 sums of the candidates whose COLUMN it owns, for the users of all ranks, from its own panels
//      (k_predict_rescore mode 1: the one-pass kernel's sums - history order, one add per row - as on one GPU; X is replicated);
//   4. one personalised exchange returns the sums to the users' ranks (U Kc 4 bytes per rank);
//   5. ranking and proof per user (mode 2); users without a proof - of all ranks - are scored exactly by every rank on its own columns
//      (the one-pass kernel on a user list), their local top-Kp lists exchanged and merged like rphip_predict_topk_dist's.
// The results are the single-GPU predict's bit for bit (tests: gloo, two and three ranks).  user_bounds[world + 1]: rank r's users are
// [user_bounds[r], user_bounds[r + 1]) - the same array on every rank; the outputs hold this rank's users.
__global__ void k_owner_fail_concat(const int32_t* __restrict__ g, int slot, int W, const int32_t* __restrict__ off, const int64_t* __restrict__ bounds,
                                    int32_t* __restrict__ out) {
  const int r = blockIdx.y;
  const int n = g[(long long)r * slot];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[off[r] + i] = (int32_t)bounds[r] + min(g[(long long)r * slot + 1 + i], (int32_t)(bounds[r + 1] - bounds[r]) - 1);  // (the clamp only ever acts on a loop-back communicator's copies)
}

extern "C" int rphip_predict_topk_owner(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, const rphip_model* rows8, const rphip_csr* X,
                                        const int64_t* user_bounds, int32_t Kp, int32_t exclude_history, int32_t* idx_dev, float* val_dev,
                                        int32_t* nnz_dev) {
  RP_ENTER(ctx);
  if (!ctx || !comm || !rows8 || !X || !user_bounds || !idx_dev || !val_dev || !nnz_dev) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (Kp < 1 || Kp > RP_K_MAX) RP_FAIL(RPHIP_ELIMIT, "Kp=%d outside 1..%d", Kp, RP_K_MAX);
  const int W = rp_comm_world(comm), me = rp_comm_rank(comm);
  if (rows8->kind != RPHIP_MODEL_DENSE || !rows8->B16 || rows8->b16_kind != 2 || !rows8->b8_scale)
    RP_FAIL(RPHIP_ESTATE, "the second model must hold the gathered int8 rows (rphip_ease_model_allgather_i8)");
  if (rows8->n_items != X->n_items) RP_FAIL(RPHIP_EINVAL, "prediction matrix has %lld items, the model %lld", (long long)X->n_items, (long long)rows8->n_items);
  if (local && (local->kind != RPHIP_MODEL_DENSE || !local->B || local->n_items != rows8->n_items || local->n_cols <= 0))
    RP_FAIL(RPHIP_EINVAL, "the first model must be this rank's columns of the same fit (rphip_ease_fit_dist), or NULL on a rank without columns");
  if (user_bounds[0] != 0 || user_bounds[W] != X->n_users) RP_FAIL(RPHIP_EINVAL, "user_bounds must run from 0 to the matrix's %lld users", (long long)X->n_users);
  for (int r = 0; r < W; ++r)
    if (user_bounds[r + 1] < user_bounds[r]) RP_FAIL(RPHIP_EINVAL, "user_bounds must not decrease");
  const int Kc = ((Kp + 92 + 63) / 64) * 64;
  if (Kc > 256 || (int64_t)4 * Kc > X->n_items || (exclude_history && !X->sorted_rows) || W > RPS_MISC_INTS - RPS_USER - 1)
    RP_FAIL(RPHIP_ELIMIT, "Kp=%d, %lld items, %d ranks%s: outside what this route serves (use rphip_predict_topk_dist)", Kp, (long long)X->n_items, W,
            (exclude_history && !X->sorted_rows) ? ", history removal on unsorted rows" : "");
  RP_TRY(rp_set_device(ctx));
  hipStream_t st = ctx->stream;
  const int64_t U = X->n_users, ub = user_bounds[me], n_me = user_bounds[me + 1] - ub;
  int64_t n_max = 0;
  for (int r = 0; r < W; ++r) n_max = std::max(n_max, user_bounds[r + 1] - user_bounds[r]);
  if (n_max == 0) return RPHIP_OK;
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  if (n_me > 0) {
    hipMemsetAsync(idx_dev, 0xFF, (size_t)n_me * Kp * 4, st);
    hipMemsetAsync(val_dev, 0, (size_t)n_me * Kp * 4, st);
    hipMemsetAsync(nnz_dev, 0, (size_t)n_me * 4, st);
  }
  struct hold {  // released on every way out
    rphip_ctx* c;
    std::vector<void*> p;
    ~hold() {
      hipStreamSynchronize(c->stream);
      for (void* q : p) rp_dev_free(c, q);
    }
    int get(size_t bytes, void** out) {
      RP_TRY(rp_dev_alloc(c, std::max<size_t>(bytes, 256), out));
      p.push_back(*out);
      return RPHIP_OK;
    }
  } H{ctx, {}};
  // candidates of all ranks: [W][n_max][Kc] columns and approximate scores, [W][n_max] counts
  const size_t slotA = (size_t)n_max * Kc;
  int32_t *A_idx = nullptr, *A_nnz = nullptr, *my_idx = nullptr, *my_nnz = nullptr, *fail = nullptr, *n_fail = nullptr;
  float *A_val = nullptr, *my_val = nullptr, *part_s = nullptr, *part_r = nullptr;
  RP_TRY(H.get(64, (void**)&n_fail));  // [0] count of users without a proof, [1] the first pass's ticket, [8] the ranks' agreement
  int arc = RPHIP_OK;
  auto get = [&](size_t bytes, void** o) {
    if (arc == RPHIP_OK) arc = H.get(bytes, o);
  };
  get(slotA * W * 4, (void**)&A_idx);
  get(slotA * W * 4, (void**)&A_val);
  get((size_t)n_max * W * 4, (void**)&A_nnz);
  get(slotA * 4, (void**)&my_idx);
  get(slotA * 4, (void**)&my_val);
  get((size_t)n_max * 4, (void**)&my_nnz);
  get((size_t)(n_max + 1) * 4 + 64, (void**)&fail);  // [0] count, [1 ..] local ids (the all-gather's slot)
  get((size_t)std::max<int64_t>(U, 1) * Kc * 4, (void**)&part_s);
  get((size_t)std::max<int64_t>(n_me, 1) * Kc * 4 * W, (void**)&part_r);
  const int fslot = (int)n_max + 1;
  int32_t* G = nullptr;  // [W][fslot]: every rank's users without a proof
  get((size_t)fslot * W * 4, (void**)&G);
  // No rank may leave alone once the collectives have begun (the others would wait in the next one: ADVICE round 5).  Everything a
  // single rank can trip over is therefore decided HERE: its allocations, and the limits that depend on its own columns (the exact
  // pass's tiles and the two merges' LDS).  The ranks sum two 0 / 1 flags and return RPHIP_ENOMEM / RPHIP_ELIMIT together - the
  // caller can then take rphip_predict_topk_dist.  agree() is repeated after the second group of allocations below.
  rp_pred_params PL;
  if (local) rp_pred_fill(PL, local, X, 0, U);
  const int T_l = local ? PL.n_tiles : 1;
  const size_t mlds_t = (size_t)T_l * Kp * 8 + rp_select_smem_bytes(T_l * Kp) + (size_t)Kp * 8;
  const size_t mlds_w = (size_t)W * Kp * 8 + rp_select_smem_bytes(W * Kp) + (size_t)Kp * 8;
  const bool lim_bad = T_l > 15 || (T_l > 1 && (mlds_t > 160 * 1024 || T_l * Kp >= RPS_FLAG)) || mlds_w > 160 * 1024 || W * Kp >= RPS_FLAG;
  auto agree = [&](bool mem_bad, bool limit_bad, int32_t* sums) -> int {
    int32_t h[2] = {mem_bad ? 1 : 0, limit_bad ? 1 : 0};
    RP_HIP(hipMemcpyAsync(n_fail + 8, h, 8, hipMemcpyHostToDevice, st));
    RP_HIP(hipStreamSynchronize(st));
    RP_TRY(rp_comm_all_reduce_sum_i32(comm, n_fail + 8, 2, st));
    RP_HIP(hipMemcpyAsync(sums, n_fail + 8, 8, hipMemcpyDeviceToHost, st));
    RP_HIP(hipStreamSynchronize(st));
    return RPHIP_OK;
  };
  {
    int32_t h_sum[2] = {1, 1};
    RP_TRY(agree(arc != RPHIP_OK, lim_bad, h_sum));
    if (h_sum[1] != 0)
      RP_FAIL(RPHIP_ELIMIT, "%d of %d ranks: Kp=%d with %d column tiles here and %d ranks is outside what the exact pass and its merges serve: use rphip_predict_topk_dist",
              (int)h_sum[1], W, Kp, T_l, W);
    if (h_sum[0] != 0)
      RP_FAIL(RPHIP_ENOMEM, "%d of %d ranks could not allocate the candidate and part buffers (%zu bytes here): use rphip_predict_topk_dist", (int)h_sum[0], W,
              slotA * W * 8 + (size_t)U * Kc * 4 + (size_t)n_me * Kc * 4 * W);
  }
  hipMemsetAsync(n_fail, 0, 64, st);
  hipMemsetAsync(my_nnz, 0, (size_t)n_max * 4, st);
  hipMemsetAsync(my_idx, 0xFF, slotA * 4, st);
  hipMemsetAsync(my_val, 0, slotA * 4, st);
  rp_pred_params P8;  // the gathered rows: every column, no layout
  rp_pred_fill(P8, rows8, X, ub, ub + n_me);
  // ---- 1. first pass for this rank's users
  if (n_me > 0) RP_TRY(rp_first_pass_i8(ctx, rows8, X, P8, ub, n_me, Kc, exclude_history, my_idx, my_val, my_nnz, fail + 1, n_fail, false));
  // ---- 2. everybody's candidates
  const bool loop = rp_comm_is_loopback(comm);
  if (loop) rp_comm_loop_replicate(comm, true);  // (an emulated rank scores its own users' candidates in the others' places: the same work)
  RP_TRY(rp_comm_all_gather(comm, my_idx, A_idx, (int64_t)slotA * 4, st));
  RP_TRY(rp_comm_all_gather(comm, my_val, A_val, (int64_t)slotA * 4, st));
  RP_TRY(rp_comm_all_gather(comm, my_nnz, A_nnz, (int64_t)n_max * 4, st));
  if (loop) rp_comm_loop_replicate(comm, false);
  // ---- 3. the exact sums of the candidates in this rank's columns, users of all ranks
  auto fill_rescore = [&](rp_rescore_params& R, int r) {
    R.x_ptr = X->x_ptr;
    R.x_idx = X->x_idx;
    R.delta = rows8->b16_delta;
    R.rowmax = rows8->b16_rowmax;
    R.cand_idx = A_idx + (size_t)r * slotA;
    R.cand_val = A_val + (size_t)r * slotA;
    R.cand_nnz = A_nnz + (size_t)r * n_max;
    R.user_begin = (int)user_bounds[r];
    R.n_users = (int)(user_bounds[r + 1] - user_bounds[r]);
    R.Kp = Kp;
    R.Kc = Kc;
  };
  for (int r = 0; r < W; ++r) {
    const int64_t n_r = user_bounds[r + 1] - user_bounds[r];
    if (n_r == 0) continue;
    float* dst = part_s + (size_t)user_bounds[r] * Kc;
    if (!local) {  // no columns here: every part is zero
      hipMemsetAsync(dst, 0, (size_t)n_r * Kc * 4, st);
      continue;
    }
    rp_rescore_params R{};
    fill_rescore(R, r);
    R.B = local->B;
    R.ld = local->ld;
    R.col_begin = PL.col_begin;
    R.cyc_nb = PL.cyc_nb;
    R.cyc_w = PL.cyc_w;
    R.cyc_r = PL.cyc_r;
    R.n_own_cols = (int)local->n_cols;
    R.part_out = dst;
    rp_launch_rescore_kc<1>(ctx, R);
  }
  RP_HIP(hipGetLastError());
  // ---- 4. the parts to the users' ranks
  std::vector<int64_t> soff(W), sby(W), roff(W), rby(W);
  for (int r = 0; r < W; ++r) {
    soff[r] = user_bounds[r] * (int64_t)Kc * 4;
    sby[r] = (user_bounds[r + 1] - user_bounds[r]) * (int64_t)Kc * 4;
    roff[r] = (int64_t)r * n_me * Kc * 4;
    rby[r] = n_me * (int64_t)Kc * 4;
  }
  RP_TRY(rp_comm_exchange(comm, part_s, soff.data(), sby.data(), part_r, roff.data(), rby.data(), st));
  ctx->owner_stats[3] = 0;
  if (loop && rows8->B && n_me > 0) {
    // (diagnostic: one rank of several measured alone, tools/dist_rank_emulate.py) what the other ranks would have sent never existed.
    // When the int8 rows come with their f32 matrix - the emulation passes the single-GPU model - the parts of this rank's users are
    // formed from it, all columns, so that the ranking below sees the true scores and sends the true few users to the exact pass; the
    // time of this stand-in is recorded (owner_stats[3], microseconds) and the emulation subtracts it.
    hipEvent_t o0 = nullptr, o1 = nullptr;
    RP_HIP(hipEventCreate(&o0));
    RP_HIP(hipEventCreate(&o1));
    RP_HIP(hipEventRecord(o0, st));
    hipMemsetAsync(part_r, 0, (size_t)n_me * Kc * 4 * W, st);
    rp_rescore_params R{};
    fill_rescore(R, me);
    R.cand_idx = my_idx;
    R.cand_val = my_val;
    R.cand_nnz = my_nnz;
    R.B = rows8->B;
    R.ld = rows8->ld;
    R.col_begin = 0;
    R.n_own_cols = (int)rows8->n_items;
    R.part_out = part_r;
    rp_launch_rescore_kc<1>(ctx, R);
    RP_HIP(hipEventRecord(o1, st));
    RP_HIP(hipEventSynchronize(o1));
    float oms = 0.0f;
    hipEventElapsedTime(&oms, o0, o1);
    hipEventDestroy(o0);
    hipEventDestroy(o1);
    ctx->owner_stats[3] = (int64_t)(oms * 1000.0f);
  }
  // ---- 5. ranking and proof for this rank's users
  if (n_me > 0) {
    rp_rescore_params R{};
    fill_rescore(R, me);
    R.cand_idx = my_idx;
    R.cand_val = my_val;
    R.cand_nnz = my_nnz;
    R.part_in = part_r;
    R.n_parts = W;
    R.out_idx = idx_dev;
    R.out_val = val_dev;
    R.out_nnz = nnz_dev;
    R.fail_list = fail + 1;
    R.n_fail = n_fail;
    R.fail_base = 0;
    rp_launch_rescore_kc<2>(ctx, R);
    RP_HIP(hipGetLastError());
  }
  // ---- the users without a proof, of all ranks: [count, local ids] per rank
  RP_HIP(hipMemcpyAsync(fail, n_fail, 4, hipMemcpyDeviceToDevice, st));
  int32_t *glist = nullptr, *goff_dev = nullptr;
  int64_t* bounds_dev = nullptr;
  RP_TRY(rp_comm_all_gather(comm, fail, G, (int64_t)fslot * 4, st));
  std::vector<int32_t> cnt(W, 0), off(W + 1, 0);
  for (int r = 0; r < W; ++r) RP_HIP(hipMemcpyAsync(&cnt[r], G + (size_t)r * fslot, 4, hipMemcpyDeviceToHost, st));
  RP_HIP(hipStreamSynchronize(st));
  for (int r = 0; r < W; ++r) {
    if (cnt[r] < 0 || cnt[r] > user_bounds[r + 1] - user_bounds[r]) RP_FAIL(RPHIP_EHIP, "rank %d reports %d users without a proof of its %lld", r, cnt[r], (long long)(user_bounds[r + 1] - user_bounds[r]));
    off[r + 1] = off[r] + cnt[r];
  }
  const int n_f = off[W], f_me = cnt[me];
  ctx->owner_stats[0] = n_f;
  ctx->owner_stats[1] = f_me;
  ctx->owner_stats[2] = Kc;
  if (n_f > 0) {
    // the second group of allocations (sizes known only now; n_f and the counts are the same numbers on every rank): agreed as above
    int32_t *L_idx = nullptr, *L_nnz = nullptr, *t_idx = nullptr, *t_nnz = nullptr, *ci = nullptr, *cn = nullptr;
    float *L_val = nullptr, *t_val = nullptr, *cv = nullptr;
    char* recvb = nullptr;
    const int64_t cs = std::max(f_me, 1);
    const int64_t slot = cs * ((int64_t)Kp * 8 + 4);
    arc = RPHIP_OK;
    get((size_t)n_f * 4, (void**)&glist);
    get((size_t)(W + 1) * 4, (void**)&goff_dev);
    get((size_t)(W + 1) * 8, (void**)&bounds_dev);
    get((size_t)n_f * Kp * 4, (void**)&L_idx);
    get((size_t)n_f * Kp * 4, (void**)&L_val);
    get((size_t)n_f * 4, (void**)&L_nnz);
    if (local && T_l > 1) {
      get((size_t)n_f * T_l * Kp * 4, (void**)&t_idx);
      get((size_t)n_f * T_l * Kp * 4, (void**)&t_val);
      get((size_t)n_f * T_l * 4, (void**)&t_nnz);
    }
    get((size_t)slot * W, (void**)&recvb);
    if (f_me > 0) {
      get((size_t)f_me * W * Kp * 4, (void**)&ci);
      get((size_t)f_me * W * Kp * 4, (void**)&cv);
      get((size_t)f_me * W * 4, (void**)&cn);
    }
    {
      int32_t h_sum[2] = {1, 1};
      RP_TRY(agree(arc != RPHIP_OK, false, h_sum));
      if (h_sum[0] != 0)
        RP_FAIL(RPHIP_ENOMEM, "%d of %d ranks could not allocate the exact pass's lists for %d users without a proof: use rphip_predict_topk_dist", (int)h_sum[0], W, n_f);
    }
    RP_HIP(hipMemcpyAsync(goff_dev, off.data(), (size_t)(W + 1) * 4, hipMemcpyHostToDevice, st));
    RP_HIP(hipMemcpyAsync(bounds_dev, user_bounds, (size_t)(W + 1) * 8, hipMemcpyHostToDevice, st));
    RP_HIP(hipStreamSynchronize(st));  // (the host arrays are read by the copies)
    hipLaunchKernelGGL(k_owner_fail_concat, dim3(64, W), dim3(256), 0, st, G, fslot, W, goff_dev, bounds_dev, glist);
    // local top-Kp of every such user over this rank's columns: [n_f][Kp] by position in the list
    hipMemsetAsync(L_nnz, 0, (size_t)n_f * 4, st);
    if (local) {
      rp_pred_params F = PL;
      F.Kp = Kp;
      F.exclude_history = exclude_history;
      F.mode = 0;
      F.user_begin = 0;
      F.n_users = n_f;
      F.user_list = glist;
      F.n_list = nullptr;
      const int T = F.n_tiles;
      const size_t lds1 = rp_pred_lds(F);
      RP_HIP(hipFuncSetAttribute((const void*)k_predict_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
      if (T == 1) {
        F.list_compact = 2;
        F.out_idx = L_idx;
        F.out_val = L_val;
        F.out_nnz = L_nnz;
        hipLaunchKernelGGL(k_predict_tile<true>, dim3((unsigned)std::min<int64_t>(n_f, (int64_t)ctx->n_cu * 8)), dim3(RP_PRED_THREADS), lds1, st, F);
      } else {
        hipMemsetAsync(t_nnz, 0, (size_t)n_f * T * 4, st);
        F.list_compact = 1;
        F.out_idx = t_idx;
        F.out_val = t_val;
        F.out_nnz = t_nnz;
        hipLaunchKernelGGL(k_predict_tile<true>, dim3((unsigned)std::min<int64_t>((int64_t)n_f * T, (int64_t)ctx->n_cu * 8)), dim3(RP_PRED_THREADS), lds1, st, F);
        const size_t mlds_p = mlds_t;  // (checked before the first collective)
        RP_HIP(hipFuncSetAttribute((const void*)k_predict_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds_p));
        // (the tiles of a block-cyclic slab are not consecutive column ranges: sort_cols; rows by list position: no user list)
        hipLaunchKernelGGL(k_predict_merge, dim3((unsigned)std::min<int64_t>(n_f, (int64_t)ctx->n_cu * 8)), dim3(RP_PRED_THREADS), mlds_p, st, t_idx, t_val, t_nnz, T,
                           Kp, n_f, nullptr, nullptr, L_idx, L_val, L_nnz, 1);
      }
      RP_HIP(hipGetLastError());
    }
    // the lists to the users' ranks: slots [source][f_me][Kp] columns, scores, [f_me] counts - k_lists_to_users' layout
    for (int sec = 0; sec < 3; ++sec) {  // columns, scores, counts
      const int64_t per = sec < 2 ? (int64_t)Kp * 4 : 4;
      const char* sendb = sec == 0 ? (const char*)L_idx : (sec == 1 ? (const char*)L_val : (const char*)L_nnz);
      for (int r = 0; r < W; ++r) {
        soff[r] = (int64_t)off[r] * per;
        sby[r] = (int64_t)cnt[r] * per;
        roff[r] = (int64_t)r * slot + (sec == 0 ? 0 : (sec == 1 ? cs * Kp * 4 : cs * Kp * 8));
        rby[r] = (int64_t)f_me * per;
      }
      RP_TRY(rp_comm_exchange(comm, sendb, soff.data(), sby.data(), recvb, roff.data(), rby.data(), st));
    }
    if (f_me > 0) {
      const long long total = (long long)f_me * W * Kp;
      hipLaunchKernelGGL(k_lists_to_users, dim3((unsigned)std::min<long long>(8192, rp_cdiv(total, 256))), dim3(256), 0, st, recvb, (long long)slot, W, (int)cs, f_me,
                         (int)Kp, ci, cv, cn);
      const size_t mlds = mlds_w;  // (checked before the first collective)
      RP_HIP(hipFuncSetAttribute((const void*)k_predict_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
      // rows of this rank's users without a proof: user_list = their local ids (behind the count in `fail`), lists by position
      hipLaunchKernelGGL(k_predict_merge, dim3((unsigned)std::min<int64_t>(f_me, (int64_t)ctx->n_cu * 8)), dim3(RP_PRED_THREADS), mlds, st, ci, cv, cn, W, (int)Kp, f_me,
                         fail + 1, nullptr, idx_dev, val_dev, nnz_dev, 1, 0, 1);
      RP_HIP(hipGetLastError());
    }
  }
  rp_time_end(ctx, ev, RPHIP_T_PREDICT_DENSE, 0.0);
  return RPHIP_OK;
}

// (diagnostic, include/recpack_hip_diag.h) the last rphip_predict_topk_owner of this context: [0] users without a proof over all
// ranks, [1] this rank's, [2] candidates per user, [3] loop-back communicators only: microseconds of the stand-in for the other ranks' parts
extern "C" int rphip_predict_owner_stats(rphip_ctx* ctx, int64_t* out4) {
  RP_ENTER(ctx);
  if (!ctx || !out4) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  ::memcpy(out4, ctx->owner_stats, sizeof(ctx->owner_stats));
  return RPHIP_OK;
}

// (diagnostic, include/recpack_hip_diag.h) the last single-GPU predict of this context
extern "C" int rphip_predict_stats(rphip_ctx* ctx, int64_t* out6) {
  RP_ENTER(ctx);
  if (!ctx || !out6) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  ::memcpy(out6, ctx->predict_stats, sizeof(ctx->predict_stats));
  return RPHIP_OK;
}

extern "C" int rphip_predict_topk_dev(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                      int64_t user_end, int32_t Kp, int32_t exclude_history, int32_t* idx_dev, float* val_dev,
                                      int32_t* nnz_dev) {
  RP_ENTER(ctx);
  return rp_predict_topk_impl(ctx, model, Xpred, user_begin, user_end, Kp, exclude_history, nullptr, idx_dev, val_dev, nnz_dev);
}

extern "C" int rphip_predict_topk_filtered_dev(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                               int64_t user_end, int32_t Kp, int32_t exclude_history, const uint8_t* item_keep_dev,
                                               int32_t* idx_dev, float* val_dev, int32_t* nnz_dev) {
  RP_ENTER(ctx);
  return rp_predict_topk_impl(ctx, model, Xpred, user_begin, user_end, Kp, exclude_history, item_keep_dev, idx_dev, val_dev, nnz_dev);
}

extern "C" int rphip_predict_topk(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                  int64_t user_end, int32_t Kp, int32_t exclude_history, int32_t* idx_host, float* val_host,
                                  int32_t* nnz_host) {
  RP_ENTER(ctx);
  return rphip_predict_topk_filtered(ctx, model, Xpred, user_begin, user_end, Kp, exclude_history, nullptr, idx_host, val_host, nnz_host);
}

extern "C" int rphip_predict_topk_filtered(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                           int64_t user_end, int32_t Kp, int32_t exclude_history, const uint8_t* item_keep_host,
                                           int32_t* idx_host, float* val_host, int32_t* nnz_host) {
  RP_ENTER(ctx);
  RP_TRY(rp_pred_check(ctx, model, Xpred, user_begin, user_end));
  RP_TRY(rp_model_gather_wait(ctx, model, true));
  if (!idx_host || !val_host || !nnz_host) RP_FAIL(RPHIP_EINVAL, "NULL output");
  if (Kp < 1) RP_FAIL(RPHIP_EINVAL, "Kp=%d must be at least 1", Kp);  // (beyond RP_K_MAX: rp_predict_topk_big)
  RP_TRY(rp_set_device(ctx));
  const int64_t n = user_end - user_begin;
  if (n == 0) return RPHIP_OK;
  char* ob = nullptr;
  RP_TRY(rp_scratch(ctx, 3, (size_t)n * Kp * 8 + (size_t)n * 4 + 64, (void**)&ob));
  int32_t* di = (int32_t*)ob;
  float* dv = (float*)(ob + (size_t)n * Kp * 4);
  int32_t* dn = (int32_t*)(ob + (size_t)n * Kp * 8);
  uint8_t* keep_dev = nullptr;
  if (item_keep_host) {
    RP_TRY(rp_scratch(ctx, 1, (size_t)model->n_items, (void**)&keep_dev));
    RP_HIP(hipMemcpyAsync(keep_dev, item_keep_host, (size_t)model->n_items, hipMemcpyHostToDevice, ctx->stream));
  }
  RP_TRY(rp_predict_topk_impl(ctx, model, Xpred, user_begin, user_end, Kp, exclude_history, keep_dev, di, dv, dn));
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
  RP_ENTER(ctx);
  RP_TRY(rp_pred_check(ctx, model, Xpred, user_begin, user_end));
  RP_TRY(rp_model_gather_wait(ctx, model, true));
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
  const int grid = (int)std::min<int64_t>(n, (int64_t)ctx->n_cu * 8);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  RP_TRY(rp_pred_launch(ctx, P, grid));
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
  RP_ENTER(ctx);
  RP_TRY(rp_pred_check(ctx, model, Xpred, user_begin, user_end));
  RP_TRY(rp_model_gather_wait(ctx, model, true));
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
  const int grid = (int)std::min<int64_t>(n, (int64_t)ctx->n_cu * 8);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  RP_TRY(rp_pred_launch(ctx, P, grid));
  rp_time_end(ctx, ev, RPHIP_T_PREDICT, rp_pred_work(Xpred, model, n, 0) + 12.0 * (double)total);
  RP_HIP(hipMemcpyAsync(indices_host, di, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipMemcpyAsync(data_host, dv, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

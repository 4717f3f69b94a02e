// The merge of a row's per-tile candidate lists for ANY K (round 6; recpack takes any K: recpack/util.py:63-66, base.py:302-304).
//
// The selections of rp_select.h number their candidates in 16 bits with a flag (RPS_FLAG): a row's T per-tile lists may hold at most
// 32 767 candidates.  Here nothing is numbered.  One workgroup per row walks the lists where they lie in memory:
//   * an MSD radix select over the 64-bit key (11-bit digits, at most six passes, each a histogram of the candidates that share the
//     decided prefix) finds the K-th key;
//   * a last pass in position order - tiles are ascending column ranges and a tile's list is in column order, so position order IS
//     column order - emits everything above it and, of the candidates that tie with it, the first ones: ties to the smaller column,
//     as everywhere;
//   * a row with at most K candidates (the common case at such K) is copied out.
// Shared by the binary fit (rp_itemknn.hip), normalize_X (rp_itemknn_w.hip) and the valued / time-aware fits (rp_itemknn_v.hip).
// A rare route: simple, not tuned.
#pragma once
#include "rp_common.h"
#include "rp_wave.h"

constexpr int RP_MBIG_THREADS = 256;

struct rp_mbig_smem {
  uint32_t hist[2048];
  int off[17], res[4], w[2][RP_MBIG_THREADS / 64];
};

// n_lists lists of a row: list t holds cnt(t) candidates at addresses t * stride + [0, cnt(t)).  key_at(address) -> uint64 (larger
// is better), emit(position in the output row, address).  K_sel candidates are selected (all of them if there are fewer); returns how
// many were emitted.  Every thread of the workgroup must call it.
template <class Cnt, class KeyAt, class Emit>
__device__ int rp_merge_big_row(rp_mbig_smem& S, int n_lists, int stride, int K_sel, Cnt cnt, KeyAt key_at, Emit emit) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const int T = n_lists;
  __syncthreads();
  if (tid == 0) {
    int o = 0;
    for (int t = 0; t < T; ++t) {
      S.off[t] = o;
      o += cnt(t);
    }
    S.off[T] = o;
  }
  __syncthreads();
  const int n = S.off[T];
  auto at = [&](int w) {
    int t = 0;
    while (t + 1 < T && w >= S.off[t + 1]) ++t;
    return t * stride + (w - S.off[t]);
  };
  if (K_sel <= 0) return 0;
  const bool all = n <= K_sel;
  uint64_t prefix = 0;
  int need = K_sel, sh = 0;
  bool fin_all = false;
  if (!all) {
    for (int level = 0; level < 6; ++level) {
      const int hi = 64 - 11 * level, width = min(11, hi), shift = hi - width;
      for (int i = tid; i < 2048; i += RP_MBIG_THREADS) S.hist[i] = 0u;
      __syncthreads();
      for (int w = tid; w < n; w += RP_MBIG_THREADS) {
        const uint64_t k = key_at(at(w));
        if (level == 0 || (k >> hi) == (prefix >> hi)) atomicAdd(&S.hist[(uint32_t)(k >> shift) & ((1u << width) - 1u)], 1u);
      }
      __syncthreads();
      if (wave == 0) {  // the bin where the count from the top reaches `need`: lane l owns the 32 bins below 2047 - 32 l
        int h[32], sum = 0;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
          h[e] = (int)S.hist[2047 - 32 * lane - e];
          sum += h[e];
        }
        int above = rp_wave_incl_scan(sum, lane) - sum;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
          if (above < need && need <= above + h[e]) {
            S.res[0] = 2047 - 32 * lane - e;
            S.res[1] = above;
            S.res[2] = h[e];
          }
          above += h[e];
        }
      }
      __syncthreads();
      need -= S.res[1];
      prefix |= (uint64_t)(uint32_t)S.res[0] << shift;
      sh = shift;
      if (S.res[2] == need) {  // the whole boundary bin is selected
        fin_all = true;
        break;
      }
      __syncthreads();
    }
  }
  // ---- emission in position order; `need` of the members of the boundary group (all of them when fin_all)
  int run_sel = 0, run_mem = 0;
  for (int base = 0; base < n; base += RP_MBIG_THREADS) {
    const int w = base + tid;
    int a = 0;
    bool above = false, member = false;
    if (w < n) {
      a = at(w);
      if (all) {
        above = true;
      } else {
        const uint64_t k = key_at(a);
        above = (k >> sh) > (prefix >> sh);
        member = (k >> sh) == (prefix >> sh);
      }
    }
    const uint64_t mm = __ballot(member);
    if (lane == 0) S.w[0][wave] = __popcll(mm);
    __syncthreads();
    int mem_before = run_mem + __popcll(mm & lt_mask);
    int mem_total = 0;
    for (int q = 0; q < RP_MBIG_THREADS / 64; ++q) {
      if (q < wave) mem_before += S.w[0][q];
      mem_total += S.w[0][q];
    }
    const bool sel = above || (member && (fin_all || mem_before < need));
    const uint64_t sm = __ballot(sel);
    if (lane == 0) S.w[1][wave] = __popcll(sm);
    __syncthreads();
    int pos = run_sel + __popcll(sm & lt_mask);
    int sel_total = 0;
    for (int q = 0; q < RP_MBIG_THREADS / 64; ++q) {
      if (q < wave) pos += S.w[1][q];
      sel_total += S.w[1][q];
    }
    if (sel) emit(pos, a);
    run_sel += sel_total;
    run_mem += mem_total;
    __syncthreads();
  }
  return run_sel;
}

// Wave-private running top-K pool (no workgroup barriers): the selection machinery of the wave-per-row fit
// kernel and the wave-per-user predict kernel.
//
// A wave streams candidates (key, column) in ASCENDING column order.  Candidates whose key beats the current
// threshold are appended to an LDS pool; when the pool is full it is cut to the K best by an exact wave-level
// radix select (8-bit digits, 256-bin LDS histogram, boundary group of <= 64 resolved by ranking in registers).
// After a cut the threshold is the K-th key, and a later candidate must be STRICTLY larger to enter: it has a
// larger column than everything in the pool, so on equal keys it loses - which is exactly the order
// "larger key first, then smaller column" of recpack/util.py:50-96 as DESIGN.md restates it.
#pragma once

#include "rp_common.h"

// LDS pointers carry their address space: a plain (generic) pointer kept in a struct or made volatile compiles to
// FLAT loads/stores, which count on vmcnt AND lgkmcnt - every wait on one then also drains the global loads in flight.
#define RP_LDS __attribute__((address_space(3)))
// LDS accesses of one wave execute in program order, so wave-synchronous code (lane A writes, lane B reads, no
// barrier) only needs the COMPILER to keep the order.  `volatile` does that too, but the memory legalizer then puts an
// s_waitcnt vmcnt(0) behind every such access, which drains the global loads a kernel has in flight; a wavefront-scope
// fence orders the accesses at no cost.
#define RP_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

template <typename KeyT>
struct rp_wpool {
  RP_LDS KeyT* key;        // [CAP]
  RP_LDS uint32_t* pay;    // [CAP] payload; the column is pay >> col_shift
  RP_LDS uint32_t* hist;   // [256]
  RP_LDS uint16_t* gidx;   // [64]
  int n;            // entries in the pool (wave-uniform)
  KeyT tau;         // admission threshold: a candidate enters if key > tau, or key == tau and column < tau_col
  uint32_t tau_col; // (0 / 0 = everything positive)
};

template <typename KeyT>
__device__ __forceinline__ KeyT rp_shfl_key(KeyT v, int src);
template <>
__device__ __forceinline__ uint32_t rp_shfl_key<uint32_t>(uint32_t v, int src) {
  return (uint32_t)__shfl((int)v, src);
}
template <>
__device__ __forceinline__ uint64_t rp_shfl_key<uint64_t>(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

// Wave-wide inclusive scans on the VALU (DPP row shifts + row broadcasts, gfx9 encodings): no LDS round trips, where
// a __shfl_up ladder is six dependent ds_bpermute.  Out-of-row sources read 0 (old = 0, bound_ctrl off).
#define RP_DPP_ROW_SHR(n) (0x110 + (n))
#define RP_DPP_ROW_BCAST15 0x142
#define RP_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ int rp_wave_incl_scan(int v, int /*lane*/) {
  v += __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_SHR(1), 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_SHR(2), 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_SHR(4), 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_SHR(8), 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_BCAST15, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_BCAST31, 0xc, 0xf, false);
  return v;
}
// inclusive running maximum of non-negative values
__device__ __forceinline__ int rp_wave_incl_max(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_SHR(1), 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_SHR(2), 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_SHR(4), 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_SHR(8), 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_BCAST15, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, RP_DPP_ROW_BCAST31, 0xc, 0xf, false));
  return v;
}

// OR over the wave (returned in every lane's SGPR-uniform result)
__device__ __forceinline__ uint32_t rp_wave_or(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RP_DPP_ROW_SHR(1), 0xf, 0xf, false);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RP_DPP_ROW_SHR(2), 0xf, 0xf, false);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RP_DPP_ROW_SHR(4), 0xf, 0xf, false);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RP_DPP_ROW_SHR(8), 0xf, 0xf, false);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RP_DPP_ROW_BCAST15, 0xa, 0xf, false);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RP_DPP_ROW_BCAST31, 0xc, 0xf, false);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

#ifndef RP_TRIM_SLACK
#define RP_TRIM_SLACK 96  // free entries a trim must leave (64 are needed by the push that asked; measured 64 / 96 / 128: see DESIGN.md)
#endif

template <typename KeyT>
struct rp_wcut {
  KeyT tau;
  uint32_t tau_col;
  int n;
};

// The work of a cut on plain values (pointers and sizes in, threshold out): if the compiler keeps this large function
// out of line, the pool struct itself is never address-taken and stays in registers in the callers' hot loops.
template <typename KeyT, int COL_SHIFT>
__device__ rp_wcut<KeyT> rp_wpool_cut_impl(RP_LDS KeyT* pkey_, RP_LDS uint32_t* ppay_, RP_LDS uint32_t* phist_, RP_LDS uint16_t* pgidx_,
                                           int n, int K, int lane) {
  constexpr int KEY_LEVELS = sizeof(KeyT);  // 8-bit digits
  RP_LDS KeyT* key = pkey_;
  RP_LDS uint32_t* pay = ppay_;
  RP_LDS uint32_t* hist = phist_;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  KeyT pkey = 0;
  uint32_t pic = 0;  // decided prefix of ~column
  int need = K;
  KeyT tkey = 0;
  uint32_t tcol = 0xffffffffu;  // selected <=> key > tkey || (key == tkey && col <= tcol)
  bool done = false;
  // Skip the digits every key shares (fixed-point sums are far below the overflow bound, so the top bytes agree):
  // OR of key ^ key[0] over the pool, first differing byte = first level worth a pass.
  int level0 = 0;
  {
    const KeyT k0 = key[0];
    KeyT diff = 0;
    for (int e = lane; e < n; e += 64) diff |= key[e] ^ k0;
    uint32_t dlo = rp_wave_or((uint32_t)diff), dhi = 0;
    if (sizeof(KeyT) == 8) dhi = rp_wave_or((uint32_t)((uint64_t)diff >> 32));
    const uint64_t d64 = sizeof(KeyT) == 8 ? (((uint64_t)dhi << 32) | dlo) : ((uint64_t)dlo << 32);
    level0 = d64 ? (__builtin_clzll(d64) >> 3) : KEY_LEVELS;  // all keys equal: straight to the column digits
    if (level0 > 0) pkey = level0 >= KEY_LEVELS ? k0 : (KeyT)(k0 & ~((((KeyT)1) << (8 * (KEY_LEVELS - level0))) - 1));
  }
  for (int level = level0; level < KEY_LEVELS + 4 && !done; ++level) {
    const bool is_col = level >= KEY_LEVELS;
    const int shift = is_col ? 8 * (3 - (level - KEY_LEVELS)) : 8 * (KEY_LEVELS - 1 - level);
#pragma unroll
    for (int b = 0; b < 4; ++b) hist[lane * 4 + b] = 0;
    RP_WAVE_FENCE();
    for (int e = lane; e < n; e += 64) {
      const KeyT k = key[e];
      const uint32_t ic = ~(pay[e] >> COL_SHIFT);
      bool in_group;
      uint32_t d;
      if (!is_col) {
        in_group = (level == 0) || ((k >> shift) >> 8) == ((pkey >> shift) >> 8);  // the skipped prefix is shared by all
        d = (uint32_t)(k >> shift) & 255u;
      } else {
        in_group = (k == pkey) && ((level == KEY_LEVELS) || ((ic >> shift) >> 8) == ((pic >> shift) >> 8));
        d = (ic >> shift) & 255u;
      }
      if (in_group) atomicAdd((uint32_t*)&phist_[d], 1u);
    }
    RP_WAVE_FENCE();
    // lane owns bins 255-4*lane .. 252-4*lane (descending)
    int h[4];
    int sum = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      h[b] = (int)hist[255 - 4 * lane - b];
      sum += h[b];
    }
    int above = rp_wave_incl_scan(sum, lane) - sum;
    int fbin = -1, fabove = 0, fgroup = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (above < need && need <= above + h[b]) {
        fbin = 255 - 4 * lane - b;
        fabove = above;
        fgroup = h[b];
      }
      above += h[b];
    }
    const uint64_t fm = __ballot(fbin >= 0);
    const int src = __ffsll((unsigned long long)fm) - 1;
    const int bin = __shfl(fbin, src);
    const int group = __shfl(fgroup, src);
    need -= __shfl(fabove, src);
    if (!is_col)
      pkey |= (KeyT)bin << shift;
    else
      pic |= (uint32_t)bin << shift;
    if (group == need) {  // the whole boundary group is selected
      if (!is_col) {
        tkey = pkey;
        tcol = 0xffffffffu;
      } else {
        tkey = pkey;
        tcol = ~pic;  // ~col >= pic  <=>  col <= ~pic
      }
      done = true;
    } else if (group <= 64) {
      // exact: gather the group (one member per lane), rank it, the member of rank need-1 is the threshold
      int base = 0;
      for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        bool mem = false;
        if (e < n) {
          const KeyT k = key[e];
          const uint32_t ic = ~(pay[e] >> COL_SHIFT);
          if (!is_col)
            mem = (k >> shift) == (pkey >> shift);
          else
            mem = (k == pkey) && ((ic >> shift) == (pic >> shift));
        }
        const uint64_t m = __ballot(mem);
        if (mem) pgidx_[base + __popcll(m & lt_mask)] = (uint16_t)e;
        base += __popcll(m);
      }
      RP_WAVE_FENCE();
      KeyT mk = 0;
      uint32_t mic = 0;
      if (lane < group) {
        const int e = pgidx_[lane];
        mk = key[e];
        mic = ~(pay[e] >> COL_SHIFT);
      }
      int rank = 0;
      for (int m = 0; m < group; ++m) {  // v_readlane broadcasts (m is uniform): no LDS round trips
        KeyT ok = (KeyT)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mk, m);
        if (sizeof(KeyT) == 8) ok |= (KeyT)((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)mk >> 32), m) << 32);
        const uint32_t oic = (uint32_t)__builtin_amdgcn_readlane((int)mic, m);
        rank += ((ok > mk) || (ok == mk && oic > mic)) ? 1 : 0;
      }
      const uint64_t tm = __ballot(lane < group && rank == need - 1);
      const int tsrc = __ffsll((unsigned long long)tm) - 1;
      tkey = rp_shfl_key<KeyT>(mk, tsrc);
      tcol = ~(uint32_t)__shfl((int)mic, tsrc);
      done = true;
    }
  }
  // rebuild the pool in place, 64 entries per round: round r writes at most 64 selected entries at positions
  // <= 64 r + 63, all of which this or an earlier round has already read (one wave: LDS accesses are in order)
  int out = 0;
  for (int e0 = 0; e0 < n; e0 += 64) {
    const int e = e0 + lane;
    const KeyT rk = (e < n) ? key[e] : (KeyT)0;
    const uint32_t rp = (e < n) ? pay[e] : 0u;
    const uint32_t c = rp >> COL_SHIFT;
    const bool sel = (e < n) && ((rk > tkey) || (rk == tkey && c <= tcol));
    const uint64_t m = __ballot(sel);
    if (sel) {
      const int pos = out + __popcll(m & lt_mask);
      key[pos] = rk;
      pay[pos] = rp;
    }
    out += __popcll(m);
  }
  RP_WAVE_FENCE();
  rp_wcut<KeyT> r;
  r.n = out;  // == K
  r.tau = tkey;
  r.tau_col = tcol;
  return r;
}

// A cheaper way to make room: drop every entry whose key lies below the lower edge of the radix bin that holds the K-th best
// key (digits resolved level by level until at most `limit` entries stay).  The K best are all kept - plus whoever shares the
// boundary bin - and the edge becomes the admission bound; nothing is ranked and the columns are never looked at, which is most
// of an exact cut's work.  Returns n = -1 when the keys cannot be separated that way (massive ties, or an edge of 0): the caller
// then cuts exactly.
template <typename KeyT, int COL_SHIFT>
__device__ rp_wcut<KeyT> rp_wpool_trim_impl(RP_LDS KeyT* pkey_, RP_LDS uint32_t* ppay_, RP_LDS uint32_t* phist_, int n, int K, int limit,
                                            int lane) {
  constexpr int KEY_LEVELS = sizeof(KeyT);
  RP_LDS KeyT* key = pkey_;
  RP_LDS uint32_t* pay = ppay_;
  RP_LDS uint32_t* hist = phist_;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  rp_wcut<KeyT> r;
  r.n = -1;
  r.tau = 0;
  r.tau_col = 0;
  KeyT pkey = 0;
  int need = K;
  int level0 = 0;
  {
    const KeyT k0 = key[0];
    KeyT diff = 0;
    for (int e = lane; e < n; e += 64) diff |= key[e] ^ k0;
    uint32_t dlo = rp_wave_or((uint32_t)diff), dhi = 0;
    if (sizeof(KeyT) == 8) dhi = rp_wave_or((uint32_t)((uint64_t)diff >> 32));
    const uint64_t d64 = sizeof(KeyT) == 8 ? (((uint64_t)dhi << 32) | dlo) : ((uint64_t)dlo << 32);
    if (!d64) return r;  // all keys equal
    level0 = __builtin_clzll(d64) >> 3;
    if (level0 > 0) pkey = (KeyT)(k0 & ~((((KeyT)1) << (8 * (KEY_LEVELS - level0))) - 1));
  }
  KeyT edge = 0;
  bool found = false;
  for (int level = level0; level < KEY_LEVELS && !found; ++level) {
    const int shift = 8 * (KEY_LEVELS - 1 - level);
#pragma unroll
    for (int b = 0; b < 4; ++b) hist[lane * 4 + b] = 0;
    RP_WAVE_FENCE();
    for (int e = lane; e < n; e += 64) {
      const KeyT k = key[e];
      const bool in_group = (level == 0) || ((k >> shift) >> 8) == ((pkey >> shift) >> 8);
      if (in_group) atomicAdd((uint32_t*)&phist_[(uint32_t)(k >> shift) & 255u], 1u);
    }
    RP_WAVE_FENCE();
    int h[4];
    int sum = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      h[b] = (int)hist[255 - 4 * lane - b];
      sum += h[b];
    }
    int above = rp_wave_incl_scan(sum, lane) - sum;
    int fbin = -1, fabove = 0, fgroup = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (above < need && need <= above + h[b]) {
        fbin = 255 - 4 * lane - b;
        fabove = above;
        fgroup = h[b];
      }
      above += h[b];
    }
    const uint64_t fm = __ballot(fbin >= 0);
    const int src = __ffsll((unsigned long long)fm) - 1;
    const int bin = __shfl(fbin, src);
    const int group = __shfl(fgroup, src);
    const int ab = __shfl(fabove, src);
    pkey |= (KeyT)bin << shift;
    // kept if the walk stops here: the K - need entries above the group, the bins above the boundary one, the boundary bin
    if ((K - need) + ab + group <= limit) {
      edge = pkey;
      found = true;
    }
    need -= ab;
  }
  if (!found || edge == 0) return r;
  int out = 0;
  for (int e0 = 0; e0 < n; e0 += 64) {
    const int e = e0 + lane;
    const KeyT rk = (e < n) ? key[e] : (KeyT)0;
    const uint32_t rp = (e < n) ? pay[e] : 0u;
    const bool sel = (e < n) && rk >= edge;
    const uint64_t m = __ballot(sel);
    if (sel) {
      const int pos = out + __popcll(m & lt_mask);
      key[pos] = rk;
      pay[pos] = rp;
    }
    out += __popcll(m);
  }
  RP_WAVE_FENCE();
  r.n = out;
  r.tau = (KeyT)(edge - 1);  // key > tau  <=>  key >= edge
  r.tau_col = 0;
  return r;
}

// Cut the pool to its K best entries by (key desc, column asc); requires pool.n > K.  Sets pool.tau / tau_col.
template <typename KeyT, int COL_SHIFT>
__device__ __forceinline__ void rp_wpool_cut(rp_wpool<KeyT>& P, int K, int lane) {
  const rp_wcut<KeyT> r = rp_wpool_cut_impl<KeyT, COL_SHIFT>(P.key, P.pay, P.hist, P.gidx, P.n, K, lane);
  P.n = r.n;
  P.tau = r.tau;
  P.tau_col = r.tau_col;
}

// Is a candidate still able to make the top K?  Exact for candidates arriving in ANY column order: ties with the
// K-th key are admitted only when their column is smaller than the K-th entry's.
template <typename KeyT, int COL_SHIFT>
__device__ __forceinline__ bool rp_wpool_admits(const rp_wpool<KeyT>& P, KeyT k, uint32_t pay) {
  return (k > P.tau) || (k == P.tau && k != 0 && (pay >> COL_SHIFT) < P.tau_col);
}

// Append the lanes' candidates (keep == true) to the pool; cuts first when fewer than 64 slots are free.
template <typename KeyT, int COL_SHIFT>
__device__ __forceinline__ void rp_wpool_push(rp_wpool<KeyT>& P, int K, int cap, bool keep, KeyT k, uint32_t pay, int lane) {
  uint64_t m = __ballot(keep);
  if (m == 0) return;
  if (P.n + 64 > cap && P.n > K) {  // cap >= K + 64
    // room for this push and a few more: a trim to at most cap - RP_TRIM_SLACK entries when the keys allow it, an exact cut otherwise
    const rp_wcut<KeyT> t = (cap - RP_TRIM_SLACK > K) ? rp_wpool_trim_impl<KeyT, COL_SHIFT>(P.key, P.pay, P.hist, P.n, K, cap - RP_TRIM_SLACK, lane) : rp_wcut<KeyT>{0, 0, -1};
    if (t.n >= 0) {
      P.n = t.n;
      P.tau = t.tau;
      P.tau_col = t.tau_col;
    } else {
      rp_wpool_cut<KeyT, COL_SHIFT>(P, K, lane);
    }
    keep = keep && rp_wpool_admits<KeyT, COL_SHIFT>(P, k, pay);
    m = __ballot(keep);
    if (m == 0) return;
  }
  if (keep) {
    const int pos = P.n + __popcll(m & ((1ull << lane) - 1ull));
    P.key[pos] = k;
    P.pay[pos] = pay;
  }
  P.n += __popcll(m);
}

// The similarities of one row lie within a few binades, so the sign and most exponent bits of their keys agree and the
// first 11-bit radix digit of the raw key separates nothing.  One pass finds the bits all candidates share (AND and OR
// over the keys); dropping them keeps the order and lets the first digit already resolve mantissa bits.
// red[0..3] = {and_lo, and_hi, or_lo, or_hi}, preset to {~0, ~0, 0, 0} by the caller before the barrier that precedes
// this call.  All threads must call; returns the number of shared leading bits (0..63).
template <int NTHR, class Src>
__device__ int rp_key_prefix(Src& src, int n_slots, uint32_t* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint64_t a = ~0ull, o = 0ull;
  src.sh = 0;
  for (int w = tid; w < n_slots; w += NTHR) {
    if (!src.valid(w)) continue;
    const uint64_t k = src.key(w);
    a &= k;
    o |= k;
  }
  const uint32_t alo = ~rp_wave_or(~(uint32_t)a), ahi = ~rp_wave_or(~(uint32_t)(a >> 32));
  const uint32_t olo = rp_wave_or((uint32_t)o), ohi = rp_wave_or((uint32_t)(o >> 32));
  if (lane == 0) {
    atomicAnd(&red[0], alo);
    atomicAnd(&red[1], ahi);
    atomicOr(&red[2], olo);
    atomicOr(&red[3], ohi);
  }
  __syncthreads();
  const uint64_t A = ((uint64_t)red[1] << 32) | red[0], O = ((uint64_t)red[3] << 32) | red[2];
  const uint64_t d = A ^ O;
  return d ? min(63, (int)__builtin_clzll(d)) : 0;  // all keys equal (or no candidate): nothing to gain
}

// time_aware_item_knn/decay_functions.py:24-252 on one distance d > 0 (N = the largest possible distance)
__device__ __forceinline__ double rp_pair_decay(int kind, double a, double N, double d) {
  switch (kind) {
    case RPHIP_DECAY_EXPONENTIAL: return exp(-a * d);
    case RPHIP_DECAY_CONVEX: return pow(a, d);
    case RPHIP_DECAY_CONCAVE: return 1.0 - pow(a, 1.0 - (d / N));
    case RPHIP_DECAY_LOG: return log(((a - 1.0) * (1.0 - d / N)) + 1.0) / log(a);
    case RPHIP_DECAY_LINEAR: {
      const double r = 1.0 - (d / N) * a;
      return r < 0.0 ? 0.0 : r;
    }
    case RPHIP_DECAY_INVERSE: return 1.0 / d;
    default: return 1.0;  // RPHIP_DECAY_NONE
  }
}


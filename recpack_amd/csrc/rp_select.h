// Exact per-row top-K selection inside one workgroup (K3 / the fused top-K of K8, K9).
//
// Replaces get_top_K_ranks / get_top_K_values (recpack/util.py:50-96): keep the K largest candidates of
// a row.  Order: larger key first, then SMALLER column (the reference's tie order is an artefact of
// numpy's introselect).  The candidates live in `n_slots` slots of an LDS tile (dense
// accumulators) or of a compact list; a slot is described by the Src functor:
//     bool     valid(int slot)   slot holds a candidate
//     uint64_t key(int slot)     order-preserving 64-bit key (larger = better)
//     uint32_t col(int slot)     column id (unique inside the row)
// Slots must be in increasing column order; the selected entries are emitted in slot (= column) order,
// so the output rows are canonical CSR rows.
//
// Method: MSD radix select on the composite (key, ~col), 12 bits per level, histogram in LDS; as soon as
// the undecided group fits RP_SORT_CAP entries it is sorted exactly (bitonic) to read off the threshold.
// Then one predicate pass builds a 64-slot bitmask per wave-step, a 16-entry scan orders the waves and a
// last pass emits.  Every candidate is classified exactly - no sampling, no tolerance.
#pragma once

#include "rp_common.h"

struct rp_select_smem {
  uint32_t* hist;    // RP_HIST_BINS u32, overlays skey/scol
  uint64_t* skey;    // RP_SORT_CAP
  uint32_t* scol;    // RP_SORT_CAP
  uint64_t* mask;    // ceil(n_slots / 64)
  int* misc;         // 64 ints
};

// bytes of LDS the selection needs besides the tile itself
__host__ __device__ constexpr size_t rp_select_smem_bytes(int n_slots) {
  return (size_t)RP_SORT_CAP * 12 + (size_t)((n_slots + 63) / 64) * 8 + 64 * 4;
}

__device__ __forceinline__ rp_select_smem rp_select_carve(char* base, int n_slots) {
  rp_select_smem s;
  s.skey = reinterpret_cast<uint64_t*>(base);
  s.scol = reinterpret_cast<uint32_t*>(base + (size_t)RP_SORT_CAP * 8);
  s.hist = reinterpret_cast<uint32_t*>(base);  // 16 KiB <= 24 KiB
  s.mask = reinterpret_cast<uint64_t*>(base + (size_t)RP_SORT_CAP * 12);
  s.misc = reinterpret_cast<int*>(base + (size_t)RP_SORT_CAP * 12 + (size_t)((n_slots + 63) / 64) * 8);
  return s;
}

// level table: 6 levels over the key (12,12,12,12,12,4 bits from the top), 3 over ~col (12,12,8)
__device__ __forceinline__ void rp_level(int level, int* shift, int* width, bool* is_col) {
  if (level < 5) {
    *shift = 52 - 12 * level;
    *width = 12;
    *is_col = false;
  } else if (level == 5) {
    *shift = 0;
    *width = 4;
    *is_col = false;
  } else if (level == 6) {
    *shift = 20;
    *width = 12;
    *is_col = true;
  } else if (level == 7) {
    *shift = 8;
    *width = 12;
    *is_col = true;
  } else {
    *shift = 0;
    *width = 8;
    *is_col = true;
  }
}

__device__ __forceinline__ int rp_wave_sum(int v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// misc layout
enum { RPS_NVALID = 0, RPS_BIN = 1, RPS_ABOVE = 2, RPS_CNT = 3, RPS_GROUP = 4, RPS_WCNT = 16 /* 16..31 */, RPS_RED = 32 /* 32..47 */, RPS_USER = 48 /* 48..63: caller's */ };

// Returns K' (number of selected entries).  After the call, s.mask holds one bit per slot (1 = selected)
// and s.misc[RPS_WCNT + w] the exclusive offset of wave w's first slot.  Wave w owns the slots
// [w*per_wave, (w+1)*per_wave), per_wave = ceil(n_slots / nwaves / 64) * 64.
template <int NTHR, class Src>
__device__ int rp_select_topk(const Src& src, int n_slots, int K, rp_select_smem s) {
  static_assert(RP_HIST_BINS % NTHR == 0, "bins must divide over the threads");
  constexpr int BPT = RP_HIST_BINS / NTHR;  // histogram bins owned by one thread
  const int tid = threadIdx.x, nthr = NTHR;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
  const int per_wave = ((n_slots + nwaves * 64 - 1) / (nwaves * 64)) * 64;
  const int w_begin = wave * per_wave;
  const int w_end = min(n_slots, w_begin + per_wave);

  // ---- count candidates
  int cnt = 0;
  for (int b = w_begin; b < w_end; b += 64) {
    int slot = b + lane;
    bool v = (slot < w_end) && src.valid(slot);
    cnt += v ? 1 : 0;
  }
  cnt = rp_wave_sum(cnt);
  if (lane == 0) s.misc[RPS_RED + wave] = cnt;
  __syncthreads();
  int n_valid = 0;
  for (int w = 0; w < nwaves; ++w) n_valid += s.misc[RPS_RED + w];
  const int kprime = min(K, n_valid);
  __syncthreads();

  uint64_t tkey = 0;   // threshold composite: selected <=> (key, ~col) >= (tkey, tcol)
  uint32_t tcol = 0;
  if (kprime < n_valid) {
    uint64_t pkey = 0;  // decided prefix
    uint32_t pcol = 0;
    int need = kprime;
    for (int level = 0; level < 9; ++level) {
      int shift, width;
      bool is_col;
      rp_level(level, &shift, &width, &is_col);
      const int nb = 1 << width;
      for (int i = tid; i < RP_HIST_BINS; i += nthr) s.hist[i] = 0;
      if (tid == 0) {
        s.misc[RPS_BIN] = -1;
        s.misc[RPS_CNT] = 0;
      }
      __syncthreads();
      // histogram of this level's digit over the undecided group
      for (int b = w_begin; b < w_end; b += 64) {
        int slot = b + lane;
        if (slot < w_end && src.valid(slot)) {
          uint64_t k = src.key(slot);
          uint32_t ic = ~src.col(slot);
          bool in_group;
          uint32_t d;
          if (!is_col) {
            in_group = (level == 0) || ((k >> (shift + width)) == (pkey >> (shift + width)));
            d = (uint32_t)(k >> shift) & (nb - 1);
          } else {
            in_group = (k == pkey) && ((level == 6) || ((ic >> (shift + width)) == (pcol >> (shift + width))));
            d = (ic >> shift) & (nb - 1);
          }
          if (in_group) atomicAdd(&s.hist[d], 1u);
        }
      }
      __syncthreads();
      // find the bin where the count from the top reaches `need`: thread t owns BPT bins, descending
      {
        int d_hi = RP_HIST_BINS - 1 - BPT * tid;
        int h[BPT];
        int sum = 0;
#pragma unroll
        for (int e = 0; e < BPT; ++e) {
          int d = d_hi - e;
          h[e] = (d < nb) ? (int)s.hist[d] : 0;
          sum += h[e];
        }
        // exclusive prefix over threads (= entries in higher bins)
        int inc = sum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
          int t = __shfl_up(inc, dd);
          if (lane >= dd) inc += t;
        }
        if (lane == 63) s.misc[RPS_RED + wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += s.misc[RPS_RED + w];
        int above = woff + inc - sum;
#pragma unroll
        for (int e = 0; e < BPT; ++e) {
          if (above < need && need <= above + h[e]) {
            s.misc[RPS_BIN] = d_hi - e;
            s.misc[RPS_ABOVE] = above;
            s.misc[RPS_GROUP] = h[e];
          }
          above += h[e];
        }
        __syncthreads();
      }
      const int bin = s.misc[RPS_BIN];
      const int above = s.misc[RPS_ABOVE];
      const int group = s.misc[RPS_GROUP];
      need -= above;
      if (!is_col)
        pkey |= (uint64_t)bin << shift;
      else
        pcol |= (uint32_t)bin << shift;
      if (group == need) {  // the whole group is selected: threshold = smallest composite with this prefix
        tkey = pkey;
        tcol = pcol;
        break;
      }
      if (group <= RP_SORT_CAP) {
        // exact: gather the group, sort descending, read the need-th
        __syncthreads();  // hist is dead from here (skey overlays it)
        for (int b = w_begin; b < w_end; b += 64) {
          int slot = b + lane;
          if (slot < w_end && src.valid(slot)) {
            uint64_t k = src.key(slot);
            uint32_t ic = ~src.col(slot);
            bool in_group;
            if (!is_col)
              in_group = (k >> shift) == (pkey >> shift);
            else
              in_group = (k == pkey) && ((ic >> shift) == (pcol >> shift));
            if (in_group) {
              int p = atomicAdd(&s.misc[RPS_CNT], 1);
              s.skey[p] = k;
              s.scol[p] = ic;
            }
          }
        }
        __syncthreads();
        int n2 = 1;
        while (n2 < group) n2 <<= 1;
        for (int i = group + tid; i < n2; i += nthr) {
          s.skey[i] = 0;
          s.scol[i] = 0;
        }
        __syncthreads();
        for (int k2 = 2; k2 <= n2; k2 <<= 1) {
          for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += nthr) {
              int ixj = i ^ j;
              if (ixj > i) {
                uint64_t ka = s.skey[i], kb = s.skey[ixj];
                uint32_t ca = s.scol[i], cb = s.scol[ixj];
                bool a_lt_b = (ka < kb) || (ka == kb && ca < cb);
                bool desc = ((i & k2) == 0);  // descending in the first half
                if (desc ? a_lt_b : !a_lt_b) {
                  s.skey[i] = kb;
                  s.skey[ixj] = ka;
                  s.scol[i] = cb;
                  s.scol[ixj] = ca;
                }
              }
            }
            __syncthreads();
          }
        }
        tkey = s.skey[need - 1];
        tcol = s.scol[need - 1];
        __syncthreads();
        break;
      }
      __syncthreads();
    }
  }

  // ---- predicate pass: bitmask + per-wave counts
  int wc = 0;
  for (int b = w_begin; b < w_begin + per_wave; b += 64) {
    int slot = b + lane;
    bool sel = false;
    if (slot < w_end && src.valid(slot)) {
      if (kprime == n_valid) {
        sel = true;
      } else {
        uint64_t k = src.key(slot);
        uint32_t ic = ~src.col(slot);
        sel = (k > tkey) || (k == tkey && ic >= tcol);
      }
    }
    uint64_t m = __ballot(sel);
    if (b < n_slots && lane == 0) s.mask[b >> 6] = m;
    wc += __popcll(m);
  }
  if (lane == 0) s.misc[RPS_RED + wave] = wc;
  __syncthreads();
  if (tid == 0) {
    int off = 0;
    for (int w = 0; w < nwaves; ++w) {
      s.misc[RPS_WCNT + w] = off;
      off += s.misc[RPS_RED + w];
    }
    s.misc[RPS_NVALID] = off;
  }
  __syncthreads();
  return kprime;
}

// Emit the selected slots in column order: emit(pos, slot).
template <class Emit>
__device__ __forceinline__ void rp_select_emit(int n_slots, rp_select_smem s, Emit emit) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
  const int per_wave = ((n_slots + nwaves * 64 - 1) / (nwaves * 64)) * 64;
  const int w_begin = wave * per_wave;
  int off = s.misc[RPS_WCNT + wave];
  for (int b = w_begin; b < w_begin + per_wave && b < n_slots; b += 64) {
    uint64_t m = s.mask[b >> 6];
    if ((m >> lane) & 1ull) {
      int pos = off + __popcll(m & ((1ull << lane) - 1ull));
      emit(pos, b + lane);
    }
    off += __popcll(m);
  }
}

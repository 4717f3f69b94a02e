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
// Method
//   1. compaction: every wave scans its contiguous share of the slots ONCE and writes the valid slot ids
//      (u16) to its segment of an LDS list - all later passes touch candidates only (rows are ~5-10 % dense);
//      with PRE the caller has already filled the segments (first-touch lists built while accumulating, in any
//      order) and the results are emitted through rp_select_emit_sorted, which restores the column order;
//   2. MSD radix select on the composite (key, ~col), 11 bits per level, histogram in LDS, until the boundary bin
//      holds at most RP_SORT_CAP candidates (or is selected as a whole);
//   3. ONE classification pass: candidates above the boundary bin are flagged (bit 15 of the list entry), members
//      of the bin are gathered with their list positions and ordered exactly (ranking in registers for <= 64,
//      bitonic sort otherwise): the best of them are flagged too; a scan over the waves' counts orders the emit.
// Every candidate is classified exactly - no sampling, no tolerance.
#pragma once
#include <type_traits>
#include <utility>

#include "rp_common.h"

struct rp_select_smem {
  uint32_t* hist;  // RP_HIST_BINS u32, overlays skey/scol
  uint64_t* skey;  // RP_SORT_CAP
  uint32_t* scol;  // RP_SORT_CAP
  uint16_t* list;  // rp_list_cap(n_slots) entries: wave w owns [w*per_wave, (w+1)*per_wave)
  int* misc;       // RPS_MISC_INTS ints
};

constexpr int RPS_MISC_INTS = 96;
constexpr int RPS_FLAG = 0x8000;
constexpr int RPS_EMPTY = 0x7fff;  // PRE lists: unused entry of a partly filled block (no slot is that large)

// Position of a wave's e-th list entry.  Scan-built lists: wave w owns [w*seg, (w+1)*seg).  PRE (first-touch) lists
// are a shared pool of 64-entry blocks handed out while accumulating; for the selection wave w takes blocks
// w, w+NW, w+2NW, .. whoever filled them, so the passes stay balanced.
template <int NW, bool PRE>
__device__ __forceinline__ int rp_list_pos(int wave, int seg, int e) {
  return PRE ? ((((e >> 6) * NW + wave) << 6) | (e & 63)) : wave * seg + e;
}

__host__ __device__ constexpr int rp_per_wave(int n_slots, int nwaves) {
  return ((n_slots + nwaves * 64 - 1) / (nwaves * 64)) * 64;
}
__host__ __device__ constexpr int rp_list_cap(int n_slots, int nwaves) { return rp_per_wave(n_slots, nwaves) * nwaves; }

// bytes of LDS the selection needs besides the tile itself (nwaves = threads / 64).  seg_cap = list entries per
// wave; 0 = enough for every slot of the wave's share (no overflow possible).
__host__ __device__ constexpr size_t rp_select_smem_bytes(int n_slots, int nwaves = 16, int seg_cap = 0, int bins = RP_HIST_BINS) {
  return (size_t)bins * 4 + (size_t)(seg_cap ? seg_cap * nwaves : rp_list_cap(n_slots, nwaves)) * 2 + RPS_MISC_INTS * 4;
}

// `bins` = histogram bins of the radix levels (rp_select_topk's 1 << BITS); the boundary bin is ordered exactly once it
// holds at most bins / 4 candidates, whose (key, ~col, list position) overlay the histogram
__device__ __forceinline__ rp_select_smem rp_select_carve(char* base, int n_slots, int nwaves = 16, int seg_cap = 0,
                                                          int bins = RP_HIST_BINS) {
  const size_t list_entries = seg_cap ? (size_t)seg_cap * nwaves : (size_t)rp_list_cap(n_slots, nwaves);
  rp_select_smem s;
  s.hist = reinterpret_cast<uint32_t*>(base);
  s.skey = reinterpret_cast<uint64_t*>(base);
  s.scol = reinterpret_cast<uint32_t*>(base + (size_t)(bins / 4) * 8);
  s.list = reinterpret_cast<uint16_t*>(base + (size_t)bins * 4);
  s.misc = reinterpret_cast<int*>(base + (size_t)bins * 4 + list_entries * 2);
  return s;
}

// level table for BITS-bit digits: ceil(64 / BITS) levels over the key from the top, then ceil(32 / BITS) over ~col
// (BITS = 11: 11,11,11,11,11,9 and 11,11,10)
template <int BITS = 11>
__device__ __forceinline__ void rp_level(int level, int* shift, int* width, bool* is_col) {
  constexpr int NK = (64 + BITS - 1) / BITS, NC = (32 + BITS - 1) / BITS;
  if (level < NK) {
    *is_col = false;
    *shift = (level < NK - 1) ? 64 - BITS * (level + 1) : 0;
    *width = (level < NK - 1) ? BITS : 64 - BITS * (NK - 1);
  } else {
    const int l = level - NK;
    *is_col = true;
    *shift = (l < NC - 1) ? 32 - BITS * (l + 1) : 0;
    *width = (l < NC - 1) ? BITS : 32 - BITS * (NC - 1);
  }
}

__device__ __forceinline__ int rp_wave_sum(int v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// misc layout (ints)
enum {
  RPS_NSEL = 0,
  RPS_BIN = 1,
  RPS_ABOVE = 2,
  RPS_CNT = 3,
  RPS_GROUP = 4,
  RPS_TK = 8,     // 8..10: threshold handed from the ranking wave to the workgroup
  RPS_WOFF = 16,  // 16..31: exclusive offset of wave w's selected entries
  RPS_RED = 32,   // 32..47: scratch for reductions over waves
  RPS_LCNT = 48,  // 48..63: candidates in wave w's list segment
  RPS_USER = 64   // 64..95: the caller's
};

// Returns K' (number of selected entries), or -1 when a wave found more candidates than its list segment holds
// (seg_cap > 0 only; nothing is selected then and the caller must clear the tile densely).  PRE: the caller filled
// `pre_blocks` 64-entry blocks of the list itself (unused entries = RPS_EMPTY) and passes the number of candidates
// its wave appended (pre_cnt) and whether the pool overflowed (pre_over).  Afterwards the list holds the candidates
// (RPS_LCNT + w entries to visit per wave), selected ones flagged with RPS_FLAG; use rp_select_emit /
// rp_select_emit_sorted / rp_select_foreach with the same seg_cap.
// NCACHE: the (slot, key, ~column) of a lane's first 4 * NCACHE list entries stay in registers from the first radix level on - a key
// costs a global gather (the column weight), and the levels and the classification pass would each evaluate it again.
// a selection source whose key is (slot-local data) x (a weight gathered from global memory) may expose the two halves apart:
//   double colw(slot)            the gathered weight (any slot < n_slots is a valid address, wanted or not)
//   uint64_t key_w(slot, colw)   the key from it
template <class S, class = void>
struct rp_src_has_colw : std::false_type {};
template <class S>
struct rp_src_has_colw<S, std::void_t<decltype(std::declval<const S&>().colw(0))>> : std::true_type {};

template <int NTHR, class Src, bool PRE = false, int BITS = 11, int NCACHE = 0>
__device__ int rp_select_topk(const Src& src, int n_slots, int K, rp_select_smem s, int seg_cap = 0, int pre_cnt = 0,
                              bool pre_over = false, int pre_blocks = 0) {
  constexpr int BINS = 1 << BITS, SCAP = BINS / 4;  // histogram bins; largest boundary bin that is ordered exactly
  constexpr int NK = (64 + BITS - 1) / BITS, NLEV = NK + (32 + BITS - 1) / BITS;
  static_assert(BINS % NTHR == 0, "bins must divide over the threads");
  static_assert(SCAP * 14 <= BINS * 4, "sort buffers (key, ~col, list position) overlay the histogram");
  constexpr int BPT = BINS / NTHR;  // histogram bins owned by one thread
  constexpr int NW = NTHR / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int per_wave = rp_per_wave(n_slots, NW);
  const int w_begin = wave * per_wave;
  const int w_end = min(n_slots, w_begin + per_wave);
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const int seg = seg_cap ? seg_cap : per_wave;
  auto lpos = [&](int e) { return rp_list_pos<NW, PRE>(wave, seg, e); };

  // ---- 1. compaction
  int cnt = 0;       // list entries this wave visits
  int n_mine = 0;    // candidates among them
  bool over = false;
  if constexpr (PRE) {
    cnt = ((pre_blocks - wave + NW - 1) / NW) * 64;
    n_mine = pre_cnt;  // what this wave appended: the sum over the waves is what counts
    over = pre_over;
  } else {
    for (int b = w_begin; b < w_end; b += 64) {
      const int slot = b + lane;
      const bool v = (slot < w_end) && src.valid(slot);
      const uint64_t m = __ballot(v);
      const int add = __popcll(m);
      if (cnt + add > seg) {
        over = true;
        break;
      }
      if (v) s.list[lpos(cnt + __popcll(m & lt_mask))] = (uint16_t)slot;
      cnt += add;
    }
    n_mine = cnt;
  }
  if (lane == 0) {
    s.misc[RPS_RED + wave] = over ? -1 : n_mine;
    s.misc[RPS_LCNT + wave] = over ? 0 : cnt;
  }
  __syncthreads();
  int n_valid = 0;
  bool any_over = false;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int c = s.misc[RPS_RED + w];
    any_over |= (c < 0);
    n_valid += c;
  }
  if (any_over) {
    __syncthreads();
    return -1;
  }
  const int kprime = min(K, n_valid);
  __syncthreads();

  // outcome of the radix levels: the boundary bin (prefix fin_pkey/fin_pcol down to fin_shift) holds fin_group
  // candidates of which fin_need are selected
  bool have_fin = false, fin_is_col = false, fin_all = false;
  int fin_shift = 0, fin_group = 0, fin_need = 0;
  uint64_t fin_pkey = 0;
  uint32_t fin_pcol = 0;
  // the cached entries (NCACHE > 0): filled once, before the first level
  int c_sl[NCACHE > 0 ? NCACHE : 1][4];
  bool c_ok[NCACHE > 0 ? NCACHE : 1][4];
  uint64_t c_k[NCACHE > 0 ? NCACHE : 1][4];
  uint32_t c_ic[NCACHE > 0 ? NCACHE : 1][4];
  // four list entries of this lane at e0 .. e0 + 192: evaluated
  auto entries4 = [&](int e0, int(&sl)[4], bool(&ok)[4], uint64_t(&kk)[4], uint32_t(&ic)[4], bool want_key) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 64 + lane;
      sl[u] = (e < cnt) ? (int)s.list[lpos(e)] : RPS_EMPTY;
      ok[u] = PRE ? sl[u] != RPS_EMPTY : e < cnt;
    }
    if constexpr (rp_src_has_colw<Src>::value) {
      // (round 6) the key's global operand - the column weight - of all four entries is REQUESTED before the first key is formed, and
      // unconditionally (a lane without an entry reads slot 0): with the load inside `ok ? key(slot) : 0` every gather sat behind its
      // own branch and the compiler put s_waitcnt vmcnt(0) in front of each use - eight dependent L2 round trips per thread in the
      // light rows' selection, a third of that kernel
      double cw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) cw[u] = src.colw(ok[u] ? sl[u] : 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) kk[u] = (ok[u] && want_key) ? src.key_w(sl[u], cw[u]) : 0ull;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) kk[u] = (ok[u] && want_key) ? src.key(sl[u]) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) ic[u] = ok[u] ? ~src.col(sl[u]) : 0u;
  };
  if (kprime < n_valid) {
    if constexpr (NCACHE > 0) {
#pragma unroll
      for (int it = 0; it < NCACHE; ++it) entries4(it * 256, c_sl[it], c_ok[it], c_k[it], c_ic[it], true);
    }
    uint64_t pkey = 0;  // decided prefix
    uint32_t pcol = 0;
    int need = kprime;
    for (int level = 0; level < NLEV; ++level) {
      int shift, width;
      bool is_col;
      rp_level<BITS>(level, &shift, &width, &is_col);
      const int nb = 1 << width;
      for (int i = tid; i < BINS; i += NTHR) s.hist[i] = 0;
      if (tid == 0) {
        s.misc[RPS_BIN] = -1;
        s.misc[RPS_CNT] = 0;
      }
      __syncthreads();
      // histogram of this level's digit over the undecided group (4 candidates per lane in flight:
      // the key needs a global gather)
      auto hist4 = [&](const bool(&ok)[4], const uint64_t(&kk)[4], const uint32_t(&icv)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
          const uint64_t k = kk[u];
          const uint32_t ic = icv[u];
          bool in_group;
          uint32_t d;
          if (!is_col) {
            in_group = (level == 0) || ((k >> (shift + width)) == (pkey >> (shift + width)));
            d = (uint32_t)(k >> shift) & (nb - 1);
          } else {
            in_group = (k == pkey) && ((level == NK) || ((ic >> (shift + width)) == (pcol >> (shift + width))));
            d = (ic >> shift) & (nb - 1);
          }
          if (in_group) atomicAdd(&s.hist[d], 1u);
        }
      };
      if constexpr (NCACHE > 0) {
#pragma unroll
        for (int it = 0; it < NCACHE; ++it)
          if (it * 256 < cnt) hist4(c_ok[it], c_k[it], c_ic[it]);
      }
      for (int e0 = NCACHE * 256; e0 < cnt; e0 += 256) {
        int sl[4];
        bool ok[4];
        uint64_t kk[4];
        uint32_t ic[4];
        entries4(e0, sl, ok, kk, ic, true);
        hist4(ok, kk, ic);
      }
      __syncthreads();
      // find the bin where the count from the top reaches `need`: thread t owns BPT bins, descending
      {
        const int d_hi = BINS - 1 - BPT * tid;
        int h[BPT];
        int sum = 0;
#pragma unroll
        for (int e = 0; e < BPT; ++e) {
          const int d = d_hi - e;
          h[e] = (d < nb) ? (int)s.hist[d] : 0;
          sum += h[e];
        }
        int inc = sum;  // inclusive prefix over the wave (= entries in this and higher bins)
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
          const int t = __shfl_up(inc, dd);
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
      if (group == need || group <= SCAP) {
        // decided: everything above the boundary bin is selected; the bin itself entirely (group == need) or after an
        // exact ordering of its members (fused pass below)
        fin_shift = shift;
        fin_is_col = is_col;
        fin_all = (group == need);
        fin_group = group;
        fin_need = need;
        fin_pkey = pkey;
        fin_pcol = pcol;
        have_fin = true;
        __syncthreads();  // hist is dead from here (the sort buffers overlay it)
        break;
      }
      __syncthreads();
    }
  }

  // ---- 3. fused classification pass: one key evaluation per candidate.  Above the boundary bin -> flagged now;
  // members of the bin -> flagged (whole bin selected) or gathered with their list position for the exact step.
  int wc = 0;
  uint16_t* spos = reinterpret_cast<uint16_t*>(s.scol + SCAP);  // [SCAP] list positions of the members
  auto classify4 = [&](int e0, const int(&sl)[4], const bool(&ok)[4], const uint64_t(&kk)[4], const uint32_t(&icv)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bool sel = false;
      if (ok[u]) {
        if (!have_fin) {
          sel = true;  // fewer candidates than K: all of them
        } else {
          const uint64_t k = kk[u];
          const uint32_t ic = icv[u];
          bool above, member;
          if (!fin_is_col) {
            above = (k >> fin_shift) > (fin_pkey >> fin_shift);
            member = (k >> fin_shift) == (fin_pkey >> fin_shift);
          } else {
            above = (k > fin_pkey) || (k == fin_pkey && (ic >> fin_shift) > (fin_pcol >> fin_shift));
            member = (k == fin_pkey) && ((ic >> fin_shift) == (fin_pcol >> fin_shift));
          }
          sel = above || (member && fin_all);
          if (member && !fin_all) {
            const int p = atomicAdd(&s.misc[RPS_CNT], 1);
            s.skey[p] = k;
            s.scol[p] = ic;
            spos[p] = (uint16_t)lpos(e0 + u * 64 + lane);
          }
        }
        if (sel) s.list[lpos(e0 + u * 64 + lane)] = (uint16_t)(sl[u] | RPS_FLAG);
      }
      wc += __popcll(__ballot(sel));
    }
  };
  if constexpr (NCACHE > 0) {
    if (have_fin) {  // (the cache was filled)
#pragma unroll
      for (int it = 0; it < NCACHE; ++it)
        if (it * 256 < cnt) classify4(it * 256, c_sl[it], c_ok[it], c_k[it], c_ic[it]);
    }
  }
  for (int e0 = (NCACHE > 0 && have_fin) ? NCACHE * 256 : 0; e0 < cnt; e0 += 256) {
    int sl[4];
    bool ok[4];
    uint64_t kk[4];
    uint32_t ic[4];
    entries4(e0, sl, ok, kk, ic, have_fin);
    classify4(e0, sl, ok, kk, ic);
  }
  if (lane == 0) s.misc[RPS_RED + wave] = wc;
  __syncthreads();
  if (have_fin && !fin_all) {
    // ---- 4. exact order of the boundary bin's members: the fin_need best of them are selected too
    const int group = fin_group, need = fin_need;
    auto owner = [&](int pos) { return PRE ? ((pos >> 6) % NW) : (pos / seg); };
    if (group <= 64) {
      if (wave == 0) {  // one wave ranks the members in registers
        uint64_t mk = 0;
        uint32_t mic = 0;
        if (lane < group) {
          mk = s.skey[lane];
          mic = s.scol[lane];
        }
        int rank = 0;
        for (int m2 = 0; m2 < group; ++m2) {
          const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mk, m2);
          const uint32_t ohi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mk >> 32), m2);
          const uint32_t oic = (uint32_t)__builtin_amdgcn_readlane((int)mic, m2);
          const uint64_t ok2 = ((uint64_t)ohi << 32) | olo;
          rank += ((ok2 > mk) || (ok2 == mk && oic > mic)) ? 1 : 0;
        }
        if (lane < group && rank < need) {
          const int pos = spos[lane];
          s.list[pos] = (uint16_t)(s.list[pos] | RPS_FLAG);
          atomicAdd(&s.misc[RPS_RED + owner(pos)], 1);
        }
      }
    } else {
      int n2 = 1;
      while (n2 < group) n2 <<= 1;
      for (int i = group + tid; i < n2; i += NTHR) {
        s.skey[i] = 0;
        s.scol[i] = 0;
        spos[i] = 0;
      }
      __syncthreads();
      for (int k2 = 2; k2 <= n2; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < n2; i += NTHR) {
            const int ixj = i ^ j;
            if (ixj > i) {
              const uint64_t ka = s.skey[i], kb = s.skey[ixj];
              const uint32_t ca = s.scol[i], cb = s.scol[ixj];
              const bool a_lt_b = (ka < kb) || (ka == kb && ca < cb);
              const bool desc = ((i & k2) == 0);  // descending in the first half
              if (desc ? a_lt_b : !a_lt_b) {
                const uint16_t pa = spos[i], pb = spos[ixj];
                s.skey[i] = kb;
                s.skey[ixj] = ka;
                s.scol[i] = cb;
                s.scol[ixj] = ca;
                spos[i] = pb;
                spos[ixj] = pa;
              }
            }
          }
          __syncthreads();
        }
      }
      for (int i = tid; i < need; i += NTHR) {
        const int pos = spos[i];
        s.list[pos] = (uint16_t)(s.list[pos] | RPS_FLAG);
        atomicAdd(&s.misc[RPS_RED + owner(pos)], 1);
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    int off = 0;
    for (int w = 0; w < NW; ++w) {
      s.misc[RPS_WOFF + w] = off;
      off += s.misc[RPS_RED + w];
    }
    s.misc[RPS_NSEL] = off;
  }
  __syncthreads();
  return kprime;
}

// Emit the selected slots in column order: emit(pos, slot).
template <int NTHR, class Emit>
__device__ __forceinline__ void rp_select_emit(int n_slots, rp_select_smem s, Emit emit, int seg_cap = 0) {
  constexpr int NW = NTHR / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint16_t* mylist = s.list + wave * (seg_cap ? seg_cap : rp_per_wave(n_slots, NW));
  const int cnt = s.misc[RPS_LCNT + wave];
  int off = s.misc[RPS_WOFF + wave];
  for (int e0 = 0; e0 < cnt; e0 += 64) {
    const int e = e0 + lane;
    const int v = (e < cnt) ? (int)mylist[e] : 0;
    const bool sel = (v & RPS_FLAG) != 0;
    const uint64_t m = __ballot(sel);
    if (sel) emit(off + __popcll(m & ((1ull << lane) - 1ull)), v & (RPS_FLAG - 1));
    off += __popcll(m);
  }
}

// Emit the selected slots in column order when the list segments are NOT column-ordered (first-touch lists):
// the flagged slots are gathered into a small LDS array (over the dead histogram), every thread ranks its entries
// by counting the smaller slots, emit(rank, slot).  Needs K <= 2 * bins selected entries (u16 each, over the histogram).
template <int NTHR, class Emit>
__device__ __forceinline__ void rp_select_emit_sorted(rp_select_smem s, Emit emit) {
  constexpr int NW = NTHR / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint16_t* buf = reinterpret_cast<uint16_t*>(s.hist);
  const int cnt = s.misc[RPS_LCNT + wave];
  int off = s.misc[RPS_WOFF + wave];
  const int n_sel = s.misc[RPS_NSEL];
  for (int e0 = 0; e0 < cnt; e0 += 64) {
    const int v = (int)s.list[rp_list_pos<NW, true>(wave, 0, e0 + lane)];  // cnt is a multiple of 64
    const bool sel = (v & RPS_FLAG) != 0;
    const uint64_t m = __ballot(sel);
    if (sel) buf[off + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(v & (RPS_FLAG - 1));
    off += __popcll(m);
  }
  __syncthreads();
  const uint2* buf4 = reinterpret_cast<const uint2*>(s.hist);
  for (int e = tid; e < n_sel; e += NTHR) {
    const uint32_t slot = buf[e];
    int rank = 0;
    for (int o = 0; o < n_sel; o += 4) {  // (the words past n_sel hold stale bytes: masked by the index)
      const uint2 q = buf4[o >> 2];
      rank += ((q.x & 0xffffu) < slot ? 1 : 0) + ((q.x >> 16) < slot && o + 1 < n_sel ? 1 : 0) +
              ((q.y & 0xffffu) < slot && o + 2 < n_sel ? 1 : 0) + ((q.y >> 16) < slot && o + 3 < n_sel ? 1 : 0);
    }
    emit(rank, (int)slot);
  }
}

// PRE lists, the same in column order through a BITMAP (one bit per slot, over the dead histogram: n_slots <= 32 * bins): the
// flagged entries set their bits, a scan over the words' popcounts gives every thread the output position of its first bit, and
// the bits are emitted in slot (= column) order.  Three barriers and a few dozen instructions per thread where ranking K slots
// against each other (rp_select_emit_sorted) reads the whole selection once per selected slot: 2.1 of the light kernel's 9 ms.
template <int NTHR, class Emit>
__device__ __forceinline__ void rp_select_emit_bitmap(int n_slots, rp_select_smem s, Emit emit) {
  constexpr int NW = NTHR / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* bm = s.hist;
  const int nw = (n_slots + 31) >> 5;
  for (int i = tid; i < nw; i += NTHR) bm[i] = 0u;
  __syncthreads();
  const int cnt = s.misc[RPS_LCNT + wave];
  for (int e = lane; e < cnt; e += 64) {
    const int v = (int)s.list[rp_list_pos<NW, true>(wave, 0, e)];
    if (v & RPS_FLAG) {
      const int slot = v & (RPS_FLAG - 1);
      atomicOr(&bm[slot >> 5], 1u << (slot & 31));
    }
  }
  __syncthreads();
  const int wpt = (nw + NTHR - 1) / NTHR;  // consecutive words per thread
  const int w0 = tid * wpt, w1 = min(nw, w0 + wpt);
  int mine = 0;
  for (int w = w0; w < w1; ++w) mine += __popc(bm[w]);
  int inc = mine;  // inclusive prefix over the wave
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const int t = __shfl_up(inc, dd);
    if (lane >= dd) inc += t;
  }
  if (lane == 63) s.misc[RPS_RED + wave] = inc;
  __syncthreads();
  int off = inc - mine;
  for (int w2 = 0; w2 < wave; ++w2) off += s.misc[RPS_RED + w2];
  for (int w = w0; w < w1; ++w) {
    uint32_t bits = bm[w];
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1u;
      emit(off++, w * 32 + b);
    }
  }
}

// PRE lists: visit every candidate slot (selected or not).
template <int NTHR, class F>
__device__ __forceinline__ void rp_select_foreach_pre(rp_select_smem s, F f) {
  constexpr int NW = NTHR / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cnt = s.misc[RPS_LCNT + wave];
  for (int e = lane; e < cnt; e += 64) {
    const int v = (int)s.list[rp_list_pos<NW, true>(wave, 0, e)] & (RPS_FLAG - 1);
    if (v != RPS_EMPTY) f(v);
  }
}

// Visit every candidate slot (selected or not): f(slot).  Used to clear the touched accumulators.
template <int NTHR, class F>
__device__ __forceinline__ void rp_select_foreach(int n_slots, rp_select_smem s, F f, int seg_cap = 0) {
  constexpr int NW = NTHR / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint16_t* mylist = s.list + wave * (seg_cap ? seg_cap : rp_per_wave(n_slots, NW));
  const int cnt = s.misc[RPS_LCNT + wave];
  for (int e = lane; e < cnt; e += 64) f((int)(mylist[e] & (RPS_FLAG - 1)));
}

// ---- selection straight from a dense tile (no candidate list) ---------------------------------------------------------
// For rows whose tile is mostly non-zero (heavy item rows) a candidate list is as large as the tile itself and only
// costs LDS.  Here every pass walks the slots (thread t takes slots t, t + NTHR, ..: the key's global operand, e.g.
// 1/c_j, is then read coalesced), selected slots are recorded in a bitmap and emitted in slot (= column) order.
// Same exact rule as rp_select_topk: larger key first, then smaller column.
constexpr int RP_TSEL_LIST = 1024;  // candidates the sampled-threshold pass may collect (key 8 B + slot 2 B each)
struct rp_tsel_smem {
  uint32_t* hist;    // RP_HIST_BINS u32, overlays skey / scol / spos of the boundary bin
  uint64_t* skey;
  uint32_t* scol;
  uint32_t* bitmap;  // (n_slots + 31) / 32 words
  int* misc;         // RPS_MISC_INTS
  uint64_t* lkey;    // [RP_TSEL_LIST] keys of the collected candidates
  uint16_t* lslot;   // [RP_TSEL_LIST] their slots
};

__host__ __device__ constexpr size_t rp_tsel_smem_bytes(int n_slots) {
  return (size_t)RP_HIST_BINS * 4 + (size_t)((n_slots + 31) / 32 + 4) * 4 + RPS_MISC_INTS * 4 + (size_t)RP_TSEL_LIST * 10 + 16;
}

__device__ __forceinline__ rp_tsel_smem rp_tsel_carve(char* base, int n_slots) {
  rp_tsel_smem s;
  s.hist = reinterpret_cast<uint32_t*>(base);
  s.skey = reinterpret_cast<uint64_t*>(base);
  s.scol = reinterpret_cast<uint32_t*>(base + (size_t)RP_SORT_CAP * 8);
  s.bitmap = reinterpret_cast<uint32_t*>(base + (size_t)RP_HIST_BINS * 4);
  s.misc = reinterpret_cast<int*>(base + (size_t)RP_HIST_BINS * 4 + (size_t)((n_slots + 31) / 32 + 4) * 4);
  char* l = base + (size_t)RP_HIST_BINS * 4 + (size_t)((n_slots + 31) / 32 + 4) * 4 + RPS_MISC_INTS * 4;
  l += (8 - (reinterpret_cast<uintptr_t>(l) & 7)) & 7;
  s.lkey = reinterpret_cast<uint64_t*>(l);
  s.lslot = reinterpret_cast<uint16_t*>(l + (size_t)RP_TSEL_LIST * 8);
  return s;
}

// the collected candidates as a selection source: slot p of the list = tile slot lslot[p]
template <class Src>
struct rp_tlist_src {
  const Src* tile;
  const uint64_t* lkey;
  const uint16_t* lslot;
  int n;
  __device__ __forceinline__ bool valid(int p) const { return p < n; }
  __device__ __forceinline__ uint64_t key(int p) const { return lkey[p]; }
  __device__ __forceinline__ uint32_t col(int p) const { return tile->col((int)lslot[p]); }
};

// Exact selection over the slots of `src`: the radix levels and the classification pass each walk ALL slots and evaluate
// every valid slot's key.  mark(slot) records a selected slot.  Returns K'.
template <int NTHR, class Src, class Mark>
__device__ int rp_select_tile_full(const Src& src, int n_slots, int K, rp_tsel_smem s, Mark mark) {
  static_assert(RP_HIST_BINS % NTHR == 0, "bins must divide over the threads");
  constexpr int BPT = RP_HIST_BINS / NTHR;
  constexpr int NW = NTHR / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint16_t* spos = reinterpret_cast<uint16_t*>(s.scol + RP_SORT_CAP);

  bool have_fin = false, fin_is_col = false, fin_all = false;
  int fin_shift = 0, fin_group = 0, fin_need = 0;
  uint64_t fin_pkey = 0;
  uint32_t fin_pcol = 0;
  int n_valid = 0, kprime = 0;
  {
    uint64_t pkey = 0;
    uint32_t pcol = 0;
    int need = 0;
    for (int level = 0; level < 9; ++level) {
      int shift, width;
      bool is_col;
      rp_level(level, &shift, &width, &is_col);
      const int nb = 1 << width;
      for (int i = tid; i < RP_HIST_BINS; i += NTHR) s.hist[i] = 0;
      if (tid == 0) {
        s.misc[RPS_BIN] = -1;
        s.misc[RPS_CNT] = 0;
      }
      __syncthreads();
      int mine = 0;
      for (int w = tid; w < n_slots; w += NTHR) {
        if (!src.valid(w)) continue;
        const uint64_t k = src.key(w);
        const uint32_t ic = ~src.col(w);
        bool in_group;
        uint32_t d;
        if (!is_col) {
          in_group = (level == 0) || ((k >> (shift + width)) == (pkey >> (shift + width)));
          d = (uint32_t)(k >> shift) & (nb - 1);
        } else {
          in_group = (k == pkey) && ((level == 6) || ((ic >> (shift + width)) == (pcol >> (shift + width))));
          d = (ic >> shift) & (nb - 1);
        }
        if (in_group) {
          atomicAdd(&s.hist[d], 1u);
          ++mine;
        }
      }
      if (level == 0) {  // the first pass also counts the candidates
        mine = rp_wave_sum(mine);
        if (lane == 0) s.misc[RPS_RED + wave] = mine;
      }
      __syncthreads();
      if (level == 0) {
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) n_valid += s.misc[RPS_RED + w2];
        kprime = min(K, n_valid);
        need = kprime;
        __syncthreads();
        if (kprime == n_valid) break;  // fewer candidates than K: all of them
      }
      {
        const int d_hi = RP_HIST_BINS - 1 - BPT * tid;
        int h[BPT];
        int sum = 0;
#pragma unroll
        for (int e = 0; e < BPT; ++e) {
          const int d = d_hi - e;
          h[e] = (d < nb) ? (int)s.hist[d] : 0;
          sum += h[e];
        }
        int inc = sum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
          const int t = __shfl_up(inc, dd);
          if (lane >= dd) inc += t;
        }
        if (lane == 63) s.misc[RPS_RED + wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int w2 = 0; w2 < wave; ++w2) woff += s.misc[RPS_RED + w2];
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
      if (group == need || group <= RP_SORT_CAP) {
        fin_shift = shift;
        fin_is_col = is_col;
        fin_all = (group == need);
        fin_group = group;
        fin_need = need;
        fin_pkey = pkey;
        fin_pcol = pcol;
        have_fin = true;
        __syncthreads();  // hist is dead from here
        break;
      }
      __syncthreads();
    }
  }
  // classification pass
  for (int w = tid; w < n_slots; w += NTHR) {
    if (!src.valid(w)) continue;
    bool sel = true;
    if (have_fin) {
      const uint64_t k = src.key(w);
      const uint32_t ic = ~src.col(w);
      bool above, member;
      if (!fin_is_col) {
        above = (k >> fin_shift) > (fin_pkey >> fin_shift);
        member = (k >> fin_shift) == (fin_pkey >> fin_shift);
      } else {
        above = (k > fin_pkey) || (k == fin_pkey && (ic >> fin_shift) > (fin_pcol >> fin_shift));
        member = (k == fin_pkey) && ((ic >> fin_shift) == (fin_pcol >> fin_shift));
      }
      sel = above || (member && fin_all);
      if (member && !fin_all) {
        const int p = atomicAdd(&s.misc[RPS_CNT], 1);
        s.skey[p] = k;
        s.scol[p] = ic;
        spos[p] = (uint16_t)w;
      }
    }
    if (sel) mark(w);
  }
  __syncthreads();
  if (have_fin && !fin_all) {
    const int group = fin_group, need = fin_need;
    if (group <= 64) {
      if (wave == 0) {
        uint64_t mk = 0;
        uint32_t mic = 0;
        if (lane < group) {
          mk = s.skey[lane];
          mic = s.scol[lane];
        }
        int rank = 0;
        for (int m2 = 0; m2 < group; ++m2) {
          const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mk, m2);
          const uint32_t ohi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mk >> 32), m2);
          const uint32_t oic = (uint32_t)__builtin_amdgcn_readlane((int)mic, m2);
          const uint64_t ok2 = ((uint64_t)ohi << 32) | olo;
          rank += ((ok2 > mk) || (ok2 == mk && oic > mic)) ? 1 : 0;
        }
        if (lane < group && rank < need) mark((int)spos[lane]);
      }
    } else {
      int n2 = 1;
      while (n2 < group) n2 <<= 1;
      for (int i = group + tid; i < n2; i += NTHR) {
        s.skey[i] = 0;
        s.scol[i] = 0;
        spos[i] = 0;
      }
      __syncthreads();
      for (int k2 = 2; k2 <= n2; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < n2; i += NTHR) {
            const int ixj = i ^ j;
            if (ixj > i) {
              const uint64_t ka = s.skey[i], kb = s.skey[ixj];
              const uint32_t ca = s.scol[i], cb = s.scol[ixj];
              const bool a_lt_b = (ka < kb) || (ka == kb && ca < cb);
              const bool desc = ((i & k2) == 0);
              if (desc ? a_lt_b : !a_lt_b) {
                const uint16_t pa = spos[i], pb = spos[ixj];
                s.skey[i] = kb;
                s.skey[ixj] = ka;
                s.scol[i] = cb;
                s.scol[ixj] = ca;
                spos[i] = pb;
                spos[ixj] = pa;
              }
            }
          }
          __syncthreads();
        }
      }
      for (int i = tid; i < need; i += NTHR) mark((int)spos[i]);
    }
    __syncthreads();
  }
  return kprime;
}

// Returns K' and leaves the selected slots in s.bitmap (bit w of word w/32).  The bitmap must be all zero on entry;
// rp_select_tile_emit clears it again.
//
// Sampled threshold first: the full routine evaluates every slot's key (an unpack, a gather of the column weight, a float64
// product) once per radix level and once more to classify - three to four times per slot.  Here one slot in eight is
// histogrammed by its first radix digit, the digit d* that holds rank ~ 1.5 K / 8 + 16 of the SAMPLE becomes a lower bound
// for the K-th key of the tile, and ONE pass over all slots collects the slots whose first digit is >= d* (with their keys)
// and counts the valid ones.  If at least K and at most RP_TSEL_LIST slots were collected, the K best of the tile are
// among them by construction and the exact selection runs on that short list with cached keys (the same routine, a list
// source).  Otherwise (a sample that misled, a crowded digit) the full routine runs: exactness never depends on the sample.
template <int NTHR, class Src>
__device__ int rp_select_tile(const Src& src, int n_slots, int K, rp_tsel_smem s) {
  constexpr int SMP = 8;
  constexpr int NW = NTHR / 64;
  constexpr int BPT = RP_HIST_BINS / NTHR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto mark_tile = [&](int w) { atomicOr(&s.bitmap[w >> 5], 1u << (w & 31)); };
  int shift0, width0;
  bool is_col0;
  rp_level(0, &shift0, &width0, &is_col0);
  // ---- sample
  for (int i = tid; i < RP_HIST_BINS; i += NTHR) s.hist[i] = 0;
  if (tid == 0) {
    s.misc[RPS_BIN] = -1;
    s.misc[RPS_CNT] = 0;
  }
  __syncthreads();
  for (int w = tid * SMP; w < n_slots; w += NTHR * SMP) {
    if (!src.valid(w)) continue;
    atomicAdd(&s.hist[(uint32_t)(src.key(w) >> shift0) & ((1u << width0) - 1u)], 1u);
  }
  __syncthreads();
  {
    const int want = (3 * K) / (2 * SMP) + 16;  // sample rank whose digit becomes the bound
    const int d_hi = RP_HIST_BINS - 1 - BPT * tid;
    int h[BPT];
    int sum = 0;
#pragma unroll
    for (int e = 0; e < BPT; ++e) {
      h[e] = (int)s.hist[d_hi - e];
      sum += h[e];
    }
    int inc = sum;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const int t = __shfl_up(inc, dd);
      if (lane >= dd) inc += t;
    }
    if (lane == 63) s.misc[RPS_RED + wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w2 = 0; w2 < wave; ++w2) woff += s.misc[RPS_RED + w2];
    int above = woff + inc - sum;
#pragma unroll
    for (int e = 0; e < BPT; ++e) {
      if (above < want && want <= above + h[e]) s.misc[RPS_BIN] = d_hi - e;
      above += h[e];
    }
    __syncthreads();
  }
  const int dstar = s.misc[RPS_BIN];  // -1: the sample holds fewer than `want` candidates - collect everything
  // ---- collect
  int mine = 0;
  for (int w = tid; w < n_slots; w += NTHR) {
    if (!src.valid(w)) continue;
    ++mine;
    const uint64_t k = src.key(w);
    if ((int)((uint32_t)(k >> shift0) & ((1u << width0) - 1u)) >= dstar) {
      const int p = atomicAdd(&s.misc[RPS_CNT], 1);
      if (p < RP_TSEL_LIST) {
        s.lkey[p] = k;
        s.lslot[p] = (uint16_t)w;
      }
    }
  }
  mine = rp_wave_sum(mine);
  if (lane == 0) s.misc[RPS_RED + wave] = mine;
  __syncthreads();
  int n_valid = 0;
#pragma unroll
  for (int w2 = 0; w2 < NW; ++w2) n_valid += s.misc[RPS_RED + w2];
  const int n_list = s.misc[RPS_CNT];
  __syncthreads();
  if (n_valid <= K) {  // fewer candidates than K: all of them
    for (int w = tid; w < n_slots; w += NTHR)
      if (src.valid(w)) mark_tile(w);
    __syncthreads();
    return n_valid;
  }
  if (n_list >= K && n_list <= RP_TSEL_LIST) {
    rp_tlist_src<Src> ls{&src, s.lkey, s.lslot, n_list};
    const uint16_t* lslot = s.lslot;
    const int kp = rp_select_tile_full<NTHR>(ls, n_list, K, s, [&](int p) { mark_tile((int)lslot[p]); });
    return kp;
  }
  const int kp = rp_select_tile_full<NTHR>(src, n_slots, K, s, mark_tile);
  return kp;
}

// emit(pos, slot) for every selected slot in slot order; clears the bitmap
template <int NTHR, class Emit>
__device__ __forceinline__ void rp_select_tile_emit(int n_slots, rp_tsel_smem s, Emit emit) {
  constexpr int NW = NTHR / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_words = (n_slots + 31) >> 5;
  const int per_wave = ((n_words + NW * 64 - 1) / (NW * 64)) * 64;  // words per wave, a multiple of 64
  const int wb = wave * per_wave, we = min(n_words, wb + per_wave);
  int cnt = 0;
  for (int w0 = wb; w0 < we; w0 += 64) {
    const int wi = w0 + lane;
    cnt += (wi < we) ? __popc(s.bitmap[wi]) : 0;
  }
  cnt = rp_wave_sum(cnt);
  if (lane == 0) s.misc[RPS_RED + wave] = cnt;
  __syncthreads();
  int off = 0;
  for (int w2 = 0; w2 < wave; ++w2) off += s.misc[RPS_RED + w2];
  for (int w0 = wb; w0 < we; w0 += 64) {
    const int wi = w0 + lane;
    uint32_t bits = (wi < we) ? s.bitmap[wi] : 0u;
    const int c = __popc(bits);
    int inc = c;  // inclusive prefix over the lanes
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const int t = __shfl_up(inc, dd);
      if (lane >= dd) inc += t;
    }
    int pos = off + inc - c;
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1u;
      emit(pos++, wi * 32 + b);
    }
    off += __shfl(inc, 63);
    if (wi < we) s.bitmap[wi] = 0u;
  }
  __syncthreads();
}

// K1-K3: row-blocked Gram X^T X + cosine key + per-row top-K, fused (ItemKNN cosine fit).
//
// Replaces compute_cosine_similarity (recpack/algorithms/nearest_neighbour.py:69-84: sklearn
// cosine_similarity(X.T) -> scipy csr_matmat, then setdiag(0)) and get_top_K_values (recpack/util.py:50-96,
// called at nearest_neighbour.py:217).  The I x I similarity matrix is never materialised:
//
//   task = (item row i, column tile t, slice of the users of i)
//   accumulate  c_ij = |users(i) & users(j)| for j in tile t into a dense counter tile in LDS: bytes for rows of at most 255 users,
//               16-bit pairs for the others, 32 bits for the sums of sliced rows and the dense Gram of EASE (one LDS atomic per
//               scalar product).  A wave takes its even share of the task's users (one per lane, at most 64 per batch), cuts their
//               tile segments into chunks of eight 16-bit tile-relative columns (K0's x_rel: one 16-byte load) and deals chunk c to
//               lane c mod 64
//   select      exact top-K of key(i,j) = fl(c_ij^2) * fl(1/c_j)  (monotone in cosine for fixed i), ties by
//               smaller j (rp_select.h); emit (j, c_ij / sqrt(c_i c_j)) in column order - several tiles: per-tile lists of
//               (j, c_ij), merged per row by k_itemknn_merge[_wave]
//
// Rows whose user list is long are cut into slices (partial tiles summed by a second launch); tasks are handed out heaviest
// first through ticket counters (one per XCD for the heavy instance) to persistent workgroups.
#include <math.h>

#include <algorithm>

#include "rp_comm.h"
#include "rp_common.h"
#include "rp_gemm.h"
#include "rp_merge_big.h"
#include "rp_select.h"
#include "rp_wave.h"

struct rp_task {
  int32_t row;      // item row
  int32_t tile;     // column tile
  int32_t p_begin;  // slice of xt_idx (mode 0/1)  |  first partial slot (mode 2)
  int32_t p_end;    //                              |  one past the last partial slot
  int32_t slot;     // mode 1: partial slot to write
  int32_t mode;     // 0 = whole row: accumulate + select; 1 = slice: accumulate + dump; 2 = sum partials + select
};

struct rp_fit_params {
  const int32_t* x_ptr;
  const int32_t* x_idx;
  const uint16_t* x_rel;   // x_idx relative to the entry's column tile (rphip_csr::x_rel): what the accumulate loop reads
  const int32_t* xt_ptr;
  const int32_t* xt_idx;
  const int32_t* deg;
  const double* inv_deg;   // column weight of the key: 1/c_j (cosine) or c_j^-alpha (conditional probability)
  const int32_t* tile_ptr;
  int n_items, n_tiles, tile_w, K, row_begin;
  int cand_stride;         // n_tiles > 1: entries between two tiles' lists of a row (K; min(K, tile_w) on the large-K route)
  int nnz;
  int key_pow;             // 2 = cosine, 1 = conditional probability
  const rp_task* tasks;
  const int32_t* n_tasks;
  int32_t* ticket;
  int32_t* out_idx;   // n_tiles == 1: final [rows*K]; else per-tile candidates [rows*T*K]
  double* out_val;    // n_tiles == 1 only
  uint32_t* out_cnt;  // n_tiles > 1: candidate counts c_ij
  int32_t* out_nnz;   // [rows] or [rows*T]
  uint32_t* part;     // partial tiles [slot][tile_w]
  rp_task* spill_tasks;          // light kernel: tasks whose candidate list overflowed are appended here ...
  int32_t* spill_count;          // ... (= the heavy kernel's task array and task count)
  unsigned long long* n_products;  // light kernel: += scalar products (roofline split); may be NULL
  float* dense;       // K4 (EASE): write the Gram row tile as f32 into dense[row*dense_ld + col] instead of selecting
  long long dense_ld;
  float diag_add;     // lambda, added on the diagonal of the dense Gram
};

// Selection key of entry (i, j), monotone in the similarity for a fixed row i:
//   cosine (pw = 2):                 fl(c*c) * fl(1/c_j)        (nearest_neighbour.py:69-84)
//   conditional probability (pw = 1): fl(c) * c_j^-alpha         (nearest_neighbour.py:22-66; alpha = pop_discount)
__device__ __forceinline__ uint64_t rp_fit_key(uint32_t c, double colw, int pw = 2) {
  // k in [2^-31, 2^32): the exponent offset from 992 fits 6 bits; shift the 58
  // significant bits to the top so that the first radix level already resolves 6 mantissa bits.
  double k = (double)((uint64_t)c * (uint64_t)(pw == 2 ? c : 1u)) * colw;
  uint64_t b = (uint64_t)__double_as_longlong(k);
  return (b - (992ull << 52)) << 6;
}

// accumulator tile: CB-bit counters, 32 / CB of them per LDS word: u32 (any row), u16 pairs (rows and slices with at most
// 65 535 users: c_ij <= c_i) or bytes (at most 255 users)
template <int CB>
struct rp_dense_src {
  const uint32_t* acc;
  const double* inv_deg;
  int t0, row, n_cols, key_pow;
  __device__ __forceinline__ uint32_t count(int w) const {
    if (CB == 8) return (acc[w >> 2] >> (8 * (w & 3))) & 255u;
    if (CB == 16) return (acc[w >> 1] >> (16 * (w & 1))) & 65535u;
    return acc[w];
  }
  __device__ __forceinline__ bool valid(int w) const { return count(w) != 0u && (t0 + w) != row; }
  __device__ __forceinline__ uint64_t key(int w) const { return rp_fit_key(count(w), inv_deg[t0 + w], key_pow); }
  __device__ __forceinline__ double colw(int w) const { return inv_deg[t0 + w]; }
  __device__ __forceinline__ uint64_t key_w(int w, double cw) const { return rp_fit_key(count(w), cw, key_pow); }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)(t0 + w); }
};

#ifndef RP_FIT_CH
#define RP_FIT_CH 8  // entries per chunk of the accumulate loop (8 or 16: one or two 16-byte loads of x_rel)
#endif
constexpr int RP_TICKET_CHUNK = 8;      // light tasks a workgroup draws per ticket atomic
constexpr int RP_LIGHT_THREADS = 256;   // light rows: 4-wave workgroups, several per CU
constexpr int RP_LIGHT_SEG = 1024;      // candidate-list entries per wave (4096 per task); overflow -> heavy kernel
constexpr int RP_LIGHT_MAX_DEG = 255;   // byte counters
constexpr int RP_LIGHT_BITS = 10;       // 1024-bin histogram: 25.6 + 4 + 8 KB per workgroup, four of them per CU
constexpr int RP_PAIR_THREADS = 1024;   // heavy rows: u16 counter pairs, no candidate list, two workgroups per CU
constexpr int RP_PAIR_MAX_USERS = 65535;

// FINALIZE only gives the second launch (sum of partial tiles + select) its own name in profiles.
// NTHR / CB (counter bits), three instances of the accumulate + select kernel:
//   <.,1024,32>  u32 counters, one workgroup per CU: the finalize launch (sums of partial tiles can exceed 65 535) and
//                the dense Gram of EASE
//   <., 512,16>  heavy rows: u16 counter pairs and NO candidate list (their tiles are mostly non-zero: the selection
//                walks the tile itself, rp_select_tile) - 63 KB of LDS, so two workgroups share a CU and one row's
//                selection (VALU/LDS) overlaps another row's accumulation (memory)
//   <., 256, 8>  rows with at most 255 users: byte counters, first-touch candidate lists, three workgroups per CU
template <bool FINALIZE, int NTHR, int CB>
__global__ void __launch_bounds__(NTHR, (CB == 16 ? 8 : 1)) k_itemknn_fit(rp_fit_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = NTHR / 64;
  constexpr bool BYTES = (CB == 8), PAIRS = (CB == 16);
  // selection straight from the tile (rp_select_tile: sampled threshold, no candidate list): the heavy rows' u16 tiles and - round 6 -
  // the finalize launch, whose rows (more than 65 535 users) have tiles without a zero: the list-driven selection evaluated every
  // one of 25 600 keys once per radix level, 0.47 ms for a launch of a few dozen tasks (half of K1-K3 at 100K x 20K)
  constexpr bool TSEL = PAIRS || (FINALIZE && CB == 32);
  constexpr int SEG = BYTES ? RP_LIGHT_SEG : 0;
  constexpr int SBITS = BYTES ? RP_LIGHT_BITS : 11;  // radix digit of the selection (light rows: 10 -> 4 KB histogram)
  uint32_t* acc = reinterpret_cast<uint32_t*>(smem);
  const int W = P.tile_w;
  const int acc_words = (W * CB + 31) / 32;
  char* sel_base = smem + (((size_t)acc_words * 4 + 15) & ~(size_t)15);
  rp_tsel_smem tsel = rp_tsel_carve(sel_base, W);
  rp_select_smem sel;
  if constexpr (TSEL) {  // no list: only the histogram (scratch of the accumulate phase) and misc are shared
    sel.hist = tsel.hist;
    sel.skey = tsel.skey;
    sel.scol = tsel.scol;
    sel.list = nullptr;
    sel.misc = tsel.misc;
    for (int w = threadIdx.x; w < (W + 31) / 32 + 4; w += NTHR) tsel.bitmap[w] = 0u;
  } else {
    sel = rp_select_carve(sel_base, W, NW, SEG, 1 << SBITS);
  }
  int* s_task = &sel.misc[RPS_USER];  // all LDS lives in the dynamic region (keeps its base 16-B aligned)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = P.n_tiles;

  for (int w = tid; w < acc_words; w += NTHR) acc[w] = 0u;
  if (tid == 0) sel.misc[RPS_USER + 16] = 0;  // first-touch block counter (light rows)
  const int n_tasks = *P.n_tasks;

  // Tasks come from a ticket counter (heaviest first).  The NEXT task is fetched by one lane of the last
  // wave while the current one is processed (two LDS slots), so the ticket atomic and the descriptor load
  // never sit on the critical path.
  // (tickets are drawn RP_TICKET_CHUNK at a time: one returning atomic on ONE word per task was the light kernel's bound - a word
  // takes a few tens of such atomics per microsecond, and 313 835 tasks at C4 spent 8 of the kernel's 10.3 ms queueing for it)
  int tk_next = 0, tk_end = 0, xq = 0;  // (live in the one thread that fetches)
  auto fetch_task = [&](int q) {
    int* d = s_task + q * 8;
    constexpr int TCH = BYTES ? RP_TICKET_CHUNK : 1;  // (heavy tasks are long and sorted heaviest first: one at a time)
    int ti;
    if constexpr (!BYTES && !FINALIZE) {
      // XCD-aware hand-out (heavy rows, dense Gram; round 4): the T column-tile tasks of one (row, user range) are consecutive in
      // the task array; group g of T tasks belongs to the XCD g mod 8, which has its own ticket counter - the workgroups that share
      // an L2 (blockIdx mod 8, MI355X_MICROARCH.md) take the tiles of one row at about the same time: its users' ids, their tile
      // pointers (one 4 (T + 1)-byte record serves all T tiles) and the lines of a user row that straddle two tiles are fetched
      // over the fabric once instead of T times.  A workgroup whose own list is exhausted takes from the next XCD's.
      // Measured at 1M x 100K: heavy rows 8.92 -> 8.66 ms, the dense Gram of EASE 18.7 -> 16.9 ms.
      ti = n_tasks;
      while (xq < 8) {
        const int x = ((int)blockIdx.x + xq) & 7;
        const int n = atomicAdd(P.ticket + x, 1);
        const int cand = ((n / T) * 8 + x) * T + n % T;
        if (cand < n_tasks) {
          ti = cand;
          break;
        }
        ++xq;
      }
    } else {
      if (tk_next == tk_end) {
        tk_next = atomicAdd(P.ticket, TCH);
        tk_end = tk_next + TCH;
      }
      ti = tk_next++;
    }
    if (ti < n_tasks) {
      const rp_task t = P.tasks[ti];
      d[0] = t.row;
      d[1] = t.tile;
      d[2] = t.p_begin;
      d[3] = t.p_end;
      d[4] = t.slot;
      d[5] = t.mode;
      d[6] = 1;
    } else {
      d[6] = 0;
    }
  };
  if (tid == NTHR - 64) fetch_task(0);
  int par = 0;
  long long n_prod_all = 0;  // this lane's share of the products of all the workgroup's tasks

  while (true) {
    __syncthreads();
    const int* cur = s_task + par * 8;
    if (!cur[6]) break;
    rp_task tk;
    tk.row = cur[0];
    tk.tile = cur[1];
    tk.p_begin = cur[2];
    tk.p_end = cur[3];
    tk.slot = cur[4];
    tk.mode = cur[5];
    par ^= 1;
    if (tid == NTHR - 64) fetch_task(par);
    const int row = tk.row, tile = tk.tile;
    const int t0 = tile * W;
    const int n_cols = min(W, P.n_items - t0);
    // light rows: first-touch candidate list = pool of NBLK blocks of 64 entries shared by the waves
    constexpr int NBLK = (SEG * NW) / 64;
    int* blk_cnt = &sel.misc[RPS_USER + 16];  // blocks handed out (reset at the end of every task)
    int lcnt = 0;             // candidates this wave appended
    int blk = NBLK, fill = 64;  // current block (none yet) and its fill

    if (tk.mode != 2) {
      // ---- accumulate: users p in [p_begin, p_end) of item `row`.
      // A wave takes 64 users (one per lane: row pointer pair), cuts their index segments into chunks of
      // CH entries, and hands chunk c to lane c % 64: every lane then issues CH independent index loads
      // before the first LDS atomic, instead of one dependent load per user per round.
      constexpr int CH = RP_FIT_CH;
      RP_LDS uint32_t* own_sc = (RP_LDS uint32_t*)sel.hist + wave * 64;  // the histogram is idle here
      RP_LDS uint16_t* ft_list = (RP_LDS uint16_t*)sel.list;  // light rows: first-touch candidates
      const int diag = row - t0;
      int n_prod = 0;
      // users per wave and batch: 64 - but a light row (at most 255 users) is spread EVENLY over the four waves (round 5): in batches of
      // 64 a row of 87 users gave wave 0 its 64, wave 1 the other 23 and left two waves idle through the task's longest phase
      // (the heavier instances alike: a row - or slice - with fewer than 64 NW users is dealt out evenly)
      const int bw = max(1, min(64, (tk.p_end - tk.p_begin + NW - 1) / NW));
      const int stride = NW * bw;
      // software pipeline over the wave's batches of 64 users: the user ids of batch n+2 and the row pointers of
      // batch n+1 are in flight while batch n is expanded into chunks
      auto load_user = [&](int bs) -> int {
        const int p = bs + lane;
        return (lane < bw && p < tk.p_end) ? P.xt_idx[p] : -1;
      };
      auto load_seg = [&](int u, int& s, int& e) {
        s = 0;
        e = 0;
        if (u >= 0) {
          if (T == 1) {
            s = P.x_ptr[u];
            e = P.x_ptr[u + 1];
          } else {
            const int32_t* tp = P.tile_ptr + (int64_t)u * (T + 1) + tile;
            s = tp[0];
            e = tp[1];
          }
        }
      };
      int base = tk.p_begin + wave * bw;
      int u_next = load_user(base + stride);
      int s_cur, e_cur;
      load_seg(load_user(base), s_cur, e_cur);
      for (; base < tk.p_end; base += stride) {
        // chunks of eight 16-bit tile-relative columns = ONE 16-byte load, on the dword grid of x_rel: a slice that starts at an odd
        // entry begins one entry into its first chunk (`lead`, masked)
        const int lead = (e_cur > s_cur) ? (s_cur & 1) : 0;
        const int s = s_cur - lead, len = e_cur - s_cur + lead;
        n_prod += e_cur - s_cur;
        load_seg(u_next, s_cur, e_cur);           // batch n+1
        u_next = load_user(base + 2 * stride);    // batch n+2
        const int nch = (len + CH - 1) / CH;
        const int len_lead = len | (lead << 24);  // (one broadcast for both)
        const int inc = rp_wave_incl_scan(nch, lane);  // DPP: no LDS round trips
        const int choff = inc - nch;       // first chunk of this lane's user
        const int n_chunks = __builtin_amdgcn_readlane(inc, 63);
        // chunk c goes to lane c % 64; the index loads of step c0+64 are issued before the atomics of step c0.
        // Owner of chunk c = largest lane l with choff[l] <= c: every owner whose first chunk falls into this step
        // writes its lane id at that chunk's slot of a 64-word LDS scratch, a running maximum (DPP) fills the gaps;
        // slot 0 starts from the owner of the previous step's last chunk.  Steps are fetched in increasing order.
        int carry = 0;
        // One chunk step = (raw loaded entries, the valid range [lo, n) among them).  The 16-byte load of a step is ALWAYS issued (a
        // step past the batch's chunks, or a lane without a chunk, reads the head of the array): with the loads under branches the
        // compiler cannot count the loads in flight and puts s_waitcnt vmcnt(0) in front of every use, which makes the step being
        // applied wait for the step just requested.  The raw registers are only touched when the step is applied (masking
        // included), so the next step's loads stay in flight meanwhile.  x_rel is padded: no chunk reads past its end.
        typedef int v4i_a4 __attribute__((ext_vector_type(4), aligned(4)));
        struct step_t {
          v4i_a4 a[CH / 8];
          int n, lo;
        };
        auto fetch = [&](int c0, step_t& S) {
          int n = 0, pos = 0, lo_e = 0;
          if (c0 < n_chunks) {  // (uniform; no loads inside)
            const int c = c0 + lane;
            own_sc[lane] = (lane == 0) ? (uint32_t)carry : 0u;
            RP_WAVE_FENCE();
            if (nch > 0 && (unsigned)(choff - c0) < 64u) own_sc[choff - c0] = (uint32_t)lane;
            RP_WAVE_FENCE();
            const int lo = rp_wave_incl_max((int)own_sc[lane]);
            carry = __builtin_amdgcn_readlane(lo, 63);
            const int us = __shfl(s, lo), ull = __shfl(len_lead, lo), co = __shfl(choff, lo);
            const int ul = ull & 0xffffff;
            const int k0 = (c - co) * CH;
            n = (c < n_chunks) ? min(CH, ul - k0) : 0;
            lo_e = (k0 == 0) ? (ull >> 24) : 0;
            pos = us + k0;
          }
          S.n = n;
          S.lo = lo_e;
          const uint16_t* src = P.x_rel + ((n > 0) ? pos : 0);  // (pos is even: a dword-aligned address)
#pragma unroll
          for (int h = 0; h < CH / 8; ++h) {
            S.a[h] = *reinterpret_cast<const v4i_a4*>(src + 8 * h);
          }
        };
        auto unpack = [&](const step_t& S, int (&jv)[CH]) {  // columns inside the tile; -1 = no entry
          int w[CH / 2];
#pragma unroll
          for (int h = 0; h < CH / 8; ++h) {
            w[4 * h] = S.a[h].x;
            w[4 * h + 1] = S.a[h].y;
            w[4 * h + 2] = S.a[h].z;
            w[4 * h + 3] = S.a[h].w;
          }
#pragma unroll
          for (int e = 0; e < CH; ++e) {
            const int v = (e & 1) ? (int)((uint32_t)w[e >> 1] >> 16) : (w[e >> 1] & 0xffff);
            jv[e] = (e >= S.lo && e < S.n) ? v : -1;
          }
        };
        auto apply = [&](const step_t& S) {
          int jv[CH];
          unpack(S, jv);
          if constexpr (BYTES) {
            // returning adds: the lane that finds its byte at zero is the column's first toucher and appends the
            // slot to the candidate list - the selection then never scans the tile.  The list is a pool of 64-entry
            // blocks; a wave takes a new block (one LDS atomic) when its current one is full.
            uint32_t old[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) {
              const int j = jv[e];
              old[e] = (jv[e] >= 0) ? atomicAdd(&acc[j >> 2], 1u << (8 * (j & 3))) : 0xffffffffu;
            }
            // the step's first touches are appended in ONE go: a lane counts its own (up to CH), a scan over the lanes gives it
            // its offset behind the current block's fill, new blocks are drawn with one LDS atomic for all of them (one ballot,
            // prefix and block check per ENTRY before: 120 instructions per step against 55)
            uint32_t fmask = 0;
#pragma unroll
            for (int e = 0; e < CH; ++e) {
              const int j = jv[e];
              const bool first = jv[e] >= 0 && ((old[e] >> (8 * (j & 3))) & 255u) == 0u && j != diag;
              fmask |= first ? (1u << e) : 0u;
            }
            const int c = __popc(fmask);
            const int inc = rp_wave_incl_scan(c, lane);
            const int tot = __builtin_amdgcn_readlane(inc, 63);
            if (tot) {  // (wave-uniform)
              const int total = fill + tot;
              int nb0 = 0;
              if (total > 64) {
                int b = 0;
                if (lane == 0) b = atomicAdd(blk_cnt, (total - 1) >> 6);  // blocks beyond the current one
                nb0 = __builtin_amdgcn_readfirstlane(b);
              }
              int g = fill + inc - c;  // position of this lane's first new entry, counted from the current block's start
#pragma unroll
              for (int e = 0; e < CH; ++e) {
                if (fmask & (1u << e)) {
                  const int bk = (g < 64) ? blk : nb0 + ((g - 64) >> 6);
                  if (bk < NBLK) ft_list[bk * 64 + (g & 63)] = (uint16_t)jv[e];
                  ++g;
                }
              }
              if (total > 64) {
                blk = nb0 + ((total - 65) >> 6);
                fill = ((total - 65) & 63) + 1;
              } else {
                fill = total;
              }
              lcnt += tot;
            }
          } else if constexpr (PAIRS) {
#pragma unroll
            for (int e = 0; e < CH; ++e)
              if (jv[e] >= 0) {
                const int j = jv[e];
                atomicAdd(&acc[j >> 1], 1u << (16 * (j & 1)));
              }
          } else {
#pragma unroll
            for (int e = 0; e < CH; ++e)
              if (jv[e] >= 0) atomicAdd(&acc[jv[e]], 1u);
          }
        };
        // the loads of step c0 + 64 are in flight while step c0 is applied (two register sets, used alternately)
        step_t sa, sb;
        fetch(0, sa);
        for (int c0 = 0; c0 < n_chunks; c0 += 128) {
          fetch(c0 + 64, sb);
          apply(sa);
          fetch(c0 + 128, sa);
          if (c0 + 64 < n_chunks) apply(sb);
        }
      }
      if (BYTES && blk < NBLK && lane >= fill) ft_list[blk * 64 + lane] = (uint16_t)RPS_EMPTY;  // rest of the last block
      if (BYTES) n_prod_all += n_prod;  // products done by the light kernel (roofline split between the two kernels)
      __syncthreads();
      if (!BYTES && tk.mode == 1) {
        uint32_t* dst = P.part + (int64_t)tk.slot * W;
        if constexpr (PAIRS) {
          for (int w = tid; w < W; w += NTHR) dst[w] = (acc[w >> 1] >> (16 * (w & 1))) & 65535u;
          __syncthreads();
          for (int w = tid; w < acc_words; w += NTHR) acc[w] = 0u;
        } else {
          for (int w = tid; w < W; w += NTHR) {
            dst[w] = acc[w];
            acc[w] = 0u;
          }
        }
        continue;
      }
    } else if constexpr (CB == 32) {
      // (round 6) the sum of a row's partial tiles: ONE workgroup reads slices x W words, and with one load per iteration in flight
      // this loop was the whole 0.46 ms of the finalize launch (35 slices of the most popular row at 1M x 100K: 875 dependent round
      // trips per thread).  Eight slices of two columns per thread are requested before the first is added.
      const uint32_t* __restrict__ pp = P.part;
      for (int w = tid; w < W; w += 2 * NTHR) {
        const int w2 = w + NTHR;
        const bool two = w2 < W;
        uint32_t s0 = 0, s1 = 0;
        int sl = tk.p_begin;
        for (; sl + 8 <= tk.p_end; sl += 8) {
          uint32_t v0[8], v1[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            v0[q] = pp[(int64_t)(sl + q) * W + w];
            v1[q] = two ? pp[(int64_t)(sl + q) * W + w2] : 0u;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            s0 += v0[q];
            s1 += v1[q];
          }
        }
        for (; sl < tk.p_end; ++sl) {
          s0 += pp[(int64_t)sl * W + w];
          if (two) s1 += pp[(int64_t)sl * W + w2];
        }
        acc[w] = s0;
        if (two) acc[w2] = s1;
      }
      __syncthreads();
    }

    if (CB == 32 && P.dense) {
      // ---- K4: dense Gram row (ease.py:81 `(X.T @ X).toarray()`), lambda fused on the diagonal
      float* g = P.dense + (long long)row * P.dense_ld + t0;
      for (int w = tid; w < n_cols; w += NTHR) {
        g[w] = (float)acc[w] + ((t0 + w == row) ? P.diag_add : 0.0f);
        acc[w] = 0u;
      }
      continue;
    }
    // ---- select
    rp_dense_src<CB> src{acc, P.inv_deg, t0, row, n_cols, P.key_pow};
    int kp;
    if constexpr (BYTES) {
      const int handed = *blk_cnt;  // (the barrier after the accumulate phase has passed)
      kp = rp_select_topk<NTHR, rp_dense_src<CB>, true, SBITS, 2>(src, n_cols, P.K, sel, SEG, lcnt, handed > NBLK, min(handed, NBLK));
      if (tid == 0) *blk_cnt = 0;  // every wave has read it before the barriers inside the selection
    } else if constexpr (TSEL) {
      kp = rp_select_tile<NTHR>(src, n_cols, P.K, tsel);
    } else {
      kp = rp_select_topk<NTHR>(src, n_cols, P.K, sel, SEG);
    }
    if (BYTES && kp < 0) {
      // more candidates than the list holds: hand the task to the heavy kernel (it runs after this one)
      if (tid == 0) {
        const int pos = atomicAdd(P.spill_count, 1);
        rp_task t2 = tk;
        t2.mode = 0;
        P.spill_tasks[pos] = t2;
      }
      __syncthreads();
      for (int w = tid; w < acc_words; w += NTHR) acc[w] = 0u;
      continue;
    }
    const int rl = row - P.row_begin;
    if (T == 1) {
      int32_t* oi = P.out_idx + (int64_t)rl * P.K;
      double* ov = P.out_val + (int64_t)rl * P.K;
      const double ci = (double)P.deg[row];
      auto emit = [&](int pos, int w) {
        const int j = t0 + w;
        oi[pos] = j;
        ov[pos] = (P.key_pow == 2) ? (double)src.count(w) / sqrt(ci * (double)P.deg[j])
                                   : ((double)src.count(w) * P.inv_deg[j]) / ci;
      };
      if constexpr (BYTES)
        rp_select_emit_bitmap<NTHR>(n_cols, sel, emit);
      else if constexpr (TSEL)
        rp_select_tile_emit<NTHR>(n_cols, tsel, emit);
      else
        rp_select_emit<NTHR>(n_cols, sel, emit, SEG);
      if (tid == 0) P.out_nnz[rl] = kp;
    } else {
      const int64_t o = ((int64_t)rl * T + tile) * P.cand_stride;
      auto emit = [&](int pos, int w) {
        P.out_idx[o + pos] = t0 + w;
        P.out_cnt[o + pos] = src.count(w);
      };
      if constexpr (BYTES) {
        rp_select_emit_bitmap<NTHR>(n_cols, sel, emit);
      } else if constexpr (TSEL)
        rp_select_tile_emit<NTHR>(n_cols, tsel, emit);
      else
        rp_select_emit<NTHR>(n_cols, sel, emit, SEG);
      if (tid == 0) P.out_nnz[(int64_t)rl * T + tile] = kp;
    }
    // clear only what was touched: the candidates (every wave its own list segment) and the diagonal; without a list
    // the whole tile.  Byte tiles: the emit above still reads neighbouring bytes of a word, so clear after a barrier.
    if (BYTES || TSEL) __syncthreads();
    if constexpr (TSEL) {
      for (int w = tid; w < acc_words; w += NTHR) acc[w] = 0u;
    } else {
      if (BYTES)
        rp_select_foreach_pre<NTHR>(sel, [&](int w) { acc[w >> 2] = 0u; });
      else
        rp_select_foreach<NTHR>(n_cols, sel, [&](int w) { acc[w] = 0u; }, SEG);
      if (tid == 0 && (unsigned)(row - t0) < (unsigned)n_cols) acc[BYTES ? ((row - t0) >> 2) : (row - t0)] = 0u;
    }
  }
  if (BYTES && P.n_products) {  // one atomic per wave and launch
    long long v = n_prod_all;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0 && v) atomicAdd(P.n_products, (unsigned long long)v);
  }
}

// ---- merge of per-tile candidates (n_tiles > 1) ---------------------------------------------------
struct rp_list_src {
  const int32_t* col_;
  const uint32_t* cnt_;
  const double* inv_deg;
  int n, key_pow;
  __device__ __forceinline__ bool valid(int w) const { return w < n; }
  __device__ __forceinline__ uint64_t key(int w) const { return rp_fit_key(cnt_[w], inv_deg[col_[w]], key_pow); }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)col_[w]; }
};

struct rp_merge_params {
  const int32_t* cand_idx;
  const uint32_t* cand_cnt;
  const int32_t* cand_nnz;
  const int32_t* deg;
  const double* inv_deg;
  int n_tiles, K, row_begin, n_rows, key_pow;
  int stride;  // k_itemknn_merge_big: entries between two tiles' lists of a row (the other merges: K)
  int32_t* out_idx;
  double* out_val;
  int32_t* out_nnz;
};

constexpr int RP_MERGE_THREADS = 256;  // a row's T*K candidates are few: small workgroups, many per CU

// The per-tile lists of one row where they lie in memory (large K, round 5): slot w of the row's concatenated lists is entry
// w - off[t] of tile t's list - the candidates are not staged in LDS (T * K * 8 bytes: 160 KB at K = 5000 and four tiles), only the
// selection's own list of slot numbers is (2 bytes per candidate: T * K < 32 768).  A key costs two loads more than from LDS; the
// lists of a row are L2-resident after the first radix level.
struct rp_glist_src {
  const int32_t* idx_;   // the row's lists: tile t at idx_ + t * K
  const uint32_t* cnt_;
  const int* off;        // [T + 1] in LDS
  const double* inv_deg;
  int T, K, n, key_pow;
  __device__ __forceinline__ int at(int w) const {
    int t = 0;
    while (t + 1 < T && w >= off[t + 1]) ++t;
    return t * K + (w - off[t]);
  }
  __device__ __forceinline__ bool valid(int w) const { return w < n; }
  __device__ __forceinline__ uint64_t key(int w) const {
    const int a = at(w);
    return rp_fit_key(cnt_[a], inv_deg[idx_[a]], key_pow);
  }
  __device__ __forceinline__ uint32_t col(int w) const { return (uint32_t)idx_[at(w)]; }
};

template <bool GLOBAL>
__global__ void __launch_bounds__(RP_MERGE_THREADS) k_itemknn_merge(rp_merge_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = P.n_tiles, K = P.K;
  const int cap = T * K;
  int32_t* lcol = reinterpret_cast<int32_t*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 4);
  rp_select_smem sel = rp_select_carve(smem + (GLOBAL ? 0 : (size_t)cap * 8), cap, RP_MERGE_THREADS / 64);
  int* s_off = &sel.misc[RPS_USER];  // n_tiles + 1 <= 16 entries
  const int tid = threadIdx.x;
  for (int rl = blockIdx.x; rl < P.n_rows; rl += gridDim.x) {
    __syncthreads();
    if (tid == 0) {
      int o = 0;
      for (int t = 0; t < T; ++t) {
        s_off[t] = o;
        o += P.cand_nnz[(int64_t)rl * T + t];
      }
      s_off[T] = o;
    }
    __syncthreads();
    const int n = s_off[T];
    int32_t* oi = P.out_idx + (int64_t)rl * K;
    double* ov = P.out_val + (int64_t)rl * K;
    const double ci = (double)P.deg[P.row_begin + rl];
    if constexpr (GLOBAL) {
      const int32_t* gi = P.cand_idx + (int64_t)rl * T * K;
      const uint32_t* gc = P.cand_cnt + (int64_t)rl * T * K;
      rp_glist_src src{gi, gc, s_off, P.inv_deg, T, K, n, P.key_pow};
      const int kp = rp_select_topk<RP_MERGE_THREADS>(src, n, K, sel);
      rp_select_emit<RP_MERGE_THREADS>(n, sel, [&](int pos, int w) {
        const int a = src.at(w);
        const int j = gi[a];
        const uint32_t c = gc[a];
        oi[pos] = j;
        ov[pos] = (P.key_pow == 2) ? (double)c / sqrt(ci * (double)P.deg[j]) : ((double)c * P.inv_deg[j]) / ci;
      });
      if (tid == 0) P.out_nnz[rl] = kp;
    } else {
      for (int t = 0; t < T; ++t) {
        const int cn = s_off[t + 1] - s_off[t];
        const int64_t src = ((int64_t)rl * T + t) * K;
        for (int k = tid; k < cn; k += RP_MERGE_THREADS) {
          lcol[s_off[t] + k] = P.cand_idx[src + k];
          lcnt[s_off[t] + k] = P.cand_cnt[src + k];
        }
      }
      __syncthreads();
      rp_list_src src{lcol, lcnt, P.inv_deg, n, P.key_pow};
      const int kp = rp_select_topk<RP_MERGE_THREADS>(src, n, K, sel);
      rp_select_emit<RP_MERGE_THREADS>(n, sel, [&](int pos, int w) {
        const int j = lcol[w];
        oi[pos] = j;
        ov[pos] = (P.key_pow == 2) ? (double)lcnt[w] / sqrt(ci * (double)P.deg[j]) : ((double)lcnt[w] * P.inv_deg[j]) / ci;
      });
      if (tid == 0) P.out_nnz[rl] = kp;
    }
  }
}

// The same merge with one WAVE per row and no barriers: the T*K candidates stream through a wave-private top-K pool
// (rp_wave.h, the pool of the predict kernel: exact radix-select cuts, admission exact for any arrival order).  The
// payload is the candidate's position t*K + k, which orders like its column (tiles are ascending column ranges and a
// tile's candidates are stored in column order), so ties go to the smaller column as everywhere else.  For K <= 384.
constexpr int RP_MERGE_WAVE_KMAX = 384;
__host__ __device__ constexpr int rp_merge_wave_cap(int K) { return ((K + 63) / 64) * 64 + 128; }
__host__ __device__ constexpr size_t rp_merge_wave_bytes(int K) { return (size_t)rp_merge_wave_cap(K) * 12 + 1024 + 128; }

__global__ void __launch_bounds__(64) k_itemknn_merge_wave(rp_merge_params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = P.n_tiles, K = P.K, lane = threadIdx.x;
  const int cap = rp_merge_wave_cap(K);
  rp_wpool<uint64_t> pool;
  pool.key = (RP_LDS uint64_t*)smem;
  pool.pay = (RP_LDS uint32_t*)(pool.key + cap);
  pool.hist = pool.pay + cap;
  pool.gidx = (RP_LDS uint16_t*)(pool.hist + 256);
  for (int rl = blockIdx.x; rl < P.n_rows; rl += gridDim.x) {
    pool.n = 0;
    pool.tau = 0;
    pool.tau_col = 0;
    const int64_t base = (int64_t)rl * T * K;
    for (int t = 0; t < T; ++t) {
      const int cn = P.cand_nnz[(int64_t)rl * T + t];
      for (int k0 = 0; k0 < cn; k0 += 256) {  // four candidates per lane: the loads of a kind go out together
        int col[4];
        uint32_t cnt[4];
        double iw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = k0 + 64 * r + lane;
          col[r] = (k < cn) ? P.cand_idx[base + t * K + k] : -1;
          cnt[r] = (k < cn) ? P.cand_cnt[base + t * K + k] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) iw[r] = (col[r] >= 0) ? P.inv_deg[col[r]] : 0.0;
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
          if (k0 + 64 * r >= cn) break;  // (wave-uniform)
          const uint32_t q = (uint32_t)(t * K + k0 + 64 * r + lane);
          const uint64_t key = (col[0] >= 0) ? rp_fit_key(cnt[0], iw[0], P.key_pow) : 0ull;
          const bool keep = col[0] >= 0 && rp_wpool_admits<uint64_t, 0>(pool, key, q);
          rp_wpool_push<uint64_t, 0>(pool, K, cap, keep, key, q, lane);
          col[0] = col[1]; col[1] = col[2]; col[2] = col[3];
          cnt[0] = cnt[1]; cnt[1] = cnt[2]; cnt[2] = cnt[3];
          iw[0] = iw[1]; iw[1] = iw[2]; iw[2] = iw[3];
        }
      }
    }
    RP_WAVE_FENCE();
    if (pool.n > K) rp_wpool_cut<uint64_t, 0>(pool, K, lane);
    const int n_out = pool.n;
    RP_LDS uint32_t* pp = pool.pay;
    if (lane < 4) pp[n_out + lane] = 0xffffffffu;  // pads the last group of four (n_out + 4 <= cap)
    RP_WAVE_FENCE();
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    int32_t* oi = P.out_idx + (int64_t)rl * K;
    double* ov = P.out_val + (int64_t)rl * K;
    const double ci = (double)P.deg[P.row_begin + rl];
    for (int e = lane; e < n_out; e += 64) {
      const uint32_t q = pp[e];
      int rank = 0;  // position order = column order
      for (int o = 0; o < n_out; o += 4) {
        const v4u w = *reinterpret_cast<RP_LDS v4u*>(&pp[o]);
        rank += (w.x < q ? 1 : 0) + (w.y < q ? 1 : 0) + (w.z < q ? 1 : 0) + (w.w < q ? 1 : 0);
      }
      const int j = P.cand_idx[base + q];
      const double c = (double)P.cand_cnt[base + q];
      oi[rank] = j;
      ov[rank] = (P.key_pow == 2) ? c / sqrt(ci * (double)P.deg[j]) : (c * P.inv_deg[j]) / ci;
    }
    if (lane == 0) P.out_nnz[rl] = n_out;
    RP_WAVE_FENCE();
  }
}

// ---- (round 6) the merge for ANY K: K x tiles >= 32 768 candidates per row (ItemKNN(K=20000) at 100 000 items): rp_merge_big.h
__global__ void __launch_bounds__(RP_MBIG_THREADS) k_itemknn_merge_big(rp_merge_params P) {
  __shared__ rp_mbig_smem S;
  const int T = P.n_tiles, K = P.K, St = P.stride;
  for (int rl = blockIdx.x; rl < P.n_rows; rl += gridDim.x) {
    const int32_t* gi = P.cand_idx + (int64_t)rl * T * St;
    const uint32_t* gc = P.cand_cnt + (int64_t)rl * T * St;
    const int32_t* cn = P.cand_nnz + (int64_t)rl * T;
    int32_t* oi = P.out_idx + (int64_t)rl * K;
    double* ov = P.out_val + (int64_t)rl * K;
    const double ci = (double)P.deg[P.row_begin + rl];
    const int n_out = rp_merge_big_row(
        S, T, St, K, [&](int t) { return cn[t]; }, [&](int a) { return rp_fit_key(gc[a], P.inv_deg[gi[a]], P.key_pow); },
        [&](int pos, int a) {
          const int j = gi[a];
          const uint32_t c = gc[a];
          oi[pos] = j;
          ov[pos] = (P.key_pow == 2) ? (double)c / sqrt(ci * (double)P.deg[j]) : ((double)c * P.inv_deg[j]) / ci;
        });
    if (threadIdx.x == 0) P.out_nnz[rl] = n_out;
  }
}

// ---- planning: order rows by degree (descending, coarse buckets), cut long rows into slices -------
constexpr int PLAN_BUCKETS = RP_PLAN_BUCKETS;  // 0, then 8 per octave

__device__ __forceinline__ int rp_deg_bucket(int d) {
  if (d <= 0) return 0;
  if (d < 8) return d;  // 1..7
  const int lg = 31 - __clz(d);           // >= 3
  return 8 * (lg - 2) + ((d >> (lg - 3)) & 7);  // 8.. ; monotone in d
}

__global__ void k_plan_hist(const int32_t* __restrict__ deg, int row_begin, int n_rows, int32_t* __restrict__ bucket_cnt) {
  __shared__ int h[PLAN_BUCKETS];
  for (int i = threadIdx.x; i < PLAN_BUCKETS; i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x)
    atomicAdd(&h[rp_deg_bucket(deg[row_begin + r])], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < PLAN_BUCKETS; i += blockDim.x)
    if (h[i]) atomicAdd(&bucket_cnt[i], h[i]);
}

// bucket_cnt -> start offsets in DESCENDING bucket order (in place)
__global__ void k_plan_offsets(int32_t* __restrict__ bucket_cnt) {
  if (threadIdx.x == 0) {
    int off = 0;
    for (int b = PLAN_BUCKETS - 1; b >= 0; --b) {
      int c = bucket_cnt[b];
      bucket_cnt[b] = off;
      off += c;
    }
  }
}

// (round 6: one returning global atomic per ROW on a few dozen bucket words - most rows share the buckets of the small degrees - took
// 0.21 ms at 100 000 rows, two thirds of the whole plan; now the rank inside the block by LDS atomics and one global atomic per
// (block, bucket))
__global__ void k_plan_scatter(const int32_t* __restrict__ deg, int row_begin, int n_rows, int32_t* __restrict__ bucket_off,
                               int32_t* __restrict__ order) {
  __shared__ int h[PLAN_BUCKETS];
  for (int r0 = blockIdx.x * blockDim.x; r0 < n_rows; r0 += gridDim.x * blockDim.x) {
    for (int i = threadIdx.x; i < PLAN_BUCKETS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int r = r0 + threadIdx.x;
    int b = 0, rk = 0;
    if (r < n_rows) {
      b = rp_deg_bucket(deg[row_begin + r]);
      rk = atomicAdd(&h[b], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PLAN_BUCKETS; i += blockDim.x)
      if (h[i]) h[i] = atomicAdd(&bucket_off[i], h[i]);
    __syncthreads();
    if (r < n_rows) order[h[b] + rk] = row_begin + r;
    __syncthreads();
  }
}

// per ordered row: number of accumulate tasks, finalize tasks and partial slots
__global__ void k_plan_count(const int32_t* __restrict__ deg, const int32_t* __restrict__ order, int n_rows, int n_tiles,
                             int slice_users, int heavy_thr, int light_max, int32_t* __restrict__ n_acc, int32_t* __restrict__ n_fin,
                             int32_t* __restrict__ n_slot, int32_t* __restrict__ n_light) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_rows; k += gridDim.x * blockDim.x) {
    const int d = deg[order[k]];
    int a = 0, f = 0, s = 0, l = 0;
    if (d > 0) {
      if (d <= light_max) {
        l = n_tiles;  // light kernel: one task per column tile
      } else if (d > heavy_thr) {
        const int ns = (d + slice_users - 1) / slice_users;
        a = ns * n_tiles;
        f = n_tiles;
        s = ns * n_tiles;
      } else {
        a = n_tiles;
      }
    }
    n_acc[k] = a;
    n_fin[k] = f;
    n_slot[k] = s;
    n_light[k] = l;
  }
}

__global__ void k_plan_write(const int32_t* __restrict__ deg, const int32_t* __restrict__ xt_ptr, const int32_t* __restrict__ order,
                             int n_rows, int n_tiles, int slice_users, int heavy_thr, int light_max,
                             const int32_t* __restrict__ off_acc, const int32_t* __restrict__ off_fin,
                             const int32_t* __restrict__ off_slot, const int32_t* __restrict__ off_light,
                             rp_task* __restrict__ acc_tasks, rp_task* __restrict__ fin_tasks, rp_task* __restrict__ light_tasks) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_rows; k += gridDim.x * blockDim.x) {
    const int row = order[k];
    const int d = deg[row];
    if (d <= 0) continue;
    if (d <= light_max) {
      int ol = off_light[k];
      for (int t = 0; t < n_tiles; ++t) {
        rp_task tk;
        tk.row = row;
        tk.tile = t;
        tk.p_begin = xt_ptr[row];
        tk.p_end = xt_ptr[row] + d;
        tk.slot = -1;
        tk.mode = 0;
        light_tasks[ol++] = tk;
      }
      continue;
    }
    const int p0 = xt_ptr[row];
    int oa = off_acc[k];
    if (d > heavy_thr) {
      const int ns = (d + slice_users - 1) / slice_users;
      int of = off_fin[k];
      int os = off_slot[k];
      for (int s = 0; s < ns; ++s)  // (the tiles of one slice side by side: one XCD takes them together)
        for (int t = 0; t < n_tiles; ++t) {
          rp_task tk;
          tk.row = row;
          tk.tile = t;
          tk.p_begin = p0 + s * slice_users;
          tk.p_end = p0 + min(d, (s + 1) * slice_users);
          tk.slot = os + t * ns + s;
          tk.mode = 1;
          acc_tasks[oa++] = tk;
        }
      for (int t = 0; t < n_tiles; ++t) {
        rp_task fk;
        fk.row = row;
        fk.tile = t;
        fk.p_begin = os + t * ns;
        fk.p_end = os + (t + 1) * ns;
        fk.slot = -1;
        fk.mode = 2;
        fin_tasks[of++] = fk;
      }
    } else {
      for (int t = 0; t < n_tiles; ++t) {
        rp_task tk;
        tk.row = row;
        tk.tile = t;
        tk.p_begin = p0;
        tk.p_end = p0 + d;
        tk.slot = -1;
        tk.mode = 0;
        acc_tasks[oa++] = tk;
      }
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------
int rp_model_alloc_topk(rphip_ctx* ctx, int64_t n_items, int64_t r0, int64_t r1, int32_t K, rphip_model** out) {
  rphip_model* m = new rphip_model();
  m->ctx = ctx;
  m->kind = RPHIP_MODEL_TOPK;
  m->n_items = n_items;
  m->row_begin = r0;
  m->row_end = r1;
  m->K = K;
  const size_t n = (size_t)std::max<int64_t>(1, (r1 - r0) * (int64_t)K);
  const int e1 = rp_dev_alloc(ctx, n * sizeof(int32_t), (void**)&m->idx);
  const int e2 = rp_dev_alloc(ctx, n * sizeof(double), (void**)&m->val);
  int e3 = rp_dev_alloc(ctx, n * sizeof(float), (void**)&m->val32);
  if (e3 == RPHIP_OK) e3 = rp_dev_alloc(ctx, (n + 8) * sizeof(int2), (void**)&m->pairs);  // +8: aligned 4-entry chunk loads may overrun
  const int e4 = rp_dev_alloc(ctx, (size_t)std::max<int64_t>(1, r1 - r0) * sizeof(int32_t), (void**)&m->nnz);
  if (e1 != RPHIP_OK || e2 != RPHIP_OK || e3 != RPHIP_OK || e4 != RPHIP_OK) {
    rphip_model_destroy(m);
    RP_FAIL(RPHIP_ENOMEM, "hipMalloc of the neighbour lists failed (%lld rows x K=%d)", (long long)(r1 - r0), K);
  }
  *out = m;
  return RPHIP_OK;
}

extern "C" int rphip_model_destroy(rphip_model* m) {
  if (!m) return RPHIP_OK;
  RP_ENTER(m->ctx);  // (the block cache, the stream and a pending gather belong to the context: a destructor on another thread waits its turn)
  if (m->ctx) hipSetDevice(m->ctx->device);
  if (m->ctx && m->borrows_B) hipStreamSynchronize(m->ctx->stream);  // the caller's B may still be read by a launch in flight
  if (m->ctx && m->gather_ev) rp_model_gather_wait(m->ctx, m, false);   // (a gathered model whose f32 rows are still arriving)
  rp_dev_free(m->ctx, m->idx);
  rp_dev_free(m->ctx, m->val);
  rp_dev_free(m->ctx, m->val32);
  rp_dev_free(m->ctx, m->pairs);
  rp_dev_free(m->ctx, m->toff);
  rp_dev_free(m->ctx, m->rowmax);
  rp_dev_free(m->ctx, m->nnz);
  if (!m->borrows_B) rp_dev_free(m->ctx, m->B);
  rp_dev_free(m->ctx, m->B16);
  rp_dev_free(m->ctx, m->b16_delta);
  rp_dev_free(m->ctx, m->b16_rowmax);
  rp_dev_free(m->ctx, m->b8_scale);
  delete m;
  return RPHIP_OK;
}

extern "C" int rphip_model_empty_rows(rphip_ctx* ctx, const rphip_model* m, int64_t* out) {
  RP_ENTER(ctx);
  if (!ctx || !m || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (m->empty_rows < 0 && m->kind == RPHIP_MODEL_DENSE && m->B && !m->is_gram && m->n_cols == 0) {  // (a gathered model: counted when asked for)
    RP_TRY(rp_set_device(ctx));
    RP_TRY(rp_model_gather_wait(ctx, m, true));
    RP_TRY(rp_dense_count_empty(ctx, const_cast<rphip_model*>(m)));
  }
  *out = m->empty_rows;
  return RPHIP_OK;
}

extern "C" int rphip_model_info(const rphip_model* m, int32_t* kind, int64_t* n_items, int64_t* row_begin, int64_t* row_end,
                                int32_t* K) {
  if (!m) RP_FAIL(RPHIP_EINVAL, "model is NULL");
  if (kind) *kind = m->kind;
  if (n_items) *n_items = m->n_items;
  if (row_begin) *row_begin = m->row_begin;
  if (row_end) *row_end = m->row_end;
  if (K) *K = m->K;
  return RPHIP_OK;
}

// (the u32 instances: the list-driven selection of the dense-capable instance, or the tile selection of the finalize launch, whose
// fixed parts - histogram and collected list - are the larger ones on small tiles)
static size_t rp_fit_lds_bytes(int tile_w) { return (size_t)tile_w * 4 + std::max(rp_select_smem_bytes(tile_w), rp_tsel_smem_bytes(tile_w) + 16); }

// rows [row_begin, row_begin + n_rows) in descending degree order (coarse buckets): `bucket` = PLAN_BUCKETS ints of scratch
int rp_plan_row_order(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int n_rows, int32_t* bucket, int32_t* order) {
  hipStream_t st = ctx->stream;
  RP_HIP(hipMemsetAsync(bucket, 0, PLAN_BUCKETS * 4, st));
  const int pb = 256, pg = (int)std::min<int64_t>(1024, rp_cdiv(n_rows, pb));
  hipLaunchKernelGGL(k_plan_hist, dim3(pg), dim3(pb), 0, st, X->deg, (int)row_begin, n_rows, bucket);
  hipLaunchKernelGGL(k_plan_offsets, dim3(1), dim3(64), 0, st, bucket);
  hipLaunchKernelGGL(k_plan_scatter, dim3(pg), dim3(pb), 0, st, X->deg, (int)row_begin, n_rows, bucket, order);
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}

// K1-K3 / K4 for the item rows [row_begin, row_begin + n_rows): plan, the three instances, the merge of the per-tile lists.  The
// neighbour lists go to out_idx / out_val / out_nnz (the first row's place); `big`: the large-K route (per-tile lists of
// min(K, tile) entries, k_itemknn_merge_big).
static int rp_gram_rows(rphip_ctx* ctx, const rphip_csr* X, const rphip_csr* Xc, int64_t row_begin, int n_rows, int32_t K, float* dense,
                        int64_t dense_ld, float diag_add, const double* colw, int key_pow, int32_t* out_idx, double* out_val, int32_t* out_nnz,
                        bool big) {
  const int T = X->n_tiles, W = X->tile_w;
  const int32_t Keff = K;
  const int32_t Ks = big ? std::min<int32_t>(Keff, W) : Keff;
  hipStream_t st = ctx->stream;
  // ---- plan
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  // (neighbour lists: a row or slice handled by the u16-pair kernel may have at most 65 535 users -> slices <= 32 767)
  const int slice_users = (int)std::min<int64_t>(dense ? (1 << 20) : RP_PAIR_MAX_USERS / 2,
                                                  std::max<int64_t>(1024, X->nnz / ((int64_t)ctx->n_cu * 8)));
  const int heavy_thr = 2 * slice_users;
  // upper bounds: every row n_tiles tasks, plus slices
  const int64_t max_slices = (X->nnz / slice_users + X->nnz / heavy_thr + 2) * T;
  const int64_t max_acc = (int64_t)n_rows * T + max_slices;
  const int64_t max_fin = std::min<int64_t>((int64_t)n_rows, X->nnz / heavy_thr + 1) * T;
  char* sc = nullptr;
  // light rows (<= 255 users): 256-thread workgroups with byte counters, several per CU; a task whose candidate
  // list overflows is appended to the heavy kernel's task array, which therefore has room for all of them
  const bool use_light = !dense;
  const int light_max = use_light ? RP_LIGHT_MAX_DEG : 0;
  const int64_t max_light = use_light ? (int64_t)n_rows * T : 0;
  const size_t plan_bytes = (size_t)PLAN_BUCKETS * 4 + (size_t)n_rows * 4 * 5 + (size_t)(n_rows + 1) * 4 * 4 + 64 +
                            (size_t)(max_acc + max_light + max_fin + max_light) * sizeof(rp_task) + 512;
  RP_TRY(rp_scratch(ctx, 0, plan_bytes, (void**)&sc));
  size_t o = 0;
  auto carve = [&](size_t bytes) {
    char* p = sc + o;
    o += (bytes + 15) & ~(size_t)15;
    return p;
  };
  int32_t* bucket = (int32_t*)carve(PLAN_BUCKETS * 4);
  int32_t* order = (int32_t*)carve((size_t)n_rows * 4);
  int32_t* n_acc = (int32_t*)carve((size_t)n_rows * 4);
  int32_t* n_fin = (int32_t*)carve((size_t)n_rows * 4);
  int32_t* n_slot = (int32_t*)carve((size_t)n_rows * 4);
  int32_t* off_acc = (int32_t*)carve((size_t)(n_rows + 1) * 4);
  int32_t* off_fin = (int32_t*)carve((size_t)(n_rows + 1) * 4);
  int32_t* off_slot = (int32_t*)carve((size_t)(n_rows + 1) * 4);
  int32_t* n_light = (int32_t*)carve((size_t)n_rows * 4);
  int32_t* off_light = (int32_t*)carve((size_t)(n_rows + 1) * 4);
  int32_t* tickets = (int32_t*)carve(64);  // [0..7] heavy rows / dense Gram, one per XCD; [8] finalize; [9] light rows
  rp_task* acc_tasks = (rp_task*)carve((size_t)(max_acc + max_light) * sizeof(rp_task));
  rp_task* fin_tasks = (rp_task*)carve((size_t)max_fin * sizeof(rp_task));
  rp_task* light_tasks = (rp_task*)carve((size_t)std::max<int64_t>(1, max_light) * sizeof(rp_task));
  hipMemsetAsync(bucket, 0, PLAN_BUCKETS * 4, st);
  hipMemsetAsync(tickets, 0, 64, st);
  const int pb = 256, pg = (int)std::min<int64_t>(1024, rp_cdiv(n_rows, pb));
  hipLaunchKernelGGL(k_plan_hist, dim3(pg), dim3(pb), 0, st, X->deg, (int)row_begin, n_rows, bucket);
  hipLaunchKernelGGL(k_plan_offsets, dim3(1), dim3(64), 0, st, bucket);
  hipLaunchKernelGGL(k_plan_scatter, dim3(pg), dim3(pb), 0, st, X->deg, (int)row_begin, n_rows, bucket, order);
  hipLaunchKernelGGL(k_plan_count, dim3(pg), dim3(pb), 0, st, X->deg, order, n_rows, T, slice_users, heavy_thr, light_max, n_acc,
                     n_fin, n_slot, n_light);
  RP_TRY(rp_exclusive_scan_i32(ctx, n_acc, off_acc, n_rows));
  RP_TRY(rp_exclusive_scan_i32(ctx, n_fin, off_fin, n_rows));
  RP_TRY(rp_exclusive_scan_i32(ctx, n_slot, off_slot, n_rows));
  RP_TRY(rp_exclusive_scan_i32(ctx, n_light, off_light, n_rows));
  hipLaunchKernelGGL(k_plan_write, dim3(pg), dim3(pb), 0, st, X->deg, X->xt_ptr, order, n_rows, T, slice_users, heavy_thr, light_max,
                     off_acc, off_fin, off_slot, off_light, acc_tasks, fin_tasks, light_tasks);
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, RPHIP_T_FIT_PLAN, 0.0);

  // ---- partial tiles and per-tile candidates
  uint32_t* part = nullptr;
  RP_TRY(rp_scratch(ctx, 1, (size_t)std::max<int64_t>(1, max_slices) * W * 4, (void**)&part));
  int32_t* cand_idx = nullptr;
  uint32_t* cand_cnt = nullptr;
  int32_t* cand_nnz = nullptr;
  if (T > 1 && !dense) {
    char* cb = nullptr;
    const size_t n_c = (size_t)n_rows * T * Ks;
    RP_TRY(rp_scratch(ctx, 2, n_c * 8 + (size_t)n_rows * T * 4 + 64, (void**)&cb));
    cand_idx = (int32_t*)cb;
    cand_cnt = (uint32_t*)(cb + n_c * 4);
    cand_nnz = (int32_t*)(cb + n_c * 8);
    hipMemsetAsync(cand_nnz, 0, (size_t)n_rows * T * 4, st);
  }

  rp_fit_params P;
  P.x_ptr = Xc->x_ptr;
  P.x_idx = Xc->x_idx;
  P.x_rel = Xc->x_rel;
  P.xt_ptr = X->xt_ptr;
  P.xt_idx = X->xt_idx;
  P.deg = X->deg;
  P.inv_deg = colw ? colw : X->inv_deg;
  P.key_pow = key_pow;
  P.tile_ptr = Xc->tile_ptr;
  P.n_items = (int)X->n_items;
  P.nnz = (int)std::max<int64_t>(Xc->nnz, 8);  // (matrices with fewer entries carry a padded copy of their indices)
  P.n_tiles = T;
  P.tile_w = W;
  P.K = K;
  P.cand_stride = Ks;
  P.row_begin = (int)row_begin;
  P.part = part;
  P.dense = dense;
  P.dense_ld = dense_ld;
  P.diag_add = diag_add;
  if (dense) {
    P.out_idx = nullptr;
    P.out_val = nullptr;
    P.out_cnt = nullptr;
    P.out_nnz = nullptr;
  } else if (T == 1) {
    P.out_idx = out_idx;
    P.out_val = out_val;
    P.out_cnt = nullptr;
    P.out_nnz = out_nnz;
  } else {
    P.out_idx = cand_idx;
    P.out_val = nullptr;
    P.out_cnt = cand_cnt;
    P.out_nnz = cand_nnz;
  }
  P.spill_tasks = acc_tasks;
  P.spill_count = off_acc + n_rows;
  P.n_products = nullptr;
  const size_t lds = rp_fit_lds_bytes(W);
  RP_HIP(hipFuncSetAttribute((const void*)k_itemknn_fit<false, RP_FIT_THREADS, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  RP_HIP(hipFuncSetAttribute((const void*)k_itemknn_fit<true, RP_FIT_THREADS, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = ctx->n_cu;
  // heavy rows of the neighbour-list path: u16 pairs, no list
  const size_t plds = ((((size_t)W * 2 + 3) & ~(size_t)3) + 15 & ~(size_t)15) + rp_tsel_smem_bytes(W);
  RP_HIP(hipFuncSetAttribute((const void*)k_itemknn_fit<false, RP_PAIR_THREADS, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
  const int pair_per_cu = std::max<int>(1, std::min<int>(4, (int)((160 * 1024) / plds)));

  if (use_light) {
    // ---- light rows first: their overflow tasks join the heavy kernel's list
    const size_t llds = (((size_t)W + 15) & ~(size_t)15) + rp_select_smem_bytes(W, RP_LIGHT_THREADS / 64, RP_LIGHT_SEG, 1 << RP_LIGHT_BITS);
    RP_HIP(hipFuncSetAttribute((const void*)k_itemknn_fit<false, RP_LIGHT_THREADS, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
    const int per_cu = std::max<int>(1, std::min<int>(8, (int)((160 * 1024) / llds)));
    RP_HIP(hipMemsetAsync(ctx->dev_counters, 0, 64, st));
    RP_HIP(hipMemcpyAsync(ctx->dev_counters + 1, off_light + n_rows, 4, hipMemcpyDeviceToDevice, st));
    rp_fit_params L = P;
    L.tasks = light_tasks;
    L.n_tasks = off_light + n_rows;
    L.ticket = tickets + 9;
    L.n_products = ctx->dev_counters;
    rp_time_begin(ctx, &ev);
    hipLaunchKernelGGL((k_itemknn_fit<false, RP_LIGHT_THREADS, 8>), dim3(ctx->n_cu * per_cu), dim3(RP_LIGHT_THREADS), llds, st, L);
    RP_HIP(hipGetLastError());
    rp_time_end(ctx, ev, RPHIP_T_GRAM_TOPK_WAVE, 0.0);
  }

  rp_time_begin(ctx, &ev);
  P.tasks = acc_tasks;
  P.n_tasks = off_acc + n_rows;
  P.ticket = tickets;
  if (dense)
    hipLaunchKernelGGL((k_itemknn_fit<false, RP_FIT_THREADS, 32>), dim3(grid), dim3(RP_FIT_THREADS), lds, st, P);
  else
    hipLaunchKernelGGL((k_itemknn_fit<false, RP_PAIR_THREADS, 16>), dim3(grid * pair_per_cu), dim3(RP_PAIR_THREADS), plds, st, P);
  RP_HIP(hipGetLastError());
  // algorithmic bytes of the SpGEMM: 4*P_shard + 8*nnz + 4*(U+I+2) + 8*rows*K; the
  // caller scales P by the shard's share, here the full-matrix figure is attributed when rows cover all.
  double work = 8.0 * (double)X->nnz + 4.0 * (double)(X->n_users + X->n_items + 2) + 8.0 * (double)n_rows * K;
  if (X->n_products >= 0) work += 4.0 * (double)X->n_products * ((double)n_rows / (double)X->n_items);
  if (dense) work = 2.0 * (double)(X->n_products >= 0 ? X->n_products : 0) * ((double)n_rows / (double)X->n_items) + 4.0 * (double)n_rows * (double)X->n_items;
  rp_time_end(ctx, ev, dense ? RPHIP_T_GRAM_DENSE : RPHIP_T_GRAM_TOPK, work);
  rp_time_begin(ctx, &ev);
  P.tasks = fin_tasks;
  P.n_tasks = off_fin + n_rows;
  P.ticket = tickets + 8;
  hipLaunchKernelGGL((k_itemknn_fit<true, RP_FIT_THREADS, 32>), dim3(grid), dim3(RP_FIT_THREADS), lds, st, P);
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, dense ? RPHIP_T_GRAM_DENSE : RPHIP_T_GRAM_FINALIZE, 0.0);

  if (dense) return RPHIP_OK;
  if (T > 1) {
    rp_time_begin(ctx, &ev);
    rp_merge_params M;
    M.cand_idx = cand_idx;
    M.cand_cnt = cand_cnt;
    M.cand_nnz = cand_nnz;
    M.deg = X->deg;
    M.inv_deg = colw ? colw : X->inv_deg;
    M.key_pow = key_pow;
    M.n_tiles = T;
    M.K = Keff;
    M.row_begin = (int)row_begin;
    M.n_rows = n_rows;
    M.stride = Ks;
    M.out_idx = out_idx;
    M.out_val = out_val;
    M.out_nnz = out_nnz;
    if (big) {
      hipLaunchKernelGGL(k_itemknn_merge_big, dim3((unsigned)std::min<int64_t>(n_rows, (int64_t)ctx->n_cu * 8)), dim3(RP_MBIG_THREADS), 0, st, M);
      RP_HIP(hipGetLastError());
      rp_time_end(ctx, ev, RPHIP_T_TOPK_MERGE, 16.0 * (double)n_rows * T * Ks);
    } else if (Keff <= RP_MERGE_WAVE_KMAX) {
      const size_t wl = rp_merge_wave_bytes(Keff);
      RP_HIP(hipFuncSetAttribute((const void*)k_itemknn_merge_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl));
      const int wpc = std::max<int>(1, std::min<int>(24, (int)((160 * 1024) / wl)));
      hipLaunchKernelGGL(k_itemknn_merge_wave, dim3((unsigned)std::min<int64_t>(n_rows, (int64_t)ctx->n_cu * wpc)), dim3(64), wl, st, M);
      RP_HIP(hipGetLastError());
      rp_time_end(ctx, ev, RPHIP_T_TOPK_MERGE, 16.0 * (double)n_rows * T * Keff);
    } else {
    const size_t sel_lds = rp_select_smem_bytes(T * Keff, RP_MERGE_THREADS / 64);
    const bool in_lds = (size_t)T * Keff * 8 + sel_lds <= (size_t)160 * 1024;
    const size_t mlds = in_lds ? (size_t)T * Keff * 8 + sel_lds : sel_lds;
    if (mlds > 160 * 1024 || T * Keff >= RPS_FLAG) {
      RP_FAIL(RPHIP_ELIMIT, "K=%d with %d column tiles needs %zu bytes of LDS for the merge", K, T, mlds);
    }
    const int mper = std::max<int>(1, std::min<int>(8, (int)((160 * 1024) / mlds)));
    const int mg = (int)std::min<int64_t>(n_rows, (int64_t)ctx->n_cu * mper);
    if (in_lds) {
      RP_HIP(hipFuncSetAttribute((const void*)k_itemknn_merge<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
      hipLaunchKernelGGL(k_itemknn_merge<false>, dim3(mg), dim3(RP_MERGE_THREADS), mlds, st, M);
    } else {  // (the lists stay in memory: large K)
      RP_HIP(hipFuncSetAttribute((const void*)k_itemknn_merge<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
      hipLaunchKernelGGL(k_itemknn_merge<true>, dim3(mg), dim3(RP_MERGE_THREADS), mlds, st, M);
    }
    RP_HIP(hipGetLastError());
    rp_time_end(ctx, ev, RPHIP_T_TOPK_MERGE, 16.0 * (double)n_rows * T * Keff);
    }
  }
  return RPHIP_OK;
}

// Shared driver of K1-K3 (out != NULL: neighbour lists) and K4 (dense != NULL: dense f32 Gram rows + lambda).
// Xc (dense only): take the USER ROWS from another matrix of the same shape - rows of X^T Xc instead of X^T X
// (EASE_XY's Y^T X); NULL = X itself.
static int rp_gram_core(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K, rphip_model** out,
                        float* dense, int64_t dense_ld, float diag_add, const double* colw = nullptr, int key_pow = 2,
                        const rphip_csr* Xc = nullptr) {
  if (!ctx || !X || (!out && !dense)) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (out) *out = nullptr;
  if (!Xc) Xc = X;
  if (Xc->n_users != X->n_users || Xc->n_items != X->n_items) RP_FAIL(RPHIP_EINVAL, "the two matrices must have the same shape");
  if (row_begin < 0 || row_end > X->n_items || row_begin > row_end)
    RP_FAIL(RPHIP_EINVAL, "row range [%lld,%lld) outside 0..%lld", (long long)row_begin, (long long)row_end, (long long)X->n_items);
  if (K < 1) RP_FAIL(RPHIP_EINVAL, "K must be >= 1 (got %d)", K);
  RP_NEED_DEGREES(X);
  RP_NEED_DEGREES(Xc);
  if (X->xt_row_end >= 0 && (row_begin < X->xt_row_begin || row_end > X->xt_row_end))
    RP_FAIL(RPHIP_ESTATE, "the matrix was built for item rows [%lld,%lld) only (rphip_csr_from_device_rows)", (long long)X->xt_row_begin,
            (long long)X->xt_row_end);
  RP_TRY(rp_set_device(ctx));
  // a row has at most n_items-1 neighbours
  const int32_t Keff = (int32_t)std::min<int64_t>(K, std::max<int64_t>(1, X->n_items - 1));
  const int n_rows = (int)(row_end - row_begin);
  const int T = X->n_tiles, W = X->tile_w;
  // (round 4) one column tile - up to RP_TILE_MAX items - takes ANY K, as the reference does (recpack/algorithms/base.py:302-304,
  // recpack/util.py:63-66): the selection runs on the tile itself and writes the row.  With several tiles a row's per-tile
  // candidate lists meet in the LDS of the merge: K <= RP_K_MAX there.
  // (round 5) ... any K with K x tiles < 32 768 (the selection's list of slot numbers): the merge reads the per-tile lists where
  // they lie when they do not fit the LDS (k_itemknn_merge<true>) - K = 8 191 at 100 000 items, 4 095 at 200 000
  // (round 6) ... and any K beyond that: per-tile lists of min(K, tile) entries merged where they lie by k_itemknn_merge_big, the rows
  // taken in chunks that bound the lists' memory - ItemKNN(K=20000) at 100 000 items
  const bool big_k = T > 1 && (int64_t)T * Keff >= RPS_FLAG;

  K = Keff;  // the model's row stride; rphip_model_info reports it
  if (T > 15) RP_FAIL(RPHIP_ELIMIT, "%d column tiles (n_items=%lld) exceed the supported 15", T, (long long)X->n_items);
  rphip_model* m = nullptr;
  hipStream_t st = ctx->stream;
  if (!dense) {
    RP_TRY(rp_model_alloc_topk(ctx, X->n_items, row_begin, row_end, K, &m));
    const size_t nk = (size_t)std::max<int64_t>(1, (int64_t)n_rows * K);
    hipMemsetAsync(m->idx, 0xFF, nk * sizeof(int32_t), st);
    hipMemsetAsync(m->val, 0, nk * sizeof(double), st);
    hipMemsetAsync(m->nnz, 0, (size_t)std::max(1, n_rows) * sizeof(int32_t), st);
  }
  if (n_rows == 0 || X->nnz == 0 || Xc->nnz == 0) {
    if (dense) return RPHIP_OK;  // rows without users keep what the caller initialised (zeros + lambda)
    int r = rp_model_finalize_topk(ctx, m);
    if (r != RPHIP_OK) {
      rphip_model_destroy(m);
      return r;
    }
    *out = m;
    return RPHIP_OK;
  }

  // ---- the rows, in chunks where the per-tile candidate lists of all rows at once would be too much memory (T x min(K, tile) x 8 bytes per
  // row: 64 GB at K = 20 000 and 100 000 items)
  int rc = RPHIP_OK;
  if (dense) {
    rc = rp_gram_rows(ctx, X, Xc, row_begin, n_rows, K, dense, dense_ld, diag_add, colw, key_pow, nullptr, nullptr, nullptr, false);
    return rc;
  }
  {
    const int64_t per_row = T > 1 ? (int64_t)T * (big_k ? std::min<int64_t>(K, W) : K) * 8 + (int64_t)T * 4 : 0;
    const int64_t budget = (int64_t)6 << 30;
    const int64_t R = (per_row > 0 && per_row * n_rows > budget) ? std::max<int64_t>(1, budget / per_row) : n_rows;
    for (int64_t c0 = 0; c0 < n_rows && rc == RPHIP_OK; c0 += R)
      rc = rp_gram_rows(ctx, X, Xc, row_begin + c0, (int)std::min<int64_t>(R, n_rows - c0), K, nullptr, 0, 0.0f, colw, key_pow, m->idx + c0 * K,
                        m->val + c0 * K, m->nnz + c0, big_k);
  }
  if (rc != RPHIP_OK) {
    rphip_model_destroy(m);
    return rc;
  }
  {
    int r = rp_model_finalize_topk(ctx, m);
    if (r != RPHIP_OK) {
      rphip_model_destroy(m);
      return r;
    }
  }
  *out = m;
  return RPHIP_OK;
}


extern "C" int rphip_itemknn_fit(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K,
                                 rphip_model** out) {
  RP_ENTER(ctx);
  if (!out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  return rp_gram_core(ctx, X, row_begin, row_end, K, out, nullptr, 0, 0.0f);
}

extern "C" int rphip_itemknn_fit_ex(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K,
                                    int32_t similarity, double pop_discount, int32_t normalize_X, rphip_model** out) {
  RP_ENTER(ctx);
  if (!ctx || !X || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (similarity != RPHIP_SIM_COSINE && similarity != RPHIP_SIM_CONDITIONAL_PROBABILITY) RP_FAIL(RPHIP_EINVAL, "unknown similarity %d", similarity);
  if (similarity == RPHIP_SIM_COSINE && !normalize_X) return rp_gram_core(ctx, X, row_begin, row_end, K, out, nullptr, 0, 0.0f);
  RP_TRY(rp_set_device(ctx));
  std::vector<double> w;
  const int64_t n = X->n_items;
  if (similarity == RPHIP_SIM_CONDITIONAL_PROBABILITY) {
    if (!(pop_discount >= 0.0 && pop_discount <= 1.0)) RP_FAIL(RPHIP_EINVAL, "pop_discount must be in [0, 1] (got %g)", pop_discount);
    // column weights c_j^-alpha on the HOST with the C library's pow - the same function numpy calls, so that the
    // selection keys are bit-identical to the restated reference (a device pow could differ in the last bit)
    std::vector<int32_t> deg((size_t)std::max<int64_t>(n, 1));
    w.resize((size_t)std::max<int64_t>(n, 1));
    if (n > 0) RP_HIP(hipMemcpyAsync(deg.data(), X->deg, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RP_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t j = 0; j < n; ++j)
      w[(size_t)j] = deg[(size_t)j] > 0 ? (pop_discount == 0.0 ? 1.0 : pow((double)deg[(size_t)j], -pop_discount)) : 0.0;
  }
  if (normalize_X) {
    const int rc = rp_gram_weighted(ctx, X, row_begin, row_end, K, similarity, w.empty() ? nullptr : w.data(), out);
    RP_HIP(hipStreamSynchronize(ctx->stream));  // the host weights above must outlive their copy
    return rc;
  }
  double* dw = nullptr;
  RP_TRY(rp_scratch(ctx, 6, (size_t)std::max<int64_t>(n, 1) * 8, (void**)&dw));
  if (n > 0) RP_HIP(hipMemcpyAsync(dw, w.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return rp_gram_core(ctx, X, row_begin, row_end, K, out, nullptr, 0, 0.0f, dw, 1);
}

int rp_gram_dense(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, float* G, int64_t ld, float diag_add,
                  const rphip_csr* Xcols) {
  if (!G) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  return rp_gram_core(ctx, X, row_begin, row_end, 1, nullptr, G, ld, diag_add, nullptr, 2, Xcols);
}

extern "C" int rphip_model_topk_download(rphip_ctx* ctx, const rphip_model* m, int32_t* idx_host, double* val_host,
                                         int32_t* nnz_host) {
  RP_ENTER(ctx);
  if (!ctx || !m) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (m->kind != RPHIP_MODEL_TOPK) RP_FAIL(RPHIP_ESTATE, "model is not a neighbour-list model");
  RP_TRY(rp_set_device(ctx));
  const int64_t rows = m->row_end - m->row_begin;
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  if (rows > 0) {
    if (idx_host) RP_HIP(hipMemcpyAsync(idx_host, m->idx, (size_t)rows * m->K * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (val_host) RP_HIP(hipMemcpyAsync(val_host, m->val, (size_t)rows * m->K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (nnz_host) RP_HIP(hipMemcpyAsync(nnz_host, m->nnz, (size_t)rows * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  rp_time_end(ctx, ev, RPHIP_T_D2H, (double)rows * m->K * 12.0);
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

extern "C" int rphip_model_topk_device(const rphip_model* m, void** idx_dev, void** val_dev, void** nnz_dev, int64_t* rows,
                                       int32_t* K) {
  if (!m) RP_FAIL(RPHIP_EINVAL, "model is NULL");
  if (m->kind != RPHIP_MODEL_TOPK) RP_FAIL(RPHIP_ESTATE, "model is not a neighbour-list model");
  if (idx_dev) *idx_dev = m->idx;
  if (val_dev) *val_dev = m->val;
  if (nnz_dev) *nnz_dev = m->nnz;
  if (rows) *rows = m->row_end - m->row_begin;
  if (K) *K = m->K;
  return RPHIP_OK;
}

static int rp_model_from_topk(rphip_ctx* ctx, int64_t n_items, int32_t K, const int32_t* idx, const double* val,
                              const int32_t* nnz, hipMemcpyKind kind, rphip_model** out) {
  if (!ctx || !idx || !val || !nnz || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (n_items < 0 || K < 1) RP_FAIL(RPHIP_EINVAL, "bad shape (%lld x %d)", (long long)n_items, K);
  RP_TRY(rp_set_device(ctx));
  rphip_model* m = nullptr;
  RP_TRY(rp_model_alloc_topk(ctx, n_items, 0, n_items, K, &m));
  const size_t nk = (size_t)n_items * K;
  if (nk) {
    hipEvent_t ev;
    rp_time_begin(ctx, &ev);
    hipError_t e1 = hipMemcpyAsync(m->idx, idx, nk * sizeof(int32_t), kind, ctx->stream);
    hipError_t e2 = hipMemcpyAsync(m->val, val, nk * sizeof(double), kind, ctx->stream);
    hipError_t e3 = hipMemcpyAsync(m->nnz, nnz, (size_t)n_items * sizeof(int32_t), kind, ctx->stream);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
      rphip_model_destroy(m);
      RP_FAIL(RPHIP_EHIP, "copy of the neighbour lists failed");
    }
    if (kind == hipMemcpyHostToDevice) rp_time_end(ctx, ev, RPHIP_T_H2D, (double)nk * 12.0);
    else ctx->ev_pool.push_back(ev);
    int r = rp_model_finalize_topk(ctx, m);
    if (r != RPHIP_OK) {
      rphip_model_destroy(m);
      return r;
    }
    if (kind == hipMemcpyHostToDevice) RP_HIP(hipStreamSynchronize(ctx->stream));
  }
  *out = m;
  return RPHIP_OK;
}

// ---- multi-GPU fit: the ranks' row shards into one model ---------------------------------------------
// Every rank fitted the item rows [bounds[r], bounds[r + 1]) (rphip_itemknn_fit* with a row range); the neighbour lists are
// exchanged with ONE all-gather through the communicator - the shards packed [idx | val | nnz] and padded to the longest one,
// I * K * 12 bytes in all - and land directly in the new model's arrays (one unpack kernel), which is then finalised for predict.
__global__ void k_pack_shard(const int32_t* __restrict__ idx, const double* __restrict__ val, const int32_t* __restrict__ nnz, long long rows, int K,
                             long long max_rows, char* __restrict__ out) {
  int32_t* oi = reinterpret_cast<int32_t*>(out + max_rows * K * 8);
  double* ov = reinterpret_cast<double*>(out);  // (the 8-byte part first: alignment)
  int32_t* on = reinterpret_cast<int32_t*>(out + max_rows * K * 12);
  const long long n = rows * K;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    oi[e] = idx[e];
    ov[e] = val[e];
    if (e < rows) on[e] = nnz[e];
  }
}

struct rp_bounds32 {
  long long b[33];
};

__global__ void k_unpack_shards(const char* __restrict__ in, long long slot, long long max_rows, int K, int W, rp_bounds32 bd, int32_t* __restrict__ idx,
                                double* __restrict__ val, int32_t* __restrict__ nnz) {
  const int r = blockIdx.y;
  const long long r0 = bd.b[r], rows = bd.b[r + 1] - r0;
  const char* src = in + (long long)r * slot;
  const double* sv = reinterpret_cast<const double*>(src);
  const int32_t* si = reinterpret_cast<const int32_t*>(src + max_rows * K * 8);
  const int32_t* sn = reinterpret_cast<const int32_t*>(src + max_rows * K * 12);
  const long long n = rows * K;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    idx[r0 * K + e] = si[e];
    val[r0 * K + e] = sv[e];
    if (e < rows) nnz[r0 + e] = sn[e];
  }
}

extern "C" int rphip_itemknn_gather(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, const int64_t* row_bounds, rphip_model** out) {
  RP_ENTER(ctx);
  if (!ctx || !comm || !local || !row_bounds || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (local->kind != RPHIP_MODEL_TOPK) RP_FAIL(RPHIP_ESTATE, "not a neighbour-list model");
  const int W = rp_comm_world(comm), me = rp_comm_rank(comm);
  if (W > 32) RP_FAIL(RPHIP_ELIMIT, "at most 32 ranks");
  const int64_t I = local->n_items;
  const int K = local->K;
  rp_bounds32 bd;
  long long max_rows = 0;
  for (int r = 0; r <= W; ++r) {
    bd.b[r] = row_bounds[r];
    if (r > 0) {
      if (row_bounds[r] < row_bounds[r - 1]) RP_FAIL(RPHIP_EINVAL, "row bounds must not decrease");
      max_rows = std::max<long long>(max_rows, row_bounds[r] - row_bounds[r - 1]);
    }
  }
  if (row_bounds[0] != 0 || row_bounds[W] != I) RP_FAIL(RPHIP_EINVAL, "the row bounds must cover [0, %lld)", (long long)I);
  if (local->row_begin != row_bounds[me] || local->row_end != row_bounds[me + 1])
    RP_FAIL(RPHIP_EINVAL, "this rank's model holds rows [%lld,%lld), the bounds say [%lld,%lld)", (long long)local->row_begin, (long long)local->row_end,
            (long long)row_bounds[me], (long long)row_bounds[me + 1]);
  RP_TRY(rp_set_device(ctx));
  hipStream_t st = ctx->stream;
  rphip_model* m = nullptr;
  RP_TRY(rp_model_alloc_topk(ctx, I, 0, I, K, &m));
  const long long slot = ((max_rows * K * 12 + max_rows * 4 + 255) / 256) * 256;
  char *sendb = nullptr, *recvb = nullptr;
  int rc = rp_dev_alloc(ctx, (size_t)std::max<long long>(slot, 256), (void**)&sendb);
  if (rc == RPHIP_OK) rc = rp_dev_alloc(ctx, (size_t)std::max<long long>(slot, 256) * W, (void**)&recvb);
  if (rc == RPHIP_OK && max_rows > 0) {
    hipEvent_t ev;
    rp_time_begin(ctx, &ev);
    const long long rows = local->row_end - local->row_begin;
    // (a loop-back run hands every slot this rank's own send buffer: the part of it behind the rank's rows must read as empty lists)
    if (rp_comm_is_loopback(comm)) hipMemsetAsync(sendb, 0, (size_t)slot, st);
    if (rows > 0)
      hipLaunchKernelGGL(k_pack_shard, dim3((unsigned)std::min<long long>(4096, rp_cdiv(rows * K, 256))), dim3(256), 0, st, local->idx, local->val, local->nnz,
                         rows, K, max_rows, sendb);
    rc = rp_comm_all_gather(comm, sendb, recvb, slot, st);
    if (rc == RPHIP_OK) {
      hipLaunchKernelGGL(k_unpack_shards, dim3((unsigned)std::min<long long>(2048, std::max<long long>(1, rp_cdiv(max_rows * K, 256))), W), dim3(256), 0, st,
                         recvb, slot, max_rows, K, W, bd, m->idx, m->val, m->nnz);
      if (hipGetLastError() != hipSuccess) rc = RPHIP_EHIP;
    }
    rp_time_end(ctx, ev, RPHIP_T_TOPK_MERGE, (double)I * K * 24.0);
    if (rc == RPHIP_OK) rc = rp_model_finalize_topk(ctx, m);
  }
  hipStreamSynchronize(st);
  rp_dev_free(ctx, sendb);
  rp_dev_free(ctx, recvb);
  if (rc != RPHIP_OK) {
    rphip_model_destroy(m);
    return rc;
  }
  *out = m;
  return RPHIP_OK;
}

extern "C" int rphip_model_from_topk_host(rphip_ctx* ctx, int64_t n_items, int32_t K, const int32_t* idx_host,
                                          const double* val_host, const int32_t* nnz_host, rphip_model** out) {
  RP_ENTER(ctx);
  return rp_model_from_topk(ctx, n_items, K, idx_host, val_host, nnz_host, hipMemcpyHostToDevice, out);
}

extern "C" int rphip_model_from_topk_device(rphip_ctx* ctx, int64_t n_items, int32_t K, const int32_t* idx_dev,
                                            const double* val_dev, const int32_t* nnz_dev, rphip_model** out) {
  RP_ENTER(ctx);
  return rp_model_from_topk(ctx, n_items, K, idx_dev, val_dev, nnz_dev, hipMemcpyDeviceToDevice, out);
}

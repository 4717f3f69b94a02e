// Internal definitions shared by the translation units of librecpack_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/recpack_hip.h"
#include "../../include/recpack_hip_diag.h"

// ---- errors --------------------------------------------------------------------------------------
void rp_set_error(const char* fmt, ...);

#define RP_FAIL(code, ...)      \
  do {                          \
    rp_set_error(__VA_ARGS__);  \
    return (code);              \
  } while (0)

#define RP_HIP(expr)                                                                           \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      rp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return (_e == hipErrorOutOfMemory) ? RPHIP_ENOMEM : RPHIP_EHIP;                          \
    }                                                                                          \
  } while (0)

#define RP_TRY(expr)        \
  do {                      \
    int _r = (expr);        \
    if (_r != RPHIP_OK) return _r; \
  } while (0)

// every entry point that reads deg / inv_deg of a matrix: a shard's degrees are partial until the caller's all-reduce
#define RP_NEED_DEGREES(X)                                                                                              \
  do {                                                                                                                  \
    if ((X)->deg_partial)                                                                                               \
      RP_FAIL(RPHIP_ESTATE, "the shard's degrees are not complete: all-reduce rphip_csr_degrees_device over the shards, " \
                            "then rphip_csr_finish_degrees");                                                           \
  } while (0)

// Diagnostic switches (RP_CHOL_TRACE, RP_INV_TRACE, RP_CHOL_MEASURE_SCALES, RP_N22_PLANES, RP_EASE_SCALE_ROWS, RP_K9_I8_KC): read from
// the environment only by the diagnostic build (make diag: -DRP_DIAG, librecpack_hip_diag.so); the shipped library has none.
#ifdef RP_DIAG
#include <stdlib.h>
static inline const char* rp_diag_env(const char* name) { return getenv(name); }
#else
static inline const char* rp_diag_env(const char*) { return nullptr; }
#endif

// ---- geometry ------------------------------------------------------------------------------------
constexpr int RP_WAVE = 64;
constexpr int RP_FIT_THREADS = 1024;      // 16 waves, one workgroup per CU (LDS-bound)
constexpr int RP_TILE_MAX = 25600;        // accumulator columns per LDS tile (100 KiB of u32/f32 + 50 KiB candidate list)
constexpr int RP_SORT_CAP = 512;          // entries of the in-LDS exact sort (6 KiB, overlays the histogram)
constexpr int RP_HIST_BINS = 2048;        // radix-select digit: 11 bits
constexpr int RP_K_MAX = 2048;            // largest effective K / Kp of one call

// ---- objects -------------------------------------------------------------------------------------
struct rp_buf {  // grow-only device scratch
  void* p = nullptr;
  size_t cap = 0;
};

// Every entry point that takes a context holds its lock for the duration of the call: a context (its stream, scratch buffers, block
// cache, timing slots - and the lazily built acceleration structures of its models: predict's tile offsets and quantised copy of B)
// may be used from several threads, whose calls are serialised.  Recursive: entry points call each other.
#define RP_ENTER(ctx)                                          \
  std::unique_lock<std::recursive_mutex> rp_enter_lock_;       \
  if (ctx) rp_enter_lock_ = std::unique_lock<std::recursive_mutex>((ctx)->api_mu)

struct rphip_ctx {
  std::recursive_mutex api_mu;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 0;
  size_t hbm_bytes = 0;
  char arch[64] = {0};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  rphip_timings tm;
  rp_buf scratch[16];  // 8..11: plane workspaces of the large products (rp_ease.hip, k_gemm_hp): main stream A/B, side stream A/B; 12..13: the
                       // update stream of the distributed solve (rp_ease_dist.hip)
  unsigned long long* dev_counters = nullptr;  // 8 x u64 on the device: [0] products done by the wave fit kernel, [1] its rows
  // pending event pairs are resolved lazily (no sync inside *_dev calls)
  struct pending_t { hipEvent_t a, b; int slot; };
  std::vector<pending_t> pending;
  std::vector<hipEvent_t> ev_pool;
  // device allocations of models / matrices: freed blocks are kept (up to cache_limit bytes) and handed out again
  // - a fit in a loop allocates nothing after its first pass (rp_dev_alloc / rp_dev_free)
  // (round 4) a cached block also serves SMALLER requests: best fit, split when the request is at least 1/16 of the block (the rest
  // stays cached; small requests never carve up a large block), neighbours of one hipMalloc region are joined again when freed;
  // only whole regions go back to the driver
  struct block_t { char* p; size_t bytes; char* region; size_t region_bytes; std::vector<hipEvent_t> freed; };
  struct live_t { size_t bytes; char* region; size_t region_bytes; };
  std::vector<block_t> free_blocks;
  std::unordered_map<void*, live_t> live_blocks;
  size_t cached_bytes = 0;
  size_t cache_limit = (size_t)1 << 62;  // set to half of the device's memory by rphip_ctx_create
  // predict options (rphip_ctx_set_option)
  int predict_exact64 = 0;   // 1: the wave predict kernel sums in 64 bits (exact) instead of the per-user-scaled 32 bits
  int predict_half = 1;      // dense models: K9 as two passes (fp16 rows for a candidate list, then exact f32 scores of the candidates)
  int predict_half_w = 11200;  // columns per tile of that first pass (0: the model's own tiling, one workgroup per CU)
  int predict_int8 = 1;      // ... on int8 rows (one scale per row) instead of fp16 ones: half the bytes again, a longer candidate list
  int predict_user_tasks = 1;  // ... with one task per USER (rp_predict_i8.hip) instead of one per (user, column tile) + a merge kernel
  int predict_second_list = 1;  // ... and users without a proof try once more with a list of 256 before the exact pass takes them
  int predict_second_list_hist = 1024;  // ... if their history is no longer than this (one workgroup per user: a long user would be the whole launch)
  int64_t predict_stats[6] = {-1, -1, 0, 0, 0, 0};  // the last single-GPU predict: users sent to the exact pass, users of the call (-1: not the two-pass route), Kc, 1 = per-user first pass, users given a second list, of them proven by it
  int64_t owner_stats[4] = {0, 0, 0, 0};  // the last rphip_predict_topk_owner: users without a proof (all ranks / this rank), Kc, loop-back: us of the stand-in for the others' parts
  int comm_broadcast_split = 1;  // a broadcast of >= 8 MiB among more than two ranks travels as scatter + all-gather (rp_comm_broadcast)
  int pw_wt32 = 4096;        // tile width of the 32-bit sums: 2048 or 4096 (tuning)
  int gemm_mode = 2;         // operands of the solve's products: 0 fp32 MFMA (cross-check), 2 two fp16 planes (3 products; default)
  int gemm_supertile = 0;    // 1: GEMM workgroups walk the tiles in XCD-round-robin 8 x 8 supertiles (0: row-major 2-D grid)
  int gemm_xcd_block = 4;    // k_gemm_hp16w: an XCD's 32 concurrent tiles form a block of this many tile rows x 32 / this tile columns (0: plain grid)
  int gemm_presplit = 1;     // mode 2: products with many tiles cut their operands into planes once (workspace) and run k_gemm_hp
  int gemm_wide = 1;         // the plane product kernel on 256 x 128 tiles with three LDS stages (k_gemm_hp16w) instead of 128 x 128 / two
  size_t gemm_ws_bytes = (size_t)4 << 30;  // workspace per operand of those products (sets their K chunk)
  int chol_sub = 8;          // tiles per sub-panel of the blocked Cholesky (0: one level of panels)
  int chol_panel = 0;        // tiles per panel of the blocked Cholesky (0: 32 from 640 tiles, else 16)
  int inv_panel = 16;        // tiles per panel of N = L^-T
  int inv_n22_early = 4;  // the diagonal-block chains of the inverse's first panels are issued during the factorisation, on a stream of their own (0: all ahead of the inverse's products, as until round 5)
  int64_t gemm_plane_cache = 1;  // the inverse keeps the planes of N = L^-T between its panels: 0 off, 1 as much as fits, > 1 at most that many bytes
  unsigned* gemm_amax_ring = nullptr;  // mode 2: [RP_AMAX_SLOTS][2] words, the operands' largest magnitudes per launch
  unsigned long long gemm_amax_seq = 0;
  int64_t dist_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // rphip_ease_fit_dist's last run on this context (rphip_ease_fit_dist_stats)
  int gemm_kc = 1024;        // K steps between two flushes of the MFMA accumulators into the compensated master sums (round 3: 1024;
                             // certified column error at C3 / C4, lambda 10 and 200: 512: <= 7.4e-6, 1024: <= 1.2e-5, 2048: <= 1.9e-5; -2 % time)
};

constexpr int RP_AMAX_SLOTS = 4096;

struct rphip_csr {
  rphip_ctx* ctx = nullptr;
  int64_t n_users = 0, n_items = 0, nnz = 0;
  bool owns_x = false;
  bool x_idx_padded = false;       // x_idx is the library's padded copy (matrices with fewer than 8 entries)
  const int32_t* x_ptr = nullptr;  // [U+1]
  const int32_t* x_idx = nullptr;  // [nnz]
  int32_t* xt_ptr = nullptr;       // [I+1]
  int32_t* xt_idx = nullptr;       // [nnz] users of each item (any order inside a row)
  int32_t* deg = nullptr;          // [I] item degrees
  double* inv_deg = nullptr;       // [I] fl(1/c_j) (0 for empty items)
  int64_t* row_cost = nullptr;     // [I] w_i = sum_{u in i} d_u
  int64_t n_products = -1;         // P = sum_u d_u^2
  // column tiling (n_items > RP_TILE_MAX): tile_ptr[u*(T+1)+t] = first position of row u with col >= t*W
  int n_tiles = 1;
  int tile_w = 0;
  int32_t* tile_ptr = nullptr;
  uint16_t* x_rel = nullptr;       // [nnz rounded up to 8, + 8] column of every entry RELATIVE to its tile (x_idx - tile * tile_w < 65536): what the
                                   // fit kernels' accumulate loops read - half the bytes and half the load instructions of x_idx
  bool sorted_rows = false;
  int max_user_deg = 0;            // longest history (bounds |score| for predict's fixed-point scale)
  int64_t xt_row_begin = 0, xt_row_end = -1;  // item rows of X^T that were filled (-1: all); fit is limited to them
  bool deg_partial = false;        // rphip_csr_from_device_shard: deg holds the rows [xt_row_begin, xt_row_end) only until the
                                   // caller's all-reduce + rphip_csr_finish_degrees
  // optional float64 values of the stored entries (rphip_csr_set_values): the reference's weighted matrices
  // (time decay, time_aware_item_knn/base.py:181-201).  NULL: the matrix is binary.
  double* x_val = nullptr;         // [nnz], order of x_idx
  double val_max_abs = 0.0;
  bool val_neg = false;            // some value is negative
};

struct rphip_model {
  rphip_ctx* ctx = nullptr;
  int kind = 0;
  int64_t n_items = 0;
  int64_t row_begin = 0, row_end = 0;
  int32_t K = 0;
  // TOPK
  int32_t* idx = nullptr;   // [rows*K]
  double* val = nullptr;    // [rows*K]
  float* val32 = nullptr;   // [rows*K] (predict operand)
  int2* pairs = nullptr;    // [rows*K] {column, float bits}: what the wave-per-user predict kernel streams
  mutable uint16_t* toff = nullptr; // [n_items * (n_wt + 1)] per-row offsets of the predict column tiles (built on first use)
  mutable int toff_wt = 0;          // tile width toff was built for
  mutable float* rowmax = nullptr;  // [n_items] largest |similarity| of each list (built with toff)
  int32_t* nnz = nullptr;   // [rows]
  bool cols_sorted = false; // every row's columns strictly increasing (checked when the model is finalised)
  float vmax_abs = 0.0f;    // largest |similarity| (f32): with the longest history it bounds |score|
  // DENSE
  float* B = nullptr;       // [n_items * ld]
  bool borrows_B = false;   // B belongs to the caller (rphip_model_from_dense_device)
  bool is_gram = false;     // B holds the plain Gram X^T X (rphip_ease_gram_create): input of rphip_ease_fit_gram, not a model to predict with
  int64_t ld = 0;
  int64_t col_begin = 0, n_cols = 0;  // n_cols > 0: B holds only the columns [col_begin, col_begin + n_cols) (rphip_model_from_dense_cols)
  // n_cols > 0 and cyc_nb > 0: the columns are block-cyclic panels of cyc_nb columns (rphip_ease_fit_dist): local column l is item
  // rp_cyc_panel(l / cyc_nb, cyc_rank, cyc_world) * cyc_nb + l % cyc_nb (below); the n_cols items are a prefix of the local columns
  int32_t cyc_nb = 0, cyc_world = 1, cyc_rank = 0;
  // (the `mutable` members are acceleration structures built on first use by predict under the owning context's lock, RP_ENTER)
  // fp16 copy of B for the first pass of the two-pass K9 (built on first use, rp_predict.hip): B16 = fp16(B * b16_scale), the same
  // leading dimension; b16_delta[i] = the largest rounding error of row i (in B's units), b16_rowmax[i] = its largest |entry|
  mutable _Float16* B16 = nullptr;   // (b16_kind 2: int8 rows, q = rint(b / b8_scale[row]) in [-127, 127], the same leading dimension in BYTES)
  mutable int b16_kind = 0;          // 1: fp16 copy, 2: int8 copy (round 4); the copy is rebuilt when the context asks for the other kind
  mutable float* b8_scale = nullptr; // int8 copy: the rows' steps (rowmax / 127)
  mutable float b16_scale = 0.0f;
  mutable float* b16_delta = nullptr;
  mutable float* b16_rowmax = nullptr;
  mutable bool b16_failed = false;  // the copy could not be allocated: the one-pass kernel is used
  mutable int64_t empty_rows = -1;  // rows without any stored similarity (-1: not counted)
  // rphip_ease_model_allgather (round 5): the int8 rows of the gathered model are complete when the call returns, its f32 rows
  // arrive on a stream of their own behind it - whoever reads B first waits for gather_ev (rp_model_gather_wait)
  mutable hipEvent_t gather_ev = nullptr;
  mutable hipStream_t gather_stream = nullptr;
  mutable void* gather_tmp[2] = {nullptr, nullptr};  // the f32 strips' staging, released with the event
  bool rows_i8_only = false;  // rphip_ease_model_allgather_i8: B is NULL - the int8 rows, their scales and bounds only (rphip_predict_topk_owner)
};

// B of a gathered model is about to be read: on_stream = by launches on the context's stream (they are ordered behind the gather),
// otherwise by the host (the call waits).  A no-op for every other model.   (rp_ease_dist.hip)
int rp_model_gather_wait(rphip_ctx* ctx, const rphip_model* m, bool on_stream);

// ---- helpers (rp_api.hip) ------------------------------------------------------------------------
int rp_scratch(rphip_ctx* ctx, int slot, size_t bytes, void** out);
int rp_scratch_release_planes(rphip_ctx* ctx);
int rp_dev_alloc(rphip_ctx* ctx, size_t bytes, void** out);  // hipMalloc through the context's block cache
void rp_dev_free(rphip_ctx* ctx, void* p);                   // NULL is fine; blocks not from rp_dev_alloc go to hipFree
void rp_dev_trim(rphip_ctx* ctx);                            // release every cached block (whole regions; parts of a split region stay)
int rp_dev_cache_serves(rphip_ctx* ctx, size_t bytes, int n);  // how many of n requests of that size the cache would serve
void rp_time_begin(rphip_ctx* ctx, hipEvent_t* a);
void rp_time_end(rphip_ctx* ctx, hipEvent_t a, int slot, double work);
int rp_resolve_timings(rphip_ctx* ctx);
int rp_set_device(const rphip_ctx* ctx);

// device-wide exclusive scan of int32 (n < 2^31, total < 2^31): out[i] = sum_{k<i} in[k], out[n] = total
int rp_exclusive_scan_i32(rphip_ctx* ctx, const int32_t* in, int32_t* out, int64_t n);

// rp_itemknn.hip / rp_itemknn_w.hip
constexpr int RP_PLAN_BUCKETS = 264;  // degree buckets of the row ordering: 0, then 8 per octave
int rp_model_alloc_topk(rphip_ctx* ctx, int64_t n_items, int64_t r0, int64_t r1, int32_t K, rphip_model** out);
int rp_plan_row_order(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int n_rows, int32_t* bucket, int32_t* order);
int rp_gram_weighted(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K, int32_t similarity,
                     const double* cp_colw_host, rphip_model** out);

// rp_csr.hip
int rp_csr_prepare(rphip_ctx* ctx, rphip_csr* m);

// rp_itemknn.hip: K4, dense f32 Gram rows [row_begin,row_end) of X^T X into G (row-major, leading dimension ld),
// diag_add on the diagonal; rows of items nobody interacted with are left untouched
// Xcols != NULL: rows of X^T Xcols (the user rows come from Xcols, a matrix of the same shape)
int rp_gram_dense(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, float* G, int64_t ld, float diag_add,
                  const rphip_csr* Xcols = nullptr);

// rp_predict.hip: val (f64) -> val32 + pairs, column-order check; call once the lists are complete
int rp_model_finalize_topk(rphip_ctx* ctx, rphip_model* m);
// out[u * (T + 1) + t] = first position of (column-sorted) row u with column >= t * W   (rp_itemknn_v.hip)
int rp_user_tile_ptr(rphip_ctx* ctx, const rphip_csr* X, int T, int W, int32_t* out);

static inline int64_t rp_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- the panel layout of the distributed EASE solve (rp_ease_dist.hip, rp_predict.hip, rp_ease.hip) --------------------------------
// 1-D block-cyclic with REFLECTED odd rounds (round 4): round q = panels [q W, (q + 1) W) of the matrix; rank r holds the panel at
// position r of an even round and W - 1 - r of an odd one, as its local panel q.  The solve works on the lower triangle, where panel
// p carries np - p tiles: plain cyclic (rank p mod W) gave rank 0 a third more update flop than rank W - 1 at W = 8 and 49 panels;
// reflected, every pair of rounds gives every rank the same count.  Local panels stay in the matrix's order.
__host__ __device__ inline int rp_cyc_pos(int q, int r, int W) { return (q & 1) ? W - 1 - r : r; }
__host__ __device__ inline int rp_cyc_panel(int q, int r, int W) { return q * W + rp_cyc_pos(q, r, W); }  // the matrix's panel of rank r's local panel q
__host__ __device__ inline int rp_cyc_owner(int p, int W) { return rp_cyc_pos(p / W, p % W, W); }          // (the reflection is its own inverse)
__host__ __device__ inline int rp_cyc_count(int np, int r, int W) {  // panels < np that rank r holds
  return np / W + ((np % W) && rp_cyc_pos(np / W, r, W) < np % W ? 1 : 0);
}

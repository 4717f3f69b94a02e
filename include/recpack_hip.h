/*
 * recpack_hip.h - C ABI of librecpack_hip.so: the MI355X (gfx950) implementation of recpack's
 * item-similarity hot path (ItemKNN cosine, EASE; fit + predict).
 *
 * The reference (LienM/recpack v0.3.6) is pure Python and has no FFI; its boundary for this path is the
 * class protocol Algorithm.fit(X)/predict(X) (recpack/algorithms/base.py:165-217).  The entry points
 * below are what a binding for that protocol needs; each cites the reference code it replaces
 * (paths relative to the reference root).  The ctypes binding that a recpack maintainer would add is in
 * INTEGRATION.md; the one this repo ships is recpack_amd/_lib.py.
 *
 * Conventions
 *  - every function returns 0 on success or a negative RPHIP_E* code; rphip_last_error() gives the
 *    thread-local message of the last failure;
 *  - plain pointers and sizes only; "host" pointers are caller-owned host memory, "dev" pointers are
 *    device memory on the context's GPU (e.g. a torch tensor's data_ptr());
 *  - indices are int32, row pointers int32 or int64 (flag); nnz must be < 2^31;
 *  - a context is bound to one GPU and one HIP stream.  Every entry point that takes a context holds the context's lock for the
 *    duration of the call (round 5): a context - and the models and matrices created on it, whose acceleration structures
 *    (predict's tile offsets, the quantised copy of a dense model) are built on first use - may be used from several threads, whose
 *    calls are serialised; host-output calls are synchronous on return, *_dev calls are ordered on the context's stream, with ONE
 *    exception: the fused predict of a DENSE model in two passes (option "predict_half", the default) reads back the number of
 *    users without a proof before it launches their exact pass, i.e. rphip_predict_topk*_dev waits for the stream once per call;
 *    rphip_model_destroy / rphip_csr_destroy take the same lock (round 6: a destructor running on a garbage collector's thread waits
 *    for the call in flight).  NOT covered: rphip_ctx_destroy and rphip_comm_destroy - the caller destroys a context or a
 *    communicator only when no call that uses it is in flight (a communicator may outlive its context and so cannot lock it);
 *  - objects (ctx, csr, model) are opaque and freed by their *_destroy.
 */
#ifndef RECPACK_HIP_H
#define RECPACK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPHIP_VERSION 100 /* 0.1.0 */

/* error codes */
#define RPHIP_OK 0
#define RPHIP_EINVAL (-1)     /* bad argument */
#define RPHIP_EHIP (-2)       /* HIP runtime failure (message has the hipError string) */
#define RPHIP_ENOMEM (-3)     /* device allocation failed */
#define RPHIP_ENODEV (-4)     /* no usable GPU */
#define RPHIP_EUNSORTED (-5)  /* column indices of a row are not strictly increasing */
#define RPHIP_ELIMIT (-6)     /* size outside what this build supports (see message) */
#define RPHIP_ESTATE (-7)     /* object in the wrong state for the call */
#define RPHIP_ENUMERIC (-8)   /* factorisation broke down (matrix not positive definite) */

typedef struct rphip_ctx rphip_ctx;
typedef struct rphip_csr rphip_csr;
typedef struct rphip_model rphip_model;

/* model kinds */
#define RPHIP_MODEL_TOPK 1  /* I x K neighbour lists (ItemKNN) */
#define RPHIP_MODEL_DENSE 2 /* dense I x I fp32 item-item matrix (EASE) */

/* timing slots of rphip_get_timings (milliseconds, HIP events on the context's stream) */
enum {
  RPHIP_T_CSR_TRANSPOSE = 0, /* K0: degree histogram + scan + scatter (X -> X^T) */
  RPHIP_T_FIT_PLAN = 1,      /* row costs, task list */
  RPHIP_T_GRAM_TOPK = 2,     /* K1-K3: row-blocked Gram + cosine key + per-row top-K (the SpGEMM) */
  RPHIP_T_TOPK_MERGE = 3,    /* merge of per-column-tile candidates (only when I > one LDS tile) */
  RPHIP_T_PREDICT = 4,       /* K8 (neighbour-list models) and the merges */
  RPHIP_T_GRAM_DENSE = 5,    /* K4 */
  RPHIP_T_CHOLESKY = 6,      /* K5 */
  RPHIP_T_INVERSE = 7,       /* K6 */
  RPHIP_T_EASE_SCALE = 8,    /* K7 */
  RPHIP_T_H2D = 9,
  RPHIP_T_D2H = 10,
  RPHIP_T_GRAM_FINALIZE = 11, /* second launch of K1-K3: sum of the partial tiles of sliced rows + top-K */
  RPHIP_T_GRAM_TOPK_WAVE = 12, /* K1-K3 for rows with <= 255 users: 256-thread workgroups, byte counters */
  RPHIP_T_SWEEP_UPDATE = 13,  /* rphip_ease_fit_dist: the symmetric rank-nb updates of the own panels (the n^3 / world part) */
  RPHIP_T_SWEEP_CHAIN = 14,   /* rphip_ease_fit_dist: pivot inverse, gather / broadcast of the pivot column, T, look-ahead updates */
  RPHIP_T_MIRROR = 15,        /* rphip_ease_fit_dist: the exchange that fills the upper triangle */
  RPHIP_T_PREDICT_DENSE = 16, /* K9 (dense models) */
  RPHIP_T_COUNT = 24
};

typedef struct rphip_timings {
  double ms[RPHIP_T_COUNT];       /* accumulated since the last rphip_reset_timings */
  int64_t launches[RPHIP_T_COUNT];
  double work[RPHIP_T_COUNT];     /* algorithmic bytes (HBM-bound slots) or flops (MFMA slots), same period */
} rphip_timings;

/* ---- library / context ------------------------------------------------------------------------- */
int rphip_version(void);
const char* rphip_last_error(void);
int rphip_device_count(int* n_out);

/* device: HIP ordinal.  stream: a hipStream_t to run on (e.g. torch's current stream) or NULL to
 * create a private one. */
int rphip_ctx_create(int device, void* stream, rphip_ctx** out);
int rphip_ctx_destroy(rphip_ctx* ctx);
int rphip_ctx_sync(rphip_ctx* ctx);

/* Device blocks of destroyed models / matrices are kept by the context (up to "cache_limit_bytes", default half of the device's memory) and
 * handed out again for a request of the same size, so a fit repeated in a loop allocates nothing after its first pass.  This is
 * not a nicety: the driver wipes freed memory in the background, and a hipMalloc that follows a hipFree of tens of GB waits for it
 * (measured on MI355X, tools/alloc_bench.hip: 40 GiB allocated right after 40 GiB were freed: 1.8 - 6 s; from clean memory 0.3 ms).
 * An EASE fit + predict at 100 000 items cycles through 40 GB (B), 40 GB (the inverse's plane cache) and 20 GB (the fp16 copy).
 * rphip_ctx_trim releases the blocks to the driver (e.g. before another library allocates most of the HBM). */
int rphip_ctx_trim(rphip_ctx* ctx);

/* Options of a context (name, integer value):
 *   "predict_exact64"   0 (default) | 1.  The fused predict + top-Kp of neighbour-list models (rphip_predict_topk*) sums
 *                       scores as integers.  Default: 32-bit sums with one power-of-two scale PER USER - a score is off by at
 *                       most d_u * M_u * 2^-30 (d_u = history length, M_u = the user's score bound, the sum of his history
 *                       lists' largest |similarity|), i.e. the size of the f32 rounding of the similarities and of the f32
 *                       output.  1: 64-bit sums with one scale per call - exact sums (the reference's X @ S, base.py:248, in
 *                       float64 differs from either only by the f32 storage of S).  Valued prediction matrices
 *                       (rphip_csr_set_values) always use the 64-bit sums.  Both are independent of the order of the adds:
 *                       predict is bit-reproducible run to run.
 *   "predict_tile32"    2048 | 4096 (default): column tile of the 32-bit sums (tuning).
 *   "predict_half"      1 (default) | 0.  Fused predict + top-Kp of DENSE models (EASE; X @ B of base.py:248 is HBM-bound on the rows
 *                       of B) in two passes: the rows are first read from an fp16 copy of B (built on first use, half the bytes) to
 *                       keep Kp + 28..91 candidates per user by approximate score; the candidates are then scored exactly (the f32
 *                       sums of the one-pass kernel, same order, bit for bit) and a per-user bound on |approximate - exact| - the
 *                       sum over the history of each row's largest rounding error - proves that no column outside the list can be
 *                       in the exact top-Kp; a user whose list cannot prove it (< 0.1 % measured) is served by the one-pass kernel.
 *                       The results are those of the one-pass kernel.  Costs n_items * ld * 2 bytes beside B; skipped (as with 0)
 *                       when that does not fit or Kp > 488.
 *   "predict_half_tile" columns per tile of that first pass (default 11200, at most 16384; 0 = the model's own tiling).  Tiles of
 *                       half the LDS let two workgroups share a CU: one streams rows while the other selects.
 *   "predict_int8"      1 (default) | 0.  The first pass of "predict_half" on an INT8 copy of B instead of the fp16 one (a quarter of
 *                       B's bytes: q = rint(b / s_row) in [-127, 127], s_row = the row's largest magnitude / 127; n_items * ld bytes
 *                       beside B).  The bound on |approximate - exact| is the sum of the rows' measured half-steps, 2^-8 of their
 *                       largest entries, so the candidate list is longer (Kp + 92 rounded up to 64: 192 at Kp = 100, which proves
 *                       98.5 % of the users of the 200K x 50K and 1M x 100K workloads; the others take the one-pass kernel) - the
 *                       results are still those of the one-pass kernel, bit for bit.  Needs "predict_half_tile" > 0; a context
 *                       that switches the option has the copy rebuilt on the next call.
 *   "predict_user_tasks"  1 (default) | 0.  The int8 first pass with one task per USER (round 5): a workgroup walks all column tiles of a
 *                       user, keeps a tile's sums in registers, hands the columns above a running lower bound of the user's Kc-th best
 *                       score to ONE candidate list in LDS and cuts it to Kc entries itself - no per-tile lists, no merge kernel.  0: one
 *                       task per (user, column tile) and a merge (round 4; also taken when history removal meets unsorted rows, and for
 *                       candidate lists longer than 256; until round 6 also with an item filter).  The results are identical either way.
 *   "predict_second_list"  1 (default) | 0 | n > 1.  With "predict_user_tasks": the users whose candidate list proved nothing (1 - 3 %:
 *                       history removal and item filters take the best columns away) go through both passes once more, alone, with a
 *                       list of 256 candidates, before the exact one-pass kernel takes who is left (it reads four times the bytes per
 *                       row).  Users with more than n history items (1: 1024) skip it: a workgroup walks a user alone, and among ten
 *                       thousand users a long one would be the whole launch.  The results are identical either way.
 *   "comm_broadcast_split"  1 (default) | 0.  Broadcasts of 8 MiB and more among MORE than two ranks travel as scatter + all-gather
 *                       (grouped point-to-point sends on RCCL: no link of a fully connected xGMI node carries more than 2 / W of the
 *                       message); 0: the transport's own broadcast for every size.  bench.py switches it off when operation 4 of
 *                       rphip_comm_selftest (the large broadcast) is the only one that fails.
 *   "gemm_mode"         2 (default) | 0: how the O(n^3) products of the EASE solve (and rphip_gemm_nt_f32) multiply their fp32 operands.
 *                       2: two fp16 planes per operand, both rounded to nearest (x = h1 + h2 to 2^-24 |x|), three plane products per
 *                          step on v_mfma_f32_16x16x32_f16; each operand is scaled by a power of two taken from its largest magnitude
 *                          over the rows and K range the launch reads (one extra pass over the operands), so fp16's range is never
 *                          left; fp32 accumulate.  Entries below 2^-18 of that magnitude are held to an absolute 2^-40 of it
 *                          instead of fp32's relative 2^-24 (rows of very different size: use mode 0).
 *                          0: the fp32-input MFMA (v_mfma_f32_32x32x2_f32): the cross-check.  Error of one product, relative to
 *                          sum |a||b|: the same for both (2e-8 .. 2.5e-7); speed on MI355X 360 / 113 TFLOP/s fp32-equivalent.
 *                          (Round 2's three-bf16-plane mode 1 was removed in round 4: slower than 2, no more exact than 0.)
 *   "gemm_presplit"     1 (default) | 0.  gemm_mode 2: products with at least 512 tiles cut their operands into planes ONCE per K chunk
 *                       (a workspace laid out in MFMA fragment order) and run a product kernel staged by LDS-DMA loads with no
 *                       conversion arithmetic (363 against 225 TFLOP/s fp32-equivalent on a square product); 0: every workgroup
 *                       converts the tiles it stages.
 *   "gemm_workspace_bytes"  bytes of plane workspace per operand of those products (default 4 GiB, at least 64 MiB): sets the K
 *                       chunk; a product on a side stream has a second pair.  rphip_ctx_trim releases them.
 *   "gemm_wide"         1 (default) | 0.  The plane product kernel on 256 x 128 tiles, eight waves, three LDS stages with counted
 *                       waits (k_gemm_hp16w; bit-identical results, +2..6 %) or on 128 x 128 tiles with two stages.
 *   "chol_sub_panel"    8 (default) | 0 .. 32 tiles.  The blocked Cholesky of the EASE solve works in panels of 32 tiles (16 below 640
 *                       tiles) and, inside a panel, in sub-panels of this many: a block's left-looking products reach back to its
 *                       sub-panel only, a finished sub-panel updates the rest of the panel in one wide product.  0 (or a value
 *                       that does not divide the panel) = one level.
 *   "chol_panel"        0 (default: 32 tiles from 640 tiles = 81 920 items, 16 below) | 8 | 16 | 32 | 64: tiles per panel of that Cholesky
 *                       (tuning; the trailing update of a finished panel is one product with K = 128 x this).
 *   "inv_panel"         16 (default) | 8 | 32: tiles per column panel of N = L^-T in the inverse (tuning).
 *   "inv_n22_early"     4 (default) | 0 | n.  The diagonal blocks of the inverse's first n panels - chains of small products that need
 *                       only the factor's rows inside the panel - are issued during the factorisation, as soon as it has passed the
 *                       panel, on a stream of their own: the inverse then starts with its products.  The other panels' chains run ahead
 *                       of the inverse's products as before (all of them early: the factorisation loses more than the inverse gains).
 *                       Same kernels, same results.
 *   "gemm_plane_cache"  1 (default) | 0 | bytes.  The inverse N = L^-T keeps the fp16 planes of the part of N it has finished
 *                       (4 bytes per element of an n x n square, from the context's block cache) instead of cutting it anew for
 *                       every panel: 1 = as much as the free memory holds (everything at 100 000 items; a prefix of the panels
 *                       when the matrix itself fills the GPU), a larger value = at most that many bytes, 0 = off.  One scale per
 *                       2048-column chunk of N instead of one per product: a fit with the whole cache, a part of it or none agrees
 *                       to the rounding of the plane products (<= 1e-6 of max |B|), not bit for bit - which is why the cache's
 *                       size is derived from the device's memory and the matrix, not from what happens to be free (if that much
 *                       cannot be allocated after all, the solve runs without the cache).
 *   "gemm_supertile"    0 (default) | 1.  Tile order of the GEMM's workgroups: supertiles of 64 tiles handed to the 8 XCDs round-robin
 *                       (each XCD's L2 then holds the 8 + 8 panels of an 8 x 8 block of tiles), or the plain row-major grid
 *                       (measured equal within 1 %: the products are not bound by operand traffic).
 *   "gemm_xcd_block"    4 (default) | 0 | 1 | 2 | 8 | 16 | 32: order of the wide plane product kernel's tiles.  Workgroups are dealt to the 8 XCDs round-robin
 *                       and an XCD (32 CUs, its own L2) runs 32 tiles at a time; with a value R > 0 those 32 tiles form a block of R tile
 *                       rows x 32 / R tile columns, so that an XCD's L2 serves R + 32 / R operand panels per K step instead of 17 - 33 (0: the
 *                       plain row-major grid).  Results are bit-identical.
 *   "gemm_flush_k"      K steps between two flushes of the MFMA accumulators into the compensated (hi + lo) master sums of the
 *                       EASE solve's GEMM (default 1024, a multiple of 16; a value beyond any K = plain fp32 accumulation).  Certified
 *                       column error at 50K / 100K items, l2 = 10 and 200: 512: <= 7.4e-6, 1024: <= 1.2e-5, 2048: <= 1.9e-5.
 *   "cache_limit_bytes" see rphip_ctx_trim. */
int rphip_ctx_set_option(rphip_ctx* ctx, const char* name, int64_t value);
int rphip_get_timings(rphip_ctx* ctx, rphip_timings* out);
int rphip_reset_timings(rphip_ctx* ctx);
/* counters of the last rphip_itemknn_fit (int64[8]): [0] scalar products done by the light-row kernel,
 * [1] its tasks (row, tile); the rest is reserved.  Used to split the SpGEMM's algorithmic bytes between its two kernels. */
int rphip_get_counters(rphip_ctx* ctx, int64_t* out8);
/* number of CUs, bytes of HBM, gfx arch string (into buf) */
int rphip_device_info(rphip_ctx* ctx, int* n_cu, int64_t* hbm_bytes, char* arch_buf, int arch_buf_len);

/* ---- interaction matrix --------------------------------------------------------------------------
 * Replaces InteractionMatrix.values -> to_csr_matrix(binary=True)
 * (recpack/matrix/interaction_matrix.py:202-217, recpack/matrix/util.py:27-48, recpack/util.py:99-109):
 * only the sparsity pattern is taken (binary), `data` is never shipped.
 * Builds X^T (the CSC view) on the device (K0) and the item degrees c = diag(X^T X).
 * indices inside a row may be in any order unless n_items exceeds one LDS tile (then they must be
 * increasing: RPHIP_EUNSORTED otherwise; scipy's canonical CSR satisfies this).
 */
int rphip_csr_from_host(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t nnz,
                        const void* indptr_host, int indptr_is_int64, const int32_t* indices_host,
                        rphip_csr** out);
/* same, from device arrays that are already resident (indptr int32[n_users+1], indices int32[nnz]);
 * the arrays are NOT copied and must outlive the object. */
int rphip_csr_from_device(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t nnz,
                          const int32_t* indptr_dev, const int32_t* indices_dev, rphip_csr** out);
/* same for one shard of a multi-GPU fit: X^T (the costly scattered writes of K0) is filled for the item rows
 * [fit_row_begin, fit_row_end) only, which is all rphip_itemknn_fit(row range inside it) reads; degrees, tile pointers
 * and predict are complete.  fit_row_end < 0 = all rows. */
int rphip_csr_from_device_rows(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t nnz, const int32_t* indptr_dev,
                               const int32_t* indices_dev, int64_t fit_row_begin, int64_t fit_row_end, rphip_csr** out);
/* one shard of a multi-GPU fit with the degree pass SHARDED as well: the histogram and the in-row ranks of K0 are computed for the
 * item rows [fit_row_begin, fit_row_end) only (1 / world of K0's atomics).  The handle is not usable for a fit until the caller
 * has summed the degree arrays of all shards - rphip_csr_degrees_device gives the int32[n_items] device array for ONE
 * all-reduce (RCCL over xGMI through torch.distributed in recpack_amd/dist.py::csr_for_shard; 4 * n_items bytes) - and called
 * rphip_csr_finish_degrees.  (The reference has no multi-GPU path: base.py:454-455.) */
int rphip_csr_from_device_shard(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t nnz, const int32_t* indptr_dev,
                                const int32_t* indices_dev, int64_t fit_row_begin, int64_t fit_row_end, rphip_csr** out);
int rphip_csr_degrees_device(rphip_csr* X, int32_t** deg_dev);
int rphip_csr_finish_degrees(rphip_ctx* ctx, rphip_csr* X);
/* from raw (user id, item id) pairs, int64 as pandas holds them, in any order and with repeats: what
 * InteractionMatrix.values builds from its DataFrame (recpack/matrix/interaction_matrix.py:202-217:
 * csr_matrix((ones, (uid, iid)), shape), duplicates summed) before the path binarises it - here the DISTINCT pairs,
 * rows sorted by item, built on the device (64-bit key sort + flagged compaction).  Ids outside the shape: RPHIP_EINVAL. */
int rphip_csr_from_coo_host(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t n_pairs, const int64_t* uid_host,
                            const int64_t* iid_host, rphip_csr** out);
int rphip_csr_from_coo_device(rphip_ctx* ctx, int64_t n_users, int64_t n_items, int64_t n_pairs, const int64_t* uid_dev,
                              const int64_t* iid_dev, rphip_csr** out);
/* the CSR arrays of a device matrix to the host: indptr int32[n_users+1], indices int32[nnz] */
int rphip_csr_download(rphip_ctx* ctx, const rphip_csr* m, int32_t* indptr_host, int32_t* indices_host);
int rphip_csr_destroy(rphip_csr* m);
int rphip_csr_shape(const rphip_csr* m, int64_t* n_users, int64_t* n_items, int64_t* nnz);
/* item degrees c_j (int32[n_items]) to host; sum_u d_u^2 (the number of scalar products P) */
int rphip_csr_item_degrees(rphip_ctx* ctx, const rphip_csr* m, int32_t* deg_host);
int rphip_csr_num_products(rphip_ctx* ctx, const rphip_csr* m, int64_t* p_out);

/* ---- ItemKNN fit ----------------------------------------------------------------------------------
 * Replaces ItemKNN._fit for similarity="cosine" (recpack/algorithms/nearest_neighbour.py:204-224):
 * compute_cosine_similarity (:69-84, sklearn cosine_similarity + setdiag(0)) and get_top_K_values
 * (recpack/util.py:50-96), fused: the I x I similarity matrix is never materialised.
 * Computes the neighbour lists of item rows [row_begin, row_end) (the shard of this GPU; the whole
 * matrix is 0..n_items).  Result: per row min(K, #co-occurring items) neighbours, sorted by column,
 * value c_ij / sqrt(c_i c_j) in float64.  Ties at the K-th position: larger similarity first, then
 * the SMALLER column index.
 */
int rphip_itemknn_fit(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K,
                      rphip_model** out);
/* Same with the similarity of ItemKNN's constructor (nearest_neighbour.py:172-202):
 *   RPHIP_SIM_COSINE                   as above;
 *   RPHIP_SIM_CONDITIONAL_PROBABILITY  sim(i,j) = c_ij / (c_i * c_j^pop_discount), compute_conditional_probability
 *                                      (nearest_neighbour.py:22-66) on the binarised matrix; pop_discount in [0,1].
 * normalize_X != 0 (nearest_neighbour.py:207-210): the binarised rows are l1-normalised first, X[u,i] = 1/d_u; the
 * Gram then is a float64 sum over the common users (cosine: sum 1/d_u^2 over sqrt(G_ii G_jj); conditional
 * probability: sum 1/d_u over c_i c_j^pop_discount).  Values agree with the reference to 1e-11 relative; the last
 * bits depend on the order of the adds, as they do in scipy. */
#define RPHIP_SIM_COSINE 0
#define RPHIP_SIM_CONDITIONAL_PROBABILITY 1
int rphip_itemknn_fit_ex(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K,
                         int32_t similarity, double pop_discount, int32_t normalize_X, rphip_model** out);
/* ---- valued matrices: the time-aware ItemKNN family --------------------------------------------------
 * TARSItemKNN (recpack/algorithms/time_aware_item_knn/base.py:34-201) weighs every interaction by a decay of its age
 * (_add_decay_to_interaction_matrix, :181-195; the host class computes the weights), fits cosine / conditional
 * probability / pearson on the weighted matrix (:166-179) and multiplies the weighted history with the neighbour
 * lists (:139-150).
 *
 * rphip_csr_set_values attaches float64 values to the stored entries of a matrix handle (values_host[k] belongs to
 * indices[k] as uploaded; NULL detaches them).  The binary fits above ignore them, as the reference's ItemKNN / EASE
 * binarise their input (base.py:141-151).  EVERY predict entry point multiplies a neighbour-list model with the valued
 * matrix when values are attached: score[u, j] = sum_i X[u, i] * S[i, j], each term the float64 product rounded once
 * into the fixed-point sum (same bit-reproducible sums as the binary case); a dense model refuses a valued matrix.
 *
 * rphip_itemknn_fit_values: G_ij = sum over the users u of item i that hold j of L_ui * X_uj with L = X
 * (left_binary = 0) or L = 1 (left_binary != 0: to_binary(X).T @ X, nearest_neighbour.py:45), then
 * S_ij = G_ij * row_scale[i] (NULL: 1), diagonal excluded, top K per row with get_top_K_values' handling of stored
 * zeros (recpack/util.py:80-96; see rp_itemknn_v.hip).  The host class passes
 *   cosine / pearson          columns of the (centred) weighted matrix scaled to unit length, left_binary = 0
 *                             (sklearn's cosine_similarity normalises first; nearest_neighbour.py:69-118)
 *   conditional probability   the weighted matrix, left_binary = 1, row_scale[i] = 1 / c_i   (:22-66, pop_discount 0)
 * Rows must be column-sorted.  Values agree with the reference to 1e-11 relative (float64 sums in a different order).
 */
int rphip_csr_set_values(rphip_ctx* ctx, rphip_csr* X, const double* values_host);
int rphip_itemknn_fit_values(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K,
                             int32_t left_binary, const double* row_scale_host, rphip_model** out);
/* TARSItemKNNCoocDistance._fit (time_aware_item_knn/base.py:260-294; TARSItemKNNXia, xia_2010.py): the values of X are the
 * last timestamps in units of decay_interval, all > 0 (the host drops the others, as the reference's `> 0` masks do):
 *   S_ij = row_scale[i] * sum over the users u of item i that hold j, t_uj != t_ui, of decay(|t_uj - t_ui|)
 * decay = one of decay_functions.py:24-252 with parameter `decay` and, for the bounded ones, max_distance (the largest
 * possible distance, base.py:268); pairs with equal timestamps add nothing (their difference is dropped by the sparse
 * subtraction, base.py:279).  row_scale NULL ("cooc") or 1 / number of users of item i ("conditional_probability").
 * Top K of the positive sums per row.
 */
#define RPHIP_DECAY_NONE 1
#define RPHIP_DECAY_EXPONENTIAL 2
#define RPHIP_DECAY_CONVEX 3
#define RPHIP_DECAY_CONCAVE 4
#define RPHIP_DECAY_LOG 5
#define RPHIP_DECAY_LINEAR 6
#define RPHIP_DECAY_INVERSE 7
int rphip_itemknn_fit_timedist(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K,
                               int32_t decay_kind, double decay, double max_distance, const double* row_scale_host,
                               rphip_model** out);
/* TARSItemKNNHermann._fit (time_aware_item_knn/hermann_2010.py:64-103), values of X = last timestamps in units of
 * decay_interval, all > 0, now = max timestamp + 1 / decay_interval (:70):
 *   S_ij = (1 / n_ij) * sum over the users u holding i and j with t_uj != t_ui of 1 / (|t_uj - t_ui| + now - min(t_ui, t_uj))
 * n_ij = number of users that hold both items (:97-98).  Top K of the positive entries per row. */
int rphip_itemknn_fit_hermann(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int32_t K, double now,
                              rphip_model** out);

/* neighbour lists to host: idx int32[rows*K] (-1 padded), val float64[rows*K] (0 padded), nnz int32[rows] */
int rphip_model_topk_download(rphip_ctx* ctx, const rphip_model* m, int32_t* idx_host, double* val_host,
                              int32_t* nnz_host);
/* device pointers of the same arrays (for collectives on them): idx int32, val float64, nnz int32 */
int rphip_model_topk_device(const rphip_model* m, void** idx_dev, void** val_dev, void** nnz_dev,
                            int64_t* rows, int32_t* K);
/* build a predict-ready model from assembled neighbour lists (all n_items rows) in host / device memory;
 * the arrays are copied. */
int rphip_model_from_topk_host(rphip_ctx* ctx, int64_t n_items, int32_t K, const int32_t* idx_host,
                               const double* val_host, const int32_t* nnz_host, rphip_model** out);
int rphip_model_from_topk_device(rphip_ctx* ctx, int64_t n_items, int32_t K, const int32_t* idx_dev,
                                 const double* val_dev, const int32_t* nnz_dev, rphip_model** out);
int rphip_model_destroy(rphip_model* m);
int rphip_model_info(const rphip_model* m, int32_t* kind, int64_t* n_items, int64_t* row_begin,
                     int64_t* row_end, int32_t* K);

/* ---- predict --------------------------------------------------------------------------------------
 * Replaces ItemSimilarityMatrixAlgorithm._predict, scores = X @ similarity_matrix_
 * (recpack/algorithms/base.py:237-255) for users [user_begin, user_end) of Xpred.
 *
 * rphip_predict_topk fuses the per-user top-Kp that the reference applies later in
 * MetricTopK.calculate (recpack/metrics/base.py:189, recpack/util.py:50-77) and, when
 * exclude_history != 0, the history removal of Pipeline (recpack/pipelines/pipeline.py:174-175).
 * Outputs (host): idx int32[n*Kp] (-1 padded), val float32[n*Kp], nnz int32[n]; entries of a row are
 * sorted by column.  Ties: larger score first, then smaller column.
 */
int rphip_predict_topk(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                       int64_t user_end, int32_t Kp, int32_t exclude_history, int32_t* idx_host,
                       float* val_host, int32_t* nnz_host);
int rphip_predict_topk_dev(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred,
                           int64_t user_begin, int64_t user_end, int32_t Kp, int32_t exclude_history,
                           int32_t* idx_dev, float* val_dev, int32_t* nnz_dev);
/* The same with an item filter applied to the scores before the top-Kp: only items with item_keep[j] != 0
 * (n_items bytes; NULL = every item) can be recommended.  This is ExcludeItems / SelectItems
 * (recpack/postprocessing/filters.py:58-101: X_pred.multiply(mask)) fused into the kernel, the step Pipeline runs
 * between predict and the metrics (recpack/pipelines/pipeline.py:170-180). */
int rphip_predict_topk_filtered(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred, int64_t user_begin,
                                int64_t user_end, int32_t Kp, int32_t exclude_history, const uint8_t* item_keep_host,
                                int32_t* idx_host, float* val_host, int32_t* nnz_host);
int rphip_predict_topk_filtered_dev(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred,
                                    int64_t user_begin, int64_t user_end, int32_t Kp, int32_t exclude_history,
                                    const uint8_t* item_keep_dev, int32_t* idx_dev, float* val_dev, int32_t* nnz_dev);
/* exact reference semantics (every non-zero score): two phases.  _count fills row_nnz_host
 * (int64[n]) and returns the total; _fill writes CSR indices/data (float64) given the caller's
 * indptr (exclusive scan of row_nnz). */
/* LIMITS (the reference accepts any K: base.py:302-304, util.py:63-66).  Every fit - rphip_itemknn_fit[_ex] with and without normalize_X,
 * rphip_itemknn_fit_values / _timedist / _hermann - takes any K at any item count (round 6: from K x column tiles = 32 768 the per-tile
 * lists hold min(K, tile) entries and are merged where they lie, rp_merge_big.h, the rows in chunks of bounded scratch -
 * ItemKNN(K=20000) at 100 000 items is a 48 GB model).  The Kp of rphip_predict_topk[_filtered][_dev]: any (round 6: beyond 2048, what the
 * fused kernels' LDS buffers hold, the full scores of predict() are formed per chunk of users and cut to Kp on the device - float64 sums
 * rounded to float32; models on a column slab return RPHIP_ELIMIT there); the caller's arrays hold n x Kp entries whatever the item count.
 * The distributed routes (rphip_predict_topk_owner / _dist, rphip_predict_merge_topk) keep Kp <= 2048. */
int rphip_predict_full_count(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred,
                             int64_t user_begin, int64_t user_end, int64_t* row_nnz_host, int64_t* total_out);
int rphip_predict_full_fill(rphip_ctx* ctx, const rphip_model* model, const rphip_csr* Xpred,
                            int64_t user_begin, int64_t user_end, const int64_t* indptr_host,
                            int32_t* indices_host, double* data_host);

/* ---- EASE fit -------------------------------------------------------------------------------------
 * Replaces EASE._fit (recpack/algorithms/ease.py:63-95): G = X^T X (dense), P = (G + l2 I)^-1 by
 * blocked Cholesky on MFMA tiles instead of np.linalg.inv (:82), B_ij = -P_ij / P_jj, B_ii = 0 (:85-86),
 * optional column weights c_j^-alpha (:88-90).  l2 is rounded to float32 first, as the reference does
 * (l2 * np.identity(dtype=float32), :82).  B stays on the device (fp32).
 */
int rphip_ease_fit(rphip_ctx* ctx, const rphip_csr* X, double l2, double alpha, rphip_model** out);

/* Sweep over l2 with the Gram computed once (pipelines grid-search l2, e.g. [10, 100, 1000],
 * recpack doc/source/guides.create_algorithm.rst:681; the reference recomputes X^T X in every EASE._fit,
 * ease.py:81).  rphip_ease_gram_create keeps G = X^T X (float32, no shift) on the device behind a model
 * handle that only rphip_ease_fit_gram, rphip_model_dense_device and rphip_model_destroy accept;
 * rphip_ease_fit_gram(l2, alpha) copies it, adds l2 to the diagonal and runs the solve and the scaling of
 * rphip_ease_fit: same result, bit for bit.  X must be the matrix the Gram was built from (its item
 * degrees feed the alpha weights).
 */
int rphip_ease_gram_create(rphip_ctx* ctx, const rphip_csr* X, rphip_model** out);
int rphip_ease_fit_gram(rphip_ctx* ctx, const rphip_csr* X, const rphip_model* gram, double l2, double alpha,
                        rphip_model** out);
/* EASE_XY (recpack/algorithms/experimental/ease_xy.py:61-96): reconstruct a second matrix y from X,
 * B = P (X^T y) - P diag(diag(P X^T y) / diag(P)) with P = (X^T X + l2 I)^-1, columns scaled by c_j^-alpha;
 * X and y are binary matrices of the same shape.  Three n x n float32 matrices during the fit. */
int rphip_ease_xy_fit(rphip_ctx* ctx, const rphip_csr* X, const rphip_csr* Y, double l2, double alpha, rphip_model** out);
/* dense B (row-major float32[n_items*n_items]) to host - the lazy similarity_matrix_ (ease.py:92) */
int rphip_model_dense_download(rphip_ctx* ctx, const rphip_model* m, float* B_host);
int rphip_model_dense_device(const rphip_model* m, void** B_dev, int64_t* n_items, int64_t* ld);
/* a dense model from a host matrix (row-major float32[n_items*n_items], copied): used after host-side edits of B
 * such as EASE's density pruning (ease.py:97-109) */
int rphip_model_from_dense_host(rphip_ctx* ctx, int64_t n_items, const float* B_host, rphip_model** out);
/* density pruning on the device (ease.py:97-109): keep the `keep` largest |B_ij| of a dense model, zero the rest
 * (keep = min(int(density * n * n), nnz) in the reference).  Entries equal to the threshold are kept up to the exact
 * count in no particular order, as np.argpartition does.  nnz_before / nnz_after may be NULL. */
int rphip_model_dense_prune(rphip_ctx* ctx, rphip_model* m, int64_t keep, int64_t* nnz_before, int64_t* nnz_after);
/* The caller has rewritten the matrix of a dense model that BORROWS its memory (rphip_model_from_dense_cols and the other borrowing
 * constructors): the quantised copy the two-pass predict built from the old values (option "predict_half") is dropped and rebuilt
 * on the next predict, and the empty-row count is taken again.  Without this call the proof of the second pass would use the old
 * rows' error bounds.  (No reference counterpart: recpack's similarity_matrix_ is an ordinary attribute.) */
int rphip_model_dense_changed(rphip_ctx* ctx, rphip_model* m);
/* number of similarity rows without any stored value (dense models; -1 if not counted): what
 * ItemSimilarityMatrixAlgorithm._check_fit_complete warns about (recpack/algorithms/base.py:257-279) */
int rphip_model_empty_rows(rphip_ctx* ctx, const rphip_model* m, int64_t* out);

/* The building block of K5/K6: C = alpha * A * B^T + beta * C in fp32 on MFMA tiles (v_mfma_f32_32x32x2_f32,
 * exact fp32 FMA chains).  A is M x K, B is N x K, C is M x N, all row-major device memory; M and N must be
 * multiples of 128, K of 16, lda/ldb of 4.  Stands where the reference relies on LAPACK/BLAS inside
 * np.linalg.inv (recpack/algorithms/ease.py:82); exported for tests and roofline measurements. */
int rphip_gemm_nt_f32(rphip_ctx* ctx, int64_t M, int64_t N, int64_t K, float alpha, const float* A_dev, int64_t lda,
                      const float* B_dev, int64_t ldb, float beta, float* C_dev, int64_t ldc);

/* ---- column slabs of a dense model (multi-GPU predict: SURVEY.md 8(e); the reference has no multi-GPU path, base.py:454-455) ----
 *   rphip_model_from_dense_cols   a model on a slab of item COLUMNS of B (all rows; row-major, caller-owned device memory, not
 *                                 copied: it must outlive the model): predict scores the slab's columns only
 *   rphip_predict_merge_topk      per-user merge of the ranks' local top-Kp lists (metrics/base.py:189 on the union)
 * (Round 2's torch-driven solve - rphip_ease_gram_rows, rphip_transpose_f32, rphip_spd_inverse_f32, rphip_ease_scale_cols - was
 * removed in round 4: the distributed solve lives in the library, rphip_ease_fit_dist below.) */
int rphip_model_from_dense_cols(rphip_ctx* ctx, int64_t n_items, int64_t col_begin, int64_t n_cols, float* B_dev, int64_t ld,
                                rphip_model** out);
int rphip_predict_merge_topk(rphip_ctx* ctx, int32_t n_lists, int32_t Kp, int64_t n_users, const int32_t* cand_idx_dev,
                             const float* cand_val_dev, const int32_t* cand_nnz_dev, int32_t* idx_dev, float* val_dev, int32_t* nnz_dev);


/* ---- multi-GPU inside the library: a communicator and the distributed EASE fit -----------------------------------------
 * One process per GPU.  An rphip_comm is a group of `world` ranks with stream-ordered collectives on device memory:
 *   rphip_comm_rccl_*   the library's own RCCL communicator (RCCL over xGMI; librccl is opened at run time, the same copy the
 *                       process already holds when torch loaded one).  Rank 0 draws the 128-byte id, the caller hands it to
 *                       every rank (any out-of-band channel: torch.distributed's store, MPI, a file), every rank creates.
 *   rphip_comm_from_ops the caller's own transport as a table of C callbacks (the tests run the same solve over gloo with the
 *                       ranks sharing one GPU).  A callback enqueues its operation on `stream` (a hipStream_t) or completes it
 *                       before returning; buffers are device pointers on the caller's device; returns 0 on success.
 * The reference has no multi-GPU path (recpack/algorithms/base.py:454-455, 686-690 leave it as a TODO). */
typedef struct rphip_comm rphip_comm;
typedef struct rphip_comm_ops {
  void* self;
  int32_t world, rank;
  /* buf (bytes) of rank `root` to every rank's buf */
  int (*broadcast)(void* self, void* buf, int64_t bytes, int32_t root, void* stream);
  /* recv = every rank's send (bytes_per_rank each), in rank order; send may not alias recv */
  int (*all_gather)(void* self, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
  /* rank r sends send + send_off[d] (send_bytes[d] bytes) to every rank d and receives recv_bytes[s] bytes from every rank s at
   * recv + recv_off[s]; the arrays have `world` entries, the entry for the rank itself is a local copy */
  int (*exchange)(void* self, const void* send, const int64_t* send_off, const int64_t* send_bytes, void* recv, const int64_t* recv_off,
                  const int64_t* recv_bytes, void* stream);
  /* element-wise sum of int32 buffers over the ranks, in place */
  int (*all_reduce_sum_i32)(void* self, void* buf, int64_t count, void* stream);
} rphip_comm_ops;
#define RPHIP_COMM_ID_BYTES 128
int rphip_comm_rccl_unique_id(void* id_out /* RPHIP_COMM_ID_BYTES */);
int rphip_comm_rccl_create(rphip_ctx* ctx, int32_t world, int32_t rank, const void* id, rphip_comm** out);
int rphip_comm_from_ops(rphip_ctx* ctx, const rphip_comm_ops* ops, rphip_comm** out);
int rphip_comm_destroy(rphip_comm* comm);
int rphip_comm_info(const rphip_comm* comm, int32_t* world, int32_t* rank, int32_t* is_rccl);
/* int32 sum over the ranks of a device vector, in place (the item degrees of a sharded K0: rphip_csr_degrees_device) */
int rphip_comm_all_reduce_sum_i32(rphip_ctx* ctx, rphip_comm* comm, int32_t* buf_dev, int64_t count);
/* Self-test of a communicator, ONE operation per call: op 0 = broadcast (from the last rank), 1 = all-gather, 2 = personalised
 * exchange, 3 = int32 all-reduce, 4 = the broadcast of a large message (12 MiB or more: the library sends it as scatter + all-gather so
 * that no xGMI link carries more than 2 / world of it), on `words` int32 per rank (per piece for the exchange).  The operation runs once on the context's
 * stream on a known pattern, the call waits for it and checks every received word on the device: *mismatches = words that differ
 * (0 = works), *ms = its wall time.  Every rank must make the same calls.  A transport that hangs hangs HERE, in a call that names
 * the operation: put a watchdog around it (bench.py --gpus N does, 30 s per operation, before its warm-up) rather than find out
 * in the first pivot of a solve.  The reference has no multi-GPU path (recpack/algorithms/base.py:454-455). */
int rphip_comm_selftest(rphip_ctx* ctx, rphip_comm* comm, int32_t op, int64_t words, int64_t* mismatches, double* ms);

/* Multi-GPU ItemKNN fit: every rank has fitted the item rows [row_bounds[r], row_bounds[r + 1]) (rphip_itemknn_fit* with a row
 * range; row_bounds has world + 1 entries, 0 .. n_items); the shards are exchanged with ONE all-gather through the communicator
 * (n_items * K * 12 bytes in all, padded to the longest shard) straight into a new model holding all rows, ready for predict
 * (the item rows of nearest_neighbour.py:204-224 are independent of each other; the reference has no multi-GPU path). */
int rphip_itemknn_gather(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, const int64_t* row_bounds, rphip_model** out);

/* EASE fit over the ranks of `comm` with the n x n matrix SHARDED in memory (replaces EASE._fit,
 * ease.py:63-95, whose np.linalg.inv is the n^3 step).  Layout: panels of nb = 128 * panel_tiles columns, 1-D block-cyclic with
 * reflected odd rounds (rphip_model_dense_layout: every pair of rounds gives every rank the same share of the lower triangle -
 * plain cyclic left rank 0 a third more update flop than the last rank at 8 ranks and 49 panels); a rank stores its
 * panels side by side, all n_pad rows (n_pad = n rounded up to nb), and only the lower triangle is worked on.
 * Solve: symmetric block Gauss-Jordan sweeps (Goodnight 1979) - for pivot panel K: P = A_KK^-1 (blocked Cholesky on MFMA
 * tiles, on the owner), V = the pivot's full column A[:, K] (the part left of the diagonal gathered from the row block K of
 * everybody's panels, the part below it broadcast by the owner with P), T_j = V_j P for the own panels j, then the symmetric
 * rank-nb update A_ij -= V_i T_j^T on the own lower tiles, A_iK = T_i, A_KK = -P.  n^3 / world flop per rank (the count of
 * potrf + trtri + lauum), n^2 * 4 bytes received per rank over the whole solve; the next pivot's chain (its panel update, the
 * pivot inverse, the gather / broadcast) runs beside the current update on other streams (look-ahead of one panel), nothing
 * is read back by the host before the end.  The finished lower triangle is mirrored across the ranks (one personalised
 * exchange per `world` panels), K7 runs on the panels.  panel_tiles <= 0: chosen by the library (16, fewer for small n).
 * The model holds this rank's columns of B (rphip_model_dense_layout); predict against it scores those columns only
 * (a local top-Kp per user; rphip_predict_merge_topk merges the ranks' lists).  *out = NULL on a rank that owns only padding. */
int rphip_ease_fit_dist(rphip_ctx* ctx, rphip_comm* comm, const rphip_csr* X, double l2, double alpha, int32_t panel_tiles,
                        rphip_model** out);
/* The ranks' column panels of rphip_ease_fit_dist into ONE full dense model on every rank (strips of rows, one all-gather each):
 * while 3 n^2 floats fit a GPU the fastest multi-GPU predict is the single-GPU rphip_predict_topk* on a share of the USERS - each
 * rank then streams full-width rows of B for its own users only.  `local`: this rank's model (NULL on a rank that owns only
 * padding); n_items and panel_cols as rphip_model_dense_layout reports them.  rphip_predict_topk_dist (below) stays for matrices
 * that do not fit.  The reference has no multi-GPU path (recpack/algorithms/base.py:454-455). */
int rphip_ease_model_allgather(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, int64_t n_items, int64_t panel_cols,
                               rphip_model** out);
/* (round 5) The gathered model's int8 rows ALONE - their scales and error bounds with them, n^2 bytes where the full model takes
 * 5 n^2: what rphip_predict_topk_owner needs on every rank.  The result's f32 matrix is absent: every other entry point refuses it
 * (RPHIP_ESTATE).  Needs the options predict_half and predict_int8 (the defaults) and more than one rank. */
int rphip_ease_model_allgather_i8(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, int64_t n_items, int64_t panel_cols,
                                  rphip_model** out);
/* Fused predict + top-Kp for this rank's users [user_bounds[rank], user_bounds[rank + 1]) (the same world + 1 bounds on every rank,
 * from 0 to n_users; the prediction matrix replicated) with the exact scores formed by the OWNERS of the columns: every rank runs
 * the int8 first pass for its users on `rows_i8` (rphip_ease_model_allgather_i8), the candidate lists are all-gathered, every rank
 * sums the f32 entries of the candidates whose column lies in `local` (its model of rphip_ease_fit_dist; NULL on a rank that owns
 * only padding) for the users of ALL ranks - history order, one add per row: the single-GPU sums - one personalised exchange returns
 * them, the user's rank ranks and proves its list; users without a proof are scored exactly by every rank on its own columns and
 * their local lists merged on their rank.  The result is the single-GPU rphip_predict_topk's (base.py:248 + metrics/base.py:189) bit
 * for bit; no rank ever holds the f32 matrix.  idx / val / nnz (device): n * Kp / n entries for the rank's n users.  RPHIP_ELIMIT
 * outside the first pass's range (Kp > 164, fewer than 4 * (Kp + 92) items, history removal on unsorted rows) and RPHIP_ENOMEM (agreed
 * on by all ranks) when the buffers do not fit: rphip_predict_topk_dist serves those. */
int rphip_predict_topk_owner(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, const rphip_model* rows_i8, const rphip_csr* Xpred,
                             const int64_t* user_bounds, int32_t Kp, int32_t exclude_history, int32_t* idx_dev, float* val_dev,
                             int32_t* nnz_dev);
/* Fused predict + top-Kp with the model's columns spread over the ranks of `comm` (the models of rphip_ease_fit_dist; `model` NULL
 * on a rank that owns only padding): every rank scores the users against its own columns, the local lists travel to the rank
 * that owns the user (one personalised exchange per chunk of users) and are merged there by (score desc, column asc) - the
 * single-GPU result (base.py:248 + metrics/base.py:189).  Rank r ends with the users [r * per, (r + 1) * per), per =
 * ceil(n_users / world): *user_begin_out, *n_users_out say which; idx / val / nnz (device) must hold per * Kp / per entries. */
int rphip_predict_topk_dist(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* model, const rphip_csr* Xpred, int32_t Kp,
                            int32_t exclude_history, int64_t* user_begin_out, int64_t* n_users_out, int32_t* idx_dev, float* val_dev,
                            int32_t* nnz_dev);
/* Layout of a dense model's columns.  panel > 0 (rphip_ease_fit_dist): block-cyclic panels of `panel` columns with reflected odd
 * rounds - local panel q = l / panel is the matrix's panel q * world + rank for even q, q * world + (world - 1 - rank) for odd q
 * (every pair of rounds gives every rank the same share of the lower triangle), item = that panel * panel + l % panel.  Otherwise
 * col_begin + l.  n_local_cols of the local columns are items (the rest is padding). */
int rphip_model_dense_layout(const rphip_model* m, int64_t* n_local_cols, int64_t* panel, int32_t* world, int32_t* rank, int64_t* col_begin);
#ifdef __cplusplus
}
#endif
#endif /* RECPACK_HIP_H */

// The NT product of the EASE solve and its callers' view of it (rp_ease.hip implements, rp_ease_dist.hip drives it too).
#pragma once
#include "rp_common.h"

constexpr int GT = 128;   // tile (and diagonal block) size
constexpr int GBK = 16;   // K step

// Planes of an operand kept between products.  Every panel of the inverse multiplies by ALL of N = L^-T found so far: cut anew
// per product that was 0.65 TB of reads and writes per 100 000-item solve (k_split_planes + k_gemm_amax: 0.6 s of kernel time).
// The cache holds k_split_planes' layout for rows [0, K) x columns [0, K) of the operand with one power-of-two scale per chunk
// of w columns; a finished panel is cut into it once.
struct rp_plane_cache {
  unsigned char* W = nullptr;  // [plane 2][32-row block K/32][K16 step K/16][1 KB]
  long long ps = 0;            // plane stride in bytes
  int K = 0;                   // rows and columns covered (a multiple of 128)
  int w = 0;                   // chunk width: a power of two
  unsigned* amax = nullptr;    // [ceil(K / w)] largest magnitude of each chunk (fp32 bits)
};

struct rp_gemm_args {
  const float* A;
  long long lda;
  const float* B;
  long long ldb;
  float* C;
  long long ldc;
  int M, N;          // multiples of 128
  int k0, k1;        // K range, multiples of 16
  int kb_m, kb_n;    // k_begin = max(k0, kb_m * (m_off + m0), kb_n * (n_off + n0))
  int m_off, n_off;
  int ke_n;          // k_end = min(k1, ke_n * (n_off + n0 + 128)): B is lower triangular (row n holds columns <= n)
  int tri;           // 0: all tiles; 1: only tiles with n_off + n0 <= m_off + m0; 2: strictly below the diagonal;
                     // 3: the diagonal tiles only (grid.x = 1), written to C as a stack of blocks [M][128] (ldc = 128);
                     // 4 (the wide plane product, both operands cached): as 1, the diagonal tiles written to the stack Cd instead of C
  float* Cd;         // tri = 4: [tile][128][128]
  int n_skip0, n_skip1;  // tiles whose first column n0 lies in [n_skip0, n_skip1) are not computed (0, 0: every tile the tri rule leaves)
  int m_skip0, m_skip1;  // ... and tiles (128-row halves of the wide kernel's) whose first row m0 lies in [m_skip0, m_skip1)
  float alpha, beta;
  // split-K (skinny products with few tiles and a long K): blockIdx.z = s handles K chunk s and writes its partial
  // tile to part[s][M][N] (alpha = 1, beta = 0); k_splitk_reduce applies alpha / beta to C afterwards
  float* part;
  int k_chunk;
  int kc;            // K steps between two flushes into the master sums (set by rp_gemm_nt from the context: GKC by default)
  const unsigned* amax;  // k_gemm_nt_h2: device words [max |A|, max |B|] over the ranges read (k_gemm_amax); set by rp_gemm_nt
  int sw;            // tiles per row of a supertile (64 tiles: sw x 64/sw), 0 = plain row-major 2-D grid; set by rp_gemm_nt
  // C's columns are LOCAL columns of a block-cyclic layout (rp_ease_dist.hip): panels of cyc_nb columns, panel p of the matrix on
  // rank rp_cyc_owner(p) at local panel p / cyc_w (rp_common.h).  Local column l = cyc_l0 + n0 of rank cyc_r is then global column
  // rp_cyc_panel(l / cyc_nb, cyc_r, cyc_w) * cyc_nb + l % cyc_nb, which is what the `tri` tests compare with the row.  0 = off.
  int cyc_nb, cyc_w, cyc_r, cyc_l0;
  int no_splitk;     // never split K (the product runs beside others that use the shared partial-sum scratch)
  int ws_slot;       // scratch slots ws_slot, ws_slot + 1 hold this product's plane workspaces (0: 8 / 9 on the context's stream, 10 / 11 elsewhere)
  const rp_plane_cache* ca;  // A's planes are read from this cache (A unused; M and k1 <= ca->K; the wide product kernel only)
  int cb;            // ... and B's as well (B is the same matrix)
  int f32;           // this product on the fp32 MFMA kernel (k_gemm_nt) whatever the context's gemm_mode
  // (round 4) the scales of a chain's products without a measuring pass: amax_known = device words [max |A|, max |B|] that already
  // bound what the launch reads (no memset, no k_gemm_amax); amax_out / amax_out2 = pairs of words that take max |C| of what this
  // launch writes (atomicMax on both words of each; the fp32 kernel only) - the producer of a panel's columns measures them
  const unsigned* amax_known;
  unsigned* amax_out;
  unsigned* amax_out2;
};

// C = alpha * A * B^T + beta * C on `st` (the context's stream when NULL)
int rp_gemm_nt(rphip_ctx* ctx, const rp_gemm_args& g, hipStream_t st = nullptr, float* part_at = nullptr);
// In-place inverse of the SPD matrix A (n_pad x n_pad, a multiple of 128; reads the lower triangle).  info_dev == NULL: the
// call waits for the result and fails with RPHIP_ENUMERIC when a pivot is not positive; otherwise nothing is waited for and a
// failing pivot's index + 1 is left in *info_dev (device int, not cleared here).
// `on`: the stream to run on (the context's when NULL); lower_only: the upper half of the result is NOT mirrored from the lower one
// (the caller's next pass does it: k_ease_scale_sym)
int rp_spd_inverse(rphip_ctx* ctx, float* A, int n_pad, long long ld, int* info_dev = nullptr, hipStream_t on = nullptr, bool lower_only = false);
// rows [r0, r1) of G + lam I (identity block on the padding rows >= n_items) into G (row-major, ld), all n_cols columns
int rp_ease_gram_rows(rphip_ctx* ctx, const rphip_csr* X, int64_t r0, int64_t r1, int64_t n_cols, float lam, float* G, int64_t ld);
int rp_transpose_f32(rphip_ctx* ctx, const float* src, int64_t lds, int64_t rows, int64_t cols, float* dst, int64_t ldd, hipStream_t st = nullptr);
int rp_dense_count_empty(rphip_ctx* ctx, rphip_model* m);

/* recpack_hip_diag.h - measuring instruments and test hooks of librecpack_hip.so.
 *
 * NOT part of the drop-in boundary (include/recpack_hip.h): nothing here stands for a function of the reference; recpack_amd's
 * estimators never call these.  They are exported by the same library so that tools/ and tests/ can reach single building blocks:
 *   rphip_comm_loopback_create   one rank of an N-rank run measured alone on one GPU (tools/dist_rank_emulate.py)
 *   rphip_ease_fit_dist_stats    byte / flop counters of the last distributed fit (tests: memory bound of the sharded solve)
 *   rphip_potrf_trtri_128        the 128 x 128 diagonal-block kernel alone (tools/potrf_check.py, tests/test_gpu_ease.py)
 *   rphip_model_adopt_gather     hand a pending panel gather to another model (tools/dist_rank_emulate.py)
 *   rphip_gemm_nt_f32_ex         the NT product with the triangular K clipping of the solve (tests/test_gpu_ease.py)
 *   rphip_predict_stats          users the two-pass predict sent to the exact pass in the last call (tests/test_gpu_ease.py)
 * The environment switches RP_CHOL_TRACE, RP_INV_TRACE, RP_CHOL_MEASURE_SCALES, RP_N22_PLANES, RP_EASE_SCALE_ROWS, RP_K9_I8_KC
 * exist only in the diagnostic build (make -C recpack_amd/csrc diag -> librecpack_hip_diag.so, selected with RECPACK_HIP_LIB).
 */
#ifndef RECPACK_HIP_DIAG_H
#define RECPACK_HIP_DIAG_H
#include "recpack_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A LOOP-BACK communicator for measuring ONE rank of a `world`-rank run on a single GPU (tools/dist_rank_emulate.py): every
 * operation is stream-ordered and asynchronous like the RCCL ones, moves no data between ranks - what a rank would receive is
 * filled with its own data or small noise of the right size - and holds its stream for latency_us + bytes / gbytes_per_s, where
 * gbytes_per_s is the rate of ONE link of a fully connected node (xGMI: every pair of GPUs has its own) and bytes what the busiest
 * link carries: a broadcast its whole message, an all-gather one rank's part, an exchange the largest piece to or from one peer.
 * The results of a run on it are meaningless; its kernel sequence, launch for launch, is rank `rank`'s.  The reference has no multi-GPU path (recpack/algorithms/base.py:454-455 TODO). */
int rphip_comm_loopback_create(rphip_ctx* ctx, int32_t world, int32_t rank, double gbytes_per_s, double latency_us, rphip_comm** out);

/* numbers of the last rphip_ease_fit_dist on this context: [0] slab bytes, [1] peak bytes of everything else the fit held
 * (panel buffers, staging, plane workspaces), [2] bytes received through the communicator, [3] bytes sent, [4] pivots,
 * [5] panel width, [6] n_pad, [7] flop of the rank's products */
int rphip_ease_fit_dist_stats(rphip_ctx* ctx, int64_t* out8);

/* Cholesky factor (lower) of a 128 x 128 block in place and its inverse into W (ld 128), in float64 internally;
 * *info_dev (device int32) receives pivot + 1 if the block is not positive definite. */
int rphip_potrf_trtri_128(rphip_ctx* ctx, float* A_dev, int64_t lda, float* W_dev, int32_t* info_dev);
/* rphip_gemm_nt_f32 with per-tile K clipping for triangular operands (see rp_ease.hip). */
int rphip_gemm_nt_f32_ex(rphip_ctx* ctx, int64_t M, int64_t N, int64_t k_begin, int64_t k_end, float alpha, const float* A_dev,
                         int64_t lda, const float* B_dev, int64_t ldb, float beta, float* C_dev, int64_t ldc, int32_t k_from_row,
                         int32_t k_from_col, int64_t m_off, int64_t n_off, int32_t lower_only);


/* `dst` takes over the pending f32 gather of `src` (a model of rphip_ease_model_allgather whose f32 rows are still arriving on their
 * own stream): tools/dist_rank_emulate.py predicts on the real matrix of a single-GPU fit while the emulated rank's gather - of noise -
 * is in flight, and that predict's rescoring has to wait for it as the rank's own would. */
int rphip_model_adopt_gather(rphip_model* dst, rphip_model* src);

/* The last rphip_predict_topk_owner of this context: out4[0] users without a proof over all ranks (they took the exact pass on every
 * rank's columns), [1] this rank's, [2] candidates per user, [3] on a loop-back communicator: microseconds of the stand-in that forms the other ranks'
 * parts from the f32 matrix passed with the int8 rows (tools/dist_rank_emulate.py subtracts it). */
int rphip_predict_owner_stats(rphip_ctx* ctx, int64_t* out4);

/* The last rphip_predict_topk* call of this context on one GPU: out6[0] users the two-pass predict could not prove from its candidate
 * lists (they were served by the exact one-pass kernel: same results, three times the traffic), [1] users of the call, [2] candidates per
 * user, [3] 1 when the first pass was the one-task-per-user kernel (rp_predict_i8.hip), [4] users that were given a second list (256
 * candidates), [5] those of them it proved.  [0] = [1] = -1: the call did not take the two-pass route.
 * (tests/test_gpu_ease.py: a tiling whose last wave owns four threads must not send every user to the exact pass - round 6.) */
int rphip_predict_stats(rphip_ctx* ctx, int64_t* out6);

#ifdef __cplusplus
}
#endif
#endif /* RECPACK_HIP_DIAG_H */

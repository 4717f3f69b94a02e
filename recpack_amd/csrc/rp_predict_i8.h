// K9's first pass with one task per user (rp_predict_i8.hip); driven by rp_predict_topk_impl (rp_predict.hip).
#pragma once
#include "rp_common.h"

struct rp_i8_params {
  const int32_t* x_ptr;   // prediction input (CSR, binary); rows ascending when exclude_history
  const int32_t* x_idx;
  const int8_t* B8;       // int8 rows of the model (leading dimension ld, in bytes), entry = q * b8_scale[row]
  long long ld;
  const float* b8_scale;
  int n_items;            // columns of the model (or of the rank's slab)
  int n_tiles, tile_w;    // rp_predict_user_i8_geometry
  int col_begin;          // dense column slab: column 0 is item col_begin
  int cyc_nb, cyc_w, cyc_r;  // cyc_nb > 0: block-cyclic panels (rp_common.h)
  int exclude_history;
  const uint8_t* keep;    // optional item filter [items of the whole model]: only items with keep[j] != 0 are candidates
  int user_begin, n_users;
  const int32_t* user_list;  // optional: the pass serves users user_begin + user_list[k], k < n_users; outputs are indexed by k
  int max_hist;           // > 0: a user with a longer history is not served (appended to the fail list): the second list leaves the long ones to the exact pass, whose tasks are (user, tile)
  int Kc;                 // candidates per user for the rescoring (<= 256)
  int32_t* out_idx;       // [n_users][Kc] global column ids (row k: the k-th user of the pass)
  float* out_val;         // [n_users][Kc] approximate scores
  int32_t* out_nnz;       // [n_users]; -1 = the user's list overflowed (he is on the fail list)
  int32_t* fail_list;
  int32_t* n_fail;
  int fail_base;
  int32_t* ticket;        // (set by the launcher)
};

size_t rp_predict_user_i8_lds();
int rp_predict_user_i8_geometry(int64_t n_cols, int* n_tiles, int* tile_w);
// one launch over all users of P; ticket_dev = one device word of scratch
int rp_predict_user_i8(rphip_ctx* ctx, const rp_i8_params& P, int32_t* ticket_dev, bool leave_room = false);

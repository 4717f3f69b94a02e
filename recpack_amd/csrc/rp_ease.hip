// EASE (K4-K7, K9) - placeholder translation unit until the dense path lands.
#include "rp_common.h"

extern "C" int rphip_ease_fit(rphip_ctx* ctx, const rphip_csr* X, double l2, double alpha, rphip_model** out) {
  (void)ctx; (void)X; (void)l2; (void)alpha;
  if (out) *out = nullptr;
  RP_FAIL(RPHIP_ESTATE, "rphip_ease_fit: the dense EASE path is not part of this build yet");
}

extern "C" int rphip_model_dense_download(rphip_ctx* ctx, const rphip_model* m, float* B_host) {
  (void)ctx; (void)m; (void)B_host;
  RP_FAIL(RPHIP_ESTATE, "rphip_model_dense_download: the dense EASE path is not part of this build yet");
}

extern "C" int rphip_model_dense_device(const rphip_model* m, void** B_dev, int64_t* n_items, int64_t* ld) {
  (void)m; (void)B_dev; (void)n_items; (void)ld;
  RP_FAIL(RPHIP_ESTATE, "rphip_model_dense_device: the dense EASE path is not part of this build yet");
}

// MFMA EPS kernels (placeholder until the first MFMA family lands): everything is routed to the
// generic kernels.
#include "common.h"

int eps_fwd_mfma(const void*, const void*, void*, const EpsP&, int, int, hipStream_t) {
  return DCTN_ERR_UNSUPPORTED;
}
size_t eps_bwd_mfma_workspace(const EpsP&, int, int, int, int) { return 0; }
int eps_bwd_mfma(const void*, const void*, const void*, void*, void*, void*, size_t, const EpsP&,
                 int, int, hipStream_t) {
  return DCTN_ERR_UNSUPPORTED;
}

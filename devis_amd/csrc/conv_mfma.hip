// MFMA implicit-GEMM convolution kernels (bf16, channels-last).  Shapes they do not take
// return DEVIS_ERR_UNSUPPORTED and the caller routes to the direct kernels.
#include "common.h"

namespace devis {

int conv_fwd_mfma(const devis_conv_desc*, const void*, const void*, const void*, const float*, const float*,
                  const float*, const void*, void*, hipStream_t) { return DEVIS_ERR_UNSUPPORTED; }
int conv_dgrad_mfma(const devis_conv_desc*, const void*, const void*, void*, void*, hipStream_t) { return DEVIS_ERR_UNSUPPORTED; }
size_t conv_wgrad_mfma_workspace(const devis_conv_desc*) { return 0; }
int conv_wgrad_mfma(const devis_conv_desc*, const void*, const void*, const float*, const float*, const void*,
                    float*, float*, void*, size_t, hipStream_t) { return DEVIS_ERR_UNSUPPORTED; }

}  // namespace devis

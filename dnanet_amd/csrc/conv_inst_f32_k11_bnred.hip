// instantiations: conv_igemm<float, 1, 1> as the transposed convolution's data gradient with BatchNorm backward's sums (CONV_EPI_BNRED)
#include "conv_igemm.h"
int conv_igemm_run_f32_k11_bnred(const ConvParams& p, int ht, int cot, hipStream_t s) {
  if (ht == 5 && cot == 32) return conv_igemm_launch<float, 1, 1, 5, 32, CONV_EPI_BNRED>(p, s);
  if (ht == 5 && cot == 64) return conv_igemm_launch<float, 1, 1, 5, 64, CONV_EPI_BNRED>(p, s);
  if (ht == 2 && cot == 128) return conv_igemm_launch<float, 1, 1, 2, 128, CONV_EPI_BNRED, 2, 2>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm: no instantiation for ht=%d cot=%d", ht, cot);
}

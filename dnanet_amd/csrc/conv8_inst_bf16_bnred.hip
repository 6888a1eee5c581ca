// instantiations: conv_igemm8<__bf16> (3x5, 8-wave pipelined) as a data gradient with BatchNorm backward's sums (CONV_EPI_BNRED)
#include "conv_igemm8.h"
int conv_igemm8_run_bf16_bnred(const ConvParams& p, int ht, hipStream_t s) {
  if (ht == 5) return conv_igemm8_launch<__bf16, 5, false, CONV_EPI_BNRED>(p, s);
  if (ht == 3) return conv_igemm8_launch<__bf16, 3, true, CONV_EPI_BNRED>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm8: no instantiation for ht=%d", ht);
}

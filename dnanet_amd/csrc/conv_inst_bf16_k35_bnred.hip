// instantiations: conv_igemm<__bf16, 3, 5> as a data gradient that also forms BatchNorm backward's sums (CONV_EPI_BNRED)
#include "conv_igemm.h"
int conv_igemm_run_bf16_k35_bnred(const ConvParams& p, int ht, int cot, hipStream_t s) {
  if (ht == 5 && cot == 32) return conv_igemm_launch<__bf16, 3, 5, 5, 32, CONV_EPI_BNRED>(p, s);
  if (ht == 5 && cot == 64) return conv_igemm_launch<__bf16, 3, 5, 5, 64, CONV_EPI_BNRED>(p, s);
  if (ht == 3 && cot == 32) return conv_igemm_launch<__bf16, 3, 5, 3, 32, CONV_EPI_BNRED>(p, s);
  if (ht == 3 && cot == 64) return conv_igemm_launch<__bf16, 3, 5, 3, 64, CONV_EPI_BNRED, 3>(p, s);
  if (ht == 6 && cot == 32) return conv_igemm_launch<__bf16, 3, 5, 6, 32, CONV_EPI_BNRED, 3>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm: no instantiation for ht=%d cot=%d", ht, cot);
}

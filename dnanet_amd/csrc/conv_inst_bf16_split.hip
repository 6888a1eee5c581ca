// instantiations: the hi + lo split epilogues (bf16x3 inference mode) of the 4-wave kernels, 3x5, 1x3 and 1x1
#include "conv_igemm.h"
#define DN_SPLIT_K35(HT_, COT_)                                                                           \
  if (ht == HT_ && cot == COT_)                                                                           \
    return p.pool ? conv_igemm_launch<__bf16, 3, 5, HT_, COT_, CONV_EPI_POOL_SPLIT>(p, s)                \
                  : conv_igemm_launch<__bf16, 3, 5, HT_, COT_, CONV_EPI_SPLIT>(p, s);
int conv_igemm_run_bf16_k35_split(const ConvParams& p, int ht, int cot, hipStream_t s) {
  DN_SPLIT_K35(5, 32) DN_SPLIT_K35(5, 64) DN_SPLIT_K35(3, 32) DN_SPLIT_K35(3, 64) DN_SPLIT_K35(6, 32)
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm (split): no instantiation for ht=%d cot=%d", ht, cot);
}
int conv_igemm_run_bf16_k13_split(const ConvParams& p, int ht, int cot, hipStream_t s) {
  if (ht == 5 && cot == 32) return p.pool ? conv_igemm_launch<__bf16, 1, 3, 5, 32, CONV_EPI_POOL_SPLIT>(p, s) : conv_igemm_launch<__bf16, 1, 3, 5, 32, CONV_EPI_SPLIT>(p, s);
  if (ht == 5 && cot == 64) return p.pool ? conv_igemm_launch<__bf16, 1, 3, 5, 64, CONV_EPI_POOL_SPLIT>(p, s) : conv_igemm_launch<__bf16, 1, 3, 5, 64, CONV_EPI_SPLIT>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm (split): no instantiation for ht=%d cot=%d", ht, cot);
}
int conv_igemm_run_bf16_k11_split(const ConvParams& p, int ht, int cot, hipStream_t s) {
  if (ht == 5 && cot == 32) return conv_igemm_launch<__bf16, 1, 1, 5, 32, CONV_EPI_SPLIT>(p, s);
  if (ht == 5 && cot == 64) return conv_igemm_launch<__bf16, 1, 1, 5, 64, CONV_EPI_SPLIT>(p, s);
  if (ht == 2 && cot == 128) return conv_igemm_launch<__bf16, 1, 1, 2, 128, CONV_EPI_SPLIT, 2, 2>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm (split): no instantiation for ht=%d cot=%d", ht, cot);
}

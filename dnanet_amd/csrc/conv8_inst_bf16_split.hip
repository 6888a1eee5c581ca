// instantiations: the hi + lo split epilogues (bf16x3 inference mode) of the 8-wave 3x5 kernel, 5-row tiles
#include "conv_igemm8.h"
int conv_igemm8_run_bf16_split(const ConvParams& p, int ht, hipStream_t s) {
  if (ht == 5) return p.pool ? conv_igemm8_launch<__bf16, 5, false, CONV_EPI_POOL_SPLIT>(p, s) : conv_igemm8_launch<__bf16, 5, false, CONV_EPI_SPLIT>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm8 (split): no instantiation for ht=%d", ht);
}

// instantiations: conv_igemm8<__bf16> (3x5, 8-wave pipelined): 5-row tiles step-pipelined, 3-row tiles chunk-pipelined
#include "conv_igemm8.h"
int conv_igemm8_run_bf16(const ConvParams& p, int ht, hipStream_t s) {
  if (ht == 5) return p.stats ? conv_igemm8_launch<__bf16, 5, false, CONV_EPI_STATS>(p, s) : p.pool ? conv_igemm8_launch<__bf16, 5, false, CONV_EPI_POOL>(p, s) : conv_igemm8_launch<__bf16, 5>(p, s);
  if (ht == 3) return p.stats ? conv_igemm8_launch<__bf16, 3, true, CONV_EPI_STATS>(p, s) : p.pool ? conv_igemm8_launch<__bf16, 3, true, CONV_EPI_POOL>(p, s) : conv_igemm8_launch<__bf16, 3, true>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm8: no instantiation for ht=%d", ht);
}

// generated instantiation: conv_igemm<__bf16, 3, 5>
#include "conv_igemm.h"
int conv_igemm_run_bf16_k35(const ConvParams& p, int ht, int cot, hipStream_t s) {
  if (ht == 5 && cot == 32) return p.stats ? conv_igemm_launch<__bf16, 3, 5, 5, 32, CONV_EPI_STATS>(p, s) : p.head_w ? conv_igemm_launch<__bf16, 3, 5, 5, 32, CONV_EPI_HEAD>(p, s) : p.pool ? conv_igemm_launch<__bf16, 3, 5, 5, 32, CONV_EPI_POOL>(p, s) : conv_igemm_launch<__bf16, 3, 5, 5, 32>(p, s);
  if (ht == 5 && cot == 64) return p.stats ? conv_igemm_launch<__bf16, 3, 5, 5, 64, CONV_EPI_STATS>(p, s) : p.head_w ? conv_igemm_launch<__bf16, 3, 5, 5, 64, CONV_EPI_HEAD>(p, s) : p.pool ? conv_igemm_launch<__bf16, 3, 5, 5, 64, CONV_EPI_POOL>(p, s) : conv_igemm_launch<__bf16, 3, 5, 5, 64>(p, s);
  if (ht == 3 && cot == 32) return p.stats ? conv_igemm_launch<__bf16, 3, 5, 3, 32, CONV_EPI_STATS>(p, s) : p.head_w ? conv_igemm_launch<__bf16, 3, 5, 3, 32, CONV_EPI_HEAD>(p, s) : p.pool ? conv_igemm_launch<__bf16, 3, 5, 3, 32, CONV_EPI_POOL>(p, s) : conv_igemm_launch<__bf16, 3, 5, 3, 32>(p, s);
  if (ht == 3 && cot == 64) return p.stats ? conv_igemm_launch<__bf16, 3, 5, 3, 64, CONV_EPI_STATS, 3>(p, s) : p.head_w ? conv_igemm_launch<__bf16, 3, 5, 3, 64, CONV_EPI_HEAD>(p, s) : p.pool ? conv_igemm_launch<__bf16, 3, 5, 3, 64, CONV_EPI_POOL>(p, s) : conv_igemm_launch<__bf16, 3, 5, 3, 64>(p, s);
  if (ht == 6 && cot == 32) return p.stats ? conv_igemm_launch<__bf16, 3, 5, 6, 32, CONV_EPI_STATS, 3>(p, s) : p.head_w ? conv_igemm_launch<__bf16, 3, 5, 6, 32, CONV_EPI_HEAD>(p, s) : p.pool ? conv_igemm_launch<__bf16, 3, 5, 6, 32, CONV_EPI_POOL>(p, s) : conv_igemm_launch<__bf16, 3, 5, 6, 32>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm: no instantiation for ht=%d cot=%d", ht, cot);
}

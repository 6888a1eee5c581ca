// generated instantiation: conv_igemm<__bf16, 1, 3>
#include "conv_igemm.h"
int conv_igemm_run_bf16_k13(const ConvParams& p, int ht, int cot, hipStream_t s) {
  if (ht == 5 && cot == 32) return p.stats ? conv_igemm_launch<__bf16, 1, 3, 5, 32, true>(p, s) : conv_igemm_launch<__bf16, 1, 3, 5, 32>(p, s);
  if (ht == 5 && cot == 64) return p.stats ? conv_igemm_launch<__bf16, 1, 3, 5, 64, true>(p, s) : conv_igemm_launch<__bf16, 1, 3, 5, 64>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm: no instantiation for ht=%d cot=%d", ht, cot);
}

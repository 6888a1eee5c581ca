// generated instantiation: conv_igemm<__bf16, 1, 1>
#include <cstdlib>
#include "conv_igemm.h"
int conv_igemm_run_bf16_k11(const ConvParams& p, int ht, int cot, hipStream_t s) {
  if (ht == 5 && cot == 32) return conv_igemm_launch<__bf16, 1, 1, 5, 32>(p, s);
  if (ht == 5 && cot == 64) return conv_igemm_launch<__bf16, 1, 1, 5, 64>(p, s);
  // two K planes per barrier: a plane is only 8 MFMAs per wave here
  const bool kp1 = dn_tune().k11_kp1;                           // tuning aid: one plane per barrier
  if (ht == 2 && cot == 128) return kp1 ? conv_igemm_launch<__bf16, 1, 1, 2, 128>(p, s)
                                        : conv_igemm_launch<__bf16, 1, 1, 2, 128, CONV_EPI_PLAIN, 2, 2>(p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_igemm: no instantiation for ht=%d cot=%d", ht, cot);
}

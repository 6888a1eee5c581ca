// generated instantiation: conv_wgrad<__bf16, 1, 3>
#include <algorithm>
#include "conv_wgrad.h"
int conv_wgrad_run_bf16_k13(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s) {
  return conv_wgrad_launch<__bf16, 1, 3>(p, slab_bytes, nslab, s);
}

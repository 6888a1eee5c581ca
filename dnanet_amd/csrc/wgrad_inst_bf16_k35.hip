// generated instantiation: conv_wgrad<__bf16, 3, 5>
#include <algorithm>
#include "conv_wgrad.h"
int conv_wgrad_run_bf16_k35(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s) {
  return conv_wgrad_launch<__bf16, 3, 5>(p, slab_bytes, nslab, s);
}

// generated instantiation: conv_wgrad<float, 1, 3>
#include <algorithm>
#include "conv_wgrad.h"
int conv_wgrad_run_f32_k13(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s) {
  return conv_wgrad_launch<float, 1, 3>(p, slab_bytes, nslab, s);
}

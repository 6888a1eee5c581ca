// instantiations: conv_wgrad<__bf16, 3, 5> with 5-, 3- and 1-row strips
#include <algorithm>
#include "conv_wgrad.h"
int conv_wgrad_run_bf16_k35(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s) {
  // same row-tile choice as the forward convolution (api.hip run_igemm): 6-dye profiles take 3-row tiles
  if (p.co_pad <= 32) {            // at most 32 output channels (wgrad_api.hip pads to the tile it expects here)
    if (p.H % 5 == 0) return conv_wgrad_launch<__bf16, 3, 5, 5, 32>(p, slab_bytes, nslab, s);
    if (p.H % 3 == 0) return conv_wgrad_launch<__bf16, 3, 5, 3, 32>(p, slab_bytes, nslab, s);
    return conv_wgrad_launch<__bf16, 3, 5, 1, 32>(p, slab_bytes, nslab, s);
  }
  if (p.H % 5 == 0) return conv_wgrad_launch<__bf16, 3, 5, 5>(p, slab_bytes, nslab, s);
  if (p.H % 3 == 0) return conv_wgrad_launch<__bf16, 3, 5, 3>(p, slab_bytes, nslab, s);
  return conv_wgrad_launch<__bf16, 3, 5, 1>(p, slab_bytes, nslab, s);
}

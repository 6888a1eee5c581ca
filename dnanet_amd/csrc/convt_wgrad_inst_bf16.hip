// instantiations: convt_wgrad<__bf16> -- 256 x 256, 128 x 128 and 64 x 64 tiles of dW
#include <algorithm>
#include "convt_wgrad.h"
int convt_wgrad_run_bf16(ConvtWgradParams& p, int cout, long slab_bytes, int* nslab, hipStream_t s) {
  const int rows = 2 * p.dy_nplanes * 16, cols = p.x_nplanes * 16;
  // tile = the largest one both dimensions fill (the shipped net: 512 x 512, 256 x 256, 128 x 128, 64 x 64 outputs)
  if (rows >= 256 && cols >= 256) return convt_wgrad_launch<__bf16, 256, 256, 4, 2, 1, 32, 1>(p, cout, slab_bytes, nslab, s);
  if (rows >= 128 && cols >= 128) return convt_wgrad_launch<__bf16, 128, 128, 2, 2, 1, 32, 2>(p, cout, slab_bytes, nslab, s);
  return convt_wgrad_launch<__bf16, 64, 64, 1, 1, 4, 64, 2>(p, cout, slab_bytes, nslab, s);
}

// instantiations: convt_wgrad<float> -- 256 x 256, 128 x 128 and 64 x 64 tiles of dW
#include <algorithm>
#include "convt_wgrad.h"
int convt_wgrad_run_f32(ConvtWgradParams& p, int cout, long slab_bytes, int* nslab, hipStream_t s) {
  const int rows = 2 * p.dy_nplanes * 8, cols = p.x_nplanes * 8;
  if (rows >= 256 && cols >= 256) return convt_wgrad_launch<float, 256, 256, 4, 2, 1, 16, 1>(p, cout, slab_bytes, nslab, s);
  if (rows >= 128 && cols >= 128) return convt_wgrad_launch<float, 128, 128, 2, 2, 1, 16, 2>(p, cout, slab_bytes, nslab, s);
  return convt_wgrad_launch<float, 64, 64, 1, 1, 4, 16, 2>(p, cout, slab_bytes, nslab, s);
}

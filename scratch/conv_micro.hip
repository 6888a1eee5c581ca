// micro-benchmark of conv_igemm_kernel<bf16,3,5,5,COT> on a shallow layer, with ablations (-DCV_ABLATE_*)
#include <cstdio>
#include <cstdarg>
#include <algorithm>
#include "conv_igemm.h"
int dn_set_error(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); return code; }
#ifndef CIN
#define CIN 32
#endif
#ifndef COUT
#define COUT 32
#endif
#ifndef WID
#define WID 4096
#endif
int main() {
  const int N = 256, H = CM_H, W = WID;
  const long xrecs = (long)N * (CIN / 16) * H * W, yrecs = (long)N * (COUT / 16) * H * W;
  void *x, *y, *w; float* b;
  hipMalloc(&x, xrecs * 32); hipMalloc(&y, yrecs * 32); hipMalloc(&w, (long)(CIN / 16) * 15 * COUT * 32); hipMalloc(&b, COUT * 4);
  hipMemset(x, 0x11, xrecs * 32); hipMemset(w, 0x22, (long)(CIN / 16) * 15 * COUT * 32); hipMemset(b, 0, COUT * 4);
  ConvParams p{};
  p.x = x; p.x_planes_total = CIN / 16; p.x_plane0 = 0; p.x_nplanes = CIN / 16; p.x_gplanes = 1 << 30; p.x_wmul = 1;
  p.w = w; p.bias = b;
  p.y = y; p.y_planes_total = COUT / 16; p.y_plane0 = 0; p.y_nplanes = COUT / 16; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.N = N; p.H = H; p.W = W; p.nchunk = CIN / 16; p.cout_pad = COUT; p.relu = 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) conv_igemm_launch<__bf16, 3, 5, CM_HT, (COUT % 64 == 0 ? 64 : 32)>(p, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    if (rep) printf("%s %d->%d W=%d: %.1f us  %.1f TF/s  %.0f GB/s\n", VARIANT, CIN, COUT, W, ms * 1e3,
                    2.0 * N * H * W * CIN * COUT * 15 / ms / 1e9, (xrecs + yrecs) * 32 / ms / 1e6);
  }
}

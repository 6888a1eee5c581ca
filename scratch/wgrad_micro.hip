// micro-benchmark of conv_wgrad_kernel<bf16,3,5> with ablations (build 3 times with -D flags)
#include <cstdio>
#include <cstdarg>
#include <algorithm>
#include <vector>
#include "conv_wgrad.h"
int dn_set_error(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); return code; }
int main() {
  const int N = 256, H = WG_H, W = 256, C = 512;
  const long recs = (long)N * (C / 16) * H * W;
  void *dy, *a; float* slab;
  hipMalloc(&dy, recs * 32); hipMalloc(&a, recs * 32); hipMalloc(&slab, 256L << 20);
  {  // random bf16 in [-2, 2): constant data clocks ~20 % higher than real activations
    std::vector<unsigned short> hbuf(recs * 16);
    unsigned st = 12345u;
    for (auto& v : hbuf) { st = st * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((st >> 9) & 0x4ffu) + ((st >> 31) << 15)); }
    hipMemcpy(dy, hbuf.data(), recs * 32, hipMemcpyHostToDevice);
    for (auto& v : hbuf) { st = st * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((st >> 9) & 0x4ffu) + ((st >> 31) << 15)); }
    hipMemcpy(a, hbuf.data(), recs * 32, hipMemcpyHostToDevice);
  }
  WgradParams p{};
  p.dy = dy; p.dy_pt = C / 16; p.dy_p0 = 0; p.dy_nplanes = C / 16; p.dy_gplanes = 1 << 30; p.dy_wmul = 1;
  p.a = a; p.a_pt = C / 16; p.a_p0 = 0; p.a_nplanes = C / 16; p.slab = slab; p.N = N; p.H = H; p.W = W; p.co_pad = C; p.ci_pad = C;
  int nslab;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) conv_wgrad_launch<__bf16, 3, 5, WG_HT>(p, 256L << 20, &nslab, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    if (rep) printf("%s: %.1f us  %.1f TF/s (nslab %d)\n", VARIANT, ms * 1e3, 2.0 * N * H * W * C * C * 15 / ms / 1e9, nslab);
  }
  return 0;
}

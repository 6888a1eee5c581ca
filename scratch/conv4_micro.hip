// micro-benchmark of the 4-wave conv_igemm_kernel<bf16,3,5,HT,64> on the 512->512 bottleneck layer (random data),
// with -DC8_PROFILE_WAIT: share of wave time between "barrier + stage + wait + barrier" and the rest, shader clock
#include <cstdio>
#include <cstring>
#include <cstdarg>
#include <vector>
#include <algorithm>
#include "conv_igemm.h"
int dn_set_error(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); return code; }
#ifndef HROWS
#define HROWS 6
#endif
#ifndef HTILE
#define HTILE 3
#endif
int main() {
  const int N = 256, H = HROWS, W = 256, CIN = 512, COUT = 512;
  const long xrecs = (long)N * (CIN / 16) * H * W, yrecs = (long)N * (COUT / 16) * H * W;
  const long wbytes = (long)(CIN / 16) * 15 * COUT * 32;
  void *x, *y, *w; float* b;
  hipMalloc(&x, xrecs * 32); hipMalloc(&y, yrecs * 32); hipMalloc(&w, wbytes); hipMalloc(&b, COUT * 4);
  std::vector<unsigned short> hx(xrecs * 16), hw(wbytes / 2);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f; unsigned u = __builtin_bit_cast(unsigned, f); return (unsigned short)(u >> 16); };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd();
  hipMemcpy(x, hx.data(), xrecs * 32, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice); hipMemset(b, 0, COUT * 4);
  ConvParams p{};
  p.x = x; p.x_planes_total = CIN / 16; p.x_plane0 = 0; p.x_nplanes = CIN / 16; p.x_gplanes = 1 << 30; p.x_wmul = 1;
  p.w = w; p.bias = b;
  p.y = y; p.y_planes_total = COUT / 16; p.y_plane0 = 0; p.y_nplanes = COUT / 16; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.N = N; p.H = H; p.W = W; p.nchunk = CIN / 16; p.cout_pad = COUT; p.relu = 1;
#ifdef C8_PROFILE_WAIT
  const long nw = (long)N * ((H + HTILE - 1) / HTILE) * 2 * 8 * 4 + 64; long long* dbg; hipMallocManaged(&dbg, nw * 3 * 8); memset(dbg, 0, nw * 3 * 8); p.dbg = dbg;
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) conv_igemm_launch<__bf16, 3, 5, HTILE, 64>(p, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
#ifdef C8_PROFILE_WAIT
    if (rep) { hipDeviceSynchronize(); double a = 0, wl = 0, c = 0; long cnt = 0;
      for (long i = 0; i < nw; ++i) if (dbg[3*i+2]) { a += dbg[3*i]; wl += dbg[3*i+1]; c += dbg[3*i+2]; ++cnt; }
      printf("waves %ld: main loop %.0f cycles, stage+wait+barriers %.1f%%, shader clock %.0f MHz\n", cnt, c / cnt, 100 * a / c, c / wl * 100.0); }
#endif
    if (rep) printf("conv4 H=%d ht%d %d->%d W=%d: %.1f us  %.1f TF/s nominal\n", H, HTILE, CIN, COUT, W, ms * 1e3, 2.0 * N * H * W * CIN * COUT * 15 / ms / 1e9);
  }
}

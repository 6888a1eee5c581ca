// micro-benchmark of conv_igemm8_kernel<bf16,5> on the 512->512 bottleneck layer (random data)
#include <cstdio>
#include <cstring>
#include <cstdarg>
#include <vector>
#include <algorithm>
#include "conv_igemm8.h"
int dn_set_error(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); return code; }
int main() {
  const int N = 256, H = 5, W = 256, CIN = 512, COUT = 512;
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
  const long nw = 2048L * 8; long long* dbg; hipMallocManaged(&dbg, 2 * nw * 3 * 8); memset(dbg, 0, 2 * nw * 3 * 8); p.dbg = dbg;
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) conv_igemm8_launch<__bf16, 5>(p, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
#ifdef C8_PROFILE_WAIT
    if (rep) { hipDeviceSynchronize(); double a = 0, b = 0, c = 0; double wl = 0; for (long i = 0; i < nw; ++i) { a += dbg[3*i]; b += (double)(dbg[3*i+1] >> 24); wl += (double)(dbg[3*i+1] & 0xffffff); c += dbg[3*i+2]; }
      printf("shader clock during the main loop: %.0f MHz (clock64 / wall_clock64 at 100 MHz)\n", c / wl * 100.0);
      { double pr = 0, e1 = 0, e2 = 0; const long long* d2 = dbg + 3 * nw; for (long i = 0; i < nw; ++i) { pr += d2[3*i]; e1 += d2[3*i+1]; e2 += d2[3*i+2]; }
        printf("per wave: prologue %.0f cycles, epilogue %.0f issued / %.0f acknowledged\n", pr / nw, e1 / nw, e2 / nw); }
      printf("per wave: main loop %.0f cycles, vmcnt wait %.1f%%, barrier %.1f%%\n", c / nw, 100 * a / c, 100 * b / c); }
#endif
    if (rep) printf("conv8 %d->%d W=%d: %.1f us  %.1f TF/s nominal\n", CIN, COUT, W, ms * 1e3, 2.0 * N * H * W * CIN * COUT * 15 / ms / 1e9);
  }
}

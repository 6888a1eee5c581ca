// Shader-clock probe: one wave samples clock64() (shader clock) against wall_clock64() (constant 100 MHz)
// while the wgrad kernel runs on another stream, to see what the chip clocks at under MFMA + staging load.
#include <cstdio>
#include <cstdarg>
#include <algorithm>
#include "conv_wgrad.h"
int dn_set_error(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); return code; }
__global__ void probe(long long* out, int nsamples, long long ticks_per_sample) {
  long long w0 = wall_clock64();
  for (int i = 0; i < nsamples; ++i) {
    while (wall_clock64() - w0 < ticks_per_sample * i) __builtin_amdgcn_s_sleep(8);
    out[2 * i] = wall_clock64(); out[2 * i + 1] = clock64();
  }
}
int main() {
  const int N = 256, H = 5, W = 256, C = 512;
  const long recs = (long)N * (C / 16) * H * W;
  void *dy, *a; float* slab; long long* samples;
  hipMalloc(&dy, recs * 32); hipMalloc(&a, recs * 32); hipMalloc(&slab, 256L << 20);
  const int NS = 120; hipMallocManaged(&samples, NS * 16);
  hipMemset(dy, 0x11, recs * 32); hipMemset(a, 0x22, recs * 32);
  WgradParams p{};
  p.dy = dy; p.dy_pt = C / 16; p.dy_p0 = 0; p.dy_nplanes = C / 16; p.dy_gplanes = 1 << 30; p.dy_wmul = 1;
  p.a = a; p.a_pt = C / 16; p.a_p0 = 0; p.a_nplanes = C / 16; p.slab = slab; p.N = N; p.H = H; p.W = W; p.co_pad = C; p.ci_pad = C;
  int nslab; hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  for (int i = 0; i < 3; ++i) conv_wgrad_launch<__bf16, 3, 5, WG_HT>(p, 256L << 20, &nslab, s1);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s2, samples, NS, 10000LL);   // a sample every 100 us, 12 ms in all
  for (int i = 0; i < 4; ++i) conv_wgrad_launch<__bf16, 3, 5, WG_HT>(p, 256L << 20, &nslab, s1);
  hipDeviceSynchronize();
  for (int i = 1; i < NS; ++i) {
    double dt = (samples[2 * i] - samples[2 * i - 2]) / 100.0;   // us
    double mhz = (samples[2 * i + 1] - samples[2 * i - 1]) / dt;
    printf("%s%.0f", i % 20 == 1 ? "\n" : " ", mhz);
  }
  printf("\n");
}

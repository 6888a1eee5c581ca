// micro-benchmark of conv_first_mfma_kernel<bf16,3,5,6> (256 x 6 x 4096 -> 32 channels) with ablations:
//   -DCF_ABLATE_NO_LOAD  -DCF_ABLATE_NO_MFMA  -DCV_ABLATE_NO_STORE
#include <cstdio>
#include <cstdarg>
#include <algorithm>
#include "conv_first_mfma.hip"
int dn_set_error(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); return code; }
int main() {
  const int N = 256, H = 6, W = 4096, C = 32;
  float *x, *wf, *bias; void* y;
  hipMalloc(&x, (size_t)N * H * W * 4); hipMalloc(&wf, 15 * 32 * 4); hipMalloc(&bias, 32 * 4); hipMalloc(&y, (size_t)N * 2 * H * W * 32);
  hipMemset(x, 0x3c, (size_t)N * H * W * 4); hipMemset(wf, 0x3c, 15 * 32 * 4); hipMemset(bias, 0, 128);
  ConvParams p{};
  p.bias = bias; p.y = y; p.y_planes_total = 2; p.y_plane0 = 0; p.y_nplanes = 2; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.N = N; p.H = H; p.W = W; p.cout_pad = C; p.relu = 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) conv_first_mfma_run(x, wf, p, 3, 5, DN_BF16, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%8.1f us per launch, %.2f TB/s written\n", ms * 100, (double)N * H * W * 64 / (ms / 10 * 1e-3) / 1e12);
  }
  // the same work as four concurrent partial launches on four streams (are the phases of co-resident blocks in lockstep?)
  hipStream_t st[4]; for (auto& q : st) hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
  const int tiles = N * (W / 128) * 1;
  for (int rep = 0; rep < 2; ++rep) {
    hipDeviceSynchronize();
    hipEventRecord(e0, st[0]);
    for (int i = 0; i < 10; ++i)
      for (int q = 0; q < 4; ++q) {
        ConvParams pq = p; pq.x_plane0 = q * (tiles / 4);
        hipLaunchKernelGGL((conv_first_mfma_kernel<__bf16, 3, 5, 6, CONV_EPI_PLAIN>), dim3(256, 1), dim3(256), 0, st[q], x, wf, [&]{ pq.ntw = W / 128; pq.nth = 1; return pq; }());
      }
    for (int q = 1; q < 4; ++q) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, st[q]); hipStreamWaitEvent(st[0], e, 0); }
    hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("4 streams x quarter launches: %8.1f us per full pass\n", ms * 100);
  }
  return 0;
}

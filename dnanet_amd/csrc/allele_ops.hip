// Device side of the allele hand-off: probability maps -> per-(dye, panel allele) peak table.
//
// Replaces the pixel loops of NearestBasePairCaller.translate_pixels_to_alleles (reference
// DNAnet/allele_callers.py:63-125): per dye row, pixels with probability >= 0.5 (:80) form runs of consecutive scan
// columns (:87); a run's mean base pair (:90) picks the panel allele with the nearest base pair on that dye (:122-123,
// first-listed wins a tie); the allele keeps the highest RFU seen in any of its runs (:94-98).  What goes back to the
// host is, per profile and dye, two small integer rows indexed by the dye's sorted allele table -- the maximum RFU
// and the first scan column that called the allele (which restores the reference's insertion order) -- instead of
// H x W probabilities walked in Python.
#include <climits>
#include "common.h"
#include "ops.h"

namespace {

constexpr int AL_THREADS = 256;

// one block per (dye row, profile); a thread owns the runs that START in its slice of the row
__global__ __launch_bounds__(AL_THREADS) void allele_runs_kernel(const float* __restrict__ probs, const float* __restrict__ rfu,
                                                                 const double* __restrict__ scaler, long scaler_stride,
                                                                 const double* __restrict__ table_bp,
                                                                 const int* __restrict__ table_order,
                                                                 const int* __restrict__ table_off, int H, int W, int amax,
                                                                 int* __restrict__ out_rfu, int* __restrict__ out_first) {
  const int h = blockIdx.x, n = blockIdx.y;
  const float* prow = probs + ((long)n * H + h) * W;
  const float* xrow = rfu + ((long)n * H + h) * W;
  const double* bp = scaler + (long)n * scaler_stride;
  const int t0 = table_off[h], tn = table_off[h + 1] - t0;
  if (tn <= 0) return;
  const double* tb = table_bp + t0;
  const int* to = table_order + t0;
  int* orfu = out_rfu + ((long)n * H + h) * amax;
  int* ofirst = out_first + ((long)n * H + h) * amax;
  const int per = (W + AL_THREADS - 1) / AL_THREADS;
  const int w_begin = threadIdx.x * per, w_end = min(W, w_begin + per);
  for (int w = w_begin; w < w_end; ++w) {
    if (!(prow[w] >= 0.5f) || (w > 0 && prow[w - 1] >= 0.5f)) continue;   // not a run start
    double sum = 0.0;
    float mx = xrow[w];
    int e = w;
    for (; e < W && prow[e] >= 0.5f; ++e) { sum += bp[e]; mx = fmaxf(mx, xrow[e]); }
    const double mean = sum / (double)(e - w);
    // nearest base pair: lower bound in the sorted table, then the closer neighbour; equal distances -> the allele
    // listed first in the panel
    int lo = 0, hi = tn;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (tb[mid] < mean) lo = mid + 1; else hi = mid; }
    int best = lo < tn ? lo : tn - 1;
    if (lo > 0 && lo < tn) {
      const double dl = fabs(tb[lo - 1] - mean), dr = fabs(tb[lo] - mean);
      if (dl < dr || (dl == dr && to[lo - 1] < to[lo])) best = lo - 1;
    }
    atomicMax(&orfu[best], (int)mx);        // table rows start at 0: a negative peak keeps 0, like max(rfus.get(k, 0), ...)
    atomicMin(&ofirst[best], w);
  }
}

// Pixel confusion counts of one profile per block (pixel_metrics.py:9-74): a pixel is predicted positive when
// p >= thr_ge (precision / recall / F1) resp. p > thr_gt (IoU); truth is 1 / 0, anything else counts for neither class.
// counts[image] = { tp, fp, fn } for ">=", { tp, fp, fn } for ">", then the number of truth values that are neither 0 nor 1
// (a mask the caller should not use these counts for).  Integer sums: order independent.
__global__ __launch_bounds__(256) void pixel_confusion_kernel(const float* __restrict__ probs, const signed char* __restrict__ truth,
                                                              long pixels, float thr_ge, float thr_gt, int* __restrict__ counts) {
  const float* p = probs + (long)blockIdx.x * pixels;
  const signed char* t = truth + (long)blockIdx.x * pixels;
  int c[7] = {0, 0, 0, 0, 0, 0, 0};
  for (long i = threadIdx.x; i < pixels; i += 256) {
    const float pv = p[i];
    const int one = t[i] == 1, zero = t[i] == 0;
    const int ge = pv >= thr_ge, gt = pv > thr_gt;
    c[0] += ge & one; c[1] += ge & zero; c[2] += (ge ^ 1) & one;
    c[3] += gt & one; c[4] += gt & zero; c[5] += (gt ^ 1) & one;
    c[6] += (one | zero) ^ 1;
  }
  __shared__ int red[4][7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    int v = c[k];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 7) counts[blockIdx.x * 7 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

}  // namespace

extern "C" {

int dn_call_alleles(const float* probs, const float* rfu, const double* scaler, long scaler_stride, const double* table_bp,
                    const int* table_order, const int* table_off, int N, int H, int W, int amax, int* out_rfu, int* out_first,
                    void* stream) {
  DN_CHECK_ARG(probs && rfu && scaler && table_bp && table_order && table_off && out_rfu && out_first, "dn_call_alleles: null pointer");
  DN_CHECK_ARG(N > 0 && N < 65536 && H > 0 && W > 0 && amax > 0, "dn_call_alleles: bad shape N=%d H=%d W=%d amax=%d", N, H, W, amax);
  DN_CHECK_ARG(scaler_stride == 0 || scaler_stride >= W, "dn_call_alleles: scaler stride %ld shorter than a row", scaler_stride);
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = (size_t)N * H * amax * sizeof(int);
  DN_CHECK_HIP(hipMemsetAsync(out_rfu, 0, bytes, s));
  DN_CHECK_HIP(hipMemsetAsync(out_first, 0x7f, bytes, s));      // 0x7f7f7f7f = "never called"
  hipLaunchKernelGGL(allele_runs_kernel, dim3(H, N), dim3(AL_THREADS), 0, s, probs, rfu, scaler, scaler_stride, table_bp,
                     table_order, table_off, H, W, amax, out_rfu, out_first);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

int dn_pixel_confusion(const float* probs, const signed char* truth, int n_images, long pixels, float thr_ge, float thr_gt,
                       int* counts, void* stream) {
  DN_CHECK_ARG(probs && truth && counts, "dn_pixel_confusion: null pointer");
  DN_CHECK_ARG(n_images > 0 && pixels > 0 && pixels < (1L << 31), "dn_pixel_confusion: bad shape %d x %ld", n_images, pixels);
  hipLaunchKernelGGL(pixel_confusion_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream, probs, truth, pixels, thr_ge,
                     thr_gt, counts);
  DN_LAUNCH_CHECK();
  return DN_OK;
}


}  // extern "C"

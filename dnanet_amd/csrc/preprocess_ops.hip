// Ingest resampling: raw scans -> 4096 scans spaced evenly in base pairs (reference DNAnet/data/utils.py:239-279
// rescale_dye + DNAnet/data/data_models/hid_image.py:168-183 _rescale_profile, :111 the scaler), batched.
#include "common.h"
#include "ops.h"

namespace {

// One block per profile.  bp[lo..hi] (the span between the first and last size-standard peak) is strictly increasing
// and everything outside it is 0, so the reference's argsort + searchsorted over the whole array is a lower-bound
// search inside the span; the nearer of the two bracketing scans wins, the lower one on a tie.
__global__ __launch_bounds__(256) void rescale_profiles_kernel(const float* __restrict__ raw, const double* __restrict__ bp,
                                                               const int* __restrict__ span, const double* __restrict__ target,
                                                               int rows, int length, int nout, float* __restrict__ out,
                                                               double* __restrict__ scaler, int* __restrict__ idx_out) {
  const int n = blockIdx.x;
  const double* b = bp + (long)n * length;
  const float* r = raw + (long)n * rows * length;
  const int lo = span[2 * n], hi = span[2 * n + 1];
  const bool no_zeros = hi - lo + 1 == length;          // the clip(ins, 1, len - 1) of the reference, in span terms
  for (int j = threadIdx.x; j < nout; j += blockDim.x) {
    const double t = target[j];
    int a = lo, e = hi + 1;                              // first position in [lo, hi] with b >= t, hi + 1 if none
    while (a < e) {
      const int mid = (a + e) >> 1;
      if (b[mid] < t) a = mid + 1; else e = mid;
    }
    int pos = min(a, hi);
    if (no_zeros && pos == lo) pos = lo + 1;
    int pick = pos;
    if (pos - 1 >= lo) {
      const double dl = fabs(b[pos - 1] - t), dr = fabs(b[pos] - t);
      if (dl <= dr) pick = pos - 1;
    }
    for (int row = 0; row < rows; ++row) out[((long)n * rows + row) * nout + j] = r[(long)row * length + pick];
    scaler[(long)n * nout + j] = b[pick];
    if (idx_out) idx_out[(long)n * nout + j] = pick;
  }
}

}  // namespace

extern "C" {

int dn_rescale_profiles(const float* raw, const double* basepairs, const int* span, const double* target, int n_profiles,
                        int rows, int length, int n_out, float* out, double* scaler, int* idx_out, void* stream) {
  DN_CHECK_ARG(raw && basepairs && span && target && out && scaler, "dn_rescale_profiles: null pointer");
  DN_CHECK_ARG(n_profiles > 0 && rows > 0 && length > 1 && n_out > 0, "dn_rescale_profiles: bad shape %d x %d x %d -> %d",
               n_profiles, rows, length, n_out);
  hipLaunchKernelGGL(rescale_profiles_kernel, dim3(n_profiles), dim3(256), 0, (hipStream_t)stream, raw, basepairs, span, target,
                     rows, length, n_out, out, scaler, idx_out);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

}  // extern "C"

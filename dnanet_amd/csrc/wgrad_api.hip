// C ABI of the weight-gradient GEMM: dispatch + fixed-order reduction of the split-K slabs.
#include "common.h"
#include "ops.h"

namespace {

// dw (canonical parameter layout) = sum over slabs; one thread per (tap, row, col) of the padded tile space
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int taps, int co_pad,
                                                           int ci_pad, float* dw, int Cout, int Cin, int convt, int split,
                                                           int half, int gsize) {
  const long per = (long)taps * co_pad * ci_pad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % ci_pad);
    long r = i / ci_pad;
    const int co = (int)(r % co_pad);
    const int tap = (int)(r / co_pad);
    long dst = -1;
    if (!convt) {
      int cic = ci;
      if (split > 0 && half != split) {             // concat input stored with padded halves
        const int grp = ci / half, within = ci % half;
        cic = (grp < 2 && within < split) ? grp * split + within : -1;
      }
      if (co < Cout && cic >= 0 && cic < Cin) dst = ((long)co * Cin + cic) * taps + tap;
    } else {                                        // ConvTranspose2d weight [Cin][Cout][1][2]; rows = k*gsize + co
      const int k = co / gsize, c = co % gsize;
      if (k < 2 && c < Cout && ci < Cin) dst = ((long)ci * Cout + c) * 2 + k;
    }
    if (dst < 0) continue;
    // many slabs (small tensors, few threads): eight independent chains keep loads in flight; fixed order => reproducible
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 7 < nslab; b += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] += slab[(long)(b + k) * per + i];
    }
    for (; b < nslab; ++b) c[0] += slab[(long)b * per + i];
    dw[dst] = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
  }
}

// Conv2d form with coalesced traffic on both sides: one block per (co, 64 input channels); slab reads run along
// ci (256-byte runs), the sums are turned through LDS, and dw[co][ci0..ci0+63][taps] is written as one run.
__global__ __launch_bounds__(256) void wgrad_reduce_conv_kernel(const float* __restrict__ slab, int nslab, int taps, int co_pad,
                                                                int ci_pad, float* dw, int Cout, int Cin, int split, int half) {
  __shared__ float tile[25 * 64];
  const int nci = (ci_pad + 63) / 64;
  const int co = blockIdx.x / nci, ci0 = (blockIdx.x % nci) * 64;
  if (co >= Cout) return;
  const long per = (long)taps * co_pad * ci_pad;
  for (int e = threadIdx.x; e < taps * 64; e += 256) {
    const int tap = e >> 6, cil = e & 63;
    float s = 0.f;
    if (ci0 + cil < ci_pad) {
      const float* src = slab + ((long)tap * co_pad + co) * ci_pad + ci0 + cil;
      // four independent chains keep the slab loads in flight (a single chain is one load latency per slab); fixed order
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int b = 0;
      for (; b + 3 < nslab; b += 4) {
        s0 += src[(long)b * per];
        s1 += src[(long)(b + 1) * per];
        s2 += src[(long)(b + 2) * per];
        s3 += src[(long)(b + 3) * per];
      }
      for (; b < nslab; ++b) s0 += src[(long)b * per];
      s = (s0 + s1) + (s2 + s3);
    }
    tile[tap * 64 + cil] = s;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < taps * 64; o += 256) {
    const int cil = o / taps, tap = o - cil * taps;
    const int ci = ci0 + cil;
    int cic = ci;
    if (split > 0 && half != split) {               // concat input stored with padded halves
      const int grp = ci / half, within = ci % half;
      cic = (grp < 2 && within < split) ? grp * split + within : -1;
    }
    if (ci < ci_pad && cic >= 0 && cic < Cin) dw[((long)co * Cin + cic) * taps + tap] = tile[tap * 64 + cil];
  }
}

// ConvTranspose2d form for MANY slabs of a small tensor (convt_wgrad.h: up to 512 splits of a 64 x 64 .. 512 x 512 output):
// a block sums 16 consecutive slab elements over 16 slab lanes (slab b goes to lane b % 16, four independent chains per
// lane), then the lanes are added in a fixed order -- 256 x more threads per element than one chain per element, which left
// the 64 -> 32 layer's reduction (16 blocks walking 2048 slabs) longer than its GEMM.  Bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_convt_kernel(const float* __restrict__ slab, int nslab, int co_pad, int ci_pad,
                                                                 float* dw, int Cout, int Cin, int gsize) {
  __shared__ float part[16][17];
  const long per = (long)co_pad * ci_pad;
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + e;
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < per) {
    int b = sl;
    for (; b + 48 < nslab; b += 64) {
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] += slab[(long)(b + 16 * k) * per + i];
    }
    for (; b < nslab; b += 16) c[0] += slab[(long)b * per + i];
  }
  part[sl][e] = (c[0] + c[1]) + (c[2] + c[3]);
  __syncthreads();
  if (threadIdx.x < 16 && i < per) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += part[k][e];
    const int ci = (int)(i % ci_pad), row = (int)(i / ci_pad);
    const int k = row / gsize, co = row % gsize;
    if (k < 2 && co < Cout && ci < Cin) dw[((long)ci * Cout + co) * 2 + k] = s;
  }
}

int run_wgrad(WgradParams& p, long ws_bytes, int kh, int kw, int dtype, int* nslab, hipStream_t s) {
  if (dtype == DN_BF16) {
    if (kh == 3 && kw == 5) return conv_wgrad_run_bf16_k35(p, ws_bytes, nslab, s);
    if (kh == 1 && kw == 3) return conv_wgrad_run_bf16_k13(p, ws_bytes, nslab, s);
    if (kh == 1 && kw == 1) return conv_wgrad_run_bf16_k11(p, ws_bytes, nslab, s);
  } else if (dtype == DN_F32) {
    if (kh == 3 && kw == 5) return conv_wgrad_run_f32_k35(p, ws_bytes, nslab, s);
    if (kh == 1 && kw == 3) return conv_wgrad_run_f32_k13(p, ws_bytes, nslab, s);
    if (kh == 1 && kw == 1) return conv_wgrad_run_f32_k11(p, ws_bytes, nslab, s);
  } else {
    return dn_set_error(DN_ERR_ARG, "bad dtype %d", dtype);
  }
  return dn_set_error(DN_ERR_UNSUPPORTED, "weight gradient for kernel size (%d,%d) is not built", kh, kw);
}

inline int tile_pad(int c, int tile) { return (c + tile - 1) / tile * tile; }

}  // namespace

extern "C" {

int dn_conv2d_wgrad(dn_tensor dy, dn_tensor a, float* dw, void* workspace, long workspace_bytes, int N, int H, int W, int kh,
                    int kw, int cin, int concat_split, int dtype, void* stream) {
  DN_CHECK_ARG(dy.base && a.base && dw && workspace, "dn_conv2d_wgrad: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && W > 0 && cin > 0, "dn_conv2d_wgrad: bad shape");
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv2d_wgrad: bad dtype %d", dtype);
  const int epr = dtype == DN_BF16 ? 16 : 8;
  // output-channel tile: 64 for bf16, 32 for fp32 and for bf16 3x5 layers with at most 32 output channels (wgrad_inst_bf16_k35.hip)
  const int co_t = dtype == DN_BF16 && !(kh == 3 && kw == 5 && dy.channels <= 32 && !dn_tune().wgrad_no_cot32) ? 64 : 32;
  WgradParams p{};
  p.dy = dy.base; p.dy_pt = dy.planes_total; p.dy_p0 = dy.plane0; p.dy_nplanes = dn_planes(dy.channels, dtype);
  p.dy_gplanes = 1 << 30; p.dy_wmul = 1;
  p.a = a.base; p.a_pt = a.planes_total; p.a_p0 = a.plane0; p.a_nplanes = dn_planes(a.channels, dtype);
  p.slab = (float*)workspace; p.N = N; p.H = H; p.W = W;
  p.co_pad = tile_pad(dy.channels, co_t); p.ci_pad = tile_pad(p.a_nplanes * epr, wg_ci_tile(kh * kw, dtype == DN_BF16 ? 2 : 4));
  DN_CHECK_ARG(dy.plane0 + p.dy_nplanes <= dy.planes_total && a.plane0 + p.a_nplanes <= a.planes_total,
               "dn_conv2d_wgrad: planes outside buffer");
  int nslab = 0;
  int rc = run_wgrad(p, workspace_bytes, kh, kw, dtype, &nslab, (hipStream_t)stream);
  if (rc) return rc;
  const int half = concat_split > 0 ? dn_planes(concat_split, dtype) * epr : 0;
  if (kh * kw <= 25 && nslab <= 64) {      // big tensors, few slabs: coalesced on both sides
    const int nci = (p.ci_pad + 63) / 64;
    hipLaunchKernelGGL(wgrad_reduce_conv_kernel, dim3((unsigned)(p.co_pad * nci)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, nslab, kh * kw, p.co_pad, p.ci_pad, dw, dy.channels, cin, concat_split, half);
  } else {
    const long per = (long)kh * kw * p.co_pad * p.ci_pad;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long>((per + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, nslab, kh * kw, p.co_pad, p.ci_pad, dw, dy.channels, cin, 0, concat_split, half, 1);
  }
  DN_LAUNCH_CHECK();
  return DN_OK;
}

int dn_conv_transpose_1x2_wgrad(dn_tensor dy, dn_tensor x, float* dw, void* workspace, long workspace_bytes, int N, int H,
                                int Win, int dtype, void* stream) {
  DN_CHECK_ARG(dy.base && x.base && dw && workspace, "dn_conv_transpose_1x2_wgrad: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && Win > 0, "dn_conv_transpose_1x2_wgrad: bad shape");
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv_transpose_1x2_wgrad: bad dtype %d", dtype);
  const int epr = dtype == DN_BF16 ? 16 : 8;
  // the large-tile GEMM (convt_wgrad.h) when the image is whole strips of pixels (every level of a U-Net on 4096 scans)
  if ((H * Win) % 64 == 0 && !dn_tune().no_convt_gemm) {
    ConvtWgradParams q{};
    q.dy = dy.base; q.dy_pt = dy.planes_total; q.dy_p0 = dy.plane0; q.dy_nplanes = dn_planes(dy.channels, dtype);
    q.x = x.base; q.x_pt = x.planes_total; q.x_p0 = x.plane0; q.x_nplanes = dn_planes(x.channels, dtype);
    DN_CHECK_ARG(dy.plane0 >= 0 && x.plane0 >= 0 && dy.plane0 + q.dy_nplanes <= dy.planes_total && x.plane0 + q.x_nplanes <= x.planes_total,
                 "dn_conv_transpose_1x2_wgrad: planes outside buffer");
    q.slab = (float*)workspace; q.N = N; q.HW = H * Win;
    int nslab = 0;
    const int rc = dtype == DN_BF16 ? convt_wgrad_run_bf16(q, dy.channels, workspace_bytes, &nslab, (hipStream_t)stream)
                                    : convt_wgrad_run_f32(q, dy.channels, workspace_bytes, &nslab, (hipStream_t)stream);
    if (rc) return rc;
    const long per = (long)2 * q.gsize * q.ci_pad;
    hipLaunchKernelGGL(wgrad_reduce_convt_kernel, dim3((unsigned)((per + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, nslab, 2 * q.gsize, q.ci_pad, dw, dy.channels, x.channels, q.gsize);
    DN_LAUNCH_CHECK();
    return DN_OK;
  }
  const int co_t = dtype == DN_BF16 ? 64 : 32;
  const int gsize = tile_pad(dy.channels, co_t);
  WgradParams p{};
  p.dy = dy.base; p.dy_pt = dy.planes_total; p.dy_p0 = dy.plane0; p.dy_nplanes = dn_planes(dy.channels, dtype);
  p.dy_gplanes = gsize / epr; p.dy_wmul = 2;
  p.a = x.base; p.a_pt = x.planes_total; p.a_p0 = x.plane0; p.a_nplanes = dn_planes(x.channels, dtype);
  p.slab = (float*)workspace; p.N = N; p.H = H; p.W = Win;
  p.co_pad = 2 * gsize; p.ci_pad = tile_pad(p.a_nplanes * epr, wg_ci_tile(1, dtype == DN_BF16 ? 2 : 4));
  int nslab = 0;
  int rc = run_wgrad(p, workspace_bytes, 1, 1, dtype, &nslab, (hipStream_t)stream);
  if (rc) return rc;
  const long per = (long)p.co_pad * p.ci_pad;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long>((per + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, nslab, 1, p.co_pad, p.ci_pad, dw, dy.channels, x.channels, 1, 0, 0, gsize);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

}  // extern "C"

// First convolution of the U-Net (Cin = 1: nn.Conv2d(1, F, k, padding=k//2) on the raw fp32 RFU, reference
// unet_architecture.py:14-16 with trainable_unet.py:70-79) on the fp32 matrix cores.
//
// The input stays fp32 (raw RFU reach 3e4: bf16's 8 bits cannot hold them), so the contraction runs on
// v_mfma_f32_32x32x2_f32 in both storage modes: M = 32 output channels, N = 32 scan columns, K = the KH*KW taps, two per
// MFMA (15 taps = 8 MFMAs per 32 pixels; at the 157 TFLOP/s fp32 matrix peak the whole layer is ~45 us for 256 six-dye
// profiles, under the ~80 us its output takes to write).  A block = 4 waves x 32 columns x HT dye rows; the HT + KH - 1
// input rows of the tile (4 KiB) are staged through LDS with the zero border written in, a lane's tap weights live in
// registers, and the output leaves through the shared convolution epilogue (conv_igemm.h): bias + ReLU, 1-KiB stores,
// optionally BatchNorm batch statistics (training forward).  The VALU kernel this replaces ran at 1.6 TB/s.
#include "conv_igemm.h"
#include "ops.h"

template <typename T, int KH, int KW, int HT, int EPI>
__global__ __launch_bounds__(256, HT > 3 ? 2 : EPI == CONV_EPI_PLAIN ? 4 : 3) void conv_first_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wf,
                                                              const ConvParams p) {
  constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2, KS = (TAPS + 1) / 2, COT = 32;
  constexpr int EPR = RecTraits<T>::EPR, NPL = COT / EPR;
  constexpr int ROWS = HT + KH - 1, QW = 128 + KW - 1;
  constexpr bool STATS = EPI == CONV_EPI_STATS;
  __shared__ float xs[ROWS * QW];
  __shared__ __attribute__((aligned(16))) unsigned char epi[4 * NPL * 1536 + 4 * COT * 2 * 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int cot = blockIdx.y;
  const int ntiles = p.N * p.nth * p.ntw;

  // this lane's taps: K step s contracts taps 2s (lower half-wave) and 2s + 1 (upper); a tap past the end has weight 0
  float a[KS];
  int boff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int tap = 2 * s + half;
    const int tc = tap < TAPS ? tap : TAPS - 1;
    a[s] = tap < TAPS ? wf[tap * p.cout_pad + cot * COT + l31] : 0.f;
    boff[s] = (tc / KW) * QW + tc % KW + 32 * wave + l31;
  }

  // A block walks tiles blockIdx.x, + gridDim.x, ... and keeps the NEXT tile's input in flight (registers) while the matrix
  // cores work on the current one; the epilogue's stores are fire-and-forget.  One tile per block (the first version)
  // ran load -> MFMA -> store strictly one after the other in every co-resident block at the same time: 183 us = the sum
  // of the three phases (44 + 86 + 51 us, scratch/first_micro.hip) for a 403 MB write.
  constexpr int NPRE = (ROWS * QW + 255) / 256;
  float pre[NPRE];
  int pre_r[NPRE], pre_q[NPRE];            // a thread's elements of the tile image: the same (row, column) for every tile
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {
    const int i = tid + k * 256;
    pre_r[k] = i / QW - PH;
    pre_q[k] = i % QW - PW;
  }
  auto prefetch = [&](int ptile) {
    int pt = ptile;
    const int tw = pt % p.ntw; pt /= p.ntw;
    const int th = pt % p.nth;
    const int n = pt / p.nth;
    const float* xn = x + (long)n * p.H * p.W;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int i = tid + k * 256;
      const int h = th * HT + pre_r[k], w = tw * 128 + pre_q[k];
#ifndef CF_ABLATE_NO_LOAD     // (ablation hooks: scratch/first_micro.hip)
      pre[k] = (ptile < ntiles && i < ROWS * QW && h >= 0 && h < p.H && w >= 0 && w < p.W) ? xn[(long)h * p.W + w] : 0.f;
#else
      pre[k] = (float)(h + w);
#endif
    }
  };
  float* stat_lds = (float*)(epi + 4 * NPL * 1024);
  unsigned char* pool_lds = epi + 4 * NPL * 1024 + 4 * COT * 2 * 4;

  prefetch(blockIdx.x + p.x_plane0);     // (x_plane0: first tile of a partial launch, 0 otherwise)
  for (int ptile = blockIdx.x + p.x_plane0; ptile < ntiles; ptile += gridDim.x) {
    int pt = ptile;
    const int tw = pt % p.ntw; pt /= p.ntw;
    const int th = pt % p.nth;
    const int n = pt / p.nth;
    const int w0 = tw * 128, h0 = th * HT;
    __syncthreads();                     // the previous tile's MFMAs (and its statistics sum) are done with the LDS
#pragma unroll
    for (int k = 0; k < NPRE; ++k)
      if (tid + k * 256 < ROWS * QW) xs[tid + k * 256] = pre[k];
    __syncthreads();
    prefetch(ptile + gridDim.x);

    f32x16_t acc[HT][1];
#pragma unroll
    for (int hh = 0; hh < HT; ++hh) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[hh][0][i] = 0.f;
#ifndef CF_ABLATE_NO_MFMA
#pragma unroll
      for (int s = 0; s < KS; ++s)
        acc[hh][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], xs[hh * QW + boff[s]], acc[hh][0], 0, 0, 0);
#else
      acc[hh][0][0] = a[hh % KS] + xs[hh * QW + boff[0]];
#endif
    }

    conv_epilogue_store<T, HT, COT, EPI>(p, acc, epi + wave * (NPL * 1024), stat_lds + wave * COT * 2,
                                         pool_lds + wave * (NPL * 512), n, h0, w0 + 32 * wave, cot, lane);
    if constexpr (STATS) {
      __syncthreads();
      if (tid < COT * 2) {
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) t += stat_lds[wv * COT * 2 + tid];
        p.stats[((long)ptile * p.cout_pad + cot * COT) * 2 + tid] = t;
      }
    }
  }
}

template <typename T, int KH, int KW, int HT>
static int launch_ht(const float* x, const float* wf, ConvParams p, hipStream_t s) {
  p.ntw = (p.W + 127) / 128;
  p.nth = (p.H + HT - 1) / HT;
  const long tiles = (long)p.N * p.ntw * p.nth;
  if (tiles > 0x7fffffffL || p.cout_pad / 32 > 65535) return dn_set_error(DN_ERR_ARG, "conv_first: grid out of range");
  // persistent blocks: four per CU (256 CUs), each walking every (grid)th tile
  const dim3 grid((unsigned)std::min<long>(tiles, 1024), (unsigned)(p.cout_pad / 32));
  if constexpr (sizeof(T) == 2) {
    if (p.y_lo_off) {          // hi + lo split output (bf16x3 inference mode)
      hipLaunchKernelGGL((conv_first_mfma_kernel<T, KH, KW, HT, CONV_EPI_SPLIT>), grid, dim3(256), 0, s, x, wf, p);
      DN_LAUNCH_CHECK();
      return DN_OK;
    }
  }
  if (p.stats) hipLaunchKernelGGL((conv_first_mfma_kernel<T, KH, KW, HT, CONV_EPI_STATS>), grid, dim3(256), 0, s, x, wf, p);
  else hipLaunchKernelGGL((conv_first_mfma_kernel<T, KH, KW, HT, CONV_EPI_PLAIN>), grid, dim3(256), 0, s, x, wf, p);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

int conv_first_row_tile(int H) { return H % 5 == 0 ? 5 : H % 3 == 0 ? 3 : 1; }   // 6-row tiles: 96 AGPR + 102 VGPR = 2 waves per SIMD, half the blocks in flight

template <typename T, int KH, int KW>
static int launch_k(const float* x, const float* wf, const ConvParams& p, hipStream_t s) {
  switch (conv_first_row_tile(p.H)) {
    case 5: return launch_ht<T, KH, KW, 5>(x, wf, p, s);
    case 6: return launch_ht<T, KH, KW, 6>(x, wf, p, s);
    case 3: return launch_ht<T, KH, KW, 3>(x, wf, p, s);
    default: return launch_ht<T, KH, KW, 1>(x, wf, p, s);
  }
}

// p carries the output side (y, bias, relu, N, H, W, cout_pad, stats); kernel sizes (3,5) and (1,3)
int conv_first_mfma_run(const float* x, const float* wf, const ConvParams& p, int kh, int kw, int dtype, hipStream_t s) {
  if (kh == 3 && kw == 5) return dtype == DN_BF16 ? launch_k<__bf16, 3, 5>(x, wf, p, s) : launch_k<float, 3, 5>(x, wf, p, s);
  if (kh == 1 && kw == 3) return dtype == DN_BF16 ? launch_k<__bf16, 1, 3>(x, wf, p, s) : launch_k<float, 1, 3>(x, wf, p, s);
  return dn_set_error(DN_ERR_UNSUPPORTED, "conv_first: kernel size (%d,%d) not built for the matrix-core path", kh, kw);
}

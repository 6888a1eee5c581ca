// Implicit-GEMM convolution over the dye x scan axes for gfx950 (MI355X), MFMA 32x32 tiles.
//
// Replaces nn.Conv2d(k=(KH,KW), padding=k//2) of DoubleConv (reference
// DNAnet/models/segmentation/unet_architecture.py:14-20), and -- with KH=KW=1 and the column
// multiplier / plane-group addressing below -- nn.ConvTranspose2d(k=(1,2), s=(1,2)) (:54-57) and
// the data-gradient of both.
//
// GEMM view: D[co][pixel] = sum_{tap, ci} Wp[tap][co][ci] * X[pixel + tap][ci]
//   A operand (M = 32 output channels)  : packed weights, LDS image [tap][co][32 B]
//   B operand (N = 32 consecutive scan columns of one dye row) : activations, LDS image
//                                          [row][col + halo][32 B]
//   K step = one 32-byte record = one plane (16 bf16 / 8 fp32 input channels).
// Block = 4 waves; tile = HT dye rows x 128 scan columns x COT output channels; wave w owns
// columns [32w, 32w+32) of every row, i.e. HT x COT/32 accumulator tiles. A staged activation
// tile is reused by all KH*KW taps, so LDS traffic per MFMA is < 0.5 ds_read_b128.
// 16-byte pieces inside a record are XOR-swizzled by bit 3 of the record index, which makes every
// ds_read_b128 of a fragment conflict-free for any tap shift (MI355X_MICROARCH.md, LDS table).
#pragma once
#include "common.h"

struct ConvParams {
  const void* x; int x_planes_total, x_plane0, x_nplanes, x_gplanes, x_wmul;
  const void* w;            // packed [nchunk][TAPS][cout_pad][32 B]
  const float* bias;        // [cout_pad] (virtual channel order) or nullptr
  void* y; int y_planes_total, y_plane0, y_nplanes, y_gplanes, y_wmul;
  int N, H, W;              // pixel grid of the GEMM (per group)
  int nchunk;               // K chunks = virtual input planes
  int cout_pad;             // multiple of COT
  int relu;
  int ntw, nth;             // column / row tiles per image
};

template <typename T> struct Mma32;
template <> struct Mma32<__bf16> {
  __device__ __forceinline__ static void run(const uint4& a, const uint4& b, f32x16_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                  acc, 0, 0, 0);
  }
};
template <> struct Mma32<float> {
  // 8 fp32 channels per record: lane half h holds channels 4h..4h+3; MFMA j contracts {j, 4+j}.
  __device__ __forceinline__ static void run(const uint4& a, const uint4& b, f32x16_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

template <typename T, int KH, int KW, int HT, int COT>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
  constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
  constexpr int ROWS = HT + KH - 1;
  constexpr int QW = 128 + KW - 1;
  constexpr int NCG = COT / 32;
  constexpr int EPR = RecTraits<T>::EPR;
  constexpr int ACT_BYTES = ROWS * QW * 32;
  constexpr int W_BYTES = TAPS * COT * 32;
  constexpr int XP = 2 * (KW - 1);          // pieces per tile row beyond the first 256
  constexpr int TPI = 128 / COT;            // taps staged per 256-piece pass
  constexpr int W_ITERS = (TAPS + TPI - 1) / TPI;
  constexpr int SB = 4;   // staging batch (pieces per thread in flight)
  __shared__ __attribute__((aligned(16))) unsigned char smem[ACT_BYTES + W_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  const int npt = p.N * p.nth * p.ntw;
  const int cot = blockIdx.x / npt;
  int pt = blockIdx.x - cot * npt;
  const int tw = pt % p.ntw; pt /= p.ntw;
  const int th = pt % p.nth;
  const int n = pt / p.nth;
  const int w0 = tw * 128, h0 = th * HT;
  const int xrow = p.W * p.x_wmul;

  // per-thread staging offsets (bytes)
  const int a_q = tid >> 1, a_pc = tid & 1;
  const int a_w = w0 + a_q - PW;
  const bool a_ok = a_w >= 0 && a_w < p.W;
  const long a_goff = ((long)a_w * p.x_wmul) * 32 + 16 * a_pc;
  const int a_loff = a_q * 32 + 16 * (a_pc ^ ((a_q >> 3) & 1));
  const int x_r = XP > 0 ? tid / (XP > 0 ? XP : 1) : 0;
  const int x_i = 256 + (XP > 0 ? tid % (XP > 0 ? XP : 1) : 0);
  const int x_q = x_i >> 1, x_pc = x_i & 1;
  const int x_w = w0 + x_q - PW;
  const bool x_ok = x_w >= 0 && x_w < p.W;
  const long x_goff = ((long)x_w * p.x_wmul) * 32 + 16 * x_pc;
  const int x_loff = x_q * 32 + 16 * (x_pc ^ ((x_q >> 3) & 1));
  const int w_tap = tid / (2 * COT), w_co = (tid >> 1) % COT, w_pc = tid & 1;
  const long w_goff = ((long)w_tap * p.cout_pad + w_co) * 32 + 16 * w_pc;
  const int w_loff = (w_tap * COT + w_co) * 32 + 16 * (w_pc ^ ((w_co >> 3) & 1));

  f32x16_t acc[HT][NCG];
#pragma unroll
  for (int a = 0; a < HT; ++a)
#pragma unroll
    for (int b = 0; b < NCG; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const unsigned char* xb = (const unsigned char*)p.x;
  const unsigned char* wb = (const unsigned char*)p.w;

  for (int ck = 0; ck < p.nchunk; ++ck) {
    const int grp = ck / p.x_gplanes;
    const int plane = ck - grp * p.x_gplanes;
    if (plane >= p.x_nplanes) continue;   // padding plane of a virtual group: weights are zero, data absent
    const long plane_base = ((long)(n * p.x_planes_total + p.x_plane0 + plane) * p.H) * xrow;
    __syncthreads();   // previous chunk's fragment reads are done
    // ---- stage the activation tile (rows h0-PH .. h0+HT-1+PH, columns w0-PW .. w0+127+PW).
    // Thread t moves 16-byte piece t of every tile row (a row is QW*2 = 256 + 2*(KW-1) pieces); the
    // few pieces past 256 are moved by threads 0..ROWS*XP-1 in one extra pass. Addresses are
    // "uniform row base + per-thread offset", so nothing but two offsets lives across the K loop.
    const unsigned char* xplane = xb + plane_base * 32 + (long)grp * 32;
#pragma unroll
    for (int r0 = 0; r0 < ROWS; r0 += SB) {
      uint4 v[SB];
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        const int r = r0 + k;
        const int h = h0 + r - PH;
        v[k] = make_uint4(0, 0, 0, 0);
        if (r < ROWS && a_ok && h >= 0 && h < p.H) v[k] = *(const uint4*)(xplane + (long)h * xrow * 32 + a_goff);
      }
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        const int r = r0 + k;
        if (r < ROWS) *(uint4*)(smem + r * QW * 32 + a_loff) = v[k];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (XP > 0) {
      if (tid < ROWS * XP) {
        const int h = h0 + x_r - PH;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (x_ok && h >= 0 && h < p.H) v = *(const uint4*)(xplane + (long)h * xrow * 32 + x_goff);
        *(uint4*)(smem + x_r * QW * 32 + x_loff) = v;
      }
    }
    // ---- stage the weight tile [TAPS][COT] records of this chunk / output-channel tile
    {
      const unsigned char* wsrc = wb + ((long)ck * TAPS * p.cout_pad + (long)cot * COT) * 32;
#pragma unroll
      for (int it0 = 0; it0 < W_ITERS; it0 += SB) {
        uint4 v[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int tap = (it0 + k) * TPI + w_tap;
          v[k] = make_uint4(0, 0, 0, 0);
          if (it0 + k < W_ITERS && tap < TAPS) v[k] = *(const uint4*)(wsrc + (long)((it0 + k) * TPI) * p.cout_pad * 32 + w_goff);
        }
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int tap = (it0 + k) * TPI + w_tap;
          if (it0 + k < W_ITERS && tap < TAPS) *(uint4*)(smem + ACT_BYTES + (it0 + k) * 4096 + w_loff) = v[k];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    // ---- MFMA over all taps
#pragma unroll
    for (int dwi = 0; dwi < KW; ++dwi) {
      uint4 bf[ROWS];
      const int q = 32 * wave + l31 + dwi;
      const int boff = q * 32 + 16 * (half ^ ((q >> 3) & 1));
#pragma unroll
      for (int r = 0; r < ROWS; ++r) bf[r] = *(const uint4*)(smem + r * QW * 32 + boff);
#pragma unroll
      for (int dhi = 0; dhi < KH; ++dhi) {
        uint4 af[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
          const int col = cg * 32 + l31;
          af[cg] = *(const uint4*)(smem + ACT_BYTES + ((dhi * KW + dwi) * COT + col) * 32 +
                                   16 * (half ^ ((col >> 3) & 1)));
        }
#pragma unroll
        for (int hh = 0; hh < HT; ++hh) {
          const int hin = h0 + hh + dhi - PH;
          if (h0 + hh < p.H && hin >= 0 && hin < p.H) {   // block-uniform
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) Mma32<T>::run(af[cg], bf[hh + dhi], acc[hh][cg]);
          }
        }
      }
    }
  }

  // ---- epilogue: bias (+ folded BN) + ReLU, store 4 consecutive channels per lane and group
  const int wcol = w0 + 32 * wave + l31;
  unsigned char* yb = (unsigned char*)p.y;
  const int yrow = p.W * p.y_wmul;
  if (wcol < p.W) {
#pragma unroll
    for (int hh = 0; hh < HT; ++hh) {
      const int h = h0 + hh;
      if (h < p.H) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co0 = cot * COT + cg * 32 + 8 * g + 4 * half;
            const int vplane = co0 / EPR, within = co0 % EPR;
            const int ygrp = vplane / p.y_gplanes;
            const int yplane = vplane - ygrp * p.y_gplanes;
            if (yplane < p.y_nplanes) {
              float v[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                v[i] = acc[hh][cg][4 * g + i];
                if (p.bias) v[i] += p.bias[co0 + i];
                if (p.relu) v[i] = fmaxf(v[i], 0.f);
              }
              const long rec = ((long)(n * p.y_planes_total + p.y_plane0 + yplane) * p.H + h) * yrow +
                               (long)wcol * p.y_wmul + ygrp;
              unsigned char* dst = yb + rec * 32 + within * sizeof(T);
              if constexpr (sizeof(T) == 4) {
                *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
              } else {
                __bf16 b0 = (__bf16)v[0], b1 = (__bf16)v[1], b2 = (__bf16)v[2], b3 = (__bf16)v[3];
                uint2 u;
                u.x = (uint32_t)__builtin_bit_cast(unsigned short, b0) | ((uint32_t)__builtin_bit_cast(unsigned short, b1) << 16);
                u.y = (uint32_t)__builtin_bit_cast(unsigned short, b2) | ((uint32_t)__builtin_bit_cast(unsigned short, b3) << 16);
                *(uint2*)dst = u;
              }
            }
          }
        }
      }
    }
  }
}

template <typename T, int KH, int KW, int HT, int COT>
int conv_igemm_launch(ConvParams p, hipStream_t stream) {
  p.ntw = (p.W + 127) / 128;
  p.nth = (p.H + HT - 1) / HT;
  const long blocks = (long)p.N * p.ntw * p.nth * (p.cout_pad / COT);
  if (blocks <= 0 || blocks > 0x7fffffffL) return dn_set_error(DN_ERR_ARG, "conv grid out of range: %ld blocks", blocks);
  hipLaunchKernelGGL((conv_igemm_kernel<T, KH, KW, HT, COT>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

// Weight gradient of the dye x scan convolution as a split-K MFMA GEMM for gfx950.
//
// Replaces the weight half of autograd's convolution_backward for nn.Conv2d(k=(KH,KW), padding=k//2)
// (reference unet_architecture.py:14-20, driven by loss.backward() at trainable_unet.py:342) and,
// with KH=KW=1 and the column multiplier, for nn.ConvTranspose2d(k=(1,2), s=(1,2)) (:54-57).
//
//   dW[tap][co][ci] = sum_{n,h,w} dY[n,h,w][co] * A[n,h+dh,w+dw][ci]
// GEMM per tap: M = co (operand dY^T), N = ci (operand A shifted by the tap), K = pixels.
// Both operands sit in HBM pixel-major ([pixel][16 ch] records), so K is the *slow* index of every
// fragment: bf16 fragments are fetched with the gfx950 transposing LDS read ds_read_b64_tr_b16
// (4 pixels x 16 channels per 16-lane group, delivered channel-per-lane); fp32 fragments need one
// value per lane (v_mfma_f32_32x32x2_f32) and use plain ds_read_b32.
// Block = 4 waves, tile = CO_T x CI_T x all taps, K-loop over 128-column strips of one dye row
// (plus KH-1 halo rows of A). Waves split the taps (3x5: 4/4/4/3); for kernels with < 4 taps they
// split the K steps instead and write one partial slab each.
// Split-K partials go to a slab [slab][tap][co_pad][ci_pad] that a second kernel sums in fixed order
// (bitwise reproducible; no float atomics) and scatters into the canonical parameter layout.
#pragma once
#include "common.h"

__device__ __attribute__((aligned(16))) const unsigned int dn_zero_page_wg[8] = {0, 0, 0, 0, 0, 0, 0, 0};

struct WgradParams {
  const void* dy; int dy_pt, dy_p0, dy_nplanes, dy_gplanes, dy_wmul;
  const void* a;  int a_pt, a_p0, a_nplanes;
  float* slab;
  int N, H, W;
  int co_pad, ci_pad;
  int nsplit, nstrips, ntw;
};

template <typename T> struct WgTraits;
template <> struct WgTraits<__bf16> { static constexpr int CO_T = 64, CI_T = 32, KSTEP = 16, BANKW = 256, PHASE = 128; };
template <> struct WgTraits<float>  { static constexpr int CO_T = 32, CI_T = 32, KSTEP = 2,  BANKW = 128, PHASE = 32; };

__device__ __forceinline__ uint2 lds_tr16_b64(const unsigned char* p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

template <typename T, int KH, int KW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p) {
  using TR = WgTraits<T>;
  constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
  constexpr int EPR = RecTraits<T>::EPR;
  constexpr int CO_T = TR::CO_T, CI_T = TR::CI_T, KSTEP = TR::KSTEP;
  constexpr int NCOG = CO_T / 32, NCIG = CI_T / 32;
  constexpr int CO_PL = CO_T / EPR, CI_PL = CI_T / EPR;     // planes per tile
  constexpr int PLG = 32 / EPR;                             // planes per 32-channel group
  constexpr int QW = 128 + KW - 1;
  constexpr int DY_RAW = 128 * 32, A_RAW = KH * QW * 32;
  constexpr int DY_PS = DY_RAW + (TR::PHASE - DY_RAW % TR::BANKW + TR::BANKW) % TR::BANKW;   // plane strides chosen so the
  constexpr int A_PS = A_RAW + (TR::PHASE - A_RAW % TR::BANKW + TR::BANKW) % TR::BANKW;      // planes of one read hit distinct banks
  constexpr int DY_BYTES = CO_PL * DY_PS, A_BYTES = CI_PL * A_PS;
  constexpr bool KSPLIT = TAPS < 4;                         // waves split K steps instead of taps
  constexpr int WT = KSPLIT ? TAPS : (TAPS + 3) / 4;        // taps per wave
  constexpr int NK = 128 / KSTEP;
  constexpr int ST_ITERS = (DY_BYTES + A_BYTES + 4095) / 4096;
  __shared__ __attribute__((aligned(16))) unsigned char smem[ST_ITERS * 4096];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int n_cit = p.ci_pad / CI_T, n_cot = p.co_pad / CO_T;
  // tiles are the fast block index: blocks that run together walk the SAME strips (same split) for different
  // (co, ci) tiles, so a strip is fetched from HBM about once per XCD L2 instead of once per tile
  int b = blockIdx.x;
  const int cit = b % n_cit; b /= n_cit;
  const int cot = b % n_cot;
  const int split = b / n_cot;

  f32x16_t acc[WT][NCOG][NCIG];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int a = 0; a < NCOG; ++a)
#pragma unroll
      for (int c = 0; c < NCIG; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][a][c][r] = 0.f;

  const unsigned char* dyb = (const unsigned char*)p.dy;
  const unsigned char* ab = (const unsigned char*)p.a;
  const long dy_row = (long)p.W * p.dy_wmul;

  // staging plan: LDS piece P = (it*4 + wave)*64 + lane of the linear image [dY planes | A planes] (plane strides
  // padded, pad pieces are zero-filled).  st_off = byte offset of the source relative to the strip's base address,
  // st_meta = column | row << 8 | isA << 12, or -1 for padding / absent planes.
  int st_off[ST_ITERS], st_meta[ST_ITERS];
#pragma unroll
  for (int it = 0; it < ST_ITERS; ++it) {
    const int byte = ((it * 4 + wave) * 64 + lane) * 16;
    int off = 0, meta = -1;
    if (byte < DY_BYTES) {
      const int pl = byte / DY_PS, within = byte - pl * DY_PS;
      if (within < DY_RAW) {
        const int q = within >> 5, pc = (within >> 4) & 1;
        const int vp = cot * CO_PL + pl;
        const int grp = vp / p.dy_gplanes, plane = vp - grp * p.dy_gplanes;
        if (plane < p.dy_nplanes) {
          off = (int)((long)plane * p.H * dy_row * 32) + (q * p.dy_wmul + grp) * 32 + 16 * pc;
          meta = q;
        }
      }
    } else if (byte < DY_BYTES + A_BYTES) {
      const int b2 = byte - DY_BYTES;
      const int pl = b2 / A_PS, within = b2 - pl * A_PS;
      if (within < A_RAW) {
        const int rq = within >> 5, pc = (within >> 4) & 1;
        const int row = rq / QW, q = rq - row * QW;
        const int plane = cit * CI_PL + pl;
        if (plane < p.a_nplanes) {
          off = (int)((long)plane * p.H * p.W * 32) + ((row - PH) * p.W + (q - PW)) * 32 + 16 * pc;
          meta = q | (row << 8) | (1 << 12);
        }
      }
    }
    st_off[it] = off;
    st_meta[it] = meta;
  }

  for (int strip = split; strip < p.nstrips; strip += p.nsplit) {
    const int tw = strip % p.ntw;
    const int h = (strip / p.ntw) % p.H;
    const int n = strip / (p.ntw * p.H);
    const int w0 = tw * 128;
    __syncthreads();
#ifndef WG_ABLATE_NO_STAGE
    // ---- stage dY (CO_PL planes x 128 px) and A (CI_PL planes x KH rows x QW px) straight into LDS with
    // global_load_lds_dwordx4: every piece of the strip is in flight at once and costs no VGPRs.
    {
      const unsigned char* dy_base = dyb + (((long)(n * p.dy_pt + p.dy_p0) * p.H + h) * dy_row + (long)w0 * p.dy_wmul) * 32;
      const unsigned char* a_base = ab + (((long)(n * p.a_pt + p.a_p0) * p.H + h) * p.W + w0) * 32;
#pragma unroll
      for (int it = 0; it < ST_ITERS; ++it) {
        const int meta = st_meta[it];
        const unsigned char* src = (const unsigned char*)dn_zero_page_wg;
        if (meta >= 0) {
          const int q = meta & 0xff, row = (meta >> 8) & 0xf;
          if (meta & (1 << 12)) {                       // A piece: image row h + row - PH, column w0 + q - PW
            const int hh = h + row - PH, w = w0 + q - PW;
            if (hh >= 0 && hh < p.H && w >= 0 && w < p.W) src = a_base + st_off[it];
          } else if (w0 + q < p.W) {
            src = dy_base + st_off[it];
          }
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + (it * 4 + wave) * 1024), 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
#ifndef WG_ABLATE_NO_COMPUTE
    // ---- MFMA over the strip's pixels
    if constexpr (sizeof(T) == 2) {
      const int g = lane >> 4, idx = lane & 15;
      const int toff = (8 * (g >> 1) + (idx >> 2)) * 32 + 8 * (idx & 3);   // pixel (8*khalf + q), columns 4p..4p+3
      const int gpl = g & 1;
#pragma unroll 2
      for (int ks0 = 0; ks0 < NK; ks0 += (KSPLIT ? 4 : 1)) {
        const int ks = KSPLIT ? ks0 + wave : ks0;
        uint4 af[NCOG];
#pragma unroll
        for (int cg = 0; cg < NCOG; ++cg) {
          const unsigned char* base = smem + (cg * PLG + gpl) * DY_PS + ks * 16 * 32 + toff;
          const uint2 lo = lds_tr16_b64(base), hi = lds_tr16_b64(base + 4 * 32);
          af[cg] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int i = 0; i < WT; ++i) {
          const int tap = KSPLIT ? i : wave + 4 * i;
          if (tap < TAPS) {
            const int dh = tap / KW, dw = tap - dh * KW;
            uint4 bf[NCIG];
#pragma unroll
            for (int cg = 0; cg < NCIG; ++cg) {
              const unsigned char* base = smem + DY_BYTES + (cg * PLG + gpl) * A_PS + (dh * QW + ks * 16 + dw) * 32 + toff;
              const uint2 lo = lds_tr16_b64(base), hi = lds_tr16_b64(base + 4 * 32);
              bf[cg] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int a = 0; a < NCOG; ++a)
#pragma unroll
              for (int c = 0; c < NCIG; ++c)
                acc[i][a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[a]),
                                                                       __builtin_bit_cast(bf16x8_t, bf[c]), acc[i][a][c], 0, 0, 0);
          }
        }
      }
    } else {
      const int foff = (l31 >> 3) * 0 + (l31 & 7) * 4;   // channel inside its plane
      const int fpl = l31 >> 3;
#pragma unroll 4
      for (int ks0 = 0; ks0 < NK; ks0 += (KSPLIT ? 4 : 1)) {
        const int ks = KSPLIT ? ks0 + wave : ks0;
        const int px = ks * 2 + half;
        float af[NCOG];
#pragma unroll
        for (int cg = 0; cg < NCOG; ++cg) af[cg] = *(const float*)(smem + (cg * PLG + fpl) * DY_PS + px * 32 + foff);
#pragma unroll
        for (int i = 0; i < WT; ++i) {
          const int tap = KSPLIT ? i : wave + 4 * i;
          if (tap < TAPS) {
            const int dh = tap / KW, dw = tap - dh * KW;
            float bf[NCIG];
#pragma unroll
            for (int cg = 0; cg < NCIG; ++cg)
              bf[cg] = *(const float*)(smem + DY_BYTES + (cg * PLG + fpl) * A_PS + (dh * QW + px + dw) * 32 + foff);
#pragma unroll
            for (int a = 0; a < NCOG; ++a)
#pragma unroll
              for (int c = 0; c < NCIG; ++c)
                acc[i][a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[c], acc[i][a][c], 0, 0, 0);
          }
        }
      }
    }
#endif
  }

  // ---- write this block's partial tile(s)
  const int slab_idx = KSPLIT ? split * 4 + wave : split;
  float* slab = p.slab + (long)slab_idx * TAPS * p.co_pad * p.ci_pad;
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int tap = KSPLIT ? i : wave + 4 * i;
    if (tap < TAPS) {
#pragma unroll
      for (int a = 0; a < NCOG; ++a)
#pragma unroll
        for (int c = 0; c < NCIG; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = cot * CO_T + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int ci = cit * CI_T + c * 32 + l31;
            slab[((long)tap * p.co_pad + co) * p.ci_pad + ci] = acc[i][a][c][r];
          }
    }
  }
}

template <typename T, int KH, int KW>
int conv_wgrad_launch(WgradParams p, long slab_bytes, int* nslab_out, hipStream_t stream) {
  using TR = WgTraits<T>;
  constexpr int TAPS = KH * KW;
  constexpr bool KSPLIT = TAPS < 4;
  p.ntw = (p.W + 127) / 128;
  p.nstrips = p.N * p.H * p.ntw;
  p.co_pad = (p.co_pad + TR::CO_T - 1) / TR::CO_T * TR::CO_T;
  p.ci_pad = (p.ci_pad + TR::CI_T - 1) / TR::CI_T * TR::CI_T;
  const long tiles = (long)(p.co_pad / TR::CO_T) * (p.ci_pad / TR::CI_T);
  const long per_slab = (long)TAPS * p.co_pad * p.ci_pad * 4 * (KSPLIT ? 4 : 1);
  long nsplit = std::max<long>(1, (1024 + tiles - 1) / tiles);
  nsplit = std::min<long>(nsplit, std::max<long>(1, p.nstrips / 16));   // >= 16 strips per block: slabs cost a round trip
  nsplit = std::min<long>(nsplit, slab_bytes / per_slab);
  if (nsplit < 1) return dn_set_error(DN_ERR_ARG, "conv_wgrad: workspace of %ld bytes is smaller than one partial slab (%ld)",
                                      slab_bytes, per_slab);
  p.nsplit = (int)nsplit;
  *nslab_out = (int)nsplit * (KSPLIT ? 4 : 1);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, KH, KW>), dim3((unsigned)(tiles * nsplit)), dim3(256), 0, stream, p);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

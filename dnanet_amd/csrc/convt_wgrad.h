// Weight gradient of the transposed convolution as ONE large-tile split-K GEMM for gfx950.
//
// Replaces the weight half of autograd's backward of nn.ConvTranspose2d(k=(1,2), s=(1,2)) (reference
// unet_architecture.py:54-57, driven by loss.backward() at trainable_unet.py:342):
//
//   dW[ci][co][k] = sum_{n,h,w} x[n,h,w][ci] * dY[n,h,2w+k][co]
//
// GEMM: M = (co, k) rows (operand dY), N = ci (operand x), K = pixels.  There are no taps and no halo, so a plane of
// one image is a contiguous run of H*W records and the K loop walks strips of SW consecutive pixels of it.  Both
// operands are pixel-major in HBM, i.e. K is the slow index of every fragment: bf16 fragments come out of LDS through
// the transposing read ds_read_b64_tr_b16, fp32 fragments are one value per lane (see conv_wgrad.h).
//
// Why a kernel of its own (conv_wgrad.h's 1x1 instantiation did this job before): that one computes 64 x 128 tiles with
// one K step per wave and barrier, so dY is streamed from L2 once per input-channel tile and x once per output-channel
// tile -- 4.8 GB through L2 and 2.2 GB from HBM (PMC) for 0.8 GB of operands on the 512 -> 256 layer, at 1.35 TB/s.
// Here a block owns a TM x TN tile of dW large enough that the whole output is 1-4 tiles (256 x 256 for the two deep
// layers: 8 waves x 128 accumulator registers, one workgroup per CU), so an operand is read at most twice, and the K
// range of a split is tied to an XCD (split = 8 * group + blockIdx % 8): the tiles of a split run on the same L2 at the
// same time and a strip comes from HBM once.  A row group of 32 holds BOTH output columns k of 16 (8) output channels,
// i.e. both 32-byte halves of every 64-byte pair of dY records: no cache line is fetched for half of its bytes.
// Split-K partials go to slabs [slab][row = k * gsize + co][ci] that wgrad_reduce_kernel sums in fixed order (bitwise
// reproducible; no float atomics).
#pragma once
#include "conv_wgrad.h"

struct ConvtWgradParams {
  const void* dy; int dy_pt, dy_p0, dy_nplanes;     // dY: [N][planes][H][2*Win] records
  const void* x;  int x_pt, x_p0, x_nplanes;        // x : [N][planes][H][Win] records
  float* slab;
  int N, HW;                                        // images, pixels per image of x (H * Win)
  int gsize;                                        // rows per output column k in a slab (slab rows = 2 * gsize)
  int ci_pad;                                       // slab columns
  int nsplit, nstrips, spi;                         // K splits, strips in all, strips per image
  int n_mt, n_nt;                                   // tiles along M and N
};

// TM x TN: block tile (rows of (co, k) x input channels); WM x WN x WK: waves along M, N and over the K steps of a strip
template <typename T, int TM, int TN, int WM, int WN, int WK, int SW, int MINB>
__global__ __launch_bounds__(64 * WM * WN * WK, MINB) void convt_wgrad_kernel(const ConvtWgradParams p) {
  using TR = WgTraits<T>;
  constexpr int EPR = RecTraits<T>::EPR, KSTEP = TR::KSTEP;
  constexpr int NW = WM * WN * WK;
  constexpr int AM = TM / WM / 32, AN = TN / WN / 32;       // 32-channel groups per wave
  constexpr int PLG = 32 / EPR;                             // planes per 32-channel group
  constexpr int M_PL = TM / EPR, N_PL = TN / EPR;           // row planes / input-channel planes of the tile
  constexpr int RAW = SW * 32;
  constexpr int PS = RAW + (TR::PHASE - RAW % TR::BANKW + TR::BANKW) % TR::BANKW;   // plane stride: the planes of one read hit distinct banks
  constexpr int M_BYTES = (M_PL * PS + 1023) / 1024 * 1024; // the dY region ends on a 1-KiB piece boundary: one descriptor per wave-instruction
  constexpr int IMG_BYTES = M_BYTES + N_PL * PS;
  constexpr int ST_ITERS = (IMG_BYTES + NW * 1024 - 1) / (NW * 1024);
  constexpr int BUF_BYTES = ST_ITERS * NW * 1024;
  constexpr int NBUF = MINB * 3 * BUF_BYTES <= 160 * 1024 ? 3 : 2;   // strips in LDS: one being read, the others in flight (a 4th measured nothing)
  static_assert(MINB * NBUF * BUF_BYTES <= 160 * 1024, "LDS budget");
  static_assert((NBUF - 2) * ST_ITERS < 64, "vmcnt is a 6-bit counter");
  constexpr int NK = SW / KSTEP;                            // K steps per strip
  static_assert(NK % WK == 0 && AM >= 1 && AN >= 1 && TM % (WM * 32) == 0 && TN % (WN * 32) == 0, "tile / wave shape");
  constexpr int NIT = NK / WK;                              // K steps per wave and strip
  constexpr int RED_BYTES = WM * WN * (WK - 1) * AM * AN * 16 * 64 * 4;      // cross-wave reduction of the K-split waves (epilogue)
  constexpr int SMEM_BYTES = NBUF * BUF_BYTES > RED_BYTES ? NBUF * BUF_BYTES : RED_BYTES;
  static_assert(MINB * SMEM_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem_all[SMEM_BYTES];
  const unsigned smem_lds = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem_all);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);

  // block -> (tile, split): blocks b and b + 8 share an XCD; the split is tied to the XCD, tiles are the fast index
  const int tiles = p.n_mt * p.n_nt;
  const int j = blockIdx.x >> 3;
  const int t = j % tiles;
  const int nt = t % p.n_nt, mt = t / p.n_nt;
  const int split = (j / tiles) * 8 + (blockIdx.x & 7);
  if (split >= p.nsplit) return;                            // whole block: the grid is padded to a multiple of 8 splits

  f32x16_t acc[AM][AN];
#pragma unroll
  for (int a = 0; a < AM; ++a)
#pragma unroll
    for (int c = 0; c < AN; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  // staging plan: LDS piece (it * NW + wave) * 64 + lane of the image [row planes | pad | input-channel planes], plane
  // stride PS.  Row plane t of the tile = (output-channel plane (mt * M_PL + t) >> 1, output column k = (mt * M_PL + t) & 1):
  // dY record (plane, pixel q, k) sits at ((plane * HW + q) * 2 + k) * 32 inside the image.  Offsets are relative to the
  // strip's first pixel; absent planes and the padding between planes are zero-filled (offset out of range).
  unsigned st_off[ST_ITERS];
#pragma unroll
  for (int it = 0; it < ST_ITERS; ++it) {
    const int byte = ((it * NW + wave) * 64 + lane) * 16;
    unsigned off = DN_OOB;
    if (byte < M_BYTES) {
      const int pl = byte / PS, within = byte - pl * PS;
      if (pl < M_PL && within < RAW) {
        const int q = within >> 5, pc = (within >> 4) & 1;
        const int rp = mt * M_PL + pl, plane = rp >> 1, k = rp & 1;
        if (plane < p.dy_nplanes) off = (unsigned)((((long)plane * p.HW + q) * 2 + k) * 32 + 16 * pc);
      }
    } else if (byte < IMG_BYTES) {
      const int b2 = byte - M_BYTES;
      const int pl = b2 / PS, within = b2 - pl * PS;
      if (within < RAW) {
        const int q = within >> 5, pc = (within >> 4) & 1;
        const int plane = nt * N_PL + pl;
        if (plane < p.x_nplanes) off = (unsigned)(((long)plane * p.HW + q) * 32 + 16 * pc);
      }
    }
    st_off[it] = off;
  }
  const unsigned char* dyb = (const unsigned char*)p.dy;
  const unsigned char* xb = (const unsigned char*)p.x;
  auto stage = [&](int strip, int buf) {
    const bool exists = strip < p.nstrips;                  // a strip past the end is "staged" as zeros nobody reads: vmcnt stays uniform
    const int n = exists ? strip / p.spi : 0;
    const long px0 = exists ? (long)(strip - n * p.spi) * SW : 0;
    dn_srd_t srd_dy = dn_make_srd(dyb + ((long)(n * p.dy_pt + p.dy_p0) * p.HW * 2 + px0 * 2) * 32);
    dn_srd_t srd_x = dn_make_srd(xb + ((long)(n * p.x_pt + p.x_p0) * p.HW + px0) * 32);
    if (!exists) { srd_dy.z = 0; srd_x.z = 0; }
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
      const bool is_x = (it * NW + wave) * 1024 >= M_BYTES;               // wave-uniform
      dn_buffer_lds16(st_off[it], is_x ? srd_x : srd_dy,
                      __builtin_amdgcn_readfirstlane(smem_lds + buf * BUF_BYTES + (it * NW + wave) * 1024));
    }
  };

  // fragment addressing.  bf16 (ds_read_b64_tr_b16): 16-lane group g reads pixels 8 * (g >> 1).. of plane (g & 1) of the
  // 32-channel group; fp32: lane l31 reads channel l31 & 7 of plane l31 >> 3 at pixel 2 * ks + half.
  const int gpl_c = (lane >> 4) & 1;
  const int frag_off = sizeof(T) == 2 ? gpl_c * PS + (8 * (lane >> 5) + ((lane & 15) >> 2)) * 32 + 8 * (lane & 3)
                                      : (l31 >> 3) * PS + half * 32 + (l31 & 7) * 4;
  const int m_off = (wm * AM) * PLG * PS, n_off = M_BYTES + (wn * AN) * PLG * PS;
  using Frag = std::conditional_t<sizeof(T) == 2, uint4, float>;
  auto rd = [&](const unsigned char* base) -> Frag {
    if constexpr (sizeof(T) == 2) {
      const uint2 lo = lds_tr16_b64(base), hi = lds_tr16_b64(base + 4 * 32);
      return make_uint4(lo.x, lo.y, hi.x, hi.y);
    } else {
      return *(const float*)base;
    }
  };
  auto mma = [&](const Frag& a, const Frag& b, f32x16_t& c) {
    if constexpr (sizeof(T) == 2)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  };

  // LDS-DMA completes in issue order, so "at most (NBUF-2)*ST_ITERS of my loads outstanding" means the oldest strip in
  // flight has landed
  int buf = 0;
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) stage(split + i * p.nsplit, i);
  for (int strip = split; strip < p.nstrips; strip += p.nsplit, buf = buf + 1 == NBUF ? 0 : buf + 1) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * ST_ITERS) : "memory");
    __syncthreads();                                        // everyone's pieces landed; everyone left the oldest buffer
    stage(strip + (NBUF - 1) * p.nsplit, buf == 0 ? NBUF - 1 : buf - 1);
    const unsigned char* smem = smem_all + buf * BUF_BYTES + frag_off;
    Frag af[2][AM], bf[2][AN];
    auto load = [&](int it, Frag (&a)[AM], Frag (&b)[AN]) {
      const int ks = it * WK + wk;
      const int koff = ks * KSTEP * 32;
#pragma unroll
      for (int g = 0; g < AM; ++g) a[g] = rd(smem + m_off + g * PLG * PS + koff);
#pragma unroll
      for (int g = 0; g < AN; ++g) b[g] = rd(smem + n_off + g * PLG * PS + koff);
    };
    load(0, af[0], bf[0]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it + 1 < NIT) load(it + 1, af[(it + 1) & 1], bf[(it + 1) & 1]);
#pragma unroll
      for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int c = 0; c < AN; ++c) mma(af[it & 1][a], bf[it & 1][c], acc[a][c]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may outlive the wave

  if constexpr (WK > 1) {
    // the WK waves of a (wm, wn) position hold partial sums over different K steps: add them up through LDS in a fixed
    // order (wk = 0, 1, 2, ...), so the block writes ONE slab -- 4 x fewer partials for the reduction kernel to walk
    constexpr int PER_WAVE = AM * AN * 16 * 64;             // floats
    __syncthreads();                                        // every wave is done with the staged strips, every DMA has landed
    float* red = (float*)smem_all + ((wm * WN + wn) * (WK - 1)) * PER_WAVE;
    if (wk > 0) {
#pragma unroll
      for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int c = 0; c < AN; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(wk - 1) * PER_WAVE + ((a * AN + c) * 16 + r) * 64 + lane] = acc[a][c][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int w = 1; w < WK; ++w)
#pragma unroll
      for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int c = 0; c < AN; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][c][r] += red[(w - 1) * PER_WAVE + ((a * AN + c) * 16 + r) * 64 + lane];
  }

  // ---- this wave's partial tile -> its slab.  Accumulator register r of lane (l31, half): row (r & 3) + 8 (r >> 2) + 4 half
  // of the 32-row group, column l31.  Row rr of the group lives in row plane rp = group's first plane + rr / EPR:
  // output channel (rp >> 1) * EPR + rr % EPR, output column k = rp & 1 -> slab row k * gsize + channel.
  float* slab = p.slab + (long)split * 2 * p.gsize * p.ci_pad;
#pragma unroll
  for (int a = 0; a < AM; ++a)
#pragma unroll
    for (int c = 0; c < AN; ++c) {
      const int ci = nt * TN + (wn * AN + c) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int rp = mt * M_PL + (wm * AM + a) * PLG + rr / EPR;
        const int co = (rp >> 1) * EPR + rr % EPR, k = rp & 1;
        if (co < p.gsize && ci < p.ci_pad) slab[((long)k * p.gsize + co) * p.ci_pad + ci] = acc[a][c][r];
      }
    }
}

// slabs a launch writes = nsplit; ci_pad / gsize are set here (the reduction needs them)
template <typename T, int TM, int TN, int WM, int WN, int WK, int SW, int MINB>
int convt_wgrad_launch(ConvtWgradParams& p, int cout, long slab_bytes, int* nslab_out, hipStream_t stream) {
  constexpr int EPR = RecTraits<T>::EPR;
  if (p.HW % SW) return dn_set_error(DN_ERR_UNSUPPORTED, "convt_wgrad: H * W = %d is not a multiple of the strip width %d", p.HW, SW);
  if ((long)p.dy_pt * p.HW * 2 * 32 >= (1L << 31) || (long)p.x_pt * p.HW * 32 >= (1L << 31))
    return dn_set_error(DN_ERR_ARG, "convt_wgrad: one image of an operand exceeds 2 GiB (32-bit buffer offsets)");
  const int rows = 2 * p.dy_nplanes * EPR;                   // (co, k) rows that exist (whole planes)
  p.n_mt = (rows + TM - 1) / TM;
  p.n_nt = (p.x_nplanes * EPR + TN - 1) / TN;
  p.gsize = (cout + 31) / 32 * 32;
  if (p.gsize < p.dy_nplanes * EPR) p.gsize = p.dy_nplanes * EPR;
  p.ci_pad = p.n_nt * TN;
  p.spi = p.HW / SW;
  p.nstrips = p.N * p.spi;
  const long tiles = (long)p.n_mt * p.n_nt;
  const long per_slab = (long)2 * p.gsize * p.ci_pad * 4;
  long nsplit = std::max<long>(8, (256L * MINB / tiles + 7) / 8 * 8);      // every CU busy; a multiple of 8 (XCD-tied splits)
  while (nsplit > 8 && (nsplit * per_slab > slab_bytes || p.nstrips / nsplit < 4)) nsplit -= 8;
  if (nsplit * per_slab > slab_bytes)
    return dn_set_error(DN_ERR_ARG, "convt_wgrad: workspace of %ld bytes is smaller than %ld partial slabs (%ld)", slab_bytes, nsplit,
                        nsplit * per_slab);
  p.nsplit = (int)nsplit;
  *nslab_out = (int)nsplit;
  hipLaunchKernelGGL((convt_wgrad_kernel<T, TM, TN, WM, WN, WK, SW, MINB>), dim3((unsigned)(tiles * nsplit)), dim3(64 * WM * WN * WK), 0,
                     stream, p);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

// 3x5 implicit-GEMM convolution, 8-wave software-pipelined variant for the mid / deep U-Net levels (gfx950).
//
// Same operator, operands, LDS swizzle and epilogue as conv_igemm.h (reference unet_architecture.py:14-20); what
// changes is the schedule.  conv_igemm.h stages one K chunk, waits, computes, and relies on the CU's second
// workgroup to cover the load phase; per chunk each of the two workgroups pulls its own weight tile, and the L2->LDS
// path (about 20 B/clk per CU) ends up as busy as the matrix pipe.  Here ONE workgroup of 8 waves owns the CU:
//   * tile = HT dye rows x 256 scan columns x 64 output channels: the 30-KB weight tile of a chunk is shared by twice
//     the pixels (bytes staged per MFMA x0.73);
//   * the activation tile is double buffered: chunk k+1 streams in (LDS-DMA) during the first two of the five
//     kernel-column steps of chunk k;
//   * the weight tile is kept once, as five kernel-column slices [dwi][dhi][co]: the slice a step has just finished
//     with is refilled with the slice the same step needs four steps later (next chunk) -- a ring with a fixed lead;
//   * one barrier per kernel-column step; every wave issues the same number of LDS-DMA pieces per step (dummy pieces
//     go to a scrap KiB), so "my loads older than the newest N have landed" is a compile-time vmcnt per step.
// LDS: 2 x act (<= 57 KiB) + 30 KiB weights + 1 KiB scrap <= 146 KiB of the CU's 160.
#pragma once
#include <type_traits>
#include "conv_igemm.h"

template <typename T, int HT, int EPI = CONV_EPI_PLAIN>
__global__ __launch_bounds__(512, 1) void conv_igemm8_kernel(const ConvParams p) {
  constexpr bool STATS = EPI == CONV_EPI_STATS;
  constexpr int KH = 3, KW = 5, PH = 1, PW = 2, COT = 64, NCG = 2, TW = 256;
  constexpr int ROWS = HT + KH - 1;
  constexpr int QW = TW + KW - 1;
  constexpr int EPR = RecTraits<T>::EPR;
  constexpr int ACT_RAW = ROWS * QW * 32;
  constexpr int ACT_PIECES = (ACT_RAW + 1023) / 1024;
  constexpr int ACT_STEPS = 2;                                        // kernel-column steps that carry act pieces
  constexpr int APW = (ACT_PIECES + 8 * ACT_STEPS - 1) / (8 * ACT_STEPS);   // act pieces per wave per such step
  constexpr int ACT_BUF = ACT_PIECES * 1024;
  constexpr int W_SLICE = KH * COT * 32;                              // one kernel column: 6 pieces, waves 0..5
  constexpr int W_OFF = 2 * ACT_BUF, SCRAP_OFF = W_OFF + KW * W_SLICE;
  constexpr int SMEM_BYTES = SCRAP_OFF + 1024;
  static_assert(W_SLICE == 6 * 1024 && SMEM_BYTES <= 160 * 1024, "LDS plan");
  static_assert(8 * (COT / EPR) * 1536 + 8 * COT * 8 <= SMEM_BYTES, "epilogue scratch (stores, pooled, statistics) fits the tile buffers");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  const unsigned smem_lds = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
#ifdef C8_PROFILE_WAIT
  const long long prof_entry = clock64();
#endif

  // block -> tile map: as conv_igemm.h (channel tiles fast inside an XCD)
  const int npt = p.N * p.nth * p.ntw;
  const int ncot = p.cout_pad / COT;
  const int j = blockIdx.x >> 3;
  const int cot = j % ncot;
  int pt = (j / ncot) * 8 + (blockIdx.x & 7);
  if (pt >= npt) return;
  const int ptile = pt;      // index of this tile's BatchNorm partial (STATS)
  const int tw = pt % p.ntw; pt /= p.ntw;
  const int th = pt % p.nth;
  const int n = pt / p.nth;
  const int w0 = tw * TW, h0 = th * HT;
  const int xrow = p.W * p.x_wmul;

  // staging plan, fixed for the block.  Activation piece P = i*8 + wave (i < ACT_STEPS*APW) of the linear tile image;
  // weight piece = wave (< 6) of a kernel-column slice, relative to that slice's first tap.
  unsigned a_off[ACT_STEPS * APW];
  {
    const int xrow_b = xrow * 32;
#pragma unroll
    for (int i = 0; i < ACT_STEPS * APW; ++i) {
      const int byte = ((i * 8 + wave) * 64 + lane) * 16;
      unsigned off = DN_OOB;
      if (byte < ACT_RAW) {
        const int rq = byte >> 5, pcl = (byte >> 4) & 1;
        const int r = rq / QW, q = rq - r * QW;
        const int h = h0 + r - PH, w = w0 + q - PW;
        if (h >= 0 && h < p.H && w >= 0 && w < p.W) off = h * xrow_b + w * p.x_wmul * 32 + 16 * (pcl ^ ((q >> 3) & 1));
      }
      a_off[i] = off;
    }
  }
  unsigned w_off = DN_OOB;
  if (wave < 6) {
    const int byte = (wave * 64 + lane) * 16;
    const int tc = byte >> 5, pcl = (byte >> 4) & 1;
    const int dhi = tc / COT, co = tc - dhi * COT;
    w_off = (unsigned)((dhi * KW * p.cout_pad + co) * 32 + 16 * (pcl ^ ((co >> 3) & 1)));
  }

  const unsigned char* xb = (const unsigned char*)p.x;
  const unsigned char* wb = (const unsigned char*)p.w;
  // a chunk that does not exist (past the end / padding plane of a virtual group) reads from an empty buffer = zeros
  auto chunk_exists = [&](int ck) { return ck < p.nchunk && (ck % p.x_gplanes) < p.x_nplanes; };
  auto srd_act = [&](int ck) {
    const int grp = ck / p.x_gplanes, plane = ck - grp * p.x_gplanes;
    const bool ex = chunk_exists(ck);
    dn_srd_t s = dn_make_srd(xb + (ex ? (((long)(n * p.x_planes_total + p.x_plane0 + plane + ((grp & 1) ? p.x_lo_off : 0)) * p.H) * xrow + (long)grp * p.x_gstride) * 32 : 0));
    if (!ex) s.z = 0;
    return s;
  };
  auto srd_wslice = [&](int ck, int dwi) {
    const bool ex = chunk_exists(ck);
    dn_srd_t s = dn_make_srd(wb + (ex ? (((long)ck * KH * KW + dwi) * p.cout_pad + (long)cot * COT) * 32 : 0));
    if (!ex) s.z = 0;
    return s;
  };
  auto issue_act = [&](const dn_srd_t& srd, int abuf, int step) {     // APW pieces of one activation image
#ifdef C8_ABLATE_NO_STAGE   // (ablation hook: scratch/conv8_micro.hip)
    return;
#endif
#pragma unroll
    for (int k = 0; k < APW; ++k) {
      const int i = step * APW + k;
      const bool real = (i * 8 + wave) < ACT_PIECES;                  // wave-uniform; the rest lands on the scrap KiB
      dn_buffer_lds16(a_off[i], srd, __builtin_amdgcn_readfirstlane(
                          smem_lds + (real ? abuf * ACT_BUF + (i * 8 + wave) * 1024 : SCRAP_OFF)));
    }
  };
  auto issue_w = [&](const dn_srd_t& srd, int dwi) {                  // one piece of one weight slice
#ifdef C8_ABLATE_NO_STAGE
    return;
#endif
    dn_buffer_lds16(w_off, srd, __builtin_amdgcn_readfirstlane(smem_lds + (wave < 6 ? W_OFF + dwi * W_SLICE + wave * 1024 : SCRAP_OFF)));
  };

  f32x16_t acc[HT][NCG];
#pragma unroll
  for (int a = 0; a < HT; ++a)
#pragma unroll
    for (int b = 0; b < NCG; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // ---- prologue: chunk 0 entirely
  {
    const dn_srd_t sa = srd_act(0);
#pragma unroll
    for (int s = 0; s < ACT_STEPS; ++s) issue_act(sa, 0, s);
#pragma unroll
    for (int dwi = 0; dwi < KW; ++dwi) issue_w(srd_wslice(0, dwi), dwi);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

#ifdef C8_PROFILE_WAIT
  long long prof_wait = 0, prof_bar = 0;
  const long long prof_t0 = clock64(), prof_w0 = wall_clock64();
#endif
  auto mainloop = [&](auto skip_top_c, auto skip_bot_c) {
    constexpr bool SKIP_TOP = decltype(skip_top_c)::value, SKIP_BOT = decltype(skip_bot_c)::value;
    // fragments of the NEXT step are read at the end of the current one (activation rows: all; weights: kernel row 0),
    // so that after a barrier the matrix pipe starts at once instead of waiting for all 8 waves' LDS reads
    uint4 bf[ROWS], af0[NCG];
    auto load_bf = [&](const unsigned char* act, int dwi) {
      const int q = 32 * wave + l31 + dwi;
      const int boff = q * 32 + 16 * (half ^ ((q >> 3) & 1));
#pragma unroll
      for (int r = 0; r < ROWS; ++r) bf[r] = *(const uint4*)(act + r * QW * 32 + boff);
    };
    auto load_af = [&](uint4 (&af)[NCG], int dwi, int dhi) {
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        const int col = cg * 32 + l31;
        af[cg] = *(const uint4*)(smem + W_OFF + dwi * W_SLICE + (dhi * COT + col) * 32 + 16 * (half ^ ((col >> 3) & 1)));
      }
    };
    __syncthreads();   // prologue pieces of every wave have landed
    load_bf(smem, 0);
    load_af(af0, 0, 0);
    for (int ck = 0; ck < p.nchunk; ++ck) {
      const unsigned char* act = smem + (ck & 1) * ACT_BUF;
      const unsigned char* act_next = smem + ((ck + 1) & 1) * ACT_BUF;
      const dn_srd_t sa_next = srd_act(ck + 1);
      auto step = [&](auto dwi_c) {
        constexpr int dwi = decltype(dwi_c)::value;
        // Per wave and step the issue order is [act pieces (steps 0, 1 only), weight piece].  At the top of a step the
        // weight slice of the NEXT step (issued three steps ago) must have landed -- and at step 4 the next chunk's
        // activations (issued in steps 0-1, older); everything issued after that may still be in flight:
        constexpr int NEWER = dwi == 0 ? 2 : dwi == 1 ? 2 + APW : dwi == 2 ? 2 + 2 * APW : dwi == 3 ? 2 + APW : 2;
#ifdef C8_PROFILE_WAIT
        const long long tw0 = clock64();
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER) : "memory");
#ifdef C8_PROFILE_WAIT
        const long long tw1 = clock64();
#endif
#ifndef C8_ABLATE_NO_BARRIER
        __syncthreads();   // everyone's pieces landed; everyone is done with the previous step's slice / act image
#endif
#ifdef C8_PROFILE_WAIT
        prof_wait += tw1 - tw0; prof_bar += clock64() - tw1;
#endif
        if (dwi < ACT_STEPS) issue_act(sa_next, (ck + 1) & 1, dwi);
        // refill the slice the previous step finished with: step 0 -> slice 4 of THIS chunk, step d -> slice d-1 of the next
        if (dwi == 0) issue_w(srd_wslice(ck, KW - 1), KW - 1);
        else issue_w(srd_wslice(ck + 1, dwi - 1), dwi - 1);

        // weight fragments rotate through two register sets: row 1 is read before row 0's MFMAs, row 2 after them
        uint4 af1[NCG];
        load_af(af1, dwi, 1);
#pragma unroll
        for (int dhi = 0; dhi < KH; ++dhi) {
#pragma unroll
          for (int hh = 0; hh < HT; ++hh) {
            const int hin = hh + dhi - PH;
            if (!(SKIP_TOP && hin < 0) && !(SKIP_BOT && hin >= HT)) {   // compile time
#pragma unroll
              for (int cg = 0; cg < NCG; ++cg) Mma32<T>::run(dhi == 1 ? af1[cg] : af0[cg], bf[hh + dhi], acc[hh][cg]);
            }
          }
          if (dhi == 0) load_af(af0, dwi, 2);
        }
        // next step's fragments
        if (dwi + 1 < KW) load_bf(act, dwi + 1); else load_bf(act_next, 0);
        load_af(af0, (dwi + 1) % KW, 0);
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
    }
  };
  {
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    const bool top = h0 == 0, bot = h0 + HT >= p.H;   // block-uniform
    if (top && bot) mainloop(yes{}, yes{});
    else if (top) mainloop(yes{}, no{});
    else if (bot) mainloop(no{}, yes{});
    else mainloop(no{}, no{});
  }

  constexpr int NPL = COT / EPR;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef C8_PROFILE_WAIT
  if (lane == 0 && p.dbg) {
    long long* d = p.dbg + ((long)blockIdx.x * 8 + wave) * 3;
    d[0] = prof_wait; d[1] = (wall_clock64() - prof_w0) | ((long long)prof_bar << 24); d[2] = clock64() - prof_t0;
  }
  const long long prof_t1 = clock64();
#endif
  float* stat_lds = (float*)(smem + 8 * NPL * 1024);                // [wave][COT][2], past every wave's scratch
  unsigned char* pool_lds = smem + 8 * NPL * 1024 + 8 * COT * 2 * 4;   // [wave][NPL][512 B]
  conv_epilogue_store<T, HT, COT, EPI>(p, acc, smem + wave * (NPL * 1024), stat_lds + wave * COT * 2,
                                       pool_lds + wave * (NPL * 512), n, h0, w0 + 32 * wave, cot, lane);
#ifdef C8_PROFILE_WAIT
  if (lane == 0 && p.dbg) {
    long long* d = p.dbg + 3L * 8 * gridDim.x + ((long)blockIdx.x * 8 + wave) * 3;
    d[0] = prof_t0 - prof_entry;           // prologue
    d[1] = clock64() - prof_t1;            // epilogue up to the last store issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    d[2] = clock64() - prof_t1;            // ... up to the last store acknowledged
  }
#endif
  // BatchNorm statistics of this 256-column tile: sum the 8 waves' (sum, sum of squares), fixed order
  if constexpr (STATS || EPI == CONV_EPI_BNRED) {
    __syncthreads();
    if (tid < COT * 2) {
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) t += stat_lds[wv * COT * 2 + tid];
      p.stats[((long)ptile * p.cout_pad + cot * COT) * 2 + tid] = t;
    }
  }
}

// ---- chunk-granular variant for 3-row tiles (6-dye profiles): a kernel-column step of a 3-row tile is only ~16 MFMAs
// per wave, too few to pay for a barrier each, and its LDS images are small enough to double buffer BOTH operands
// (2 x 41.6 KiB activations + 2 x 30 KiB weights = 143 KiB).  One barrier per K chunk; the whole next chunk streams
// in while the current one (84-90 MFMAs per wave) is consumed.
template <typename T, int HT, int EPI = CONV_EPI_PLAIN>
__global__ __launch_bounds__(512, 1) void conv_igemm8c_kernel(const ConvParams p) {
  constexpr bool STATS = EPI == CONV_EPI_STATS;
  constexpr int KH = 3, KW = 5, TAPS = 15, PH = 1, PW = 2, COT = 64, NCG = 2, TW = 256;
  constexpr int ROWS = HT + KH - 1;
  constexpr int QW = TW + KW - 1;
  constexpr int EPR = RecTraits<T>::EPR;
  constexpr int ACT_RAW = ROWS * QW * 32;
  constexpr int ACT_BYTES = (ACT_RAW + 1023) / 1024 * 1024;
  constexpr int W_BYTES = TAPS * COT * 32;
  constexpr int ST_ITERS = (ACT_BYTES + W_BYTES + 8191) / 8192;       // 1-KiB pieces per wave (8 waves)
  constexpr int BUF_BYTES = ST_ITERS * 8192;
  static_assert(2 * BUF_BYTES <= 160 * 1024, "both operands double buffered");
  static_assert(8 * (COT / EPR) * 1536 + 8 * COT * 8 <= 2 * BUF_BYTES, "epilogue scratch (stores, pooled, statistics) fits the tile buffers");
  __shared__ __attribute__((aligned(16))) unsigned char smem_all[2 * BUF_BYTES];
  const unsigned smem_lds = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem_all);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  const int npt = p.N * p.nth * p.ntw;
  const int ncot = p.cout_pad / COT;
  const int j = blockIdx.x >> 3;
  const int cot = j % ncot;
  int pt = (j / ncot) * 8 + (blockIdx.x & 7);
  if (pt >= npt) return;
  const int ptile = pt;      // index of this tile's BatchNorm partial (STATS)
  const int tw = pt % p.ntw; pt /= p.ntw;
  const int th = pt % p.nth;
  const int n = pt / p.nth;
  const int w0 = tw * TW, h0 = th * HT;
  const int xrow = p.W * p.x_wmul;

  unsigned st_off[ST_ITERS];
  {
    const int xrow_b = xrow * 32;
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
      const int byte = ((it * 8 + wave) * 64 + lane) * 16;
      unsigned off = DN_OOB;
      if (byte < ACT_RAW) {
        const int rq = byte >> 5, pcl = (byte >> 4) & 1;
        const int r = rq / QW, q = rq - r * QW;
        const int h = h0 + r - PH, w = w0 + q - PW;
        if (h >= 0 && h < p.H && w >= 0 && w < p.W) off = h * xrow_b + w * p.x_wmul * 32 + 16 * (pcl ^ ((q >> 3) & 1));
      } else if (byte >= ACT_BYTES && byte < ACT_BYTES + W_BYTES) {
        const int bw = byte - ACT_BYTES;
        const int tc = bw >> 5, pcl = (bw >> 4) & 1;
        const int tap = tc / COT, co = tc - tap * COT;
        off = (tap * p.cout_pad + co) * 32 + 16 * (pcl ^ ((co >> 3) & 1));
      }
      st_off[it] = off;
    }
  }

  const unsigned char* xb = (const unsigned char*)p.x;
  const unsigned char* wb = (const unsigned char*)p.w;
  auto stage = [&](int ck, int buf) {
    const int grp = ck / p.x_gplanes;
    const int plane = ck - grp * p.x_gplanes;
    const bool exists = ck < p.nchunk && plane < p.x_nplanes;
    dn_srd_t srd_x = dn_make_srd(xb + (exists ? (((long)(n * p.x_planes_total + p.x_plane0 + plane + ((grp & 1) ? p.x_lo_off : 0)) * p.H) * xrow + (long)grp * p.x_gstride) * 32 : 0));
    dn_srd_t srd_w = dn_make_srd(wb + (exists ? ((long)ck * TAPS * p.cout_pad + (long)cot * COT) * 32 : 0));
    if (!exists) { srd_x.z = 0; srd_w.z = 0; }
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
      const bool is_w = (it * 8 + wave) * 1024 >= ACT_BYTES;          // wave-uniform
      dn_buffer_lds16(st_off[it], is_w ? srd_w : srd_x,
                      __builtin_amdgcn_readfirstlane(smem_lds + buf * BUF_BYTES + (it * 8 + wave) * 1024));
    }
  };

  f32x16_t acc[HT][NCG];
#pragma unroll
  for (int a = 0; a < HT; ++a)
#pragma unroll
    for (int b = 0; b < NCG; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  stage(0, 0);
  auto mainloop = [&](auto skip_top_c, auto skip_bot_c) {
    constexpr bool SKIP_TOP = decltype(skip_top_c)::value, SKIP_BOT = decltype(skip_bot_c)::value;
    for (int ck = 0; ck < p.nchunk; ++ck) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // chunk ck landed for everyone; everyone left the other image
      stage(ck + 1, (ck + 1) & 1);
      const unsigned char* smem = smem_all + (ck & 1) * BUF_BYTES;
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
            af[cg] = *(const uint4*)(smem + ACT_BYTES + ((dhi * KW + dwi) * COT + col) * 32 + 16 * (half ^ ((col >> 3) & 1)));
          }
#pragma unroll
          for (int hh = 0; hh < HT; ++hh) {
            const int hin = hh + dhi - PH;
            if (!(SKIP_TOP && hin < 0) && !(SKIP_BOT && hin >= HT)) {   // compile time
#pragma unroll
              for (int cg = 0; cg < NCG; ++cg) Mma32<T>::run(af[cg], bf[hh + dhi], acc[hh][cg]);
            }
          }
        }
      }
    }
  };
  {
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    const bool top = h0 == 0, bot = h0 + HT >= p.H;   // block-uniform
    if (top && bot) mainloop(yes{}, yes{});
    else if (top) mainloop(yes{}, no{});
    else if (bot) mainloop(no{}, yes{});
    else mainloop(no{}, no{});
  }

  constexpr int NPL = COT / EPR;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* stat_lds = (float*)(smem_all + 8 * NPL * 1024);            // [wave][COT][2], past every wave's scratch
  unsigned char* pool_lds = smem_all + 8 * NPL * 1024 + 8 * COT * 2 * 4;   // [wave][NPL][512 B]
  conv_epilogue_store<T, HT, COT, EPI>(p, acc, smem_all + wave * (NPL * 1024), stat_lds + wave * COT * 2,
                                       pool_lds + wave * (NPL * 512), n, h0, w0 + 32 * wave, cot, lane);
  // BatchNorm statistics of this 256-column tile: sum the 8 waves' (sum, sum of squares), fixed order
  if constexpr (STATS || EPI == CONV_EPI_BNRED) {
    __syncthreads();
    if (tid < COT * 2) {
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) t += stat_lds[wv * COT * 2 + tid];
      p.stats[((long)ptile * p.cout_pad + cot * COT) * 2 + tid] = t;
    }
  }
}

// the 8-wave kernel pays off once a block has enough K chunks to pipeline and the row holds whole 256-column tiles
inline bool conv_igemm8_applies(const ConvParams& p, int kh, int kw) {
  return kh == 3 && kw == 5 && p.cout_pad % 64 == 0 && p.W % 256 == 0 && p.nchunk >= 4 && p.x_wmul == 1 && p.y_wmul == 1;
}

template <typename T, int HT, bool CHUNKED = false, int EPI = CONV_EPI_PLAIN>
int conv_igemm8_launch(ConvParams p, hipStream_t stream) {
  p.ntw = p.W / 256;
  p.nth = (p.H + HT - 1) / HT;
  const long blocks = ((long)p.N * p.ntw * p.nth + 7) / 8 * 8 * (p.cout_pad / 64);
  if (blocks <= 0 || blocks > 0x7fffffffL) return dn_set_error(DN_ERR_ARG, "conv grid out of range: %ld blocks", blocks);
  if ((long)p.H * p.W * 32 >= (1L << 31)) return dn_set_error(DN_ERR_ARG, "conv: one plane exceeds 2 GiB");
  if constexpr (CHUNKED) hipLaunchKernelGGL((conv_igemm8c_kernel<T, HT, EPI>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
  else hipLaunchKernelGGL((conv_igemm8_kernel<T, HT, EPI>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

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
// Block = 4 waves, tile = CO_T x CI_T x all taps, K-loop over SW-column strips of one dye row
// (plus KH-1 halo rows of A), software pipelined through NBUF LDS buffers filled by LDS-DMA: strips s+1..
// streams in while strip s feeds the MFMAs, one barrier per strip; two such workgroups share a CU.
// Waves split the taps (3x5: 4/4/4/3); for kernels with < 4 taps they split the K steps instead and
// write one partial slab each.
// Split-K partials go to a slab [slab][tap][co_pad][ci_pad] that a second kernel sums in fixed order
// (bitwise reproducible; no float atomics) and scatters into the canonical parameter layout.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "common.h"

struct WgradParams {
  const void* dy; int dy_pt, dy_p0, dy_nplanes, dy_gplanes, dy_wmul;
  const void* a;  int a_pt, a_p0, a_nplanes;
  float* slab;
  int N, H, W;
  int co_pad, ci_pad;
  int nsplit, nstrips, ntw, nth;
  int skip_halo;            // leave out the products with zero halo rows (set by the launcher; tuning switch)
  int xcd_split;            // 1: block -> (tile, split) map that gives every XCD its own K ranges (set by the launcher)
};

template <typename T> struct WgTraits;
template <> struct WgTraits<__bf16> { static constexpr int CO_T = 64, CI_T = 32, KSTEP = 16, BANKW = 256, PHASE = 128; };
template <> struct WgTraits<float>  { static constexpr int CO_T = 32, CI_T = 32, KSTEP = 2,  BANKW = 128, PHASE = 32; };

// strip = HT dye rows x SW scan columns of one image.  Row tiles (HT > 1) exist for the 3-row kernels: the KH-1 halo
// rows of A are then fetched once per HT output rows instead of once per row, which is what moves the kernel from
// the L2 bandwidth bound to the MFMA bound (bytes staged per flop: x0.6 at HT = 5).
// input-channel tile: a single-tap (1x1 / transposed-convolution) gradient has one accumulator set per wave, so it can
// afford 128 input channels per tile -- dY is then re-read 4x less often (it is a plain [co x pixels] x [pixels x ci]
// GEMM that is bound by how often the operands are re-streamed from L2)
constexpr int wg_ci_tile(int taps, int elem_bytes) { return taps == 1 ? (elem_bytes == 2 ? 128 : 64) : 32; }   // fp32: LDS budget
template <typename T> constexpr int wg_strip_width(int ht) { return ht == 1 ? 64 : (sizeof(T) == 2 ? 32 : 16); }

__device__ __forceinline__ uint2 lds_tr16_b64(const unsigned char* p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

// COT = output channels per block tile: 64 for bf16 (two 32-channel groups share every A fragment), 32 for layers with at
// most 32 output channels (a 64-wide tile would stage and multiply 32 channels of zeros) and for fp32.
template <typename T, int KH, int KW, int HT, int COT = WgTraits<T>::CO_T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p) {
  using TR = WgTraits<T>;
  constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
  constexpr int EPR = RecTraits<T>::EPR;
  constexpr int CO_T = COT, CI_T = wg_ci_tile(TAPS, sizeof(T)), KSTEP = TR::KSTEP;
  constexpr int NCOG = CO_T / 32, NCIG = CI_T / 32;
  constexpr int CO_PL = CO_T / EPR, CI_PL = CI_T / EPR;     // planes per tile
  constexpr int PLG = 32 / EPR;                             // planes per 32-channel group
  constexpr int SW = wg_strip_width<T>(HT);                 // strip width in pixels
  constexpr int QW = SW + KW - 1, AR = HT + KH - 1;         // A image: AR rows x QW columns per plane
  constexpr int DY_RAW = HT * SW * 32, A_RAW = AR * QW * 32;
  constexpr int DY_PS = DY_RAW + (TR::PHASE - DY_RAW % TR::BANKW + TR::BANKW) % TR::BANKW;   // plane strides chosen so the
  constexpr int A_PS = A_RAW + (TR::PHASE - A_RAW % TR::BANKW + TR::BANKW) % TR::BANKW;      // planes of one read hit distinct banks
  constexpr int DY_BYTES = (CO_PL * DY_PS + 1023) / 1024 * 1024, A_BYTES = CI_PL * A_PS;   // dY region ends on a 1-KB piece boundary
  constexpr bool KSPLIT = TAPS < 4;                         // waves split K steps instead of taps
  // COL4 (fp32 3x5): wave w owns kernel column w (KH accumulators) in every K step, and the KH taps of the LAST column are
  // shared by K: in each group of four K steps, wave w multiplies them for step w (KH more accumulators, partial sums that
  // the epilogue adds up through LDS in a fixed order).  15 MFMAs per wave and four K steps instead of 16 / 16 / 16 / 12:
  // no wave waits for the others.  fp32 has the registers for it (6 accumulators = 96); the bf16 tile (two output-channel
  // groups per tap) does not, and keeps the 4/4/4/3 map below.
  constexpr bool COL4 = sizeof(T) == 4 && KH == 3 && KW == 5 && (SW / TR::KSTEP) % 4 == 0;
  constexpr int WT = KSPLIT ? TAPS : (COL4 ? 2 * KH : (TAPS + 3) / 4);        // accumulator sets (taps) per wave
  constexpr int NK = SW / KSTEP;                            // K steps per row of the strip
  static_assert(!KSPLIT || (HT == 1 && NK % 4 == 0), "K steps must split over 4 waves");
  constexpr int ST_ITERS = (DY_BYTES + A_BYTES + 4095) / 4096;
  constexpr int BUF_BYTES = ST_ITERS * 4096;
  constexpr int NBUF = 2 * 3 * BUF_BYTES <= 160 * 1024 ? 3 : 2;   // strips in LDS: one being read, NBUF-1 in flight
  static_assert(2 * NBUF * BUF_BYTES <= 160 * 1024, "two workgroups per CU");
  static_assert((NBUF - 2) * ST_ITERS < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) unsigned char smem_all[NBUF * BUF_BYTES];
  const unsigned smem_lds = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem_all);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int n_cit = p.ci_pad / CI_T, n_cot = p.co_pad / CO_T;
  // tiles are the fast block index: blocks that run together walk the SAME strips (same split) for different
  // (co, ci) tiles, so a strip is fetched from HBM about once per XCD L2 instead of once per tile.  Blocks b and b + 8
  // share an XCD (and its L2): with xcd_split the split index is tied to the XCD (split = 8 * group + b % 8), so that an
  // XCD walks its own K ranges for ALL tiles and no strip is fetched by more than one XCD -- measured before: 2 x FETCH_SIZE
  // = 2.5 GB per launch against 0.8 GB of operands, dY being pulled into all eight L2s.
  int cit, cot, split;
  if (p.xcd_split) {
    const int j = blockIdx.x >> 3, tiles = n_cit * n_cot;
    const int t = j % tiles;
    cit = t % n_cit; cot = t / n_cit;
    split = (j / tiles) * 8 + (blockIdx.x & 7);
    if (split >= p.nsplit) return;          // whole block: the grid is padded to a multiple of 8 splits
  } else {
    int b = blockIdx.x;
    cit = b % n_cit; b /= n_cit;
    cot = b % n_cot;
    split = b / n_cot;
  }

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

  // this wave's taps as byte offsets into an A plane image.  A wave with fewer real taps than WT (3x5: wave 3) repeats
  // its last one into an accumulator that is never stored, which keeps the MFMA loop free of branches.
  // 3x5: accumulator i < 3 of wave w is tap (kernel row i, column w) -- the kernel row of an accumulator is a compile-time
  // constant, which lets the products with a zero halo row be left out below -- and accumulator 3 is column 4 of kernel
  // row w (waves 0..2; wave 3 has none).
  static_assert(KSPLIT || (KH == 3 && KW == 5 && (WT == 4 || COL4)), "tap-to-wave map is written for 3x5");
  auto tap_of = [&](int i) {
    if (COL4) return i < KH ? i * KW + wave : (i - KH) * KW + KW - 1;
    return KSPLIT ? i : (i < KH ? i * KW + wave : min(wave, KH - 1) * KW + KW - 1);
  };
  int tap_off[WT];
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int tap = tap_of(i);
    const int dh = tap / KW, dw = tap - dh * KW;
    tap_off[i] = __builtin_amdgcn_readfirstlane((dh * QW + dw) * 32);
  }
  const bool last_tap_real = KSPLIT || COL4 || wave < KH;           // wave-uniform (same waves under either map)
  (void)last_tap_real;
  // bf16 fragment addressing (ds_read_b64_tr_b16): 16-lane group g reads pixels 8*(g>>1).. of plane (g&1)
  const int gpl_c = (lane >> 4) & 1;
  const int frag_off = gpl_c * DY_PS + (8 * (lane >> 5) + ((lane & 15) >> 2)) * 32 + 8 * (lane & 3);
  (void)frag_off;

  // staging plan: LDS piece P = (it*4 + wave)*64 + lane of the linear image [dY planes | A planes] (plane strides
  // padded; pad pieces and absent planes are zero-filled).  st_off = byte offset of the piece relative to (row h0,
  // column w0) of the image's first plane, st_row / st_col = its row and column relative to (h0, w0) for the image
  // border checks (the halo makes them negative; a piece that never exists gets a row no image has).
  unsigned st_off[ST_ITERS];
  int st_row[ST_ITERS], st_col[ST_ITERS];
#pragma unroll
  for (int it = 0; it < ST_ITERS; ++it) {
    const int byte = ((it * 4 + wave) * 64 + lane) * 16;
    unsigned off = 0;
    int row = 1 << 30, col = 0;
    if (byte < DY_BYTES) {
      const int pl = byte / DY_PS, within = byte - pl * DY_PS;
      if (pl < CO_PL && within < DY_RAW) {
        const int rec = within >> 5, pc = (within >> 4) & 1;
        const int r = rec / SW, q = rec - r * SW;
        const int vp = cot * CO_PL + pl;
        const int grp = vp / p.dy_gplanes, plane = vp - grp * p.dy_gplanes;
        if (plane < p.dy_nplanes) {
          off = (unsigned)((((long)plane * p.H + r) * dy_row + q * p.dy_wmul + grp) * 32 + 16 * pc);
          row = r; col = q;
        }
      }
    } else if (byte < DY_BYTES + A_BYTES) {
      const int b2 = byte - DY_BYTES;
      const int pl = b2 / A_PS, within = b2 - pl * A_PS;
      if (within < A_RAW) {
        const int rec = within >> 5, pc = (within >> 4) & 1;
        const int r = rec / QW, q = rec - r * QW;
        const int plane = cit * CI_PL + pl;
        if (plane < p.a_nplanes) {
          off = (unsigned)((((long)plane * p.H + (r - PH)) * p.W + (q - PW)) * 32 + 16 * pc);   // may wrap; the sum below does not
          row = r - PH; col = q - PW;
        }
      }
    }
    st_off[it] = off; st_row[it] = row; st_col[it] = col;
  }

  // staging of one strip = uniform set-up (SALU) + ST_ITERS independent pieces, so that the pieces of the next strip
  // can be sprinkled between the MFMAs of the current one instead of stalling the matrix pipe in one burst
  struct StageCtx { dn_srd_t srd_dy, srd_a; unsigned so_dy, so_a, lds0; int h0, w0; };
  auto stage_begin = [&](int strip, int buf) {
    StageCtx c;
    // the row tile is the fast index: with nsplit a multiple of nth, all strips of a block (split + k * nsplit) sit at
    // the same height in the image, so "my strips touch the top / bottom border" is a per-block constant
    const int th = strip % p.nth, rest = strip / p.nth;
    const int tw = rest % p.ntw;
    const int n = rest / p.ntw;
    c.h0 = th * HT;
    c.w0 = tw * SW;
    c.srd_dy = dn_make_srd(dyb + (long)(n * p.dy_pt + p.dy_p0) * p.H * dy_row * 32);
    c.srd_a = dn_make_srd(ab + (long)(n * p.a_pt + p.a_p0) * p.H * p.W * 32);
    c.so_dy = (unsigned)(((long)c.h0 * dy_row + (long)c.w0 * p.dy_wmul) * 32);
    c.so_a = (unsigned)(((long)c.h0 * p.W + c.w0) * 32);
    c.lds0 = smem_lds + buf * BUF_BYTES + wave * 1024;
    return c;
  };
  auto stage_piece = [&](const StageCtx& c, int it) {
    const bool is_a = (it * 4 + wave) * 1024 >= DY_BYTES;              // wave-uniform
    const bool ok = ((unsigned)(c.h0 + st_row[it]) < (unsigned)p.H) & ((unsigned)(c.w0 + st_col[it]) < (unsigned)p.W);
    const unsigned off = ok ? st_off[it] + (is_a ? c.so_a : c.so_dy) : DN_OOB;
    dn_buffer_lds16(off, is_a ? c.srd_a : c.srd_dy, __builtin_amdgcn_readfirstlane(c.lds0 + it * 4096));
  };
  // a strip past the end is still "staged", with every offset out of range (zeros into a buffer nobody reads): the
  // loops stay branch-free and the vmcnt arithmetic is the same for every strip
  auto stage_begin_checked = [&](int strip, int buf) {
    const bool exists = strip < p.nstrips;
    StageCtx c = stage_begin(exists ? strip : 0, buf);
    if (!exists) c.h0 = 1 << 29;
    return c;
  };
  auto stage = [&](int strip, int buf) {
    const StageCtx c = stage_begin_checked(strip, buf);
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) stage_piece(c, it);
  };

  // LDS-DMA completes in issue order, so "at most (NBUF-2)*ST_ITERS of my loads outstanding" means the oldest
  // strip in flight has landed
  int buf = 0;
#ifndef WG_ABLATE_NO_STAGE
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i)
    stage(split + i * p.nsplit, i);
#endif
  // A strip at the top (bottom) of the image pairs its first (last) row with a zero halo row of A for kernel row 0
  // (KH-1): those products are left out at compile time for accumulators 0..2 (12 of the 15 taps) -- 11 % of the MFMAs
  // of a 5-row image.  Chosen once per block (the variants live outside the strip loop, inside it they cost 150
  // spilled registers): the strips of a block all sit at the same height of the image (see stage_begin).
  // LAST: this wave's last accumulator is a real tap (3x5, 4/4/4/3 map: waves 0..2).  A compile-time variant as well, so
  // that the wave without one neither reads those fragments nor carries a branch per K step in its MFMA stream.
  auto strips = [&](auto top_c, auto bot_c, auto last_c) {
  constexpr bool TOP = decltype(top_c)::value, BOT = decltype(bot_c)::value, LAST = decltype(last_c)::value;
  auto dead = [&](int i, int r) -> bool {       // (accumulator, strip row) sees only a zero row of A
    if (KSPLIT || (!COL4 && i >= KH)) return false;        // 4/4/4/3 map: accumulator 3's kernel row depends on the wave
    const int krow = i % KH;                               // COL4: accumulators KH.. are the last column's rows
    return (TOP && r == 0 && krow == 0) || (BOT && r == HT - 1 && krow == KH - 1);
  };
  for (int strip = split; strip < p.nstrips; strip += p.nsplit, buf = buf + 1 == NBUF ? 0 : buf + 1) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * ST_ITERS) : "memory");
    __syncthreads();                                    // everyone's pieces landed; everyone left the oldest buffer
    const int nxt_strip = strip + (NBUF - 1) * p.nsplit, nxt_buf = buf == 0 ? NBUF - 1 : buf - 1;
    (void)nxt_strip; (void)nxt_buf;
#ifndef WG_ABLATE_NO_STAGE
    if (sizeof(T) != 2) stage(nxt_strip, nxt_buf);
#endif
    const unsigned char* smem = smem_all + buf * BUF_BYTES;
    (void)smem;
#ifndef WG_ABLATE_NO_COMPUTE
    // ---- MFMA over the strip's pixels
    if constexpr (sizeof(T) == 2 && !KSPLIT) {
      // Fragment reads are what bounds this kernel (ds_read_b64_tr_b16: 1.5 per MFMA when every (tap, strip row) fetches
      // its own A fragment).  Accumulator i < KH of a wave is tap (kernel row i, column = wave): at strip row r it multiplies
      // A row r + i, so A row rho of one K step serves (i, r) = (0, rho), (1, rho - 1), (2, rho - 2).  The strip rows are
      // therefore the INNER loop and the wave keeps a sliding window of KH A-row fragments: HT + KH - 1 row fragments per
      // K step instead of KH * HT (3 rows: 28 instead of 36 reads per 24 MFMAs, 5 rows: 44 instead of 60 per 40).
      // Everything is unrolled: g[][] are plain registers whose live ranges the compiler sees.
      static_assert(NCIG == 1 && WT == KH + 1, "window scheme: one input-channel group, KH row taps + one extra per wave");
      constexpr int NIT = HT * NK;
      const unsigned char* dbase0 = smem + frag_off;
      const unsigned char* abase0 = smem + frag_off + DY_BYTES + (A_PS - DY_PS) * gpl_c;
      auto rd = [&](const unsigned char* base) {
#ifdef WG_ABLATE_NO_FRAG          // (ablation hook, scratch/wgrad_micro.hip: no fragment reads)
        const unsigned v = (unsigned)(size_t)base;
        return make_uint4(v, v, v, v);
#else
        const uint2 lo = lds_tr16_b64(base), hi = lds_tr16_b64(base + 4 * 32);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
#endif
      };
      uint4 af[2][NCOG], bx[2], g[NK][HT + KH - 1];
      auto load_step = [&](int s) {
        const int ks = s / HT, r = s - ks * HT;
#pragma unroll
        for (int cg = 0; cg < NCOG; ++cg) af[s & 1][cg] = rd(dbase0 + cg * PLG * DY_PS + (r * SW + ks * 16) * 32);
        if constexpr (LAST) bx[s & 1] = rd(abase0 + tap_off[KH] + (r * QW + ks * 16) * 32);
#pragma unroll
        for (int rho = 0; rho < HT + KH - 1; ++rho) {
          if (rho > r + KH - 1 || (r > 0 && rho < r + KH - 1)) continue;        // new rows of the window only
          if ((TOP && rho == 0) || (BOT && rho == HT + KH - 2)) continue;       // zero halo row: nobody multiplies it
          g[ks][rho] = rd(abase0 + tap_off[0] + (rho * QW + ks * 16) * 32);
        }
      };
      constexpr int PPI = (ST_ITERS + NIT - 1) / NIT;   // pieces of the next strip staged per K step
      load_step(0);
#ifndef WG_ABLATE_NO_STAGE
      const StageCtx sc = stage_begin_checked(nxt_strip, nxt_buf);
#endif
#pragma unroll
      for (int s = 0; s < NIT; ++s) {
        if (s + 1 < NIT) load_step(s + 1);
#ifndef WG_ABLATE_NO_STAGE
#pragma unroll
        for (int k = s * PPI; k < (s + 1) * PPI && k < ST_ITERS; ++k) stage_piece(sc, k);
#endif
        const int ks = s / HT, r = s - ks * HT;
#pragma unroll
        for (int i = 0; i < KH; ++i) {
          if (dead(i, r)) continue;
#pragma unroll
          for (int a = 0; a < NCOG; ++a)
            acc[i][a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[s & 1][a]),
                                                                   __builtin_bit_cast(bf16x8_t, g[ks][r + i]), acc[i][a][0], 0, 0, 0);
        }
        // the extra tap (column KW-1 of kernel row `wave`); the wave without one runs the variant compiled without it
        if constexpr (LAST) {
#pragma unroll
          for (int a = 0; a < NCOG; ++a)
            acc[KH][a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[s & 1][a]),
                                                                    __builtin_bit_cast(bf16x8_t, bx[s & 1]), acc[KH][a][0], 0, 0, 0);
        }
      }
    } else if constexpr (sizeof(T) == 2) {
      // register double buffering: the fragments of K step it+1 are read while the MFMAs of step it run
      constexpr int NIT = KSPLIT ? NK / 4 : HT * NK;
      auto load_frags = [&](int it, uint4 (&af)[NCOG], uint4 (&bf)[WT][NCIG]) {
        const int r = KSPLIT ? 0 : it / NK;
        const int ks = KSPLIT ? it * 4 + wave : it - r * NK;
        const unsigned char* dbase = smem + frag_off + (r * SW + ks * 16) * 32;
        const unsigned char* abase = smem + frag_off + DY_BYTES + (A_PS - DY_PS) * gpl_c + (r * QW + ks * 16) * 32;
#pragma unroll
        for (int cg = 0; cg < NCOG; ++cg) {
          const unsigned char* base = dbase + cg * PLG * DY_PS;
          const uint2 lo = lds_tr16_b64(base), hi = lds_tr16_b64(base + 4 * 32);
          af[cg] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int i = 0; i < WT; ++i) {
          if (i < KH && dead(i, r)) continue;              // compile time: nobody multiplies these
#pragma unroll
          for (int cg = 0; cg < NCIG; ++cg) {
            const unsigned char* base = abase + cg * PLG * A_PS + tap_off[i];
            const uint2 lo = lds_tr16_b64(base), hi = lds_tr16_b64(base + 4 * 32);
            bf[i][cg] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
        }
      };
      constexpr int PPI = (ST_ITERS + NIT - 1) / NIT;   // pieces of the next strip staged per K step
      uint4 af[2][NCOG], bf[2][WT][NCIG];
      load_frags(0, af[0], bf[0]);
#ifndef WG_ABLATE_NO_STAGE
      const StageCtx sc = stage_begin_checked(nxt_strip, nxt_buf);
#endif
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (it + 1 < NIT) load_frags(it + 1, af[(it + 1) & 1], bf[(it + 1) & 1]);
#ifndef WG_ABLATE_NO_STAGE
#pragma unroll
        for (int k = it * PPI; k < (it + 1) * PPI && k < ST_ITERS; ++k) stage_piece(sc, k);
#endif
#pragma unroll
        for (int i = 0; i < WT; ++i) {
          // the repeated tap of the short wave: its fragments are read (no branch in the read stream) but the MFMAs,
          // which are what the power budget pays for, are skipped
          if (i + 1 == WT && !last_tap_real) continue;
          if (dead(i, KSPLIT ? 0 : it / NK)) continue;
#pragma unroll
          for (int a = 0; a < NCOG; ++a)
#pragma unroll
            for (int c = 0; c < NCIG; ++c)
              acc[i][a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[it & 1][a]),
                                                                     __builtin_bit_cast(bf16x8_t, bf[it & 1][i][c]),
                                                                     acc[i][a][c], 0, 0, 0);
        }
      }
    } else {
      // fp32: one value per lane and operand (v_mfma_f32_32x32x2_f32, K step = 2 pixels): lane l31 holds channel l31 & 7 of
      // plane l31 >> 3 at pixel 2 * ks + half.  The K steps of a strip are fully unrolled and register double buffered --
      // the fragments of step it + 1 are read while the MFMAs of step it run -- and the loop is ONE basic block: the wave
      // with three real taps multiplies its repeated tap into the accumulator that is never stored instead of branching
      // around it.  (The plain loop -- read, wait, multiply, a branch around the odd tap -- left a K step's LDS latency
      // exposed twice: 0.76 of the fp32 matrix peak on every layer; the fp32 pipe is not power limited, the spare MFMA of
      // one wave in four costs no time.)
      constexpr int NIT = KSPLIT ? NK / 4 : HT * NK;
      constexpr int WM = COL4 ? KH : WT;                    // accumulators multiplied in every K step
      const unsigned char* dbase0 = smem + (l31 >> 3) * DY_PS + half * 32 + (l31 & 7) * 4;
      const unsigned char* abase0 = smem + DY_BYTES + (l31 >> 3) * A_PS + half * 32 + (l31 & 7) * 4;
      const unsigned char* abase[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) abase[i] = abase0 + tap_off[i] + (COL4 && i >= KH ? wave * 64 : 0);   // the shared column: my K step of four
      const unsigned char* dbase4 = dbase0 + wave * 64;
      (void)dbase4;
      auto load_frags = [&](int it, float (&af)[NCOG], float (&bf)[WM][NCIG]) {
        const int r = KSPLIT ? 0 : it / NK;
        const int ks = KSPLIT ? it * 4 + wave : it - r * NK;
#pragma unroll
        for (int cg = 0; cg < NCOG; ++cg) af[cg] = *(const float*)(dbase0 + cg * PLG * DY_PS + (r * SW + ks * 2) * 32);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          if (dead(i, r)) continue;                          // compile time: nobody multiplies these
#pragma unroll
          for (int cg = 0; cg < NCIG; ++cg) bf[i][cg] = *(const float*)(abase[i] + cg * PLG * A_PS + (r * QW + ks * 2) * 32);
        }
      };
      // the shared column's fragments for the group of four K steps starting at it4 (a group never straddles a strip row)
      auto load_frags4 = [&](int it4, float (&af)[NCOG], float (&bf)[KH][NCIG]) {
        const int r = it4 / NK, ks = it4 - r * NK;
#pragma unroll
        for (int cg = 0; cg < NCOG; ++cg) af[cg] = *(const float*)(dbase4 + cg * PLG * DY_PS + (r * SW + ks * 2) * 32);
#pragma unroll
        for (int i = 0; i < KH; ++i) {
          if (dead(KH + i, r)) continue;
#pragma unroll
          for (int cg = 0; cg < NCIG; ++cg) bf[i][cg] = *(const float*)(abase[COL4 ? KH + i : 0] + cg * PLG * A_PS + (r * QW + ks * 2) * 32);
        }
      };
      float af[2][NCOG], bf[2][WM][NCIG], af4[NCOG], bf4[KH][NCIG];
      (void)af4; (void)bf4;
      load_frags(0, af[0], bf[0]);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (COL4 && it % 4 == 0) load_frags4(it, af4, bf4);
        if (it + 1 < NIT) load_frags(it + 1, af[(it + 1) & 1], bf[(it + 1) & 1]);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          if (dead(i, KSPLIT ? 0 : it / NK)) continue;
#pragma unroll
          for (int a = 0; a < NCOG; ++a)
#pragma unroll
            for (int c = 0; c < NCIG; ++c)
              acc[i][a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[it & 1][a], bf[it & 1][i][c], acc[i][a][c], 0, 0, 0);
        }
        if (COL4 && it % 4 == 3) {
#pragma unroll
          for (int i = 0; i < KH; ++i) {
            if (dead(KH + i, it / NK)) continue;
#pragma unroll
            for (int a = 0; a < NCOG; ++a)
#pragma unroll
              for (int c = 0; c < NCIG; ++c)
                acc[COL4 ? KH + i : 0][a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af4[a], bf4[i][c], acc[COL4 ? KH + i : 0][a][c], 0, 0, 0);
          }
        }
      }
    }
#endif
  }
  };
  {
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    const bool fixed_height = p.skip_halo == 1 && p.nsplit % p.nth == 0;
    const int th = split % p.nth;
    const bool top = fixed_height && th == 0, bot = fixed_height && th == p.nth - 1;   // block-uniform
    constexpr bool WINDOW = sizeof(T) == 2 && !KSPLIT;      // the only path with a per-wave LAST variant
    auto run = [&](auto t, auto b) {
      if (!WINDOW || last_tap_real) strips(t, b, yes{}); else strips(t, b, no{});
    };
    if constexpr (KH == 1) strips(no{}, no{}, yes{});
    else if (top && bot) run(yes{}, yes{});
    else if (top) run(yes{}, no{});
    else if (bot) run(no{}, yes{});
    else run(no{}, no{});
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave
  if constexpr (COL4) {
    // the shared column: accumulator KH + j of every wave is a partial sum of tap (kernel row j, last column) over that wave's
    // K steps.  All partials go through LDS; wave j < KH adds the four of row j in the fixed order wave 0, 1, 2, 3 and stores
    // the tap below (bitwise reproducible).
    constexpr int PER = NCOG * NCIG * 16 * 64;               // floats of one accumulator set of one wave
    static_assert(4 * KH * PER * 4 <= NBUF * BUF_BYTES, "cross-wave sums fit the staging buffers");
    float* red = (float*)smem_all;
    __syncthreads();                                         // every wave is done with the staged strips, every DMA has landed
#pragma unroll
    for (int j = 0; j < KH; ++j)
#pragma unroll
      for (int a = 0; a < NCOG; ++a)
#pragma unroll
        for (int c = 0; c < NCIG; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((wave * KH + j) * PER) + ((a * NCIG + c) * 16 + r) * 64 + lane] = acc[KH + j][a][c][r];
    __syncthreads();
    if (wave < KH) {
#pragma unroll
      for (int a = 0; a < NCOG; ++a)
#pragma unroll
        for (int c = 0; c < NCIG; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) t += red[((wv * KH + wave) * PER) + ((a * NCIG + c) * 16 + r) * 64 + lane];
            acc[KH][a][c][r] = t;                             // accumulator KH now holds tap (kernel row `wave`, last column), complete
          }
    }
  }
  // ---- write this block's partial tile(s)
  const int slab_idx = KSPLIT ? split * 4 + wave : split;
  float* slab = p.slab + (long)slab_idx * TAPS * p.co_pad * p.ci_pad;
#pragma unroll
  for (int i = 0; i < (COL4 ? KH + 1 : WT); ++i) {
    // COL4: after the cross-wave sum accumulator KH is the last column's tap of kernel row `wave` (waves 0..KH-1)
    const int tap = COL4 && i == KH ? min(wave, KH - 1) * KW + KW - 1 : tap_of(i);
    if (COL4 ? (i < KH || wave < KH) : (i + 1 < WT || last_tap_real)) {
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

template <typename T, int KH, int KW, int HT = 1, int COT = WgTraits<T>::CO_T>
int conv_wgrad_launch(WgradParams p, long slab_bytes, int* nslab_out, hipStream_t stream) {
  using TR = WgTraits<T>;
  (void)sizeof(TR);
  constexpr int TAPS = KH * KW;
  constexpr bool KSPLIT = TAPS < 4;
  constexpr int SW = wg_strip_width<T>(HT);
  p.ntw = (p.W + SW - 1) / SW;
  p.nth = (p.H + HT - 1) / HT;
  if ((long)p.dy_pt * p.H * p.W * p.dy_wmul * 32 >= (1L << 31) || (long)p.a_pt * p.H * p.W * 32 >= (1L << 31))
    return dn_set_error(DN_ERR_ARG, "conv_wgrad: one image of an operand exceeds 2 GiB (32-bit buffer offsets)");
  p.nstrips = p.N * p.nth * p.ntw;
  p.co_pad = (p.co_pad + COT - 1) / COT * COT;
  constexpr int CI_T = wg_ci_tile(TAPS, sizeof(T));
  p.ci_pad = (p.ci_pad + CI_T - 1) / CI_T * CI_T;
  const long tiles = (long)(p.co_pad / COT) * (p.ci_pad / CI_T);
  const long per_slab = (long)TAPS * p.co_pad * p.ci_pad * 4 * (KSPLIT ? 4 : 1);
  const long target = dn_tune().wgrad_blocks > 0 ? dn_tune().wgrad_blocks : 512;   // two resident workgroups per CU
  long nsplit = std::max<long>(1, (target + tiles - 1) / tiles);
  // >= 2048 pixels per block: every slab costs a round trip through HBM.  Layers with few tiles get their blocks from
  // the split alone, and at train.py's batch of 16 that floor left CUs without work (64 -> 64 channels at 2048 columns:
  // 320 blocks); their slabs are small, so they may split further (measured at 16 profiles: fp32 234 -> 164 us on that
  // layer, 143 -> 122 us on 32 -> 32; bf16 49 -> 41 us on 64 -> 128).  At 256 profiles neither floor binds.
  const long small_px = sizeof(T) == 4 ? (tiles <= 8 ? 640 : 2048) : (tiles >= 2 && tiles <= 16 ? 1024 : 2048);
  const long min_px = dn_tune().wgrad_minpx > 0 ? dn_tune().wgrad_minpx : small_px;
  nsplit = std::min<long>(nsplit, std::max<long>(1, (long)p.nstrips * HT * SW / min_px));
  nsplit = std::min<long>(nsplit, slab_bytes / per_slab);
  // splits tied to XCDs need a multiple of 8 of them (else some XCDs would idle): round up when the workspace allows
  bool xcd = !dn_tune().wgrad_no_xcd;
  if (xcd && nsplit % 8) {
    const long up = (nsplit + 7) / 8 * 8;
    if (up <= slab_bytes / per_slab && up <= std::max<long>(1, (long)p.nstrips * HT * SW / (min_px / 2))) nsplit = up; else xcd = false;
  }
  if (nsplit >= p.nth) nsplit -= nsplit % p.nth;   // blocks of one height in the image (halo-row skipping)
  if (nsplit % 8) xcd = false;
  if (nsplit < 1) return dn_set_error(DN_ERR_ARG, "conv_wgrad: workspace of %ld bytes is smaller than one partial slab (%ld)",
                                      slab_bytes, per_slab);
  p.skip_halo = dn_tune().wgrad_noskip ? 0 : 1;
  p.nsplit = (int)nsplit;
  p.xcd_split = xcd ? 1 : 0;
  *nslab_out = (int)nsplit * (KSPLIT ? 4 : 1);
  const long grid = p.xcd_split ? tiles * ((nsplit + 7) / 8 * 8) : tiles * nsplit;
  hipLaunchKernelGGL((conv_wgrad_kernel<T, KH, KW, HT, COT>), dim3((unsigned)grid), dim3(256), 0, stream, p);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

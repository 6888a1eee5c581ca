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
#include <type_traits>
#include "common.h"

struct ConvParams {
  const void* x; int x_planes_total, x_plane0, x_nplanes, x_gplanes, x_wmul;
  // virtual K plane ck -> group ck / x_gplanes, plane ck % x_gplanes.  A group is shifted by x_gstride records (the
  // transposed convolution's data gradient: group k = output column 2w + k) and odd groups by x_lo_off planes (hi + lo
  // split activations, K walk [hi | lo | hi] against the weight image [W_hi | W_hi | W_lo]; 0 otherwise)
  int x_gstride, x_lo_off;
  const void* w;            // packed [nchunk][TAPS][cout_pad][32 B]
  const float* bias;        // [cout_pad] (virtual channel order) or nullptr
  void* y; int y_planes_total, y_plane0, y_nplanes, y_gplanes, y_wmul;
  int y_lo_off, pool_lo_off; // split epilogues: planes between a tensor's hi part and its lo part (pool_lo_off 0: pooled output is plain bf16)
  int head_exact;            // head epilogue: do not round the activation to the storage type first (split mode)
  int N, H, W;              // pixel grid of the GEMM (per group)
  int nchunk;               // K chunks = virtual input planes
  int cout_pad;             // multiple of COT
  int relu;
  int ntw, nth;             // column / row tiles per image
  float* stats;             // BatchNorm partial sums [pixel tile][cout_pad][2] (CONV_EPI_STATS / CONV_EPI_BNRED), else unused
  // CONV_EPI_BNRED (a data gradient whose output is dL/d(activation) of a BatchNorm + ReLU unit): that unit's saved
  // convolution output (same channels and pixel grid as y) and its per-channel vectors [cout_pad]
  const void* bn_y; int bn_planes_total, bn_plane0;
  const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
  void* pool; int pool_planes_total, pool_plane0;   // MaxPool2d((1,2)) of y, W/2 columns (CONV_EPI_POOL), else unused
  const float* head_w; const float* head_b;         // 1x1 head over the tile's channels (CONV_EPI_HEAD): y is not stored,
  float* head_logits; float* head_probs;            // logits / sigmoid go to [N][H][W] fp32 (either may be null)
#ifdef C8_PROFILE_WAIT
  long long* dbg;           // [waves][3]: cycles in vmcnt waits, in barriers, total (scratch/conv8_micro.hip)
#endif
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

// What the epilogue does besides storing y
constexpr int CONV_EPI_PLAIN = 0;   // nothing
constexpr int CONV_EPI_STATS = 1;   // + BatchNorm batch statistics of y (training forward)
constexpr int CONV_EPI_POOL = 2;    // + MaxPool2d((1,2)) of y into a second buffer (encoder blocks, eval mode)
constexpr int CONV_EPI_HEAD = 3;    // instead of storing y: the 1x1 output convolution + sigmoid over it (last block, eval mode)
constexpr int CONV_EPI_SPLIT = 4;   // bf16 only: y = hi + lo, two bf16 tensors p.y_lo_off planes apart (~16 mantissa bits kept)
constexpr int CONV_EPI_POOL_SPLIT = 5;   // SPLIT + POOL (the pooled tensor split too when p.pool_lo_off > 0)
constexpr int CONV_EPI_BNRED = 6;   // + the two sums of BatchNorm's backward over the stored y (= g of the unit that produced this
                                    //   convolution's input): sum dz and sum dz * xhat, dz = g * (bn_y * scale + shift > 0)

// Head form of the epilogue (unet_architecture.py:94-97 + trainable_unet.py:409 fused into the last DoubleConv): the
// block holds every channel of its pixels (cout_pad == COT), a lane 16 of them, so logit = sum_c w_c * act_c + b is a
// 16-term dot product per lane plus one exchange between the wave halves.  The activation is rounded to the storage
// type first, i.e. the head sees what it would have read back from HBM; 64 B/pixel of writes become 4 (or 8).
template <typename T, int HT, int COT>
__device__ __forceinline__ void conv_epilogue_head(const ConvParams& p, f32x16_t (&acc)[HT][COT / 32], int n, int h0, int wbase,
                                                   int lane) {
  constexpr int NCG = COT / 32;
  const int l31 = lane & 31, half = lane >> 5;
  float bias_r[NCG][16], hw[NCG][16];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = cg * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
      bias_r[cg][r] = p.bias ? p.bias[ch] : 0.f;
      hw[cg][r] = p.head_w[ch];
    }
  const float hb = p.head_b[0];
#pragma unroll
  for (int hh = 0; hh < HT; ++hh) {
    const int h = h0 + hh;
    if (h < p.H) {                                    // block-uniform
      float dot = 0.f;
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[hh][cg][r] + bias_r[cg][r];
          if (p.relu) v = fmaxf(v, 0.f);
          if constexpr (sizeof(T) == 2) { if (!p.head_exact) v = (float)(__bf16)v; }
          dot = fmaf(v, hw[cg][r], dot);
        }
      dot += __shfl_xor(dot, 32);
      if (half == 0 && wbase + l31 < p.W) {
        const long idx = ((long)n * p.H + h) * p.W + wbase + l31;
        const float z = dot + hb;
        if (p.head_logits) p.head_logits[idx] = z;
        if (p.head_probs) p.head_probs[idx] = 1.f / (1.f + expf(-z));
      }
    }
  }
}

// Epilogue shared by the 4-wave and 8-wave kernels: bias (+ folded BN) + ReLU in registers, then through a wave-private
// LDS scratch so that every global store instruction writes 1 KiB of consecutive 16-byte pieces (32 pixels of one
// plane row) instead of 8-byte fragments at a 32-byte stride.  A lane's bias values are fetched once, before the rows.
// Plain form: one pass over the rows, all COT channels of a row per scratch round trip (COT/EPR KiB per wave).
// STATS form: one pass per 32-channel group, so that a lane holds only 16 running (sum, sum of squares) pairs of the
// values AS STORED (bf16-rounded) -- BatchNorm's batch statistics; they end up in stat_wave[channel of the tile][2]
// (LDS, this wave's).
// POOL form: the plain form, plus max over each even/odd column pair (neighbouring lanes; rounding is monotonic, so the
// max of the fp32 values rounds to the max of the stored ones) through a second scratch (COT/EPR x 512 B per wave) into
// p.pool -- unet_architecture.py:37 without a second pass over the skip tensor.
// PAIRS (1x1 GEMMs only): the transposed convolution's output, y_wmul == 2 -- GEMM plane 2j + k is output column 2w + k of
// output plane j (DN_PACK_CONVT_FWD row order), so the two planes of a pair are stored together: 16 pixels x 64 B = 1 KiB
// of consecutive bytes per store instruction instead of 32-byte pieces at a 64-byte stride.
template <typename T, int HT, int COT, int EPI, bool PAIRS = false>
__device__ __forceinline__ void conv_epilogue_store(const ConvParams& p, f32x16_t (&acc)[HT][COT / 32], unsigned char* scratch,
                                                    float* stat_wave, unsigned char* pool_scratch, int n, int h0, int wbase,
                                                    int cot, int lane) {
  constexpr bool STATS = EPI == CONV_EPI_STATS, POOL = EPI == CONV_EPI_POOL || EPI == CONV_EPI_POOL_SPLIT;
  constexpr bool SPLIT = EPI == CONV_EPI_SPLIT || EPI == CONV_EPI_POOL_SPLIT;
  constexpr bool BNRED = EPI == CONV_EPI_BNRED;
  static_assert(!SPLIT || sizeof(T) == 2, "hi + lo split exists for bf16 storage only");
  constexpr int NPART = SPLIT ? 2 : 1;          // part 0 stores bf16(v), part 1 bf16(v - bf16(v)) y_lo_off planes further
  constexpr int NCG = COT / 32, EPR = RecTraits<T>::EPR, GPL = 32 / EPR;
  // where the statistics are accumulated: inside the store loop (16 + 16 live sums per lane, hence one pass per channel
  // group), or -- for the 3-row x 64-channel tile, where those 40 registers cost the third workgroup per CU (+160..240 us
  // per layer) -- in a second walk over the accumulators, four channels at a time, after the plain store loop
  constexpr bool STATS_WALK2 = STATS && HT == 3 && COT == 64;
  constexpr bool STATS_INLOOP = STATS && !STATS_WALK2;
  constexpr int NPASS = STATS_INLOOP ? NCG : 1, CPP = NCG / NPASS;  // passes over the rows, channel groups per pass
  const int l31 = lane & 31, half = lane >> 5;
  unsigned char* yb = (unsigned char*)p.y;
  const int yrow = p.W * p.y_wmul;
  const bool col_ok = wbase + l31 < p.W;
  // ReLU as ONE v_max per value against a per-launch floor (0 or -inf) instead of a max and a select on p.relu
  const float floor_v = p.relu ? 0.f : -__builtin_inff();
  // BNRED (a data gradient: no bias, no ReLU -- the dispatcher checks): the consumer unit's per-channel vectors
  // [scale | shift | mean | invstd][COT] go to LDS once per block, in the pooled-output scratch this form does not use
  float* bn_k = nullptr;
  (void)bn_k;
  if constexpr (BNRED) {
    bn_k = (float*)(pool_scratch - (threadIdx.x >> 6) * ((COT / EPR) * 512));
    if (threadIdx.x < COT) {
      const int ch = cot * COT + threadIdx.x;
      bn_k[threadIdx.x] = p.bn_scale[ch]; bn_k[COT + threadIdx.x] = p.bn_shift[ch];
      bn_k[2 * COT + threadIdx.x] = p.bn_mean[ch]; bn_k[3 * COT + threadIdx.x] = p.bn_invstd[ch];
    }
  }
  // ... and the unit's saved convolution output at this lane's pixels, four channels (one lane group g of a 32-channel
  // accumulator tile) per piece and row: the first BN_PF groups are requested here, before the stores, so that they have
  // arrived when the sums are formed; group k + BN_PF is requested when group k is done
  using bn_piece_t = std::conditional_t<sizeof(T) == 2, uint2, uint4>;
  constexpr int BN_NG = NCG * 4, BN_REGS = HT * (int)(sizeof(bn_piece_t) / 4);
  constexpr int BN_BUDGET = 36;   // registers for pieces in flight: what the 3-row x 64-channel tile has left at three workgroups per CU
  constexpr int BN_PF = !BNRED ? 1 : (BN_BUDGET / BN_REGS >= BN_NG ? BN_NG : (BN_BUDGET / BN_REGS >= 2 ? BN_BUDGET / BN_REGS : 2));
  bn_piece_t bn_yv[BN_PF][HT];
  (void)bn_yv;
  auto bn_fetch = [&](int gi) {
    const int ch = cot * COT + (gi >> 2) * 32 + 8 * (gi & 3) + 4 * half;
    const int plane = ch / EPR;                                  // wave-uniform: 4 * half stays inside a record
#pragma unroll
    for (int hh = 0; hh < HT; ++hh) {
      bn_yv[gi % BN_PF][hh] = bn_piece_t{};
      if (plane < p.y_nplanes && h0 + hh < p.H && col_ok)
        bn_yv[gi % BN_PF][hh] = *(const bn_piece_t*)((const unsigned char*)p.bn_y +
            ((((long)n * p.bn_planes_total + p.bn_plane0 + plane) * p.H + h0 + hh) * p.W + wbase + l31) * 32 + (ch % EPR) * sizeof(T));
    }
  };
  if constexpr (BNRED) {
#pragma unroll
    for (int gi = 0; gi < BN_PF; ++gi) bn_fetch(gi);
  }
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    float ssum[STATS_INLOOP ? 16 : 1] = {}, ssq[STATS_INLOOP ? 16 : 1] = {};
    (void)ssum; (void)ssq;
    float bias_r[CPP][BNRED ? 1 : 16];
#pragma unroll
    for (int c = 0; c < CPP; ++c)
#pragma unroll
      for (int r = 0; r < (BNRED ? 1 : 16); ++r)
        bias_r[c][r] = p.bias ? p.bias[cot * COT + (ps * CPP + c) * 32 + 8 * (r >> 2) + 4 * half + (r & 3)] : 0.f;
#pragma unroll
    for (int hh = 0; hh < HT; ++hh) {
      const int h = h0 + hh;
      if (h < p.H) {                                  // block-uniform
#pragma unroll
       for (int part = 0; part < NPART; ++part) {
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cl = c * 32 + 8 * g + 4 * half;   // channel inside the pass
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if constexpr (BNRED) v[i] = acc[hh][c][4 * g + i];
              else v[i] = fmaxf(acc[hh][ps * CPP + c][4 * g + i] + bias_r[c][4 * g + i], floor_v);
            }
            unsigned char* dst = scratch + (cl / EPR) * 1024 + l31 * 32 + (cl % EPR) * sizeof(T);
            if constexpr (sizeof(T) == 4) {
              *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            } else {
              __bf16 bq[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
              if (SPLIT && part == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bq[i] = (__bf16)(v[i] - (float)bq[i]);
              }
              uint2 u;
              u.x = (uint32_t)__builtin_bit_cast(unsigned short, bq[0]) | ((uint32_t)__builtin_bit_cast(unsigned short, bq[1]) << 16);
              u.y = (uint32_t)__builtin_bit_cast(unsigned short, bq[2]) | ((uint32_t)__builtin_bit_cast(unsigned short, bq[3]) << 16);
              *(uint2*)dst = u;
              if constexpr (STATS_INLOOP) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (float)bq[i];     // statistics of the values as stored
              }
            }
            if constexpr (STATS_INLOOP) {
              if (col_ok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { ssum[4 * g + i] += v[i]; ssq[4 * g + i] = fmaf(v[i], v[i], ssq[4 * g + i]); }
              }
            }
            if constexpr (POOL) {
              float m[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) m[i] = fmaxf(v[i], __shfl_xor(v[i], 1));
              if ((l31 & 1) == 0) {
                unsigned char* pd = pool_scratch + (cl / EPR) * 512 + (l31 >> 1) * 32 + (cl % EPR) * sizeof(T);
                if constexpr (sizeof(T) == 4) {
                  *(float4*)pd = make_float4(m[0], m[1], m[2], m[3]);
                } else {
                  __bf16 bq[4] = {(__bf16)m[0], (__bf16)m[1], (__bf16)m[2], (__bf16)m[3]};
                  if (SPLIT && part == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) bq[i] = (__bf16)(m[i] - (float)bq[i]);
                  }
                  uint2 u;
                  u.x = (uint32_t)__builtin_bit_cast(unsigned short, bq[0]) | ((uint32_t)__builtin_bit_cast(unsigned short, bq[1]) << 16);
                  u.y = (uint32_t)__builtin_bit_cast(unsigned short, bq[2]) | ((uint32_t)__builtin_bit_cast(unsigned short, bq[3]) << 16);
                  *(uint2*)pd = u;
                }
              }
            }
          }
        }
        if (PAIRS && p.y_wmul == 2) {                   // block-uniform
          static_assert(!PAIRS || (CPP * GPL) % 2 == 0, "a tile holds whole plane pairs");
#pragma unroll
          for (int pp = 0; pp < CPP * GPL / 2; ++pp) {
            const int yplane = ((cot * NCG + ps * CPP) * GPL) / 2 + pp;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const int px = hf * 16 + (lane >> 2), k = (lane >> 1) & 1, pc = lane & 1;
              const uint4 val = *(const uint4*)(scratch + (2 * pp + k) * 1024 + px * 32 + pc * 16);
              if (yplane < p.y_nplanes && wbase + px < p.W) {
                const long rec = ((long)(n * p.y_planes_total + p.y_plane0 + yplane + part * p.y_lo_off) * p.H + h) * yrow + (long)(wbase + px) * 2 + k;
                *(uint4*)(yb + rec * 32 + 16 * pc) = val;
              }
            }
          }
        } else
#pragma unroll
        for (int pl = 0; pl < CPP * GPL; ++pl) {
          const int vplane = (cot * NCG + ps * CPP) * GPL + pl;
          const int ygrp = vplane / p.y_gplanes;
          const int yplane = vplane - ygrp * p.y_gplanes;
          const int px = lane >> 1, pc = lane & 1;
          const uint4 val = *(const uint4*)(scratch + pl * 1024 + lane * 16);
#ifdef CV_ABLATE_NO_STORE
          if (val.x == 0x12345678u)
#endif
          if (yplane < p.y_nplanes && wbase + px < p.W) {
            const long rec = ((long)(n * p.y_planes_total + p.y_plane0 + yplane + part * p.y_lo_off) * p.H + h) * yrow +
                             (long)(wbase + px) * p.y_wmul + ygrp;
            *(uint4*)(yb + rec * 32 + 16 * pc) = val;
          }
        }
        if constexpr (POOL) {
          // 16 pooled pixels x 32 B per plane: the lower half-wave stores 512 consecutive bytes per plane
          const int wp = p.W >> 1, px = (lane & 31) >> 1, pc = lane & 1;
#pragma unroll
          for (int pl = 0; pl < CPP * GPL; ++pl) {
            const int yplane = (cot * NCG + ps * CPP) * GPL + pl;
            const uint4 val = *(const uint4*)(pool_scratch + pl * 512 + (lane & 31) * 16);
            if (lane < 32 && yplane < p.y_nplanes && (wbase >> 1) + px < wp && (part == 0 || p.pool_lo_off > 0)) {
              const long rec = ((long)(n * p.pool_planes_total + p.pool_plane0 + yplane + part * p.pool_lo_off) * p.H + h) * wp + (wbase >> 1) + px;
              *(uint4*)((unsigned char*)p.pool + rec * 32 + 16 * pc) = val;
            }
          }
        }
       }
      }
    }
    if constexpr (STATS || BNRED) {
      // Sum over the 32 pixel lanes of each half-wave, four channels (a lane's accumulator group g) at a time: the eight
      // values [s0..s3, q0..q3] go through a butterfly that halves the number of live values per step -- lane bit 0
      // decides who keeps the sums and who the squares, bit 1 who keeps channels 0,1 and who 2,3 (DPP quad permutes) --
      // then the two survivors are summed over the row (row_ror 4 / 8) and over the two rows of the half.  26 VALU
      // instructions per group instead of 48 (one full 5-step reduction per value); lanes 0..3 of each half publish.
      auto dpp_add = [](float keep, float give, auto ctrl) {
        return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), decltype(ctrl)::value, 0xF, 0xF, true));
      };
      using QP1 = std::integral_constant<int, 0xB1>;     // quad_perm [1,0,3,2]
      using QP2 = std::integral_constant<int, 0x4E>;     // quad_perm [2,3,0,1]
      using ROR4 = std::integral_constant<int, 0x124>;
      using ROR8 = std::integral_constant<int, 0x128>;
      const bool b0 = lane & 1, b1 = lane & 2;
      auto publish4 = [&](int cgi, int g, const float (&s4)[4], const float (&q4)[4]) {
        float A[4], B[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) A[j] = dpp_add(b0 ? q4[j] : s4[j], b0 ? s4[j] : q4[j], QP1{});
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          B[j] = dpp_add(b1 ? A[2 + j] : A[j], b1 ? A[j] : A[2 + j], QP2{});
          B[j] = dpp_add(B[j], B[j], ROR4{});
          B[j] = dpp_add(B[j], B[j], ROR8{});
          B[j] += __shfl_xor(B[j], 16);
        }
        if (l31 < 4) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int ch = cgi * 32 + 8 * g + 4 * half + (b1 ? 2 : 0) + j;
            stat_wave[ch * 2 + (b0 ? 1 : 0)] = B[j];
          }
        }
      };
      if constexpr (BNRED) {
        // BatchNorm backward's two sums for the unit whose activation gradient this tile just stored (train_ops.hip,
        // bn_bwd_reduce_kernel), on the values as stored.  A further walk over the accumulators, four channels (one 8- or
        // 16-byte piece of the unit's saved convolution output per row) at a time.  A lane sums dz and dz * y over its
        // rows and turns the second into sum dz * xhat = (sum dz*y - mean * sum dz) * invstd before the lanes are added
        // up: two per-channel vectors less in the inner loop, and at most 6 terms under the subtraction.
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        __syncthreads();                                               // bn_k is filled (all waves are past their stores)
#pragma unroll
        for (int gi = 0; gi < BN_NG; ++gi) {
          const int c = gi >> 2, g = gi & 3, cl = c * 32 + 8 * g + 4 * half;
          float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
          if ((cot * COT + cl) / EPR < p.y_nplanes) {   // wave-uniform
            const float4 k0 = *(const float4*)(bn_k + cl), k1 = *(const float4*)(bn_k + COT + cl);
            const float4 k2 = *(const float4*)(bn_k + 2 * COT + cl), k3 = *(const float4*)(bn_k + 3 * COT + cl);
            const float sc[4] = {k0.x, k0.y, k0.z, k0.w}, sh[4] = {k1.x, k1.y, k1.z, k1.w};
            const float mu[4] = {k2.x, k2.y, k2.z, k2.w}, is[4] = {k3.x, k3.y, k3.z, k3.w};
            const bn_piece_t (&yv)[HT] = bn_yv[gi % BN_PF];
#pragma unroll
            for (int hh = 0; hh < HT; ++hh) {
              if (h0 + hh < p.H) {                      // block-uniform
                float fy[4];
                if constexpr (sizeof(T) == 2) {
                  fy[0] = __uint_as_float(yv[hh].x << 16); fy[1] = __uint_as_float(yv[hh].x & 0xffff0000u);
                  fy[2] = __uint_as_float(yv[hh].y << 16); fy[3] = __uint_as_float(yv[hh].y & 0xffff0000u);
                } else {
                  fy[0] = __uint_as_float(yv[hh].x); fy[1] = __uint_as_float(yv[hh].y);
                  fy[2] = __uint_as_float(yv[hh].z); fy[3] = __uint_as_float(yv[hh].w);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                  f32x2_t v = {acc[hh][c][4 * g + 2 * k], acc[hh][c][4 * g + 2 * k + 1]};
                  if constexpr (sizeof(T) == 2) {       // the gradient as stored
                    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                    v = (f32x2_t){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
                  }
#pragma unroll
                  for (int i = 0; i < 2; ++i) {
                    const int j = 2 * k + i;
                    const float dz = (col_ok && fmaf(fy[j], sc[j], sh[j]) > 0.f) ? v[i] : 0.f;   // the forward's own pre-ReLU value
                    s4[j] += dz;
                    q4[j] = fmaf(dz, fy[j], q4[j]);
                  }
                }
              }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) q4[j] = (q4[j] - mu[j] * s4[j]) * is[j];
          }
          if (gi + BN_PF < BN_NG) bn_fetch(gi + BN_PF);
          publish4(c, g, s4, q4);
        }
      } else if constexpr (STATS_INLOOP) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float s4[4] = {ssum[4 * g], ssum[4 * g + 1], ssum[4 * g + 2], ssum[4 * g + 3]};
          const float q4[4] = {ssq[4 * g], ssq[4 * g + 1], ssq[4 * g + 2], ssq[4 * g + 3]};
          publish4(ps, g, s4, q4);
        }
      } else {
        // second walk over the accumulators, two values per instruction where the ISA has packed fp32 math
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        const bool all_cols = wbase + 31 < p.W;     // wave-uniform: no column of this wave hangs over the image edge
        auto walk = [&](auto masked_c) {
          constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
          for (int c = 0; c < CPP; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x2_t s2[2] = {{0.f, 0.f}, {0.f, 0.f}}, q2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
              for (int hh = 0; hh < HT; ++hh) {
                if (h0 + hh < p.H) {                      // block-uniform
#pragma unroll
                  for (int k = 0; k < 2; ++k) {
                    const int r = 4 * g + 2 * k;
                    f32x2_t v = {acc[hh][ps * CPP + c][r], acc[hh][ps * CPP + c][r + 1]};
                    v += (f32x2_t){bias_r[c][r], bias_r[c][r + 1]};
                    v = (f32x2_t){fmaxf(v.x, floor_v), fmaxf(v.y, floor_v)};
                    if constexpr (sizeof(T) == 2) {       // statistics of the values as stored
                      const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                      v = (f32x2_t){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
                    }
                    if constexpr (MASKED) v = col_ok ? v : (f32x2_t){0.f, 0.f};
                    s2[k] += v;
                    q2[k] = __builtin_elementwise_fma(v, v, q2[k]);
                  }
                }
              }
              const float s4[4] = {s2[0].x, s2[0].y, s2[1].x, s2[1].y}, q4[4] = {q2[0].x, q2[0].y, q2[1].x, q2[1].y};
              publish4(ps * CPP + c, g, s4, q4);
            }
        };
        if (all_cols) walk(std::false_type{}); else walk(std::true_type{});
      }
    }
  }
}

// A tile whose halo row lies above / below the image (every tile of the shipped 5-dye case, half the tiles of the
// 6-dye one) leaves out the MFMAs of the (kernel row, output row) pairs that only see that zero row: 2 of 15 for 3x5
// with both.  The choice is per block, so the chunk loop exists in four compile-time variants and each block runs
// one; inside a variant the MFMA loop has no branches, so the fragment reads and MFMAs of a whole chunk are
// scheduled as one basic block.
// KP = K planes (chunks) staged and consumed per barrier: a 1x1 GEMM has only HT * COT/32 MFMAs per wave and plane, too
// few to pay for a barrier each.
template <typename T, int KH, int KW, int HT, int COT, int EPI = CONV_EPI_PLAIN, int MINB = 2, int KP = 1>
__global__ __launch_bounds__(256, MINB) void conv_igemm_kernel(const ConvParams p) {
  constexpr bool STATS = EPI == CONV_EPI_STATS;
  constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
  constexpr int ROWS = HT + KH - 1;
  constexpr int QW = 128 + KW - 1;
  constexpr int NCG = COT / 32;
  constexpr int EPR = RecTraits<T>::EPR;
  constexpr int ACT_RAW = ROWS * QW * 32;
  constexpr int ACT_BYTES = (ACT_RAW + 1023) / 1024 * 1024;          // weight image starts on a 1-KiB piece boundary
  constexpr int W_BYTES = TAPS * COT * 32;
  constexpr int PL_ITERS = (ACT_BYTES + W_BYTES + 4095) / 4096;      // wave-instructions of 1 KiB per wave and plane
  constexpr int PL_BYTES = PL_ITERS * 4096;                          // LDS image of one plane
  constexpr int ST_ITERS = KP * PL_ITERS;
  constexpr int EPI_BYTES = 4 * (COT / EPR) * 1536 + 4 * COT * 2 * 4;  // epilogue: 4 waves x COT channels (+ pooled) + statistics
  constexpr int BUF_BYTES = KP * PL_BYTES;
  // K chunks are software pipelined through NBUF LDS images when two workgroups of them fit a CU (1x1 / 1x3 kernels:
  // a chunk is only HT*NCG MFMAs per wave, far less than a load latency); the 3x5 images are too big for that and
  // rely on the CU's other workgroup to cover their load phase.
  // MINB = 1 (one workgroup per CU, 512 registers per lane: the 6-row x 64-channel tile) has nobody else to cover its
  // loads, and room for two images of its own.
  constexpr int NBUF = MINB * 3 * BUF_BYTES <= 160 * 1024 ? 3 : (MINB * 2 * BUF_BYTES <= 160 * 1024 ? 2 : 1);
  static_assert((NBUF - 1) * ST_ITERS < 64, "vmcnt is a 6-bit counter");
  constexpr int SMEM_BYTES = NBUF * BUF_BYTES > EPI_BYTES ? NBUF * BUF_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem_all[SMEM_BYTES];
  const unsigned smem_lds = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem_all);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // block -> tile map, XCD aware (blocks b and b+8 share an XCD and its L2): inside an XCD the output-channel tile is
  // the fast index, so the blocks that need the same activation tile run at the same time on the same L2 and the
  // tile comes from HBM once instead of once per channel tile; pixel tiles are dealt round-robin to the 8 XCDs.
  const int npt = p.N * p.nth * p.ntw;
  const int ncot = p.cout_pad / COT;
  const int j = blockIdx.x >> 3;
  const int cot = j % ncot;
  int pt = (j / ncot) * 8 + (blockIdx.x & 7);
  if (pt >= npt) return;                 // whole block: the grid is padded to a multiple of 8 pixel tiles
  const int ptile = pt;
  (void)ptile;
  const int tw = pt % p.ntw; pt /= p.ntw;
  const int th = pt % p.nth;
  const int n = pt / p.nth;
  const int w0 = tw * 128, h0 = th * HT;
  const int xrow = p.W * p.x_wmul;

  // per-thread staging plan: LDS piece P = (it*4 + wave)*64 + lane of the linear image [act tile | pad | weight tile];
  // st_off[it] = byte offset of its source inside the current plane / weight block, DN_OOB = zero fill.  The tile
  // position is fixed for the block, so the plan is made once and a chunk costs SALU work only (two SRD bases).
  unsigned st_off[PL_ITERS];
  {
    const int xrow_b = p.W * p.x_wmul * 32;
#pragma unroll
    for (int it = 0; it < PL_ITERS; ++it) {
      const int byte = ((it * 4 + wave) * 64 + lane) * 16;
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

  f32x16_t acc[HT][NCG];
#pragma unroll
  for (int a = 0; a < HT; ++a)
#pragma unroll
    for (int b = 0; b < NCG; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const unsigned char* xb = (const unsigned char*)p.x;
  const unsigned char* wb = (const unsigned char*)p.w;

  // stage chunk ck into LDS image buf (LDS-DMA: no VGPR round trip, all pieces of the chunk in flight at once).  The
  // LDS image is lane-linear per wave-instruction, so the XOR swizzle is applied to the per-lane SOURCE offset
  // (cdna_hip_programming.md, rule 21).  A chunk that does not exist (past the end, or the padding plane of a virtual
  // group, whose weights are zero) is staged from an empty buffer = all zeros, which keeps loops and vmcnt uniform.
  auto stage = [&](int ckp, int buf) {
#pragma unroll
   for (int kp = 0; kp < KP; ++kp) {
    const int ck = ckp * KP + kp;
    const int grp = ck / p.x_gplanes;
    const int plane = ck - grp * p.x_gplanes;
    const bool exists = ck < p.nchunk && plane < p.x_nplanes;
    const long plane_base = ((long)(n * p.x_planes_total + p.x_plane0 + plane + ((grp & 1) ? p.x_lo_off : 0)) * p.H) * xrow;
    dn_srd_t srd_x = dn_make_srd(xb + (exists ? plane_base * 32 + (long)grp * p.x_gstride * 32 : 0));
    dn_srd_t srd_w = dn_make_srd(wb + (exists ? ((long)ck * TAPS * p.cout_pad + (long)cot * COT) * 32 : 0));
    if (!exists) { srd_x.z = 0; srd_w.z = 0; }
#ifndef CV_ABLATE_NO_STAGE   // (ablation hooks: scratch/conv_micro.hip)
#pragma unroll
    for (int it = 0; it < PL_ITERS; ++it) {
      const bool is_w = (it * 4 + wave) * 1024 >= ACT_BYTES;          // wave-uniform
      dn_buffer_lds16(st_off[it], is_w ? srd_w : srd_x,
                      __builtin_amdgcn_readfirstlane(smem_lds + buf * BUF_BYTES + kp * PL_BYTES + (it * 4 + wave) * 1024));
    }
#endif
   }
  };

#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) stage(i, i);
#ifdef C8_PROFILE_WAIT
  long long prof_wait = 0;
  const long long prof_t0 = clock64(), prof_w0 = wall_clock64();
#endif
  auto mainloop = [&](auto skip_top_c, auto skip_bot_c) {
  constexpr bool SKIP_TOP = decltype(skip_top_c)::value, SKIP_BOT = decltype(skip_bot_c)::value;
  int buf = 0;
  const int nsteps = (p.nchunk + KP - 1) / KP;
  for (int ck = 0; ck < nsteps; ++ck, buf = buf + 1 == NBUF ? 0 : buf + 1) {
    if constexpr (NBUF == 1) {
#ifdef C8_PROFILE_WAIT
      const long long tp0 = clock64();
#endif
      __syncthreads();   // previous chunk's fragment reads are done
      stage(ck, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#ifdef C8_PROFILE_WAIT
      prof_wait += clock64() - tp0;
#endif
    } else {
      // LDS-DMA completes in issue order: all but the newest NBUF-2 chunks in flight have landed
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * ST_ITERS) : "memory");
      __syncthreads();   // everyone's pieces of chunk ck landed; everyone left the image of chunk ck-1
      stage(ck + NBUF - 1, buf == 0 ? NBUF - 1 : buf - 1);
    }
#pragma unroll
   for (int kp = 0; kp < KP; ++kp) {
    const unsigned char* smem = smem_all + buf * BUF_BYTES + kp * PL_BYTES;
    // ---- MFMA over all taps
#ifndef CV_ABLATE_NO_COMPUTE
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
          const int hin = hh + dhi - PH;                      // input row relative to the tile
          if (!(SKIP_TOP && hin < 0) && !(SKIP_BOT && hin >= HT)) {   // compile time
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) Mma32<T>::run(af[cg], bf[hh + dhi], acc[hh][cg]);
          }
        }
      }
    }
#else
    (void)smem;
#endif
   }
  }

  };
  {
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    const bool top = KH > 1 && h0 == 0, bot = KH > 1 && h0 + HT >= p.H;   // block-uniform
    if constexpr (KH == 1) mainloop(no{}, no{});
    else if (top && bot) mainloop(yes{}, yes{});
    else if (top) mainloop(yes{}, no{});
    else if (bot) mainloop(no{}, yes{});
    else mainloop(no{}, no{});
  }

  // ---- epilogue: bias (+ folded BN) + ReLU in registers, then through a wave-private LDS scratch so that
  // every global store instruction writes 1 KiB of consecutive 16-byte pieces (32 pixels of one plane row)
  // instead of 8-byte fragments at a 32-byte stride.
#ifdef C8_PROFILE_WAIT
  if (lane == 0 && p.dbg) {
    long long* d = p.dbg + ((long)blockIdx.x * 4 + wave) * 3;
    d[0] = prof_wait; d[1] = wall_clock64() - prof_w0; d[2] = clock64() - prof_t0;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may land on the scratch (or outlive the wave)
  __syncthreads();                                  // all waves are done reading the staged tiles
  constexpr int NPL = COT / EPR;                    // planes of the tile's channels = KiB of scratch per wave
  float* stat_lds = (float*)(smem_all + 4 * NPL * 1024);            // [wave][COT][2], past every wave's scratch
  unsigned char* pool_lds = smem_all + 4 * NPL * 1024 + 4 * COT * 2 * 4;   // [wave][NPL][512 B]
  if constexpr (EPI == CONV_EPI_HEAD) conv_epilogue_head<T, HT, COT>(p, acc, n, h0, w0 + 32 * wave, lane);
  else conv_epilogue_store<T, HT, COT, EPI, KH == 1 && KW == 1>(p, acc, smem_all + wave * (NPL * 1024), stat_lds + wave * COT * 2,
                                            pool_lds + wave * (NPL * 512), n, h0, w0 + 32 * wave, cot, lane);
  if constexpr (STATS || EPI == CONV_EPI_BNRED) {
    // BatchNorm statistics of this tile: partial[pixel tile][channel][sum, sum of squares] (bn_finalize_kernel's layout);
    // BNRED: [sum dz, sum dz * xhat] (bn_bwd_finalize_kernel's)
    __syncthreads();
    if (tid < COT * 2) {
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) t += stat_lds[wv * COT * 2 + tid];
      p.stats[((long)ptile * p.cout_pad + cot * COT) * 2 + tid] = t;
    }
  }
}

template <typename T, int KH, int KW, int HT, int COT, int EPI = CONV_EPI_PLAIN, int MINB = 2, int KP = 1>
int conv_igemm_launch(ConvParams p, hipStream_t stream) {
  p.ntw = (p.W + 127) / 128;
  p.nth = (p.H + HT - 1) / HT;
  const long blocks = ((long)p.N * p.ntw * p.nth + 7) / 8 * 8 * (p.cout_pad / COT);
  if (blocks <= 0 || blocks > 0x7fffffffL) return dn_set_error(DN_ERR_ARG, "conv grid out of range: %ld blocks", blocks);
  if ((long)p.H * p.W * p.x_wmul * 32 >= (1L << 31)) return dn_set_error(DN_ERR_ARG, "conv: one plane exceeds 2 GiB");
  hipLaunchKernelGGL((conv_igemm_kernel<T, KH, KW, HT, COT, EPI, MINB, KP>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

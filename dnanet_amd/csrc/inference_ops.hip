// Memory-bound stages of the U-Net forward pass on plane-record buffers (see common.h):
//   first convolution (Cin = 1, raw fp32 RFU input), MaxPool2d((1,2)), 1x1 head + sigmoid,
//   weight packing for the implicit-GEMM kernel.
// Reference: DNAnet/models/segmentation/unet_architecture.py:14-20 (first conv), :37 (pool),
// :94-97 (head); DNAnet/models/segmentation/trainable_unet.py:401-409 (sigmoid).
#include "common.h"
#include "ops.h"

// ------------------------------------------------------------------------------------------------
// first convolution: x [N][H][W] fp32 -> y planes.  One thread = one 16-byte output piece (8 bf16 / 4 fp32 channels) of
// one scan column, walking down the dye rows: its KH*KW x PN weights live in registers for the whole walk, the KH x KW
// input window slides (KW new x values per output, coalesced across lanes), stores are 16 B per lane, contiguous.
// Grid (column blocks, planes, images): no division anywhere.
// wf: [taps][cpad] fp32 (already multiplied by the folded BN scale when inference), bias [cpad]
// ------------------------------------------------------------------------------------------------
template <typename T, int KH, int KW>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ wf,
                                                         const float* __restrict__ bias, unsigned char* y,
                                                         int y_planes_total, int y_plane0, int N, int H, int W,
                                                         int cpad, int relu) {
  constexpr int EPR = RecTraits<T>::EPR, PN = Piece<T>::N, TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
  const int plane = blockIdx.y, n = blockIdx.z;
  const int pc = threadIdx.x & 1, w = blockIdx.x * 128 + (threadIdx.x >> 1);
  if (w >= W) return;
  const int c0 = plane * EPR + pc * PN;
  float wt[TAPS][PN], b[PN];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < PN; ++j) wt[t][j] = wf[t * cpad + c0 + j];
#pragma unroll
  for (int j = 0; j < PN; ++j) b[j] = bias ? bias[c0 + j] : 0.f;
  const float* xn = x + (long)n * H * W;
  auto load_row = [&](int hh, float (&row)[KW]) {
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int ww = w + k - PW;
      row[k] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? xn[hh * W + ww] : 0.f;
    }
  };
  float win[KH][KW];                       // rows h-PH .. h+PH of the window
#pragma unroll
  for (int r = 0; r < KH - 1; ++r) load_row(r - PH, win[r + 1]);      // pre-load so that the first shift lines up
  unsigned char* dst = y + ((((long)n * y_planes_total + y_plane0 + plane) * H) * (long)W + w) * 32 + 16 * pc;
  for (int h = 0; h < H; ++h, dst += (long)W * 32) {
#pragma unroll
    for (int r = 0; r < KH - 1; ++r)
#pragma unroll
      for (int k = 0; k < KW; ++k) win[r][k] = win[r + 1][k];
    load_row(h + KH - 1 - PH, win[KH - 1]);
    float acc[PN];
#pragma unroll
    for (int j = 0; j < PN; ++j) acc[j] = b[j];
#pragma unroll
    for (int r = 0; r < KH; ++r)
#pragma unroll
      for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int j = 0; j < PN; ++j) acc[j] = fmaf(win[r][k], wt[r * KW + k][j], acc[j]);
    if (relu) {
#pragma unroll
      for (int j = 0; j < PN; ++j) acc[j] = fmaxf(acc[j], 0.f);
    }
    *(uint4*)dst = Piece<T>::pack(acc);
  }
}

template <typename T>
static int conv_first_t(const float* x, const float* wf, const float* bias, dn_tensor y, int N, int H, int W, int cout,
                        int kh, int kw, int relu, hipStream_t s) {
  constexpr int EPR = RecTraits<T>::EPR;
  const int nplanes = (cout + EPR - 1) / EPR;
  const int cpad = dn_channel_pad(cout);
  if (N >= 65536 || nplanes >= 65536) return dn_set_error(DN_ERR_ARG, "conv_first: batch or channels too large for the grid");
  dim3 grid((unsigned)((W + 127) / 128), nplanes, N);
#define LAUNCH(KH, KW) hipLaunchKernelGGL((conv_first_kernel<T, KH, KW>), grid, dim3(256), 0, s, x, wf, bias, \
    (unsigned char*)y.base, y.planes_total, y.plane0, N, H, W, cpad, relu)
  if (kh == 3 && kw == 5) LAUNCH(3, 5);
  else if (kh == 1 && kw == 3) LAUNCH(1, 3);
  else if (kh == 1 && kw == 1) LAUNCH(1, 1);
  else return dn_set_error(DN_ERR_UNSUPPORTED, "conv_first: kernel size (%d,%d) not built", kh, kw);
#undef LAUNCH
  DN_LAUNCH_CHECK();
  return DN_OK;
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d((1,2)): one thread = one 16-byte output piece
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const unsigned char* x, int x_planes_total, int x_plane0,
                                                      unsigned char* y, int y_planes_total, int y_plane0, int nplanes,
                                                      int N, int H, int Wout) {
  const long total = (long)N * nplanes * H * Wout * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int pc = (int)(i & 1);
    long r = i >> 1;
    const int w = (int)(r % Wout); r /= Wout;
    const int h = (int)(r % H); r /= H;
    const int pl = (int)(r % nplanes);
    const long n = r / nplanes;
    const unsigned char* src = x + ((((n * x_planes_total + x_plane0 + pl) * H + h) * (long)(2 * Wout)) + 2 * w) * 32 + 16 * pc;
    const uint4 a = *(const uint4*)src, b = *(const uint4*)(src + 32);
    float fa[Piece<T>::N], fb[Piece<T>::N];
    Piece<T>::unpack(a, fa); Piece<T>::unpack(b, fb);
#pragma unroll
    for (int k = 0; k < Piece<T>::N; ++k) fa[k] = fmaxf(fa[k], fb[k]);
    *(uint4*)(y + ((((n * y_planes_total + y_plane0 + pl) * H + h) * (long)Wout) + w) * 32 + 16 * pc) = Piece<T>::pack(fa);
  }
}

// ------------------------------------------------------------------------------------------------
// head: logits[n,h,w] = b + sum_c x[n,c,h,w] * w[c];  probs = sigmoid(logits)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_kernel(const unsigned char* x, int x_planes_total, int x_plane0, int nplanes,
                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                   float* logits, float* probs, int N, int H, int W, int lo_off) {
  // lo_off > 0: x is a hi + lo split tensor (bf16x3), its lo part lo_off planes further: both parts enter the dot product
  constexpr int EPR = RecTraits<T>::EPR, PN = Piece<T>::N;
  const long total = (long)N * H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long)gridDim.x * blockDim.x) {
    const long hw = pix % ((long)H * W);
    const long n = pix / ((long)H * W);
    float acc = b[0];
    for (int pl = 0; pl < nplanes; ++pl) {
      const unsigned char* src = x + (((n * x_planes_total + x_plane0 + pl) * (long)H * W) + hw) * 32;
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        float f[PN];
        Piece<T>::unpack(*(const uint4*)(src + 16 * pc), f);
#pragma unroll
        for (int k = 0; k < PN; ++k) acc = fmaf(f[k], w[pl * EPR + pc * PN + k], acc);
        if (lo_off > 0) {
          Piece<T>::unpack(*(const uint4*)(src + (long)lo_off * H * W * 32 + 16 * pc), f);
#pragma unroll
          for (int k = 0; k < PN; ++k) acc = fmaf(f[k], w[pl * EPR + pc * PN + k], acc);
        }
      }
    }
    if (logits) logits[pix] = acc;
    if (probs) probs[pix] = 1.f / (1.f + expf(-acc));
  }
}

// ------------------------------------------------------------------------------------------------
// weight packing: canonical fp32 parameter -> [nchunk][taps][rows_pad][EPR] of T
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pack_element(const float* __restrict__ w, const float* __restrict__ scale, void* out, int cout,
                                             int cin, int taps, int mode, int rows_pad, int nchunk, int gsize, long i) {
  constexpr int EPR = RecTraits<T>::EPR;
  if (mode >= DN_PACK_VEC) {                   // fp32 side images
    float v = 0.f;
    if (mode == DN_PACK_VEC) {                 // [rows_pad]
      if (i < cout) v = w[i];
    } else if (mode == DN_PACK_UP_BIAS) {      // [2 * gsize] in the row order of DN_PACK_CONVT_FWD
      const int co = (int)(i / (2 * EPR)) * EPR + (int)(i % EPR);
      if (co < cout) v = w[co];
    } else {                                   // DN_PACK_FIRST: [taps][rows_pad]
      const int co = (int)(i % rows_pad), tap = (int)(i / rows_pad);
      if (co < cout) v = w[(long)co * taps + tap] * (scale ? scale[co] : 1.f);
    }
    ((float*)out)[i] = v;
    return;
  }
  const int e = (int)(i % EPR);
  long r = i / EPR;
  const int row = (int)(r % rows_pad); r /= rows_pad;
  const int tap = (int)(r % taps);
  const int ck = (int)(r / taps);
  float v = 0.f;
  if (mode == DN_PACK_FWD) {                 // rows = co, K = ci
    const int co = row, ci = ck * EPR + e;
    if (co < cout && ci < cin) v = w[((long)co * cin + ci) * taps + tap] * (scale ? scale[co] : 1.f);
  } else if (mode == DN_PACK_DGRAD) {        // rows = ci, K = co, taps flipped
    const int ci = row, co = ck * EPR + e;
    if (co < cout && ci < cin) v = w[((long)co * cin + ci) * taps + (taps - 1 - tap)];
  } else if (mode == DN_PACK_CONVT_FWD) {    // w: [cin][cout][1][2]; rows = [output plane][k][EPR channels], K = ci:
    // the two output columns of a plane are neighbouring GEMM planes, so one tile holds both and its epilogue writes
    // whole 64-byte pixel pairs (conv_epilogue_store, PAIRS)
    const int k = (row / EPR) & 1, co = (row / (2 * EPR)) * EPR + row % EPR, ci = ck * EPR + e;
    if (co < cout && ci < cin) v = w[((long)ci * cout + co) * 2 + k];
  } else {                                   // DN_PACK_CONVT_DGRAD: rows = ci, K = (k, co) virtual planes
    const int gpl = gsize / EPR;
    const int k = ck / gpl, co = (ck % gpl) * EPR + e, ci = row;
    if (k < 2 && co < cout && ci < cin) v = w[((long)ci * cout + co) * 2 + k];
  }
  ((T*)out)[i] = dn_from_float<T>(v);
}

template <typename T>
__host__ __device__ inline long pack_total(int taps, int mode, int rows_pad, int nchunk, int gsize) {
  if (mode == DN_PACK_VEC) return rows_pad;
  if (mode == DN_PACK_UP_BIAS) return 2L * gsize;
  if (mode == DN_PACK_FIRST) return (long)taps * rows_pad;
  return (long)nchunk * taps * rows_pad * RecTraits<T>::EPR;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                          T* out, int cout, int cin, int taps, int mode, int rows_pad,
                                                          int nchunk, int gsize) {
  const long total = pack_total<T>(taps, mode, rows_pad, nchunk, gsize);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    pack_element<T>(w, scale, out, cout, cin, taps, mode, rows_pad, nchunk, gsize, i);
}

// every image of a job table in one launch: block -> job by binary search on the jobs' first blocks
constexpr int PACK_BLOCK_ELEMS = 2048;
template <typename T>
__global__ __launch_bounds__(256) void pack_batched_kernel(const dn_pack_job* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const dn_pack_job j = jobs[lo];
  const long total = pack_total<T>(j.taps, j.mode, j.rows_pad, j.nchunk, j.gsize);
  const long base = (long)((int)blockIdx.x - j.block0) * PACK_BLOCK_ELEMS;
#pragma unroll
  for (int k = 0; k < PACK_BLOCK_ELEMS / 256; ++k) {
    const long i = base + k * 256 + threadIdx.x;
    if (i < total) pack_element<T>(j.w, j.scale, j.out, j.cout, j.cin, j.taps, j.mode, j.rows_pad, j.nchunk, j.gsize, i);
  }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

// matrix-core path (conv_first_mfma.hip), optionally with BatchNorm statistics
static int conv_first_impl(const float* x, const float* wf, const float* bias, dn_tensor y, int N, int H, int W, int cout, int kh,
                           int kw, int relu, int dtype, float* stats, long stats_capacity, int* ntiles, void* stream) {
  const int yp = dn_planes(cout, dtype);
  DN_CHECK_ARG(y.plane0 >= 0 && y.plane0 + yp <= y.planes_total, "dn_conv_first: y planes [%d,%d) outside buffer of %d", y.plane0,
               y.plane0 + yp, y.planes_total);
  ConvParams p{};
  p.bias = bias;
  p.y = y.base; p.y_planes_total = y.planes_total; p.y_plane0 = y.plane0; p.y_nplanes = yp; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.N = N; p.H = H; p.W = W; p.cout_pad = dn_channel_pad(cout); p.relu = relu;
  if (stats) {
    const int ht = conv_first_row_tile(H);
    const long tiles = (long)N * ((H + ht - 1) / ht) * ((W + 127) / 128);
    DN_CHECK_ARG(tiles * p.cout_pad * 2 <= stats_capacity, "dn_conv_first_stats: partial buffer of %ld floats, need %ld",
                 stats_capacity, tiles * p.cout_pad * 2);
    p.stats = stats;
    *ntiles = (int)tiles;
  }
  return conv_first_mfma_run(x, wf, p, kh, kw, dtype, (hipStream_t)stream);
}

int dn_conv_first_stats(const float* x, const float* wf, const float* bias, dn_tensor y, int N, int H, int W, int cout, int kh,
                        int kw, int dtype, float* partial, long partial_floats, int* ntiles, void* stream) {
  DN_CHECK_ARG(x && wf && y.base && partial && ntiles, "dn_conv_first_stats: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && W > 0 && cout > 0 && y.channels == cout, "dn_conv_first_stats: bad shape");
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv_first_stats: bad dtype %d", dtype);
  DN_CHECK_ARG((kh == 3 && kw == 5) || (kh == 1 && kw == 3), "dn_conv_first_stats: kernel size (%d,%d) has no statistics epilogue", kh, kw);
  return conv_first_impl(x, wf, bias, y, N, H, W, cout, kh, kw, 0, dtype, partial, partial_floats, ntiles, stream);
}

int dn_conv_first(const float* x, const float* wf, const float* bias, dn_tensor y, int N, int H, int W, int cout, int kh,
                  int kw, int relu, int dtype, void* stream) {
  DN_CHECK_ARG(x && wf && y.base, "dn_conv_first: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && W > 0 && cout > 0, "dn_conv_first: bad shape N=%d H=%d W=%d cout=%d", N, H, W, cout);
  DN_CHECK_ARG(y.channels == cout, "dn_conv_first: y.channels=%d != cout=%d", y.channels, cout);
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv_first: bad dtype %d", dtype);
  if ((kh == 3 && kw == 5) || (kh == 1 && kw == 3))
    return conv_first_impl(x, wf, bias, y, N, H, W, cout, kh, kw, relu, dtype, nullptr, 0, nullptr, stream);
  if (dtype == DN_BF16) return conv_first_t<__bf16>(x, wf, bias, y, N, H, W, cout, kh, kw, relu, (hipStream_t)stream);
  if (dtype == DN_F32) return conv_first_t<float>(x, wf, bias, y, N, H, W, cout, kh, kw, relu, (hipStream_t)stream);
  return dn_set_error(DN_ERR_ARG, "dn_conv_first: bad dtype %d", dtype);
}

int dn_conv_first_x3(const float* x, const float* wf, const float* bias, dn_tensor_x3 y, int N, int H, int W, int cout, int kh,
                     int kw, int relu, void* stream) {
  DN_CHECK_ARG(x && wf && y.base, "dn_conv_first_x3: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && W > 0 && cout > 0 && y.channels == cout, "dn_conv_first_x3: bad shape N=%d H=%d W=%d cout=%d", N, H, W, cout);
  DN_CHECK_ARG((kh == 3 && kw == 5) || (kh == 1 && kw == 3), "dn_conv_first_x3: kernel size (%d,%d) is not built (have (3,5), (1,3))", kh, kw);
  const int yp = dn_planes(cout, DN_BF16);
  DN_CHECK_ARG(y.plane0 >= 0 && y.lo_off >= yp && y.plane0 + y.lo_off + yp <= y.planes_total,
               "dn_conv_first_x3: y planes [%d,%d) + lo part at +%d outside buffer of %d", y.plane0, y.plane0 + yp, y.lo_off, y.planes_total);
  ConvParams p{};
  p.bias = bias;
  p.y = y.base; p.y_planes_total = y.planes_total; p.y_plane0 = y.plane0; p.y_nplanes = yp; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.y_lo_off = y.lo_off;
  p.N = N; p.H = H; p.W = W; p.cout_pad = dn_channel_pad(cout); p.relu = relu;
  return conv_first_mfma_run(x, wf, p, kh, kw, DN_BF16, (hipStream_t)stream);
}

int dn_maxpool_1x2(dn_tensor x, dn_tensor y, int N, int H, int Wout, int dtype, void* stream) {
  DN_CHECK_ARG(x.base && y.base, "dn_maxpool_1x2: null pointer");
  DN_CHECK_ARG(x.channels == y.channels && N > 0 && H > 0 && Wout > 0, "dn_maxpool_1x2: bad shape");
  const int epr = dtype == DN_BF16 ? 16 : 8;
  const int nplanes = (x.channels + epr - 1) / epr;
  const long total = (long)N * nplanes * H * Wout * 2;
  const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 8192);
  if (dtype == DN_BF16)
    hipLaunchKernelGGL(maxpool_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x.base,
                       x.planes_total, x.plane0, (unsigned char*)y.base, y.planes_total, y.plane0, nplanes, N, H, Wout);
  else if (dtype == DN_F32)
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x.base,
                       x.planes_total, x.plane0, (unsigned char*)y.base, y.planes_total, y.plane0, nplanes, N, H, Wout);
  else return dn_set_error(DN_ERR_ARG, "dn_maxpool_1x2: bad dtype %d", dtype);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

int dn_head_1x1(dn_tensor x, const float* w, const float* b, float* logits, float* probs, int N, int H, int W, int dtype,
                void* stream) {
  DN_CHECK_ARG(x.base && w && b && (logits || probs), "dn_head_1x1: null pointer");
  const int epr = dtype == DN_BF16 ? 16 : 8;
  const int nplanes = (x.channels + epr - 1) / epr;
  const long total = (long)N * H * W;
  const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 8192);
  if (dtype == DN_BF16)
    hipLaunchKernelGGL(head_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x.base,
                       x.planes_total, x.plane0, nplanes, w, b, logits, probs, N, H, W, 0);
  else if (dtype == DN_F32)
    hipLaunchKernelGGL(head_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x.base,
                       x.planes_total, x.plane0, nplanes, w, b, logits, probs, N, H, W, 0);
  else return dn_set_error(DN_ERR_ARG, "dn_head_1x1: bad dtype %d", dtype);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

int dn_head_1x1_x3(dn_tensor_x3 x, const float* w, const float* b, float* logits, float* probs, int N, int H, int W, void* stream) {
  DN_CHECK_ARG(x.base && w && b && (logits || probs), "dn_head_1x1_x3: null pointer");
  const int nplanes = dn_planes(x.channels, DN_BF16);
  DN_CHECK_ARG(x.plane0 >= 0 && (x.lo_off == 0 || x.lo_off >= nplanes) && x.plane0 + x.lo_off + nplanes <= x.planes_total,
               "dn_head_1x1_x3: planes outside buffer");
  const long total = (long)N * H * W;
  const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(head_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x.base,
                     x.planes_total, x.plane0, nplanes, w, b, logits, probs, N, H, W, x.lo_off);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

long dn_packed_weight_bytes(int cout, int cin, int kh, int kw, int mode, int dtype) {
  const int epr = dtype == DN_BF16 ? 16 : 8;
  int rows_pad, nchunk, taps = kh * kw;
  if (mode == DN_PACK_FWD) { rows_pad = dn_channel_pad(cout); nchunk = (cin + epr - 1) / epr; }
  else if (mode == DN_PACK_DGRAD) { rows_pad = dn_channel_pad(cin); nchunk = (cout + epr - 1) / epr; }
  else if (mode == DN_PACK_CONVT_FWD) { rows_pad = 2 * dn_channel_pad(cout); nchunk = (cin + epr - 1) / epr; taps = 1; }
  else if (mode == DN_PACK_CONVT_DGRAD) { rows_pad = dn_channel_pad(cin); nchunk = 2 * (dn_channel_pad(cout) / epr); taps = 1; }
  else return -1;
  return (long)nchunk * taps * rows_pad * 32;
}

int dn_pack_weight(const float* w, const float* scale, void* packed, int cout, int cin, int kh, int kw, int mode, int dtype,
                   void* stream) {
  DN_CHECK_ARG(w && packed, "dn_pack_weight: null pointer");
  const int epr = dtype == DN_BF16 ? 16 : 8;
  int rows_pad, nchunk, taps = kh * kw, gsize = dn_channel_pad(cout);
  if (mode == DN_PACK_FWD) { rows_pad = dn_channel_pad(cout); nchunk = (cin + epr - 1) / epr; }
  else if (mode == DN_PACK_DGRAD) { rows_pad = dn_channel_pad(cin); nchunk = (cout + epr - 1) / epr; }
  else if (mode == DN_PACK_CONVT_FWD) { rows_pad = 2 * gsize; nchunk = (cin + epr - 1) / epr; taps = 1; }
  else if (mode == DN_PACK_CONVT_DGRAD) { rows_pad = dn_channel_pad(cin); nchunk = 2 * (gsize / epr); taps = 1; }
  else return dn_set_error(DN_ERR_ARG, "dn_pack_weight: bad mode %d", mode);
  const long total = (long)nchunk * taps * rows_pad * epr;
  const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 4096);
  if (dtype == DN_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, scale, (__bf16*)packed,
                       cout, cin, taps, mode, rows_pad, nchunk, gsize);
  else if (dtype == DN_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, scale, (float*)packed,
                       cout, cin, taps, mode, rows_pad, nchunk, gsize);
  else return dn_set_error(DN_ERR_ARG, "dn_pack_weight: bad dtype %d", dtype);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

static int pack_geometry(int cout, int cin, int mode, int epr, int* rows_pad, int* nchunk, int* gsize, int* taps) {
  *gsize = dn_channel_pad(cout);
  if (mode == DN_PACK_FWD) { *rows_pad = dn_channel_pad(cout); *nchunk = (cin + epr - 1) / epr; }
  else if (mode == DN_PACK_DGRAD) { *rows_pad = dn_channel_pad(cin); *nchunk = (cout + epr - 1) / epr; }
  else if (mode == DN_PACK_CONVT_FWD) { *rows_pad = 2 * *gsize; *nchunk = (cin + epr - 1) / epr; *taps = 1; }
  else if (mode == DN_PACK_CONVT_DGRAD) { *rows_pad = dn_channel_pad(cin); *nchunk = 2 * (*gsize / epr); *taps = 1; }
  else if (mode == DN_PACK_VEC || mode == DN_PACK_UP_BIAS || mode == DN_PACK_FIRST) { *rows_pad = *gsize; *nchunk = 1; }
  else return -1;
  return 0;
}

long dn_pack_jobs_prepare(dn_pack_job* jobs, int njobs, int dtype) {
  if (!jobs || njobs <= 0 || (dtype != DN_BF16 && dtype != DN_F32)) { dn_set_error(DN_ERR_ARG, "dn_pack_jobs_prepare: bad argument"); return -1; }
  const int epr = dtype == DN_BF16 ? 16 : 8;
  long block = 0;
  for (int i = 0; i < njobs; ++i) {
    dn_pack_job& j = jobs[i];
    if (!j.w || !j.out || j.cout <= 0 || j.cin <= 0 || j.taps <= 0 ||
        pack_geometry(j.cout, j.cin, j.mode, epr, &j.rows_pad, &j.nchunk, &j.gsize, &j.taps)) {
      dn_set_error(DN_ERR_ARG, "dn_pack_jobs_prepare: job %d is malformed (mode %d)", i, j.mode);
      return -1;
    }
    const long total = dtype == DN_BF16 ? pack_total<__bf16>(j.taps, j.mode, j.rows_pad, j.nchunk, j.gsize)
                                        : pack_total<float>(j.taps, j.mode, j.rows_pad, j.nchunk, j.gsize);
    j.block0 = (int)block;
    block += (total + PACK_BLOCK_ELEMS - 1) / PACK_BLOCK_ELEMS;
    if (block > 0x7fffffffL) { dn_set_error(DN_ERR_ARG, "dn_pack_jobs_prepare: too many blocks"); return -1; }
  }
  return block;
}

int dn_pack_batched(const dn_pack_job* jobs_device, int njobs, long grid_blocks, int dtype, void* stream) {
  DN_CHECK_ARG(jobs_device && njobs > 0 && grid_blocks > 0 && grid_blocks <= 0x7fffffffL, "dn_pack_batched: bad argument");
  if (dtype == DN_BF16)
    hipLaunchKernelGGL(pack_batched_kernel<__bf16>, dim3((unsigned)grid_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs);
  else if (dtype == DN_F32)
    hipLaunchKernelGGL(pack_batched_kernel<float>, dim3((unsigned)grid_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs);
  else return dn_set_error(DN_ERR_ARG, "dn_pack_batched: bad dtype %d", dtype);
  DN_LAUNCH_CHECK();
  return DN_OK;
}

}  // extern "C"

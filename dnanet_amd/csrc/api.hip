// C-ABI glue: error text, convolution dispatch. See include/dnanet_hip.h.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "ops.h"

static thread_local char g_err[512] = "";

// the DN_TUNE_* switches are read here, once, when the library is loaded
__attribute__((constructor)) static void dn_read_tuning_switches() { (void)dn_tune(); }

int dn_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// Which instantiation a convolution goes to.  One function, used by the dispatcher and by dn_conv2d_variant (the
// name bench.py groups launches by), so the two cannot drift apart.
struct ConvVariant { bool wave8; int ht, cot; };
static ConvVariant choose_variant(const ConvParams& p, int kh, int kw, int dtype) {
  ConvVariant v{false, 5, (p.cout_pad % 64 == 0) ? 64 : 32};
  if (kh == 3 && kw == 5 && p.H % 3 == 0 && p.H % 5 != 0) v.ht = 3;   // e.g. 6 dye rows = two 3-row tiles
  const DnTune& tune = dn_tune();
  if (tune.cot == 32) v.cot = 32;
  // narrow (32-channel) tiles have room for 6 accumulator rows: one 6-row tile instead of two 3-row ones halves the
  // per-block fixed cost that dominates the 2-4 chunk layers
  if (kh == 3 && kw == 5 && v.cot == 32 && v.ht == 3 && p.H % 6 == 0) v.ht = 6;
  // a 1x1 GEMM (transposed convolution) is bound by how often the activation tile is re-streamed from L2: 128 output
  // channels x 2 rows halves that against 64 x 5 (rows are 128-pixel slices of the flattened plane, so H is even)
  if (!tune.no_cot128 && kh == 1 && kw == 1 && p.cout_pad % 128 == 0 && p.H % 2 == 0) { v.ht = 2; v.cot = 128; return v; }
  // measured (N=256, W=4096 net): the 8-wave kernels win for 5-row tiles and for fp32; bf16 3-row tiles (6 dyes) are
  // faster on the 4-wave kernel at every level
  v.wave8 = !tune.no_conv8 && v.cot == 64 && conv_igemm8_applies(p, kh, kw) &&
            (dtype == DN_F32 || (dtype == DN_BF16 && (v.ht == 5 || tune.conv8_ht3)));
  if (p.head_w) v.wave8 = false;     // the fused head lives in the 4-wave kernel's epilogue
  if (p.y_lo_off && v.ht != 5) v.wave8 = false;   // split epilogue: the 8-wave kernel is built for 5-row tiles only
  if (p.stats && tune.stats_4wave) v.wave8 = false;
  // Small batches (train.py's 16 profiles): the wide tiles leave CUs idle -- every 8-wave launch of the shipped net
  // is exactly 128 workgroups at N = 16.  Step down to the tile that still gives two workgroups per CU.
  if (p.N > 0) {
    const long rows = (long)p.N * ((p.H + v.ht - 1) / v.ht);
    const long blocks8 = rows * (p.W / 256) * (p.cout_pad / 64);
    const long blocks64 = rows * ((p.W + 127) / 128) * (p.cout_pad / 64);
    if (v.wave8 && blocks8 < 512) v.wave8 = false;
    if (!v.wave8 && v.cot == 64 && blocks64 < 512) v.cot = 32;
  }
  return v;
}

static int run_igemm(const ConvParams& p, int kh, int kw, int dtype, hipStream_t s) {
  if (dtype != DN_BF16 && dtype != DN_F32) return dn_set_error(DN_ERR_ARG, "bad dtype %d", dtype);
  const ConvVariant v = choose_variant(p, kh, kw, dtype);
  const int ht = v.ht, cot = v.cot;
  if (p.bn_y) {                      // data gradient + BatchNorm backward sums of the unit it feeds
    if (kh == 1 && kw == 1) return dtype == DN_BF16 ? conv_igemm_run_bf16_k11_bnred(p, ht, cot, s) : conv_igemm_run_f32_k11_bnred(p, ht, cot, s);
    if (!(kh == 3 && kw == 5)) return dn_set_error(DN_ERR_UNSUPPORTED, "BatchNorm-sums epilogue: kernel size (%d,%d) is not built (have (3,5), (1,1))", kh, kw);
    if (p.bias || p.relu || p.y_wmul != 1 || !p.stats) return dn_set_error(DN_ERR_ARG, "BatchNorm-sums epilogue: a plain data gradient only");
    if (v.wave8) return dtype == DN_BF16 ? conv_igemm8_run_bf16_bnred(p, ht, s) : conv_igemm8_run_f32_bnred(p, ht, s);
    return dtype == DN_BF16 ? conv_igemm_run_bf16_k35_bnred(p, ht, cot, s) : conv_igemm_run_f32_k35_bnred(p, ht, cot, s);
  }
  if (p.y_lo_off) {                  // hi + lo split output (bf16x3 inference mode)
    if (dtype != DN_BF16) return dn_set_error(DN_ERR_ARG, "split output needs bf16 storage");
    if (v.wave8) return conv_igemm8_run_bf16_split(p, ht, s);
    if (kh == 3 && kw == 5) return conv_igemm_run_bf16_k35_split(p, ht, cot, s);
    if (kh == 1 && kw == 3) return conv_igemm_run_bf16_k13_split(p, ht, cot, s);
    if (kh == 1 && kw == 1) return conv_igemm_run_bf16_k11_split(p, ht, cot, s);
    return dn_set_error(DN_ERR_UNSUPPORTED, "split output: kernel size (%d,%d) is not built (have (3,5), (1,3), (1,1))", kh, kw);
  }
  if (v.wave8) return dtype == DN_BF16 ? conv_igemm8_run_bf16(p, ht, s) : conv_igemm8_run_f32(p, ht, s);
  if (dtype == DN_BF16) {
    if (kh == 3 && kw == 5) return conv_igemm_run_bf16_k35(p, ht, cot, s);
    if (kh == 1 && kw == 3) return conv_igemm_run_bf16_k13(p, ht, cot, s);
    if (kh == 1 && kw == 1) return conv_igemm_run_bf16_k11(p, ht, cot, s);
  } else if (dtype == DN_F32) {
    if (kh == 3 && kw == 5) return conv_igemm_run_f32_k35(p, ht, cot, s);
    if (kh == 1 && kw == 3) return conv_igemm_run_f32_k13(p, ht, cot, s);
    if (kh == 1 && kw == 1) return conv_igemm_run_f32_k11(p, ht, cot, s);
  } else {
    return dn_set_error(DN_ERR_ARG, "bad dtype %d", dtype);
  }
  return dn_set_error(DN_ERR_UNSUPPORTED, "convolution kernel size (%d,%d) is not built (have (3,5), (1,3), (1,1))", kh, kw);
}

// A 1x1 GEMM has no halo, so the H x W pixels of a plane are one contiguous run: re-tile them as rows of 128
// pixels so that 5-row tiles are always full (6 dye rows would otherwise leave a 1-row tile).
static void flatten_rows(ConvParams& p) {
  if (p.W % 128 == 0 && p.W > 128) { p.H = p.H * (p.W / 128); p.W = 128; }
}

extern "C" {

const char* dn_last_error(void) { return g_err; }
int dn_abi_version(void) { return 4; }   // 4: the *_bn_sums entry points (round 3)

int dn_conv2d_variant(int cin, int cout, int N, int H, int W, int kh, int kw, int dtype, char* buf, int buf_len) {
  DN_CHECK_ARG(buf && buf_len > 0, "dn_conv2d_variant: no buffer");
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv2d_variant: bad dtype %d", dtype);
  ConvParams p{};
  p.N = N; p.H = H; p.W = W; p.nchunk = dn_planes(cin, dtype); p.cout_pad = dn_channel_pad(cout); p.x_wmul = 1; p.y_wmul = 1;
  const ConvVariant v = choose_variant(p, kh, kw, dtype);
  snprintf(buf, buf_len, "%s<%s,%dx%d,ht%d,cot%d>", v.wave8 ? (v.ht == 3 ? "conv_igemm8c" : "conv_igemm8") : "conv_igemm",
           dtype == DN_BF16 ? "bf16" : "f32", kh, kw, v.ht, v.cot);
  return DN_OK;
}

struct BnSums { dn_tensor y; const float *scale, *shift, *mean, *invstd; };   // the unit a data gradient feeds (CONV_EPI_BNRED)
static int conv2d_impl(dn_tensor x, const void* packed_w, const float* bias, dn_tensor y, int N, int H, int W, int kh, int kw,
                       int relu, int dtype, float* stats, long stats_capacity, int* ntiles, void* stream,
                       const dn_tensor* pooled = nullptr, const BnSums* bn = nullptr);

// The fused head needs the whole channel dimension of a pixel in one block: the tile the dispatcher picks for this
// shape must be as wide as the padded channel count.
static bool head_fits(int cin, int cout, int N, int H, int W, int kh, int kw, int dtype) {
  if (!((kh == 3 && kw == 5) || (kh == 1 && kw == 3)) || cin <= 1 || cout <= 0 || N <= 0 || H <= 0 || W <= 0) return false;
  if (dtype != DN_BF16 && dtype != DN_F32) return false;
  if (dn_tune().no_head_fuse) return false;
  ConvParams p{};
  static const float one = 1.f;
  p.N = N; p.H = H; p.W = W; p.nchunk = dn_planes(cin, dtype); p.cout_pad = dn_channel_pad(cout); p.x_wmul = 1; p.y_wmul = 1;
  p.head_w = &one;
  const ConvVariant v = choose_variant(p, kh, kw, dtype);
  return !v.wave8 && v.cot == p.cout_pad;
}

int dn_conv2d_head_supported(int cin, int cout, int N, int H, int W, int kh, int kw, int dtype) {
  return head_fits(cin, cout, N, H, W, kh, kw, dtype) ? 1 : 0;
}

int dn_conv2d_head(dn_tensor x, const void* packed_w, const float* bias, int cout, const float* head_w, const float* head_b,
                   float* logits, float* probs, int N, int H, int W, int kh, int kw, int relu, int dtype, void* stream) {
  DN_CHECK_ARG(x.base && packed_w && head_w && head_b && (logits || probs), "dn_conv2d_head: null pointer");
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv2d_head: bad dtype %d", dtype);
  if (!head_fits(x.channels, cout, N, H, W, kh, kw, dtype))
    return dn_set_error(DN_ERR_UNSUPPORTED, "dn_conv2d_head: no single-tile kernel for cout=%d at N=%d H=%d W=%d k=(%d,%d); "
                        "use dn_conv2d + dn_head_1x1", cout, N, H, W, kh, kw);
  const int xp = dn_planes(x.channels, dtype);
  DN_CHECK_ARG(x.plane0 >= 0 && x.plane0 + xp <= x.planes_total, "dn_conv2d_head: x planes [%d,%d) outside buffer of %d",
               x.plane0, x.plane0 + xp, x.planes_total);
  ConvParams p{};
  p.x = x.base; p.x_planes_total = x.planes_total; p.x_plane0 = x.plane0; p.x_nplanes = xp; p.x_gplanes = 1 << 30; p.x_wmul = 1;
  p.w = packed_w; p.bias = bias;
  p.y = nullptr; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.N = N; p.H = H; p.W = W; p.nchunk = xp; p.cout_pad = dn_channel_pad(cout); p.relu = relu;
  p.head_w = head_w; p.head_b = head_b; p.head_logits = logits; p.head_probs = probs;
  return run_igemm(p, kh, kw, dtype, (hipStream_t)stream);
}

int dn_conv2d_pool(dn_tensor x, const void* packed_w, const float* bias, dn_tensor y, dn_tensor pooled, int N, int H, int W,
                   int kh, int kw, int relu, int dtype, void* stream) {
  DN_CHECK_ARG(pooled.base, "dn_conv2d_pool: null pointer");
  DN_CHECK_ARG((kh == 3 && kw == 5) || (kh == 1 && kw == 3), "dn_conv2d_pool: kernel size (%d,%d) has no pooling epilogue", kh, kw);
  DN_CHECK_ARG(W % 2 == 0 && pooled.channels == y.channels, "dn_conv2d_pool: W=%d must be even and pooled.channels=%d == y.channels=%d",
               W, pooled.channels, y.channels);
  const int pp = dn_planes(pooled.channels, dtype);
  DN_CHECK_ARG(pooled.plane0 >= 0 && pooled.plane0 + pp <= pooled.planes_total, "dn_conv2d_pool: pooled planes [%d,%d) outside buffer of %d",
               pooled.plane0, pooled.plane0 + pp, pooled.planes_total);
  return conv2d_impl(x, packed_w, bias, y, N, H, W, kh, kw, relu, dtype, nullptr, 0, nullptr, stream, &pooled);
}

int dn_conv2d_stats(dn_tensor x, const void* packed_w, const float* bias, dn_tensor y, int N, int H, int W, int kh, int kw,
                    int dtype, float* partial, long partial_floats, int* ntiles, void* stream) {
  DN_CHECK_ARG(partial && ntiles, "dn_conv2d_stats: null pointer");
  DN_CHECK_ARG((kh == 3 && kw == 5) || (kh == 1 && kw == 3), "dn_conv2d_stats: kernel size (%d,%d) has no statistics epilogue", kh, kw);
  return conv2d_impl(x, packed_w, bias, y, N, H, W, kh, kw, 0, dtype, partial, partial_floats, ntiles, stream);
}

int dn_conv2d_dgrad_bn_sums(dn_tensor dy, const void* packed_wd, dn_tensor dx, dn_tensor bn_y, const float* scale,
                            const float* shift, const float* mean, const float* invstd, int N, int H, int W, int kh, int kw,
                            int dtype, float* partial, long partial_floats, int* ntiles, void* stream) {
  DN_CHECK_ARG(partial && ntiles && bn_y.base && scale && shift && mean && invstd, "dn_conv2d_dgrad_bn_sums: null pointer");
  DN_CHECK_ARG(kh == 3 && kw == 5, "dn_conv2d_dgrad_bn_sums: kernel size (%d,%d) has no BatchNorm-sums epilogue", kh, kw);
  DN_CHECK_ARG(bn_y.channels == dx.channels, "dn_conv2d_dgrad_bn_sums: bn_y has %d channels, dx %d", bn_y.channels, dx.channels);
  const int bp = dn_planes(bn_y.channels, dtype);
  DN_CHECK_ARG(bn_y.plane0 >= 0 && bn_y.plane0 + bp <= bn_y.planes_total, "dn_conv2d_dgrad_bn_sums: bn_y planes [%d,%d) outside buffer of %d",
               bn_y.plane0, bn_y.plane0 + bp, bn_y.planes_total);
  const BnSums bn{bn_y, scale, shift, mean, invstd};
  return conv2d_impl(dy, packed_wd, nullptr, dx, N, H, W, kh, kw, 0, dtype, partial, partial_floats, ntiles, stream, nullptr, &bn);
}

int dn_conv2d(dn_tensor x, const void* packed_w, const float* bias, dn_tensor y, int N, int H, int W, int kh, int kw,
              int relu, int dtype, void* stream) {
  return conv2d_impl(x, packed_w, bias, y, N, H, W, kh, kw, relu, dtype, nullptr, 0, nullptr, stream);
}

int dn_conv_transpose_1x2(dn_tensor x, const void* packed_w, const float* bias2, dn_tensor y, int N, int H, int Win,
                          int dtype, void* stream) {
  DN_CHECK_ARG(x.base && packed_w && y.base, "dn_conv_transpose_1x2: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && Win > 0, "dn_conv_transpose_1x2: bad shape");
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv_transpose_1x2: bad dtype %d", dtype);
  const int xp = dn_planes(x.channels, dtype), yp = dn_planes(y.channels, dtype);
  DN_CHECK_ARG(x.plane0 + xp <= x.planes_total && y.plane0 + yp <= y.planes_total, "dn_conv_transpose_1x2: planes outside buffer");
  const int gsize = dn_channel_pad(y.channels);
  ConvParams p{};
  p.x = x.base; p.x_planes_total = x.planes_total; p.x_plane0 = x.plane0; p.x_nplanes = xp; p.x_gplanes = 1 << 30; p.x_wmul = 1;
  p.w = packed_w; p.bias = bias2;
  // y_wmul = 2: GEMM plane 2j + k is output column 2w + k of plane j (the epilogue stores the pair together)
  p.y = y.base; p.y_planes_total = y.planes_total; p.y_plane0 = y.plane0; p.y_nplanes = yp; p.y_gplanes = 1 << 30; p.y_wmul = 2;
  p.N = N; p.H = H; p.W = Win; p.nchunk = xp; p.cout_pad = 2 * gsize; p.relu = 0;
  flatten_rows(p);
  return run_igemm(p, 1, 1, dtype, (hipStream_t)stream);
}

static int conv_transpose_dgrad_impl(dn_tensor dy, const void* packed_w, dn_tensor dx, int N, int H, int Win, int dtype, void* stream,
                                     const BnSums* bn, float* sums, long sums_floats, int* ntiles);

int dn_conv_transpose_1x2_dgrad(dn_tensor dy, const void* packed_w, dn_tensor dx, int N, int H, int Win, int dtype,
                                void* stream) {
  return conv_transpose_dgrad_impl(dy, packed_w, dx, N, H, Win, dtype, stream, nullptr, nullptr, 0, nullptr);
}

int dn_conv_transpose_1x2_dgrad_bn_sums(dn_tensor dy, const void* packed_w, dn_tensor dx, dn_tensor bn_y, const float* scale,
                                        const float* shift, const float* mean, const float* invstd, int N, int H, int Win, int dtype,
                                        float* sums, long sums_floats, int* ntiles, void* stream) {
  DN_CHECK_ARG(sums && ntiles && bn_y.base && scale && shift && mean && invstd, "dn_conv_transpose_1x2_dgrad_bn_sums: null pointer");
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv_transpose_1x2_dgrad_bn_sums: bad dtype %d", dtype);
  DN_CHECK_ARG(bn_y.channels == dx.channels, "dn_conv_transpose_1x2_dgrad_bn_sums: bn_y has %d channels, dx %d", bn_y.channels, dx.channels);
  const int bp = dn_planes(bn_y.channels, dtype);
  DN_CHECK_ARG(bn_y.plane0 >= 0 && bn_y.plane0 + bp <= bn_y.planes_total, "dn_conv_transpose_1x2_dgrad_bn_sums: bn_y planes [%d,%d) outside buffer of %d",
               bn_y.plane0, bn_y.plane0 + bp, bn_y.planes_total);
  const BnSums bn{bn_y, scale, shift, mean, invstd};
  return conv_transpose_dgrad_impl(dy, packed_w, dx, N, H, Win, dtype, stream, &bn, sums, sums_floats, ntiles);
}

static int conv_transpose_dgrad_impl(dn_tensor dy, const void* packed_w, dn_tensor dx, int N, int H, int Win, int dtype, void* stream,
                                     const BnSums* bn, float* sums, long sums_floats, int* ntiles) {
  DN_CHECK_ARG(dy.base && packed_w && dx.base, "dn_conv_transpose_1x2_dgrad: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && Win > 0, "dn_conv_transpose_1x2_dgrad: bad shape");
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv_transpose_1x2_dgrad: bad dtype %d", dtype);
  const int epr = dtype == DN_BF16 ? 16 : 8;
  const int yp = dn_planes(dy.channels, dtype), xp = dn_planes(dx.channels, dtype);
  DN_CHECK_ARG(dy.plane0 + yp <= dy.planes_total && dx.plane0 + xp <= dx.planes_total, "dn_conv_transpose_1x2_dgrad: planes outside buffer");
  const int gsize = dn_channel_pad(dy.channels);
  ConvParams p{};
  p.x = dy.base; p.x_planes_total = dy.planes_total; p.x_plane0 = dy.plane0; p.x_nplanes = yp; p.x_gplanes = gsize / epr; p.x_wmul = 2; p.x_gstride = 1;
  p.w = packed_w; p.bias = nullptr;
  p.y = dx.base; p.y_planes_total = dx.planes_total; p.y_plane0 = dx.plane0; p.y_nplanes = xp; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.N = N; p.H = H; p.W = Win; p.nchunk = 2 * (gsize / epr); p.cout_pad = dn_channel_pad(dx.channels); p.relu = 0;
  flatten_rows(p);
  if (bn) {        // (the unit's saved convolution output has dx's pixel grid, so the flattened rows address it as well)
    p.bn_y = bn->y.base; p.bn_planes_total = bn->y.planes_total; p.bn_plane0 = bn->y.plane0;
    p.bn_scale = bn->scale; p.bn_shift = bn->shift; p.bn_mean = bn->mean; p.bn_invstd = bn->invstd;
    p.stats = sums;
    const ConvVariant v = choose_variant(p, 1, 1, dtype);
    const long tiles = (long)N * ((p.H + v.ht - 1) / v.ht) * ((p.W + 127) / 128);
    DN_CHECK_ARG(tiles * p.cout_pad * 2 <= sums_floats, "dn_conv_transpose_1x2_dgrad_bn_sums: sums buffer of %ld floats, need %ld",
                 sums_floats, tiles * p.cout_pad * 2);
    *ntiles = (int)tiles;
  }
  return run_igemm(p, 1, 1, dtype, (hipStream_t)stream);
}

// ---- bf16x3: hi + lo split activations and weights (inference).  See include/dnanet_hip.h.
// K walk of a split input [hi | lo | hi] against the image [W_hi | W_hi | W_lo]; of a plain input [x | x] against [W_hi | W_lo].
static int x3_input(ConvParams& p, const dn_tensor_x3& x, const char* who) {
  const int xp = dn_planes(x.channels, DN_BF16);
  if (!x.base || x.plane0 < 0 || x.lo_off < 0 || x.plane0 + (x.lo_off ? x.lo_off : 0) + xp > x.planes_total)
    return dn_set_error(DN_ERR_ARG, "%s: x planes [%d,%d) (+ lo part at +%d) outside buffer of %d", who, x.plane0, x.plane0 + xp,
                        x.lo_off, x.planes_total);
  if (x.lo_off && x.lo_off < xp) return dn_set_error(DN_ERR_ARG, "%s: x.lo_off %d overlaps the %d hi planes", who, x.lo_off, xp);
  p.x = x.base; p.x_planes_total = x.planes_total; p.x_plane0 = x.plane0; p.x_nplanes = xp; p.x_gplanes = xp; p.x_wmul = 1;
  p.x_gstride = 0; p.x_lo_off = x.lo_off;
  p.nchunk = (x.lo_off ? 3 : 2) * xp;
  return DN_OK;
}
static int x3_output(ConvParams& p, const dn_tensor_x3& y, const char* who) {
  const int yp = dn_planes(y.channels, DN_BF16);
  if (!y.base || y.plane0 < 0 || y.lo_off < yp || y.plane0 + y.lo_off + yp > y.planes_total)
    return dn_set_error(DN_ERR_ARG, "%s: y planes [%d,%d) + lo part at +%d outside buffer of %d (a split output needs lo_off >= its planes)",
                        who, y.plane0, y.plane0 + yp, y.lo_off, y.planes_total);
  p.y = y.base; p.y_planes_total = y.planes_total; p.y_plane0 = y.plane0; p.y_nplanes = yp; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.y_lo_off = y.lo_off;
  return DN_OK;
}

int dn_conv2d_x3(dn_tensor_x3 x, const void* packed_w, const float* bias, dn_tensor_x3 y, const dn_tensor_x3* pooled, int N, int H,
                 int W, int kh, int kw, int relu, void* stream) {
  DN_CHECK_ARG(packed_w, "dn_conv2d_x3: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && W > 0 && x.channels > 0 && y.channels > 0, "dn_conv2d_x3: bad shape N=%d H=%d W=%d", N, H, W);
  DN_CHECK_ARG((kh == 3 && kw == 5) || (kh == 1 && kw == 3), "dn_conv2d_x3: kernel size (%d,%d) is not built (have (3,5), (1,3))", kh, kw);
  ConvParams p{};
  if (int rc = x3_input(p, x, "dn_conv2d_x3")) return rc;
  if (int rc = x3_output(p, y, "dn_conv2d_x3")) return rc;
  p.w = packed_w; p.bias = bias;
  p.N = N; p.H = H; p.W = W; p.cout_pad = dn_channel_pad(y.channels); p.relu = relu;
  if (pooled) {
    const int pp = dn_planes(pooled->channels, DN_BF16);
    DN_CHECK_ARG(pooled->base && W % 2 == 0 && pooled->channels == y.channels, "dn_conv2d_x3: pooled tensor needs W even and y's channels");
    DN_CHECK_ARG(pooled->plane0 >= 0 && (pooled->lo_off == 0 || pooled->lo_off >= pp) &&
                 pooled->plane0 + pooled->lo_off + pp <= pooled->planes_total, "dn_conv2d_x3: pooled planes outside buffer");
    p.pool = pooled->base; p.pool_planes_total = pooled->planes_total; p.pool_plane0 = pooled->plane0; p.pool_lo_off = pooled->lo_off;
  }
  return run_igemm(p, kh, kw, DN_BF16, (hipStream_t)stream);
}

int dn_conv2d_head_x3(dn_tensor_x3 x, const void* packed_w, const float* bias, int cout, const float* head_w, const float* head_b,
                      float* logits, float* probs, int N, int H, int W, int kh, int kw, int relu, void* stream) {
  DN_CHECK_ARG(packed_w && head_w && head_b && (logits || probs), "dn_conv2d_head_x3: null pointer");
  if (!head_fits(x.channels, cout, N, H, W, kh, kw, DN_BF16))
    return dn_set_error(DN_ERR_UNSUPPORTED, "dn_conv2d_head_x3: no single-tile kernel for cout=%d at N=%d H=%d W=%d k=(%d,%d)", cout, N,
                        H, W, kh, kw);
  ConvParams p{};
  if (int rc = x3_input(p, x, "dn_conv2d_head_x3")) return rc;
  p.w = packed_w; p.bias = bias;
  p.y = nullptr; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.N = N; p.H = H; p.W = W; p.cout_pad = dn_channel_pad(cout); p.relu = relu;
  p.head_w = head_w; p.head_b = head_b; p.head_logits = logits; p.head_probs = probs; p.head_exact = 1;
  return run_igemm(p, kh, kw, DN_BF16, (hipStream_t)stream);
}

int dn_conv_transpose_1x2_x3(dn_tensor_x3 x, const void* packed_w, const float* bias2, dn_tensor_x3 y, int N, int H, int Win,
                             void* stream) {
  DN_CHECK_ARG(packed_w, "dn_conv_transpose_1x2_x3: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && Win > 0, "dn_conv_transpose_1x2_x3: bad shape");
  ConvParams p{};
  if (int rc = x3_input(p, x, "dn_conv_transpose_1x2_x3")) return rc;
  if (int rc = x3_output(p, y, "dn_conv_transpose_1x2_x3")) return rc;
  p.w = packed_w; p.bias = bias2;
  p.y_wmul = 2;        // GEMM plane 2j + k is output column 2w + k of plane j (the epilogue stores the pair together)
  p.N = N; p.H = H; p.W = Win; p.cout_pad = 2 * dn_channel_pad(y.channels); p.relu = 0;
  flatten_rows(p);
  return run_igemm(p, 1, 1, DN_BF16, (hipStream_t)stream);
}

}  // extern "C"

static int conv2d_impl(dn_tensor x, const void* packed_w, const float* bias, dn_tensor y, int N, int H, int W, int kh, int kw,
                       int relu, int dtype, float* stats, long stats_capacity, int* ntiles, void* stream, const dn_tensor* pooled,
                       const BnSums* bn) {
  DN_CHECK_ARG(x.base && packed_w && y.base, "dn_conv2d: null pointer");
  DN_CHECK_ARG(N > 0 && H > 0 && W > 0 && x.channels > 0 && y.channels > 0, "dn_conv2d: bad shape N=%d H=%d W=%d", N, H, W);
  DN_CHECK_ARG(dtype == DN_BF16 || dtype == DN_F32, "dn_conv2d: bad dtype %d", dtype);
  const int xp = dn_planes(x.channels, dtype), yp = dn_planes(y.channels, dtype);
  DN_CHECK_ARG(x.plane0 >= 0 && x.plane0 + xp <= x.planes_total, "dn_conv2d: x planes [%d,%d) outside buffer of %d",
               x.plane0, x.plane0 + xp, x.planes_total);
  DN_CHECK_ARG(y.plane0 >= 0 && y.plane0 + yp <= y.planes_total, "dn_conv2d: y planes [%d,%d) outside buffer of %d",
               y.plane0, y.plane0 + yp, y.planes_total);
  ConvParams p{};
  p.x = x.base; p.x_planes_total = x.planes_total; p.x_plane0 = x.plane0; p.x_nplanes = xp; p.x_gplanes = 1 << 30; p.x_wmul = 1;
  p.w = packed_w; p.bias = bias;
  p.y = y.base; p.y_planes_total = y.planes_total; p.y_plane0 = y.plane0; p.y_nplanes = yp; p.y_gplanes = 1 << 30; p.y_wmul = 1;
  p.N = N; p.H = H; p.W = W; p.nchunk = xp; p.cout_pad = dn_channel_pad(y.channels); p.relu = relu;
  if (pooled) { p.pool = pooled->base; p.pool_planes_total = pooled->planes_total; p.pool_plane0 = pooled->plane0; }
  if (bn) {
    p.bn_y = bn->y.base; p.bn_planes_total = bn->y.planes_total; p.bn_plane0 = bn->y.plane0;
    p.bn_scale = bn->scale; p.bn_shift = bn->shift; p.bn_mean = bn->mean; p.bn_invstd = bn->invstd;
  }
  if (stats) {
    p.stats = stats;
    const ConvVariant v = choose_variant(p, kh, kw, dtype);
    const long tiles = (long)N * ((H + v.ht - 1) / v.ht) * (v.wave8 ? W / 256 : (W + 127) / 128);   // one partial per block tile
    DN_CHECK_ARG(tiles * p.cout_pad * 2 <= stats_capacity, "dn_conv2d_stats: partial buffer of %ld floats, need %ld", stats_capacity,
                 tiles * p.cout_pad * 2);
    *ntiles = (int)tiles;
  }
  return run_igemm(p, kh, kw, dtype, (hipStream_t)stream);
}


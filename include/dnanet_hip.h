/* dnanet_hip.h -- C ABI of libdnanet_hip.so: the MI355X (gfx950) kernels behind DNANet's U-Net
 * peak-segmentation hot path.
 *
 * The reference (NetherlandsForensicInstitute/DNANet) is pure Python on torch.nn; its "FFI" for
 * this path is the set of torch.nn / torch.optim operators it calls.  Each entry point below
 * replaces one of those operator calls and cites it (paths relative to the reference root).
 * Signatures are plain C: device pointers, sizes, an opaque hipStream_t (void*).  No entry point
 * allocates or frees device memory; workspaces are passed in.  All return 0 on success, non-zero
 * on error (dn_last_error() gives the text).  Kernels are launched on the given stream and the
 * call returns without synchronising.
 *
 * Activation layout ("plane-record", internal to the library, never escapes to the caller's API):
 *   buffer[N][planes_total][H][W][32 bytes], one 32-byte record = 16 bf16 or 8 fp32 channels.
 *   A logical tensor of C channels occupies planes [plane0, plane0 + ceil(C/epr)) of a buffer; the
 *   skip-concat of DecoderBlock (unet_architecture.py:62) is two tensors in one buffer (zero copy).
 *   Channels past C inside the last plane are kept at zero by every kernel.
 * Per-channel vectors (bias, scale, shift, mean ...) are fp32 arrays of dn_channel_pad(C) floats,
 * zero past C.
 */
#ifndef DNANET_HIP_H
#define DNANET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DN_DTYPE_F32 0   /* fp32 storage, v_mfma_f32_32x32x2_f32 */
#define DN_DTYPE_BF16 1  /* bf16 storage, v_mfma_f32_32x32x16_bf16, fp32 accumulate */

#define DN_PACK_FWD 0          /* Conv2d weight [Cout][Cin][KH][KW] for the forward GEMM */
#define DN_PACK_DGRAD 1        /* same weight, flipped/transposed, for the data gradient */
#define DN_PACK_CONVT_FWD 2    /* ConvTranspose2d weight [Cin][Cout][1][2] as a 1x1 GEMM to 2*Cout */
#define DN_PACK_CONVT_DGRAD 3  /* its data gradient: 1x1 GEMM from the two output columns to Cin */
/* fp32 side images, only through dn_pack_batched: */
#define DN_PACK_VEC 4          /* a [cout] vector (bias, head weight) zero-padded to dn_channel_pad(cout) floats */
#define DN_PACK_UP_BIAS 5      /* ConvTranspose2d bias [cout] -> [2][dn_channel_pad(cout)], repeated per output column */
#define DN_PACK_FIRST 6        /* first conv weight [cout][1][KH][KW] -> [KH*KW][dn_channel_pad(cout)] floats */

typedef struct dn_tensor {
  void* base;        /* device pointer to the buffer */
  int planes_total;  /* planes per image in the buffer */
  int plane0;        /* first plane of this tensor */
  int channels;      /* logical channel count C */
} dn_tensor;

/* A bf16 tensor kept as hi + lo (bf16x3 inference mode): value = hi + lo with hi = bf16(v), lo = bf16(v - hi), i.e. ~16
 * mantissa bits.  The lo part sits `lo_off` planes after the hi part in the same buffer (lo_off >= the tensor's planes, so
 * that a concat buffer can hold [up_hi | skip_hi | up_lo | skip_lo]); lo_off = 0 marks a plain bf16 tensor. */
typedef struct dn_tensor_x3 {
  void* base;
  int planes_total;
  int plane0;
  int channels;
  int lo_off;
} dn_tensor_x3;

/* return codes (common.h has the same values under DN_OK / DN_ERR_*) */
#define DN_STATUS_OK 0
#define DN_STATUS_BAD_ARGUMENT 1
#define DN_STATUS_UNSUPPORTED 2   /* no kernel built for this shape / kernel size / mode */
#define DN_STATUS_HIP_ERROR 3

const char* dn_last_error(void);
int dn_abi_version(void);   /* 4 since the *_bn_sums entry points were added; entry points are only ever added */
static inline int dn_channel_pad(int c) { return (c + 31) / 32 * 32; }
static inline int dn_planes(int c, int dtype) { int e = dtype == DN_DTYPE_BF16 ? 16 : 8; return (c + e - 1) / e; }

/* ---- parameter staging ------------------------------------------------------------------- */
/* Bytes of the packed image of a weight (see DN_PACK_*). */
long dn_packed_weight_bytes(int cout, int cin, int kh, int kw, int mode, int dtype);
/* Pack a canonical fp32 parameter (state_dict layout, unet_architecture.py:14-20,54-57) into the
 * GEMM image; `scale` (nullable, [cout]) folds eval-mode BatchNorm (unet_architecture.py:17,21):
 * w' = w * gamma / sqrt(running_var + eps). */
int dn_pack_weight(const float* w, const float* scale, void* packed, int cout, int cin, int kh, int kw, int mode,
                   int dtype, void* stream);

/* All the images a training step needs, re-packed by ONE launch (the weights change every step; 40+ separate launches
 * cost more than the packing at small batches).  jobs: device copy of an array prepared by dn_pack_jobs_prepare,
 * which fills the derived fields (taps = kh*kw goes in; rows_pad, nchunk, gsize, block0 come out) and returns the
 * grid size, or -1 on a bad mode.  Pointers are device pointers and must stay valid for every later launch. */
typedef struct dn_pack_job {
  const float* w;        /* canonical fp32 parameter */
  const float* scale;    /* nullable, [cout]: folded BatchNorm scale (DN_PACK_FWD, DN_PACK_FIRST) */
  void* out;             /* packed image (dtype for modes 0-3, fp32 for modes 4-6) */
  int cout, cin, taps, mode;
  int rows_pad, nchunk, gsize, block0;
} dn_pack_job;
long dn_pack_jobs_prepare(dn_pack_job* jobs_host, int njobs, int dtype);
int dn_pack_batched(const dn_pack_job* jobs_device, int njobs, long grid_blocks, int dtype, void* stream);

/* Name of the kernel instantiation dn_conv2d dispatches a (cin -> cout, N x H x W, kh x kw) convolution to, e.g.
 * "conv_igemm8<bf16,3x5,ht5,cot64>".  No counterpart in the reference (torch picks its cuDNN / MIOpen algorithm
 * internally, unet_architecture.py:14-16 only names nn.Conv2d); bench.py groups its per-launch timings by it. */
int dn_conv2d_variant(int cin, int cout, int N, int H, int W, int kh, int kw, int dtype, char* buf, int buf_len);

/* ---- forward operators -------------------------------------------------------------------- */
/* nn.Conv2d(1, cout, (kh,kw), padding=k//2) on the raw fp32 input x[N][1][H][W]
 * (unet_architecture.py:14-16 for encoders.0; trainable_unet.py:70-79 for the input).
 * wf: [kh*kw][dn_channel_pad(cout)] fp32 (BN scale already folded in for inference), bias likewise. */
int dn_conv_first(const float* x, const float* wf, const float* bias, dn_tensor y, int N, int H, int W, int cout, int kh,
                  int kw, int relu, int dtype, void* stream);
/* dn_conv_first without ReLU plus the BatchNorm batch statistics of its output, as dn_conv2d_stats (training forward of the
 * first DoubleConv unit). */
int dn_conv_first_stats(const float* x, const float* wf, const float* bias, dn_tensor y, int N, int H, int W, int cout, int kh,
                        int kw, int dtype, float* partial, long partial_floats, int* ntiles, void* stream);
/* nn.Conv2d(cin, cout, (kh,kw), padding=k//2) [+ folded BatchNorm2d + ReLU when relu=1]
 * (unet_architecture.py:14-21): y = act(conv(x, packed_w) + bias).  packed_w from DN_PACK_FWD, or
 * DN_PACK_DGRAD to compute the data gradient (then x = dy, cin/cout swapped by the caller). */
int dn_conv2d(dn_tensor x, const void* packed_w, const float* bias, dn_tensor y, int N, int H, int W, int kh, int kw,
              int relu, int dtype, void* stream);
/* nn.ConvTranspose2d(cin, cout, (1,2), stride (1,2)) (unet_architecture.py:54-57):
 * y[n,co,h,2w+k] = bias[co] + sum_ci x[n,ci,h,w] * W[ci,co,0,k].  x is N x H x Win, y is N x H x 2*Win.
 * packed_w from DN_PACK_CONVT_FWD; bias2: [2*dn_channel_pad(cout)] = bias repeated per k. */
int dn_conv_transpose_1x2(dn_tensor x, const void* packed_w, const float* bias2, dn_tensor y, int N, int H, int Win,
                          int dtype, void* stream);
/* Data gradient of the above: dx[n,ci,h,w] = sum_{k,co} dy[n,co,h,2w+k] * W[ci,co,0,k]; packed_w from
 * DN_PACK_CONVT_DGRAD. */
int dn_conv_transpose_1x2_dgrad(dn_tensor dy, const void* packed_w, dn_tensor dx, int N, int H, int Win, int dtype,
                                void* stream);
/* dn_conv_transpose_1x2_dgrad together with the reduction pass of dn_bn_train_backward for the unit whose activation the
 * transposed convolution reads (the second unit of the block below): as dn_conv2d_dgrad_bn_sums, dx being that unit's
 * activation gradient.  sums[*ntiles][dn_channel_pad(C)][2]. */
int dn_conv_transpose_1x2_dgrad_bn_sums(dn_tensor dy, const void* packed_w, dn_tensor dx, dn_tensor bn_y, const float* scale,
                                        const float* shift, const float* mean, const float* invstd, int N, int H, int Win, int dtype,
                                        float* sums, long sums_floats, int* ntiles, void* stream);
/* ---- bf16x3: the same eval-mode operators on hi + lo split bf16 tensors and weights ----------------
 * fp32-grade results from the bf16 matrix cores: a product w*x is evaluated as w_hi*x_hi + w_hi*x_lo + w_lo*x_hi (three
 * bf16 MFMAs, fp32 accumulation; the dropped w_lo*x_lo term is 2^-18 relative), and every activation is written back as
 * hi + lo by the epilogue.  Replaces the same torch operators as dn_conv2d / dn_conv2d_pool / dn_conv2d_head /
 * dn_conv_first / dn_conv_transpose_1x2 above (unet_architecture.py:14-21,37,54-62,94-97) for callers that need the
 * reference's fp32 masks at more than the fp32 matrix rate.  Storage is bf16; there is no dtype argument.
 * packed_w is an ordinary DN_PACK_FWD (DN_PACK_CONVT_FWD) image over the VIRTUAL input channels, cin padded to whole planes:
 *   split input (x.lo_off > 0): [W_hi | W_hi | W_lo]  against the K walk [x_hi | x_lo | x_hi]
 *   plain input (x.lo_off = 0): [W_hi | W_lo]         against          [x    | x]
 * (BatchNorm folded into W before the split).  Outputs are always split (y.lo_off >= planes of y); `pooled` may be plain. */
int dn_conv2d_x3(dn_tensor_x3 x, const void* packed_w, const float* bias, dn_tensor_x3 y, const dn_tensor_x3* pooled, int N, int H,
                 int W, int kh, int kw, int relu, void* stream);
/* last DoubleConv convolution + 1x1 head + sigmoid on a split / plain input; the head sees the fp32 activation */
int dn_conv2d_head_x3(dn_tensor_x3 x, const void* packed_w, const float* bias, int cout, const float* head_w, const float* head_b,
                      float* logits, float* probs, int N, int H, int W, int kh, int kw, int relu, void* stream);
/* first convolution (fp32 input, fp32 matrix cores as in dn_conv_first) with a split output */
int dn_conv_first_x3(const float* x, const float* wf, const float* bias, dn_tensor_x3 y, int N, int H, int W, int cout, int kh,
                     int kw, int relu, void* stream);
int dn_conv_transpose_1x2_x3(dn_tensor_x3 x, const void* packed_w, const float* bias2, dn_tensor_x3 y, int N, int H, int Win,
                             void* stream);
/* dn_head_1x1 on a split tensor (shapes whose last convolution has no fused-head tile) */
int dn_head_1x1_x3(dn_tensor_x3 x, const float* w, const float* b, float* logits, float* probs, int N, int H, int W, void* stream);

/* nn.MaxPool2d((1,2)) (unet_architecture.py:37): y[n,c,h,w] = max(x[n,c,h,2w], x[n,c,h,2w+1]). */
int dn_maxpool_1x2(dn_tensor x, dn_tensor y, int N, int H, int Wout, int dtype, void* stream);
/* nn.Conv2d(C, 1, 1) head (unet_architecture.py:94-97) and torch.sigmoid (trainable_unet.py:409):
 * logits/probs are [N][1][H][W] fp32, either may be null. w: [dn_channel_pad(C)], b: [1]. */
int dn_head_1x1(dn_tensor x, const float* w, const float* b, float* logits, float* probs, int N, int H, int W, int dtype,
                void* stream);

/* ---- training operators ---------------------------------------------------------------------
 * Reduction workspaces: `partial` must hold dn_partial_floats(C) floats. */
static inline long dn_partial_floats(int c) { return 512L * dn_channel_pad(c) * 2 + 4096; }
int dn_reduce_blocks(long pixels);

/* nn.BatchNorm2d in training mode + ReLU (unet_architecture.py:17,21): batch mean / biased variance
 * over (N,H,W) of y, running-stat update (momentum, unbiased variance; running_* may be null),
 * a = relu(gamma * (y - mean) * invstd + beta).  Saves mean/invstd and the fused scale/shift
 * (dn_channel_pad(C) floats each) for the backward pass. */
int dn_bn_train_forward(dn_tensor y, dn_tensor a, int N, int H, int W, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                        float* scale, float* shift, float* partial, int dtype, void* stream);
/* dn_bn_train_forward for callers that cannot allocate (code under HIP graph capture, tight serving / training loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE: same operator, same results bit for bit, but the
 * reduction partials live in a 2.02 MiB array inside the library (one copy per GPU, held until the process ends) instead
 * of a caller-provided workspace; the call itself allocates nothing.  Limit: y.channels <= DN_NOALLOC_MAX_CHANNELS (512,
 * the widest layer of the shipped net), otherwise DN_STATUS_BAD_ARGUMENT.  All calls on one GPU share that array: calls
 * that overlap on one GPU (different streams or threads) race; calls on one stream are ordered and safe.  The FIRST call
 * on a device looks up that device's address of the array (hipGetSymbolAddress under a mutex: a lookup, not an
 * allocation): make one call per device before a stream capture starts, and only then capture
 * (tests/test_zz_no_alloc.py does exactly that). */
#define DN_NOALLOC_MAX_CHANNELS 512
int dn_bn_train_forward_noalloc(dn_tensor y, dn_tensor a, int N, int H, int W, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                                float* scale, float* shift, int dtype, void* stream);
/* dn_conv2d followed by nn.MaxPool2d((1,2)) (EncoderBlock, unet_architecture.py:28-42: returns (skip, pooled)) in one
 * kernel: y gets the convolution output (the skip tensor), pooled[n,c,h,w] = max(y[n,c,h,2w], y[n,c,h,2w+1]) with W/2
 * columns.  Kernel sizes (3,5) and (1,3); W even. */
int dn_conv2d_pool(dn_tensor x, const void* packed_w, const float* bias, dn_tensor y, dn_tensor pooled, int N, int H, int W,
                   int kh, int kw, int relu, int dtype, void* stream);
/* The last DoubleConv convolution with the output head fused: nn.Conv2d(C, 1, 1) (unet_architecture.py:94-97) and
 * torch.sigmoid (trainable_unet.py:409) are applied to the block's channels in the epilogue; the C-channel activation is
 * never written.  bias / packed_w as for dn_conv2d (eval-mode BatchNorm folded), head_w: [dn_channel_pad(cout)] zero
 * padded, head_b: [1]; logits / probs: [N][1][H][W] fp32, either may be null.  Needs the kernel tile to span all output
 * channels (cout <= 64 and a shape the dispatcher gives such a tile): dn_conv2d_head_supported says so beforehand,
 * otherwise DN_STATUS_UNSUPPORTED and the caller runs dn_conv2d + dn_head_1x1. */
int dn_conv2d_head_supported(int cin, int cout, int N, int H, int W, int kh, int kw, int dtype);
int dn_conv2d_head(dn_tensor x, const void* packed_w, const float* bias, int cout, const float* head_w, const float* head_b,
                   float* logits, float* probs, int N, int H, int W, int kh, int kw, int relu, int dtype, void* stream);
/* The convolution of a DoubleConv in training mode (nn.Conv2d + bias feeding nn.BatchNorm2d, unet_architecture.py:14-17)
 * with the BatchNorm batch statistics taken in the convolution's epilogue: y as dn_conv2d(relu = 0), and per pixel tile
 * and channel the sum and sum of squares of the values as stored, partial[tile][dn_channel_pad(cout)][2]
 * (partial_floats = capacity of the buffer, *ntiles = tiles written).  Kernel sizes (3,5) and (1,3). */
int dn_conv2d_stats(dn_tensor x, const void* packed_w, const float* bias, dn_tensor y, int N, int H, int W, int kh, int kw,
                    int dtype, float* partial, long partial_floats, int* ntiles, void* stream);
/* dn_bn_train_forward without its statistics pass: partial / ntiles come from dn_conv2d_stats. */
int dn_bn_train_apply(dn_tensor y, dn_tensor a, int N, int H, int W, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                      float* scale, float* shift, const float* partial, int ntiles, int dtype, void* stream);
/* dn_bn_train_apply for the second unit of an encoder block: also writes MaxPool2d((1,2)) of the activation
 * (unet_architecture.py:37) into `pooled` (W/2 columns), saving the pooling pass over a. */
int dn_bn_train_apply_pool(dn_tensor y, dn_tensor a, dn_tensor pooled, int N, int H, int W, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                           float* scale, float* shift, const float* partial, int ntiles, int dtype, void* stream);
/* Backward of ReLU + BatchNorm2d (autograd of the above, trainable_unet.py:342): on entry g = dL/da,
 * on exit g = dL/dy (in place); dgamma/dbeta are written ([C]); coef: 3*dn_channel_pad(C) scratch.
 * scale/shift/mean/invstd are the vectors dn_bn_train_forward saved: the ReLU mask is recomputed as
 * y*scale + shift > 0, so the activation tensor is not read. */
int dn_bn_train_backward(dn_tensor g, dn_tensor y, const float* scale, const float* shift, int N, int H, int W,
                         const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef,
                         float* partial, int dtype, void* stream);
/* The data gradient of a DoubleConv's second convolution (autograd of nn.Conv2d, unet_architecture.py:14-20, under
 * trainable_unet.py:342) together with the reduction pass of dn_bn_train_backward for the unit that produced the
 * convolution's input: dx = dL/d(activation of that unit) as dn_conv2d(dy, packed_wd, relu = 0) writes it, and per pixel
 * tile and channel the two sums { sum dz, sum dz * xhat } over the values of dx as stored, dz = dx * (bn_y * scale +
 * shift > 0), xhat = (bn_y - mean) * invstd -- sums[tile][dn_channel_pad(C)][2], *ntiles tiles.  bn_y: that unit's saved
 * convolution output (the channels and pixel grid of dx), scale / shift / mean / invstd: the vectors its forward saved
 * ([dn_channel_pad(C)]).  dx is read back one time less.  Kernel size (3,5). */
int dn_conv2d_dgrad_bn_sums(dn_tensor dy, const void* packed_wd, dn_tensor dx, dn_tensor bn_y, const float* scale,
                            const float* shift, const float* mean, const float* invstd, int N, int H, int W, int kh, int kw,
                            int dtype, float* sums, long sums_floats, int* ntiles, void* stream);
/* dn_bn_train_backward without its reduction pass: sums / ntiles come from dn_conv2d_dgrad_bn_sums. */
int dn_bn_train_backward_sums(dn_tensor g, dn_tensor y, const float* scale, const float* shift, int N, int H, int W,
                              const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef,
                              const float* sums, int ntiles, int dtype, void* stream);
/* ---- synchronised BatchNorm (optional, data parallel; NO COUNTERPART IN THE REFERENCE, which is single-device: this is what
 * makes "R ranks x N/R profiles" compute the batch statistics -- hence the results -- of one device at batch N, the only
 * oracle a data-parallel run has).  Forward: dn_conv2d_stats / dn_conv_first_stats -> dn_bn_partial_totals ->
 * the caller sums `totals` over the ranks (2 * dn_channel_pad(C) doubles) -> dn_bn_train_apply_totals with the global
 * count.  Backward: dn_bn_train_backward_totals -> sum over the ranks -> dn_bn_train_backward_apply_totals.  With one
 * rank the results are bitwise those of dn_bn_train_apply / dn_bn_train_backward. */
/* totals[c][2] = fixed-order fp64 sums of the [ntiles][dn_channel_pad(C)][2] partials (zero for c >= C). */
int dn_bn_partial_totals(const float* partial, int ntiles, int C, double* totals, void* stream);
/* dn_bn_train_apply(_pool) from summed totals: mean / biased variance over `count` values (the GLOBAL batch), running-stat
 * update with the unbiased variance of that count; pooled.base may be null. */
int dn_bn_train_apply_totals(dn_tensor y, dn_tensor a, dn_tensor pooled, int N, int H, int W, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                             float* scale, float* shift, const double* totals, double count, int dtype, void* stream);
/* First half of dn_bn_train_backward: totals[c][2] = { sum dz, sum dz * xhat } of this rank (g is not modified). */
int dn_bn_train_backward_totals(dn_tensor g, dn_tensor y, const float* scale, const float* shift, int N, int H, int W,
                                const float* mean, const float* invstd, float* partial, double* totals, int dtype, void* stream);
/* dn_bn_train_backward_totals without its reduction pass (sums / ntiles from dn_conv2d_dgrad_bn_sums). */
int dn_bn_backward_sums_totals(const float* sums, int ntiles, int C, double* totals, void* stream);
/* Second half: g -> dL/dy with the batch means taken from global_totals / count; dgamma / dbeta are this rank's sums
 * (local_totals), which the data-parallel gradient all-reduce adds up like every other parameter gradient. */
int dn_bn_train_backward_apply_totals(dn_tensor g, dn_tensor y, const float* scale, const float* shift, int N, int H, int W,
                                      const float* gamma, const float* mean, const float* invstd, const double* local_totals,
                                      const double* global_totals, double count, float* dgamma, float* dbeta, float* coef,
                                      int dtype, void* stream);
/* Backward of nn.MaxPool2d((1,2)) (unet_architecture.py:37) fused with the skip-connection add:
 * g = dskip (nullable .base) + route(dpool) to the first maximal element of every pair of x. */
int dn_maxpool_1x2_backward(dn_tensor x, dn_tensor dpool, dn_tensor dskip, dn_tensor g, int N, int H, int Wout, int dtype,
                            void* stream);
/* Backward of the 1x1 head (unet_architecture.py:94-97): g = dlogits * w, dw[c] = sum dlogits * x[c],
 * db = sum dlogits. */
int dn_head_1x1_backward(const float* dlogits, dn_tensor x, const float* w, dn_tensor g, float* dw, float* db, float* partial,
                         int N, int H, int W, int dtype, void* stream);
/* dn_maxpool_1x2_backward / dn_head_1x1_backward together with the reduction pass of dn_bn_train_backward for the unit
 * whose activation gradient g is (the second unit of an encoder block / of the last decoder block): besides g, per block
 * and channel { sum dz, sum dz * xhat } over the values of g as stored -- sums[*nblk][dn_channel_pad(C)][2], at most
 * dn_reduce_blocks(pixels) blocks; bn_y / scale / shift / mean / invstd as in dn_conv2d_dgrad_bn_sums.  Followed by
 * dn_bn_train_backward_sums (or dn_bn_backward_sums_totals), g is read back one time less. */
int dn_maxpool_1x2_backward_bn_sums(dn_tensor x, dn_tensor dpool, dn_tensor dskip, dn_tensor g, dn_tensor bn_y, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, int N, int H, int Wout, int dtype,
                                    float* sums, long sums_floats, int* nblk, void* stream);
int dn_head_1x1_backward_bn_sums(const float* dlogits, dn_tensor x, const float* w, dn_tensor g, float* dw, float* db, float* partial,
                                 dn_tensor bn_y, const float* scale, const float* shift, const float* mean, const float* invstd,
                                 int N, int H, int W, int dtype, float* sums, long sums_floats, int* nblk, void* stream);
/* out[c] = sum over (N,H,W) of x[c]: bias gradient of nn.ConvTranspose2d (unet_architecture.py:54-57). */
int dn_channel_sum(dn_tensor x, float* out, float* partial, int N, int H, int W, int dtype, void* stream);
/* Weight gradient of nn.Conv2d(cin, cout, (kh,kw), padding=k//2): dw[cout][cin][kh][kw] (canonical fp32, overwritten).
 * a = the convolution's input (concat_split > 0: it is the skip-concat of two tensors of that many channels);
 * workspace holds the split-K partial slabs (>= 64 MiB recommended; more = more K parallelism). */
int dn_conv2d_wgrad(dn_tensor dy, dn_tensor a, float* dw, void* workspace, long workspace_bytes, int N, int H, int W, int kh,
                    int kw, int cin, int concat_split, int dtype, void* stream);
/* Weight gradient of nn.ConvTranspose2d(cin, cout, (1,2), stride (1,2)): dw[cin][cout][1][2]. dy is N x H x 2*Win. */
int dn_conv_transpose_1x2_wgrad(dn_tensor dy, dn_tensor x, float* dw, void* workspace, long workspace_bytes, int N, int H,
                                int Win, int dtype, void* stream);
/* Weight gradient of the first convolution (Cin = 1) against the raw fp32 input x[N][1][H][W]: dw[cout][1][kh][kw].
 * partial: dn_conv_first_wgrad_blocks(N,H,W) * kh*kw * dn_channel_pad(cout) floats (at most 512 blocks);
 * partial_floats = capacity of that buffer, checked. */
int dn_conv_first_wgrad_blocks(int N, int H, int W);
int dn_conv_first_wgrad(const float* x, dn_tensor dy, float* dw, float* partial, long partial_floats, int N, int H, int W, int kh,
                        int kw, int dtype, void* stream);
/* DiceLoss.forward (loss.py:9-20) + BinaryAccuracy counts (trainable_unet.py:411-415) in one pass over the
 * whole batch: sums[0..4] = { sum p*t, sum p, sum t, #(p>=0.5)==t, loss }.  partial: 4096 floats. */
int dn_dice_forward(const float* logits, const signed char* targets, long n, float* sums, float* partial, void* stream);
/* dL/dlogits for the Dice loss given (possibly all-reduced) sums[0..2]. */
int dn_dice_backward(const float* logits, const signed char* targets, long n, const float* sums, float* dlogits, void* stream);
/* torch.optim.Adam step with coupled weight decay over a flat fp32 arena (trainable_unet.py:155-157,343),
 * grads are scaled by grad_scale first and zeroed afterwards when zero_grad != 0 (optimizer.zero_grad(), :344). */
int dn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, long step, float grad_scale, int zero_grad, void* stream);
/* The same Adam step for a caller that replays a captured launch sequence (HIP graph): everything that changes from step
 * to step lives in device memory.  *step (device) is incremented, hyper[0] (device) is the learning rate the host writes
 * when the ExponentialLR schedule changes it (trainable_unet.py:159-162,305-311); hyper[1..2] receive the bias-correction
 * scalars.  beta / eps / weight_decay are launch constants. */
int dn_adam_step_graph(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, float* hyper, long long* step,
                       float beta1, float beta2, float eps, float weight_decay, float grad_scale, int zero_grad, void* stream);
/* int16 RFU -> fp32 (get_input, trainable_unet.py:70-72). */
int dn_i16_to_f32(const short* in, float* out, long n, void* stream);

/* ---- allele hand-off -------------------------------------------------------------------------
 * NearestBasePairCaller.translate_pixels_to_alleles (DNAnet/allele_callers.py:63-125) for a batch: probs and rfu are
 * (N,H,W) fp32 (probabilities, raw RFU); scaler holds the base pair of every scan column (W doubles; one row per
 * profile, scaler_stride apart, or shared when the stride is 0).  The panel is given per dye row h as the slice
 * [table_off[h], table_off[h+1]) of table_bp (base pairs, ascending) and table_order (position in the panel, the
 * tie-break).  Per profile, dye and table entry j < amax: out_rfu = highest RFU of the runs called as j (>= 0),
 * out_first = first scan column that called j, or DN_ALLELE_ABSENT. */
#define DN_ALLELE_ABSENT 0x7f7f7f7f
int dn_call_alleles(const float* probs, const float* rfu, const double* scaler, long scaler_stride, const double* table_bp,
                    const int* table_order, const int* table_off, int N, int H, int W, int amax, int* out_rfu, int* out_first,
                    void* stream);

/* ---- ingest ------------------------------------------------------------------------------------
 * rescale_dye (DNAnet/data/utils.py:239-279) + HIDImage._rescale_profile / the scaler (hid_image.py:111,168-183) for a
 * batch: raw [n][rows][length] fp32 RFU, basepairs [n][length] float64 = get_interpolated_basepairs of each size standard
 * (0 outside the ladder), span [n][2] = first / last scan of the ladder, target [n_out] float64 = linspace(65, 475, 4096).
 * out [n][1][rows][n_out] fp32 (the network's input layout), scaler [n][n_out] float64, idx_out (nullable) [n][n_out] the
 * chosen raw scan per output scan. */
int dn_rescale_profiles(const float* raw, const double* basepairs, const int* span, const double* target, int n_profiles,
                        int rows, int length, int n_out, float* out, double* scaler, int* idx_out, void* stream);

/* Pixel metrics (DNAnet/evaluation/segmentation/pixel_metrics.py:9-74) from the probability maps still in HBM: per
 * profile counts[i][0..2] = { tp, fp, fn } with "predicted positive" = p >= thr_ge (pixel_precision / recall / F1,
 * :17-36), counts[i][3..5] the same with p > thr_gt (average_binary_iou, :65), counts[i][6] = truth values that are
 * neither 0 nor 1 (they count for neither class; the IoU of such a mask is not given by these counts).  truth: int8.
 * probs, truth: [n_images][pixels]; counts: [n_images][7]. */
int dn_pixel_confusion(const float* probs, const signed char* truth, int n_images, long pixels, float thr_ge, float thr_gt,
                       int* counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DNANET_HIP_H */

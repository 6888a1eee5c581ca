// internal: public C ABI + helpers shared by the translation units of libdnanet_hip.so
#pragma once
#include <algorithm>
#include "dnanet_hip.h"
#include "conv_igemm.h"

int conv_igemm_run_bf16_k35(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_bf16_k13(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_bf16_k11(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_bf16_k35_split(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_bf16_k13_split(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_bf16_k11_split(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_f32_k35(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_bf16_k35_bnred(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_f32_k35_bnred(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_bf16_k11_bnred(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_f32_k11_bnred(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_f32_k13(const ConvParams& p, int ht, int cot, hipStream_t s);
int conv_igemm_run_f32_k11(const ConvParams& p, int ht, int cot, hipStream_t s);

#include "conv_igemm8.h"
int conv_igemm8_run_bf16(const ConvParams& p, int ht, hipStream_t s);
int conv_igemm8_run_bf16_split(const ConvParams& p, int ht, hipStream_t s);
int conv_igemm8_run_f32(const ConvParams& p, int ht, hipStream_t s);
int conv_igemm8_run_bf16_bnred(const ConvParams& p, int ht, hipStream_t s);
int conv_igemm8_run_f32_bnred(const ConvParams& p, int ht, hipStream_t s);

#include "conv_wgrad.h"
int conv_wgrad_run_bf16_k35(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s);
int conv_wgrad_run_bf16_k13(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s);
int conv_wgrad_run_bf16_k11(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s);
int conv_wgrad_run_f32_k35(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s);
int conv_wgrad_run_f32_k13(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s);
int conv_wgrad_run_f32_k11(const WgradParams& p, long slab_bytes, int* nslab, hipStream_t s);
#include "convt_wgrad.h"
int convt_wgrad_run_bf16(ConvtWgradParams& p, int cout, long slab_bytes, int* nslab, hipStream_t s);
int convt_wgrad_run_f32(ConvtWgradParams& p, int cout, long slab_bytes, int* nslab, hipStream_t s);
int conv_first_mfma_run(const float* x, const float* wf, const ConvParams& p, int kh, int kw, int dtype, hipStream_t s);
int conv_first_row_tile(int H);

// Shared definitions for the gfx950 kernels of the DNANet U-Net hot path.
//
// Activation layout in HBM ("plane-record" layout, never visible outside the library):
//   [N][planes][H][W][record]  with a 32-byte record = 16 bf16 channels or 8 fp32 channels.
// One plane holds one K-chunk of every pixel, so an implicit-GEMM K-step stages whole, contiguous
// image-row segments; channel concatenation is plane concatenation (zero-copy skip connections).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#define DN_OK 0
#define DN_ERR_ARG 1
#define DN_ERR_UNSUPPORTED 2
#define DN_ERR_HIP 3

#define DN_F32 0    // == DN_DTYPE_F32
#define DN_BF16 1   // == DN_DTYPE_BF16

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int DN_REC_BYTES = 32;   // bytes per pixel record in a plane

template <typename T> struct RecTraits;
template <> struct RecTraits<float>  { static constexpr int EPR = 8;  static constexpr int EPP = 4; };  // elems per record / per 16-B piece
template <> struct RecTraits<__bf16> { static constexpr int EPR = 16; static constexpr int EPP = 8; };

// A logical activation tensor inside a plane-record buffer.
struct TensorRef {
  void* base;          // buffer base
  int planes_total;    // planes per image in the buffer (image stride = planes_total*H*Wrow records)
  int plane0;          // first plane of this tensor
  int nplanes;         // planes of this tensor (ceil(C / EPR))
  int H;               // rows
  int Wrow;            // records per row in the buffer
};

int dn_set_error(int code, const char* fmt, ...);

// Tuning aids (DN_TUNE_* environment variables, A/B switches for the measurements in DESIGN.md).  They are read ONCE,
// into this struct, when the library is loaded (api.hip calls dn_tune() from a constructor); the dispatch code only
// looks at the struct.  All zero = the shipped behaviour.
struct DnTune {
  int cot;              // DN_TUNE_COT=32: force the narrow output-channel tile
  bool no_cot128;       // DN_TUNE_NO_COT128: no 128-channel tile for the 1x1 GEMMs
  bool no_conv8;        // DN_TUNE_NO_CONV8: 4-wave kernel everywhere
  bool conv8_ht3;       // DN_TUNE_CONV8_HT3: 8-wave kernel for bf16 3-row tiles too
  bool stats_4wave;     // DN_TUNE_STATS_4WAVE: fused-statistics convolutions on the 4-wave kernel
  bool k11_kp1;         // DN_TUNE_K11_KP1: 1x1 GEMM stages one plane per barrier
  bool no_head_fuse;    // DN_TUNE_NO_HEAD_FUSE: dn_conv2d_head_supported() says no
  long wgrad_blocks;    // DN_TUNE_WGRAD_BLOCKS: split-K target grid size (0 = default 512)
  int wgrad_noskip;     // DN_TUNE_WGRAD_NOSKIP: 1 = multiply the zero halo rows too, 2 = interleaved tap map
  bool wgrad_no_cot32;  // DN_TUNE_WGRAD_NO_COT32: 64-wide output-channel tile also for layers with <= 32 output channels
  bool wgrad_no_xcd;    // DN_TUNE_WGRAD_NO_XCD: weight-gradient blocks in plain (tile-fast) order, splits not tied to XCDs
  long wgrad_minpx;     // DN_TUNE_WGRAD_MINPX: fewest pixels a split-K block of the 3x5 weight gradient may get (0 = default)
  bool no_convt_gemm;   // DN_TUNE_NO_CONVT_GEMM: transposed-convolution weight gradient on conv_wgrad.h's 1x1 kernel (the round-2 path)
};
inline const DnTune& dn_tune() {
  static const DnTune t = [] {
    DnTune v{};
    auto flag = [](const char* name) { const char* e = getenv(name); return e && *e && *e != '0'; };
    const char* e;
    if ((e = getenv("DN_TUNE_COT"))) v.cot = atoi(e);
    v.no_cot128 = flag("DN_TUNE_NO_COT128");
    v.no_conv8 = flag("DN_TUNE_NO_CONV8");
    v.conv8_ht3 = flag("DN_TUNE_CONV8_HT3");
    v.stats_4wave = flag("DN_TUNE_STATS_4WAVE");
    v.k11_kp1 = flag("DN_TUNE_K11_KP1");
    v.no_head_fuse = flag("DN_TUNE_NO_HEAD_FUSE");
    if ((e = getenv("DN_TUNE_WGRAD_BLOCKS"))) v.wgrad_blocks = atol(e);
    if ((e = getenv("DN_TUNE_WGRAD_NOSKIP"))) v.wgrad_noskip = atoi(e) == 2 ? 2 : 1;
    v.wgrad_no_cot32 = flag("DN_TUNE_WGRAD_NO_COT32");
    v.wgrad_no_xcd = flag("DN_TUNE_WGRAD_NO_XCD");
    v.no_convt_gemm = flag("DN_TUNE_NO_CONVT_GEMM");
    if ((e = getenv("DN_TUNE_WGRAD_MINPX"))) v.wgrad_minpx = atol(e);
    return v;
  }();
  return t;
}
#define DN_CHECK_ARG(cond, ...) do { if (!(cond)) return dn_set_error(DN_ERR_ARG, __VA_ARGS__); } while (0)
#define DN_CHECK_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
  return dn_set_error(DN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
#define DN_LAUNCH_CHECK() DN_CHECK_HIP(hipGetLastError())

// Sum over the 16 lanes of a DPP row, result in every lane of the row: four VALU adds with DPP operands (quad_perm
// [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8) instead of four trips through the LDS crossbar (__shfl_xor = ds_bpermute).
__device__ __forceinline__ float dn_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  return v;
}

// ---- LDS-DMA with buffer addressing (gfx950: buffer_load_dwordx4 ... lds)
// Written as inline asm on purpose: hipcc neither drains nor counts it, so loads stay in flight across the ds_reads of
// the tile being consumed and are waited for by hand (s_waitcnt vmcnt).  Buffer addressing keeps the per-piece VALU
// work at (almost) nothing -- a 64-bit per-lane address plus a zero-page select per 16-B piece costs more issue slots
// than the MFMAs it feeds: address = SRD base (uniform, moved by SALU) + 32-bit per-lane offset, and an offset
// >= num_records (DN_OOB) makes the hardware write zeros, which is how padding, halos and image borders are filled.
// The LDS destination is lane-linear: M0 + 16 * lane, so one wave-instruction fills 1 KiB.
#define DN_OOB 0x80000000u
typedef __attribute__((ext_vector_type(4))) unsigned dn_srd_t;
__device__ __forceinline__ dn_srd_t dn_make_srd(const void* base) {
  const unsigned long long b = (unsigned long long)base;
  dn_srd_t r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);   // stride 0: raw buffer
  r.z = DN_OOB;                                                          // num_records (bytes)
  r.w = 0x00020000u;                                                     // 32-bit untyped data format
  return r;
}
__device__ __forceinline__ void dn_buffer_lds16(unsigned voff, dn_srd_t srd, unsigned lds_addr_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr_uniform) : "memory");
}

__device__ __forceinline__ float dn_to_float(float v) { return v; }
__device__ __forceinline__ float dn_to_float(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T dn_from_float(float v);
template <> __device__ __forceinline__ float dn_from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 dn_from_float<__bf16>(float v) { return (__bf16)v; }

// 16-byte piece <-> floats
template <typename T> struct Piece;
template <> struct Piece<float> {
  static constexpr int N = 4;
  __device__ static void unpack(const uint4& p, float* f) {
    f[0] = __uint_as_float(p.x); f[1] = __uint_as_float(p.y); f[2] = __uint_as_float(p.z); f[3] = __uint_as_float(p.w);
  }
  __device__ static uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct Piece<__bf16> {
  static constexpr int N = 8;
  __device__ static void unpack(const uint4& p, float* f) {
    const uint32_t u[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(u[i] << 16);
      f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
  }
  __device__ static uint4 pack(const float* f) {
    uint32_t u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __bf16 lo = (__bf16)f[2 * i], hi = (__bf16)f[2 * i + 1];
      u[i] = (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
    }
    return make_uint4(u[0], u[1], u[2], u[3]);
  }
};

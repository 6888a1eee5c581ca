// Shared definitions for the gfx950 kernels of the DNANet U-Net hot path.
//
// Activation layout in HBM ("plane-record" layout, never visible outside the library):
//   [N][planes][H][W][record]  with a 32-byte record = 16 bf16 channels or 8 fp32 channels.
// One plane holds one K-chunk of every pixel, so an implicit-GEMM K-step stages whole, contiguous
// image-row segments; channel concatenation is plane concatenation (zero-copy skip connections).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
#define DN_CHECK_ARG(cond, ...) do { if (!(cond)) return dn_set_error(DN_ERR_ARG, __VA_ARGS__); } while (0)
#define DN_CHECK_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
  return dn_set_error(DN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
#define DN_LAUNCH_CHECK() DN_CHECK_HIP(hipGetLastError())

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

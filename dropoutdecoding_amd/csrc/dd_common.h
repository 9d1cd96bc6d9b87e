// Shared host/device helpers for libdropdec (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dropdec.h"

#define DD_WAVE 64

void dd_set_error(const char* fmt, ...);

#define DD_HIP(expr)                                                                            \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) {                                                                    \
      dd_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));       \
      return DD_EHIP;                                                                           \
    }                                                                                           \
  } while (0)

#define DD_CHECK_LAUNCH()                                                                       \
  do {                                                                                          \
    hipError_t e__ = hipGetLastError();                                                         \
    if (e__ != hipSuccess) {                                                                    \
      dd_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e__));   \
      return DD_EHIP;                                                                           \
    }                                                                                           \
  } while (0)

#define DD_REQUIRE(cond, ...)                                                                   \
  do {                                                                                          \
    if (!(cond)) {                                                                              \
      dd_set_error(__VA_ARGS__);                                                                \
      return DD_EINVAL;                                                                         \
    }                                                                                           \
  } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#ifdef __HIPCC__
// fp32 -> bf16 bits, round-to-nearest-even (finite inputs).
__device__ __forceinline__ uint32_t dd_bf16_rn(float x) {
  uint32_t u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float dd_bf16_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// x = hi + lo with hi, lo bf16: carries ~16 mantissa bits through a bf16 MFMA (DESIGN.md "Numerics").
__device__ __forceinline__ void dd_split_hl(float x, uint32_t& hi, uint32_t& lo) {
  hi = dd_bf16_rn(x);
  lo = dd_bf16_rn(x - dd_bf16_to_f32(hi));
}

// Prefill activations are stored in the SAME 16x32 operand-tile order as the weights: element (row m, col k) of a
// [M][K] plane lives at u16 offset (((m/16)*S + k/32)*64 + ((k%32)/8)*16 + m%16)*8 + k%8, S = K/32, so every A
// fragment of the prefill GEMM is one contiguous 1 KiB wave load (guide: fragment-shaped row-major loads cost 18-45 %).
__device__ __forceinline__ size_t apack_off(int m, int k, int S) {
  return ((((size_t)(m >> 4) * S + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (m & 15)) << 3) + (k & 7);
}

__device__ __forceinline__ float dd_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float dd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double dd_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#endif

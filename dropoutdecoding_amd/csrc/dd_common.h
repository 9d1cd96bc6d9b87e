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

// One sequence of a dd_sample_masks_lanes launch (dd_dropout.hip; called by the group step in dd_engine.hip and by dd_tools.hip)
struct MaskLaneArgs {
  const float* epi;
  int L;
  uint8_t* keep;
  const int32_t* argmax;
  const int32_t* topk;
  uint32_t* rng_state;
  uint8_t* drop;
  int32_t* n_drop;
  uint8_t* drop_bits;
  const int32_t* gate;
};
int dd_sample_masks_lanes(const MaskLaneArgs* lanes, int n, int k_top, const double* mprobs, int K, int mode, hipStream_t st);

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 dd_f16x8_t;
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
// Two values at once on gfx950's packed convert: v_cvt_pk_bf16_f32 rounds to nearest even like dd_bf16_rn (finite inputs), so hi / lo are the
// bits dd_split_hl gives — as packed pairs (a in the low half), 6 vector instructions instead of 18 (round 4: the prefill attention spent
// more of its time splitting operands than on the matrix cores).
typedef __bf16 dd_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float dd_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dd_split_hl2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const dd_f32x2_t v = {a, b};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dd_bf16x2_t));
  const dd_f32x2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, dd_bf16x2_t));
}
// The same for engines that keep fp16 weights (weight_format 2: fp16-native checkpoints stay exact — the reference loads
// every model with torch_dtype=float16, chair_test/chair_test.py:189-213): operands of the f16 MFMA.  hi + lo carries
// ~22 mantissa bits; |x| beyond fp16's range saturates (the reference's own fp16 activations live in that range).
typedef _Float16 dd_f16;
__device__ __forceinline__ uint32_t dd_f16_bits(float x) {
  dd_f16 h = (dd_f16)fminf(fmaxf(x, -65504.f), 65504.f);
  return (uint32_t)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float dd_f16_to_f32(uint32_t b) { return (float)__builtin_bit_cast(dd_f16, (unsigned short)b); }
__device__ __forceinline__ void dd_split_hl_f16(float x, uint32_t& hi, uint32_t& lo) {
  hi = dd_f16_bits(x);
  lo = dd_f16_bits(x - dd_f16_to_f32(hi));
}
// wf: the engine's 16-bit weight type (0 bf16, 1 fp16); activations are split in the same type
__device__ __forceinline__ void dd_split(float x, uint32_t& hi, uint32_t& lo, int wf) {
  // x becomes an opaque register value first: hipcc's default -ffp-contract=fast may otherwise fuse the multiplication that
  // produced x with the `x - hi` below into an fma in one kernel and not in another (it did, once `wf` was a template
  // constant in some kernels and a run-time value in others), and the same value must split identically everywhere —
  // rows are bit-identical across the 8 / 16 / 32-row kernels only if their operands are
  asm volatile("" : "+v"(x));
  if (wf) dd_split_hl_f16(x, hi, lo);
  else dd_split_hl(x, hi, lo);
}
__device__ __forceinline__ float dd_w16_to_f32(uint32_t b, int wf) { return wf ? dd_f16_to_f32(b) : dd_bf16_to_f32(b); }

// Prefill activations are stored in the SAME 16x32 operand-tile order as the weights: element (row m, col k) of a
// [M][K] plane lives at u16 offset (((m/16)*S + k/32)*64 + ((k%32)/8)*16 + m%16)*8 + k%8, S = K/32, so every A
// fragment of the prefill GEMM is one contiguous 1 KiB wave load (guide: fragment-shaped row-major loads cost 18-45 %).
__device__ __forceinline__ size_t apack_off(int m, int k, int S) {
  return ((((size_t)(m >> 4) * S + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (m & 15)) << 3) + (k & 7);
}

// 16x16x32 MFMA on 16-bit operand tiles of the engine's weight type
template <int WF>
__device__ __forceinline__ f32x4_t dd_mfma16(u32x4_t a, u32x4_t b, f32x4_t c) {
  if constexpr (WF) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dd_f16x8_t, a), __builtin_bit_cast(dd_f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ float dd_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// Softmax statistics of one 64-column block of a logits row, held by a 16-lane group as four values per lane (lane c: columns c, c + 16, c + 32,
// c + 48 of the block; invalid columns = -INFINITY): m = the block's maximum, s = sum exp(x - m).  The order of every operation is part of the
// result: the prefill GEMM's epilogue (dd_prefill.hip gemm_epilogue) and the stand-alone scorer (dd_dropout.hip k_row_partials) both call this.
__device__ __forceinline__ void dd_row_block_stats(const float (&x)[4], float& m, float& s) {
  float mm = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) mm = fmaxf(mm, __shfl_xor(mm, o));
  float ss = 0.f;
  if (mm != -INFINITY) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ss += expf(x[j] - mm);          // exp(-inf) = 0 for the invalid columns
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o);
  m = mm, s = ss;
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

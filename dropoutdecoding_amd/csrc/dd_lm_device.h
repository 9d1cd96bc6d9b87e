// Device helpers shared by the LM kernel files (dd_lm_kernels.hip, dd_gemv.hip, dd_attn_decode.hip, dd_prefill.hip).
#pragma once
#include "dd_common.h"

#define ROPE_HALF 64
#define HEAD_DIM 128
#define ATT_SPLIT 64      // keys per attention tile (decode and the VALU prefill attention)
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
#define RC_(...)                     \
  do {                               \
    int rc__ = (__VA_ARGS__);        \
    if (rc__ != DD_OK) return rc__;  \
  } while (0)

// fp8 (e4m3fn) x16 -> two bf16x8 MFMA operands, exact (3 mantissa bits fit bf16's 7): gfx950's
// v_cvt_scalef32_pk_bf16_fp8 turns two fp8 into one packed bf16 pair per instruction (scale 1.0) - 8 VALU ops per KiB
__device__ __forceinline__ void fp8x16_to_bf16(u32x4_t w, u32x4_t& k0, u32x4_t& k1) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  uint32_t o[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    bf2_t a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[d], 1.0f, false);
    bf2_t b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[d], 1.0f, true);
    o[2 * d] = __builtin_bit_cast(uint32_t, a);
    o[2 * d + 1] = __builtin_bit_cast(uint32_t, b);
  }
  k0 = (u32x4_t){o[0], o[1], o[2], o[3]};
  k1 = (u32x4_t){o[4], o[5], o[6], o[7]};
}

// write the hi/lo split of value y for (row m, k index k) into a packed decode operand
__device__ __forceinline__ void xop_store(u32x4_t* xop, int k, int m, float y, int wf = 0) {
  uint32_t hi, lo;
  dd_split(y, hi, lo, wf);
  uint16_t* p = (uint16_t*)xop;
  int ks = k >> 5, h = (k >> 3) & 3, j = k & 7;
  size_t base = ((size_t)ks * 64 + h * 16) * 8 + j;
  p[base + (size_t)m * 8] = (uint16_t)hi;
  p[base + (size_t)(m + 8) * 8] = (uint16_t)lo;
}
// the same into plane (m >> 3) of a multi-plane operand (planes of S * 64 tiles)
__device__ __forceinline__ void xop_store16(u32x4_t* xop, int k, int m, float y, int S, int wf = 0) {
  xop_store(xop + (size_t)(m >> 3) * S * 64, k, m & 7, y, wf);   // plane = group of the row
}

// Rotary mix of one element: q * cos -/+ q_partner * sin (HF apply_rotary_pos_emb with rotate_half: the first half of a head
// takes -partner, the second +partner), in ONE fixed form for every kernel that applies it: the partner product rounded on its
// own, then a fused multiply-add — fma(y, c, -/+ round(yp * sn)).  hipcc's default -ffp-contract=fast would otherwise pick a
// contraction per call site (it did: a four-wide epilogue got a different one for its fourth element), and a row's bits must
// not depend on which kernel roped it.
__device__ __forceinline__ float dd_rope_mix(float y, float yp, float c, float sn, bool first_half) {
  float p = yp * sn;
  asm volatile("" : "+v"(p));
  return __builtin_fmaf(y, c, first_half ? -p : p);
}

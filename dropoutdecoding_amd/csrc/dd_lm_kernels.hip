// LM kernels for the K-way masked-context decode step and the prefill pass (gfx950, wave64).
//   - decode GEMV: v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the A operand (16 output rows) and the
//     packed activation hi/lo pairs of up to 8 ensemble rows as the 16 B-operand columns; weights are
//     streamed straight to VGPRs (1 KiB per wave instruction, non-temporal), K is split over the 8 waves of a
//     workgroup and reduced through LDS in a fixed order (bit-reproducible).
//   - decode attention: one wave per (kv head, 64-key split); every K/V tile is loaded once and consumed by
//     all rows (ensemble members x GQA group) with their own drop bits; flash-decoding style combine.
//   - prefill GEMM: the same packed weights as the MFMA B operand, activations as hi/lo bf16 planes.
// Reference anchors: the third-party LM forward the reference calls at models/llava.py:294-303,350-359.
#include <type_traits>

#include "dd_lm_kernels.h"
#include "dd_gemv_slices.h"

#define ROPE_HALF 64
#define HEAD_DIM 128

// ===============================================================================================
// packing / init
// ===============================================================================================
__global__ __launch_bounds__(256) void k_pack_weight(const uint16_t* __restrict__ src, int rows, int cols,
                                                     u32x4_t* __restrict__ dst, int dst_tile0, int tile_stride,
                                                     int pack_mode, int n_src_tiles) {
  int S = cols >> 5;
  size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)n_src_tiles * S * 64;
  if (gid >= total) return;
  int lane = (int)(gid & 63);
  size_t q = gid >> 6;
  int ks = (int)(q % S);
  int nt = (int)(q / S);
  int h = lane >> 4, r = lane & 15;
  int row;
  if (pack_mode == PACK_ROPE) {
    int head = nt >> 3, tt = nt & 7;
    row = head * HEAD_DIM + (r < 8 ? tt * 8 + r : ROPE_HALF + tt * 8 + (r - 8));
  } else {
    row = nt * 16 + r;
  }
  u32x4_t v = {0u, 0u, 0u, 0u};
  if (row < rows) v = *(const u32x4_t*)(src + (size_t)row * cols + (size_t)ks * 32 + 8 * h);
  dst[((size_t)(dst_tile0 + nt * tile_stride) * S + ks) * 64 + lane] = v;
}

int ddk_pack_weight(const uint16_t* src, int rows, int cols, u32x4_t* dst, int dst_tile0, int tile_stride,
                    int pack_mode, int n_src_tiles, hipStream_t st) {
  size_t total = (size_t)n_src_tiles * (cols >> 5) * 64;
  k_pack_weight<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(src, rows, cols, dst, dst_tile0, tile_stride,
                                                                 pack_mode, n_src_tiles);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

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

__global__ __launch_bounds__(256) void k_pack_weight_fp8(const uint8_t* __restrict__ src, const float* __restrict__ rs,
                                                         int rows, int cols, u32x4_t* __restrict__ dst,
                                                         float* __restrict__ dscale, int dst_tile0, int tile_stride,
                                                         int pack_mode, int n_src_tiles) {
  int S2 = cols >> 6;
  size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)n_src_tiles * S2 * 64;
  if (gid >= total) return;
  int lane = (int)(gid & 63);
  size_t q = gid >> 6;
  int ks2 = (int)(q % S2);
  int nt = (int)(q / S2);
  int h = lane >> 4, r = lane & 15;
  int row;
  if (pack_mode == PACK_ROPE) {
    int head = nt >> 3, tt = nt & 7;
    row = head * HEAD_DIM + (r < 8 ? tt * 8 + r : ROPE_HALF + tt * 8 + (r - 8));
  } else {
    row = nt * 16 + r;
  }
  u32x4_t v = {0u, 0u, 0u, 0u};
  if (row < rows) {
    const uint8_t* p = src + (size_t)row * cols + (size_t)ks2 * 64 + 8 * h;
    u32x2_t a = *(const u32x2_t*)p, b = *(const u32x2_t*)(p + 32);
    v = (u32x4_t){a.x, a.y, b.x, b.y};
  }
  int dt = dst_tile0 + nt * tile_stride;
  dst[((size_t)dt * S2 + ks2) * 64 + lane] = v;
  if (ks2 == 0 && h == 0) dscale[(size_t)dt * 16 + r] = row < rows ? rs[row] : 0.f;
}
int ddk_pack_weight_fp8(const uint8_t* src, const float* rs, int rows, int cols, u32x4_t* dst, float* dscale,
                        int dst_tile0, int tile_stride, int pack_mode, int n_src_tiles, hipStream_t st) {
  size_t total = (size_t)n_src_tiles * (cols >> 6) * 64;
  k_pack_weight_fp8<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(src, rs, rows, cols, dst, dscale, dst_tile0,
                                                                     tile_stride, pack_mode, n_src_tiles);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
// fp8 tiles -> bf16 tiles of the plain layout (prefill GEMM operand), scales stay separate
__global__ __launch_bounds__(256) void k_dequant_tiles(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst,
                                                       size_t total, int S2) {
  size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  int lane = (int)(gid & 63);
  size_t q = gid >> 6;
  int ks2 = (int)(q % S2);
  size_t nt = q / S2;
  u32x4_t k0, k1;
  fp8x16_to_bf16(src[gid], k0, k1);
  size_t o = ((nt * (size_t)(2 * S2) + 2 * ks2) * 64) + lane;
  dst[o] = k0;
  dst[o + 64] = k1;
}
int ddk_dequant_tiles(const u32x4_t* src, u32x4_t* dst, int n_tiles, int S, hipStream_t st) {
  size_t total = (size_t)n_tiles * (S >> 1) * 64;
  k_dequant_tiles<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(src, dst, total, S >> 1);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// deterministic pseudo-random bf16 fill (sum of 4 uniforms ~ normal), synthetic-weights bench mode
__global__ __launch_bounds__(256) void k_fill_synth(uint16_t* dst, size_t n, uint32_t seed, float std, int wf) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h = hash32((uint32_t)i * 0x9e3779b9u + seed) ^ hash32((uint32_t)(i >> 32) + seed * 31u);
  uint32_t h2 = hash32(h + 0x68bc21ebu);
  float u = ((h & 0xffff) + (h >> 16) + (h2 & 0xffff) + (h2 >> 16)) * (1.0f / 65536.0f) - 2.0f;  // var = 1/3
  dst[i] = (uint16_t)(wf ? dd_f16_bits(u * 1.7320508f * std) : dd_bf16_rn(u * 1.7320508f * std));
}
// random finite e4m3fn bytes (0x7f / 0xff are NaN in the OCP encoding and are avoided)
__global__ __launch_bounds__(256) void k_fill_synth_fp8(uint8_t* dst, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h = hash32((uint32_t)i * 0x9e3779b9u + seed) ^ hash32((uint32_t)(i >> 32) + seed * 31u);
  uint8_t b = (uint8_t)(h & 0xff);
  if ((b & 0x7f) >= 0x78) b = (b & 0x80) | ((b & 0x3f) + 0x20);   // keep |value| <= 240, never NaN
  dst[i] = b;
}
int ddk_fill_synthetic_fp8(uint8_t* dst, size_t n, uint32_t seed, hipStream_t st) {
  k_fill_synth_fp8<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(dst, n, seed);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int ddk_fill_synthetic(uint16_t* dst, size_t n, uint32_t seed, float std, hipStream_t st, int wf) {
  k_fill_synth<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(dst, n, seed, std, wf);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
__global__ void k_fill_const(float* dst, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = v;
}
int ddk_fill_const_f32(float* dst, size_t n, float v, hipStream_t st) {
  k_fill_const<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(dst, n, v);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
__global__ void k_bf16_to_f32(const uint16_t* src, float* dst, int n, int wf) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = dd_w16_to_f32(src[i], wf);
}
int ddk_bf16_to_f32(const uint16_t* src, float* dst, int n, hipStream_t st, int wf) {
  k_bf16_to_f32<<<(n + 255) / 256, 256, 0, st>>>(src, dst, n, wf);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
// cos/sin[pos][i] = cos/sin(float(pos) * inv_freq[i])   (HF LlamaRotaryEmbedding.forward, fp32)
__global__ void k_rope_table(float* c, float* s, int max_seq, const float* inv_freq) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= max_seq * ROPE_HALF) return;
  float ang = __fmul_rn((float)(i / ROPE_HALF), inv_freq[i % ROPE_HALF]);
  c[i] = cosf(ang);
  s[i] = sinf(ang);
}
int ddk_rope_table(float* c, float* s, int max_seq, const float* inv_freq, hipStream_t st) {
  k_rope_table<<<(max_seq * ROPE_HALF + 255) / 256, 256, 0, st>>>(c, s, max_seq, inv_freq);
  DD_CHECK_LAUNCH();
  return DD_OK;
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

// ===============================================================================================
// decode GEMV
// ===============================================================================================
#define GEMV_WAVES 8
#define GEMV_THREADS (GEMV_WAVES * 64)
#define RC_(...)                     \
  do {                               \
    int rc__ = (__VA_ARGS__);        \
    if (rc__ != DD_OK) return rc__;  \
  } while (0)

// U   = weight tiles requested per wave before the first MFMA consumes one (loads in flight)
// NT  = non-temporal weight loads (read-once stream, keeps L2/MALL for the x operand and the KV cache)
// ILV = k-steps interleaved over the 8 waves (wave w takes steps w, w+8, ...: at any instant the workgroup reads
//       8 consecutive KiB) instead of one contiguous chunk per wave
template <int EPI, int TILES, int U, int NT, int ILV, int FP8 = 0, int PIPE = 0, int WF = 0>
__global__ __launch_bounds__(GEMV_THREADS) void k_gemv(GemvArgs a) {
  __shared__ float red[TILES * GEMV_WAVES * 256];
  __shared__ float rstd_sh[8];
  __shared__ float ssq_sh[8 * 16];
  if (a.skip_if && *a.skip_if) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S, spw = S / GEMV_WAVES;
  const int s0 = ILV ? wave : wave * spw;
  constexpr int SS = ILV ? GEMV_WAVES : 1;   // step stride of this wave
  const int tile0 = blockIdx.x * TILES;

  f32x4_t acc[TILES];
  const u32x4_t* wp[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    wp[t] = a.W + ((size_t)(tile0 + t) * S + s0) * 64 + lane;
  }
  const u32x4_t* xp = a.xop + (size_t)s0 * 64 + lane;
  auto ldw = [](const u32x4_t* p) -> u32x4_t { return NT ? __builtin_nontemporal_load(p) : *p; };

  // Everything the epilogue needs from memory is requested BEFORE the weight stream so its latency hides behind it:
  // the folded RMSNorm's rstd (wave w assembles row w's sum of squares from the producer's slots), the residual
  // input + next norm weight (EPI_RESID) and the rotary cos/sin (EPI_QKV).  The sum-of-squares slots are only
  // REQUESTED here; they are reduced after the first weight batch has been issued (loads return in order, so waiting
  // on them does not wait on the weights behind them).
  const bool has_ssq = a.ssq_in != nullptr;
  f32x4_t sv = {0.f, 0.f, 0.f, 0.f};
  // row `wave`'s slots are contiguous: one 16-byte load per lane covers 256 slots (every workgroup of the launch reads
  // these same few lines, so the request count matters: strided 4-byte reads here cost ~2 us per launch)
  if (has_ssq && 4 * lane < a.ssq_n) sv = *(const f32x4_t*)(a.ssq_in + (size_t)wave * a.ssq_ld + 4 * lane);
  float pre0 = 0.f, pre1 = 0.f;
  {
    const int em = threadIdx.x & 7, en = threadIdx.x >> 3;
    if (threadIdx.x < 128 && em < a.nb) {
      if (EPI == EPI_RESID) {
        pre0 = a.out[(size_t)em * a.ldo + tile0 * 16 + en];
        pre1 = a.normw_next[tile0 * 16 + en];
      } else if (EPI == EPI_QKV) {
        if (tile0 < a.q_tiles + a.k_tiles) {
          int ht = tile0 < a.q_tiles ? tile0 : tile0 - a.q_tiles;
          int f = (ht & 7) * 8 + (en & 7);
          const DDState* sp = a.state_rows[em] ? a.state_rows[em] : a.state;
          int pos = sp->pos;
          pre0 = a.rope_cos[(size_t)pos * ROPE_HALF + f];
          pre1 = a.rope_sin[(size_t)pos * ROPE_HALF + f];
        }
      } else if (EPI == EPI_STORE) {
        // logits of a sequence that already emitted its EOS are not overwritten by look-ahead steps (DDState::done)
        const DDState* sp = a.state_rows[em] ? a.state_rows[em] : a.state;
        if (sp && sp->done) pre0 = 1.f;
      }
    }
  }

  if constexpr (FP8) {
    // fp8 weights: one 1 KiB load = 64 k of a tile row = two MFMA k-steps; wave w takes 64-k steps w, w+8, ...
    // (uneven tails allowed: K = 11008 has 172 such steps)
    const int S2 = S >> 1;
    const u32x4_t* wq[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) wq[t] = a.W + (size_t)(tile0 + t) * S2 * 64 + lane;
    const u32x4_t* xq = a.xop + lane;
    constexpr int UF = 4;   // fp8 loads in flight per tile = 8 bf16 k-steps
    for (int s2 = wave; s2 < S2; s2 += GEMV_WAVES * UF) {
      u32x4_t wf[TILES][UF], b0[UF], b1[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        int ss = s2 + u * GEMV_WAVES;
        if (ss < S2) {
#pragma unroll
          for (int t = 0; t < TILES; ++t) wf[t][u] = ldw(wq[t] + (size_t)ss * 64);
          b0[u] = xq[(size_t)(2 * ss) * 64];
          b1[u] = xq[(size_t)(2 * ss + 1) * 64];
        }
      }
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        int ss = s2 + u * GEMV_WAVES;
        if (ss < S2) {
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            u32x4_t k0, k1;
            fp8x16_to_bf16(wf[t][u], k0, k1);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, k0),
                                                             __builtin_bit_cast(bf16x8_t, b0[u]), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, k1),
                                                             __builtin_bit_cast(bf16x8_t, b1[u]), acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
  auto finish_rstd = [&]() {
    if (has_ssq) {
      const int i0 = 4 * lane;
      float v = 0.f;
      if (i0 < a.ssq_n) v += sv.x;
      if (i0 + 1 < a.ssq_n) v += sv.y;
      if (i0 + 2 < a.ssq_n) v += sv.z;
      if (i0 + 3 < a.ssq_n) v += sv.w;
      for (int i = lane + 256; i < a.ssq_n; i += 64) v += a.ssq_in[(size_t)wave * a.ssq_ld + i];
      v = dd_wave_sum(v);
      if (lane == 0) rstd_sh[wave] = 1.0f / sqrtf(v * a.inv_k + a.eps);
    }
  };
  finish_rstd();
  if constexpr (!FP8 && PIPE == 0) {
    // batches: U steps requested together, then consumed; the other resident waves cover the drain
    int s = 0;
    for (; s + U <= spw; s += U) {
      u32x4_t b[U], w[TILES][U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) w[t][u] = ldw(wp[t] + (size_t)(s + u) * SS * 64);
        b[u] = xp[(size_t)(s + u) * SS * 64];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < TILES; ++t)
          acc[t] = dd_mfma16<WF>(w[t][u], b[u], acc[t]);
    }
    if (s < spw) {  // tail: the remaining (< U) steps requested together as well (K = 11008: 43 steps per wave)
      const int rem = spw - s;
      u32x4_t b[U], w[TILES][U];
#pragma unroll
      for (int u = 0; u < U - 1; ++u) {
        if (u < rem) {
#pragma unroll
          for (int t = 0; t < TILES; ++t) w[t][u] = ldw(wp[t] + (size_t)(s + u) * SS * 64);
          b[u] = xp[(size_t)(s + u) * SS * 64];
        }
      }
#pragma unroll
      for (int u = 0; u < U - 1; ++u) {
        if (u < rem) {
#pragma unroll
          for (int t = 0; t < TILES; ++t)
            acc[t] = dd_mfma16<WF>(w[t][u], b[u], acc[t]);
        }
      }
    }
  }
  if constexpr (!FP8 && PIPE == 1) {
    // Ring of U requests per wave: slot u is consumed by its MFMA and immediately re-requested U steps ahead, so the
    // wave always has ~U weight tiles in flight (no drain between batches).
    const int n = spw;
    u32x4_t b[U], w[TILES][U];
    auto req = [&](int u, int step) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) w[t][u] = ldw(wp[t] + (size_t)step * SS * 64);
      b[u] = xp[(size_t)step * SS * 64];
    };
    auto use = [&](int u) {
#pragma unroll
      for (int t = 0; t < TILES; ++t)
        acc[t] = dd_mfma16<WF>(w[t][u], b[u], acc[t]);
    };
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < n) req(u, u);
    int s = 0;
    for (; s + 2 * U <= n; s += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        use(u);
        req(u, s + U + u);
        __builtin_amdgcn_sched_barrier(0);   // keep consume -> re-request order (otherwise the scheduler sinks all
      }                                      // requests below the last MFMA, which is the batch order again)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s + u < n) use(u);
      if (s + U + u < n) req(u, s + U + u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s + U + u < n) use(u);
  }

#pragma unroll
  for (int t = 0; t < TILES; ++t) *(f32x4_t*)&red[(t * GEMV_WAVES + wave) * 256 + lane * 4] = acc[t];
  __syncthreads();

  // D[n][c]: lane = (n>>2)*16 + c, reg = n&3.  y[m][n] = sum_w (D_w[n][m] + D_w[n][m+8])   (hi + lo columns)
  const int t = threadIdx.x;
  // fixed order: (hi + lo) per wave, waves added in pairs, pairs in sequence — the order the slice-resident kernels
  // reproduce from partial sums (dd_gemv_slices.h): y = sum_p ((hi+lo)(2p) + (hi+lo)(2p+1))
  auto tile_sum = [&](int tt, int n, int m) -> float {
    float y = 0.f;
    int o = ((n >> 2) * 16 + m) * 4 + (n & 3);
#pragma unroll
    for (int w = 0; w < GEMV_WAVES; w += 2) {
      const float* r = &red[(tt * GEMV_WAVES + w) * 256];
      y += (r[o] + r[o + 32]) + (r[256 + o] + r[256 + o + 32]);
    }
    if (FP8) y *= a.wscale[(size_t)(tile0 + tt) * 16 + n];
    return y;
  };

  if (EPI == EPI_STORE) {
    if (t < 128) {
      int m = t & 7, n = t >> 3;
      if (m < a.nb) {
        float y = tile_sum(0, n, m);
        if (a.ssq_in) y *= rstd_sh[m];
        int col = tile0 * 16 + n;
        if (col < a.n_valid && pre0 == 0.f) a.out[(size_t)m * a.ldo + col] = y;
      }
    }
  } else if (EPI == EPI_RESID) {
    float sq = 0.f;
    int m = t & 7, n = t >> 3;
    if (t < 128 && m < a.nb) {
      float y = tile_sum(0, n, m);
      int col = tile0 * 16 + n;
      float xn = pre0 + y;
      a.out[(size_t)m * a.ldo + col] = xn;
      xop_store(a.xop_next, col, m, pre1 * xn, WF);
      sq = xn * xn;
    }
    if (t < 128) ssq_sh[n * 8 + m] = sq;
    __syncthreads();
    if (t < 8) {
      float v = 0.f;
      for (int i = 0; i < 16; ++i) v += ssq_sh[i * 8 + t];
      a.ssq_out[(size_t)t * a.ssq_ld + blockIdx.x] = v;
    }
  } else if (EPI == EPI_SILU) {
    if (t < 128) {
      int m = t & 7, n = t >> 3;
      if (m < a.nb) {
        float g = tile_sum(0, n, m), u = tile_sum(TILES - 1, n, m);
        if (a.ssq_in) {
          g *= rstd_sh[m];
          u *= rstd_sh[m];
        }
        float act = g / (1.0f + expf(-g));  // silu
        xop_store(a.xop_next, blockIdx.x * 16 + n, m, act * u, WF);
      }
    }
  } else {  // EPI_QKV
    if (t < 128) {
      int m = t & 7, n = t >> 3;
      if (m < a.nb) {
        float y = tile_sum(0, n, m);
        if (a.ssq_in) y *= rstd_sh[m];
        int nt = tile0;
        if (nt < a.q_tiles + a.k_tiles) {
          float yp = tile_sum(0, n ^ 8, m);
          if (a.ssq_in) yp *= rstd_sh[m];
          bool is_q = nt < a.q_tiles;
          int ht = is_q ? nt : nt - a.q_tiles;
          int head = ht >> 3, f = (ht & 7) * 8 + (n & 7);
          float c = pre0, sn = pre1;
          // q*cos + rotate_half(q)*sin, two rounded products then one add (HF apply_rotary_pos_emb)
          float o = (n < 8) ? __fadd_rn(__fmul_rn(y, c), __fmul_rn(-yp, sn)) : __fadd_rn(__fmul_rn(y, c), __fmul_rn(yp, sn));
          int i = (n < 8) ? f : ROPE_HALF + f;
          if (is_q) a.qbuf[(size_t)m * a.q_dim + head * HEAD_DIM + i] = o;
          else a.knew[(size_t)m * a.kv_dim + head * HEAD_DIM + i] = o;
        } else {
          int col = (nt - a.q_tiles - a.k_tiles) * 16 + n;
          a.vnew[(size_t)m * a.kv_dim + col] = y;
        }
      }
    }
  }
}

// tuning knobs (dd_set_tuning): 0 = U (4/8/16), 4 = ring (1) or batch (0) request order.
// Keys 1 (non-temporal loads) and 2 (k-step interleave) are settled at 1 and kept only as accepted no-ops.
static int g_gemv_u = 8, g_gemv_pipe = 0;
void ddk_set_tuning(int key, int value) {
  if (key == 0) g_gemv_u = value;
  else if (key == 4) g_gemv_pipe = value;
}

template <int EPI, int TILES>
static void launch_gemv(const GemvArgs& a_, hipStream_t st) {
  const GemvArgs& a = a_;
#define GV(U_, P_)                                                                             \
  do {                                                                                         \
    if (a.wf) k_gemv<EPI, TILES, U_, 1, 1, 0, P_, 1><<<a.n_tiles, GEMV_THREADS, 0, st>>>(a);   \
    else k_gemv<EPI, TILES, U_, 1, 1, 0, P_, 0><<<a.n_tiles, GEMV_THREADS, 0, st>>>(a);        \
  } while (0)
  if (a.fp8) { k_gemv<EPI, TILES, 8, 1, 1, 1><<<a.n_tiles, GEMV_THREADS, 0, st>>>(a); return; }
  const int u = g_gemv_u;
  if (g_gemv_pipe) { if (u == 4) GV(4, 1); else if (u == 16) GV(16, 1); else GV(8, 1); }
  else { if (u == 4) GV(4, 0); else if (u == 16) GV(16, 0); else GV(8, 0); }
#undef GV
}

int ddk_gemv(int epi, const GemvArgs& a, hipStream_t st) {
  DD_REQUIRE(a.S % GEMV_WAVES == 0 && a.S >= GEMV_WAVES, "gemv: K=%d must be a multiple of 256", a.S * 32);
  DD_REQUIRE(a.nb >= 1 && a.nb <= 8, "gemv: nb=%d", a.nb);
  DD_REQUIRE(!a.fp8 || a.wscale, "gemv: fp8 weights need row scales");
  DD_REQUIRE(!a.ssq_in || a.ssq_n >= 1, "gemv: ssq_n");
  switch (epi) {
    case EPI_STORE: launch_gemv<EPI_STORE, 1>(a, st); break;
    case EPI_RESID: launch_gemv<EPI_RESID, 1>(a, st); break;
    case EPI_SILU: launch_gemv<EPI_SILU, 2>(a, st); break;
    case EPI_QKV: launch_gemv<EPI_QKV, 1>(a, st); break;
    default: DD_REQUIRE(false, "gemv: unknown epilogue %d", epi);
  }
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// ===============================================================================================
// decode GEMV, NG groups of 8 rows (NG = 2 or 4; e.g. the members of NG sequences) against ONE pass over the weights.
// Same tiles, same k order and the same fixed-order reduction per output as k_gemv, so a row's result does not depend
// on which of the kernels computed it; group g's B operand is plane g of the packed operand and costs one more MFMA per
// tile step (the kernel is HBM-bound; even 4 planes keep the MFMA pipe under half busy).
// ===============================================================================================
__device__ __forceinline__ void xop_store16(u32x4_t* xop, int k, int m, float y, int S, int wf = 0) {
  xop_store(xop + (size_t)(m >> 3) * S * 64, k, m & 7, y, wf);   // plane = group of the row
}

// What the epilogue needs from memory, requested before the weight stream (k_gemv_groups) or before the partial sums
// (k_gemv_finish): residual + next norm weight (EPI_RESID), rotary cos/sin (EPI_QKV), the finished flag (EPI_STORE).
// Epilogue thread t < 128 * NG: group t >> 7, row m = 8 * group + (t & 7), column n = (t & 127) >> 3.
template <int TILES>
struct GroupsPre {
  float pre0, pre1;
  float rope_c[TILES], rope_s[TILES];
};
template <int EPI, int TILES, int NG>
__device__ __forceinline__ void groups_prefetch(const GemvArgs& a, int tile0, GroupsPre<TILES>& p) {
  const int et = threadIdx.x, eg = et >> 7, ml = et & 7, em = (eg << 3) + ml, en = (et & 127) >> 3;
  const bool erow = et < 128 * NG && ml < a.nb;
  p.pre0 = p.pre1 = 0.f;
#pragma unroll
  for (int tt = 0; tt < TILES; ++tt) p.rope_c[tt] = p.rope_s[tt] = 0.f;
  if (erow) {
    if (EPI == EPI_RESID) {
      p.pre0 = a.out[(size_t)em * a.ldo + tile0 * 16 + en];
      p.pre1 = a.normw_next[tile0 * 16 + en];
    } else if (EPI == EPI_QKV) {
      const DDState* sp = a.state_rows[em] ? a.state_rows[em] : a.state;
      const int pos = sp->pos;
#pragma unroll
      for (int tt = 0; tt < TILES; ++tt) {     // one (cos, sin) pair per tile of the workgroup (q_tiles, k_tiles are even)
        const int nt = tile0 + tt;
        if (nt < a.q_tiles + a.k_tiles) {
          int ht = nt < a.q_tiles ? nt : nt - a.q_tiles;
          int f = (ht & 7) * 8 + (en & 7);
          p.rope_c[tt] = a.rope_cos[(size_t)pos * ROPE_HALF + f];
          p.rope_s[tt] = a.rope_sin[(size_t)pos * ROPE_HALF + f];
        }
      }
    } else if (EPI == EPI_STORE) {
      const DDState* sp = a.state_rows[em] ? a.state_rows[em] : a.state;
      if (sp && sp->done) p.pre0 = 1.f;    // finished sequence: its logits stay as the EOS step left them
    }
  }
}
// folded RMSNorm: wave w assembles rstd of rows w, w + 8, ... from the producer's sum-of-squares slots
template <int NG>
__device__ __forceinline__ void groups_rstd(const GemvArgs& a, float* rstd_sh) {
  if (a.ssq_in) dd_rows_rstd<NG>(a.ssq_in, a.ssq_n, a.ssq_ld, a.inv_k, a.eps, rstd_sh);
}
// everything after the reduction; tile_sum(tt, n) = this thread's (group, row) sum for output row n of tile tt.
// wg = index of the workgroup's tile set (k_gemv_groups: blockIdx.x); rstd_sh must be visible (barrier before the call).
template <int EPI, int TILES, int NG, typename TS>
__device__ __forceinline__ void groups_epilogue(const GemvArgs& a, int wg, const GroupsPre<TILES>& p, const float* rstd_sh,
                                                float* ssq_sh, TS tile_sum) {
  const int tile0 = wg * TILES;
  const int et = threadIdx.x, eg = et >> 7, ml = et & 7, em = (eg << 3) + ml, en = (et & 127) >> 3;
  const bool erow = et < 128 * NG && ml < a.nb;
  if (EPI == EPI_STORE) {
    if (erow) {
      float y = tile_sum(0, en);
      if (a.ssq_in) y *= rstd_sh[em];
      int col = tile0 * 16 + en;
      float* row = a.out_g[eg] ? a.out_g[eg] + (size_t)ml * a.ldo : a.out + (size_t)em * a.ldo;
      if (col < a.n_valid && p.pre0 == 0.f) row[col] = y;
    }
  } else if (EPI == EPI_RESID) {
    float sq = 0.f;
    if (erow) {
      float y = tile_sum(0, en);
      int col = tile0 * 16 + en;
      float xn = p.pre0 + y;
      a.out[(size_t)em * a.ldo + col] = xn;
      xop_store16(a.xop_next, col, em, p.pre1 * xn, a.S_next, a.wf);
      sq = xn * xn;
    }
    if (et < 128 * NG) ssq_sh[en * (8 * NG) + em] = sq;
    __syncthreads();
    if (et < 8 * NG) {
      float v = 0.f;
      for (int i = 0; i < 16; ++i) v += ssq_sh[i * (8 * NG) + et];
      a.ssq_out[(size_t)et * a.ssq_ld + wg] = v;
    }
  } else if (EPI == EPI_SILU) {
    if (erow) {
      float g = tile_sum(0, en), u = tile_sum(TILES - 1, en);
      if (a.ssq_in) {
        g *= rstd_sh[em];
        u *= rstd_sh[em];
      }
      float act = g / (1.0f + expf(-g));  // silu
      xop_store16(a.xop_next, wg * 16 + en, em, act * u, a.S_next, a.wf);
    }
  } else {  // EPI_QKV
    if (erow) {
      float* kn = a.knew_g[eg] ? a.knew_g[eg] + (size_t)ml * a.kv_dim : a.knew + (size_t)em * a.kv_dim;
      float* vn = a.vnew_g[eg] ? a.vnew_g[eg] + (size_t)ml * a.kv_dim : a.vnew + (size_t)em * a.kv_dim;
#pragma unroll
      for (int tt = 0; tt < TILES; ++tt) {
        float y = tile_sum(tt, en);
        if (a.ssq_in) y *= rstd_sh[em];
        const int nt = tile0 + tt;
        if (nt < a.q_tiles + a.k_tiles) {
          float yp = tile_sum(tt, en ^ 8);
          if (a.ssq_in) yp *= rstd_sh[em];
          bool is_q = nt < a.q_tiles;
          int ht = is_q ? nt : nt - a.q_tiles;
          int head = ht >> 3, f = (ht & 7) * 8 + (en & 7);
          float c = p.rope_c[tt], sn = p.rope_s[tt];
          float o = (en < 8) ? __fadd_rn(__fmul_rn(y, c), __fmul_rn(-yp, sn)) : __fadd_rn(__fmul_rn(y, c), __fmul_rn(yp, sn));
          int i = (en < 8) ? f : ROPE_HALF + f;
          if (is_q) a.qbuf[(size_t)em * a.q_dim + head * HEAD_DIM + i] = o;
          else kn[head * HEAD_DIM + i] = o;
        } else {
          int col = (nt - a.q_tiles - a.k_tiles) * 16 + en;
          vn[col] = y;
        }
      }
    }
  }
}

template <int EPI, int TILES, int NG, int U = 4, int FP8 = 0, int WF = 0>
__global__ __launch_bounds__(GEMV_THREADS) void k_gemv_groups(GemvArgs a) {
  extern __shared__ float gg_sh[];
  float* red = gg_sh;                                   // [TILES * NG * 8 waves][256]
  float* rstd_sh = red + TILES * NG * GEMV_WAVES * 256;  // [8 * NG]
  float* ssq_sh = rstd_sh + 8 * NG;                     // [16][8 * NG]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S, spw = S / GEMV_WAVES;
  const int tile0 = blockIdx.x * TILES;
  f32x4_t acc[TILES][NG];
  const u32x4_t* wp[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[t][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    wp[t] = a.W + ((size_t)(tile0 + t) * S + wave) * 64 + lane;
  }
  const u32x4_t* xp = a.xop + (size_t)wave * 64 + lane;      // plane g: + g * S * 64
  const size_t xplane = (size_t)S * 64;
  GroupsPre<TILES> pre;
  groups_prefetch<EPI, TILES, NG>(a, tile0, pre);
  groups_rstd<NG>(a, rstd_sh);

  if constexpr (FP8) {
    // fp8 weights: one 1 KiB load = 64 k of a tile row = two MFMA k-steps, expanded exactly to bf16 in registers ONCE and
    // used for all NG operand planes; wave w takes 64-k steps w, w+8, ... (same order as k_gemv's fp8 path)
    const int S2 = S >> 1;
    const u32x4_t* wq[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) wq[t] = a.W + (size_t)(tile0 + t) * S2 * 64 + lane;
    const u32x4_t* xq = a.xop + lane;
    constexpr int UF = 2;
    for (int s2 = wave; s2 < S2; s2 += GEMV_WAVES * UF) {
      u32x4_t wf[TILES][UF], b0[UF][NG], b1[UF][NG];
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        int ss = s2 + u * GEMV_WAVES;
        if (ss < S2) {
#pragma unroll
          for (int t = 0; t < TILES; ++t) wf[t][u] = __builtin_nontemporal_load(wq[t] + (size_t)ss * 64);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            b0[u][g] = xq[(size_t)(2 * ss) * 64 + g * xplane];
            b1[u][g] = xq[(size_t)(2 * ss + 1) * 64 + g * xplane];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        int ss = s2 + u * GEMV_WAVES;
        if (ss < S2) {
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            u32x4_t k0, k1;
            fp8x16_to_bf16(wf[t][u], k0, k1);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, k0),
                                                                  __builtin_bit_cast(bf16x8_t, b0[u][g]), acc[t][g], 0, 0, 0);
              acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, k1),
                                                                  __builtin_bit_cast(bf16x8_t, b1[u][g]), acc[t][g], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  int s = FP8 ? spw : 0;
  for (; s + U <= spw; s += U) {
    u32x4_t b[U][NG], w[TILES][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) w[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)(s + u) * GEMV_WAVES * 64);
#pragma unroll
      for (int g = 0; g < NG; ++g) b[u][g] = xp[(size_t)(s + u) * GEMV_WAVES * 64 + g * xplane];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g)
          acc[t][g] = dd_mfma16<WF>(w[t][u], b[u][g], acc[t][g]);
  }
  if (s < spw) {
    const int rem = spw - s;
    u32x4_t b[U][NG], w[TILES][U];
#pragma unroll
    for (int u = 0; u < U - 1; ++u)
      if (u < rem) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) w[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)(s + u) * GEMV_WAVES * 64);
#pragma unroll
        for (int g = 0; g < NG; ++g) b[u][g] = xp[(size_t)(s + u) * GEMV_WAVES * 64 + g * xplane];
      }
#pragma unroll
    for (int u = 0; u < U - 1; ++u)
      if (u < rem) {
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc[t][g] = dd_mfma16<WF>(w[t][u], b[u][g], acc[t][g]);
      }
  }

#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int g = 0; g < NG; ++g) *(f32x4_t*)&red[((t * NG + g) * GEMV_WAVES + wave) * 256 + lane * 4] = acc[t][g];
  __syncthreads();

  const int eg = threadIdx.x >> 7, ml = threadIdx.x & 7;
  auto tile_sum = [&](int tt, int n) -> float {     // the order of k_gemv's tile_sum: (hi + lo) per wave, waves in pairs
    float y = 0.f;
    int o = ((n >> 2) * 16 + ml) * 4 + (n & 3);
#pragma unroll
    for (int w = 0; w < GEMV_WAVES; w += 2) {
      const float* r = &red[((tt * NG + eg) * GEMV_WAVES + w) * 256];
      y += (r[o] + r[o + 32]) + (r[256 + o] + r[256 + o + 32]);
    }
    if (FP8) y *= a.wscale[(size_t)(tile0 + tt) * 16 + n];
    return y;
  };
  groups_epilogue<EPI, TILES, NG>(a, blockIdx.x, pre, rstd_sh, ssq_sh, tile_sum);
}

// Second half of the slice-resident GEMV (dd_gemv_slices.h): adds the slices' partial sums in k_gemv's order and runs
// k_gemv_groups' epilogue for ONE tile set per workgroup (128 * NG threads).  Everything it needs from memory — the
// partial sums, rstd (assembled by the first kernel), residual / norm weight / rotary terms — is requested up front.
// part: [NP][n_tiles_total][NG][128], NP = 8 (single slices) or 4 (slice pairs already added by the producer).
template <int EPI, int TILES, int NG, int NP>
__global__ __launch_bounds__(128 * NG) void k_gemv_finish(GemvArgs a, const float* __restrict__ part, const float* __restrict__ rstd_g,
                                                          int n_sets) {
  __shared__ float ssq_sh[16 * 8 * NG];
  __shared__ float rstd_sh[8 * NG];
  __shared__ float y_sh[EPI == EPI_QKV ? TILES * 128 * NG : 1];   // rotary tiles: a thread needs its partner column's sum (n ^ 8)
  const int wg = blockIdx.x, tile0 = wg * TILES;
  const int et = threadIdx.x, eg = et >> 7, ml = et & 7, en = (et & 127) >> 3;
  const size_t ps = ((size_t)n_sets * TILES * NG) << 7;
  // one batch of requests: the thread's partial sums, rstd, the epilogue's operands — a single memory round trip before the
  // arithmetic
  float v[TILES][NP];
#pragma unroll
  for (int tt = 0; tt < TILES; ++tt) {
    const float* p0 = part + (((size_t)(tile0 + tt) * NG + eg) << 7) + dd_part_index(en, ml);
#pragma unroll
    for (int q = 0; q < NP; ++q) v[tt][q] = p0[(size_t)q * ps];
  }
  if (a.ssq_in && et < 8 * NG) rstd_sh[et] = rstd_g[et];
  GroupsPre<TILES> pre;
  groups_prefetch<EPI, TILES, NG>(a, tile0, pre);
  float y_own[TILES];
#pragma unroll
  for (int tt = 0; tt < TILES; ++tt) {
    float y = 0.f;
    if (NP == 8) {
#pragma unroll
      for (int q = 0; q < 8; q += 2) y += v[tt][q] + v[tt][q + 1];
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) y += v[tt][q];
    }
    y_own[tt] = y;
    if (EPI == EPI_QKV) y_sh[tt * 128 * NG + et] = y;   // the partner (same group and row, column n ^ 8) is thread et ^ 64
  }
  __syncthreads();                                   // rstd_sh, y_sh
  auto tile_sum = [&](int tt, int n) -> float { return (EPI == EPI_QKV && n != en) ? y_sh[tt * 128 * NG + (et ^ 64)] : y_own[tt]; };
  groups_epilogue<EPI, TILES, NG>(a, wg, pre, rstd_sh, ssq_sh, tile_sum);
}

template <int EPI, int TILES, int NG, int FP8>
static int launch_gemv_groups_f(const GemvArgs& a, hipStream_t st) {
  size_t smem = (size_t)(TILES * NG * GEMV_WAVES * 256 + 8 * NG + 16 * 8 * NG) * sizeof(float);
  // weight tiles requested per wave before the first MFMA: 4; 2 for the two-tile kernels with four operand planes (keeps
  // the register file at two workgroups per CU: gate/up 49 vs 53 us); 8 for o_proj / down (one workgroup per CU anyway:
  // 32.0 vs 33.0 us)
  constexpr int U = (TILES == 2 && NG == 4) ? 2 : (EPI == EPI_RESID ? 8 : 4);
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_groups<EPI, TILES, NG, U, FP8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (!FP8) DD_HIP(hipFuncSetAttribute((const void*)k_gemv_groups<EPI, TILES, NG, U, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  if (!FP8 && a.wf) k_gemv_groups<EPI, TILES, NG, U, 0, 1><<<a.n_tiles, GEMV_THREADS, smem, st>>>(a);
  else k_gemv_groups<EPI, TILES, NG, U, FP8, 0><<<a.n_tiles, GEMV_THREADS, smem, st>>>(a);
  return DD_OK;
}
template <int EPI, int TILES, int NG>
static int launch_gemv_groups(const GemvArgs& a, hipStream_t st) {
  return a.fp8 ? launch_gemv_groups_f<EPI, TILES, NG, 1>(a, st) : launch_gemv_groups_f<EPI, TILES, NG, 0>(a, st);
}

// ---- slice-resident path (dd_gemv_slices.h + k_gemv_finish): bf16 weights, the per-layer matrices at the shapes the 7B
// families have (K = 4096: 16 k-steps per slice; K = 11008 / 14336: 43 / 56, staged in chunks of 16).  Anything else runs
// through k_gemv_groups; both produce the same bits.
static int g_gemv_slices = 1;     // dd_set_tuning key 13
int g_exp_G[4] = {0, 0, 0, 0};   // dd_set_tuning keys 17..19: workgroups per slice of the 64-row kernels (qkv, o, gate/up); 0 = default
static int g_slices_only = 0;     // dd_lm_time_gemv: launch the streaming kernel without its finishing kernel (timing only)
void ddk_set_gemv_slices(int on) { g_gemv_slices = on; }
void ddk_set_slices_only(int on) { g_slices_only = on; }
#define SLICES_UNSUPPORTED 1

template <int TW, int NG, int U, int SPW, int CS, int CH, int TAG>
static int launch_slices_k(const SliceArgs& sa, int wf, hipStream_t st) {
  constexpr size_t smem = (size_t)CH * (SPW < CS ? SPW : CS) * NG * 1024;
  static bool attr = false;
  if (!attr) {
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices<TW, NG, U, SPW, CS, CH, 0, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices<TW, NG, U, SPW, CS, CH, 1, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  const int grid = (sa.halves == 2 ? 2 : 1) * (8 / CH) * sa.G;
  if (wf) k_gemv_slices<TW, NG, U, SPW, CS, CH, 1, TAG><<<grid, GEMV_THREADS, smem, st>>>(sa);
  else k_gemv_slices<TW, NG, U, SPW, CS, CH, 0, TAG><<<grid, GEMV_THREADS, smem, st>>>(sa);
  return DD_OK;
}
template <int EPI, int TILES, int NG, int NP>
static void launch_finish(const GemvArgs& a, int n_sets, hipStream_t st) {
  if (g_slices_only) return;
  k_gemv_finish<EPI, TILES, NG, NP><<<n_sets, 128 * NG, 0, st>>>(a, a.part, a.part + a.part_floats, n_sets);
}
template <int NG>
static int try_slices(int epi, const GemvArgs& a, hipStream_t st) {
  const int spw = a.S / GEMV_WAVES;
  const int nt = epi == EPI_SILU ? 2 * a.n_tiles : a.n_tiles;      // 16-row weight tiles
  if (!(spw == 16 || spw == 43 || spw == 56) || nt < 64) return SLICES_UNSUPPORTED;
  SliceArgs sa;
  sa.W = a.W, sa.xop = a.xop, sa.part = a.part, sa.S = a.S, sa.halves = 1;
  sa.ssq_in = a.ssq_in, sa.ssq_n = a.ssq_n, sa.ssq_ld = a.ssq_ld, sa.inv_k = a.inv_k, sa.eps = a.eps;
  sa.rstd_out = a.part + a.part_floats;                              // 32 floats behind the partial sums
  const size_t need8 = (size_t)8 * nt * NG * 128, need4 = need8 / 2;
  if (epi == EPI_STORE) {
    // lm_head (K = 4096): the wave-split kernel streams it at 2.8 TB/s with four planes (operand reads from L2); the slice kernels
    // with the plain-store finish: 32 rows as slice pairs, 64 rows as single slices, 16 rows stay on the wave-split kernel
    if (spw != 16 || NG < 4) return SLICES_UNSUPPORTED;
    sa.n_groups = nt;
    if constexpr (NG == 8) {
      if (a.part_floats < need8) return SLICES_UNSUPPORTED;
      sa.G = (nt + 31) / 32;
      RC_(launch_slices_k<1, 8, 8, 16, 16, 1, EPI_STORE>(sa, a.wf, st));
      launch_finish<EPI_STORE, 1, 8, 8>(a, nt, st);
    } else {
      if (a.part_floats < need4) return SLICES_UNSUPPORTED;
      sa.G = 64;
      RC_(launch_slices_k<1, NG, 8, 16, 16, 2, EPI_STORE>(sa, a.wf, st));
      launch_finish<EPI_STORE, 1, NG, 4>(a, nt, st);
    }
    return DD_OK;
  }
  if constexpr (NG == 8) {
    // 64 rows: 8 operand planes fill the LDS with one slice (16 steps x 8 KiB = 128 KiB at K = 4096; long K in chunks of 8
    // steps), one tile per wave group; tools/gemv_lab: qkv 28 us, o 10.7, gate/up 45, down 21 — 1.25 x the 32-row kernels for
    // twice the rows
    if (a.part_floats < need8) return SLICES_UNSUPPORTED;
    sa.n_groups = nt;
    if (epi == EPI_QKV) {
      if (spw != 16) return SLICES_UNSUPPORTED;
      // workgroups per slice: one round of 8 * G <= 256 workgroups (one per CU: 128 KiB of operands each) measured best inside
      // the sweep (32 lanes: 32.6 -> 31.6 ms per group step for the three choices together)
      if (g_exp_G[0] >= 0 && (nt % 16) == 0 && nt / 16 * 4 <= 256) {
        // slice pairs, one slice resident at a time (see gate/up below): two tiles per wave; tuning key 17 < 0: single slices (A/B)
        sa.G = g_exp_G[0] ? g_exp_G[0] : nt / 16;
        constexpr size_t smem = (size_t)16 * 8 * 1024;
        static bool attr = false;
        if (!attr) {
          DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_seq<8, 8, 16, 2, 0, EPI_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
          DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_seq<8, 8, 16, 2, 1, EPI_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
          attr = true;
        }
        DD_REQUIRE((nt + 8 * sa.G - 1) / (8 * sa.G) <= 2, "gemv_slices_seq: %d tiles over %d workgroups per pair", nt, sa.G);
        if (a.wf) k_gemv_slices_seq<8, 8, 16, 2, 1, EPI_QKV><<<4 * sa.G, GEMV_THREADS, smem, st>>>(sa);
        else k_gemv_slices_seq<8, 8, 16, 2, 0, EPI_QKV><<<4 * sa.G, GEMV_THREADS, smem, st>>>(sa);
        launch_finish<EPI_QKV, 1, 8, 4>(a, nt, st);
      } else {
        sa.G = g_exp_G[0] > 0 ? g_exp_G[0] : (nt + 31) / 32;
        RC_(launch_slices_k<1, 8, 8, 16, 16, 1, EPI_QKV>(sa, a.wf, st));
        launch_finish<EPI_QKV, 1, 8, 8>(a, nt, st);
      }
    } else if (epi == EPI_RESID) {
      sa.G = (nt + 7) / 8;
      if (spw == 16 && g_exp_G[1] < 0) {                       // tuning key 18 < 0: the eight-plane kernel (A/B)
        sa.G = (nt + 15) / 16;
        RC_(launch_slices_k<1, 8, 8, 16, 16, 1, EPI_RESID>(sa, a.wf, st));
      } else if (spw == 16) {
        // o_proj (33 MB): two half passes of four planes over the same tiles, paired on one XCD so that the second reads the
        // tiles from L2 (dd_gemv_slices.h `halves`): 10.4 vs 13.5 us for the eight-plane kernel (the wide matrices lose with it)
        sa.G = g_exp_G[1] ? g_exp_G[1] : (nt + 15) / 16;
        sa.halves = 2;
        RC_(launch_slices_k<1, 4, 8, 16, 16, 1, EPI_RESID>(sa, a.wf, st));
        sa.halves = 1;
      }
      else if (spw == 43) RC_(launch_slices_k<1, 8, 8, 43, 8, 1, EPI_RESID>(sa, a.wf, st));
      else RC_(launch_slices_k<1, 8, 8, 56, 8, 1, EPI_RESID>(sa, a.wf, st));
      launch_finish<EPI_RESID, 1, 8, 8>(a, nt, st);
    } else {
      if (spw != 16) return SLICES_UNSUPPORTED;
      if (g_exp_G[2] >= 0 && 4 * ((nt + 23) / 24) <= 256) {
        // slice pairs, one slice resident at a time: half the partial sums.  Three tiles per wave, as evenly as the tile count
        // allows, in ONE round of workgroups (LLaVA-7B: 58 per pair = 232): 27.25 vs 27.75 ms per 32-lane step; 64 per pair
        // (2.7 tiles per wave: uneven) 28.6, 86 (two tiles, 1.3 rounds) 29.1.  Tuning key 19 < 0: single slices (A/B)
        sa.G = g_exp_G[2] ? g_exp_G[2] : (nt + 23) / 24;
        constexpr size_t smem = (size_t)16 * 8 * 1024;
        static bool attr = false;
        if (!attr) {
          DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_seq<8, 8, 16, 3, 0, EPI_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
          DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_seq<8, 8, 16, 3, 1, EPI_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
          attr = true;
        }
        DD_REQUIRE((nt + 8 * sa.G - 1) / (8 * sa.G) <= 3, "gemv_slices_seq: %d tiles over %d workgroups per pair", nt, sa.G);
        if (a.wf) k_gemv_slices_seq<8, 8, 16, 3, 1, EPI_SILU><<<4 * sa.G, GEMV_THREADS, smem, st>>>(sa);
        else k_gemv_slices_seq<8, 8, 16, 3, 0, EPI_SILU><<<4 * sa.G, GEMV_THREADS, smem, st>>>(sa);
        launch_finish<EPI_SILU, 2, 8, 4>(a, a.n_tiles, st);
      } else {
        sa.G = g_exp_G[2] > 0 ? g_exp_G[2] : (nt + 42) / 43;
        RC_(launch_slices_k<1, 8, 8, 16, 16, 1, EPI_SILU>(sa, a.wf, st));
        launch_finish<EPI_SILU, 2, 8, 8>(a, a.n_tiles, st);
      }
    }
    return DD_OK;
  } else if (epi == EPI_QKV) {
    if (spw != 16 || (nt & 1) || a.part_floats < need8) return SLICES_UNSUPPORTED;
    sa.n_groups = nt / 2;
    sa.G = sa.n_groups >= 256 ? (sa.n_groups + 15) / 16 : (sa.n_groups + 7) / 8;     // two tile pairs per wave when there are enough
    RC_(launch_slices_k<2, NG, 8, 16, 16, 1, EPI_QKV>(sa, a.wf, st));
    launch_finish<EPI_QKV, 1, NG, 8>(a, nt, st);
  } else if (epi == EPI_RESID) {
    // K = 4096 (o_proj): the wave-split kernel in one launch is as fast as slices + finish (13.5 vs 14.2 us at four planes,
    // 10.3 vs 11.2 at two: 33 MB of weights do not amortise a second launch); the long-K matrix (down) gains 30 %
    if (a.part_floats < need8 || spw == 16) return SLICES_UNSUPPORTED;
    sa.n_groups = nt;
    sa.G = (nt + 7) / 8;                                             // one tile per wave
    if (spw == 16) RC_(launch_slices_k<1, NG, 8, 16, 16, 1, EPI_RESID>(sa, a.wf, st));
    else if (spw == 43) RC_(launch_slices_k<1, NG, 8, 43, 16, 1, EPI_RESID>(sa, a.wf, st));
    else RC_(launch_slices_k<1, NG, 8, 56, 16, 1, EPI_RESID>(sa, a.wf, st));
    launch_finish<EPI_RESID, 1, NG, 8>(a, nt, st);
  } else {  // EPI_SILU: slice pairs, one workgroup per CU
    if (spw != 16 || a.part_floats < need4) return SLICES_UNSUPPORTED;
    sa.n_groups = nt;
    sa.G = 64;
    RC_(launch_slices_k<1, NG, 8, 16, 16, 2, EPI_SILU>(sa, a.wf, st));
    launch_finish<EPI_SILU, 2, NG, 4>(a, a.n_tiles, st);
  }
  return DD_OK;
}

int ddk_gemv_groups(int epi, const GemvArgs& a, hipStream_t st) {
  DD_REQUIRE(a.S % GEMV_WAVES == 0 && a.S >= GEMV_WAVES, "gemv_groups: K=%d must be a multiple of 256", a.S * 32);
  DD_REQUIRE(a.nb >= 1 && a.nb <= 8, "gemv_groups: nb=%d rows per group", a.nb);
  DD_REQUIRE(!a.fp8 || a.wscale, "gemv_groups: fp8 weights need row scales");
  DD_REQUIRE(a.n_groups == 2 || a.n_groups == 4 || a.n_groups == 8, "gemv_groups: %d groups (2, 4 or 8)", a.n_groups);
  if (g_gemv_slices && !a.fp8 && a.part) {
    int rs = a.n_groups == 2 ? try_slices<2>(epi, a, st) : (a.n_groups == 4 ? try_slices<4>(epi, a, st) : try_slices<8>(epi, a, st));
    if (rs != SLICES_UNSUPPORTED) {
      if (rs != DD_OK) return rs;
      DD_CHECK_LAUNCH();
      return DD_OK;
    }
  }
  if (a.n_groups == 8) {
    // no 64-row kernel for this matrix (lm_head, fp8 weights, other shapes): two 32-row passes over rows 0..31 / 32..63 —
    // the same bits, since a row's result does not depend on the kernel that computed it
    for (int half = 0; half < 2; ++half) {
      GemvArgs b = a;
      b.n_groups = 4;
      if (half) {
        b.xop = a.xop + (size_t)4 * a.S * 64;
        if (a.xop_next) b.xop_next = a.xop_next + (size_t)4 * a.S_next * 64;
        if (a.out) b.out = a.out + (size_t)32 * a.ldo;
        if (a.ssq_in) b.ssq_in = a.ssq_in + (size_t)32 * a.ssq_ld;
        if (a.ssq_out) b.ssq_out = a.ssq_out + (size_t)32 * a.ssq_ld;
        if (a.qbuf) b.qbuf = a.qbuf + (size_t)32 * a.q_dim;
        if (a.knew) b.knew = a.knew + (size_t)32 * a.kv_dim;
        if (a.vnew) b.vnew = a.vnew + (size_t)32 * a.kv_dim;
        for (int i = 0; i < 32; ++i) b.state_rows[i] = a.state_rows[32 + i];
        for (int i = 0; i < 4; ++i) b.out_g[i] = a.out_g[4 + i], b.knew_g[i] = a.knew_g[4 + i], b.vnew_g[i] = a.vnew_g[4 + i];
      }
      for (int i = 32; i < 64; ++i) b.state_rows[i] = nullptr;
      for (int i = 4; i < 8; ++i) b.out_g[i] = b.knew_g[i] = b.vnew_g[i] = nullptr;
      int rc2 = ddk_gemv_groups(epi, b, st);
      if (rc2 != DD_OK) return rc2;
    }
    return DD_OK;
  }
  int rc = DD_OK;
  const bool two = a.n_groups == 2;
  switch (epi) {
    case EPI_STORE: rc = two ? launch_gemv_groups<EPI_STORE, 1, 2>(a, st) : launch_gemv_groups<EPI_STORE, 1, 4>(a, st); break;
    case EPI_RESID: rc = two ? launch_gemv_groups<EPI_RESID, 1, 2>(a, st) : launch_gemv_groups<EPI_RESID, 1, 4>(a, st); break;
    case EPI_SILU: rc = two ? launch_gemv_groups<EPI_SILU, 2, 2>(a, st) : launch_gemv_groups<EPI_SILU, 2, 4>(a, st); break;
    case EPI_QKV:
      // four planes: two tiles per workgroup share the operand reads (33.0 vs 37.8 us at LLaVA-7B shapes); with two
      // planes one tile per workgroup is faster (24.6 vs 26.2 us: the extra workgroups matter more)
      if (two || (a.q_tiles & 1) || (a.k_tiles & 1) || (a.n_tiles & 1)) {
        rc = two ? launch_gemv_groups<EPI_QKV, 1, 2>(a, st) : launch_gemv_groups<EPI_QKV, 1, 4>(a, st);
      } else {
        GemvArgs b = a;
        b.n_tiles = a.n_tiles / 2;
        rc = launch_gemv_groups<EPI_QKV, 2, 4>(b, st);
      }
      break;
    default: DD_REQUIRE(false, "gemv_groups: unknown epilogue %d", epi);
  }
  if (rc != DD_OK) return rc;
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// ===============================================================================================
// decode attention: partial (one wave per kv head x 64-key split) + combine
// ===============================================================================================
#define ATT_SPLIT 64
#define ATT_MAX_SPLITS 160

// One workgroup (4 waves) per (kv head, 64-key tile).  Every wave requests its share of the K tile (8 of the 32
// 16-byte d-chunks, keys on lanes) AND of the V tile (16 keys, two per instruction) before anything else, so the
// whole 64 KiB tile is in flight at once and the kernel pays one memory latency, not 64.  Scores are reduced over
// the four waves through LDS, softmax statistics are per tile (flash-decoding), P.V partials are reduced the same way.
// GH = q heads of the GQA group handled by one workgroup (blockIdx.z picks the slice): 32 rows per workgroup (8 members
// x 4 heads) need 122 KiB of LDS and 156 VGPRs, i.e. one workgroup per CU; two slices of 16 rows run two per CU and
// read the K/V tile twice through L2.
// ML == 1 (lanes): NBT == 1 and GH == G; blockIdx.z is the ROW of the pass = the sequence whose cache this workgroup
// reads; results go to the 8-rows-per-head layout the 8-row combine reads.
// ML == 2 (groups): NBT == 8; blockIdx.z = group * (G / GH) + GQA slice; group g's 8 rows (members of sequence g) read
// that sequence's cache and drop bits; results go to an (8 * a.lane_groups)-rows-per-head layout (row = 8 * group + member).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// (fp32 cache; the fp16 cache goes through k_attn_partial16 below)
template <int NBT, int G, int GH, int ML = 0>
__global__ __launch_bounds__(256) void k_attn_partial(AttnDecodeArgs a) {
  constexpr int R = NBT * GH;        // rows of this workgroup
  const int lane_rows = a.n_lanes > 8 ? 16 : 8;   // ML == 1: rows per q head in the buffers (what the combine is built for)
  const int RT = ML == 1 ? lane_rows * G : (ML == 2 ? 8 * a.lane_groups * G : NBT * G);   // rows per kv head in the partial buffers
  // ML == 2 with NBT < 8: the 8 members of a group are split over 8 / NBT workgroups (member offset mo); each re-reads
  // the K/V tile through L2 but carries 1 / (8 / NBT) of the LDS and VALU work, which is what bounds the 8-row variant
  constexpr int MSPLIT = ML == 2 ? 8 / NBT : 1;
  const int zz = ML == 2 ? blockIdx.z % ((G / GH) * MSPLIT) : 0;
  const int g0 = ML == 1 ? 0 : (ML == 2 ? (zz / MSPLIT) * GH : blockIdx.z * GH);
  const int mo = ML == 2 ? (zz % MSPLIT) * NBT : 0;
  const int lane_row = ML == 1 ? blockIdx.z : (ML == 2 ? blockIdx.z / ((G / GH) * MSPLIT) : 0);
  // partial-buffer row of this workgroup's row r (= local head r / NBT, member r % NBT)
  auto buf_row = [&](int r) -> int {
    if (ML == 1) return r * lane_rows + lane_row;
    if (ML == 2) return (g0 + r / NBT) * 8 * a.lane_groups + lane_row * 8 + mo + r % NBT;
    return g0 * NBT + r;
  };
  extern __shared__ __align__(16) float att_sh[];
  float* q_sh = att_sh;                       // [R][128]
  float* s_part = q_sh + R * HEAD_DIM;        // [4][R][64]
  float* p_sh = s_part + 4 * R * ATT_SPLIT;   // [64][R]
  float* o_part = p_sh + ATT_SPLIT * R;       // [4][R][128]
  if (a.skip_if && *a.skip_if) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kvh = blockIdx.x, split = blockIdx.y;
  const int T = ML ? a.lane_state[lane_row]->T : (a.state ? a.state->T : a.T), t0 = split * ATT_SPLIT;
  if (t0 >= T) return;   // shorter lane / stale graph: this tile does not exist (the combine skips it as well)
  const float* kc_l = ML ? a.lane_kc[lane_row] : a.kc;
  const float* vc_l = ML ? a.lane_vc[lane_row] : a.vc;
  const uint8_t* bits_l = ML ? a.lane_bits[lane_row] : a.drop_bits;
  const int span0 = ML ? a.lane_span_start[lane_row] : a.span_start, spanL = ML ? a.lane_span_len[lane_row] : a.span_len;
  const int q_dim = a.n_heads * HEAD_DIM;
  const int nkeys = min(ATT_SPLIT, T - t0);
  const int half = lane >> 5, dq = lane & 31;

  // 1. all K / V requests of this wave (addresses clamped to the last live key; dead keys get p = 0)
  const int kt = t0 + min(lane, nkeys - 1);
  f32x4_t k4[8], v4[8];
  {
    const float* kbase = kc_l + (((size_t)kvh * 32 + wave * 8) * a.T_cap + kt) * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) k4[i] = *(const f32x4_t*)(kbase + (size_t)i * a.T_cap * 4);
    const float* vbase = vc_l + ((size_t)kvh * a.T_cap + t0) * HEAD_DIM + dq * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int key = min(wave * 16 + 2 * j + half, nkeys - 1);
      v4[j] = *(const f32x4_t*)(vbase + (size_t)key * HEAD_DIM);
    }
  }
  uint32_t bits = 0;
  if (bits_l && lane < nkeys) {
    int ka = t0 + lane;
    if (ka >= span0 && ka < span0 + spanL) bits = bits_l[ka - span0];
  }
  // 2. q rows (r = g*NBT + m) into LDS
  for (int i = tid; i < R * HEAD_DIM; i += 256) {
    int r = i / HEAD_DIM, d = i % HEAD_DIM, g = g0 + r / NBT;
    int m = ML == 1 ? lane_row : mo + r % NBT;            // row within its group (live if < nb)
    int qrow = ML == 2 ? lane_row * 8 + m : m;            // row of the pass
    q_sh[i] = (m < a.nb) ? a.qbuf[(size_t)qrow * q_dim + (kvh * G + g) * HEAD_DIM + d] : 0.f;
  }
  __syncthreads();
  // 3. partial scores over this wave's 32 d values
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float sp = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4_t q4 = *(const f32x4_t*)&q_sh[r * HEAD_DIM + (wave * 8 + i) * 4];
      sp += q4.x * k4[i].x + q4.y * k4[i].y + q4.z * k4[i].z + q4.w * k4[i].w;
    }
    s_part[(wave * R + r) * ATT_SPLIT + lane] = sp;
  }
  __syncthreads();
  // 4. softmax statistics of the tile: wave w owns rows r = w, w+4, ...
  const float scaling = 0.08838834764831845f;  // head_dim ** -0.5
  for (int r = wave; r < R; r += 4) {
    int m = ML == 1 ? 0 : mo + r % NBT;  // lanes: bit 0 of the sequence's own (leak) bits; groups: the member's bit
    float sv = (s_part[(0 * R + r) * ATT_SPLIT + lane] + s_part[(1 * R + r) * ATT_SPLIT + lane]) +
               (s_part[(2 * R + r) * ATT_SPLIT + lane] + s_part[(3 * R + r) * ATT_SPLIT + lane]);
    sv *= scaling;
    if (lane >= nkeys || ((bits >> (a.bit0 + m)) & 1u)) sv = -INFINITY;  // zero in the 2-D mask: weight exactly 0
    float mx = dd_wave_max(sv);
    float p = (sv == -INFINITY) ? 0.f : expf(sv - mx);
    float l = dd_wave_sum(p);
    p_sh[lane * R + r] = p;
    if (lane == 0) {
      float* ml = a.part_ml + (((size_t)kvh * gridDim.y + split) * RT + buf_row(r)) * 2;
      ml[0] = mx;
      ml[1] = l;
    }
  }
  __syncthreads();
  // 5. P.V over this wave's 16 keys (lanes 0-31 even keys, 32-63 odd keys; 4 consecutive d per lane)
  f32x4_t acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int key = wave * 16 + 2 * j + half;
    const float* pr = &p_sh[min(key, ATT_SPLIT - 1) * R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] += pr[r] * v4[j];   // p = 0 for dead / dropped keys
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    f32x4_t o = acc[r];
    o.x += __shfl_xor(o.x, 32);
    o.y += __shfl_xor(o.y, 32);
    o.z += __shfl_xor(o.z, 32);
    o.w += __shfl_xor(o.w, 32);
    if (half == 0) *(f32x4_t*)&o_part[(wave * R + r) * HEAD_DIM + dq * 4] = o;
  }
  __syncthreads();
  // 6. fixed-order sum over the four waves
  for (int i = tid; i < R * HEAD_DIM; i += 256) {
    float o = (o_part[i] + o_part[R * HEAD_DIM + i]) + (o_part[2 * R * HEAD_DIM + i] + o_part[3 * R * HEAD_DIM + i]);
    if (ML) {
      int r = i / HEAD_DIM, dd = i % HEAD_DIM;
      a.part_o[(((size_t)kvh * gridDim.y + split) * RT + buf_row(r)) * HEAD_DIM + dd] = o;
    } else {
      a.part_o[(((size_t)kvh * gridDim.y + split) * RT + g0 * NBT) * HEAD_DIM + i] = o;
    }
  }
}

// Merge of the tiles of one (head, row) + the row's own new key / value (each ensemble member attends to the shared prefix + ITS
// OWN new token) + hi/lo packing for o_proj: 128 threads (thread = output dimension d), shared by the stand-alone kernel
// (tiles from the partial buffers in memory) and by the all-tiles form of k_attn_partial16 (tiles still in LDS) — ONE body, with
// explicit fused multiply-adds, so that a row's bits do not depend on which of the two produced it.
// ld_ml(tile) -> (max, sum) of the tile for this row; ld_o(tile) -> its un-normalised output at dimension d.
// sh: 2 + ATT_MAX_SPLITS + 2 + 2 floats of shared memory for this group of 128 threads.
#define ATT_COMB_SH (2 + ATT_MAX_SPLITS + 2 + 2)
template <typename LD_ML, typename LD_O>
__device__ __forceinline__ void attn_combine_core(const AttnDecodeArgs& a, int head, int kvh, int m, bool wide, int d, int splits, float* sh,
                                                  LD_ML ld_ml, LD_O ld_o, bool store = true) {
  float* red = sh;
  float* w_sh = sh + 2;
  float* mx_sh = sh + 2 + ATT_MAX_SPLITS;
  float* den_sh = mx_sh + 2;
  const int lane = d & 63, wv = d >> 6;
  const int q_dim = a.n_heads * HEAD_DIM, kv_dim = a.n_kv * HEAD_DIM;
  const float scaling = 0.08838834764831845f;
  const int grp = wide ? m >> 3 : 0;                  // group of the row (multi-group passes)
  const float* knew_r = (wide && a.knew_g[grp]) ? a.knew_g[grp] + (size_t)(m & 7) * kv_dim : a.knew + (size_t)m * kv_dim;
  const float* vnew_r = (wide && a.vnew_g[grp]) ? a.vnew_g[grp] + (size_t)(m & 7) * kv_dim : a.vnew + (size_t)m * kv_dim;
  // every load of this block is issued here, before the first dependent use
  float qd = a.qbuf[(size_t)m * q_dim + head * HEAD_DIM + d];
  float kd = knew_r[kvh * HEAD_DIM + d];
  float vd = vnew_r[kvh * HEAD_DIM + d];
  float ms0 = -INFINITY, ls0 = 0.f, ms1 = -INFINITY, ls1 = 0.f;   // two tiles per thread: up to 256 tiles
  if (d < splits) ld_ml(d, ms0, ls0);
  if (d + 128 < splits) ld_ml(d + 128, ms1, ls1);
  float part = dd_wave_sum(qd * kd);
  float mloc = dd_wave_max(fmaxf(ms0, ms1));
  if (lane == 0) { red[wv] = part; mx_sh[wv] = mloc; }
  __syncthreads();
  float s_self = (red[0] + red[1]) * scaling;
  float M = fmaxf(s_self, fmaxf(mx_sh[0], mx_sh[1]));
  float w0 = (ms0 == -INFINITY) ? 0.f : expf(ms0 - M), w1 = (ms1 == -INFINITY) ? 0.f : expf(ms1 - M);
  if (d < splits) w_sh[d] = w0;
  if (d + 128 < splits) w_sh[d + 128] = w1;
  float dl = dd_wave_sum(__builtin_fmaf(w1, ls1, w0 * ls0));
  if (lane == 0) den_sh[wv] = dl;
  __syncthreads();
  float w_self = expf(s_self - M);
  float den = w_self + (den_sh[0] + den_sh[1]);
  float num = w_self * vd;
  for (int sp = 0; sp < splits; ++sp) num = __builtin_fmaf(w_sh[sp], ld_o(sp), num);
  if (!store) return;
  if (wide) xop_store16(a.xop_out, head * HEAD_DIM + d, m, num / den, q_dim >> 5, a.wf);
  else xop_store(a.xop_out, head * HEAD_DIM + d, m, num / den, a.wf);
}

// The same tile pass for the fp16 cache on the matrix cores.  The VALU form above costs ~1,100 vector instructions per wave and
// tile at 8 rows — the grouped decode attention ran at 2.3 TB/s of K/V bytes, bound by them, not by HBM.  Here
//   S^T = K . Q^T :  A = the 16-byte chunks of K as the cache stores them (16 keys x 32 d per fragment), B = the rows' q split
//                    into fp16 hi + lo columns (8 rows x {hi, lo} = the 16 columns; 2^-22 relative), wave w takes keys 16w..16w+15;
//   O^T = V^T . P^T: A = the cache's octets of V (16 d x 32 keys per fragment), B = the tile's probabilities as fp16 hi + lo,
//                    wave w takes output dimensions 32w..32w+31, so no cross-wave reduction of the outputs is needed.
// Tile softmax statistics, masks, buffers and row maps are those of k_attn_partial; all variants (rows alone, members of one
// sequence, lanes, groups) go through this one body, so a row's bits do not depend on the pass it rides in.
// FULL: the workgroup takes ALL tiles of its (kv head, sequence) — contexts of up to ATT_FULL_TILES tiles —, keeps the tiles'
// statistics and outputs in LDS instead of the partial buffers and runs the merge itself (attn_combine_core, the body the
// stand-alone k_attn_combine runs over the buffers): no partial-buffer round trip, no second launch, the same bits.
#define ATT_FULL_TILES 12
template <int NBT, int G, int GH, int ML, int FULL = 0>
__global__ __launch_bounds__(256) void k_attn_partial16(AttnDecodeArgs a) {
  constexpr int R = NBT * GH, RB = (R + 7) / 8, RP = RB * 8;
  extern __shared__ __align__(16) float full_sh[];        // FULL: [tiles][RP][HEAD_DIM] outputs, then [tiles][RP][2] statistics
  float* const fo_sh = full_sh;
  float* const fml_sh = full_sh + (size_t)ATT_FULL_TILES * RP * HEAD_DIM;
  __shared__ float comb_sh[FULL ? 2 : 1][FULL ? ATT_COMB_SH : 1];
  const int lane_rows = a.n_lanes > 8 ? 16 : 8;
  const int RT = ML == 1 ? lane_rows * G : (ML == 2 ? 8 * a.lane_groups * G : NBT * G);
  constexpr int MSPLIT = ML == 2 ? 8 / NBT : 1;
  const int zz = ML == 2 ? blockIdx.z % ((G / GH) * MSPLIT) : 0;
  const int g0 = ML == 1 ? 0 : (ML == 2 ? (zz / MSPLIT) * GH : blockIdx.z * GH);
  const int mo = ML == 2 ? (zz % MSPLIT) * NBT : 0;
  const int lane_row = ML == 1 ? blockIdx.z : (ML == 2 ? blockIdx.z / ((G / GH) * MSPLIT) : 0);
  auto buf_row = [&](int r) -> int {
    if (ML == 1) return r * lane_rows + lane_row;
    if (ML == 2) return (g0 + r / NBT) * 8 * a.lane_groups + lane_row * 8 + mo + r % NBT;
    return g0 * NBT + r;
  };
  __shared__ __align__(16) float q_sh[RP][HEAD_DIM + 4];
  __shared__ __align__(16) float p_sh[RP][ATT_SPLIT + 4];
  __shared__ float red_sh[2][4][RP];
  if (a.skip_if && *a.skip_if) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, h4 = lane >> 4;
  const int kvh = blockIdx.x;
  const int T = ML ? a.lane_state[lane_row]->T : (a.state ? a.state->T : a.T);
  // this workgroup's key tiles: tiles_per_wg consecutive ones (the launcher sizes the grid for ONE round of workgroups: with a
  // tile per workgroup the 8-sequence pass had 1.25 rounds, a quarter-full second one)
  const int tpw = FULL ? ATT_FULL_TILES : (a.tiles_per_wg > 0 ? a.tiles_per_wg : 1);
  const int split0 = blockIdx.y * tpw, n_live = (T + ATT_SPLIT - 1) / ATT_SPLIT;
  const int split1 = min(min(split0 + tpw, FULL ? ATT_FULL_TILES : a.splits_stride), n_live);
  if (split0 >= split1) return;   // shorter lane / stale graph: these tiles do not exist (the combine skips them as well)
  const dd_half* kc_l = (const dd_half*)(ML ? a.lane_kc[lane_row] : a.kc);
  const dd_half* vc_l = (const dd_half*)(ML ? a.lane_vc[lane_row] : a.vc);
  const uint8_t* bits_l = ML ? a.lane_bits[lane_row] : a.drop_bits;
  const int span0 = ML ? a.lane_span_start[lane_row] : a.span_start, spanL = ML ? a.lane_span_len[lane_row] : a.span_len;
  const int q_dim = a.n_heads * HEAD_DIM;

  // every K / V request of a tile at once (dead keys: the last live key's chunk / whatever the octet holds — cache memory is
  // zero-initialised and only ever holds finite values; their probabilities are exactly 0)
  auto load_tile = [&](int split, u32x4_t (&kf)[4], u32x4_t (&vf)[2][2], uint32_t (&kbits)[4]) {
    const int t0 = split * ATT_SPLIT, nkeys = min(ATT_SPLIT, T - t0);
    const int key = t0 + min(16 * wave + c, nkeys - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const u32x4_t*)(kc_l + (((size_t)kvh * 16 + 4 * ks + h4) * a.T_cap + key) * 8);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
        vf[dt][k2] = *(const u32x4_t*)(vc_l + (((size_t)kvh * (a.T_cap >> 3) + (t0 >> 3) + 4 * k2 + h4) * HEAD_DIM + 32 * wave + 16 * dt + c) * 8);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {              // drop bits of this lane's keys 16 w + 4 h4 + reg
      const int kk = 16 * wave + 4 * h4 + reg, ka = t0 + kk;
      kbits[reg] = (bits_l && kk < nkeys && ka >= span0 && ka < span0 + spanL) ? bits_l[ka - span0] : 0u;
    }
  };
  u32x4_t kf[2][4], vf[2][2][2];
  uint32_t kbits[2][4];
  load_tile(split0, kf[0], vf[0], kbits[0]);
  // q rows (r = g * NBT + m) into LDS, zero for rows past R or past the live members
  for (int i = tid; i < RP * HEAD_DIM; i += 256) {
    int r = i / HEAD_DIM, d = i % HEAD_DIM, g = g0 + r / NBT;
    int m = ML == 1 ? lane_row : mo + r % NBT;
    int qrow = ML == 2 ? lane_row * 8 + m : m;
    q_sh[r][d] = (r < R && m < a.nb) ? a.qbuf[(size_t)qrow * q_dim + (kvh * G + g) * HEAD_DIM + d] : 0.f;
  }
  __syncthreads();
  // fp32 x[8] -> this lane's B column: the fp16 hi part (columns 0-7) or the lo part (columns 8-15) of row c & 7
  auto split_col = [&](const float* x) -> u32x4_t {
    dd_f16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dd_half hi = (dd_half)x[j];
      float rem = x[j] - (float)hi;
      o[j] = c < 8 ? hi : (dd_half)rem;
    }
    return __builtin_bit_cast(u32x4_t, o);
  };
  // the rows' Q^T columns do not depend on the tile
  u32x4_t qb[RB][4];
#pragma unroll
  for (int blk = 0; blk < RB; ++blk) {
    const float* qr = &q_sh[8 * blk + (c & 7)][8 * h4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
      *(f32x4_t*)&x[0] = *(const f32x4_t*)(qr + 32 * ks);
      *(f32x4_t*)&x[4] = *(const f32x4_t*)(qr + 32 * ks + 4);
      qb[blk][ks] = split_col(x);
    }
  }
  const float scaling = 0.08838834764831845f;  // head_dim ** -0.5

  auto tile = [&](int split, const u32x4_t (&kfr)[4], const u32x4_t (&vfr)[2][2], const uint32_t (&kb)[4]) {
    const int t0 = split * ATT_SPLIT, nkeys = min(ATT_SPLIT, T - t0);
    float pmax[RB], sv[RB][4];
#pragma unroll
    for (int blk = 0; blk < RB; ++blk) {
      // S^T of this wave's 16 keys against block blk's 8 rows
      f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dd_f16x8_t, kfr[ks]), __builtin_bit_cast(dd_f16x8_t, qb[blk][ks]), sacc, 0, 0, 0);
      // hi + lo columns; mask; the wave's maximum per row (lanes c < 8 carry row 8 blk + c, keys 16 w + 4 h4 + reg)
      const int row = 8 * blk + (c & 7);
      const int m = ML == 1 ? 0 : mo + row % NBT;
      float mx = -INFINITY;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float s = (sacc[reg] + __shfl_down(sacc[reg], 8)) * scaling;
        const int kk = 16 * wave + 4 * h4 + reg;
        if (kk >= nkeys || ((kb[reg] >> (a.bit0 + m)) & 1u)) s = -INFINITY;   // zero in the 2-D mask: weight exactly 0
        sv[blk][reg] = s;
        mx = fmaxf(mx, s);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (h4 == 0 && c < 8) red_sh[0][wave][row] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < RB; ++blk) {
      const int row = 8 * blk + (c & 7);
      const float M = fmaxf(fmaxf(red_sh[0][0][row], red_sh[0][1][row]), fmaxf(red_sh[0][2][row], red_sh[0][3][row]));
      pmax[blk] = M;
      float l = 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float p = (sv[blk][reg] == -INFINITY) ? 0.f : expf(sv[blk][reg] - M);
        l += p;
        if (c < 8) p_sh[row][16 * wave + 4 * h4 + reg] = p;
      }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      if (h4 == 0 && c < 8) red_sh[1][wave][row] = l;
    }
    __syncthreads();
    if (wave == 0 && h4 == 0 && c < 8) {
#pragma unroll
      for (int blk = 0; blk < RB; ++blk) {
        const int row = 8 * blk + c;
        if (row < R) {
          float* ml = FULL ? fml_sh + ((size_t)split * RP + row) * 2
                           : a.part_ml + (((size_t)kvh * a.splits_stride + split) * RT + buf_row(row)) * 2;
          ml[0] = pmax[blk];
          ml[1] = (red_sh[1][0][row] + red_sh[1][1][row]) + (red_sh[1][2][row] + red_sh[1][3][row]);
        }
      }
    }
    // O^T of this wave's 32 output dimensions: P^T columns from LDS (all 64 keys), V^T fragments from the registers
#pragma unroll
    for (int blk = 0; blk < RB; ++blk) {
      u32x4_t pb[2];
      const float* pr = &p_sh[8 * blk + (c & 7)][8 * h4];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        float x[8];
        *(f32x4_t*)&x[0] = *(const f32x4_t*)(pr + 32 * k2);
        *(f32x4_t*)&x[4] = *(const f32x4_t*)(pr + 32 * k2 + 4);
        pb[k2] = split_col(x);
      }
      const int row = 8 * blk + c;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4_t oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
          oacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dd_f16x8_t, vfr[dt][k2]), __builtin_bit_cast(dd_f16x8_t, pb[k2]), oacc, 0, 0, 0);
        f32x4_t o;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) o[reg] = oacc[reg] + __shfl_down(oacc[reg], 8);
        if (c < 8 && row < R) {
          if constexpr (FULL) *(f32x4_t*)&fo_sh[((size_t)split * RP + row) * HEAD_DIM + 32 * wave + 16 * dt + 4 * h4] = o;
          else *(f32x4_t*)&a.part_o[(((size_t)kvh * a.splits_stride + split) * RT + buf_row(row)) * HEAD_DIM + 32 * wave + 16 * dt + 4 * h4] = o;
        }
      }
    }
  };
  // tiles in pairs with the other register set prefetching: the next tile's loads are in flight while this one is computed
  for (int sp = split0; sp < split1; sp += 2) {
    if (sp + 1 < split1) load_tile(sp + 1, kf[1], vf[1], kbits[1]);
    tile(sp, kf[0], vf[0], kbits[0]);
    if (sp + 1 < split1) {
      __syncthreads();                                  // p_sh / red_sh are reused
      if (sp + 2 < split1) load_tile(sp + 2, kf[0], vf[0], kbits[0]);
      tile(sp + 1, kf[1], vf[1], kbits[1]);
      if (sp + 2 < split1) __syncthreads();
    }
  }
  if constexpr (FULL) {
    // the merge, two rows at a time (threads 0-127 / 128-255 each run the 128-thread core on a row of their own)
    __syncthreads();
    const int half = tid >> 7, d = tid & 127;
    const bool wide = ML == 2 || (ML == 1 && a.n_lanes > 8);
#pragma unroll 1
    for (int it = 0; 2 * it < R; ++it) {
      const int rr = 2 * it + half, r = rr < R ? rr : R - 1;
      const int g = g0 + r / NBT;
      const int m = ML == 2 ? lane_row * 8 + mo + r % NBT : (ML == 1 ? lane_row : r % NBT);
      const bool store = rr < R && (ML != 0 || m < a.nb);
      attn_combine_core(
          a, kvh * G + g, kvh, m, wide, d, n_live, comb_sh[half],
          [&](int t, float& mx, float& l) { mx = fml_sh[((size_t)t * RP + r) * 2], l = fml_sh[((size_t)t * RP + r) * 2 + 1]; },
          [&](int t) -> float { return fo_sh[((size_t)t * RP + r) * HEAD_DIM + d]; }, store);
      __syncthreads();                                  // the core's scratch is reused by the next pair of rows
    }
  }
}

// grid (n_heads, nb), block 128 (thread = d): the merge above over the partial buffers in memory
template <int NBT, int G>
__global__ __launch_bounds__(HEAD_DIM) void k_attn_combine(AttnDecodeArgs a, int splits_grid) {
  constexpr int R = NBT * G;
  __shared__ float sh[ATT_COMB_SH];
  if (a.skip_if && *a.skip_if) return;
  const int head = blockIdx.x, m = blockIdx.y, d = threadIdx.x, kvh = head / G, g = head % G;
  const int r = g * NBT + m;
  // `splits_grid` is the stride of the partial buffers (tiles the partial kernel was launched with); a lane's row only
  // has the tiles of its own, possibly shorter, sequence
  const int splits = a.n_lanes ? (a.lane_state[a.lane_groups ? m >> 3 : m]->T + ATT_SPLIT - 1) / ATT_SPLIT
                               : (a.state ? (a.state->T + ATT_SPLIT - 1) / ATT_SPLIT : splits_grid);
  const float* mlb = a.part_ml + ((size_t)kvh * splits_grid * R + r) * 2;
  const size_t ml_stride = (size_t)R * 2;
  const float* po = a.part_o + ((size_t)kvh * splits_grid * R + r) * HEAD_DIM + d;
  const size_t o_stride = (size_t)R * HEAD_DIM;
  attn_combine_core(
      a, head, kvh, m, NBT > 8, d, splits, sh, [&](int t, float& mx, float& l) { mx = mlb[t * ml_stride], l = mlb[t * ml_stride + 1]; },
      [&](int t) -> float { return po[(size_t)t * o_stride]; });
}

// Key tiles the partial kernel is LAUNCHED with: the live count rounded up to a multiple of 4 (workgroups of tiles past
// the sequence end return at once), so that the launch shape — and with it a captured hipGraph — stays valid for 256
// more tokens instead of 64.  The combine takes the live count from the device-side length.
int ddk_attn_grid_tiles(int T, int T_cap) {
  int tiles = (T + ATT_SPLIT - 1) / ATT_SPLIT;
  int up = (tiles + 3) / 4 * 4, cap = T_cap / ATT_SPLIT;
  return up < cap ? up : (cap > tiles ? cap : tiles);
}

// k_attn_partial16: key tiles per workgroup (the next tile's loads travel while the current one is computed)
int g_attn16_tpw = 0;   // dd_set_tuning key 21: key tiles per workgroup of the fp16-cache decode attention (0: sized for one round)
static void attn16_grid(AttnDecodeArgs& b, int splits, int wg_per_tile) {
  b.splits_stride = splits;
  // up to 4 tiles per workgroup while at least ~1,000 workgroups remain (measured on the 32-lane step: 30.1 / 29.5 / 29.3 ms with
  // 1 / 2 / 4 tiles; a single sequence's 384 tile-workgroups stay one tile each)
  int tpw = (int)((long)wg_per_tile * splits / 1024);
  tpw = tpw < 1 ? 1 : (tpw > 4 ? 4 : tpw);
  b.tiles_per_wg = g_attn16_tpw > 0 ? g_attn16_tpw : tpw;
}
// all-tiles form (FULL) of k_attn_partial16: contexts of up to ATT_FULL_TILES tiles and enough (kv head, sequence) workgroups
int g_attn16_full = 1;   // dd_set_tuning key 22
static bool attn16_full_ok(int splits, int wgs) { return g_attn16_full && splits <= ATT_FULL_TILES && wgs >= 128; }
template <int NBT, int G, int GH, int ML>
static int launch_attn16_full(const AttnDecodeArgs& a, dim3 grid, hipStream_t st) {
  constexpr int R = NBT * GH, RP = (R + 7) / 8 * 8;
  constexpr size_t smem = (size_t)ATT_FULL_TILES * RP * (HEAD_DIM + 2) * sizeof(float);
  static bool attr = false;
  if (!attr && smem > 32 * 1024) {
    DD_HIP(hipFuncSetAttribute((const void*)k_attn_partial16<NBT, G, GH, ML, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  k_attn_partial16<NBT, G, GH, ML, 1><<<grid, 256, smem, st>>>(a);
  return DD_OK;
}
template <int NBT, int G>
static int launch_attn(const AttnDecodeArgs& a, hipStream_t st) {
  constexpr int GH = (NBT * G > 16) ? 2 : G;     // at most 16 rows per workgroup
  constexpr int R = NBT * GH;
  int splits = ddk_attn_grid_tiles(a.T, a.T_cap);
  DD_REQUIRE(splits >= 1 && splits <= ATT_MAX_SPLITS, "attn: %d key tiles unsupported (1..%d)", splits, ATT_MAX_SPLITS);
  size_t smem = (size_t)(R * HEAD_DIM + 4 * R * ATT_SPLIT + ATT_SPLIT * R + 4 * R * HEAD_DIM) * sizeof(float);
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    DD_HIP(hipFuncSetAttribute((const void*)k_attn_partial<NBT, G, GH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  if (a.kv16) {
    AttnDecodeArgs b = a;
    attn16_grid(b, splits, a.n_kv * (G / GH));
    k_attn_partial16<NBT, G, GH, 0><<<dim3(a.n_kv, (splits + b.tiles_per_wg - 1) / b.tiles_per_wg, G / GH), 256, 0, st>>>(b);
  } else {
    k_attn_partial<NBT, G, GH><<<dim3(a.n_kv, splits, G / GH), 256, smem, st>>>(a);
  }
  k_attn_combine<NBT, G><<<dim3(a.n_heads, a.nb), HEAD_DIM, 0, st>>>(a, splits);
  return DD_OK;
}

// fused base pass of up to 8 sequences: one single-query attention per row, each over its own cache
template <int G>
static int launch_attn_lanes(const AttnDecodeArgs& a, hipStream_t st) {
  constexpr int R = G;
  int splits = ddk_attn_grid_tiles(a.max_T, a.T_cap);
  DD_REQUIRE(splits >= 1 && splits <= ATT_MAX_SPLITS, "attn: %d key tiles unsupported (1..%d)", splits, ATT_MAX_SPLITS);
  size_t smem = (size_t)(R * HEAD_DIM + 4 * R * ATT_SPLIT + ATT_SPLIT * R + 4 * R * HEAD_DIM) * sizeof(float);
  if (a.kv16 && attn16_full_ok(splits, a.n_kv * a.n_lanes)) {      // all tiles per workgroup, merge included: no combine launch
    return launch_attn16_full<1, G, G, 1>(a, dim3(a.n_kv, 1, a.n_lanes), st);
  }
  if (a.kv16) {
    AttnDecodeArgs b = a;
    attn16_grid(b, splits, a.n_kv * a.n_lanes);
    k_attn_partial16<1, G, G, 1><<<dim3(a.n_kv, (splits + b.tiles_per_wg - 1) / b.tiles_per_wg, a.n_lanes), 256, 0, st>>>(b);
  } else {
    k_attn_partial<1, G, G, 1><<<dim3(a.n_kv, splits, a.n_lanes), 256, smem, st>>>(a);
  }
  if (a.n_lanes > 8) k_attn_combine<16, G><<<dim3(a.n_heads, a.nb), HEAD_DIM, 0, st>>>(a, splits);
  else k_attn_combine<8, G><<<dim3(a.n_heads, a.nb), HEAD_DIM, 0, st>>>(a, splits);
  return DD_OK;
}

// multi-group pass: members of NG sequences (8 rows each), every group over its own cache
static int g_attn_msplit = 1;   // workgroups per group of 8 members in the grouped decode attention (1, 2 or 4; dd_set_tuning key 10)
void ddk_set_attn_split(int v) { g_attn_msplit = v; }

template <int G, int NG, int NBT>
static int launch_attn_groups_n(const AttnDecodeArgs& a, hipStream_t st) {
  constexpr int GH = (NBT * G > 16) ? 2 : G;
  constexpr int R = NBT * GH;
  int splits = ddk_attn_grid_tiles(a.max_T, a.T_cap);
  DD_REQUIRE(splits >= 1 && splits <= ATT_MAX_SPLITS, "attn: %d key tiles unsupported (1..%d)", splits, ATT_MAX_SPLITS);
  size_t smem = (size_t)(R * HEAD_DIM + 4 * R * ATT_SPLIT + ATT_SPLIT * R + 4 * R * HEAD_DIM) * sizeof(float);
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    DD_HIP(hipFuncSetAttribute((const void*)k_attn_partial<NBT, G, GH, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  if (a.kv16 && NBT == 8 && g_attn16_full >= 2 && attn16_full_ok(splits, a.n_kv * NG * (G / GH))) {   // key 22 = 2: measured no faster (33 vs 26 + 6 us: one workgroup per CU walks ten tiles in a row)
    return launch_attn16_full<NBT, G, GH, 2>(a, dim3(a.n_kv, 1, NG * (G / GH)), st);
  }
  if (a.kv16) {
    AttnDecodeArgs b = a;
    attn16_grid(b, splits, a.n_kv * NG * (G / GH) * (8 / NBT));
    k_attn_partial16<NBT, G, GH, 2><<<dim3(a.n_kv, (splits + b.tiles_per_wg - 1) / b.tiles_per_wg, NG * (G / GH) * (8 / NBT)), 256, 0, st>>>(b);
  } else {
    k_attn_partial<NBT, G, GH, 2><<<dim3(a.n_kv, splits, NG * (G / GH) * (8 / NBT)), 256, smem, st>>>(a);
  }
  k_attn_combine<8 * NG, G><<<dim3(a.n_heads, 8 * NG), HEAD_DIM, 0, st>>>(a, splits);
  return DD_OK;
}
// multi-group pass: members of NG sequences (8 rows each), every group over its own cache
template <int G, int NG>
static int launch_attn_groups(const AttnDecodeArgs& a, hipStream_t st) {
  if (g_attn_msplit == 4) return launch_attn_groups_n<G, NG, 2>(a, st);
  if (g_attn_msplit == 2) return launch_attn_groups_n<G, NG, 4>(a, st);
  return launch_attn_groups_n<G, NG, 8>(a, st);
}

int ddk_attn_decode(const AttnDecodeArgs& a, hipStream_t st) {
  DD_REQUIRE(a.n_heads % a.n_kv == 0, "attn: heads %d not a multiple of kv heads %d", a.n_heads, a.n_kv);
  int G = a.n_heads / a.n_kv;
  DD_REQUIRE(G == 1 || G == 2 || G == 4, "attn: GQA group %d unsupported (1, 2, 4)", G);
  int rc = DD_OK;                    // a launcher that refuses (too many key tiles, attribute failure) launches nothing
  if (a.n_lanes > 0 && a.lane_groups) {
    DD_REQUIRE((a.lane_groups == 2 || a.lane_groups == 4 || a.lane_groups == 8) && a.n_lanes == a.lane_groups && a.nb >= 1 && a.nb <= 8,
               "attn: a multi-group pass takes 2, 4 or 8 sequences of up to 8 members");
    if (a.lane_groups == 8) {
      if (G == 1) rc = launch_attn_groups<1, 8>(a, st);
      else if (G == 2) rc = launch_attn_groups<2, 8>(a, st);
      else rc = launch_attn_groups<4, 8>(a, st);
    } else if (a.lane_groups == 2) {
      if (G == 1) rc = launch_attn_groups<1, 2>(a, st);
      else if (G == 2) rc = launch_attn_groups<2, 2>(a, st);
      else rc = launch_attn_groups<4, 2>(a, st);
    } else {
      if (G == 1) rc = launch_attn_groups<1, 4>(a, st);
      else if (G == 2) rc = launch_attn_groups<2, 4>(a, st);
      else rc = launch_attn_groups<4, 4>(a, st);
    }
    if (rc != DD_OK) return rc;
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
  if (a.n_lanes > 0) {
    DD_REQUIRE(a.n_lanes <= 16 && a.nb == a.n_lanes, "attn: %d lanes for %d rows", a.n_lanes, a.nb);
    if (G == 1) rc = launch_attn_lanes<1>(a, st);
    else if (G == 2) rc = launch_attn_lanes<2>(a, st);
    else rc = launch_attn_lanes<4>(a, st);
    if (rc != DD_OK) return rc;
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
  bool one = a.nb == 1;
  if (G == 1) rc = one ? launch_attn<1, 1>(a, st) : launch_attn<8, 1>(a, st);
  else if (G == 2) rc = one ? launch_attn<1, 2>(a, st) : launch_attn<8, 2>(a, st);
  else rc = one ? launch_attn<1, 4>(a, st) : launch_attn<8, 4>(a, st);
  if (rc != DD_OK) return rc;
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// ===============================================================================================
// prefill
// ===============================================================================================
// y = w * (x * rsqrt(mean(x^2) + eps)) in HF's op order, written as hi/lo bf16 planes (and optionally fp32)
__global__ __launch_bounds__(256) void k_rmsnorm_split(const float* __restrict__ x, int d, const float* __restrict__ w,
                                                       float eps, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                       const int32_t* __restrict__ row_index, float* normed, int wf) {
  __shared__ float sh[4];
  int row = blockIdx.x;
  const float* xr = x + (size_t)(row_index ? row_index[row] : row) * d;
  float ss = 0.f;
  for (int i = threadIdx.x; i < d; i += 256) ss += xr[i] * xr[i];
  ss = dd_wave_sum(ss);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
  __syncthreads();
  float rstd = 1.0f / sqrtf((sh[0] + sh[1] + sh[2] + sh[3]) / (float)d + eps);
  const int S = d >> 5;
  for (int i8 = threadIdx.x * 8; i8 < d; i8 += 256 * 8) {   // 8 consecutive k per thread -> one 16-byte packed store
    u32x4_t vh, vl;
    uint32_t hh[8], ll[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float y = w[i8 + j] * (xr[i8 + j] * rstd);
      dd_split(y, hh[j], ll[j], wf);
      if (normed) normed[(size_t)row * d + i8 + j] = y;
    }
    if (hi) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vh[j] = hh[2 * j] | (hh[2 * j + 1] << 16);
        vl[j] = ll[2 * j] | (ll[2 * j + 1] << 16);
      }
      size_t o = apack_off(row, i8, S);
      *(u32x4_t*)(hi + o) = vh;
      *(u32x4_t*)(lo + o) = vl;
    }
  }
}
int ddk_rmsnorm_split(const float* x, int M, int d, const float* w, float eps, uint16_t* hi, uint16_t* lo,
                      const int32_t* row_index, float* normed, hipStream_t st, int wf) {
  k_rmsnorm_split<<<M, 256, 0, st>>>(x, d, w, eps, hi, lo, row_index, normed, wf);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int ddk_final_norm_rows(const float* x, int rows, int d, const float* w, float eps, float* out, hipStream_t st) {
  return ddk_rmsnorm_split(x, rows, d, w, eps, nullptr, nullptr, nullptr, out, st, 0);
}

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1 (indices stay constants whatever the body's size)
template <int B, int E, typename F>
__device__ __forceinline__ void dd_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    dd_static_for<B + 1, E>(f);
  }
}
// Epilogue of the prefill GEMMs for a wave's MI x NJ accumulator tiles (rows m_base.., 16-column tiles nt_base..).
// D[m][n]: m = 4*(lane>>4) + reg, n = lane & 15
template <int EPI, int MI, int NJ, int WF>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4_t (&acc)[MI][NJ], const int m_base, const int nt_base,
                                              const bool (&wv)[NJ], const int lane) {
  const int c = lane & 15;
  // loop order row (i, reg) outside, column tile inside: what depends on the row only (its sequence, position, cache) is
  // computed 16 times per wave, not once per element.  The three loops are expanded at compile time (dd_static_for), not left to
  // `#pragma unroll`: past the compiler's unroll budget (the 4 x 8-tile kernel with a large body: RoPE, erf, ViT scatter) a
  // loop stays a loop, the accumulators are then indexed dynamically and the WHOLE array lives in scratch — in the main loop too
  dd_static_for<0, MI>([&](auto ic_) {
    constexpr int i = decltype(ic_)::value;
    dd_static_for<0, 4>([&](auto rc_) {
      constexpr int reg = decltype(rc_)::value;
      const int row = m_base + i * 16 + 4 * (lane >> 4) + reg;
      int lrow = row, live_rows = a.M;
      float *kc_r = a.kc, *vc_r = a.vc;
      if (EPI == EPI_QKV && a.seq_rows) {     // several sequences back to back: the row's own sequence, position, liveness and cache
        const int sq = min(row / a.seq_rows, 31);
        lrow = row - sq * a.seq_rows, live_rows = a.seq_tab->T[sq];
        kc_r = a.seq_tab->kc[sq] + a.seq_off_k, vc_r = a.seq_tab->vc[sq] + a.seq_off_v;
      }
      const int pos_c = a.pos0 + max(0, min(lrow, live_rows - 1));   // EPI_QKV: clamped position (rotary table row)
      dd_static_for<0, NJ>([&](auto jc_) {
        constexpr int j = decltype(jc_)::value;
        const int nt = nt_base + j;
        const bool ok = row < a.M && wv[j];
        float y = acc[i][j][reg];
        if (a.wscale) y *= a.wscale[(size_t)(wv[j] ? nt : 0) * 16 + c];
        if (a.bias && EPI != EPI_SILU && EPI != EPI_QKV) y += a.bias[(wv[j] ? nt : 0) * 16 + c];
        if (EPI == EPI_STORE) {
          int col = nt * 16 + c;
          if (ok && col < a.n_valid) a.out[(size_t)row * a.ldo + col] = y;
        } else if (EPI == EPI_RESID) {
          int col = nt * 16 + c;
          if (ok) a.out[(size_t)row * a.ldo + col] += y;
        } else if (EPI == EPI_SILU) {
          if constexpr ((j & 1) == 0 && j + 1 < NJ) {
            float u = acc[i][j + 1][reg];
            if (a.wscale) u *= a.wscale[(size_t)(nt + 1) * 16 + c];
            float act = y / (1.0f + expf(-y));
            uint32_t h, l;
            dd_split(act * u, h, l, WF);
            int col = (nt >> 1) * 16 + c;
            if (ok) {
              size_t o = apack_off(row, col, a.ld_planes >> 5);
              a.o_hi[o] = (uint16_t)h;
              a.o_lo[o] = (uint16_t)l;
            }
          }
        } else if (EPI == EPI_ACT) {
          float v = y;
          if (a.act == 0) v = y / (1.0f + expf(-1.702f * y));                        // quick_gelu: x * sigmoid(1.702 x)
          else if (a.act == 1) v = 0.5f * y * (1.0f + erff(y * 0.70710678118654752f));  // gelu (erf form)
          uint32_t h, l;
          dd_split(v, h, l, WF);
          if (ok) {
            size_t o = apack_off(row, nt * 16 + c, a.ld_planes >> 5);
            a.o_hi[o] = (uint16_t)h;
            a.o_lo[o] = (uint16_t)l;
          }
        } else if (EPI == EPI_QKV_VIT) {
          if (ok) {
            // hp: head pitch of the q / K^T / V buffers (= hd, or hd padded to a multiple of 32 for the matrix-core attention:
            // EVA ViT-g's 88 -> 96; the pad columns stay zero)
            int col = nt * 16 + c + a.vit_col0, hd = a.vit_head_dim, hp = a.vit_head_pad ? a.vit_head_pad : hd;
            int trow = row;                    // token within its image
            float *kt_i = a.kc, *v_i = a.vc;
            bool live = true;
            if (a.vit_img_rows) {
              const int im = row / a.vit_img_rows;
              trow = row - im * a.vit_img_rows, live = trow < a.vit_T;
              kt_i = a.kc + (size_t)im * a.vit_k_stride, v_i = a.vc + (size_t)im * a.vit_v_stride;
            }
            if (!live) {
            } else if (col < a.vit_hidden) {
              a.qbuf[(size_t)row * (a.vit_hidden / hd * hp) + (col / hd) * hp + col % hd] = y * a.vit_qscale;
            } else if (col < 2 * a.vit_hidden) {
              int cc = col - a.vit_hidden, head = cc / hd, idx = cc % hd;
              kt_i[(((size_t)head * (hp >> 2) + (idx >> 2)) * a.T_cap + trow) * 4 + (idx & 3)] = y;
            } else {
              int cc = col - 2 * a.vit_hidden, head = cc / hd, idx = cc % hd;
              v_i[((size_t)head * a.T_cap + trow) * hp + idx] = y;
            }
          }
        } else {  // EPI_QKV
          float yp = __shfl_xor(y, 8);  // partner column c ^ 8 of the same row
          const bool okr = ok && lrow < live_rows;
          if (nt < a.q_tiles + a.k_tiles) {
            bool is_q = nt < a.q_tiles;
            int ht = is_q ? nt : nt - a.q_tiles;
            int head = ht >> 3, f = (ht & 7) * 8 + (c & 7);
            const int pos = pos_c;
            float cs = a.rope_cos[(size_t)pos * ROPE_HALF + f], sn = a.rope_sin[(size_t)pos * ROPE_HALF + f];
            float o = (c < 8) ? __fadd_rn(__fmul_rn(y, cs), __fmul_rn(-yp, sn)) : __fadd_rn(__fmul_rn(y, cs), __fmul_rn(yp, sn));
            int idx = (c < 8) ? f : ROPE_HALF + f;
            if (okr) {
              if (is_q) a.qbuf[(size_t)row * a.q_dim + head * HEAD_DIM + idx] = o;
              else dd_kv_store(kc_r, vc_r, a.kv16, head, idx, pos, a.T_cap, true, o);
            }
          } else if (okr) {
            int col = (nt - a.q_tiles - a.k_tiles) * 16 + c;
            int kvh = col / HEAD_DIM, idx = col % HEAD_DIM;
            int pos = a.pos0 + lrow;
            dd_kv_store(kc_r, vc_r, a.kv16, kvh, idx, pos, a.T_cap, false, y);
          }
        }
      });
    });
  });
}

// C[M][N] = (A_hi + A_lo)[M][K] . W^T, block 128x128, 4 waves (2x2) of 64x64; both operands are pre-tiled so every
// fragment is a contiguous 1 KiB wave load straight to VGPRs (L2-resident A, streamed W)

// MI x NJ = 16x16 MFMA tiles per wave (rows x cols); 4 waves as 2x2: block = (32*MI) rows x (32*NJ) cols.
// 4x4 (128x128 block) for the LM prefill; 2x2 (64x64) when the grid would otherwise be too small to fill 256 CUs
// (the ViT: M = 577, N = 1024).
template <int EPI, int MI, int NJ, int WF = 0>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware block order (1-D grid of gx * gy workgroups).  The dispatcher deals workgroup ids round-robin to the 8 XCDs,
  // each with its own L2: id -> (XCD id & 7, slot id >> 3).  An XCD is given a contiguous run of the virtual order
  // v = column block * gy + row block, so the gy row blocks that re-read one weight column block run back to back on ONE
  // XCD (one fetch into one L2) and neighbouring column blocks share the activation rows in that L2.
  int bx, by;
  {
    const int gy = a.grid_y, total = gridDim.x, per = total >> 3, id = blockIdx.x;
    const int v = a.xcd_order && id < (per << 3) ? (id & 7) * per + (id >> 3) : id;
    bx = v / gy, by = v - bx * gy;
  }
  const int m_base = by * (32 * MI) + wr * (16 * MI);
  const int nt_base = bx * (2 * NJ) + wc * NJ;
  const int S = a.S;
  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const u32x4_t* pa_hi[MI];
  const u32x4_t* pa_lo[MI];
  const int m_tiles = (a.M + 15) >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int mt = min((m_base >> 4) + i, m_tiles - 1);
    pa_hi[i] = (const u32x4_t*)a.a_hi + (size_t)mt * S * 64 + lane;
    pa_lo[i] = (const u32x4_t*)a.a_lo + (size_t)mt * S * 64 + lane;
  }
  const u32x4_t* pw[NJ];
  bool wv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    wv[j] = (nt_base + j) < a.n_tiles;
    pw[j] = a.W + ((size_t)(wv[j] ? nt_base + j : 0) * S) * 64 + lane;
  }
  // explicit two-stage register pipeline: the fragments of k-step s+1 are requested before the 32 MFMAs of step s
  // issue, so the L2 latency of one step hides behind the matrix work of the other (S is even: K multiple of 256)
  u32x4_t ahi0[MI], alo0[MI], w0[NJ], ahi1[MI], alo1[MI], w1[NJ];
  auto load = [&](u32x4_t* ahi, u32x4_t* alo, u32x4_t* w, int ks) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      ahi[i] = pa_hi[i][(size_t)ks * 64];
      alo[i] = pa_lo[i][(size_t)ks * 64];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) w[j] = pw[j][(size_t)ks * 64];
  };
  auto compute = [&](const u32x4_t* ahi, const u32x4_t* alo, const u32x4_t* w) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[i][j] = dd_mfma16<WF>(ahi[i], w[j], acc[i][j]);
        acc[i][j] = dd_mfma16<WF>(alo[i], w[j], acc[i][j]);
      }
  };
  load(ahi0, alo0, w0, 0);
  for (int ks = 0; ks < S; ks += 2) {
    load(ahi1, alo1, w1, ks + 1);
    compute(ahi0, alo0, w0);
    if (ks + 2 < S) load(ahi0, alo0, w0, ks + 2);
    compute(ahi1, alo1, w1);
  }
  gemm_epilogue<EPI, MI, NJ, WF>(a, acc, m_base, nt_base, wv, lane);
}

// The same product with a 128 x 512 block for many rows (a long prompt, or the prompts of several sequences back to back):
// 8 waves as 2 x 4, each 64 rows x 128 columns (4 x 8 accumulator tiles).  The 128 x 128 kernel asks the L2 for
// (128 x {hi, lo} + 128) x 64 B = 24 KiB per k-step of 32 and is bound by that (85 flop / B: DESIGN.md); this block asks for
// (128 x 2 + 512) x 64 B = 48 KiB for four times the flops, staged once per workgroup in LDS (the pre-tiled fragments are
// copied as they lie: a wave's fragment read is 64 consecutive 16-byte words, conflict-free) in a ring of three stages.
// Per accumulator tile the MFMA sequence is the one k_gemm issues (k ascending; hi then lo), so the results are the same bits.
#define GB_MT 8     // 16-row tiles of a block
#define GB_NT 32    // 16-column tiles of a block
#define GB_STAGE (GB_MT * 2 + GB_NT)   // 1 KiB fragments per k-step: A hi, A lo, W
template <int EPI, int WF>
__global__ __launch_bounds__(512) void k_gemm_big(GemmArgs a) {
  extern __shared__ __align__(16) u32x4_t gb_sh[];          // [3 stages][GB_STAGE fragments][64]
  constexpr int MI = 4, NJ = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  int bx, by;
  {
    const int gy = a.grid_y, total = gridDim.x, per = total >> 3, id = blockIdx.x;
    const int v = a.xcd_order && id < (per << 3) ? (id & 7) * per + (id >> 3) : id;
    bx = v / gy, by = v - bx * gy;
  }
  const int S = a.S;
  const int m_tiles = (a.M + 15) >> 4;
  // copy duty of this wave: fragment slots wave + 8 i, i < 6 -> A hi tile `wave`, A lo tile `wave`, W tiles wave + 8 (i - 2)
  const u32x4_t* src[6];
  {
    const int mt = min(by * GB_MT + wave, m_tiles - 1);
    src[0] = (const u32x4_t*)a.a_hi + (size_t)mt * S * 64 + lane;
    src[1] = (const u32x4_t*)a.a_lo + (size_t)mt * S * 64 + lane;
#pragma unroll
    for (int i = 2; i < 6; ++i) {
      const int nt = min(bx * GB_NT + wave + 8 * (i - 2), a.n_tiles - 1);
      src[i] = a.W + (size_t)nt * S * 64 + lane;
    }
  }
  u32x4_t* const my_dst = gb_sh + wave * 64 + lane;             // + stage * GB_STAGE * 64 + slot group i: A hi 0..7, A lo 8..15, W 16..47
  const int m_base = by * (16 * GB_MT) + wr * (16 * MI);
  const int nt_base = bx * GB_NT + wc * NJ;
  bool wv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) wv[j] = (nt_base + j) < a.n_tiles;
  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // three LDS stages: step ks multiplies from stage ks % 3 while the fragments of step ks + 2 travel global -> registers ->
  // stage (ks + 2) % 3, and the wave's A fragments of step ks + 1 are read from stage (ks + 1) % 3 once the step's MFMAs are
  // issued — so that after the barrier the next step starts multiplying at once instead of all 8 waves queueing on the LDS
  // for their 9 KiB first (that start-up cost 40 % of a step with two stages).
  u32x4_t pre[6];
#pragma unroll
  for (int s0 = 0; s0 < 2; ++s0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) pre[i] = src[i][(size_t)s0 * 64];
#pragma unroll
    for (int i = 0; i < 6; ++i) my_dst[s0 * (GB_STAGE * 64) + i * 8 * 64] = pre[i];
  }
  __syncthreads();
  u32x4_t ahi[MI], alo[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    ahi[i] = gb_sh[(wr * MI + i) * 64 + lane];
    alo[i] = gb_sh[(GB_MT + wr * MI + i) * 64 + lane];
  }
  int s_cur = 0;                                               // ks % 3
  for (int ks = 0; ks < S; ++ks) {
    const bool more2 = ks + 2 < S;
    if (more2) {
#pragma unroll
      for (int i = 0; i < 6; ++i) pre[i] = src[i][(size_t)(ks + 2) * 64];
    }
    const u32x4_t* st = gb_sh + s_cur * (GB_STAGE * 64);
    const int s_next = s_cur == 2 ? 0 : s_cur + 1, s_next2 = s_next == 2 ? 0 : s_next + 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const u32x4_t w = st[(2 * GB_MT + wc * NJ + j) * 64 + lane];
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = dd_mfma16<WF>(ahi[i], w, acc[i][j]);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = dd_mfma16<WF>(alo[i], w, acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < S) {
      const u32x4_t* sn = gb_sh + s_next * (GB_STAGE * 64);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        ahi[i] = sn[(wr * MI + i) * 64 + lane];
        alo[i] = sn[(GB_MT + wr * MI + i) * 64 + lane];
      }
    }
    if (more2) {
      u32x4_t* d = my_dst + s_next2 * (GB_STAGE * 64);
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i * 8 * 64] = pre[i];
    }
    __syncthreads();
    s_cur = s_next;
  }
  gemm_epilogue<EPI, MI, NJ, WF>(a, acc, m_base, nt_base, wv, lane);
}

static int g_gemm_big_rows = 1024;   // tuning key 16: rows from which ddk_gemm uses the 128 x 512 block (0: never)
void ddk_set_gemm_big_rows(int v) { g_gemm_big_rows = v; }
static int launch_gemm_big(int epi, const GemmArgs& a_, hipStream_t st);

static int g_gemm_xcd_order = 1;   // tuning key 15
void ddk_set_gemm_xcd_order(int v) { g_gemm_xcd_order = v ? 1 : 0; }
template <int MI, int NJ>
static int launch_gemm(int epi, const GemmArgs& a_, hipStream_t st) {
  GemmArgs a = a_;
  const int gx = (a.n_tiles + 2 * NJ - 1) / (2 * NJ);
  a.grid_y = (a.M + 32 * MI - 1) / (32 * MI);
  a.xcd_order = g_gemm_xcd_order && a.grid_y <= 8;   // many row blocks: the column-major runs thrash the L2 with activations (measured: 150 vs 125 ms at 2960 rows)
  dim3 grid(gx * a.grid_y);
  switch (epi) {
#define GM(E_)                                                          \
  if (a.wf) k_gemm<E_, MI, NJ, 1><<<grid, 256, 0, st>>>(a);             \
  else k_gemm<E_, MI, NJ, 0><<<grid, 256, 0, st>>>(a)
    case EPI_STORE: GM(EPI_STORE); break;
    case EPI_RESID: GM(EPI_RESID); break;
    case EPI_SILU: GM(EPI_SILU); break;
    case EPI_QKV: GM(EPI_QKV); break;
#undef GM
    case EPI_ACT: k_gemm<EPI_ACT, MI, NJ><<<grid, 256, 0, st>>>(a); break;
    case EPI_QKV_VIT: k_gemm<EPI_QKV_VIT, MI, NJ><<<grid, 256, 0, st>>>(a); break;
    default: DD_REQUIRE(false, "gemm: unknown epilogue %d", epi);
  }
  DD_CHECK_LAUNCH();
  return DD_OK;
}

static int launch_gemm_big(int epi, const GemmArgs& a_, hipStream_t st) {
  GemmArgs a = a_;
  const int gx = (a.n_tiles + GB_NT - 1) / GB_NT;
  a.grid_y = (a.M + 16 * GB_MT - 1) / (16 * GB_MT);
  a.xcd_order = g_gemm_xcd_order && a.grid_y <= 8;   // many row blocks: the column-major runs thrash the L2 with activations (measured: 150 vs 125 ms at 2960 rows)
  const size_t lds = (size_t)3 * GB_STAGE * 64 * sizeof(u32x4_t);   // 144 KiB
  dim3 grid(gx * a.grid_y);
#define GBK(E_, W_)                                                                                                          \
  do {                                                                                                                       \
    static bool attr = false;                                                                                                \
    if (!attr) {                                                                                                             \
      DD_HIP(hipFuncSetAttribute((const void*)k_gemm_big<E_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
      attr = true;                                                                                                           \
    }                                                                                                                        \
    k_gemm_big<E_, W_><<<grid, 512, lds, st>>>(a);                                                                           \
  } while (0)
#define GB(E_)                  \
  if (a.wf) GBK(E_, 1);         \
  else GBK(E_, 0)
  switch (epi) {
    case EPI_STORE: GB(EPI_STORE); break;
    case EPI_RESID: GB(EPI_RESID); break;
    case EPI_SILU: GB(EPI_SILU); break;
    case EPI_QKV: GB(EPI_QKV); break;
    case EPI_ACT: GBK(EPI_ACT, 0); break;
    case EPI_QKV_VIT: GBK(EPI_QKV_VIT, 0); break;
    default: DD_REQUIRE(false, "gemm (128 x 512 block): epilogue %d not built", epi);
  }
#undef GB
#undef GBK
  DD_CHECK_LAUNCH();
  return DD_OK;
}

int ddk_gemm(int epi, const GemmArgs& a, hipStream_t st) {
  DD_REQUIRE(a.S >= 2 && (a.S & 1) == 0, "gemm: K=%d must be a multiple of 64", a.S * 32);
  if (g_gemm_big_rows > 0 && a.M >= g_gemm_big_rows &&
      (epi == EPI_STORE || epi == EPI_RESID || epi == EPI_SILU || epi == EPI_QKV || ((epi == EPI_ACT || epi == EPI_QKV_VIT) && !a.wf)))
    return launch_gemm_big(epi, a, st);
  long big = (long)((a.n_tiles + 7) / 8) * ((a.M + 127) / 128);      // workgroups of the 128x128 tiling
  if (big >= 150) return launch_gemm<4, 4>(epi, a, st);
  return launch_gemm<2, 2>(epi, a, st);                              // 64x64 blocks: 4x the workgroups
}

// causal prefill attention, fp32 VALU; keys lane-parallel from the transposed K cache
// Each wave owns QR = 4 consecutive query rows, so every K / V tile it loads (L2-resident) is reused 4 times:
// the one-row-per-wave version was bound by L2 bandwidth (5.9 GB of tile re-reads per layer at T = 608).
#define PF_QR 4
template <int G, int XOP = 0>
__global__ __launch_bounds__(256) void k_attn_prefill(const float* __restrict__ qbuf, const float* __restrict__ kc,
                                                      const float* __restrict__ vc, int T, int T_cap, int n_heads,
                                                      uint16_t* __restrict__ o_hi, uint16_t* __restrict__ o_lo,
                                                      const uint8_t* __restrict__ drop_plane, int drop_bit,
                                                      int span_start, int span_len, int q0, u32x4_t* __restrict__ xop_out) {
  // q0: position of query row 0 (chunked prefill: rows q0 .. q0 + T - 1 attend to keys 0 .. their own position; the keys
  // before q0 are already in the cache).  T = number of query rows of this call.
  __shared__ __align__(16) float q_sh[4][PF_QR][HEAD_DIM];
  __shared__ __align__(16) float p_sh[4][ATT_SPLIT][PF_QR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = blockIdx.x, kvh = head / G;
  // groups of PF_QR rows are aligned to ABSOLUTE positions (a multiple of PF_QR), so a row is processed with the same
  // three neighbours whether it arrives in a full prefill or in a later chunk: the two give bit-identical outputs
  const int t_first = (blockIdx.y * 4 + wave) * PF_QR - (q0 & (PF_QR - 1));   // local rows t_first .. t_first + 3 (may start < 0)
  const int q_dim = n_heads * HEAD_DIM;
  if (t_first >= T) return;                                       // whole wave idle (no block-level barrier below)
  const int t_last = min(t_first + PF_QR - 1, T - 1);
  const int p_last = q0 + t_last;                                 // last key position this wave needs
  for (int i = lane; i < PF_QR * HEAD_DIM; i += 64) {
    int r = i / HEAD_DIM, dd = i % HEAD_DIM;
    q_sh[wave][r][dd] = qbuf[(size_t)max(0, min(t_first + r, T - 1)) * q_dim + head * HEAD_DIM + dd];
  }
  __builtin_amdgcn_wave_barrier();
  const float scaling = 0.08838834764831845f;
  const int half = lane >> 5, dq = lane & 31;
  float m_run[PF_QR], l_run[PF_QR];
  f32x4_t acc[PF_QR];
#pragma unroll
  for (int r = 0; r < PF_QR; ++r) {
    m_run[r] = -INFINITY;
    l_run[r] = 0.f;
    acc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  for (int t0 = 0; t0 <= p_last; t0 += ATT_SPLIT) {
    int kt = t0 + lane;
    const float* kb = kc + ((size_t)kvh * 32 * T_cap + min(kt, p_last)) * 4;
    // a zero column of the member's 2-D attention mask (first-token ensemble: llava.py:336-359 run on the prompt)
    bool key_dropped = false;
    if (drop_plane && kt >= span_start && kt < span_start + span_len)
      key_dropped = (drop_plane[kt - span_start] >> drop_bit) & 1;
    float s[PF_QR];
#pragma unroll
    for (int r = 0; r < PF_QR; ++r) s[r] = 0.f;
#pragma unroll 8
    for (int d4 = 0; d4 < 32; ++d4) {
      f32x4_t k4 = *(const f32x4_t*)(kb + (size_t)d4 * T_cap * 4);
#pragma unroll
      for (int r = 0; r < PF_QR; ++r) {
        f32x4_t q4 = *(const f32x4_t*)&q_sh[wave][r][d4 * 4];
        s[r] += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
      }
    }
#pragma unroll
    for (int r = 0; r < PF_QR; ++r) {
      bool valid = kt <= q0 + max(0, min(t_first + r, T - 1)) && !key_dropped;  // causal: the row at position p attends keys 0..p
      float sv = valid ? s[r] * scaling : -INFINITY;
      float m_new = fmaxf(m_run[r], dd_wave_max(sv));
      float p = valid ? expf(sv - m_new) : 0.f;
      float corr = (m_run[r] == -INFINITY) ? 0.f : expf(m_run[r] - m_new);
      l_run[r] = l_run[r] * corr + dd_wave_sum(p);
      acc[r] *= corr;
      m_run[r] = m_new;
      p_sh[wave][lane][r] = p;
    }
    __builtin_amdgcn_wave_barrier();
    int nkeys = min(ATT_SPLIT, p_last + 1 - t0);
    const float* vb = vc + ((size_t)kvh * T_cap + t0) * HEAD_DIM + dq * 4;
    for (int kp = 0; 2 * kp < nkeys; ++kp) {
      int key = 2 * kp + half;
      if (key < nkeys) {
        f32x4_t v4 = *(const f32x4_t*)(vb + (size_t)key * HEAD_DIM);
        f32x4_t p4 = *(const f32x4_t*)&p_sh[wave][key][0];
        acc[0] += p4.x * v4;
        acc[1] += p4.y * v4;
        acc[2] += p4.z * v4;
        acc[3] += p4.w * v4;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int r = 0; r < PF_QR; ++r) {
    f32x4_t a = acc[r];
    a.x += __shfl_xor(a.x, 32);
    a.y += __shfl_xor(a.y, 32);
    a.z += __shfl_xor(a.z, 32);
    a.w += __shfl_xor(a.w, 32);
    const int t = t_first + r;
    if (t >= 0 && t < T && half == 0) {
      float inv = 1.0f / l_run[r];
      uint32_t hh[4], ll[4];
      if (XOP) {                  // rows feed the decode GEMV next (short chunks): its packed operand planes instead
#pragma unroll
        for (int j = 0; j < 4; ++j) xop_store16(xop_out, head * HEAD_DIM + dq * 4 + j, t, a[j] * inv, q_dim >> 5);
        continue;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) dd_split_hl(a[j] * inv, hh[j], ll[j]);
      size_t o = apack_off(t, head * HEAD_DIM + dq * 4, q_dim >> 5);      // 4 consecutive k: one 8-byte packed store
      *(u32x2_t*)(o_hi + o) = (u32x2_t){hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16)};
      *(u32x2_t*)(o_lo + o) = (u32x2_t){ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16)};
    }
  }
}


// -----------------------------------------------------------------------------------------------
// prefill attention on the matrix cores.  fp32-grade through bf16 MFMA: every operand is split x = hi + lo (bf16 each) and
// a product a.b is taken as a_hi.b_hi + a_lo.b_hi + a_hi.b_lo (the lo.lo term is below fp32 rounding).
//   S^T = K . Q^T   (A = K tile [16 keys x 32 d], B = Q^T [32 d x 16 queries]):  D lane l = query l & 15, keys 4 (l >> 4) + r
//   O^T = V^T . P^T (A = V^T [16 d x 32 key slots], B = P^T [32 key slots x 16 queries])
// A wave owns 16 queries (columns of every D), so the online-softmax statistics and the rescale of O^T are per LANE; the
// 32 key slots of a PV step are ordered so that the 8 probabilities a lane group already holds (4 keys of each of the two
// S^T tiles) ARE its B operand — no transpose.  K and V tiles of 32 keys are staged in LDS once per workgroup (4 waves =
// 64 queries).  Query groups are aligned to absolute positions, extra (masked) key steps add exact zeros, so a row's
// output does not depend on the call it arrives in (chunked prefill).
// -----------------------------------------------------------------------------------------------
#define FA_KEYS 32
__device__ __forceinline__ void fa_split8(const float* v, u32x4_t& hi, u32x4_t& lo) {
  uint32_t h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dd_split_hl(v[j], h[j], l[j]);
  hi = (u32x4_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
  lo = (u32x4_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
}
#define FA_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, A), __builtin_bit_cast(bf16x8_t, B), C, 0, 0, 0)

// HD = head dimension (128: the LM; 64: the CLIP tower, bidirectional, q pre-scaled)
template <int G, int HD = 128, int KV16 = 0>
__global__ __launch_bounds__(256) void k_attn_prefill_mfma(const float* qbuf, const float* kc,
                                                           const float* vc, int T, int T_cap, int n_heads,
                                                           uint16_t* o_hi, uint16_t* o_lo,
                                                           const uint8_t* __restrict__ drop_plane, int drop_bit,
                                                           int span_start, int span_len, int q0, int causal, float scaling, int wf,
                                                           int Tk, const SeqTab* tab = nullptr, int seq_rows = 0, size_t off_k = 0,
                                                           size_t off_v = 0) {
  // Tk: number of keys when not causal (cross-attention: T queries against Tk keys of another sequence; = T for self-attention)
  // tab: the prompts of several sequences in one launch (dd_lm_prefill_group): blockIdx.z = sequence; its length and cache
  // bases come from the table, its q rows / output planes start at row blockIdx.z * seq_rows of the batch's buffers
  if (tab) {
    const int sq = blockIdx.z;
    T = tab->T[sq];
    if ((int)blockIdx.y * 64 >= T) return;
    kc = tab->kc[sq] + off_k, vc = tab->vc[sq] + off_v;
    const size_t r0 = (size_t)sq * seq_rows * (n_heads * HD);
    qbuf += r0, o_hi += r0, o_lo += r0;
    Tk = T;
  }
  constexpr int LD = HD + 4;   // padded row pitch (floats) of the staged K / V tiles: keeps the V^T reads conflict-free
  constexpr int KS = HD / 32, DT = HD / 16, C4 = HD / 4;
  __shared__ __align__(16) float Ksh[FA_KEYS * LD];
  __shared__ __align__(16) float Vsh[FA_KEYS * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x, kvh = head / G;
  const int q_dim = n_heads * HD;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int shift = q0 & 15;
  const int blk_first = blockIdx.y * 64 - shift;             // local row of the workgroup's first query (may be < 0)
  const int t_q = blk_first + wave * 16 + c16;                // this lane's query (local row), same for its 4 lane groups
  const bool q_live = t_q >= 0 && t_q < T;
  const int pos_q = q0 + max(0, min(t_q, T - 1));             // its absolute position
  const int blk_last = min(blk_first + 63, T - 1);
  if (blk_last < 0) return;
  const int p_max = causal ? q0 + blk_last : Tk - 1;          // last key any query of the workgroup attends to
  const int wave_pmax = causal ? q0 + min(blk_first + wave * 16 + 15, T - 1) : Tk - 1;  // ... of this wave
  const bool wave_live = blk_first + wave * 16 < T && blk_first + wave * 16 + 15 >= 0;

  // Q^T operands of the lane: B[k = d 8 g4 .. +8][j = query c16], four 32-d steps, hi and lo
  u32x4_t qh[KS], ql[KS];
  {
    const float* qr = qbuf + (size_t)max(0, min(t_q, T - 1)) * q_dim + head * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[8];
      *(f32x4_t*)&v[0] = *(const f32x4_t*)(qr + ks * 32 + g4 * 8);
      *(f32x4_t*)&v[4] = *(const f32x4_t*)(qr + ks * 32 + g4 * 8 + 4);
      fa_split8(v, qh[ks], ql[ks]);
    }
  }
  f32x4_t acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int t0 = 0; t0 <= p_max; t0 += FA_KEYS) {
    __syncthreads();                                          // the previous tiles are no longer being read
    // stage K (from the transposed cache [d/4][T_cap][4]) and V ([T_cap][128]) of keys t0 .. t0+31, clamped to p_max
    if constexpr (!KV16) {
      for (int i = tid; i < FA_KEYS * C4; i += 256) {
        int kk = i & 31, c = i >> 5;                            // K: 32 keys x C4 d-chunks of 4 (keys contiguous in the cache)
        int key = min(t0 + kk, p_max);
        *(f32x4_t*)&Ksh[kk * LD + c * 4] = *(const f32x4_t*)(kc + (((size_t)kvh * C4 + c) * T_cap + key) * 4);
        int d4 = i % C4, k2 = i / C4;                            // V: 32 keys x C4 float4 of a row
        int key2 = min(t0 + k2, p_max);
        *(f32x4_t*)&Vsh[k2 * LD + d4 * 4] = *(const f32x4_t*)(vc + ((size_t)kvh * T_cap + key2) * HD + d4 * 4);
      }
    } else {
      // fp16 cache (dd_lm_kernels.h layouts): K chunks of 8 d, V octets of keys per d; expanded to fp32 in the staged tiles
      constexpr int C8 = HD / 8;
      for (int i = tid; i < FA_KEYS * C8; i += 256) {
        int kk = i & 31, c = i >> 5;
        int key = min(t0 + kk, p_max);
        const f16x8_t kh = *(const f16x8_t*)((const dd_half*)kc + (((size_t)kvh * C8 + c) * T_cap + key) * 8);
        *(f32x4_t*)&Ksh[kk * LD + c * 8] = (f32x4_t){(float)kh[0], (float)kh[1], (float)kh[2], (float)kh[3]};
        *(f32x4_t*)&Ksh[kk * LD + c * 8 + 4] = (f32x4_t){(float)kh[4], (float)kh[5], (float)kh[6], (float)kh[7]};
      }
      for (int i = tid; i < (FA_KEYS / 8) * HD; i += 256) {
        int dd = i % HD, oc = i / HD;                          // octet oc = keys t0 + 8 oc .. + 7 (t0 is a multiple of 32) of dimension dd
        int octet = min((t0 >> 3) + oc, p_max >> 3);           // keys past p_max inside the last octet carry weight 0
        const f16x8_t vh = *(const f16x8_t*)((const dd_half*)vc + (((size_t)kvh * (T_cap >> 3) + octet) * HD + dd) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) Vsh[(8 * oc + j) * LD + dd] = (float)vh[j];
      }
    }
    __syncthreads();
    if (!wave_live || t0 > wave_pmax) continue;               // nothing for this wave in these keys (barriers above stay matched)

    // S^T for the two 16-key tiles
    float sv[8];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
      const float* kr = &Ksh[(kt * 16 + c16) * LD + g4 * 8];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float v[8];
        *(f32x4_t*)&v[0] = *(const f32x4_t*)(kr + ks * 32);
        *(f32x4_t*)&v[4] = *(const f32x4_t*)(kr + ks * 32 + 4);
        u32x4_t kh, kl;
        fa_split8(v, kh, kl);
        sacc = FA_MFMA(kh, qh[ks], sacc);
        sacc = FA_MFMA(kl, qh[ks], sacc);
        sacc = FA_MFMA(kh, ql[ks], sacc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = t0 + kt * 16 + g4 * 4 + r;
        bool ok = causal ? key <= pos_q : key < Tk;
        if (ok && drop_plane && key >= span_start && key < span_start + span_len) ok = !((drop_plane[key - span_start] >> drop_bit) & 1);
        sv[kt * 4 + r] = ok ? sacc[r] * scaling : -INFINITY;
      }
    }
    // online softmax of this lane's query over the 32 keys (8 in this lane, the rest in lanes ^16, ^32)
    float mx = sv[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sv[j]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    float p[8], ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p[j] = (sv[j] == -INFINITY) ? 0.f : expf(sv[j] - m_new);
      ps += p[j];
    }
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    const float corr = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    l_run = l_run * corr + ps;
    m_run = m_new;
    u32x4_t ph, pl;
    fa_split8(p, ph, pl);
    // O^T += V^T . P^T: lane group g4 holds key slots {4 g4 + j, 16 + 4 g4 + j}, j < 4, of this step
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      float v[8];
      const float* vr = &Vsh[dt * 16 + c16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = vr[(4 * g4 + j) * LD];
        v[4 + j] = vr[(16 + 4 * g4 + j) * LD];
      }
      u32x4_t vh, vl;
      fa_split8(v, vh, vl);
      f32x4_t a = acc[dt] * corr;
      a = FA_MFMA(vh, ph, a);
      a = FA_MFMA(vl, ph, a);
      a = FA_MFMA(vh, pl, a);
      acc[dt] = a;
    }
  }
  if (!q_live) return;
  const float inv = 1.0f / l_run;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    uint32_t hh[4], ll[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dd_split(acc[dt][r] * inv, hh[r], ll[r], wf);
    size_t o = apack_off(t_q, head * HD + dt * 16 + g4 * 4, q_dim >> 5);   // 4 consecutive k: one 8-byte packed store
    *(u32x2_t*)(o_hi + o) = (u32x2_t){hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16)};
    *(u32x2_t*)(o_lo + o) = (u32x2_t){ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16)};
  }
}

// bidirectional attention of the CLIP tower (head_dim 64, q pre-scaled by the QKV epilogue), same kernel
int ddk_attn_vit_mfma(const float* q, const float* kt, const float* v, int T, int Tc, int n_heads, uint16_t* o_hi, uint16_t* o_lo,
                      hipStream_t st, int head_pitch, int Tk, float scaling) {
  DD_REQUIRE(head_pitch == 64 || head_pitch == 96, "attn_vit: head pitch %d (64, or 96 for 88-wide heads)", head_pitch);
  if (Tk <= 0) Tk = T;
  if (head_pitch == 96)
    k_attn_prefill_mfma<1, 96><<<dim3(n_heads, (T + 63) / 64), 256, 0, st>>>(q, kt, v, T, Tc, n_heads, o_hi, o_lo, nullptr, 0, 0, 0, 0,
                                                                            0, scaling, 0, Tk);
  else
    k_attn_prefill_mfma<1, 64><<<dim3(n_heads, (T + 63) / 64), 256, 0, st>>>(q, kt, v, T, Tc, n_heads, o_hi, o_lo, nullptr, 0, 0, 0, 0,
                                                                          0, scaling, 0, Tk);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

__global__ void k_put_seq_tab(SeqTab tab, SeqTab* dst) {
  const int i = threadIdx.x;
  dst->T[i] = tab.T[i], dst->kc[i] = tab.kc[i], dst->vc[i] = tab.vc[i];
}
int ddk_put_seq_tab(const SeqTab& tab, SeqTab* dev, hipStream_t st) {
  k_put_seq_tab<<<1, 32, 0, st>>>(tab, dev);          // by value through the launch: no host buffer has to outlive the call
  DD_CHECK_LAUNCH();
  return DD_OK;
}
// bidirectional attention of n images in one launch (blockIdx.z = image): q rows / output planes of image i start at row
// i * img_rows, its K^T / V blocks come from the table (SeqTab::kc / vc)
int ddk_attn_vit_mfma_batch(const float* q, const SeqTab* tab, int n, int img_rows, int T, int Tc, int n_heads, uint16_t* o_hi, uint16_t* o_lo,
                            hipStream_t st, int head_pitch) {
  DD_REQUIRE(head_pitch == 64 || head_pitch == 96, "attn_vit: head pitch %d (64, or 96 for 88-wide heads)", head_pitch);
  dim3 grid(n_heads, (T + 63) / 64, n);
  if (head_pitch == 96)
    k_attn_prefill_mfma<1, 96><<<grid, 256, 0, st>>>(q, nullptr, nullptr, T, Tc, n_heads, o_hi, o_lo, nullptr, 0, 0, 0, 0, 0, 1.0f, 0, T, tab, img_rows, 0, 0);
  else
    k_attn_prefill_mfma<1, 64><<<grid, 256, 0, st>>>(q, nullptr, nullptr, T, Tc, n_heads, o_hi, o_lo, nullptr, 0, 0, 0, 0, 0, 1.0f, 0, T, tab, img_rows, 0, 0);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int ddk_prefill_mfma_enabled();
static int g_prefill_mfma = 1;   // dd_set_tuning key 12: prefill attention on the matrix cores (0: the VALU kernel)
void ddk_set_prefill_mfma(int on) { g_prefill_mfma = on; }
int ddk_prefill_mfma_enabled() { return g_prefill_mfma; }

int ddk_attn_prefill(const float* qbuf, const float* kc, const float* vc, int T, int T_cap, int n_heads, int n_kv,
                     uint16_t* o_hi, uint16_t* o_lo, const uint8_t* drop_plane, int drop_bit, int span_start,
                     int span_len, int q0, hipStream_t st, u32x4_t* xop_out, int kv16, int wf) {
  int G = n_heads / n_kv;
  DD_REQUIRE(!wf || (g_prefill_mfma && !xop_out), "attn_prefill: fp16-weight engines use the matrix-core prefill attention only");
  DD_REQUIRE(!kv16 || (g_prefill_mfma && !xop_out), "attn_prefill: the fp16 KV cache is read by the matrix-core prefill attention only");
  dim3 grid(n_heads, (T + (q0 & (PF_QR - 1)) + 4 * PF_QR - 1) / (4 * PF_QR));
#define PF_ARGS qbuf, kc, vc, T, T_cap, n_heads, o_hi, o_lo, drop_plane, drop_bit, span_start, span_len, q0, xop_out
  if (xop_out) {
    if (G == 1) k_attn_prefill<1, 1><<<grid, 256, 0, st>>>(PF_ARGS);
    else if (G == 2) k_attn_prefill<2, 1><<<grid, 256, 0, st>>>(PF_ARGS);
    else if (G == 4) k_attn_prefill<4, 1><<<grid, 256, 0, st>>>(PF_ARGS);
    else DD_REQUIRE(false, "attn_prefill: GQA group %d unsupported", G);
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
  if (g_prefill_mfma) {
    dim3 g2(n_heads, (T + (q0 & 15) + 63) / 64);
#define FA_ARGS qbuf, kc, vc, T, T_cap, n_heads, o_hi, o_lo, drop_plane, drop_bit, span_start, span_len, q0, 1, 0.08838834764831845f, wf, T
    if (kv16) {
      if (G == 1) k_attn_prefill_mfma<1, 128, 1><<<g2, 256, 0, st>>>(FA_ARGS);
      else if (G == 2) k_attn_prefill_mfma<2, 128, 1><<<g2, 256, 0, st>>>(FA_ARGS);
      else if (G == 4) k_attn_prefill_mfma<4, 128, 1><<<g2, 256, 0, st>>>(FA_ARGS);
      else DD_REQUIRE(false, "attn_prefill: GQA group %d unsupported", G);
    } else if (G == 1) k_attn_prefill_mfma<1><<<g2, 256, 0, st>>>(FA_ARGS);
    else if (G == 2) k_attn_prefill_mfma<2><<<g2, 256, 0, st>>>(FA_ARGS);
    else if (G == 4) k_attn_prefill_mfma<4><<<g2, 256, 0, st>>>(FA_ARGS);
    else DD_REQUIRE(false, "attn_prefill: GQA group %d unsupported", G);
#undef FA_ARGS
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
  if (G == 1) k_attn_prefill<1><<<grid, 256, 0, st>>>(PF_ARGS);
  else if (G == 2) k_attn_prefill<2><<<grid, 256, 0, st>>>(PF_ARGS);
  else if (G == 4) k_attn_prefill<4><<<grid, 256, 0, st>>>(PF_ARGS);
  else DD_REQUIRE(false, "attn_prefill: GQA group %d unsupported", G);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// causal prefill attention of n sequences in ONE launch (the prompts of a batch: dd_lm_prefill_group): one sequence's
// 608 rows are 320 workgroups — about one per CU — and n launches would run one after another
int ddk_attn_prefill_seqs(const float* qbuf, const SeqTab* tab, size_t off_k, size_t off_v, int n, int seq_rows, int max_T, int T_cap,
                          int n_heads, int n_kv, uint16_t* o_hi, uint16_t* o_lo, hipStream_t st, int kv16, int wf) {
  const int G = n_heads / n_kv;
  DD_REQUIRE(g_prefill_mfma, "attn_prefill_seqs: the matrix-core prefill attention is switched off");
  dim3 g2(n_heads, (max_T + 63) / 64, n);
#define FS_ARGS qbuf, nullptr, nullptr, max_T, T_cap, n_heads, o_hi, o_lo, nullptr, 0, 0, 0, 0, 1, 0.08838834764831845f, wf, max_T, tab, seq_rows, off_k, off_v
  if (kv16) {
    if (G == 1) k_attn_prefill_mfma<1, 128, 1><<<g2, 256, 0, st>>>(FS_ARGS);
    else if (G == 2) k_attn_prefill_mfma<2, 128, 1><<<g2, 256, 0, st>>>(FS_ARGS);
    else if (G == 4) k_attn_prefill_mfma<4, 128, 1><<<g2, 256, 0, st>>>(FS_ARGS);
    else DD_REQUIRE(false, "attn_prefill_seqs: GQA group %d unsupported", G);
  } else if (G == 1) k_attn_prefill_mfma<1><<<g2, 256, 0, st>>>(FS_ARGS);
  else if (G == 2) k_attn_prefill_mfma<2><<<g2, 256, 0, st>>>(FS_ARGS);
  else if (G == 4) k_attn_prefill_mfma<4><<<g2, 256, 0, st>>>(FS_ARGS);
  else DD_REQUIRE(false, "attn_prefill_seqs: GQA group %d unsupported", G);
#undef FS_ARGS
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// ===============================================================================================
// glue
// ===============================================================================================
__global__ __launch_bounds__(1024) void k_embed_rows(const uint16_t* __restrict__ embed, int d, const DDState* state,
                                                     float* __restrict__ x, const float* __restrict__ normw,
                                                     u32x4_t* __restrict__ xop, float* __restrict__ ssq, int ssq_ld,
                                                     const int32_t* __restrict__ skip_if, int wf) {
  __shared__ float sh[16];
  if (skip_if && *skip_if) return;
  int tok = state->cur_tok;
  float ss = 0.f;
  for (int i = threadIdx.x; i < d; i += 1024) {
    float e = dd_w16_to_f32(embed[(size_t)tok * d + i], wf);
    ss += e * e;
    float z = normw[i] * e;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      x[(size_t)m * d + i] = e;
      xop_store(xop, i, m, z, wf);
    }
  }
  ss = dd_wave_sum(ss);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x < 8) {
    float v = 0.f;
    for (int i = 0; i < 16; ++i) v += sh[i];
    ssq[(size_t)threadIdx.x * ssq_ld] = v;
  }
}
int ddk_embed_rows(const uint16_t* embed, int d, const DDState* state, float* x, const float* normw, u32x4_t* xop,
                   float* ssq, int ssq_ld, hipStream_t st, const int32_t* skip_if, int wf) {
  k_embed_rows<<<1, 1024, 0, st>>>(embed, d, state, x, normw, xop, ssq, ssq_ld, skip_if, wf);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
// row m embeds the current token of sequence lanes.state[m] (null: a zero row): one workgroup per row (the rows are
// independent; a row's sum of squares is reduced exactly as the one-workgroup kernel did it — thread t over columns t,
// t + 1024, ..., wave sums, the 16 waves in order — so its bits do not depend on how many rows a pass has).
// planes: rows 8g..8g+7 go to operand plane g (rows > 8), or all to the 8-row operand (rows == 8)
__global__ __launch_bounds__(1024) void k_embed_rows_lanes(const uint16_t* __restrict__ embed, int d, EmbedLanes lanes, int rows,
                                                           float* __restrict__ x, const float* __restrict__ normw,
                                                           u32x4_t* __restrict__ xop, float* __restrict__ ssq, int ssq_ld, int wf) {
  __shared__ float sh[16];
  const int m = blockIdx.x;
  const DDState* sp = lanes.state[m];
  const int tok = sp ? sp->cur_tok : -1;
  float ss = 0.f;
  for (int i = threadIdx.x; i < d; i += 1024) {
    float w = normw[i];
    float e = tok >= 0 ? dd_w16_to_f32(embed[(size_t)tok * d + i], wf) : 0.f;
    ss += e * e;
    x[(size_t)m * d + i] = e;
    if (rows > 8) xop_store16(xop, i, m, w * e, d >> 5, wf);
    else xop_store(xop, i, m, w * e, wf);
  }
  float v = dd_wave_sum(ss);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += sh[i];
    ssq[(size_t)m * ssq_ld] = t;
  }
}
int ddk_embed_rows_lanes(const uint16_t* embed, int d, const EmbedLanes& lanes, int rows, float* x, const float* normw,
                         u32x4_t* xop, float* ssq, int ssq_ld, hipStream_t st, int wf) {
  DD_REQUIRE(rows == 8 || rows == 16 || rows == 32 || rows == 64, "embed_rows_lanes: %d rows (8, 16, 32 or 64)", rows);
  k_embed_rows_lanes<<<rows, 1024, 0, st>>>(embed, d, lanes, rows, x, normw, xop, ssq, ssq_ld, wf);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
// short prompt chunks through the decode GEMVs: n (<= ROWS) rows of fp32 embeddings become the residual rows x[ROWS][d]
// (rows >= n zero), the packed operand planes of the first GEMV (w_norm * x) and the per-row sums of squares
template <int ROWS>
__global__ __launch_bounds__(1024) void k_pack_embed_rows(const float* __restrict__ rows, int n, int d, float* __restrict__ x,
                                                          const float* __restrict__ normw, u32x4_t* __restrict__ xop,
                                                          float* __restrict__ ssq, int ssq_ld, int wf) {
  __shared__ float sh[ROWS][16];
  float ss[ROWS];
#pragma unroll
  for (int m = 0; m < ROWS; ++m) ss[m] = 0.f;
  for (int i = threadIdx.x; i < d; i += 1024) {
    float w = normw[i];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      float e = m < n ? rows[(size_t)m * d + i] : 0.f;
      ss[m] += e * e;
      x[(size_t)m * d + i] = e;
      if (ROWS > 8) xop_store16(xop, i, m, w * e, d >> 5, wf);
      else xop_store(xop, i, m, w * e, wf);
    }
  }
#pragma unroll
  for (int m = 0; m < ROWS; ++m) {
    float v = dd_wave_sum(ss[m]);
    if ((threadIdx.x & 63) == 0) sh[m][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < ROWS) {
    float v = 0.f;
    for (int i = 0; i < 16; ++i) v += sh[threadIdx.x][i];
    ssq[(size_t)threadIdx.x * ssq_ld] = v;
  }
}
int ddk_pack_embed_rows(const float* rows, int n, int rows_cap, int d, float* x, const float* normw, u32x4_t* xop, float* ssq,
                        int ssq_ld, hipStream_t st, int wf) {
  if (rows_cap == 32) k_pack_embed_rows<32><<<1, 1024, 0, st>>>(rows, n, d, x, normw, xop, ssq, ssq_ld, wf);
  else if (rows_cap == 16) k_pack_embed_rows<16><<<1, 1024, 0, st>>>(rows, n, d, x, normw, xop, ssq, ssq_ld, wf);
  else k_pack_embed_rows<8><<<1, 1024, 0, st>>>(rows, n, d, x, normw, xop, ssq, ssq_ld, wf);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
// position of chunk row i = base->T + i (device-side, like every other length on the decode path)
__global__ void k_chunk_positions(DDState* rows, const DDState* base, int n) {
  int i = threadIdx.x;
  if (i < n) {
    rows[i] = *base;
    rows[i].pos = base->T + i;
    rows[i].T = base->T + i;      // keys before row i: the prefix and the chunk rows ahead of it (its own key comes as a new row)
  }
}
// the chunk's roped K rows / V rows [n][kv_dim] into the cache at positions base->T + i
__global__ __launch_bounds__(256) void k_scatter_kv_rows(const float* __restrict__ kr, const float* __restrict__ vr, int kv_dim,
                                                         float* __restrict__ kc, float* __restrict__ vc, int T_cap,
                                                         const DDState* base, int kv16) {
  const int row = blockIdx.x, T = base->T + row;
  for (int i = threadIdx.x; i < kv_dim; i += 256) {
    int kvh = i / HEAD_DIM, idx = i % HEAD_DIM;
    dd_kv_store(kc, vc, kv16, kvh, idx, T, T_cap, true, kr[(size_t)row * kv_dim + i]);
    dd_kv_store(kc, vc, kv16, kvh, idx, T, T_cap, false, vr[(size_t)row * kv_dim + i]);
  }
}
int ddk_chunk_positions(DDState* rows, const DDState* base, int n, hipStream_t st) {
  k_chunk_positions<<<1, 64, 0, st>>>(rows, base, n);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int ddk_scatter_kv_rows(const float* kr, const float* vr, int n, int kv_dim, float* kc, float* vc, int T_cap, const DDState* base,
                        hipStream_t st, int kv16) {
  k_scatter_kv_rows<<<n, 256, 0, st>>>(kr, vr, kv_dim, kc, vc, T_cap, base, kv16);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
__global__ void k_embed_tokens(const uint16_t* embed, int d, const int32_t* tokens, float* x, int wf) {
  int row = blockIdx.x;
  int tok = tokens[row];
  for (int i = threadIdx.x; i < d; i += 256) x[(size_t)row * d + i] = dd_w16_to_f32(embed[(size_t)tok * d + i], wf);
}
int ddk_embed_tokens(const uint16_t* embed, int d, const int32_t* tokens, int n, float* x, hipStream_t st, int wf) {
  k_embed_tokens<<<n, 256, 0, st>>>(embed, d, tokens, x, wf);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// rows[0][v] := (((r0 + r1) + r2) + ...) / K in fp32 — numpy's mean over axis 0 of a [K, V] float32 array
// (reference llava.py:37-52, select_by_average)
__global__ void k_mean_rows(float* rows, int K, int ld, int n, const int32_t* gate) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || (gate && *gate)) return;
  float acc = rows[i];
  for (int k = 1; k < K; ++k) acc = __fadd_rn(acc, rows[(size_t)k * ld + i]);
  rows[i] = __fdiv_rn(acc, (float)K);
}
int ddk_mean_rows(float* rows, int K, int ld, int n, const int32_t* gate, hipStream_t st) {
  k_mean_rows<<<(n + 255) / 256, 256, 0, st>>>(rows, K, ld, n, gate);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// append the chosen row's new K/V of every layer at position T  (the winner's cache, reference llava.py:373)
__global__ __launch_bounds__(256) void k_commit_kv(const float* __restrict__ knew, const float* __restrict__ vnew,
                                                   int rows_per_layer, int kv_dim, float* __restrict__ kc,
                                                   float* __restrict__ vc, size_t lsk, size_t lsv, int T_cap,
                                                   const DDState* state, int use_winner, int kv16) {
  if (state->done) return;       // finished at EOS: a look-ahead step appends nothing
  int layer = blockIdx.x;
  int row = use_winner ? state->winner : 0;
  int T = state->T;
  const float* kr = knew + ((size_t)layer * rows_per_layer + row) * kv_dim;
  const float* vr = vnew + ((size_t)layer * rows_per_layer + row) * kv_dim;
  float* kl = kc + (size_t)layer * lsk;
  float* vl = vc + (size_t)layer * lsv;
  for (int i = threadIdx.x; i < kv_dim; i += 256) {
    int kvh = i / HEAD_DIM, idx = i % HEAD_DIM;
    dd_kv_store(kl, vl, kv16, kvh, idx, T, T_cap, true, kr[i]);
    dd_kv_store(kl, vl, kv16, kvh, idx, T, T_cap, false, vr[i]);
  }
}
// the same for up to 4 sequences in one launch: grid (layers, sequences)
__global__ __launch_bounds__(256) void k_commit_kv_lanes(CommitLanes t, int rows_per_layer, int kv_dim, int T_cap) {
  const int layer = blockIdx.x, q = blockIdx.y;
  const DDState* state = t.state[q];
  if (state->done) return;
  const int row = state->winner, T = state->T;
  const float* kr = t.knew[q] + ((size_t)layer * rows_per_layer + row) * kv_dim;
  const float* vr = t.vnew[q] + ((size_t)layer * rows_per_layer + row) * kv_dim;
  float* kl = t.kc[q] + (size_t)layer * t.lsk;
  float* vl = t.vc[q] + (size_t)layer * t.lsv;
  for (int i = threadIdx.x; i < kv_dim; i += 256) {
    int kvh = i / HEAD_DIM, idx = i % HEAD_DIM;
    dd_kv_store(kl, vl, t.kv16, kvh, idx, T, T_cap, true, kr[i]);
    dd_kv_store(kl, vl, t.kv16, kvh, idx, T, T_cap, false, vr[i]);
  }
}
int ddk_commit_kv_lanes(const CommitLanes& t, int n, int n_layers, int rows_per_layer, int kv_dim, int T_cap, hipStream_t st) {
  k_commit_kv_lanes<<<dim3(n_layers, n), 256, 0, st>>>(t, rows_per_layer, kv_dim, T_cap);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int ddk_commit_kv(const float* knew, const float* vnew, int n_layers, int rows_per_layer, int kv_dim, float* kc,
                  float* vc, size_t lsk, size_t lsv, int T_cap, const DDState* state, int use_winner, hipStream_t st, int kv16) {
  k_commit_kv<<<n_layers, 256, 0, st>>>(knew, vnew, rows_per_layer, kv_dim, kc, vc, lsk, lsv, T_cap, state, use_winner, kv16);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

__global__ __launch_bounds__(256) void k_kv_sums(const float* kc, const float* vc, size_t lsk, size_t lsv, int n_kv,
                                                 int T_cap, int T, double* out, int kv16) {
  __shared__ double sh[4];
  int layer = blockIdx.x, which = blockIdx.y;
  double acc = 0;
  size_t n = (size_t)n_kv * HEAD_DIM * T;
  for (size_t i = threadIdx.x; i < n; i += 256) {
    int t = (int)(i % T);
    size_t r = i / T;  // kvh*128 + idx
    int kvh = (int)(r / HEAD_DIM), idx = (int)(r % HEAD_DIM);
    float v = dd_kv_load(kc + (size_t)layer * lsk, vc + (size_t)layer * lsv, kv16, kvh, idx, t, T_cap, which == 0);
    acc += (double)v;
  }
  acc = dd_wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[layer * 2 + which] = sh[0] + sh[1] + sh[2] + sh[3];
}
int ddk_kv_sums(const float* kc, const float* vc, int n_layers, size_t lsk, size_t lsv, int n_kv, int T_cap, int T,
                double* out, hipStream_t st, int kv16) {
  k_kv_sums<<<dim3(n_layers, 2), 256, 0, st>>>(kc, vc, lsk, lsv, n_kv, T_cap, T, out, kv16);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

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
#include "dd_lm_device.h"

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

// ===============================================================================================
// glue
// ===============================================================================================
__global__ __launch_bounds__(1024) void k_embed_rows(const uint16_t* __restrict__ embed, int d, const DDState* state,
                                                     float* __restrict__ x, const float* __restrict__ normw,
                                                     u32x4_t* __restrict__ xop, float* __restrict__ ssq, int ssq_ld,
                                                     const int32_t* __restrict__ skip_if, int wf) {
  // one workgroup per row (round 6: a single workgroup wrote all eight rows — 128 KB of fp32 and 64 K two-byte operand stores from one CU, 12 us);
  // every row's sum of squares is reduced as before (thread t over columns t, t + 1024, ..., wave sums, the 16 waves in order): the same bits
  __shared__ float sh[16];
  if (skip_if && *skip_if) return;
  const int m = blockIdx.x;
  int tok = state->cur_tok;
  float ss = 0.f;
  for (int i = threadIdx.x; i < d; i += 1024) {
    float e = dd_w16_to_f32(embed[(size_t)tok * d + i], wf);
    ss += e * e;
    float z = normw[i] * e;
    x[(size_t)m * d + i] = e;
    xop_store(xop, i, m, z, wf);
  }
  ss = dd_wave_sum(ss);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int i = 0; i < 16; ++i) v += sh[i];
    ssq[(size_t)m * ssq_ld] = v;
  }
}
int ddk_embed_rows(const uint16_t* embed, int d, const DDState* state, float* x, const float* normw, u32x4_t* xop,
                   float* ssq, int ssq_ld, hipStream_t st, const int32_t* skip_if, int wf) {
  k_embed_rows<<<8, 1024, 0, st>>>(embed, d, state, x, normw, xop, ssq, ssq_ld, skip_if, wf);
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
  DD_REQUIRE(rows == 8 || rows == 16 || rows == 32 || rows == 64 || rows == 72, "embed_rows_lanes: %d rows (8, 16, 32, 64 or 72)", rows);
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
#define COMMIT_WGS 4       // workgroups per layer (round 6: one workgroup per layer walked 4,096 scattered two-byte stores sixteen deep: 14 us)
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
  for (int i = blockIdx.y * 256 + threadIdx.x; i < kv_dim; i += gridDim.y * 256) {     // grid (layers, COMMIT_WGS): the elements are independent
    int kvh = i / HEAD_DIM, idx = i % HEAD_DIM;
    dd_kv_store(kl, vl, kv16, kvh, idx, T, T_cap, true, kr[i]);
    dd_kv_store(kl, vl, kv16, kvh, idx, T, T_cap, false, vr[i]);
  }
}
// the same for up to 4 sequences in one launch: grid (layers, sequences, COMMIT_WGS)
__global__ __launch_bounds__(256) void k_commit_kv_lanes(CommitLanes t, int rows_per_layer, int kv_dim, int T_cap) {
  const int layer = blockIdx.x, q = blockIdx.y;
  const DDState* state = t.state[q];
  if (state->done) return;
  const int row = state->winner, T = state->T;
  const float* kr = t.knew[q] + ((size_t)layer * rows_per_layer + row) * kv_dim;
  const float* vr = t.vnew[q] + ((size_t)layer * rows_per_layer + row) * kv_dim;
  float* kl = t.kc[q] + (size_t)layer * t.lsk;
  float* vl = t.vc[q] + (size_t)layer * t.lsv;
  for (int i = blockIdx.z * 256 + threadIdx.x; i < kv_dim; i += gridDim.z * 256) {
    int kvh = i / HEAD_DIM, idx = i % HEAD_DIM;
    dd_kv_store(kl, vl, t.kv16, kvh, idx, T, T_cap, true, kr[i]);
    dd_kv_store(kl, vl, t.kv16, kvh, idx, T, T_cap, false, vr[i]);
  }
}
int ddk_commit_kv_lanes(const CommitLanes& t, int n, int n_layers, int rows_per_layer, int kv_dim, int T_cap, hipStream_t st) {
  k_commit_kv_lanes<<<dim3(n_layers, n, COMMIT_WGS), 256, 0, st>>>(t, rows_per_layer, kv_dim, T_cap);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int ddk_commit_kv(const float* knew, const float* vnew, int n_layers, int rows_per_layer, int kv_dim, float* kc,
                  float* vc, size_t lsk, size_t lsv, int T_cap, const DDState* state, int use_winner, hipStream_t st, int kv16) {
  k_commit_kv<<<dim3(n_layers, COMMIT_WGS), 256, 0, st>>>(knew, vnew, rows_per_layer, kv_dim, kc, vc, lsk, lsv, T_cap, state, use_winner, kv16);
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


// Tensor-parallel seam of the prefill (dd_tp.hip): x[i] += sum over ranks (in rank order) of slot r of `gather` [W][n]
__global__ __launch_bounds__(256) void k_tp_add_rows(float* __restrict__ x, const float* __restrict__ gather, int W, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4_t y = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < W; ++r) y = y + *(const f32x4_t*)(gather + (size_t)r * n + i);
  *(f32x4_t*)(x + i) = *(const f32x4_t*)(x + i) + y;
}
int ddk_tp_add_rows(float* x, const float* gather, int W, size_t n, hipStream_t st) {
  DD_REQUIRE(x && gather && W >= 1 && n % 4 == 0, "tp_add_rows: bad arguments");
  k_tp_add_rows<<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>(x, gather, W, n);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

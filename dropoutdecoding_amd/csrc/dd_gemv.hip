// Decode GEMVs of the K-way masked-context step (gfx950, wave64): v_mfma_f32_16x16x32 with the WEIGHT tile as the A operand
// (16 output rows) and the packed activation hi/lo pairs of up to 8 ensemble rows as the 16 B-operand columns; weights are
// streamed straight to VGPRs (1 KiB per wave instruction, non-temporal).  8 rows: K split over the 8 waves of a workgroup and
// reduced through LDS in a fixed order; 16 / 32 / 64 rows: slice-resident kernels (dd_gemv_slices.h) + a finishing kernel.
// Reference anchors: the third-party LM forward the reference calls at models/llava.py:294-303,350-359.
#include <type_traits>

#include "dd_lm_kernels.h"
#include "dd_lm_device.h"
#include "dd_gemv_slices.h"

// ===============================================================================================
// decode GEMV
// ===============================================================================================
#define GEMV_WAVES 8
#define GEMV_THREADS (GEMV_WAVES * 64)
// U   = weight tiles requested per wave before the first MFMA consumes one (loads in flight)
// NT  = non-temporal weight loads (read-once stream, keeps L2/MALL for the x operand and the KV cache)
// ILV = k-steps interleaved over the 8 waves (wave w takes steps w, w+8, ...: at any instant the workgroup reads
//       8 consecutive KiB) instead of one contiguous chunk per wave
// Everything behind the LDS reduction of an 8-row pass, for the TILES tiles tile0.. of tile group tg (k_gemv: tg = blockIdx.x; k_gemv_loop walks
// several groups per workgroup).  red: [TILES][8 waves][256] accumulators as the waves stored them; rstd_sh must be visible (barrier before the call).
template <int EPI, int TILES, int FP8, int WF>
__device__ __forceinline__ void gemv8_epilogue(const GemvArgs& a, const int tile0, const int tg, const float* red, const float* rstd_sh, float* ssq_sh,
                                               const float pre0, const float pre1) {
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
      a.ssq_out[(size_t)t * a.ssq_ld + tg] = v;
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
        xop_store(a.xop_next, tg * 16 + n, m, act * u, WF);
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
          float o = dd_rope_mix(y, yp, c, sn, n < 8);     // q*cos + rotate_half(q)*sin (HF apply_rotary_pos_emb)
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

  // What the epilogue needs from memory is REQUESTED before the weight stream and USED after it — by every thread, at clamped addresses,
  // with no divergent branch around a load.  (Round 6: the form before this one loaded inside `if (threadIdx.x < 128 && em < a.nb)` and reduced
  // the rows' rstd before the stream; the compiler answers a load inside a divergent block with `s_waitcnt vmcnt(0)` at the block's end,
  // and the dependent ones — the row's state pointer out of the argument table, its position, the rotary entry at that position — queued
  // up: two to four memory round trips in EVERY workgroup before its first weight request, with the whole chip waiting at a kernel's start.)
  // Up front: the sum-of-squares slots of row `wave` (one 16-byte load per lane covers 256 slots: every workgroup of the launch reads these
  // same few lines, so the request count matters), the residual input + next norm weight (EPI_RESID), the row's state pointer (EPI_QKV /
  // EPI_STORE).  After the stream: the rstd, and the dependent loads — position -> rotary cos / sin, the finished flag — which then hit in L2.
  const bool has_ssq = a.ssq_in != nullptr;
  f32x4_t sv = {0.f, 0.f, 0.f, 0.f};
  if (has_ssq) {                                       // (wave-uniform branch)
    const int i4 = min(4 * lane, max((a.ssq_n - 1) & ~3, 0));    // lanes past the last slot re-read it: masked where the slots are added up
    sv = *(const f32x4_t*)(a.ssq_in + (size_t)wave * a.ssq_ld + i4);
  }
  const int em = threadIdx.x & 7, en = (threadIdx.x >> 3) & 15, emc = min(em, a.nb - 1);
  float pre0 = 0.f, pre1 = 0.f;
  const DDState* sp_row = nullptr;
  if (EPI == EPI_RESID) {
    pre0 = a.out[(size_t)emc * a.ldo + tile0 * 16 + en];
    pre1 = a.normw_next[tile0 * 16 + en];
  } else if (EPI == EPI_QKV || EPI == EPI_STORE) {
    sp_row = a.state_rows[emc] ? a.state_rows[emc] : a.state;
  }
  auto late_prefetch = [&]() {                         // after the weight stream: the loads that depend on loaded values
    if (EPI == EPI_QKV) {
      if (tile0 < a.q_tiles + a.k_tiles) {             // (workgroup-uniform)
        const int ht = tile0 < a.q_tiles ? tile0 : tile0 - a.q_tiles;
        const int f = (ht & 7) * 8 + (en & 7);
        const int pos = sp_row->pos;
        pre0 = a.rope_cos[(size_t)pos * ROPE_HALF + f];
        pre1 = a.rope_sin[(size_t)pos * ROPE_HALF + f];
      }
    } else if (EPI == EPI_STORE) {
      // logits of a sequence that already emitted its EOS are not overwritten by look-ahead steps (DDState::done)
      if (sp_row && sp_row->done) pre0 = 1.f;
    }
  };

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

  late_prefetch();
  finish_rstd();
#pragma unroll
  for (int t = 0; t < TILES; ++t) *(f32x4_t*)&red[(t * GEMV_WAVES + wave) * 256 + lane * 4] = acc[t];
  __syncthreads();

  gemv8_epilogue<EPI, TILES, FP8, WF>(a, tile0, blockIdx.x, red, rstd_sh, ssq_sh, pre0, pre1);
}

// The 8-row GEMV for K = 4096 with the rows' operand slice in REGISTERS and several tile groups per workgroup (round 6).  In k_gemv a workgroup
// lives for one tile group: it loads its operand fragments from L2 beside the weights (1 KiB of x per KiB of weights, for every workgroup again),
// reduces through the LDS, runs its epilogue and ends — the stream of the NEXT workgroup on that CU starts behind all of that.  Timing experiments
// (profiles/r06_lab/gemv8_ablation.log) put the operand loads at 1-2.4 us and the reduction + epilogue at 1-2 us of a 20-34 us kernel.  Here a
// workgroup keeps wave w's sixteen operand fragments (k-steps w, w + 8, ...) in 64 registers for its whole life, walks tile groups
// blockIdx.x, blockIdx.x + gridDim.x, ..., and requests the next group's first eight weight tiles BEFORE it reduces and finishes the current group, so
// the weight stream runs through reduction and epilogue.  Per tile the MFMA chain (k-steps in order) and the reduction (gemv8_epilogue) are k_gemv's:
// the same bits (tests/test_gpu_engine.py goldens, tests/test_gpu_7b_shapes_vs_oracle.py solo runs, lanes vs solo everywhere).
template <int EPI, int TILES, int WF>
__global__ __launch_bounds__(GEMV_THREADS) void k_gemv_loop(GemvArgs a, int n_groups) {
  constexpr int NS = 16, U = 8;                        // k-steps per wave (K = 4096); weight tiles per batch
  __shared__ float red[2][TILES * GEMV_WAVES * 256];
  __shared__ float rstd_sh[8];
  __shared__ float ssq_sh[8 * 16];
  if (a.skip_if && *a.skip_if) return;
  int g = blockIdx.x;
  if (g >= n_groups) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S;
  auto wptr = [&](int grp, int t) -> const u32x4_t* { return a.W + ((size_t)(grp * TILES + t) * S + wave) * 64 + lane; };
  u32x4_t wa[TILES][U], wb[TILES][U];
  auto issue = [&](u32x4_t (&w)[TILES][U], int grp, int half) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TILES; ++t) w[t][u] = __builtin_nontemporal_load(wptr(grp, t) + (size_t)(half * U + u) * GEMV_WAVES * 64);
  };
  issue(wa, g, 0);                                     // the weight stream starts before anything else is asked for
  u32x4_t xr[NS];
  {
    const u32x4_t* xp = a.xop + (size_t)wave * 64 + lane;
#pragma unroll
    for (int i = 0; i < NS; ++i) xr[i] = xp[(size_t)i * GEMV_WAVES * 64];
  }
  const bool has_ssq = a.ssq_in != nullptr;
  f32x4_t sv = {0.f, 0.f, 0.f, 0.f};
  if (has_ssq) {
    const int i4 = min(4 * lane, max((a.ssq_n - 1) & ~3, 0));
    sv = *(const f32x4_t*)(a.ssq_in + (size_t)wave * a.ssq_ld + i4);
  }
  const int em = threadIdx.x & 7, en = (threadIdx.x >> 3) & 15, emc = min(em, a.nb - 1);
  const DDState* sp_row = nullptr;
  if (EPI == EPI_QKV || EPI == EPI_STORE) sp_row = a.state_rows[emc] ? a.state_rows[emc] : a.state;
  int pos = 0;
  float done_f = 0.f;
  bool first = true;
  int par = 0;
  for (; g < n_groups; g += gridDim.x) {
    const int gn = g + gridDim.x;
    const int tile0 = g * TILES;
    issue(wb, g, 1);
    float pre0 = 0.f, pre1 = 0.f;
    if (EPI == EPI_RESID) {
      pre0 = a.out[(size_t)emc * a.ldo + tile0 * 16 + en];
      pre1 = a.normw_next[tile0 * 16 + en];
    }
    if (first) {                                       // (once per workgroup: the row's position / finished flag)
      if (EPI == EPI_QKV) pos = sp_row->pos;
      if (EPI == EPI_STORE) done_f = (sp_row && sp_row->done) ? 1.f : 0.f;
    }
    f32x4_t acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TILES; ++t) acc[t] = dd_mfma16<WF>(wa[t][u], xr[u], acc[t]);
    if (gn < n_groups) issue(wa, gn, 0);               // the next group's first batch travels through this group's reduction and epilogue
    if (EPI == EPI_QKV) {
      if (tile0 < a.q_tiles + a.k_tiles) {             // (workgroup-uniform)
        const int ht = tile0 < a.q_tiles ? tile0 : tile0 - a.q_tiles;
        const int f = (ht & 7) * 8 + (en & 7);
        pre0 = a.rope_cos[(size_t)pos * ROPE_HALF + f];
        pre1 = a.rope_sin[(size_t)pos * ROPE_HALF + f];
      }
    } else if (EPI == EPI_STORE) {
      pre0 = done_f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TILES; ++t) acc[t] = dd_mfma16<WF>(wb[t][u], xr[U + u], acc[t]);
    if (first) {
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
      first = false;
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) *(f32x4_t*)&red[par][(t * GEMV_WAVES + wave) * 256 + lane * 4] = acc[t];
    __syncthreads();
    gemv8_epilogue<EPI, TILES, 0, WF>(a, tile0, g, red[par], rstd_sh, ssq_sh, pre0, pre1);
    par ^= 1;
  }
}
// Tensor-parallel seam of a row-parallel matrix (o_proj, down_proj; dd_tp.hip): every rank's k_gemv wrote its partial product
// y_r [rows][N] (EPI_STORE) into slot r of `gather` [W][rows][N]; this adds the slots in rank order and runs k_gemv's EPI_RESID
// epilogue on the sum: x += y, the next matrix's packed operand z = normw * x, the sum-of-squares slots of the folded RMSNorm.
// One workgroup per 16-column tile like k_gemv, so the slots land where the consuming k_gemv expects them.  W = 1 reproduces
// k_gemv<EPI_RESID> bit for bit (0 + y = y).
__global__ __launch_bounds__(128) void k_tp_finish(const float* __restrict__ gather, int W, size_t slot, int nb, float* x, int ldo,
                                                   const float* __restrict__ normw, u32x4_t* xop_next, float* ssq_out, int ssq_ld,
                                                   int wf) {
  __shared__ float ssq_sh[8 * 16];
  const int t = threadIdx.x, m = t & 7, n = t >> 3, col = blockIdx.x * 16 + n;
  float sq = 0.f;
  if (m < nb) {
    float y = 0.f;
    for (int r = 0; r < W; ++r) y += gather[(size_t)r * slot + (size_t)m * ldo + col];
    const float xn = x[(size_t)m * ldo + col] + y;
    x[(size_t)m * ldo + col] = xn;
    xop_store(xop_next, col, m, normw[col] * xn, wf);
    sq = xn * xn;
  }
  ssq_sh[n * 8 + m] = sq;
  __syncthreads();
  if (t < 8) {
    float v = 0.f;
    for (int i = 0; i < 16; ++i) v += ssq_sh[i * 8 + t];
    ssq_out[(size_t)t * ssq_ld + blockIdx.x] = v;
  }
}
int ddk_tp_finish(const float* gather, int W, size_t slot_floats, int nb, float* x, int N, const float* normw, u32x4_t* xop_next,
                  float* ssq_out, int ssq_ld, int wf, hipStream_t st) {
  DD_REQUIRE(gather && W >= 1 && nb >= 1 && nb <= 8 && N % 16 == 0, "tp_finish: bad arguments");
  k_tp_finish<<<N / 16, 128, 0, st>>>(gather, W, slot_floats, nb, x, N, normw, xop_next, ssq_out, ssq_ld, wf);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// tuning knobs (dd_set_tuning): 0 = U (4/8/16), 4 = ring (1) or batch (0) request order.
// Keys 1 (non-temporal loads) and 2 (k-step interleave) are settled at 1 and kept only as accepted no-ops.
static int g_gemv_u = 8, g_gemv_pipe = 0;
void ddk_set_tuning(int key, int value) {
  if (key == 0) g_gemv_u = value;
  else if (key == 4) g_gemv_pipe = value;
}

// name (as a kernel trace prints it) of the streaming kernel the last ddk_gemv / ddk_gemv_groups call of this thread launched
static thread_local char g_last_kernel[96] = "";
const char* ddk_last_gemv_kernel() { return g_last_kernel; }
#define NOTE_KERNEL(...) snprintf(g_last_kernel, sizeof(g_last_kernel), __VA_ARGS__)

template <int EPI, int TILES>
static void launch_gemv(const GemvArgs& a_, hipStream_t st) {
  const GemvArgs& a = a_;
  NOTE_KERNEL("k_gemv<%d, %d, %d, 1, 1, %d, %d, %d>", EPI, TILES, a.fp8 ? 8 : (g_gemv_u == 4 || g_gemv_u == 16 ? g_gemv_u : 8), a.fp8 ? 1 : 0,
              a.fp8 ? 0 : (g_gemv_pipe ? 1 : 0), a.fp8 ? 0 : (a.wf ? 1 : 0));
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

int g_gemv_loop = 1;            // dd_tools_set_tuning key 54: k_gemv_loop for the K = 4096 matrices of an 8-row pass (0: k_gemv; n > 1: n workgroups)
template <int EPI, int TILES>
static void launch_gemv_loop(const GemvArgs& a, int n_groups, hipStream_t st) {
  // one workgroup per CU (165-230 VGPRs: two waves per SIMD), each walking ceil(n_groups / 256) tile groups
  const int grid = g_gemv_loop > 1 ? (g_gemv_loop < n_groups ? g_gemv_loop : n_groups) : (n_groups < 256 ? n_groups : 256);
  NOTE_KERNEL("k_gemv_loop<%d, %d, %d>", EPI, TILES, a.wf ? 1 : 0);
  if (a.wf) k_gemv_loop<EPI, TILES, 1><<<grid, GEMV_THREADS, 0, st>>>(a, n_groups);
  else k_gemv_loop<EPI, TILES, 0><<<grid, GEMV_THREADS, 0, st>>>(a, n_groups);
}

int ddk_gemv(int epi, const GemvArgs& a, hipStream_t st) {
  DD_REQUIRE(a.S % GEMV_WAVES == 0 && a.S >= GEMV_WAVES, "gemv: K=%d must be a multiple of 256", a.S * 32);
  DD_REQUIRE(a.nb >= 1 && a.nb <= 8, "gemv: nb=%d", a.nb);
  DD_REQUIRE(!a.fp8 || a.wscale, "gemv: fp8 weights need row scales");
  DD_REQUIRE(!a.ssq_in || a.ssq_n >= 1, "gemv: ssq_n");
  if (g_gemv_loop && !a.fp8 && a.S == 16 * GEMV_WAVES && a.n_tiles > 256 && (g_gemv_loop > 1 || a.n_tiles % 256 == 0)) {
    // K = 4096 and a whole number of tile groups per CU: LLaMA-7B's qkv (768 tiles: 20.3 -> 19.2 us).  Measured and left on k_gemv: gate/up (688
    // tile pairs = 2.69 per workgroup, the third round two thirds empty: 34.0 -> 40.3 us); o_proj's 256 tiles are one group per workgroup either way
    // (dd_tools_set_tuning key 54 > 1 forces the form with that many workgroups for every K = 4096 matrix)
    switch (epi) {
      case EPI_STORE: launch_gemv_loop<EPI_STORE, 1>(a, a.n_tiles, st); break;
      case EPI_RESID: launch_gemv_loop<EPI_RESID, 1>(a, a.n_tiles, st); break;
      case EPI_SILU: launch_gemv_loop<EPI_SILU, 2>(a, a.n_tiles, st); break;
      default: launch_gemv_loop<EPI_QKV, 1>(a, a.n_tiles, st); break;
    }
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
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
  const bool erow = et < 128 * NG && a.row_live(eg, ml);
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
  const bool erow = et < 128 * NG && a.row_live(eg, ml);
  if (EPI == EPI_STORE) {
    if (erow) {
      float y = tile_sum(0, en);
      if (a.ssq_in) y *= rstd_sh[em];
      int col = tile0 * 16 + en;
      float* row = a.out_g[a.slot(eg, ml)] ? a.out_g[a.slot(eg, ml)] + (size_t)a.slot_row(eg, ml) * a.ldo : a.out + (size_t)em * a.ldo;
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
      float* kn = a.knew_g[a.slot(eg, ml)] ? a.knew_g[a.slot(eg, ml)] + (size_t)a.slot_row(eg, ml) * a.kv_dim : a.knew + (size_t)em * a.kv_dim;
      float* vn = a.vnew_g[a.slot(eg, ml)] ? a.vnew_g[a.slot(eg, ml)] + (size_t)a.slot_row(eg, ml) * a.kv_dim : a.vnew + (size_t)em * a.kv_dim;
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
          float o = dd_rope_mix(y, yp, c, sn, en < 8);
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
  // (round 6: the epilogue's operands and the rows' rstd are fetched AFTER the weight stream — groups_prefetch loads inside a divergent
  // branch and follows pointers, the compiler waits for those loads where they stand: in front of the first weight request, as in k_gemv)

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

  groups_prefetch<EPI, TILES, NG>(a, tile0, pre);
  groups_rstd<NG>(a, rstd_sh);
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

// The same finishing step with FOUR output columns per thread: one 16-byte load per partial-sum slab instead of four 4-byte
// ones (the slabs' element order — dd_part_index — keeps columns n & 3 adjacent), 16-byte residual / logits accesses, 8-byte
// operand stores, the rotary partner column (n ^ 8) by one wave shuffle.  32 * NG threads per tile set: thread = (plane eg,
// column quad c4 = columns 4 c4 .. 4 c4 + 3, row ml).  Every output is computed by the same operations in the same order as in
// k_gemv_finish (and so as in k_gemv / k_gemv_groups): the bits do not change.
__device__ __forceinline__ void xop_store16x4(u32x4_t* xop, int k0, int m, const float (&y)[4], int S, int wf) {
  uint32_t hi[4], lo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dd_split(y[i], hi[i], lo[i], wf);
  uint16_t* p = (uint16_t*)(xop + (size_t)(m >> 3) * S * 64);
  const int ks = k0 >> 5, h = (k0 >> 3) & 3, j = k0 & 7, ml = m & 7;      // k0 is a multiple of 4: the four k share (ks, h)
  const size_t base = ((size_t)ks * 64 + h * 16) * 8 + j;
  *(u32x2_t*)&p[base + (size_t)ml * 8] = (u32x2_t){hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
  *(u32x2_t*)&p[base + (size_t)(ml + 8) * 8] = (u32x2_t){lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16)};
}

template <int EPI, int TILES, int NG, int NP>
__global__ __launch_bounds__(32 * NG) void k_gemv_finish4(GemvArgs a, const float* __restrict__ part, const float* __restrict__ rstd_g,
                                                          int n_sets) {
  __shared__ float ssq_sh[EPI == EPI_RESID ? 16 * 8 * NG : 1];
  __shared__ f32x4_t yq_sh[EPI == EPI_QKV ? TILES * 32 * NG : 1];   // rotary tiles: a thread needs its partner quad's sums (columns n ^ 8)
  const int wg = blockIdx.x, tile0 = wg * TILES;
  const int t = threadIdx.x, eg = t >> 5, c4 = (t & 31) >> 3, ml = t & 7, em = (eg << 3) + ml, n0 = c4 * 4;
  const bool erow = a.row_live(eg, ml);
  const size_t ps = ((size_t)n_sets * TILES * NG) << 7;
  // one batch of requests: the thread's partial sums, rstd, the epilogue's operands
  f32x4_t v[TILES][NP];
#pragma unroll
  for (int tt = 0; tt < TILES; ++tt) {
    const float* p0 = part + (((size_t)(tile0 + tt) * NG + eg) << 7) + ((c4 * 8 + ml) << 2);
#pragma unroll
    for (int q = 0; q < NP; ++q) v[tt][q] = *(const f32x4_t*)(p0 + (size_t)q * ps);
  }
  const float rstd = a.ssq_in ? rstd_g[em] : 1.0f;
  f32x4_t pre0 = {0.f, 0.f, 0.f, 0.f}, pre1 = {0.f, 0.f, 0.f, 0.f}, rc[TILES], rs[TILES];
  bool done = false;
  if (erow) {
    if (EPI == EPI_RESID) {
      pre0 = *(const f32x4_t*)&a.out[(size_t)em * a.ldo + tile0 * 16 + n0];
      pre1 = *(const f32x4_t*)&a.normw_next[tile0 * 16 + n0];
    } else if (EPI == EPI_QKV) {
      const DDState* sp = a.state_rows[em] ? a.state_rows[em] : a.state;
      const int pos = sp->pos;
#pragma unroll
      for (int tt = 0; tt < TILES; ++tt) {
        const int nt = tile0 + tt;
        rc[tt] = rs[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (nt < a.q_tiles + a.k_tiles) {
          const int ht = nt < a.q_tiles ? nt : nt - a.q_tiles;
          const int f = (ht & 7) * 8 + (n0 & 7);
          rc[tt] = *(const f32x4_t*)&a.rope_cos[(size_t)pos * ROPE_HALF + f];
          rs[tt] = *(const f32x4_t*)&a.rope_sin[(size_t)pos * ROPE_HALF + f];
        }
      }
    } else if (EPI == EPI_STORE) {
      const DDState* sp = a.state_rows[em] ? a.state_rows[em] : a.state;
      done = sp && sp->done;               // finished sequence: its logits stay as the EOS step left them
    }
  }
  float y[TILES][4];
#pragma unroll
  for (int tt = 0; tt < TILES; ++tt) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (NP == 8) {
#pragma unroll
      for (int q = 0; q < 8; q += 2) acc = acc + (v[tt][q] + v[tt][q + 1]);
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) acc = acc + v[tt][q];
    }
    y[tt][0] = acc.x, y[tt][1] = acc.y, y[tt][2] = acc.z, y[tt][3] = acc.w;
    if (a.fp8 && EPI != EPI_RESID) {         // fp8 tiles: the output rows' scales, after the sum as in k_gemv
      const f32x4_t sc = *(const f32x4_t*)&a.wscale[(size_t)(tile0 + tt) * 16 + n0];
      y[tt][0] *= sc.x, y[tt][1] *= sc.y, y[tt][2] *= sc.z, y[tt][3] *= sc.w;
    }
  }
  if (EPI == EPI_STORE) {
    if (erow && !done) {
      float* row = a.out_g[a.slot(eg, ml)] ? a.out_g[a.slot(eg, ml)] + (size_t)a.slot_row(eg, ml) * a.ldo : a.out + (size_t)em * a.ldo;
      const int col = tile0 * 16 + n0;
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = a.ssq_in ? y[0][i] * rstd : y[0][i];
      if (col + 3 < a.n_valid && (a.ldo & 3) == 0) *(f32x4_t*)&row[col] = (f32x4_t){o[0], o[1], o[2], o[3]};
      else
        for (int i = 0; i < 4; ++i)
          if (col + i < a.n_valid) row[col + i] = o[i];
    }
  } else if (EPI == EPI_RESID) {
    float sq[4] = {0.f, 0.f, 0.f, 0.f};
    if (erow) {
      const int col = tile0 * 16 + n0;
      float xn[4], z[4];
      const float p0[4] = {pre0.x, pre0.y, pre0.z, pre0.w}, p1[4] = {pre1.x, pre1.y, pre1.z, pre1.w};
      // fp8 tiles: x + scale * sum as ONE fused multiply-add — the form the 8-row kernel's epilogue compiles to (its scale multiply
      // and residual add contract), which the rows of every pass width must reproduce
      f32x4_t sc4 = {1.f, 1.f, 1.f, 1.f};
      if (a.fp8) sc4 = *(const f32x4_t*)&a.wscale[(size_t)tile0 * 16 + n0];
      const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xn[i] = a.fp8 ? __builtin_fmaf(y[0][i], sc[i], p0[i]) : p0[i] + y[0][i];
        z[i] = p1[i] * xn[i];
        sq[i] = xn[i] * xn[i];
      }
      *(f32x4_t*)&a.out[(size_t)em * a.ldo + col] = (f32x4_t){xn[0], xn[1], xn[2], xn[3]};
      xop_store16x4(a.xop_next, col, em, z, a.S_next, a.wf);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ssq_sh[(n0 + i) * (8 * NG) + em] = sq[i];
    __syncthreads();
    if (t < 8 * NG) {
      float s = 0.f;
      for (int i = 0; i < 16; ++i) s += ssq_sh[i * (8 * NG) + t];
      a.ssq_out[(size_t)t * a.ssq_ld + wg] = s;
    }
  } else if (EPI == EPI_SILU) {
    if (erow) {
      float z[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float g = y[0][i], u = y[TILES - 1][i];
        if (a.ssq_in) {
          g *= rstd;
          u *= rstd;
        }
        const float act = g / (1.0f + expf(-g));  // silu
        z[i] = act * u;
      }
      xop_store16x4(a.xop_next, wg * 16 + n0, em, z, a.S_next, a.wf);
    }
  } else {  // EPI_QKV
    float* kn = a.knew_g[a.slot(eg, ml)] ? a.knew_g[a.slot(eg, ml)] + (size_t)a.slot_row(eg, ml) * a.kv_dim : a.knew + (size_t)em * a.kv_dim;
    float* vn = a.vnew_g[a.slot(eg, ml)] ? a.vnew_g[a.slot(eg, ml)] + (size_t)a.slot_row(eg, ml) * a.kv_dim : a.vnew + (size_t)em * a.kv_dim;
#pragma unroll
    for (int tt = 0; tt < TILES; ++tt) yq_sh[tt * 32 * NG + t] = (f32x4_t){y[tt][0], y[tt][1], y[tt][2], y[tt][3]};
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < TILES; ++tt) {
      const int nt = tile0 + tt;
      float yy[4], yp[4];
      const f32x4_t pq = yq_sh[tt * 32 * NG + (t ^ 16)];   // columns n ^ 8 of the same (plane, row): the thread whose quad is c4 ^ 2
      const float pr[4] = {pq.x, pq.y, pq.z, pq.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        yy[i] = y[tt][i];
        yp[i] = pr[i];
        if (a.ssq_in) {
          yy[i] *= rstd;
          yp[i] *= rstd;
        }
      }
      if (!erow) continue;
      if (nt < a.q_tiles + a.k_tiles) {
        const bool is_q = nt < a.q_tiles;
        const int ht = is_q ? nt : nt - a.q_tiles;
        const int head = ht >> 3, f = (ht & 7) * 8 + (n0 & 7);
        const float c[4] = {rc[tt].x, rc[tt].y, rc[tt].z, rc[tt].w}, sn[4] = {rs[tt].x, rs[tt].y, rs[tt].z, rs[tt].w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          o[i] = dd_rope_mix(yy[i], yp[i], c[i], sn[i], n0 < 8);
        const int i0 = (n0 < 8) ? f : ROPE_HALF + f;
        float* dst = is_q ? a.qbuf + (size_t)em * a.q_dim + head * HEAD_DIM + i0 : kn + head * HEAD_DIM + i0;
        *(f32x4_t*)dst = (f32x4_t){o[0], o[1], o[2], o[3]};
      } else {
        const int col = (nt - a.q_tiles - a.k_tiles) * 16 + n0;
        *(f32x4_t*)&vn[col] = (f32x4_t){yy[0], yy[1], yy[2], yy[3]};
      }
    }
  }
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
  NOTE_KERNEL("k_gemv_groups<%d, %d, %d, %d, %d, %d>", EPI, TILES, NG, U, FP8, (!FP8 && a.wf) ? 1 : 0);
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
int g_exp_temporal = 0;        // dd_tools_set_tuning key 36: slice kernels load weight tiles with the default cache policy (A/B)
int g_exp_U9 = 4;                // dd_tools_set_tuning key 29: weight requests in flight per wave of the nine-plane qkv kernel (4 or 8; gate/up: always 4)
int g_exp_G[4] = {0, 0, 0, 0};   // dd_set_tuning keys 17..19: workgroups per slice of the 64-row kernels (qkv, o, gate/up); 0 = default
static int g_slices_only = 0;     // dd_lm_time_gemv: launch the streaming kernel without its finishing kernel (timing only)
void ddk_set_gemv_slices(int on) { g_gemv_slices = on; }
void ddk_set_slices_only(int on) { g_slices_only = on; }
#define SLICES_UNSUPPORTED 1

extern int g_seq_prog;
// dd_tools_set_tuning key 50: the rows' rstd in a workgroup of its own behind the streaming ones (SliceArgs::rstd_wg; default 1; 0 = workgroup 0
// does it before its own weight stream, as until round 5's last day).  Same bits either way.
int g_rstd_wg = 1;
static inline int rstd_blocks(const SliceArgs& sa) { return sa.rstd_wg && sa.ssq_in ? (sa.halves == 2 ? 2 : 1) : 0; }
template <int TW, int NG, int U, int SPW, int CS, int CH, int TAG>
static int launch_slices_k(const SliceArgs& sa, int wf, hipStream_t st) {
  constexpr size_t smem = (size_t)CH * (SPW < CS ? SPW : CS) * NG * 1024;
  // progressive stage-in (dd_gemv_slices.h PROG): whole-slice kernels with at least two ring blocks per tile group
  // Measured (profiles/r05_progressive_stage_in.log): it pays at two and four planes (16 / 32 rows: gate/up 33.7 -> 32.7 / 36.4 -> 35.0 us) and
  // costs at eight and nine (the per-block barriers of the first tile group: 64-lane step 36.4 -> 37.7 ms) — so only NG <= 4 takes it.
  constexpr int PROG_OK = (NG <= 4 && SPW <= CS && (CH * SPW) / (SPW < U ? SPW : U) >= 2) ? 1 : 0;
  static bool attr = false;
  if (!attr) {
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices<TW, NG, U, SPW, CS, CH, 0, TAG, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices<TW, NG, U, SPW, CS, CH, 1, TAG, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices<TW, NG, U, SPW, CS, CH, 0, TAG, PROG_OK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices<TW, NG, U, SPW, CS, CH, 1, TAG, PROG_OK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  const int grid = (sa.halves == 2 ? 2 : 1) * (8 / CH) * sa.G + rstd_blocks(sa);
  const int prog = g_seq_prog && PROG_OK;
  NOTE_KERNEL("k_gemv_slices<%d, %d, %d, %d, %d, %d, %d, %d, %d>", TW, NG, U, SPW, CS, CH, wf ? 1 : 0, TAG, prog);
  if (prog) {
    if (wf) k_gemv_slices<TW, NG, U, SPW, CS, CH, 1, TAG, PROG_OK><<<grid, GEMV_THREADS, smem, st>>>(sa);
    else k_gemv_slices<TW, NG, U, SPW, CS, CH, 0, TAG, PROG_OK><<<grid, GEMV_THREADS, smem, st>>>(sa);
  } else {
    if (wf) k_gemv_slices<TW, NG, U, SPW, CS, CH, 1, TAG, 0><<<grid, GEMV_THREADS, smem, st>>>(sa);
    else k_gemv_slices<TW, NG, U, SPW, CS, CH, 0, TAG, 0><<<grid, GEMV_THREADS, smem, st>>>(sa);
  }
  return DD_OK;
}
// (16 instead of 8 weight requests in flight per wave measured the same or slower: qkv 26.3 vs 25.4 us, gate/up 38.4 vs 38.0)
// dd_tools_set_tuning key 49: progressive stage-in of the operand planes (dd_gemv_slices.h PROG) in the whole-slice kernels at two and four planes
// (default 1; 0 = the blocking stage-in of rounds 2-4).  Same bits either way.
int g_seq_prog = 1;
template <int NG, int U, int MAXG, int TAG>
static int launch_slices_seq(const SliceArgs& sa, int wf, hipStream_t st) {
  constexpr size_t smem = (size_t)16 * NG * 1024;
  static bool attr = false;
  if (!attr) {
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_seq<NG, U, 16, MAXG, 0, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_seq<NG, U, 16, MAXG, 1, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  NOTE_KERNEL("k_gemv_slices_seq<%d, %d, 16, %d, %d, %d, 2>", NG, U, MAXG, wf ? 1 : 0, TAG);   // (the last argument: slices per workgroup, pairs)
  if (wf) k_gemv_slices_seq<NG, U, 16, MAXG, 1, TAG><<<4 * sa.G + rstd_blocks(sa), GEMV_THREADS, smem, st>>>(sa);
  else k_gemv_slices_seq<NG, U, 16, MAXG, 0, TAG><<<4 * sa.G + rstd_blocks(sa), GEMV_THREADS, smem, st>>>(sa);
  return DD_OK;
}

#ifdef DD_TIMING_EXPERIMENTS
// (tools library only) QUADS: four slices added up inside a workgroup, two partial sums per tile instead of four — the sums come out in an order
// no other pass width produces, so this is a timing experiment (dd_tools_set_tuning key 53: 1 = 96 / 86 workgroups per quad for
// qkv / gate-up (one / two tiles per wave), 2 = gate/up on 128 per quad)
int g_seq_quads = 0;
template <int NG, int U, int MAXG, int TAG>
static int launch_slices_seq_quads(const SliceArgs& sa, int wf, hipStream_t st) {
  constexpr size_t smem = (size_t)16 * NG * 1024;
  static bool attr = false;
  if (!attr) {
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_seq<NG, U, 16, MAXG, 0, TAG, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  NOTE_KERNEL("k_gemv_slices_seq<%d, %d, 16, %d, 0, %d, 4>", NG, U, MAXG, TAG);
  DD_REQUIRE(!wf, "quads: bf16 tiles only");
  k_gemv_slices_seq<NG, U, 16, MAXG, 0, TAG, 4><<<2 * sa.G + rstd_blocks(sa), GEMV_THREADS, smem, st>>>(sa);
  return DD_OK;
}
#endif
int g_finish4 = 15;   // dd_tools_set_tuning key 24: the four-columns-per-thread finishing kernel per epilogue (bit = EPI_*; 0: k_gemv_finish, same bits)
template <int EPI, int TILES, int NG, int NP>
static void launch_finish(const GemvArgs& a, int n_sets, hipStream_t st) {
  if (g_slices_only) return;
  if ((g_finish4 & (1 << EPI)) || a.fp8) k_gemv_finish4<EPI, TILES, NG, NP><<<n_sets, 32 * NG, 0, st>>>(a, a.part, a.part + a.part_floats, n_sets);
  else k_gemv_finish<EPI, TILES, NG, NP><<<n_sets, 128 * NG, 0, st>>>(a, a.part, a.part + a.part_floats, n_sets);
}
template <int NG, int SPW2, int CH, int UW>
static int launch_slices_fp8(const SliceArgs& sa, hipStream_t st) {
  constexpr size_t smem = (size_t)CH * 2 * SPW2 * NG * 1024;
  static bool attr = false;
  if (!attr) {
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_fp8<NG, SPW2, CH, UW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  NOTE_KERNEL("k_gemv_slices_fp8<%d, %d, %d, %d, 0, 0>", NG, SPW2, CH, UW);
  k_gemv_slices_fp8<NG, SPW2, CH, UW><<<(8 / CH) * sa.G + rstd_blocks(sa), GEMV_THREADS, smem, st>>>(sa);
  return DD_OK;
}
int g_fp8_xpf = 1;       // dd_tools_set_tuning key 52: the nine-plane fp8 kernel's operand fragments through a ring, five reads ahead of their MFMAs
                         // (default; 0: requested where the compiler puts them — one MFMA ahead; same bits)
template <int NG, int SPW2, int CS2, int UW>
static int launch_slices_fp8c(const SliceArgs& sa, hipStream_t st) {
  constexpr size_t smem = (size_t)2 * CS2 * NG * 1024;
  static bool attr = false;
  if (!attr) {
    DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_fp8c<NG, SPW2, CS2, UW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  NOTE_KERNEL("k_gemv_slices_fp8c<%d, %d, %d, %d, 0>", NG, SPW2, CS2, UW);
  k_gemv_slices_fp8c<NG, SPW2, CS2, UW><<<8 * sa.G + rstd_blocks(sa), GEMV_THREADS, smem, st>>>(sa);
  return DD_OK;
}
template <int EPI, int TILES, int NG>
static void finish_fp8(const GemvArgs& a, int n_sets, int ch, hipStream_t st) {
  if (ch == 2) launch_finish<EPI, TILES, NG, 4>(a, n_sets, st);
  else launch_finish<EPI, TILES, NG, 8>(a, n_sets, st);
}
// fp8 tiles (weight_format 1): K = 4096 (8 steps of 64 k per slice: slice pairs up to four planes, single slices at eight) and,
// for two planes, K = 14336 (28 steps); other shapes stay on the wave-split kernels — the same bits either way
template <int NG>
static int try_slices_fp8(int epi, const GemvArgs& a, hipStream_t st) {
  const int nt = epi == EPI_SILU ? 2 * a.n_tiles : a.n_tiles;
  const int spw2 = a.S / 16;
  if (a.S % 16 || nt < 64 || !(spw2 == 8 || spw2 == 28)) return SLICES_UNSUPPORTED;
  const int ch = (spw2 == 8 && NG <= 4) ? 2 : 1;
  if (a.part_floats < (size_t)(8 / ch) * nt * NG * 128) return SLICES_UNSUPPORTED;
  SliceArgs sa;
  sa.W = a.W, sa.xop = a.xop, sa.part = a.part, sa.S = a.S, sa.halves = 1, sa.n_groups = nt;
  sa.ssq_in = a.ssq_in, sa.ssq_n = a.ssq_n, sa.ssq_ld = a.ssq_ld, sa.inv_k = a.inv_k, sa.eps = a.eps;
  sa.rstd_out = a.part + a.part_floats, sa.temporal = g_exp_temporal, sa.rstd_wg = g_rstd_wg;
  const int per_set = 256 / (8 / ch);                  // one round of workgroups (one per CU at 64-128 KiB of operands)
  sa.G = (nt + 7) / 8 < per_set ? (nt + 7) / 8 : per_set;
  if (spw2 == 28) {
    // K = 14336 (down_proj of Mistral-7B): two planes hold a whole slice in LDS; four / eight planes stage it in chunks, one tile per wave
    if constexpr (NG == 2) {
      RC_(launch_slices_fp8<2, 28, 1, 7>(sa, st));
    } else {
      sa.G = (nt + 7) / 8;
      if constexpr (NG == 4) RC_(launch_slices_fp8c<4, 28, 7, 7>(sa, st));
      else RC_(launch_slices_fp8c<8, 28, 4, 4>(sa, st));
    }
  } else if constexpr (NG <= 4) {
    RC_(launch_slices_fp8<NG, 8, 2, 8>(sa, st));
  } else {
    RC_(launch_slices_fp8<NG, 8, 1, 8>(sa, st));
  }
  switch (epi) {
    case EPI_STORE: finish_fp8<EPI_STORE, 1, NG>(a, nt, ch, st); break;
    case EPI_RESID: finish_fp8<EPI_RESID, 1, NG>(a, nt, ch, st); break;
    case EPI_SILU: finish_fp8<EPI_SILU, 2, NG>(a, a.n_tiles, ch, st); break;
    default: finish_fp8<EPI_QKV, 1, NG>(a, nt, ch, st); break;
  }
  return DD_OK;
}

template <int NG>
static int try_slices(int epi, const GemvArgs& a, hipStream_t st) {
  if (a.fp8) return try_slices_fp8<NG>(epi, a, st);
  const int spw = a.S / GEMV_WAVES;
  const int nt = epi == EPI_SILU ? 2 * a.n_tiles : a.n_tiles;      // 16-row weight tiles
  if (!(spw == 16 || spw == 43 || spw == 56) || nt < 64) return SLICES_UNSUPPORTED;
  SliceArgs sa;
  sa.W = a.W, sa.xop = a.xop, sa.part = a.part, sa.S = a.S, sa.halves = 1;
  sa.ssq_in = a.ssq_in, sa.ssq_n = a.ssq_n, sa.ssq_ld = a.ssq_ld, sa.inv_k = a.inv_k, sa.eps = a.eps;
  sa.rstd_out = a.part + a.part_floats, sa.temporal = g_exp_temporal, sa.rstd_wg = g_rstd_wg;                              // 32 floats behind the partial sums
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
        DD_REQUIRE((nt + 8 * sa.G - 1) / (8 * sa.G) <= 2, "gemv_slices_seq: %d tiles over %d workgroups per pair", nt, sa.G);
        RC_(launch_slices_seq<8, 8, 2, EPI_QKV>(sa, a.wf, st));
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
        DD_REQUIRE((nt + 8 * sa.G - 1) / (8 * sa.G) <= 3, "gemv_slices_seq: %d tiles over %d workgroups per pair", nt, sa.G);
        RC_(launch_slices_seq<8, 8, 3, EPI_SILU>(sa, a.wf, st));
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

// Nine operand planes (72 rows): the members of eight sequences + ONE plane of un-masked rows that ride along (dd_engine.hip,
// "rider": the base rows of the partner group's next pass, so that a group step needs no sweep of its own for them).  The 64-row
// kernels with one more plane: a slice of K = 4096 is 144 KiB of LDS, one workgroup per CU as there; bf16 / fp16 tiles only, the
// 7B families' shapes only (the engine does not plan riders otherwise).
// (fp8 tiles: K = 4096 — 144 KiB of operands per slice, single slices — and K = 14336 in chunks; Mistral-7B's shapes, config 5)
static int try_slices9_fp8(int epi, const GemvArgs& a, hipStream_t st) {
  const int nt = epi == EPI_SILU ? 2 * a.n_tiles : a.n_tiles;
  const int spw2 = a.S / 16;
  if (a.S % 16 || nt < 64 || !(spw2 == 8 || (spw2 == 28 && epi == EPI_RESID))) return SLICES_UNSUPPORTED;
  if (a.part_floats < (size_t)8 * nt * 9 * 128) return SLICES_UNSUPPORTED;
  SliceArgs sa;
  sa.W = a.W, sa.xop = a.xop, sa.part = a.part, sa.S = a.S, sa.halves = 1, sa.n_groups = nt;
  sa.ssq_in = a.ssq_in, sa.ssq_n = a.ssq_n, sa.ssq_ld = a.ssq_ld, sa.inv_k = a.inv_k, sa.eps = a.eps;
  sa.rstd_out = a.part + a.part_floats, sa.temporal = g_exp_temporal, sa.rstd_wg = g_rstd_wg;
  if (spw2 == 28) {
    sa.G = (nt + 7) / 8;
    RC_(launch_slices_fp8c<9, 28, 4, 4>(sa, st));
  } else {
    sa.G = (nt + 7) / 8 < 32 ? (nt + 7) / 8 : 32;        // one round of workgroups (one per CU: 144 KiB of operands each)
    if (g_fp8_xpf) {
      constexpr size_t smem = (size_t)2 * 8 * 9 * 1024;
      static bool attr = false;
      if (!attr) {
        DD_HIP(hipFuncSetAttribute((const void*)k_gemv_slices_fp8<9, 8, 1, 8, 0, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr = true;
      }
      NOTE_KERNEL("k_gemv_slices_fp8<9, 8, 1, 8, 0, 6>");
      k_gemv_slices_fp8<9, 8, 1, 8, 0, 6><<<8 * sa.G + rstd_blocks(sa), GEMV_THREADS, smem, st>>>(sa);
    } else
    RC_(launch_slices_fp8<9, 8, 1, 8>(sa, st));
  }
  switch (epi) {
    case EPI_STORE: launch_finish<EPI_STORE, 1, 9, 8>(a, nt, st); break;
    case EPI_RESID: launch_finish<EPI_RESID, 1, 9, 8>(a, nt, st); break;
    case EPI_SILU: launch_finish<EPI_SILU, 2, 9, 8>(a, a.n_tiles, st); break;
    default: launch_finish<EPI_QKV, 1, 9, 8>(a, nt, st); break;
  }
  return DD_OK;
}
static int try_slices9(int epi, const GemvArgs& a, hipStream_t st) {
  if (a.fp8) return try_slices9_fp8(epi, a, st);
  const int spw = a.S / GEMV_WAVES;
  const int nt = epi == EPI_SILU ? 2 * a.n_tiles : a.n_tiles;
  if (!(spw == 16 || spw == 43 || spw == 56) || nt < 64) return SLICES_UNSUPPORTED;
  SliceArgs sa;
  sa.W = a.W, sa.xop = a.xop, sa.part = a.part, sa.S = a.S, sa.halves = 1;
  sa.ssq_in = a.ssq_in, sa.ssq_n = a.ssq_n, sa.ssq_ld = a.ssq_ld, sa.inv_k = a.inv_k, sa.eps = a.eps;
  sa.rstd_out = a.part + a.part_floats, sa.temporal = g_exp_temporal, sa.rstd_wg = g_rstd_wg;
  const size_t need8 = (size_t)8 * nt * 9 * 128;
  if (a.part_floats < need8) return SLICES_UNSUPPORTED;
  sa.n_groups = nt;
  if (epi == EPI_STORE) {
    if (spw != 16) return SLICES_UNSUPPORTED;
    sa.G = (nt + 31) / 32;
    RC_(launch_slices_k<1, 9, 8, 16, 16, 1, EPI_STORE>(sa, a.wf, st));
    launch_finish<EPI_STORE, 1, 9, 8>(a, nt, st);
  } else if (epi == EPI_QKV) {
    if (spw != 16 || (nt % 16) != 0 || nt / 16 * 4 > 256) return SLICES_UNSUPPORTED;
#ifdef DD_TIMING_EXPERIMENTS
    if (g_seq_quads && !a.wf && nt % 8 == 0 && nt / 8 <= 128) {
      sa.G = nt / 8;                                     // one tile per wave, 2 * G workgroups
      RC_(launch_slices_seq_quads<9, 4, 1, EPI_QKV>(sa, a.wf, st));
      launch_finish<EPI_QKV, 1, 9, 2>(a, nt, st);
      return DD_OK;
    }
#endif
    sa.G = g_exp_G[0] > 0 ? g_exp_G[0] : nt / 16;
    DD_REQUIRE((nt + 8 * sa.G - 1) / (8 * sa.G) <= 2, "gemv_slices_seq: %d tiles over %d workgroups per pair", nt, sa.G);
    if (g_exp_U9 == 8) RC_(launch_slices_seq<9, 8, 2, EPI_QKV>(sa, a.wf, st));
    else RC_(launch_slices_seq<9, 4, 2, EPI_QKV>(sa, a.wf, st));
    launch_finish<EPI_QKV, 1, 9, 4>(a, nt, st);
  } else if (epi == EPI_RESID) {
    if (spw == 16 && g_exp_G[1] == -2 && (nt + 8 * 32 - 1) / (8 * 32) <= 1) {
      // (A/B, tuning key 18 = -2) slice PAIRS, one slice resident at a time, one tile per wave on 4 * 32 = 128 workgroups: half the partial sums
      sa.G = 32;
      RC_(launch_slices_seq<9, 8, 1, EPI_RESID>(sa, a.wf, st));
      launch_finish<EPI_RESID, 1, 9, 4>(a, nt, st);
      return DD_OK;
    }
    if (spw == 16) {
      // two tiles per wave on 8 * 16 = 128 workgroups at N = 4096: alone 14.7 us against 11.2 with one tile per wave on 256, but inside
      // the step, beside the other branch's kernels, the half-chip grid wins (20.9 vs 21.4 ms per 32-lane step; tuning key 18)
      sa.G = g_exp_G[1] > 0 ? g_exp_G[1] : (nt + 15) / 16;
      RC_(launch_slices_k<1, 9, 8, 16, 16, 1, EPI_RESID>(sa, a.wf, st));
    } else {
      sa.G = (nt + 7) / 8;
      if (spw == 43) RC_(launch_slices_k<1, 9, 8, 43, 8, 1, EPI_RESID>(sa, a.wf, st));
      else RC_(launch_slices_k<1, 9, 8, 56, 8, 1, EPI_RESID>(sa, a.wf, st));
    }
    launch_finish<EPI_RESID, 1, 9, 8>(a, nt, st);
  } else if (spw == 16 && (4 * ((nt + 23) / 24) > 256 || g_exp_G[2] == -2)) {      // (A/B, tuning key 19 = -2: single slices for every gate/up)
    // more gate/up tiles than three per wave of one round of slice PAIRS (Mistral-7B: 1792 tiles): single slices, tiles walked per wave
    sa.G = 32;
    RC_(launch_slices_k<1, 9, 8, 16, 16, 1, EPI_SILU>(sa, a.wf, st));
    launch_finish<EPI_SILU, 2, 9, 8>(a, a.n_tiles, st);
  } else {
    if (spw != 16) return SLICES_UNSUPPORTED;
#ifdef DD_TIMING_EXPERIMENTS
    if (g_seq_quads && !a.wf) {
      sa.G = g_seq_quads == 2 ? 128 : (nt + 15) / 16;    // two tiles per wave
      if ((nt + 8 * sa.G - 1) / (8 * sa.G) <= 2) {
        RC_(launch_slices_seq_quads<9, 4, 2, EPI_SILU>(sa, a.wf, st));
        launch_finish<EPI_SILU, 2, 9, 2>(a, a.n_tiles, st);
        return DD_OK;
      }
    }
#endif
    sa.G = g_exp_G[2] > 0 ? g_exp_G[2] : (nt + 23) / 24;
    DD_REQUIRE(g_exp_G[2] == -3 || (nt + 8 * sa.G - 1) / (8 * sa.G) <= 3, "gemv_slices_seq: %d tiles over %d workgroups per pair", nt, sa.G);
    // (four weight requests in flight per wave: with eight, three tiles per wave and nine planes the kernel needs 257 registers and
    // spills 28 bytes per lane to private scratch — that instantiation is gone, build.py refuses kernels with scratch)
    if (g_exp_G[2] == -3) {
      // (A/B, tuning key 19 = -3) two tiles per wave on 4 * ceil(nt / 16) workgroups (11008: 344, more than one per CU): 196 instead of 236
      // VGPRs, so that the rider sweeps' attention (94) fits beside this kernel too
      sa.G = (nt + 15) / 16;
      RC_(launch_slices_seq<9, 4, 2, EPI_SILU>(sa, a.wf, st));
    } else if (g_exp_U9 == 8) {
      // (tuning key 29 = 8) eight weight requests in flight per wave: possible since the folded sums of two tiles share a register set (round 5)
      RC_(launch_slices_seq<9, 8, 3, EPI_SILU>(sa, a.wf, st));
    } else
    RC_(launch_slices_seq<9, 4, 3, EPI_SILU>(sa, a.wf, st));
    launch_finish<EPI_SILU, 2, 9, 4>(a, a.n_tiles, st);
  }
  return DD_OK;
}

int ddk_gemv_groups(int epi, const GemvArgs& a, hipStream_t st) {
  DD_REQUIRE(a.S % GEMV_WAVES == 0 && a.S >= GEMV_WAVES, "gemv_groups: K=%d must be a multiple of 256", a.S * 32);
  DD_REQUIRE(a.nb >= 1 && a.nb <= 8, "gemv_groups: nb=%d rows per group", a.nb);
  DD_REQUIRE(!a.fp8 || a.wscale, "gemv_groups: fp8 weights need row scales");
  DD_REQUIRE(a.n_groups == 2 || a.n_groups == 4 || a.n_groups == 8 || a.n_groups == 9, "gemv_groups: %d groups (2, 4, 8 or 9)", a.n_groups);
  DD_REQUIRE(!a.half_planes || (((a.n_groups == 8 && a.half_planes == 8) || (a.n_groups == 9 && a.half_planes == 7)) && a.nb <= 4 && !a.fp8 && a.part && g_gemv_slices),
             "gemv_groups: half planes take an eight- or nine-plane pass of 16-bit weights, K <= 4");
  if (a.n_groups == 9) {
    int rs = a.part ? try_slices9(epi, a, st) : SLICES_UNSUPPORTED;
    DD_REQUIRE(rs != SLICES_UNSUPPORTED, "gemv_groups: no nine-plane kernel for this matrix (K = %d, %d tiles, fp8 %d)", a.S * 32, a.n_tiles, a.fp8);
    if (rs != DD_OK) return rs;
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
  if (g_gemv_slices && a.part) {
    int rs = a.n_groups == 2 ? try_slices<2>(epi, a, st) : (a.n_groups == 4 ? try_slices<4>(epi, a, st) : try_slices<8>(epi, a, st));
    if (rs != SLICES_UNSUPPORTED) {
      if (rs != DD_OK) return rs;
      DD_CHECK_LAUNCH();
      return DD_OK;
    }
  }
  DD_REQUIRE(!a.half_planes, "gemv_groups: no eight-plane kernel for this matrix (half planes have no fallback)");
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


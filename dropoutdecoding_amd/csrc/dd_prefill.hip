// Prefill kernels (gfx950, wave64): RMSNorm + hi/lo split, the MFMA GEMMs over the packed weights (128 x 128 and the
// LDS-staged 128 x 512 block), causal / bidirectional prefill attention (VALU and matrix-core forms).
// Reference anchors: the prompt forward the reference calls at models/llava.py:294-303 (and the vision towers at :229-250).
#include <type_traits>

#include "dd_lm_kernels.h"
#include "dd_lm_device.h"

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
  for (int i = threadIdx.x; i < d; i += 256) ss = __builtin_fmaf(xr[i], xr[i], ss);
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
// The same values for SIXTEEN rows per workgroup (one row of operand tiles): k_rmsnorm_split's 16-byte stores of one row land in 512
// different 1 KiB tiles — 16 of a tile's 64 pieces come from 16 workgroups at 16 different times, and the planes reached HBM at 1.3 TB/s
// (72 us for the 2960 x 4096 rows of a NeXT prompt, 2.7 % of config 5; round 4).  Here a wave writes whole tiles, 1 KiB per store.
// Bit for bit k_rmsnorm_split: a row's sum of squares is its 256 partial sums (element t + 256 j into partial t, j ascending, fused
// multiply-add) reduced by the xor butterfly 32 .. 1 within each 64 and the four results added in order — lane L of the row's wave holds
// partials 4 L .. 4 L + 3, so the butterfly's steps 32 .. 4 are lane steps 8 .. 1 and its steps 2, 1 are register pairs (additions commute,
// every partner pair forms the same sum) —, and y = w * (x * rstd) is the same expression.  d % 256 == 0; rows beyond M untouched.
__global__ __launch_bounds__(512) void k_rmsnorm_split16(const float* __restrict__ x, int M, int d, const float* __restrict__ w, float eps,
                                                         uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int wf) {
  __shared__ float sh_rstd[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 16;
  const int nj = d >> 8;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int row = row0 + 2 * wave + rr;
    if (row < M) {
      const f32x4_t* xr = (const f32x4_t*)(x + (size_t)row * d) + lane;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int j0 = 0; j0 < nj; j0 += 8) {
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u < nj) v[u] = xr[(size_t)(j0 + u) * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u < nj) {
            s0 = __builtin_fmaf(v[u].x, v[u].x, s0);
            s1 = __builtin_fmaf(v[u].y, v[u].y, s1);
            s2 = __builtin_fmaf(v[u].z, v[u].z, s2);
            s3 = __builtin_fmaf(v[u].w, v[u].w, s3);
          }
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
        s3 += __shfl_xor(s3, o);
      }
      const float a0 = s0 + s2, a1 = s1 + s3;            // butterfly step 2: partial t with t ^ 2
      const float q = a0 + a1;                           // step 1; lanes 16 v .. 16 v + 15 hold the sum of partials 64 v .. 64 v + 63
      const float q0 = __shfl(q, 0), q1 = __shfl(q, 16), q2 = __shfl(q, 32), q3 = __shfl(q, 48);
      if (lane == 0) sh_rstd[2 * wave + rr] = 1.0f / sqrtf((q0 + q1 + q2 + q3) / (float)d + eps);
    }
  }
  __syncthreads();
  const int S = d >> 5;
  const int r = lane & 15, hq = lane >> 4;
  const int row = row0 + r;
  if (row >= M) return;
  const float rstd = sh_rstd[r];
  const float* xr = x + (size_t)row * d + 8 * hq;
  const float* wr = w + 8 * hq;
  u32x4_t* th = (u32x4_t*)hi + (size_t)blockIdx.x * S * 64 + lane;
  u32x4_t* tl = (u32x4_t*)lo + (size_t)blockIdx.x * S * 64 + lane;
  for (int ks0 = wave; ks0 < S; ks0 += 8 * 4) {
    f32x4_t xa[4][2], wa[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ks = ks0 + 8 * u;
      if (ks < S) {
        xa[u][0] = *(const f32x4_t*)(xr + ks * 32);
        xa[u][1] = *(const f32x4_t*)(xr + ks * 32 + 4);
        wa[u][0] = *(const f32x4_t*)(wr + ks * 32);
        wa[u][1] = *(const f32x4_t*)(wr + ks * 32 + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ks = ks0 + 8 * u;
      if (ks < S) {
        uint32_t hh[8], ll[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float y = wa[u][j >> 2][j & 3] * (xa[u][j >> 2][j & 3] * rstd);
          dd_split(y, hh[j], ll[j], wf);
        }
        u32x4_t vh, vl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vh[j] = hh[2 * j] | (hh[2 * j + 1] << 16);
          vl[j] = ll[2 * j] | (ll[2 * j + 1] << 16);
        }
        th[(size_t)ks * 64] = vh;
        tl[(size_t)ks * 64] = vl;
      }
    }
  }
}
int g_rmsnorm16 = 1;             // dd_tools_set_tuning key 45: 0 = one row per workgroup everywhere (the round-1..3 kernel)
int ddk_rmsnorm_split(const float* x, int M, int d, const float* w, float eps, uint16_t* hi, uint16_t* lo,
                      const int32_t* row_index, float* normed, hipStream_t st, int wf) {
  if (g_rmsnorm16 && hi && lo && !row_index && !normed && (d & 255) == 0 && M >= 64) {
    k_rmsnorm_split16<<<(M + 15) / 16, 512, 0, st>>>(x, M, d, w, eps, hi, lo, wf);
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
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
      float yv[EPI == EPI_STORE ? NJ : 1];                           // EPI_STORE: the row's values of the wave's tiles, for the fused row statistics
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
          yv[EPI == EPI_STORE ? j : 0] = (wv[j] && col < a.n_valid) ? y : -INFINITY;
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
      if constexpr (EPI == EPI_STORE && (NJ % 4) == 0) {
        if (a.rowstat) {                                             // (uniform: every lane of the wave takes part in the shuffles)
          dd_static_for<0, NJ / 4>([&](auto sb_) {
            constexpr int sb = decltype(sb_)::value;
            const float x4[4] = {yv[4 * sb], yv[4 * sb + 1], yv[4 * sb + 2], yv[4 * sb + 3]};
            float m, sum;
            dd_row_block_stats(x4, m, sum);
            const int cb = (nt_base >> 2) + sb;
            if (c == 0 && row < a.M && cb < a.rowstat_ld) {
              a.rowstat[((size_t)row * a.rowstat_ld + cb) * 2] = m;
              a.rowstat[((size_t)row * a.rowstat_ld + cb) * 2 + 1] = sum;
            }
          });
        }
      }
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

// Round 6: the same product with the operand fragments brought in by LDS-DMA (`global_load_lds_dwordx4`: global memory -> LDS with no
// register in between) and a 160 x 512 block.  The pre-tiled fragments are lane-linear 1 KiB pieces, which is exactly what one LDS-DMA
// wave instruction writes (wave-uniform LDS base + 16 bytes per lane), so the copy needs no registers, no ds_write pass and no
// `s_waitcnt` before the LDS writes: k_gemm_big's 24 staging VGPRs go to a fifth row of accumulator tiles per wave (MTB = 10: 5 x 8
// tiles, 160 accumulator registers).  Why 160 rows: a prefill group is 16 x 640 rows; with 128-row blocks the N = 4096 matrices
// (o_proj, down_proj) make 80 x 8 = 640 workgroups = 2.5 rounds of the 256 CUs, with 160-row blocks 64 x 8 = 512 = two full rounds
// (qkv: 7.5 -> 6 rounds, gate/up 13.4 -> 10.75), and a block's operand bytes per flop drop by 13 %.  Three LDS stages of
// (2 MTB + 32) KiB (156 KiB at MTB = 10): step ks multiplies from stage ks % 3, the A fragments of step ks + 1 are read from stage
// (ks + 1) % 3 behind the step's MFMAs (and its first two W fragments with them: the next step's matrix work then starts right behind
// the barrier), the fragments of step ks + 2 land in stage (ks + 2) % 3 meanwhile.  One raw `s_barrier` per
// step behind `s_waitcnt vmcnt(0)` (the only vector-memory operations in flight are the step's own LDS-DMA pieces; `__syncthreads()`
// would do the same here, the raw form only keeps the compiler from adding its own waits).  Per accumulator tile the MFMA sequence
// is k_gemm's (k ascending; hi then lo): the same bits (tests/test_gpu_engine.py::test_prefill_gemm_block_shapes_and_orders_give_the_same_bits).
template <int EPI, int WF, int MTB>
__global__ __launch_bounds__(512) void k_gemm_dma(GemmArgs a) {
  extern __shared__ __align__(16) u32x4_t gd_sh[];          // [3 stages][FR fragments][64]
  constexpr int MI = MTB / 2, NJ = 8;
  constexpr int FR = 2 * MTB + GB_NT;                        // 1 KiB fragments per k-step: A hi, A lo, W
  constexpr int NC = (FR + 7) / 8;                           // LDS-DMA pieces per wave and k-step
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
  // copy duty of this wave: fragments wave + 8 i (a wave whose last index falls past FR repeats its previous piece: every wave
  // issues the same number of pieces)
  const u32x4_t* src[NC];
  int dst[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    int f = wave + 8 * i;
    if (f >= FR) f -= 8;
    dst[i] = f * 64;
    if (f < MTB) src[i] = (const u32x4_t*)a.a_hi + (size_t)min(by * MTB + f, m_tiles - 1) * S * 64 + lane;
    else if (f < 2 * MTB) src[i] = (const u32x4_t*)a.a_lo + (size_t)min(by * MTB + f - MTB, m_tiles - 1) * S * 64 + lane;
    else src[i] = a.W + (size_t)min(bx * GB_NT + f - 2 * MTB, a.n_tiles - 1) * S * 64 + lane;
  }
  auto issue = [&](int ks, int stage) {
#pragma unroll
    for (int i = 0; i < NC; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)ks * 64),
                                       (__attribute__((address_space(3))) void*)(gd_sh + stage * (FR * 64) + dst[i]), 16, 0, 0);
  };
  const int m_base = by * (16 * MTB) + wr * (16 * MI);
  const int nt_base = bx * GB_NT + wc * NJ;
  bool wv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) wv[j] = (nt_base + j) < a.n_tiles;
  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  issue(0, 0);
  issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  u32x4_t ahi[MI], alo[MI], wq[2][2];
  const int w_off = (2 * MTB + wc * NJ) * 64 + lane;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    ahi[i] = gd_sh[(wr * MI + i) * 64 + lane];
    alo[i] = gd_sh[(MTB + wr * MI + i) * 64 + lane];
  }
  wq[0][0] = gd_sh[w_off];
  wq[0][1] = gd_sh[w_off + 64];
  auto mm = [&](const int jp, const u32x4_t (&w2)[2]) {     // the 4 MI MFMAs of W fragments 2 jp, 2 jp + 1 (per tile: hi, then lo)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][2 * jp + jj] = dd_mfma16<WF>(ahi[i], w2[jj], acc[i][2 * jp + jj]);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][2 * jp + jj] = dd_mfma16<WF>(alo[i], w2[jj], acc[i][2 * jp + jj]);
    }
  };
  int s_cur = 0;                                               // ks % 3
  for (int ks = 0; ks < S; ++ks) {
    const int s_next = s_cur == 2 ? 0 : s_cur + 1, s_next2 = s_next == 2 ? 0 : s_next + 1;
    // At the top the wave holds A(ks) and W fragments 0, 1 of step ks (read before the barrier that ended step ks - 1), so its matrix
    // work restarts at once; the W fragments come two at a time, the next pair requested before the current pair's 4 MI MFMAs issue
    // (left to itself the compiler reads a pair, waits for it with the matrix pipe running dry, multiplies, reads the next).
    const u32x4_t* st = gd_sh + s_cur * (FR * 64) + w_off;
    wq[1][0] = st[2 * 64];
    wq[1][1] = st[3 * 64];
    __builtin_amdgcn_sched_barrier(0);
    mm(0, wq[0]);
    // the pieces of step ks + 2 go out behind the first MFMAs: stage (ks + 2) % 3 held step ks - 1, which every wave finished reading
    // before the barrier that ended it
    if (ks + 2 < S) issue(ks + 2, s_next2);
    wq[0][0] = st[4 * 64];
    wq[0][1] = st[5 * 64];
    __builtin_amdgcn_sched_barrier(0);
    mm(1, wq[1]);
    wq[1][0] = st[6 * 64];
    wq[1][1] = st[7 * 64];
    __builtin_amdgcn_sched_barrier(0);
    mm(2, wq[0]);
    __builtin_amdgcn_sched_barrier(0);
    mm(3, wq[1]);                                              // (s_setprio(1) around the step's MFMAs: measured 2 % slower)
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < S) {                                          // stage (ks + 1) % 3 was published by the barrier that ended step ks - 1
      const u32x4_t* sn = gd_sh + s_next * (FR * 64);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        ahi[i] = sn[(wr * MI + i) * 64 + lane];
        alo[i] = sn[(MTB + wr * MI + i) * 64 + lane];
      }
      wq[0][0] = sn[w_off];
      wq[0][1] = sn[w_off + 64];
    }
    // the pieces of step ks + 2 have had three quarters of this step's MFMAs to land; nothing else of this wave is in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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

static int g_gemm_dma = 1;        // dd_set_tuning key 20: 1 = the LDS-DMA block (round 6; 160 or 128 rows by the launch's rounds; 10 / 8: always that one), 0 = the register-staged 128 x 512 block
void ddk_set_gemm_dma(int v) { g_gemm_dma = (v == 8 || v == 10) ? v : (v ? 1 : 0); }
template <int MTB>
static int launch_gemm_dma(int epi, const GemmArgs& a_, hipStream_t st) {
  GemmArgs a = a_;
  const int gx = (a.n_tiles + GB_NT - 1) / GB_NT;
  a.grid_y = (a.M + 16 * MTB - 1) / (16 * MTB);
  a.xcd_order = g_gemm_xcd_order && a.grid_y <= 8;
  const size_t lds = (size_t)3 * (2 * MTB + GB_NT) * 64 * sizeof(u32x4_t);   // 156 KiB at MTB = 10
  dim3 grid(gx * a.grid_y);
#define GDK(E_, W_)                                                                                                               \
  do {                                                                                                                            \
    static bool attr = false;                                                                                                     \
    if (!attr) {                                                                                                                  \
      DD_HIP(hipFuncSetAttribute((const void*)k_gemm_dma<E_, W_, MTB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
      attr = true;                                                                                                                \
    }                                                                                                                             \
    k_gemm_dma<E_, W_, MTB><<<grid, 512, lds, st>>>(a);                                                                           \
  } while (0)
#define GD(E_)                  \
  if (a.wf) GDK(E_, 1);         \
  else GDK(E_, 0)
  switch (epi) {
    case EPI_STORE: GD(EPI_STORE); break;
    case EPI_RESID: GD(EPI_RESID); break;
    case EPI_SILU: GD(EPI_SILU); break;
    case EPI_QKV: GD(EPI_QKV); break;
    case EPI_ACT: GDK(EPI_ACT, 0); break;
    case EPI_QKV_VIT: GDK(EPI_QKV_VIT, 0); break;
    default: DD_REQUIRE(false, "gemm (160 x 512 block): epilogue %d not built", epi);
  }
#undef GD
#undef GDK
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
  {
    if (!g_gemm_dma) return launch_gemm_big(epi, a, st);
    // 160- or 128-row blocks: whichever needs fewer (rounds of 256 workgroups) x (rows per block); a tie goes to the larger block
    // (fewer operand bytes per flop).  16 x 640 rows: 160 everywhere; one 2,960-row prompt (config 5): gate/up 160, down_proj (152 / 192
    // workgroups, less than one round either way) 128.  g_gemm_dma == 8 / 10 force one of them (A/B).
    const long gx = (a.n_tiles + GB_NT - 1) / GB_NT;
    const long r10 = (gx * ((a.M + 159) / 160) + 255) / 256 * 10, r8 = (gx * ((a.M + 127) / 128) + 255) / 256 * 8;
    const bool ten = g_gemm_dma == 10 || (g_gemm_dma != 8 && r10 <= r8);
    return ten ? launch_gemm_dma<10>(epi, a, st) : launch_gemm_dma<8>(epi, a, st);
  }
  long big = (long)((a.n_tiles + 7) / 8) * ((a.M + 127) / 128);      // workgroups of the 128x128 tiling
  // (a.rowstat: the fused row statistics need a wave that owns whole 64-column blocks — the 4 x 4 and 4 x 8 tilings)
  if (big >= 150 || (epi == EPI_STORE && a.rowstat)) return launch_gemm<4, 4>(epi, a, st);
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
__device__ __forceinline__ void fa_split8p(const float* v, u32x4_t& hi, u32x4_t& lo) {      // fa_split8's bits by packed converts
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t h, l;
    dd_split_hl2(v[2 * j], v[2 * j + 1], h, l);
    hi[j] = h, lo[j] = l;
  }
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

// -----------------------------------------------------------------------------------------------
// The same attention for the fp16 cache and heads of 128 with the K / V tiles staged as MFMA OPERANDS (round 4).  k_attn_prefill_mfma stages
// a tile as fp32 and every wave splits it hi + lo again for its own 16 queries: ~800 of the ~1,000 vector instructions a wave spends per
// 32 keys, and the V^T pieces by 4-byte LDS reads — 0.23 PFLOP/s at 2,960 positions (config 5: 938 us per layer, a sixth of the run).
// Here the staging threads split each value ONCE (the same dd_split_hl of the same fp32 value: the same operand bits) and write the A
// pieces where the waves read them with one ds_read_b128 each:
//   Kop[kt][ks][hi|lo][lane (g4 << 4) | c16] = key 16 kt + c16, d 32 ks + 8 g4 .. + 8   — one 16-byte chunk of the cache's [d/8][T][8]
//   Vop[dt][hi|lo][lane (g4 << 4) | c16]     = d 16 dt + c16, key slots {4 g4 + j, 16 + 4 g4 + j}; an octet of the cache's [T/8][d][8] is the
//                                              4-slot halves of two lanes
// and a wave owns QB blocks of 16 queries that share every operand read (QB = 2: half the LDS bytes per flop).  Per 16-query block the MFMA
// sequence, the softmax and the skipping of key tiles beyond the block are k_attn_prefill_mfma's, so a row's output has the same bits.
// The heaviest workgroups (last queries of a causal prompt) are scheduled first.
// -----------------------------------------------------------------------------------------------
// HD = 128 (the LM), 64 / 96 (the vision towers: bidirectional, Tk keys, q pre-scaled); KV16: fp16 cache layouts, else the fp32 ones
// (K transposed [d/4][T_cap][4], V [T_cap][HD]: fp32-cache engines and the towers' K / V buffers).
template <int G, int QB, int HD = 128, int KV16 = 1>
__global__ __launch_bounds__(256, QB == 1 ? ((KV16 || HD < 128) ? 4 : 3) : 2) void k_attn_prefill_mfma16(const float* qbuf, const float* kc, const float* vc, int T, int T_cap, int n_heads,
                                                             uint16_t* o_hi, uint16_t* o_lo, const uint8_t* __restrict__ drop_plane, int drop_bit,
                                                             int span_start, int span_len, int q0, int causal, float scaling, int wf, int Tk,
                                                             const SeqTab* tab, int seq_rows, size_t off_k, size_t off_v) {
  constexpr int KS = HD / 32, DT = HD / 16, C8 = HD / 8, C4 = HD / 4;
  constexpr int QW = 16 * QB, QG = 4 * QW;
  constexpr int NK = 32 * C8, NV = 4 * HD;                 // staging pieces per tile: K chunks (key, 8 d), V octets (8 keys, d)
  constexpr int ITS = (NK + 255) / 256;
  const int by = gridDim.y - 1 - blockIdx.y;
  if (tab) {
    const int sq = blockIdx.z;
    T = tab->T[sq];
    if (by * QG >= T) return;
    kc = tab->kc[sq] + off_k, vc = tab->vc[sq] + off_v;
    const size_t r0 = (size_t)sq * seq_rows * (n_heads * HD);
    qbuf += r0, o_hi += r0, o_lo += r0;
    Tk = T;
  }
  __shared__ u32x4_t Kop[2][KS][2][64];
  __shared__ u32x4_t Vop[DT][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x, kvh = head / G;
  const int q_dim = n_heads * HD;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int shift = q0 & 15;
  const int blk_first = by * QG - shift;
  const int blk_last = min(blk_first + QG - 1, T - 1);
  if (blk_last < 0) return;
  const int p_max = causal ? q0 + blk_last : Tk - 1;
  int t_q[QB], pos_q[QB], qb_pmax[QB];
  bool qb_live[QB];
  u32x4_t qh[QB][KS], ql[QB][KS];
  f32x4_t acc[QB][DT];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int first = blk_first + wave * QW + qb * 16;
    t_q[qb] = first + c16;
    pos_q[qb] = q0 + max(0, min(t_q[qb], T - 1));
    qb_pmax[qb] = causal ? q0 + min(first + 15, T - 1) : Tk - 1;
    qb_live[qb] = first < T && first + 15 >= 0;
    const float* qr = qbuf + (size_t)max(0, min(t_q[qb], T - 1)) * q_dim + head * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[8];
      *(f32x4_t*)&v[0] = *(const f32x4_t*)(qr + ks * 32 + g4 * 8);
      *(f32x4_t*)&v[4] = *(const f32x4_t*)(qr + ks * 32 + g4 * 8 + 4);
      fa_split8p(v, qh[qb][ks], ql[qb][ks]);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[qb][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    m_run[qb] = -INFINITY, l_run[qb] = 0.f;
  }
  // the staging thread's pieces of a tile: K chunk (key kk = i & 31, d-chunk c = i >> 5) and V octet (dimension dd = i % HD, octet oc = i / HD);
  // the NEXT tile's are requested before this tile's products start, so the cache latency runs under the matrix-core work
  constexpr int RW = KV16 ? 1 : 2;                         // 16-byte registers per piece: 8 halves, or 8 floats
  u32x4_t kraw[ITS][RW], vraw[ITS][RW];
  auto request = [&](int t0) {
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int i = tid + 256 * it;
      if (NK % 256 && i >= NK) break;
      const int kk = i & 31, c = i >> 5, dd = i % HD, oc = i / HD;
      const int key = min(t0 + kk, p_max);
      if constexpr (KV16) {
        kraw[it][0] = *(const u32x4_t*)((const dd_half*)kc + (((size_t)kvh * C8 + c) * T_cap + key) * 8);
        const int octet = min((t0 >> 3) + oc, p_max >> 3);             // keys past p_max inside the last octet carry weight 0
        vraw[it][0] = *(const u32x4_t*)((const dd_half*)vc + (((size_t)kvh * (T_cap >> 3) + octet) * HD + dd) * 8);
      } else {
        kraw[it][0] = *(const u32x4_t*)(kc + (((size_t)kvh * C4 + 2 * c) * T_cap + key) * 4);
        kraw[it][1] = *(const u32x4_t*)(kc + (((size_t)kvh * C4 + 2 * c + 1) * T_cap + key) * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          vraw[it][j >> 2][j & 3] = __float_as_uint(vc[((size_t)kvh * T_cap + min(t0 + 8 * oc + j, p_max)) * HD + dd]);
      }
    }
  };
  auto piece = [&](const u32x4_t (&raw)[RW], float* v) {
    if constexpr (KV16) {
      const f16x8_t h = __builtin_bit_cast(f16x8_t, raw[0]);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __uint_as_float(raw[j >> 2][j & 3]);
    }
  };
  request(0);
  for (int t0 = 0; t0 <= p_max; t0 += FA_KEYS) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int i = tid + 256 * it;
      if (NK % 256 && i >= NK) break;
      {
        const int kk = i & 31, c = i >> 5;
        float v[8];
        piece(kraw[it], v);
        u32x4_t hi, lo;
        fa_split8p(v, hi, lo);
        const int ln = ((c & 3) << 4) | (kk & 15);
        Kop[kk >> 4][c >> 2][0][ln] = hi;
        Kop[kk >> 4][c >> 2][1][ln] = lo;
      }
      {
        const int dd = i % HD, oc = i / HD;
        float v[8];
        piece(vraw[it], v);
        u32x4_t hi, lo;
        fa_split8p(v, hi, lo);
        const int half = oc >> 1, ga = (2 * oc) & 3;
        uint32_t* ph0 = (uint32_t*)&Vop[dd >> 4][0][(ga << 4) | (dd & 15)] + 2 * half;
        uint32_t* pl0 = (uint32_t*)&Vop[dd >> 4][1][(ga << 4) | (dd & 15)] + 2 * half;
        *(u32x2_t*)ph0 = (u32x2_t){hi[0], hi[1]};
        *(u32x2_t*)pl0 = (u32x2_t){lo[0], lo[1]};
        *(u32x2_t*)(ph0 + 16 * 4) = (u32x2_t){hi[2], hi[3]};      // lane group ga + 1: 16 lanes of 16 bytes on
        *(u32x2_t*)(pl0 + 16 * 4) = (u32x2_t){lo[2], lo[3]};
      }
    }
    if (t0 + FA_KEYS <= p_max) request(t0 + FA_KEYS);
    __syncthreads();
    bool use[QB], any = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) use[qb] = qb_live[qb] && t0 <= qb_pmax[qb], any = any || use[qb];
    if (!any) continue;
    f32x4_t sacc[QB][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) sacc[qb][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4_t kh = Kop[kt][ks][0][lane], kl = Kop[kt][ks][1][lane];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          if (use[qb]) {
            sacc[qb][kt] = FA_MFMA(kh, qh[qb][ks], sacc[qb][kt]);
            sacc[qb][kt] = FA_MFMA(kl, qh[qb][ks], sacc[qb][kt]);
            sacc[qb][kt] = FA_MFMA(kh, ql[qb][ks], sacc[qb][kt]);
          }
      }
    }
    u32x4_t ph[QB], pl[QB];
    float corr[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (!use[qb]) continue;
      float sv[8];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = t0 + kt * 16 + g4 * 4 + r;
          bool ok = causal ? key <= pos_q[qb] : key < Tk;
          if (ok && drop_plane && key >= span_start && key < span_start + span_len) ok = !((drop_plane[key - span_start] >> drop_bit) & 1);
          sv[kt * 4 + r] = ok ? sacc[qb][kt][r] * scaling : -INFINITY;
        }
      float mx = sv[0];
#pragma unroll
      for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sv[j]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[qb], mx);
      float p[8], ps = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p[j] = (sv[j] == -INFINITY) ? 0.f : expf(sv[j] - m_new);
        ps += p[j];
      }
      ps += __shfl_xor(ps, 16);
      ps += __shfl_xor(ps, 32);
      corr[qb] = (m_run[qb] == -INFINITY) ? 0.f : expf(m_run[qb] - m_new);
      l_run[qb] = l_run[qb] * corr[qb] + ps;
      m_run[qb] = m_new;
      fa_split8p(p, ph[qb], pl[qb]);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const u32x4_t vh = Vop[dt][0][lane], vl = Vop[dt][1][lane];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
        if (use[qb]) {
          f32x4_t a = acc[qb][dt] * corr[qb];
          a = FA_MFMA(vh, ph[qb], a);
          a = FA_MFMA(vl, ph[qb], a);
          a = FA_MFMA(vh, pl[qb], a);
          acc[qb][dt] = a;
        }
    }
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    if (!(t_q[qb] >= 0 && t_q[qb] < T)) continue;
    const float inv = 1.0f / l_run[qb];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      uint32_t hh[4], ll[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) dd_split(acc[qb][dt][r] * inv, hh[r], ll[r], wf);
      const size_t o = apack_off(t_q[qb], head * HD + dt * 16 + g4 * 4, q_dim >> 5);
      *(u32x2_t*)(o_hi + o) = (u32x2_t){hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16)};
      *(u32x2_t*)(o_lo + o) = (u32x2_t){ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16)};
    }
  }
}
int g_prefill_attn16 = 1;         // dd_tools_set_tuning key 46: query blocks per wave of k_attn_prefill_mfma16 (1 / 2); 0 = k_attn_prefill_mfma (fp32-staged tiles)
template <int QB, int KV16>
static int launch_prefill_mfma16(int G, dim3 grid, hipStream_t st, const float* qbuf, const float* kc, const float* vc, int T, int T_cap, int n_heads,
                                 uint16_t* o_hi, uint16_t* o_lo, const uint8_t* drop_plane, int drop_bit, int span_start, int span_len, int q0,
                                 int wf, const SeqTab* tab, int seq_rows, size_t off_k, size_t off_v) {
#define F16_ARGS qbuf, kc, vc, T, T_cap, n_heads, o_hi, o_lo, drop_plane, drop_bit, span_start, span_len, q0, 1, 0.08838834764831845f, wf, T, tab, seq_rows, off_k, off_v
  if (G == 1) k_attn_prefill_mfma16<1, QB, 128, KV16><<<grid, 256, 0, st>>>(F16_ARGS);
  else if (G == 2) k_attn_prefill_mfma16<2, QB, 128, KV16><<<grid, 256, 0, st>>>(F16_ARGS);
  else if (G == 4) k_attn_prefill_mfma16<4, QB, 128, KV16><<<grid, 256, 0, st>>>(F16_ARGS);
  else DD_REQUIRE(false, "attn_prefill: GQA group %d unsupported", G);
#undef F16_ARGS
  DD_CHECK_LAUNCH();
  return DD_OK;
}
template <int KV16>
static int launch_prefill_staged(int G, int n_heads, int rows, int nz, hipStream_t st, const float* qbuf, const float* kc, const float* vc, int T, int T_cap,
                                 uint16_t* o_hi, uint16_t* o_lo, const uint8_t* drop_plane, int drop_bit, int span_start, int span_len, int q0,
                                 int wf, const SeqTab* tab, int seq_rows, size_t off_k, size_t off_v) {
  const int qg = 64 * (g_prefill_attn16 == 1 ? 1 : 2);
  dim3 g3(n_heads, (rows + qg - 1) / qg, nz);
  if (g_prefill_attn16 == 1)
    return launch_prefill_mfma16<1, KV16>(G, g3, st, qbuf, kc, vc, T, T_cap, n_heads, o_hi, o_lo, drop_plane, drop_bit, span_start, span_len, q0, wf, tab, seq_rows, off_k, off_v);
  return launch_prefill_mfma16<2, KV16>(G, g3, st, qbuf, kc, vc, T, T_cap, n_heads, o_hi, o_lo, drop_plane, drop_bit, span_start, span_len, q0, wf, tab, seq_rows, off_k, off_v);
}

// bidirectional attention of the CLIP tower (head_dim 64, q pre-scaled by the QKV epilogue), same kernel
int ddk_attn_vit_mfma(const float* q, const float* kt, const float* v, int T, int Tc, int n_heads, uint16_t* o_hi, uint16_t* o_lo,
                      hipStream_t st, int head_pitch, int Tk, float scaling) {
  DD_REQUIRE(head_pitch == 64 || head_pitch == 96, "attn_vit: head pitch %d (64, or 96 for 88-wide heads)", head_pitch);
  if (Tk <= 0) Tk = T;
  if (g_prefill_attn16) {
#define VS_ARGS q, kt, v, T, Tc, n_heads, o_hi, o_lo, nullptr, 0, 0, 0, 0, 0, scaling, 0, Tk, nullptr, 0, 0, 0
    if (head_pitch == 96) k_attn_prefill_mfma16<1, 1, 96, 0><<<dim3(n_heads, (T + 63) / 64), 256, 0, st>>>(VS_ARGS);
    else k_attn_prefill_mfma16<1, 1, 64, 0><<<dim3(n_heads, (T + 63) / 64), 256, 0, st>>>(VS_ARGS);
#undef VS_ARGS
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
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
  if (g_prefill_attn16) {
#define VB_ARGS q, nullptr, nullptr, T, Tc, n_heads, o_hi, o_lo, nullptr, 0, 0, 0, 0, 0, 1.0f, 0, T, tab, img_rows, 0, 0
    if (head_pitch == 96) k_attn_prefill_mfma16<1, 1, 96, 0><<<grid, 256, 0, st>>>(VB_ARGS);
    else k_attn_prefill_mfma16<1, 1, 64, 0><<<grid, 256, 0, st>>>(VB_ARGS);
#undef VB_ARGS
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
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
    if (g_prefill_attn16) {
      if (kv16)
        return launch_prefill_staged<1>(G, n_heads, T + (q0 & 15), 1, st, qbuf, kc, vc, T, T_cap, o_hi, o_lo, drop_plane, drop_bit, span_start, span_len, q0, wf, nullptr, 0, 0, 0);
      return launch_prefill_staged<0>(G, n_heads, T + (q0 & 15), 1, st, qbuf, kc, vc, T, T_cap, o_hi, o_lo, drop_plane, drop_bit, span_start, span_len, q0, wf, nullptr, 0, 0, 0);
    }
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
  if (g_prefill_attn16) {
    if (kv16)
      return launch_prefill_staged<1>(G, n_heads, max_T, n, st, qbuf, nullptr, nullptr, max_T, T_cap, o_hi, o_lo, nullptr, 0, 0, 0, 0, wf, tab, seq_rows, off_k, off_v);
    return launch_prefill_staged<0>(G, n_heads, max_T, n, st, qbuf, nullptr, nullptr, max_T, T_cap, o_hi, o_lo, nullptr, 0, 0, 0, 0, wf, tab, seq_rows, off_k, off_v);
  }
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


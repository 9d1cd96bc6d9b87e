// Dropout-specific kernels: uncertainty scorer, top-k ids, keep set, mask sampler, vote, RNG.
// gfx950 / wave64.  Reference anchors are quoted per entry point in include/dropdec.h.
#include <stdarg.h>

#include "dd_common.h"

// ----------------------------------------------------------------------------------------------
// error string (thread local)
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void dd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dd_last_error(void) { return g_err; }
extern "C" int dd_version(void) { return 100; }
extern "C" const char* dd_arch(void) { return "gfx950"; }

// ----------------------------------------------------------------------------------------------
// block-level helpers (blockDim.x multiple of 64, <= 1024)
// ----------------------------------------------------------------------------------------------
struct ArgMax {
  float v;
  int i;
};
// ordering used everywhere: larger value first, then lower index (torch.argmax: first maximal index)
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(a.v, o);
    int oi = __shfl_xor(a.i, o);
    if (better(ov, oi, a.v, a.i)) {
      a.v = ov;
      a.i = oi;
    }
  }
  return a;
}

__device__ ArgMax block_argmax(ArgMax a, ArgMax* sh /*>= 16 entries*/) {
  a = wave_argmax(a);
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = a;
  __syncthreads();
  ArgMax r = sh[0];
  for (int j = 1; j < nw; ++j)
    if (better(sh[j].v, sh[j].i, r.v, r.i)) r = sh[j];
  return r;
}

__device__ double block_sum_d(double v, double* sh /*>= 16*/) {
  v = dd_wave_sum_d(v);
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double r = 0;
  for (int j = 0; j < nw; ++j) r += sh[j];
  return r;
}

__device__ float block_max_f(float v, float* sh) {
  v = dd_wave_max(v);
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int j = 1; j < nw; ++j) r = fmaxf(r, sh[j]);
  return r;
}

// ----------------------------------------------------------------------------------------------
// Uncertainty scorer (calculate_vision_uncertainty, reference models/llava.py:710-756)
//   pass A: per row softmax statistics (max, sum exp) as 64-column block partials + a fixed-order combine   (llava.py:722) — the partials come
//           from the lm_head GEMM's epilogue in the engine's prefill (no read of the logits), from k_row_partials for the stand-alone call
//   pass B: partial column sums of p over row chunks, B2 combine -> p_avg (llava.py:732)
//   pass C: per row epi / alea / var + the top-k ids, the row held in LDS  (llava.py:728,735-739, 428-441)
// HBM-bound: the engine reads the [L][V] fp32 logits twice (B, C); the stand-alone C-ABI call three times (A, B, C).
// ----------------------------------------------------------------------------------------------
#define UNC_THREADS 512
#define UNC_LSPLIT 8

// Round 5: the row statistics as BLOCK PARTIALS.  {max, sum exp(x - max)} per row and 64-column block (dd_row_block_stats) — produced by the lm_head
// GEMM's epilogue while the logits are still in its accumulators (dd_prefill.hip, GemmArgs::rowstat: the north star's "softmax reduction fused into
// the unembedding product"), or, for the stand-alone C-ABI scorer, by k_row_partials from the stored logits with the same arithmetic — and one
// fixed-order combine per row.  The engine's prefill then reads the [L][V] logits twice (column mean; epi + top-k) instead of three times.
__global__ __launch_bounds__(256) void k_row_partials(const float* __restrict__ logits, int V, int ld, float* __restrict__ partials, int n_cb) {
  const float* x = logits + (size_t)blockIdx.x * ld;
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  for (int cb0 = 0; cb0 < n_cb; cb0 += 16) {              // (uniform trip count: every lane takes part in the shuffles)
    const int cb = cb0 + grp;
    float x4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = cb * 64 + 16 * j + c;
      x4[j] = (cb < n_cb && col < V) ? x[col] : -INFINITY;
    }
    float m, sum;
    dd_row_block_stats(x4, m, sum);
    if (c == 0 && cb < n_cb) {
      partials[((size_t)blockIdx.x * n_cb + cb) * 2] = m;
      partials[((size_t)blockIdx.x * n_cb + cb) * 2 + 1] = sum;
    }
  }
}
// one wave per row: M = max of the block maxima; S = sum over blocks of s_b * exp(m_b - M), each lane over its blocks in index order in fp64, then the
// wave's butterfly — a fixed order, so the same bits every run
__global__ __launch_bounds__(64) void k_row_stats_combine(const float* __restrict__ partials, int n_cb, float* __restrict__ row_max,
                                                          float* __restrict__ row_sum) {
  const float* p = partials + (size_t)blockIdx.x * n_cb * 2;
  float m = -INFINITY;
  for (int b = threadIdx.x; b < n_cb; b += 64) m = fmaxf(m, p[2 * b]);
  m = dd_wave_max(m);
  double s = 0;
  for (int b = threadIdx.x; b < n_cb; b += 64) {
    const float mb = p[2 * b];
    if (mb != -INFINITY) s += (double)(p[2 * b + 1] * expf(mb - m));
  }
  s = dd_wave_sum_d(s);
  if (threadIdx.x == 0) {
    row_max[blockIdx.x] = m;
    row_sum[blockIdx.x] = (float)s;
  }
}
// top-k ids of every row (value descending, index ascending), the row read from memory (rows too long for the LDS)
__global__ __launch_bounds__(UNC_THREADS) void k_row_topk(const float* __restrict__ logits, int V, int ld, int k_top, float* __restrict__ topk_vals,
                                                          int32_t* __restrict__ topk_ids) {
  __shared__ ArgMax sh_am[16];
  const float* x = logits + (size_t)blockIdx.x * ld;
  float pv = INFINITY;
  int pi = -1;
  for (int j = 0; j < k_top; ++j) {
    ArgMax a = {-INFINITY, 0x7fffffff};
    for (int v = threadIdx.x; v < V; v += UNC_THREADS) {
      float xv = x[v];
      bool after = (xv < pv) || (xv == pv && v > pi);
      if (after && better(xv, v, a.v, a.i)) {
        a.v = xv;
        a.i = v;
      }
    }
    a = block_argmax(a, sh_am);
    pv = a.v;
    pi = a.i;
    if (threadIdx.x == 0) {
      if (topk_vals) topk_vals[(size_t)blockIdx.x * k_top + j] = a.v;
      if (topk_ids) topk_ids[(size_t)blockIdx.x * k_top + j] = a.i;
    }
  }
}

// grid (ceil(V/256), UNC_LSPLIT): partial[ls][v] = sum over rows of chunk ls of p[l][v]
__global__ __launch_bounds__(256) void k_col_partial(const float* __restrict__ logits, int L, int V, int ld,
                                                     const float* __restrict__ row_max, const float* __restrict__ row_sum,
                                                     double* __restrict__ partial) {
  int v = blockIdx.x * 256 + threadIdx.x;
  int ls = blockIdx.y;
  int per = (L + UNC_LSPLIT - 1) / UNC_LSPLIT;
  int l0 = ls * per, l1 = min(L, l0 + per);
  if (v >= V) return;
  double acc = 0;
  for (int l = l0; l < l1; ++l) acc += (double)(expf(logits[(size_t)l * ld + v] - row_max[l]) / row_sum[l]);
  partial[(size_t)ls * V + v] = acc;
}

__global__ __launch_bounds__(256) void k_col_combine(const double* __restrict__ partial, int L, int V,
                                                     float* __restrict__ p_avg) {
  int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  double acc = 0;
  for (int ls = 0; ls < UNC_LSPLIT; ++ls) acc += partial[(size_t)ls * V + v];
  p_avg[v] = (float)(acc / (double)L);
}

__global__ __launch_bounds__(UNC_THREADS) void k_row_epi(const float* __restrict__ logits, int V, int ld,
                                                         const float* __restrict__ row_max,
                                                         const float* __restrict__ row_sum,
                                                         const float* __restrict__ p_avg, float* __restrict__ var_tok,
                                                         float* __restrict__ epi_tok, float* __restrict__ alea_tok) {
  __shared__ double sh_d[16];
  const float* x = logits + (size_t)blockIdx.x * ld;
  float m = row_max[blockIdx.x], s = row_sum[blockIdx.x];
  double epi = 0, alea = 0, sp = 0, sp2 = 0;
  for (int v = threadIdx.x; v < V; v += UNC_THREADS) {
    float p = expf(x[v] - m) / s;
    float lp = logf(p + 1e-10f);
    float lpa = logf(p_avg[v] + 1e-10f);
    epi += (double)(p * (lp - lpa));
    alea += (double)(p * lp);
    sp += (double)p;
    sp2 += (double)p * (double)p;
  }
  epi = block_sum_d(epi, sh_d);
  alea = block_sum_d(alea, sh_d);
  sp = block_sum_d(sp, sh_d);
  sp2 = block_sum_d(sp2, sh_d);
  if (threadIdx.x == 0) {
    epi_tok[blockIdx.x] = (float)epi;
    alea_tok[blockIdx.x] = (float)(-alea);
    double mean = sp / V;
    var_tok[blockIdx.x] = (float)((sp2 - (double)V * mean * mean) / (double)(V - 1));  // unbiased (torch.var)
  }
}

// pass C with the row held in LDS: epi / alea / var AND the k_top selection sweeps from one read of the row (the arithmetic of k_row_epi and of
// k_row_topk, same order: same results)
__global__ __launch_bounds__(UNC_THREADS) void k_row_epi_topk_lds(const float* __restrict__ logits, int V, int ld, const float* __restrict__ row_max,
                                                                  const float* __restrict__ row_sum, const float* __restrict__ p_avg,
                                                                  float* __restrict__ var_tok, float* __restrict__ epi_tok, float* __restrict__ alea_tok,
                                                                  int k_top, float* __restrict__ topk_vals, int32_t* __restrict__ topk_ids) {
  extern __shared__ __align__(16) float row_sh[];
  __shared__ ArgMax sh_am[16];
  __shared__ double sh_d[16];
  const float* xg = logits + (size_t)blockIdx.x * ld;
  const int V4 = V >> 2;
  for (int i = threadIdx.x; i < V4; i += UNC_THREADS) *(f32x4_t*)&row_sh[4 * i] = *(const f32x4_t*)(xg + 4 * i);
  for (int v = 4 * V4 + threadIdx.x; v < V; v += UNC_THREADS) row_sh[v] = xg[v];
  __syncthreads();
  const float* x = row_sh;
  const float m = row_max[blockIdx.x], s = row_sum[blockIdx.x];
  double epi = 0, alea = 0, sp = 0, sp2 = 0;
  for (int v = threadIdx.x; v < V; v += UNC_THREADS) {
    float p = expf(x[v] - m) / s;
    float lp = logf(p + 1e-10f);
    float lpa = logf(p_avg[v] + 1e-10f);
    epi += (double)(p * (lp - lpa));
    alea += (double)(p * lp);
    sp += (double)p;
    sp2 += (double)p * (double)p;
  }
  epi = block_sum_d(epi, sh_d);
  alea = block_sum_d(alea, sh_d);
  sp = block_sum_d(sp, sh_d);
  sp2 = block_sum_d(sp2, sh_d);
  if (threadIdx.x == 0) {
    epi_tok[blockIdx.x] = (float)epi;
    alea_tok[blockIdx.x] = (float)(-alea);
    double mean = sp / V;
    var_tok[blockIdx.x] = (float)((sp2 - (double)V * mean * mean) / (double)(V - 1));  // unbiased (torch.var)
  }
  float pv = INFINITY;
  int pi = -1;
  for (int j = 0; j < k_top; ++j) {
    ArgMax a = {-INFINITY, 0x7fffffff};
    for (int v = threadIdx.x; v < V; v += UNC_THREADS) {
      float xv = x[v];
      bool after = (xv < pv) || (xv == pv && v > pi);
      if (after && better(xv, v, a.v, a.i)) {
        a.v = xv;
        a.i = v;
      }
    }
    a = block_argmax(a, sh_am);
    pv = a.v;
    pi = a.i;
    if (threadIdx.x == 0) {
      if (topk_vals) topk_vals[(size_t)blockIdx.x * k_top + j] = a.v;
      if (topk_ids) topk_ids[(size_t)blockIdx.x * k_top + j] = a.i;
    }
  }
}

__global__ __launch_bounds__(256) void k_means3(const float* a, const float* b, const float* c, int L, float* out3) {
  __shared__ double sh_d[16];
  double sa = 0, sb = 0, sc = 0;
  for (int l = threadIdx.x; l < L; l += 256) {
    sa += a[l];
    sb += b[l];
    sc += c[l];
  }
  sa = block_sum_d(sa, sh_d);
  sb = block_sum_d(sb, sh_d);
  sc = block_sum_d(sc, sh_d);
  if (threadIdx.x == 0) {
    out3[0] = (float)(sa / L);
    out3[1] = (float)(sb / L);
    out3[2] = (float)(sc / L);
  }
}

static int unc_n_cb(int V) { return (V + 63) / 64; }
extern "C" size_t dd_uncertainty_workspace_bytes(int L, int V) {
  // row_max[L] row_sum[L] p_avg[V] (fp32) + partial[UNC_LSPLIT][V] (fp64) + block partials [L][ceil(V/64)][2] (fp32), 256-byte aligned pieces
  size_t a = ((size_t)L * 4 + 255) / 256 * 256;
  size_t p = ((size_t)V * 4 + 255) / 256 * 256;
  size_t bp = ((size_t)L * unc_n_cb(V) * 8 + 255) / 256 * 256;
  return 2 * a + p + (size_t)UNC_LSPLIT * V * 8 + 256 + bp;
}
// where the block partials of a workspace for (L_cap rows, V) live — the engine hands this to the lm_head GEMM (GemmArgs::rowstat)
float* dd_uncertainty_partials(void* ws, int L_cap, int V, int* n_cb) {
  size_t a = ((size_t)L_cap * 4 + 255) / 256 * 256, p = ((size_t)V * 4 + 255) / 256 * 256;
  *n_cb = unc_n_cb(V);
  return (float*)((char*)ws + 2 * a + p + (size_t)UNC_LSPLIT * V * 8 + 256);
}

// partials_ready: the block partials already sit in the workspace laid out for L_cap rows (written by the lm_head GEMM's epilogue); null: this
// call computes them from the logits (k_row_partials: the same arithmetic, the same bits).
int dd_vision_uncertainty_impl(const float* logits, int L, int V, int ld, float* var_tok, float* epi_tok, float* alea_tok, float* scalars3,
                               int k_top, float* topk_vals, int32_t* topk_ids, void* ws, size_t ws_bytes, hipStream_t st, int L_cap,
                               bool partials_ready) {
  DD_REQUIRE(logits && var_tok && epi_tok && alea_tok && ws, "dd_vision_uncertainty: null pointer");
  DD_REQUIRE(L >= 1 && V >= 2 && ld >= V && L_cap >= L, "dd_vision_uncertainty: bad shape L=%d V=%d ld=%d", L, V, ld);
  DD_REQUIRE(k_top >= 0 && k_top <= DD_MAX_TOPK && k_top <= V, "dd_vision_uncertainty: k_top=%d out of range", k_top);
  DD_REQUIRE(ws_bytes >= dd_uncertainty_workspace_bytes(L_cap, V), "dd_vision_uncertainty: workspace too small");
  size_t a = ((size_t)L_cap * 4 + 255) / 256 * 256, p = ((size_t)V * 4 + 255) / 256 * 256;
  char* base = (char*)ws;
  float* row_max = (float*)base;
  float* row_sum = (float*)(base + a);
  float* p_avg = (float*)(base + 2 * a);
  double* partial = (double*)(base + 2 * a + p);
  int n_cb = 0;
  float* bpart = dd_uncertainty_partials(ws, L_cap, V, &n_cb);
  if (!partials_ready) {
    k_row_partials<<<L, 256, 0, st>>>(logits, V, ld, bpart, n_cb);
    DD_CHECK_LAUNCH();
  }
  k_row_stats_combine<<<L, 64, 0, st>>>(bpart, n_cb, row_max, row_sum);
  DD_CHECK_LAUNCH();
  k_col_partial<<<dim3((V + 255) / 256, UNC_LSPLIT), 256, 0, st>>>(logits, L, V, ld, row_max, row_sum, partial);
  DD_CHECK_LAUNCH();
  k_col_combine<<<(V + 255) / 256, 256, 0, st>>>(partial, L, V, p_avg);
  DD_CHECK_LAUNCH();
  const size_t row_bytes = (size_t)V * sizeof(float);
  if (row_bytes <= 150 * 1024 && (ld & 3) == 0 && ((uintptr_t)logits & 15) == 0) {
    static bool attr = false;
    if (!attr) {
      DD_HIP(hipFuncSetAttribute((const void*)k_row_epi_topk_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr = true;
    }
    k_row_epi_topk_lds<<<L, UNC_THREADS, row_bytes, st>>>(logits, V, ld, row_max, row_sum, p_avg, var_tok, epi_tok, alea_tok, k_top, topk_vals, topk_ids);
    DD_CHECK_LAUNCH();
  } else {
    k_row_epi<<<L, UNC_THREADS, 0, st>>>(logits, V, ld, row_max, row_sum, p_avg, var_tok, epi_tok, alea_tok);
    DD_CHECK_LAUNCH();
    if (k_top > 0) {
      k_row_topk<<<L, UNC_THREADS, 0, st>>>(logits, V, ld, k_top, topk_vals, topk_ids);
      DD_CHECK_LAUNCH();
    }
  }
  if (scalars3) {
    k_means3<<<1, 256, 0, st>>>(var_tok, epi_tok, alea_tok, L, scalars3);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}
extern "C" int dd_vision_uncertainty(const float* logits, int L, int V, int ld, float* var_tok, float* epi_tok,
                                     float* alea_tok, float* scalars3, int k_top, float* topk_vals,
                                     int32_t* topk_ids, void* ws, size_t ws_bytes, void* stream_) {
  return dd_vision_uncertainty_impl(logits, L, V, ld, var_tok, epi_tok, alea_tok, scalars3, k_top, topk_vals, topk_ids, ws, ws_bytes,
                                    (hipStream_t)stream_, L, false);
}

// ----------------------------------------------------------------------------------------------
// argmax over rows; keep set (get_overlap_image_tokens, reference models/llava.py:443-482)
// ----------------------------------------------------------------------------------------------
// `gate` (engine-internal launches): non-null and non-zero = the sequence has emitted its EOS; the wasted look-ahead step
// leaves every persistent buffer as the EOS step left it (dd_lm_kernels.h DDState::done)
// A thread's share of a row (1,024 threads).  `better` is a total order (value, then lower index), so the result does not depend on the order of the
// scan: 16-byte loads, four in flight per thread (round 6: the one-float-per-iteration scan took 12-18 us for a 128-KB row — 31 dependent round trips).
__device__ __forceinline__ ArgMax argmax_scan_row(const float* __restrict__ r, int V) {
  ArgMax a = {-INFINITY, 0x7fffffff};
  auto upd = [&](float xv, int v) {
    if (better(xv, v, a.v, a.i)) {
      a.v = xv;
      a.i = v;
    }
  };
  const int tid = threadIdx.x;
  int v0 = 0;
  if ((((size_t)r) & 15) == 0) {
    const int V4 = V >> 2;
    const f32x4_t* r4 = (const f32x4_t*)r;
    int q = tid;
    for (; q + 3 * 1024 < V4; q += 4 * 1024) {
      f32x4_t x4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x4[u] = r4[q + u * 1024];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) upd(x4[u][j], 4 * (q + u * 1024) + j);
    }
    for (; q < V4; q += 1024) {
      const f32x4_t x4 = r4[q];
#pragma unroll
      for (int j = 0; j < 4; ++j) upd(x4[j], 4 * q + j);
    }
    v0 = 4 * V4;
  }
  for (int v = v0 + tid; v < V; v += 1024) upd(r[v], v);
  return a;
}
__global__ __launch_bounds__(1024) void k_argmax_rows(const float* __restrict__ x, int V, int ld, int32_t* out,
                                                      const int32_t* __restrict__ gate) {
  __shared__ ArgMax sh[16];
  if (gate && *gate) return;
  ArgMax a = argmax_scan_row(x + (size_t)blockIdx.x * ld, V);
  a = block_argmax(a, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = a.i;
}

// rows of up to 4 sequences (their K member logits) in one launch: grid (R, n)
struct ArgmaxLanes {
  const float* x[8];
  int32_t* out[8];
  const int32_t* gate[8];
};
__global__ __launch_bounds__(1024) void k_argmax_rows_lanes(ArgmaxLanes t, int V, int ld) {
  __shared__ ArgMax sh[16];
  if (t.gate[blockIdx.y] && *t.gate[blockIdx.y]) return;
  ArgMax a = argmax_scan_row(t.x[blockIdx.y] + (size_t)blockIdx.x * ld, V);
  a = block_argmax(a, sh);
  if (threadIdx.x == 0) t.out[blockIdx.y][blockIdx.x] = a.i;
}
int dd_argmax_rows_lanes(const float* const* x, int32_t* const* out, const int32_t* const* gates, int n, int R, int V, int ld,
                         hipStream_t st) {
  DD_REQUIRE(x && out && n >= 1 && n <= 8 && R >= 1, "dd_argmax_rows_lanes: bad arguments");
  ArgmaxLanes t;
  memset(&t, 0, sizeof(t));
  for (int i = 0; i < n; ++i) t.x[i] = x[i], t.out[i] = out[i], t.gate[i] = gates ? gates[i] : nullptr;
  k_argmax_rows_lanes<<<dim3(R, n), 1024, 0, st>>>(t, V, ld);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

extern "C" int dd_argmax_rows(const float* x, int R, int V, int ld, int32_t* out, void* stream_) {
  DD_REQUIRE(x && out && R >= 1 && V >= 1 && ld >= V, "dd_argmax_rows: bad arguments");
  k_argmax_rows<<<R, 1024, 0, (hipStream_t)stream_>>>(x, V, ld, out, nullptr);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int dd_argmax_rows_gated(const float* x, int R, int V, int ld, int32_t* out, const int32_t* gate, hipStream_t st) {
  k_argmax_rows<<<R, 1024, 0, st>>>(x, V, ld, out, gate);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

__global__ __launch_bounds__(1024) void k_overlap_keep(const float* __restrict__ x, int V,
                                                       const int32_t* __restrict__ topk, int L, int k,
                                                       uint8_t* __restrict__ keep, int32_t* __restrict__ argmax_out,
                                                       const int32_t* __restrict__ argmax_in,
                                                       const int32_t* __restrict__ gate) {
  __shared__ ArgMax sh[16];
  if (gate && *gate) return;
  int tok;
  if (argmax_in) {
    tok = argmax_in[0];
  } else {
    ArgMax a = {-INFINITY, 0x7fffffff};
    for (int v = threadIdx.x; v < V; v += 1024) {
      float xv = x[v];
      if (better(xv, v, a.v, a.i)) {
        a.v = xv;
        a.i = v;
      }
    }
    a = block_argmax(a, sh);
    tok = a.i;
  }
  if (threadIdx.x == 0 && argmax_out) argmax_out[0] = tok;
  for (int l = threadIdx.x; l < L; l += 1024) {
    bool hit = false;
    for (int j = 0; j < k; ++j) hit |= (topk[(size_t)l * k + j] == tok);
    keep[l] = hit ? 1 : 0;
  }
}

extern "C" int dd_overlap_keep(const float* step_logits, int V, const int32_t* topk_ids, int L, int k, uint8_t* keep,
                               int32_t* argmax_out, void* stream_) {
  DD_REQUIRE(step_logits && topk_ids && keep && V >= 1 && L >= 1 && k >= 1, "dd_overlap_keep: bad arguments");
  k_overlap_keep<<<1, 1024, 0, (hipStream_t)stream_>>>(step_logits, V, topk_ids, L, k, keep, argmax_out, nullptr, nullptr);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// internal variant used by the engine: argmax already on the device
int dd_overlap_keep_from_argmax(const int32_t* argmax_dev, const int32_t* topk_ids, int L, int k, uint8_t* keep,
                                const int32_t* gate, hipStream_t st) {
  k_overlap_keep<<<1, 1024, 0, st>>>(nullptr, 0, topk_ids, L, k, keep, nullptr, argmax_dev, gate);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// ----------------------------------------------------------------------------------------------
// "epis_kl" keep set (lowest_percent_kl_indices, reference models/instructblip.py:559-578; used at :483-485):
//   kl[l] = sum_v p_step[v] * (log p_step[v] - log_softmax(image_logits[l])[v])      (F.kl_div(input = log-softmax of the
//   token's prefill logits, target = softmax of the step's un-masked logits), summed over the vocabulary)
//   keep = the int(0.1 * L) tokens with the smallest kl (value ascending, then index)
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_kl_rows(const float* __restrict__ step, const float* __restrict__ img, int V, int ld,
                                                 float* __restrict__ kl_out, const int32_t* __restrict__ gate) {
  __shared__ double sh_d[16];
  __shared__ float sh_f[16];
  if (gate && *gate) return;
  const float* x = img + (size_t)blockIdx.x * ld;
  float ms = -INFINITY, mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 512) {
    ms = fmaxf(ms, step[v]);
    mx = fmaxf(mx, x[v]);
  }
  ms = block_max_f(ms, sh_f);
  mx = block_max_f(mx, sh_f);
  double ss = 0, sx = 0;
  for (int v = threadIdx.x; v < V; v += 512) {
    ss += (double)expf(step[v] - ms);
    sx += (double)expf(x[v] - mx);
  }
  ss = block_sum_d(ss, sh_d);
  sx = block_sum_d(sx, sh_d);
  const float ls = ms + logf((float)ss), lx = mx + logf((float)sx);     // log-sum-exp of either row
  double acc = 0;
  for (int v = threadIdx.x; v < V; v += 512) {
    float lp = step[v] - ls;                                            // log p_step
    float p = expf(lp);
    if (p > 0.f) acc += (double)(p * (lp - (x[v] - lx)));               // xlogy: a zero target contributes 0
  }
  acc = block_sum_d(acc, sh_d);
  if (threadIdx.x == 0) kl_out[blockIdx.x] = (float)acc;
}
// one workgroup: keep[l] = 1 for the n_low smallest kl (rank by (value, index)); L <= 8192
__global__ __launch_bounds__(1024) void k_kl_select(const float* __restrict__ kl, int L, int n_low, uint8_t* __restrict__ keep,
                                                    const int32_t* __restrict__ gate) {
  if (gate && *gate) return;
  for (int l = threadIdx.x; l < L; l += 1024) {
    const float v = kl[l];
    int rank = 0;
    for (int j = 0; j < L; ++j) {
      const float w = kl[j];
      rank += (w < v || (w == v && j < l)) ? 1 : 0;
    }
    keep[l] = rank < n_low ? 1 : 0;
  }
}
int dd_kl_keep_impl(const float* step_logits, const float* image_logits, int L, int V, int ld, float percent, uint8_t* keep,
                    float* kl_ws, const int32_t* gate, hipStream_t st) {
  DD_REQUIRE(step_logits && image_logits && keep && kl_ws && L >= 1 && L <= 8192 && V >= 2 && ld >= V, "dd_kl_keep: bad arguments");
  k_kl_rows<<<L, 512, 0, st>>>(step_logits, image_logits, V, ld, kl_ws, gate);
  DD_CHECK_LAUNCH();
  k_kl_select<<<1, 1024, 0, st>>>(kl_ws, L, (int)(percent * (float)L), keep, gate);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
extern "C" int dd_kl_keep(const float* step_logits_dev, const float* image_logits_dev, int L, int V, int ld, uint8_t* keep_dev,
                          float* kl_dev, void* stream) {
  return dd_kl_keep_impl(step_logits_dev, image_logits_dev, L, V, ld, 0.1f, keep_dev, kl_dev, nullptr, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------------
// Device-resident generators. mt19937 (torch CPU default generator): 624 words + read index.
// Philox4x32-10 (torch GPU default generator): the same 625-word block with word 624 = PHILOX_TAG and
// words 0..3 = seed lo/hi, offset lo/hi — so every consumer (the mask sampler, the speculative step's
// backup copy, the captured graphs) carries either kind through one pointer.
// ----------------------------------------------------------------------------------------------
struct dd_rng {
  uint32_t* state;  // device, 625 words
  unsigned long long serial;   // never reused: captured decode steps are keyed on it, not on the address
  int kind;                    // 0 mt19937, 1 Philox
};
static unsigned long long g_rng_serial = 0;

#define MT_N 624
#define MT_M 397

__global__ void k_mt_seed(uint32_t* st, uint32_t seed) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    uint32_t x = seed;
    st[0] = x;
    for (int j = 1; j < MT_N; ++j) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)j;
      st[j] = x;
    }
    st[MT_N] = MT_N;
  }
}

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// ----------------------------------------------------------------------------------------------
// ONE WAVE per generator (round 5).  Until round 5 every kernel that touches a generator block was a 1,024-thread workgroup whose phases were
// fenced by s_barrier; that form is the victim of the co-residency fault of DESIGN.md 3e (its waves fall out of step across s_barrier while the
// kernels of a rider step share the CU; csrc/dd_sampler_block.h keeps it for the reproducer).  A single wavefront needs no barrier: its LDS
// operations execute in program order, so "all reads of this chunk before its writes, this chunk before the next" is the order of the instructions.
// DD_WSYNC only stops the COMPILER from moving LDS accesses across a phase boundary (other lanes' data); it emits no s_barrier.
// ----------------------------------------------------------------------------------------------
#define DD_WSYNC()                                            \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)

// Regenerate the 624 words held in LDS, one wave, in index order: new[i] = f(cur[i], cur[i+1], cur[i+397 mod 624]) reads words the sequential
// recurrence has not reached yet (i+1: read before this round's writes; i+397 < 624: a later chunk) or has already replaced (i+397-624 = i-227
// and, for i = 623, word 0: earlier rounds) — the sequential algorithm's values exactly.
// Three chunks (192 words) per round: a chunk's inputs are old words of its own and later chunks, and new words at least 227 - 63 = 164 places back,
// i.e. of chunks at least two before it — so chunks c, c + 1, c + 2 read everything first and then write, four dependent rounds instead of ten.
__device__ __forceinline__ void mt_twist_wave(uint32_t* mt) {
  const int lane = threadIdx.x;
#pragma unroll 1
  for (int b0 = 0; b0 < MT_N; b0 += 192) {
    uint32_t nv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = b0 + 64 * j + lane;
      nv[j] = 0;
      if (i < MT_N) nv[j] = mt_mix(mt[i], mt[i + 1 == MT_N ? 0 : i + 1], mt[i + MT_M >= MT_N ? i + MT_M - MT_N : i + MT_M]);
    }
    DD_WSYNC();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = b0 + 64 * j + lane;
      if (i < MT_N) mt[i] = nv[j];
    }
    DD_WSYNC();
  }
}
__device__ __forceinline__ float mt_temper_uniform(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return (float)(y & 0xFFFFFFu) * 5.9604644775390625e-08f;  // * 2^-24, exact
}

// Philox4x32-10 (Salmon et al., SC'11), first output word for counter (ctr, subseq) and key (k0, k1): the word ATen's
// random kernel hands element `subseq` of a rand_like over <= 524288 elements (counter layout of cuRAND / rocRAND).
#define PHILOX_TAG 0xFFFFFFFFu
__device__ __forceinline__ uint32_t philox_first(uint32_t k0, uint32_t k1, unsigned long long ctr, unsigned long long subseq) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = (uint32_t)subseq, c3 = (uint32_t)(subseq >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1, c3 = (uint32_t)p0, c0 = n0, c2 = n2;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  return c0;
}
// u32 -> (0,1] as rocRAND does it (one fused multiply-add), then ATen folds 1.0 back to 0.0
__device__ __forceinline__ float philox_uniform(uint32_t x) {
  float u = __builtin_fmaf((float)x, 2.3283064e-10f, 2.3283064e-10f);
  return u == 1.0f ? 0.0f : u;
}

__global__ void k_philox_seed(uint32_t* st, unsigned long long seed, unsigned long long offset) {
  for (int i = threadIdx.x; i < MT_N; i += blockDim.x) st[i] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) {
    st[0] = (uint32_t)seed, st[1] = (uint32_t)(seed >> 32), st[2] = (uint32_t)offset, st[3] = (uint32_t)(offset >> 32);
    st[MT_N] = PHILOX_TAG;
  }
}

// Fill out[0..n) (LDS or global) with the next n uniforms of the generator block in LDS (one wave); idx: the read index (wave-uniform register).
__device__ __forceinline__ void mt_fill_wave(uint32_t* mt, int& idx, float* out, int n) {
  const int lane = threadIdx.x;
  DD_WSYNC();
  if ((uint32_t)idx == PHILOX_TAG) {   // one rand_like: element t = subsequence t at the current offset; offset += 4
    const unsigned long long off = (unsigned long long)mt[2] | ((unsigned long long)mt[3] << 32);
    const uint32_t k0 = mt[0], k1 = mt[1];
    for (int t = lane; t < n; t += 64) out[t] = philox_uniform(philox_first(k0, k1, off >> 2, (unsigned long long)t));
    DD_WSYNC();
    if (lane == 0) {
      const unsigned long long o2 = off + 4ull;
      mt[2] = (uint32_t)o2, mt[3] = (uint32_t)(o2 >> 32);
    }
    DD_WSYNC();
    return;
  }
  int pos = 0;
  while (pos < n) {
    if (idx >= MT_N) {
      mt_twist_wave(mt);
      idx = 0;
    }
    const int take = min(n - pos, MT_N - idx);
    for (int t = lane; t < take; t += 64) out[pos + t] = mt_temper_uniform(mt[idx + t]);
    idx += take;
    pos += take;
  }
  DD_WSYNC();
}

// dd_rng_uniform, mt19937: one wave walks the stream (a draw of n costs n / 624 regenerations of ~1 us)
__global__ __launch_bounds__(64) void k_mt_uniform(uint32_t* st, float* out, int n) {
  __shared__ uint32_t mt[MT_N + 8];
  for (int i = threadIdx.x; i < MT_N; i += 64) mt[i] = st[i];
  int idx = (int)st[MT_N];
  mt_fill_wave(mt, idx, out, n);
  for (int i = threadIdx.x; i < MT_N; i += 64) st[i] = mt[i];
  if (threadIdx.x == 0) st[MT_N] = (uint32_t)idx;
}
// dd_rng_uniform, Philox: every element is its own subsequence — a plain grid, no LDS; the offset advances in k_philox_advance
__global__ __launch_bounds__(256) void k_philox_uniform(const uint32_t* __restrict__ st, float* __restrict__ out, int n) {
  const unsigned long long off = (unsigned long long)st[2] | ((unsigned long long)st[3] << 32);
  const uint32_t k0 = st[0], k1 = st[1];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n) out[t] = philox_uniform(philox_first(k0, k1, off >> 2, (unsigned long long)t));
}
__global__ void k_philox_advance(uint32_t* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned long long o2 = ((unsigned long long)st[2] | ((unsigned long long)st[3] << 32)) + 4ull;
    st[2] = (uint32_t)o2, st[3] = (uint32_t)(o2 >> 32);
  }
}

extern "C" int dd_rng_create(uint32_t seed, dd_rng** out) {
  DD_REQUIRE(out, "dd_rng_create: null out");
  dd_rng* r = new dd_rng();
  r->serial = ++g_rng_serial;
  r->kind = 0;
  hipError_t e = hipMalloc((void**)&r->state, (MT_N + 1) * sizeof(uint32_t));
  if (e != hipSuccess) {
    delete r;
    dd_set_error("dd_rng_create: hipMalloc -> %s", hipGetErrorString(e));
    return DD_ENOMEM;
  }
  k_mt_seed<<<1, 64>>>(r->state, seed);
  DD_CHECK_LAUNCH();
  DD_HIP(hipDeviceSynchronize());
  *out = r;
  return DD_OK;
}
extern "C" int dd_rng_create_philox(unsigned long long seed, unsigned long long offset, dd_rng** out) {
  DD_REQUIRE(out, "dd_rng_create_philox: null out");
  DD_REQUIRE((offset & 3ull) == 0, "dd_rng_create_philox: offset %llu is not a multiple of 4 (torch advances it in fours)", offset);
  dd_rng* r = new dd_rng();
  r->serial = ++g_rng_serial;
  r->kind = 1;
  hipError_t e = hipMalloc((void**)&r->state, (MT_N + 1) * sizeof(uint32_t));
  if (e != hipSuccess) {
    delete r;
    dd_set_error("dd_rng_create_philox: hipMalloc -> %s", hipGetErrorString(e));
    return DD_ENOMEM;
  }
  k_philox_seed<<<1, 64>>>(r->state, seed, offset);
  DD_CHECK_LAUNCH();
  DD_HIP(hipDeviceSynchronize());
  *out = r;
  return DD_OK;
}
extern "C" int dd_rng_destroy(dd_rng* r) {
  if (!r) return DD_OK;
  (void)hipFree(r->state);
  delete r;
  return DD_OK;
}
extern "C" int dd_rng_seed(dd_rng* r, uint32_t seed, void* stream_) {
  DD_REQUIRE(r, "dd_rng_seed: null rng");
  if (r->kind == 1)
    k_philox_seed<<<1, 64, 0, (hipStream_t)stream_>>>(r->state, (unsigned long long)seed, 0ull);   // torch.manual_seed: offset back to 0
  else
    k_mt_seed<<<1, 64, 0, (hipStream_t)stream_>>>(r->state, seed);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
extern "C" int dd_rng_uniform(dd_rng* r, float* out, int n, void* stream_) {
  DD_REQUIRE(r && out && n >= 0, "dd_rng_uniform: bad arguments");
  if (n == 0) return DD_OK;
  DD_REQUIRE(r->kind == 0 || n <= 524288, "dd_rng_uniform: a Philox draw covers at most 524288 elements per call (got %d)", n);
  if (r->kind == 1) {
    k_philox_uniform<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_>>>(r->state, out, n);
    k_philox_advance<<<1, 64, 0, (hipStream_t)stream_>>>(r->state);
  } else {
    k_mt_uniform<<<1, 64, 0, (hipStream_t)stream_>>>(r->state, out, n);
  }
  DD_CHECK_LAUNCH();
  return DD_OK;
}
uint32_t* dd_rng_state_ptr(dd_rng* r) { return r ? r->state : nullptr; }
unsigned long long dd_rng_serial(dd_rng* r) { return r ? r->serial : 0ull; }

// ----------------------------------------------------------------------------------------------
// Mask sampler: one WAVE, all K members of a step (get_image_attention_mask "epis")
// ----------------------------------------------------------------------------------------------
#define MASK_MAX_L 8192

struct MaskParams {
  const float* epi;
  int L, K, mode, rng_mode;
  const uint8_t* keep;
  const float* uniforms;  // [K][L] (injected) or nullptr
  uint32_t* rng_state;    // mt19937 state or nullptr
  uint8_t* drop;          // [K][L]
  int32_t* n_drop;        // [K]
  int32_t* idx;           // [K][L] or nullptr
  uint8_t* drop_bits;     // optional [ceil(K/8)][L]: bit (k&7) of plane k>>3 = member k dropped (engine layout)
  const int32_t* gate;    // optional: *gate != 0 -> the whole launch is a no-op (no draws: the rng stream stays put)
  const uint32_t* rng_in; // optional: read the mt19937 state from here instead of rng_state and do NOT write it back (a re-run
                          // of the draws a speculative launch already made: the stream has advanced by exactly these draws)
  float scale[64];        // f32(mprob - 0.1)   (reference llava.py:646: python double, rounded when it meets fp32)
  float q[64];            // f32(1 - mprob)     (reference instructblip.py:450); directly behind scale[]: the kernels stage both as one table
};
static_assert(offsetof(MaskParams, q) == offsetof(MaskParams, scale) + 64 * sizeof(float), "scale[] and q[] must be adjacent");

// The per-sequence operands of one workgroup's sampling, in registers.
struct MaskSeq {
  const float* epi;
  int L;
  const uint8_t* keep;
  const float* uniforms;
  uint32_t* rng_state;
  uint8_t* drop;
  int32_t* n_drop;
  int32_t* idx;
  uint8_t* drop_bits;
  const uint32_t* rng_in;
};
// C: the launch-wide constants; the per-member tables scale[] / q[] are staged in LDS by the kernel (mask_const_stage).  (A by-value
// MaskParams copy indexed by the member loop lives in private scratch — 616 bytes per lane, the only kernel of the library that had
// any; see DESIGN.md "determinism" and tests/test_gpu_sampler_repro.py.)
struct MaskConst {
  int K, mode, rng_mode;
  const float* scale;   // [K] f32(mprob - 0.1), in LDS
  const float* q;       // [K] f32(1 - mprob), in LDS
};
// One wave samples all K members of one sequence.  Dynamic LDS (smem, Lp = L rounded up to a power of two >= 64):
//   e[Lp] f32 | u[Lp] f32 (the current member's uniforms, or the sort buffer) | running[Lp] u8 | keep[Lp] u8 | bits[Lp] u8 | mt[632] u32
// (bits: the bit plane of the current eight members, built in LDS and stored once per plane instead of a global read-modify-write per member and
// position.  Measured, L = 576, K = 8: 49 -> 45 us per launch; the kernel is bound by what ONE wave can issue)
// keep_lds: the keep flags are already in LDS (the lanes kernel computes them there); else they are copied from P.keep (null: empty set).
__device__ __forceinline__ size_t sampler_lp(int L) {
  int Lp = 64;
  while (Lp < L) Lp <<= 1;
  return (size_t)Lp;
}
__device__ __forceinline__ void sample_masks_wave(const MaskConst C, const MaskSeq P, unsigned char* smem, const bool keep_lds) {
  const int L = P.L, lane = threadIdx.x;
  const int Lp = (int)sampler_lp(L);
  float* e = (float*)smem;                          // [Lp]
  float* u = e + Lp;                                // [Lp]
  uint8_t* running = (uint8_t*)(u + Lp);            // [Lp]
  uint8_t* keep = running + Lp;                     // [Lp]
  uint8_t* bits = keep + Lp;                        // [Lp]
  uint32_t* mt = (uint32_t*)(bits + Lp);            // [MT_N + 8]
  const bool no_overlap = C.mode == DD_MASK_NEXT_NO_OVERLAP || C.mode == DD_MASK_LLAVA_CUMULATIVE_NO_OVERLAP;

  for (int l = lane; l < L; l += 64) {
    e[l] = P.epi[l];
    running[l] = 0;
    bits[l] = 0;
    if (!keep_lds) keep[l] = (P.keep && !no_overlap) ? P.keep[l] : (uint8_t)0;
  }
  int idx = 0;
  if (C.rng_mode == DD_RNG_MT19937) {
    const uint32_t* src = P.rng_in ? P.rng_in : P.rng_state;
    for (int i = lane; i < MT_N; i += 64) mt[i] = src[i];
    idx = (int)src[MT_N];
  }
  DD_WSYNC();
  float lo = 0.f, hi = 0.f;
  if (C.mode != DD_MASK_IBLIP_QUANTILE) {
    float vmin = INFINITY, vmax = -INFINITY;
    for (int l = lane; l < L; l += 64) {
      vmin = fminf(vmin, e[l]);                      // torch.quantile(e, 0) == min   (llava.py:641)
      vmax = fmaxf(vmax, e[l]);                      // torch.quantile(e, 1) == max   (llava.py:642)
    }
    lo = -dd_wave_max(-vmin);
    hi = dd_wave_max(vmax);
  } else {
    // ascending bitonic sort of e into u (padded with +inf): torch.quantile sorts first.  One wave: a pass's pairs are disjoint, passes follow
    // each other in program order.
    for (int l = lane; l < Lp; l += 64) u[l] = l < L ? e[l] : INFINITY;
    DD_WSYNC();
    for (int k2 = 2; k2 <= Lp; k2 <<= 1) {
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < Lp; i += 64) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const float a = u[i], b = u[ixj];
            const bool up = ((i & k2) == 0);
            if ((a > b) == up) {
              u[i] = b;
              u[ixj] = a;
            }
          }
        }
        DD_WSYNC();
      }
    }
  }

  for (int k = 0; k < C.K; ++k) {   // DD_MASK_IBLIP_KL runs the NEXT_RESET rule: its keep flags come from dd_kl_keep instead of the overlap
    if (C.mode != DD_MASK_LLAVA_CUMULATIVE && C.mode != DD_MASK_LLAVA_CUMULATIVE_NO_OVERLAP) {  // reset: llavanext.py:546, instructblip.py:121
      for (int l = lane; l < L; l += 64) running[l] = 0;
    }
    float thr = 0.f;
    if (C.mode == DD_MASK_IBLIP_QUANTILE) {
      // torch.quantile(e, q) with linear interpolation, fp32: rank = q*(n-1); lerp(sorted[floor], sorted[ceil], frac)
      // ATen's lerp: weight < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w), contracted to one fma on the CPU build.
      const float rank = C.q[k] * (float)(L - 1);
      const float fl = floorf(rank);
      const int i0 = (int)fl, i1 = (int)ceilf(rank);
      const float w = rank - fl;
      const float a = u[i0], b = u[i1], diff = b - a;
      thr = (w < 0.5f) ? fmaf(w, diff, a) : fmaf(-diff, 1.0f - w, b);
    } else if (C.rng_mode == DD_RNG_MT19937) {
      mt_fill_wave(mt, idx, u, L);  // one rand_like(e) per member (llava.py:650)
    }
    DD_WSYNC();
    const float scale = C.scale[k];
    const float range = __fsub_rn(hi, lo);
    int cnt = 0;
    for (int base = 0; base < L; base += 256) {       // four positions per lane and round: their LDS reads and divisions overlap
      bool dropped[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int l = base + 64 * j + lane;
        dropped[j] = false;
        if (l < L) {
          bool d;
          if (C.mode == DD_MASK_IBLIP_QUANTILE) {
            d = e[l] >= thr;  // instructblip.py:453
          } else {
            const float r = (C.rng_mode == DD_RNG_MT19937) ? u[l] : P.uniforms[(size_t)k * L + l];
            // p = 0.1 + (mprob-0.1)*(clamp(e,lo,hi)-lo)/(hi-lo), every step rounded to fp32 (llava.py:646-647)
            const float c = fminf(fmaxf(e[l], lo), hi);
            const float p = __fadd_rn(0.1f, __fdiv_rn(__fmul_rn(scale, __fsub_rn(c, lo)), range));
            d = r < p;  // llava.py:653 (NaN p when hi == lo: nothing dropped)
          }
          uint8_t run = running[l] | (d ? 1 : 0);                      // llava.py:654-657, in place
          if (keep[l]) run = 0;                                        // llava.py:660 keep-restore (no-overlap modes / an empty set: keep[] is zero)
          running[l] = run;
          P.drop[(size_t)k * L + l] = run;
          dropped[j] = run != 0;
          if (P.drop_bits) {
            const uint8_t nb = (uint8_t)(((k & 7) ? bits[l] : 0) | ((run ? 1 : 0) << (k & 7)));   // (this lane wrote bits[l] for member k - 1)
            bits[l] = nb;
            if ((k & 7) == 7 || k == C.K - 1) P.drop_bits[(size_t)(k >> 3) * L + l] = nb;         // the plane is complete
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {                    // positions in order: round, then lane
        const unsigned long long b = __ballot(dropped[j]);
        if (P.idx && dropped[j]) P.idx[(size_t)k * L + cnt + __popcll(b & ((1ull << lane) - 1ull))] = base + 64 * j + lane;
        cnt += __popcll(b);
      }
    }
    if (P.idx)
      for (int l = cnt + lane; l < L; l += 64) P.idx[(size_t)k * L + l] = -1;
    if (lane == 0) P.n_drop[k] = cnt;  // masked_numbers (llava.py:661-662)
    DD_WSYNC();
  }
  if (C.rng_mode == DD_RNG_MT19937 && !P.rng_in) {
    for (int i = lane; i < MT_N; i += 64) P.rng_state[i] = mt[i];
    if (lane == 0) P.rng_state[MT_N] = (uint32_t)idx;
  }
}
static size_t sampler_wave_smem(int L) {
  size_t Lp = 64;
  while (Lp < (size_t)L) Lp <<= 1;
  return Lp * 11 + (MT_N + 8) * 4;
}

__global__ __launch_bounds__(64) void k_sample_masks(MaskParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (P.gate && *P.gate) return;          // sequence finished (EOS): draw nothing, write nothing
  __shared__ float sh_tab[128];
  {
    const float __attribute__((address_space(4)))* tab = (const float __attribute__((address_space(4)))*)(
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MaskParams, scale));
    sh_tab[threadIdx.x] = tab[threadIdx.x];
    sh_tab[threadIdx.x + 64] = tab[threadIdx.x + 64];
  }
  DD_WSYNC();
  const MaskSeq S = {P.epi, P.L, P.keep, P.uniforms, P.rng_state, P.drop, P.n_drop, P.idx, P.drop_bits, P.rng_in};
  const MaskConst C = {P.K, P.mode, P.rng_mode, sh_tab, sh_tab + 64};
  sample_masks_wave(C, S, smem, false);
}

// Group step: the keep set (models/llava.py:443-482 from the base argmax already on the device) and the K masks of up to
// 32 sequences in ONE launch, one wave per sequence, each from its own mt19937 state.
struct MaskLanes {
  MaskParams common;                 // K, mode, rng_mode, scale[], q[]; per-sequence fields below override the rest
  int n, k_top;
  const float* epi[32];
  int L[32];
  uint8_t* keep[32];
  const int32_t* argmax[32];
  const int32_t* topk[32];
  uint32_t* rng_state[32];
  uint8_t* drop[32];
  int32_t* n_drop[32];
  uint8_t* drop_bits[32];
  const int32_t* gate[32];
};
__global__ __launch_bounds__(64) void k_sample_masks_lanes(MaskLanes M) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int m = blockIdx.x;
  if (M.gate[m] && *M.gate[m]) return;    // this sequence finished (EOS): its stream and masks stay as they are
  const int L = M.L[m], k = M.k_top;
  const int tok = M.argmax[m][0];
  uint8_t* keep_sh = smem + sampler_lp(L) * 9;       // sample_masks_wave's keep[]
  const bool no_overlap = M.common.mode == DD_MASK_NEXT_NO_OVERLAP || M.common.mode == DD_MASK_LLAVA_CUMULATIVE_NO_OVERLAP;
  for (int l = threadIdx.x; l < L; l += 64) {
    bool hit = false;
    for (int j = 0; j < k; ++j) hit |= (M.topk[m][(size_t)l * k + j] == tok);
    M.keep[m][l] = hit ? 1 : 0;
    keep_sh[l] = (hit && !no_overlap) ? 1 : 0;
  }
  __shared__ float sh_tab[128];
  {  // scale[] and q[] are adjacent in the kernel arguments: 128 floats read straight from the kernarg segment
    const float __attribute__((address_space(4)))* tab = (const float __attribute__((address_space(4)))*)(
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MaskLanes, common.scale));
    sh_tab[threadIdx.x] = tab[threadIdx.x];
    sh_tab[threadIdx.x + 64] = tab[threadIdx.x + 64];
  }
  DD_WSYNC();
  // (uniforms / idx / rng_in are null for lanes — taken from the zeroed common block, not written as literal nullptr: with the
  // constants folded into the inlined body hipcc 7.2's instcombine dies on the dead injected-uniforms load)
  const MaskSeq S = {M.epi[m], L, M.keep[m], M.common.uniforms, M.rng_state[m], M.drop[m], M.n_drop[m], M.common.idx, M.drop_bits[m], M.common.rng_in};
  const MaskConst C = {M.common.K, M.common.mode, M.common.rng_mode, sh_tab, sh_tab + 64};
  sample_masks_wave(C, S, smem, true);
}
#ifdef DD_KEEP_SCRATCH_SAMPLER
#include "dd_sampler_block.h"          // the 1,024-thread forms, for tools/sampler_repro.py (dd_tools_set_tuning key 34: 1 scratch, 2 checking, 3 plain)
#endif

// (tools library) dynamic LDS the 1,024-thread forms of dd_sampler_block.h REQUEST: dd_tools_set_tuning key 48 = 0: the 76 KiB they use; 1: 156 KiB
// (round 4's fence: no kernel with more than ~3 KiB of LDS shares the CU); n > 1: n KiB (the request sweep of DESIGN.md 3e).
int g_sampler_lds_pad = 1;
#ifdef DD_KEEP_SCRATCH_SAMPLER
static size_t sampler_smem_used() { return (size_t)MASK_MAX_L * 4 * 2 + MASK_MAX_L + (MT_N + 8) * 4; }
static size_t sampler_smem() {
  const size_t used = sampler_smem_used();
  const size_t padded = (size_t)(g_sampler_lds_pad == 1 ? 156 : g_sampler_lds_pad) * 1024;
  return g_sampler_lds_pad && padded > used ? (padded > (size_t)156 * 1024 ? (size_t)156 * 1024 : padded) : used;
}
#endif
int dd_sample_masks_lanes(const MaskLaneArgs* lanes, int n, int k_top, const double* mprobs, int K, int mode, hipStream_t st) {
  DD_REQUIRE(lanes && n >= 1 && n <= 32 && K >= 1 && K <= 64 && mode >= 0 && mode <= 4, "dd_sample_masks_lanes: bad arguments");
  MaskLanes M;
  memset(&M, 0, sizeof(M));
  M.common.K = K, M.common.mode = mode, M.common.rng_mode = mode == DD_MASK_IBLIP_QUANTILE ? DD_RNG_INJECTED : DD_RNG_MT19937;
  for (int k = 0; k < K; ++k) {
    M.common.scale[k] = (float)(mprobs[k] - 0.1);
    M.common.q[k] = (float)(1.0 - mprobs[k]);
  }
  M.n = n, M.k_top = k_top;
  for (int m = 0; m < n; ++m) {
    DD_REQUIRE(lanes[m].L >= 1 && lanes[m].L <= MASK_MAX_L, "dd_sample_masks_lanes: L out of range");
    DD_REQUIRE(mode == DD_MASK_IBLIP_QUANTILE || lanes[m].rng_state, "dd_sample_masks_lanes: sequence %d needs an rng", m);
    M.epi[m] = lanes[m].epi, M.L[m] = lanes[m].L, M.keep[m] = lanes[m].keep, M.argmax[m] = lanes[m].argmax;
    M.topk[m] = lanes[m].topk, M.rng_state[m] = lanes[m].rng_state, M.drop[m] = lanes[m].drop, M.n_drop[m] = lanes[m].n_drop;
    M.drop_bits[m] = lanes[m].drop_bits, M.gate[m] = lanes[m].gate;
  }
  int Lmax = 1;
  for (int m = 0; m < n; ++m) Lmax = lanes[m].L > Lmax ? lanes[m].L : Lmax;
#ifdef DD_KEEP_SCRATCH_SAMPLER
  if (g_lanes_sampler_scratch) {
    const size_t smem = sampler_smem();
    static bool attr2 = false;
    if (!attr2) {
      DD_HIP(hipFuncSetAttribute((const void*)k_sample_masks_lanes_dbg, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      DD_HIP(hipFuncSetAttribute((const void*)k_sample_masks_lanes_scratch, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      DD_HIP(hipFuncSetAttribute((const void*)k_sample_masks_lanes_block, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      attr2 = true;
    }
    if (g_lanes_sampler_scratch == 2) {
      DD_REQUIRE(g_sampler_dbg_buf, "dd_sample_masks_lanes: the checking sampler needs dd_tools_sampler_dbg_attach first");
      k_sample_masks_lanes_dbg<<<n, MASK_THREADS, smem, st>>>(M, g_sampler_dbg_buf, ++g_sampler_dbg_tag, (int)sampler_smem_used());
    } else if (g_lanes_sampler_scratch == 3) {
      k_sample_masks_lanes_block<<<n, MASK_THREADS, smem, st>>>(M);
    } else {
      k_sample_masks_lanes_scratch<<<n, MASK_THREADS, smem, st>>>(M);
    }
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
#endif
  const size_t smem = sampler_wave_smem(Lmax);
  static bool attr_set = false;
  if (!attr_set) {
    DD_HIP(hipFuncSetAttribute((const void*)k_sample_masks_lanes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sampler_wave_smem(MASK_MAX_L)));
    attr_set = true;
  }
  k_sample_masks_lanes<<<n, 64, smem, st>>>(M);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

int dd_sample_masks_impl(const float* epi, int L, const double* mprobs, int K, const uint8_t* keep, int mode,
                         int rng_mode, const float* uniforms, uint32_t* rng_state, uint8_t* drop, int32_t* n_drop,
                         int32_t* idx, uint8_t* drop_bits, const int32_t* gate, hipStream_t st, const uint32_t* rng_in,
                         bool empty_keep) {
  DD_REQUIRE(epi && mprobs && drop && n_drop, "dd_sample_masks: null pointer");
  DD_REQUIRE(L >= 1 && L <= MASK_MAX_L, "dd_sample_masks: L=%d out of range (1..%d)", L, MASK_MAX_L);
  DD_REQUIRE(K >= 1 && K <= 64, "dd_sample_masks: K=%d out of range (1..64)", K);
  DD_REQUIRE(mode >= 0 && mode <= 5, "dd_sample_masks: unknown mode %d", mode);
  DD_REQUIRE(mode == DD_MASK_NEXT_NO_OVERLAP || mode == DD_MASK_LLAVA_CUMULATIVE_NO_OVERLAP || keep || empty_keep,
             "dd_sample_masks: keep flags required for mode %d", mode);
  if (mode != DD_MASK_IBLIP_QUANTILE) {
    DD_REQUIRE(rng_mode == DD_RNG_INJECTED || rng_mode == DD_RNG_MT19937, "dd_sample_masks: unknown rng mode %d", rng_mode);
    DD_REQUIRE(rng_mode != DD_RNG_INJECTED || uniforms, "dd_sample_masks: injected rng needs uniforms");
    DD_REQUIRE(rng_mode != DD_RNG_MT19937 || rng_state, "dd_sample_masks: mt19937 rng needs a dd_rng");
  } else {
    rng_mode = DD_RNG_INJECTED;
  }
  MaskParams P;
  memset(&P, 0, sizeof(P));
  P.epi = epi, P.L = L, P.K = K, P.mode = mode, P.rng_mode = rng_mode, P.keep = keep, P.uniforms = uniforms;
  P.rng_state = rng_state, P.drop = drop, P.n_drop = n_drop, P.idx = idx, P.drop_bits = drop_bits, P.gate = gate, P.rng_in = rng_in;
  if (empty_keep) P.keep = nullptr;
  for (int k = 0; k < K; ++k) {
    P.scale[k] = (float)(mprobs[k] - 0.1);  // double subtraction, then one rounding to fp32
    P.q[k] = (float)(1.0 - mprobs[k]);
  }
  const size_t smem = sampler_wave_smem(L);
  static bool attr_set = false;
  if (!attr_set) {
    DD_HIP(hipFuncSetAttribute((const void*)k_sample_masks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sampler_wave_smem(MASK_MAX_L)));
    attr_set = true;
  }
  k_sample_masks<<<1, 64, smem, st>>>(P);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

extern "C" int dd_sample_masks(const float* epi, int L, const double* mprobs, int K, const uint8_t* keep, int mode,
                               int rng_mode, const float* uniforms, dd_rng* rng, uint8_t* drop, int32_t* n_drop,
                               int32_t* idx, void* stream_) {
  return dd_sample_masks_impl(epi, L, mprobs, K, keep, mode, rng_mode, uniforms, rng ? rng->state : nullptr, drop,
                              n_drop, idx, nullptr, nullptr, (hipStream_t)stream_, nullptr, false);
}

// ----------------------------------------------------------------------------------------------
// Speculative single-sequence step (dd_engine.hip): the K members ran in the same sweep as the un-masked row with masks
// sampled for an EMPTY keep set.  That is the reference's result exactly when no member dropped a token of the real keep
// set (models/llava.py:660 would have restored it).  ok_out = 1: the speculative masks and member rows stand (also when the
// sequence is finished: nothing to redo); 0: the members must be re-run with the real keep set.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_spec_check(const uint8_t* __restrict__ keep, const uint8_t* __restrict__ drop_bits, int L,
                                                     int K, const int32_t* __restrict__ done, int32_t* __restrict__ ok_out,
                                                     int keep_matters, volatile int32_t* host_note) {
  __shared__ int hit;
  if (threadIdx.x == 0) hit = 0;
  __syncthreads();
  int h = 0;
  if (keep_matters && !*done) {
    const unsigned mask = (1u << K) - 1u;
    for (int l = threadIdx.x; l < L; l += 1024) h |= (keep[l] && (drop_bits[l] & mask)) ? 1 : 0;
  }
  if (h) hit = 1;                       // benign race: every writer stores 1
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ok = hit ? 0 : 1;
    ok_out[0] = ok;
    if (host_note) {                    // host-decided fallback: tell the waiting host thread (pinned, device-mapped words)
      const int seq = ok_out[2] + 1;    // checks announced so far (device copy of the counter)
      ok_out[2] = seq;
      host_note[1] = ok;
      __threadfence_system();
      host_note[0] = seq;
    }
  }
}
int dd_spec_check(const uint8_t* keep, const uint8_t* drop_bits, int L, int K, const int32_t* done, int32_t* ok_out,
                  int keep_matters, hipStream_t st, int32_t* host_note) {
  k_spec_check<<<1, 1024, 0, st>>>(keep, drop_bits, L, K, done, ok_out, keep_matters, host_note);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
__global__ void k_copy_row_gated(const float* __restrict__ src, float* __restrict__ dst, int n, const int32_t* __restrict__ gate) {
  if (gate && *gate) return;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
int dd_copy_row_gated(const float* src, float* dst, int n, const int32_t* gate, hipStream_t st) {
  k_copy_row_gated<<<(n + 255) / 256, 256, 0, st>>>(src, dst, n, gate);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// ----------------------------------------------------------------------------------------------
// vote (select_by_vote, reference models/llava.py:22-36)
// ----------------------------------------------------------------------------------------------
__global__ void k_vote(const int32_t* __restrict__ ids, int K, int32_t* __restrict__ out2, const int32_t* __restrict__ gate) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (gate && *gate) return;
  int best_k = 0, best_c = 0;
  for (int k = 0; k < K; ++k) {
    bool first = true;
    for (int j = 0; j < k; ++j) first &= (ids[j] != ids[k]);
    if (!first) continue;  // Counter keys are in first-insertion order
    int c = 0;
    for (int j = 0; j < K; ++j) c += (ids[j] == ids[k]);
    if (c > best_c) {      // strict: ties keep the earlier-inserted id (Counter.most_common)
      best_c = c;
      best_k = k;
    }
  }
  out2[0] = best_k;        // first member whose argmax is the majority id
  out2[1] = ids[best_k];
}

// the votes of up to 4 sequences in one launch (block = sequence)
struct VoteLanes {
  const int32_t* ids[8];
  int32_t* out2[8];
  const int32_t* gate[8];
};
__global__ void k_vote_lanes(VoteLanes t, int K) {
  if (threadIdx.x != 0) return;
  if (t.gate[blockIdx.x] && *t.gate[blockIdx.x]) return;
  const int32_t* ids = t.ids[blockIdx.x];
  int best_k = 0, best_c = 0;
  for (int k = 0; k < K; ++k) {
    bool first = true;
    for (int j = 0; j < k; ++j) first &= (ids[j] != ids[k]);
    if (!first) continue;
    int c = 0;
    for (int j = 0; j < K; ++j) c += (ids[j] == ids[k]);
    if (c > best_c) {
      best_c = c;
      best_k = k;
    }
  }
  t.out2[blockIdx.x][0] = best_k;
  t.out2[blockIdx.x][1] = ids[best_k];
}
int dd_vote_lanes(const int32_t* const* ids, int32_t* const* out2, const int32_t* const* gates, int n, int K, hipStream_t st) {
  DD_REQUIRE(ids && out2 && n >= 1 && n <= 8 && K >= 1 && K <= 4096, "dd_vote_lanes: bad arguments");
  VoteLanes t;
  memset(&t, 0, sizeof(t));
  for (int i = 0; i < n; ++i) t.ids[i] = ids[i], t.out2[i] = out2[i], t.gate[i] = gates ? gates[i] : nullptr;
  k_vote_lanes<<<n, 64, 0, st>>>(t, K);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

extern "C" int dd_vote(const int32_t* ids, int K, int32_t* out2, void* stream_) {
  DD_REQUIRE(ids && out2 && K >= 1 && K <= 4096, "dd_vote: bad arguments");
  k_vote<<<1, 64, 0, (hipStream_t)stream_>>>(ids, K, out2, nullptr);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int dd_vote_gated(const int32_t* ids, int K, int32_t* out2, const int32_t* gate, hipStream_t st) {
  k_vote<<<1, 64, 0, st>>>(ids, K, out2, gate);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

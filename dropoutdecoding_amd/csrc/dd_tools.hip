// Measurement hooks and experiment knobs: compiled into libdropdec_tools.so only (the product objects + this file), used by
// bench.py's roofline leg and the scripts under tools/.  Not part of the drop-in boundary: include/dropdec_tools.h.
#include "dd_engine_internal.h"
#include "../../include/dropdec_tools.h"

#define RC(expr)              \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != DD_OK) return rc__; \
  } while (0)

extern "C" int dd_lm_time_sweep(dd_lm* h, int nb, int iters, float* mean_ms, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h && h->prefilled && mean_ms && nb >= 1 && nb <= 8 && iters >= 1, "dd_lm_time_sweep: bad arguments");
  RC(lm_sweep(h, nb, nullptr, 0, h->member_logits, st));  // warm
  DD_HIP(hipEventRecord(h->ev0, st));
  for (int i = 0; i < iters; ++i) RC(lm_sweep(h, nb, nullptr, 0, h->member_logits, st));
  DD_HIP(hipEventRecord(h->ev1, st));
  DD_HIP(hipEventSynchronize(h->ev1));
  float ms = 0;
  DD_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *mean_ms = ms / iters;
  return DD_OK;
}

// Time ONE decode GEMV kind in isolation with HIP events on `stream`, cycling through the layers' weights so that
// every launch streams bytes that are not resident in the 256 MiB Infinity Cache (bench.py roofline leg).
// which: 0 qkv, 1 o_proj, 2 gate/up (+SiLU), 3 down_proj.  bytes_per_launch = algorithmic weight bytes (bf16).
extern "C" int dd_lm_time_gemv(dd_lm* h, int which, int nb, int iters, float* mean_ms, double* bytes_per_launch,
                               void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  const bool stream_only = which >= 8;      // 8 + kind: the slice-resident path's streaming kernel alone (no finishing kernel)
  if (stream_only) which -= 8;
  DD_REQUIRE(h && mean_ms && bytes_per_launch && which >= 0 && which <= 3 && ((nb >= 1 && nb <= 8) || nb == 16 || nb == 32 || nb == 64 || nb == 72) && iters >= 1,
             "dd_lm_time_gemv: bad arguments (nb 1..8, or 16 / 32 / 64 / 72 = the two- / four- / eight- / nine-group kernel)");
  struct Restore {
    ~Restore() { ddk_set_slices_only(0); }
  } restore_;
  ddk_set_slices_only(stream_only ? 1 : 0);
  const int ngroups = nb >= 16 ? nb / 8 : 0;
  const bool wide = ngroups > 0;
  if (nb >= 16) nb = 8;
  auto gemv = [&](int epi, GemvArgs& a) -> int {
    a.S_next = epi == EPI_SILU ? h->S_ff : h->S_d;
    a.n_groups = ngroups;
    a.part = h->gemv_part, a.part_floats = h->gemv_part_floats;
    return wide ? ddk_gemv_groups(epi, a, st) : ddk_gemv(epi, a, st);
  };
  const int d = h->d, dff = h->dff;
  auto launch = [&](int l) -> int {
    LayerW& w = h->lw[l % h->Lyr];
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.nb = nb, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps, a.state = h->state, a.fp8 = h->fp8;
    switch (which) {
      case 0:
        a.W = w.wqkv, a.wscale = w.s_qkv, a.S = h->S_d, a.n_tiles = h->qkv_tiles, a.xop = h->xop_d, a.ssq_in = h->ssq_a, a.ssq_n = d / 16, a.ssq_ld = d / 16;
        a.qbuf = h->qbuf, a.knew = h->knew, a.vnew = h->vnew, a.q_tiles = h->q_tiles, a.k_tiles = h->k_tiles;
        a.q_dim = h->q_dim, a.kv_dim = h->kv_dim, a.rope_cos = h->rope_cos, a.rope_sin = h->rope_sin;
        return gemv(EPI_QKV, a);
      case 1:
        a.W = w.wo, a.wscale = w.s_o, a.S = h->S_q, a.n_tiles = d / 16, a.xop = h->xop_q, a.out = h->xa, a.ldo = d;
        a.normw_next = w.norm2, a.xop_next = h->xop_d, a.ssq_out = h->ssq_b, a.ssq_ld = d / 16;
        return gemv(EPI_RESID, a);
      case 2:
        a.W = w.wgu, a.wscale = w.s_gu, a.S = h->S_d, a.n_tiles = dff / 16, a.xop = h->xop_d, a.ssq_in = h->ssq_b, a.ssq_n = d / 16, a.ssq_ld = d / 16;
        a.xop_next = h->xop_ff;
        return gemv(EPI_SILU, a);
      default:
        a.W = w.wdown, a.wscale = w.s_down, a.S = h->S_ff, a.n_tiles = d / 16, a.xop = h->xop_ff, a.out = h->xa, a.ldo = d;
        a.normw_next = w.norm1, a.xop_next = h->xop_d, a.ssq_out = h->ssq_a, a.ssq_ld = d / 16;
        return gemv(EPI_RESID, a);
    }
  };
  for (int i = 0; i < h->Lyr; ++i) RC(launch(i));  // warm (also evicts)
  DD_HIP(hipEventRecord(h->ev0, st));
  for (int i = 0; i < iters; ++i) RC(launch(i));
  DD_HIP(hipEventRecord(h->ev1, st));
  DD_HIP(hipEventSynchronize(h->ev1));
  float ms = 0;
  DD_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *mean_ms = ms / iters;
  double rows[4] = {(double)(h->q_dim + 2 * h->kv_dim) * d, (double)d * h->q_dim, 2.0 * dff * d, (double)d * dff};
  *bytes_per_launch = rows[which] * (h->fp8 ? 1.0 : 2.0);
  return DD_OK;
}

// Experiment knobs (A/B switches of kernel variants; every setting produces the same bits).  Keys as the tools/ scripts and
// DESIGN.md quote them: 0 = GEMV weight tiles in flight per wave (4/8/16), 4 = ring (1) or batch (0) request order of the 8-row
// GEMV, 9 = sequences per member sweep in dd_lm_group_step (1, 2, 4, 8), 10 = workgroups per group of 8 members in the grouped
// decode attention, 12 = prefill attention on the matrix cores, 17 / 18 / 19 = workgroups per K slice of the 64-row qkv / o_proj /
// gate-up GEMV (18 < 0: eight-plane o_proj kernel; 19 < 0: single K slices for gate/up), 21 = key tiles per workgroup of the
// fp16-cache decode attention, 22 = all-tiles form of that attention, 23 = concurrent member sweeps of a
// classic group step, 24 = four-columns-per-thread finishing kernel of the slice GEMVs (bit mask over the epilogues), 26 = rider form of the
// group step (0: the classic form always), 27 = the riding rows' attention inside the members' launches, 28 = branches of the rider
// form (1..4), 29 = weight requests in flight per wave of the nine-plane qkv / gate-up kernels (4 or 8), 30 = half planes for K <= 4, 31 = half
// planes (classic form) before the rider form for line-ups that are not whole groups of fourteen, 33 = rider rings in stages (masks
// between the stages on the caller's stream), 34 = the lanes mask sampler in one of its 1,024-thread forms (dd_sampler_block.h, this library only: 1 =
// round 3's, 616 bytes of private scratch per lane; 2 = the checking form; 3 = round 4's product kernel; 0 = the product's one-wave sampler), 48 = the
// dynamic-LDS request of those forms (0: the 76 KiB they use, 1: 156 KiB, n: n KiB); the product switches
// (8, 11, 13-16) are forwarded to dd_set_tuning.  Every call bumps the graph-key epoch: steps captured under other settings are not replayed.
extern int g_exp_G[4];
extern int g_attn16_tpw, g_attn16_full, g_finish4, g_attn16_gh_all, g_attn32_lds_pad, g_attn32_nopk, g_attn_comb_pre;
void dd_engine_set_pairs(int on);
void dd_engine_set_branches(int n);
void dd_engine_set_rider(int on);
void dd_engine_set_ride_beside(int on);
void dd_engine_set_rider_branches(int n);
void dd_engine_set_half_planes(int on);
void dd_engine_set_half_planes_first(int on);
void dd_engine_set_rider_staged(int on);
void dd_engine_set_fp32_fork(int on);
void dd_engine_set_mask_branches(int on);
void dd_engine_set_attn_masked(int mode);
void dd_engine_set_unmask(int m);
extern int g_exp_U9, g_exp_temporal, g_rmsnorm16, g_prefill_attn16, g_attn16_ride_pf, g_sampler_lds_pad, g_seq_prog, g_rstd_wg, g_fp8_xpf, g_seq_quads, g_gemv_loop;
void dd_dropout_set_lanes_sampler_scratch(int on);   // dd_dropout.hip compiled with -DDD_KEEP_SCRATCH_SAMPLER (this library only)
void dd_dropout_set_sampler_dbg(uint32_t* buf);
uint32_t dd_dropout_sampler_dbg_tag();
size_t dd_dropout_sampler_dbg_words();
// The checking sampler (tools key 34 = 2): buf_dev = dd_tools_sampler_dbg_words() zeroed uint32 words; word 0 != 0 afterwards = a generator-block
// check failed, word 1 = how many, words 2.. = the first failure's dump (dd_dropout.hip mt_dbg_check).  NULL detaches.
extern "C" int dd_tools_sampler_dbg_attach(uint32_t* buf_dev) {
  dd_dropout_set_sampler_dbg(buf_dev);
  return DD_OK;
}
extern "C" size_t dd_tools_sampler_dbg_words(void) { return dd_dropout_sampler_dbg_words(); }
extern "C" unsigned int dd_tools_sampler_dbg_launches(void) { return dd_dropout_sampler_dbg_tag(); }
extern "C" int dd_tools_set_tuning(int key, int value) {
  dd_engine_bump_epoch();
  if (key == 8 || key == 11 || (key >= 13 && key <= 16) || key == 20) return dd_set_tuning(key, value);
  DD_REQUIRE(key == 0 || key == 1 || key == 2 || key == 4 || key == 9 || key == 10 || key == 12 || (key >= 17 && key <= 19) || (key >= 21 && key <= 24) || (key >= 26 && key <= 31) || key == 33 || key == 34 || key == 36 || key == 37 || key == 38 || key == 39 || key == 40 || key == 41 || key == 42 || key == 43 || key == 45 || key == 46 || key == 47 || key == 48 || key == 49 || key == 50 || key == 52 || key == 53 || key == 54 || key == 55,
             "dd_tools_set_tuning: unknown key %d", key);
  if (key == 9) dd_engine_set_pairs(value);
  else if (key == 10) ddk_set_attn_split(value);
  else if (key == 12) ddk_set_prefill_mfma(value);
  else if (key >= 17 && key <= 19) g_exp_G[key - 17] = value;
  else if (key == 21) g_attn16_tpw = value;
  else if (key == 22) g_attn16_full = value;
  else if (key == 23) dd_engine_set_branches(value);
  else if (key == 24) g_finish4 = value;
  else if (key == 26) dd_engine_set_rider(value);
  else if (key == 27) dd_engine_set_ride_beside(value);
  else if (key == 28) dd_engine_set_rider_branches(value);
  else if (key == 29) g_exp_U9 = value;
  else if (key == 30) dd_engine_set_half_planes(value);
  else if (key == 31) dd_engine_set_half_planes_first(value);
  else if (key == 33) dd_engine_set_rider_staged(value);
  else if (key == 34) dd_dropout_set_lanes_sampler_scratch(value);
  else if (key == 36) g_exp_temporal = value;
  else if (key == 37) dd_engine_set_fp32_fork(value);
  else if (key == 38) g_attn16_gh_all = value;
  else if (key == 39) g_attn32_lds_pad = value;
  else if (key == 40) dd_engine_set_mask_branches(value);
  else if (key == 41) dd_engine_set_attn_masked(value);
  else if (key == 42) dd_engine_set_unmask(value);
  else if (key == 43) g_attn32_nopk = value;
  else if (key == 45) g_rmsnorm16 = value;
  else if (key == 46) g_prefill_attn16 = value;
  else if (key == 47) g_attn16_ride_pf = value;
  else if (key == 48) g_sampler_lds_pad = value;
  else if (key == 49) g_seq_prog = value;
  else if (key == 50) g_rstd_wg = value;
  else if (key == 52) g_fp8_xpf = value;
  else if (key == 53) g_seq_quads = value;
  else if (key == 54) g_gemv_loop = value;
  else if (key == 55) g_attn_comb_pre = value;
  else ddk_set_tuning(key, value);      // 0, 4; 1 and 2 are settled (accepted, ignored)
  return DD_OK;
}

// Name of the decode-GEMV STREAMING kernel the last ddk_gemv / ddk_gemv_groups call of this thread launched, as a kernel trace
// prints it (template arguments included): bench.py matches its roofline kernel against rocprofv3 output by this string.
const char* ddk_last_gemv_kernel();
extern "C" const char* dd_tools_last_gemv_kernel(void) { return ddk_last_gemv_kernel(); }

// Calibration: plain streaming READ bandwidth of this device over a large buffer (grid-stride 16-byte loads, 8 per
// thread in flight), timed with HIP events.  bench.py reports it next to the 8 TB/s spec peak so the roofline
// fraction can also be read against what this board actually delivers for a read-only stream.
__global__ __launch_bounds__(256) void k_stream_read(const u32x4_t* __restrict__ p, size_t n16, unsigned int* sink) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned int acc = 0;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].w;
  }
  for (; i < n16; i += stride) acc ^= p[i].x;
  if (acc == 0x9e3779b9u) sink[0] = acc;
}
extern "C" int dd_hbm_read_bench(const void* buf_dev, size_t bytes, int iters, int n_blocks, float* gbs_out, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(buf_dev && gbs_out && bytes >= (1u << 20) && iters >= 1, "dd_hbm_read_bench: bad arguments");
  unsigned int* sink = nullptr;
  hipEvent_t e0, e1;
  DD_HIP(hipMalloc((void**)&sink, 16));
  DD_HIP(hipEventCreate(&e0));
  DD_HIP(hipEventCreate(&e1));
  if (n_blocks <= 0) n_blocks = 4096;
  k_stream_read<<<n_blocks, 256, 0, st>>>((const u32x4_t*)buf_dev, bytes / 16, sink);
  DD_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) k_stream_read<<<n_blocks, 256, 0, st>>>((const u32x4_t*)buf_dev, bytes / 16, sink);
  DD_HIP(hipEventRecord(e1, st));
  DD_HIP(hipEventSynchronize(e1));
  float ms = 0;
  DD_HIP(hipEventElapsedTime(&ms, e0, e1));
  *gbs_out = (float)((double)bytes * iters / (ms * 1e-3) / 1e9);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  return DD_OK;
}


// -------------------------------------------------------------------------------------------------------------------------------
// Determinism diagnostics (DESIGN.md "Determinism"; tools/stress_lanes.py, tests/test_gpu_sampler_repro.py)
// -------------------------------------------------------------------------------------------------------------------------------
// Per-step trace: after every group_finish one 32-int record per sequence, so that two runs can be compared quantity by quantity and
// the FIRST one that differs named: [0] tokens emitted so far, [1] argmax of the un-masked row, [2] size of the keep set, [3] hash of the
// drop bit planes, [4..11] n_drop of members 0..7, [12..19] the members' argmax ids, [20] winner, [21] token, [22] mt19937 read index
// (-1 unknown), [23] hash of the un-masked row's logits.
#include <map>
struct TraceBuf {
  int32_t* buf;
  int cap;
  const uint32_t* rng_state;
};
static std::map<dd_lm*, TraceBuf> g_trace;
struct TraceLanes {
  const DDState* st[8];
  const int32_t* argmax_base[8];
  const uint8_t* keep[8];
  const uint8_t* drop_bits[8];
  const int32_t* n_drop[8];
  const int32_t* member_tok[8];
  const float* base_logits[8];
  const uint32_t* rng_state[8];
  int32_t* buf[8];
  int cap[8], L[8];
};
__global__ __launch_bounds__(256) void k_trace_step(TraceLanes t, int K, int V) {
  const int q = blockIdx.x;
  if (!t.buf[q]) return;
  const DDState* st = t.st[q];
  const int idx = st->n_tok - 2;                       // the step that just ended emitted token n_tok - 1
  if (idx < 0 || idx >= t.cap[q]) return;
  __shared__ uint32_t sh[2][4];
  uint32_t nk = 0, hb = 0, hl = 0;
  const int planes = (K + 7) / 8;
  for (int l = threadIdx.x; l < t.L[q]; l += 256) {
    nk += t.keep[q][l] ? 1u : 0u;
    for (int p = 0; p < planes; ++p) hb += (uint32_t)t.drop_bits[q][(size_t)p * t.L[q] + l] * (2654435761u * (uint32_t)(l + 1 + p * 8191));
  }
  for (int v = threadIdx.x; v < V; v += 256) hl += __float_as_uint(t.base_logits[q][v]) * (2246822519u * (uint32_t)(v + 1));
  for (int o = 32; o > 0; o >>= 1) nk += __shfl_xor(nk, o), hb += __shfl_xor(hb, o), hl += __shfl_xor(hl, o);
  __shared__ uint32_t sk[4], sb[4], sl[4];
  if ((threadIdx.x & 63) == 0) sk[threadIdx.x >> 6] = nk, sb[threadIdx.x >> 6] = hb, sl[threadIdx.x >> 6] = hl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t* r = t.buf[q] + (size_t)idx * 32;
    r[0] = st->n_tok, r[1] = t.argmax_base[q][0], r[2] = (int32_t)(sk[0] + sk[1] + sk[2] + sk[3]);
    r[3] = (int32_t)(sb[0] + sb[1] + sb[2] + sb[3]);
    for (int k = 0; k < 8; ++k) r[4 + k] = k < K ? t.n_drop[q][k] : 0, r[12 + k] = k < K ? t.member_tok[q][k] : 0;
    r[20] = st->winner, r[21] = st->cur_tok, r[22] = t.rng_state[q] ? (int32_t)t.rng_state[q][624] : -1;
    r[23] = (int32_t)(sl[0] + sl[1] + sl[2] + sl[3]);
  }
}
static int trace_hook(dd_lm* const* qs, int ng, int K, hipStream_t st) {
  TraceLanes t;
  memset(&t, 0, sizeof(t));
  bool any = false;
  for (int g = 0; g < ng && g < 8; ++g) {
    dd_lm* q = qs[g];
    auto it = g_trace.find(q);
    t.st[g] = q->state, t.argmax_base[g] = q->argmax_base, t.keep[g] = q->keep, t.drop_bits[g] = q->drop_bits, t.n_drop[g] = q->n_drop;
    t.member_tok[g] = q->member_tok, t.base_logits[g] = q->base_logits, t.L[g] = q->L;
    if (it != g_trace.end()) t.buf[g] = it->second.buf, t.cap[g] = it->second.cap, t.rng_state[g] = it->second.rng_state, any = true;
  }
  if (!any) return DD_OK;
  k_trace_step<<<ng, 256, 0, st>>>(t, K, qs[0]->V);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
uint32_t* dd_rng_state_ptr(dd_rng* r);
extern "C" int dd_tools_trace_attach(dd_lm* h, int32_t* buf_dev, int cap_steps, dd_rng* rng) {
  DD_REQUIRE(h, "dd_tools_trace_attach: null handle");
  dd_engine_bump_epoch();                               // the trace launch is part of a captured step
  if (!buf_dev || cap_steps <= 0) g_trace.erase(h);
  else g_trace[h] = {buf_dev, cap_steps, rng ? dd_rng_state_ptr(rng) : nullptr};
  dd_engine_group_finish_hook = g_trace.empty() ? nullptr : trace_hook;
  return DD_OK;
}

// LDS poison: `launches` grids of `wgs` workgroups that fill `lds_bytes` of LDS each with a quiet-NaN pattern and leave.  Enqueued on a
// side stream beside a step, they leave NaNs in whatever LDS the step's next workgroups are given: a kernel that reads LDS it did
// not write turns that into a token change.  (LDS is not cleared between workgroups on this hardware.)
__global__ __launch_bounds__(256) void k_lds_poison(int n4, uint32_t salt, uint32_t* sink) {
  extern __shared__ __align__(16) uint32_t lds_p[];
  for (int i = threadIdx.x; i < n4; i += 256) lds_p[i] = 0x7FC00000u | ((salt + i) & 0x3FFFFFu);
  __syncthreads();
  if (lds_p[(threadIdx.x * 7) % n4] == 0x12345u) sink[0] = 1;     // keep the stores
}
extern "C" int dd_tools_lds_poison(int launches, int wgs, int lds_bytes, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(launches >= 1 && wgs >= 1 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024, "dd_tools_lds_poison: bad arguments");
  static uint32_t* sink = nullptr;
  static int attr_bytes = 0;
  if (!sink) DD_HIP(hipMalloc((void**)&sink, 16));
  if (lds_bytes > attr_bytes) {
    DD_HIP(hipFuncSetAttribute((const void*)k_lds_poison, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_bytes = lds_bytes;
  }
  static uint32_t salt = 1;
  for (int i = 0; i < launches; ++i) {
    k_lds_poison<<<wgs, 256, lds_bytes, st>>>(lds_bytes / 4, salt++, sink);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// Scratch probe: a kernel shaped like the round-3 sampler (1,024 threads per workgroup, a 154-word private array indexed at run time
// = 616 bytes of scratch per lane) that writes a lane-specific pattern into its scratch, lingers (`spin` rounds of dependent LDS
// traffic with barriers), reads the pattern back and counts mismatches — the direct test of "private scratch is not private beside
// other queues' kernels".  errors_dev[0] += mismatching words.
__global__ __launch_bounds__(1024) void k_scratch_probe(int spin, uint32_t salt, const int* __restrict__ perm, unsigned int* errors) {
  __shared__ uint32_t sh[1024];
  uint32_t priv[154];
  const uint32_t me = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + salt;
  for (int i = 0; i < 154; ++i) priv[perm[i]] = me ^ (uint32_t)(i * 40503u);
  uint32_t v = me;
  for (int r = 0; r < spin; ++r) {
    sh[threadIdx.x] = v;
    __syncthreads();
    v = v * 1664525u + sh[(threadIdx.x * 33 + r) & 1023];
    __syncthreads();
  }
  unsigned int bad = 0;
  for (int i = 0; i < 154; ++i) bad += priv[perm[i]] != (me ^ (uint32_t)(i * 40503u)) ? 1u : 0u;
  if (v == 0xdeadbeefu) bad += 1u << 30;                 // keep the loop
  if (bad) atomicAdd(errors, bad);
}
extern "C" int dd_tools_scratch_probe(int launches, int wgs, int spin, unsigned int* errors_dev, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(launches >= 1 && wgs >= 1 && spin >= 0 && errors_dev, "dd_tools_scratch_probe: bad arguments");
  static int* perm = nullptr;
  if (!perm) {
    int hperm[154];
    for (int i = 0; i < 154; ++i) hperm[i] = (i * 37) % 154;      // a permutation: the index is a run-time value for the compiler
    DD_HIP(hipMalloc((void**)&perm, sizeof(hperm)));
    DD_HIP(hipMemcpy(perm, hperm, sizeof(hperm), hipMemcpyHostToDevice));
  }
  static uint32_t salt = 7;
  for (int i = 0; i < launches; ++i) {
    k_scratch_probe<<<wgs, 1024, 0, st>>>(spin, salt++, perm, errors_dev);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// The lanes mask sampler on its own (the kernel the group step launches: keep sets + K masks of n sequences, one workgroup each), for the
// unit reproducer: arrays of n device pointers / lengths as the engine would pass them.
extern "C" int dd_tools_sample_masks_lanes(int n, const float* const* epi, const int32_t* L, uint8_t* const* keep, const int32_t* const* argmax,
                                           const int32_t* const* topk, dd_rng* const* rngs, uint8_t* const* drop, int32_t* const* n_drop,
                                           uint8_t* const* drop_bits, int k_top, const double* mprobs, int K, int mode, void* stream_) {
  DD_REQUIRE(n >= 1 && n <= 32 && epi && L && keep && argmax && topk && rngs && drop && n_drop && drop_bits && mprobs, "dd_tools_sample_masks_lanes: bad arguments");
  MaskLaneArgs ml[32];
  for (int i = 0; i < n; ++i) ml[i] = {epi[i], L[i], keep[i], argmax[i], topk[i], dd_rng_state_ptr(rngs[i]), drop[i], n_drop[i], drop_bits[i], nullptr};
  return dd_sample_masks_lanes(ml, n, k_top, mprobs, K, mode, (hipStream_t)stream_);
}

// Per-stage checksums of the multi-group sweeps (dd_engine.hip dbg_sum): trace_dev [cap_sweeps][n_layers][8] uint32, zeroed by the caller;
// sweeps are numbered in the order the host enqueues them from this call on.  NULL switches it off.
extern uint32_t* g_dbg_trace;
extern int g_dbg_trace_cap, g_dbg_sweeps, g_dbg_replay;
extern "C" int dd_tools_sweep_trace(uint32_t* trace_dev, int cap_sweeps) {
  dd_engine_bump_epoch();
  g_dbg_replay = cap_sweeps < 0 ? 1 : 0;             // negative capacity: also launch every traced attention a second time (stages 11 / 12)
  if (cap_sweeps < 0) cap_sweeps = -cap_sweeps;
  g_dbg_trace = trace_dev, g_dbg_trace_cap = trace_dev ? cap_sweeps : 0, g_dbg_sweeps = 0;
  return DD_OK;
}

// Per-workgroup checksums inside the fp32-cache attention tile pass (k_attn_partial) of the traced sweeps: attn_dev [cap_sweeps][n_layers][stride_words],
// workgroup (x, y, z) of a launch at ((z * grid.y + y) * grid.x + x) * 8: K registers, V registers, q rows read from LDS, scores read, p written,
// p read, outputs read, 0.  Use together with dd_tools_sweep_trace (same sweep numbering); NULL: off.
extern uint32_t* g_dbg_attn;
extern size_t g_dbg_attn_stride;
extern "C" int dd_tools_attn_trace(uint32_t* attn_dev, size_t stride_words) {
  g_dbg_attn = attn_dev, g_dbg_attn_stride = attn_dev ? stride_words : 0;
  return DD_OK;
}

// LDS / barrier isolation probe: the exchange pattern of the fp32-cache attention tile pass (256 threads, 30,720 bytes of dynamic LDS: every
// wave writes two of eight rows of a [64][8] table, barrier, every thread reads whole rows written by the other waves) with values that
// are a function of (workgroup, round, position), verified in place for `rounds` rounds.  errors_dev[0] += mismatching words.
__global__ __launch_bounds__(256) void k_lds_barrier_probe(int rounds, uint32_t salt, unsigned int* errors) {
  extern __shared__ __align__(16) uint32_t pl[];        // 7680 words; the table sits where p_sh sits in the attention kernel
  uint32_t* tab = pl + 3072;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wg = (blockIdx.x * 2654435761u) ^ salt;
  unsigned int bad = 0;
  for (int rd = 0; rd < rounds; ++rd) {
    for (int r = wave; r < 8; r += 4) tab[lane * 8 + r] = wg + (uint32_t)(rd * 7919 + (lane * 8 + r) * 31);
    __syncthreads();
    for (int j = 0; j < 8; ++j) {
      const int key = (wave * 16 + 2 * j + (lane >> 5)) & 63;
      for (int r = 0; r < 8; ++r) bad += tab[key * 8 + r] != wg + (uint32_t)(rd * 7919 + (key * 8 + r) * 31) ? 1u : 0u;
    }
    __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
}
extern "C" int dd_tools_lds_barrier_probe(int launches, int wgs, int rounds, unsigned int* errors_dev, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(launches >= 1 && wgs >= 1 && rounds >= 1 && errors_dev, "dd_tools_lds_barrier_probe: bad arguments");
  static uint32_t salt = 11;
  for (int i = 0; i < launches; ++i) {
    k_lds_barrier_probe<<<wgs, 256, 30720, st>>>(rounds, salt++, errors_dev);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// LDS overlap probe: does a workgroup's dynamic LDS stay its own beside workgroups of ANOTHER dispatch with a different LDS size on the same CU?
// `k_lds_hold` fills all of its dynamic LDS with a pattern that is a function of (workgroup, salt, word), holds it for `hold` rounds of
// s_sleep + barrier, and verifies every word.  dd_tools_lds_overlap_probe runs a grid of 512-thread workgroups with lds_a bytes (A: the
// size of a slice-resident GEMV's operand planes) on one stream and, concurrently, `launches_b` grids of 256-thread workgroups with lds_b
// bytes (B: the fp32-cache attention tile pass's size) on another; errors_dev[0] / [1] += mismatching words seen by A / B workgroups,
// errors_dev[2] / [3] = lowest / highest mismatching word offset seen by B (atomicMin / atomicMax).
__global__ void k_lds_hold(int words, int hold, uint32_t salt, unsigned int* errors, int who) {
  extern __shared__ __align__(16) uint32_t pl[];
  const uint32_t me = (blockIdx.x * 2654435761u) ^ (salt * 40503u);
  for (int i = threadIdx.x; i < words; i += blockDim.x) pl[i] = me + (uint32_t)i * 2246822519u;
  __syncthreads();
  for (int h = 0; h < hold; ++h) {
    __builtin_amdgcn_s_sleep(64);
    __syncthreads();
  }
  unsigned int bad = 0, lo = 0xFFFFFFFFu, hi = 0;
  for (int i = threadIdx.x; i < words; i += blockDim.x)
    if (pl[i] != me + (uint32_t)i * 2246822519u) {
      ++bad;
      lo = (unsigned)i < lo ? (unsigned)i : lo, hi = (unsigned)i > hi ? (unsigned)i : hi;
    }
  if (bad) {
    atomicAdd(errors + who, bad);
    if (who == 1) atomicMin(errors + 2, lo), atomicMax(errors + 3, hi);
  }
}
extern "C" int dd_tools_lds_overlap_probe(int lds_a, int wgs_a, int hold_a, int lds_b, int wgs_b, int hold_b, int launches_b, unsigned int* errors_dev,
                                          void* stream_a, void* stream_b) {
  DD_REQUIRE(lds_a >= 0 && lds_a <= 160 * 1024 && lds_b >= 4 && lds_b <= 160 * 1024 && errors_dev && wgs_b >= 1 && launches_b >= 1, "dd_tools_lds_overlap_probe: bad arguments");
  static int attr_bytes = 0;
  const int mx = lds_a > lds_b ? lds_a : lds_b;
  if (mx > attr_bytes) {
    DD_HIP(hipFuncSetAttribute((const void*)k_lds_hold, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    attr_bytes = mx;
  }
  static uint32_t salt = 3;
  if (lds_a > 0 && wgs_a > 0) {
    k_lds_hold<<<wgs_a, 512, lds_a, (hipStream_t)stream_a>>>(lds_a / 4, hold_a, salt++, errors_dev, 0);
    DD_CHECK_LAUNCH();
  }
  for (int i = 0; i < launches_b; ++i) {
    k_lds_hold<<<wgs_b, 256, lds_b, (hipStream_t)stream_b>>>(lds_b / 4, hold_b, salt++, errors_dev, 1);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// Register / load probes for the co-residency question (tools/sampler_repro.py): VALU-only 256-thread workgroups shaped like the fp32-cache attention
// tile pass (about 120 VGPRs, 30,720 bytes of dynamic LDS so that they fit beside a slice-resident GEMV workgroup on a CU).
//   k_vgpr_hold:  56 registers per lane hold a pattern across `hold` rounds of s_sleep + barrier, then are verified.
//   k_gload_hold: 16 outstanding 16-byte global loads per lane from a buffer whose contents are a function of the address, requested up front
//                 (as the tile pass requests its K / V rows), consumed after `hold` rounds; verified against the function.
// errors_dev[0] += mismatching words.
__global__ __launch_bounds__(256) void k_vgpr_hold(int hold, uint32_t salt, unsigned int* errors) {
  extern __shared__ __align__(16) uint32_t pl[];
  uint32_t r[56];
  const uint32_t me = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u) ^ salt;
#pragma unroll
  for (int i = 0; i < 56; ++i) {
    r[i] = me + (uint32_t)i * 40503u;
    asm volatile("" : "+v"(r[i]));
  }
  pl[threadIdx.x] = me;
  for (int h = 0; h < hold; ++h) {
    __builtin_amdgcn_s_sleep(32);
    __syncthreads();
  }
  unsigned int bad = pl[threadIdx.x] != me ? 1u : 0u;
#pragma unroll
  for (int i = 0; i < 56; ++i) {
    asm volatile("" : "+v"(r[i]));
    bad += r[i] != me + (uint32_t)i * 40503u ? 1u : 0u;
  }
  if (bad) atomicAdd(errors, bad);
}
__global__ __launch_bounds__(256) void k_gload_hold(const u32x4_t* __restrict__ buf, size_t n16, int hold, uint32_t salt, unsigned int* errors) {
  extern __shared__ __align__(16) uint32_t pl[];
  u32x4_t v[16];
  size_t idx[16];
  const size_t base = ((size_t)(blockIdx.x * 256u + threadIdx.x) * 2654435761u + salt) % n16;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    idx[i] = (base + (size_t)i * 8191u * 64u) % n16;      // strided like rows of a cache
    v[i] = buf[idx[i]];
  }
  pl[threadIdx.x] = salt;
  for (int h = 0; h < hold; ++h) {
    __builtin_amdgcn_s_sleep(16);
    __syncthreads();
  }
  unsigned int bad = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t w = (uint32_t)(idx[i] * 4u) * 2246822519u;
    bad += (v[i].x != w) + (v[i].y != w + 1u) + (v[i].z != w + 2u) + (v[i].w != w + 3u);
  }
  if (bad) atomicAdd(errors, bad);
}
__global__ void k_gload_fill(u32x4_t* buf, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint32_t w = (uint32_t)(i * 4u) * 2246822519u;
    buf[i] = (u32x4_t){w, w + 1u, w + 2u, w + 3u};
  }
}
extern "C" int dd_tools_hold_probe(int kind, int launches, int wgs, int hold, unsigned int* errors_dev, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE((kind == 0 || kind == 1) && launches >= 1 && wgs >= 1 && hold >= 0 && errors_dev, "dd_tools_hold_probe: bad arguments");
  static u32x4_t* buf = nullptr;
  const size_t n16 = (size_t)64 << 20;                    // 1 GiB
  if (kind == 1 && !buf) {
    DD_HIP(hipMalloc((void**)&buf, n16 * 16));
    k_gload_fill<<<4096, 256, 0, st>>>(buf, n16);
    DD_CHECK_LAUNCH();
    DD_HIP(hipStreamSynchronize(st));
  }
  static uint32_t salt = 5;
  for (int i = 0; i < launches; ++i) {
    if (kind == 0) k_vgpr_hold<<<wgs, 256, 30720, st>>>(hold, salt++, errors_dev);
    else k_gload_hold<<<wgs, 256, 30720, st>>>(buf, n16, hold, salt++, errors_dev);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// Packed-FP32 probe: v_pk_fma_f32 chains (plain and with the broadcast op_sel forms the fp32 attention tile pass compiles to) against the same
// multiply-adds issued as scalar v_fma_f32, on identical operands, `iters` rounds per lane; 256-thread workgroups with 30,720 bytes of dynamic LDS
// (they fit beside a slice-resident GEMV workgroup).  errors_dev[0] += lanes whose packed and scalar results differ in any bit.
typedef float dd_f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk_probe(int iters, uint32_t salt, unsigned int* errors) {
  extern __shared__ __align__(16) uint32_t pl[];
  const uint32_t me = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u) ^ salt;
  float a0 = 0.5f + (float)(me & 1023u) * (1.0f / 1024.0f), a1 = 0.25f + (float)((me >> 10) & 1023u) * (1.0f / 2048.0f);
  float b0 = 1.0f - (float)((me >> 20) & 255u) * (1.0f / 512.0f), b1 = 0.75f + (float)((me >> 5) & 511u) * (1.0f / 4096.0f);
  dd_f32x2 accp = {0.f, 0.f}, accb = {0.f, 0.f};
  float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
  pl[threadIdx.x] = me;
  for (int it = 0; it < iters; ++it) {
    dd_f32x2 x = {a0, a1}, y = {b0, b1};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(accp) : "v"(x), "v"(y));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(accb) : "v"(x), "v"(y));     // second operand: its low half for both lanes
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(a0), "v"(b0));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(a1), "v"(b1));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t0) : "v"(a0), "v"(b0));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t1) : "v"(a1), "v"(b0));
    a0 = a0 * 0.999f + 0.001f, a1 = a1 * 1.0005f - 0.0003f, b0 = b0 * 0.9995f + 0.0004f, b1 = b1 * 1.0002f - 0.0001f;
    if ((it & 63) == 63) __syncthreads();
  }
  const bool bad = __float_as_uint(accp.x) != __float_as_uint(s0) || __float_as_uint(accp.y) != __float_as_uint(s1) ||
                   __float_as_uint(accb.x) != __float_as_uint(t0) || __float_as_uint(accb.y) != __float_as_uint(t1) || pl[threadIdx.x] != me;
  if (bad) atomicAdd(errors, 1u);
}
extern "C" int dd_tools_pk_probe(int launches, int wgs, int iters, unsigned int* errors_dev, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(launches >= 1 && wgs >= 1 && iters >= 1 && errors_dev, "dd_tools_pk_probe: bad arguments");
  static uint32_t salt = 17;
  for (int i = 0; i < launches; ++i) {
    k_pk_probe<<<wgs, 256, 30720, st>>>(iters, salt++, errors_dev);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// P.V probe: step 5 of the fp32-cache attention tile pass as it is written there — eight rows' accumulators += p (read from LDS, one 32-byte row
// per key) x V (sixteen bytes per lane and key, in registers); the compiler turns the vector form into v_pk_fma_f32 with broadcast op_sel forms fed
// by ds_read_b128 — next to the SAME sums issued as scalar v_fma_f32, compared bit for bit, `iters` rounds.  errors_dev[0] += lanes that differ.
__global__ __launch_bounds__(256) void k_pv_probe(int iters, uint32_t salt, unsigned int* errors) {
  extern __shared__ __align__(16) float pf[];
  float* p_sh = pf + 3072;                             // where the tile pass keeps its probabilities: [64 keys][8 rows]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const uint32_t me = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u) ^ salt;
  f32x4_t v4[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    v4[j] = (f32x4_t){0.5f + (float)((me >> j) & 255u) * (1.0f / 256.0f), -0.25f + (float)((me >> (j + 3)) & 127u) * (1.0f / 128.0f),
                      1.0f - (float)((me >> (j + 7)) & 63u) * (1.0f / 64.0f), 0.125f + (float)((me >> (j + 11)) & 31u) * (1.0f / 32.0f)};
  unsigned int bad = 0;
  for (int it = 0; it < iters; ++it) {
    for (int r = wave; r < 8; r += 4) p_sh[lane * 8 + r] = (float)((lane * 8 + r + it * 13 + (int)(salt & 31u)) & 511) * (1.0f / 512.0f);
    __syncthreads();
    f32x4_t acc[8], ref[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = ref[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = wave * 16 + 2 * j + half;
      const float* pr = &p_sh[key * 8];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] += pr[r] * v4[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = wave * 16 + 2 * j + half;
      const float* pr = &p_sh[key * 8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float pv = pr[r];
        asm volatile("" : "+v"(pv));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[r].x) : "v"(pv), "v"(v4[j].x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[r].y) : "v"(pv), "v"(v4[j].y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[r].z) : "v"(pv), "v"(v4[j].z));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[r].w) : "v"(pv), "v"(v4[j].w));
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
      bad += (__float_as_uint(acc[r].x) != __float_as_uint(ref[r].x)) | (__float_as_uint(acc[r].y) != __float_as_uint(ref[r].y)) |
             (__float_as_uint(acc[r].z) != __float_as_uint(ref[r].z)) | (__float_as_uint(acc[r].w) != __float_as_uint(ref[r].w));
#pragma unroll
    for (int j = 0; j < 8; ++j) v4[j] = v4[j] * 0.9995f + 0.0003f;
    __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
}
extern "C" int dd_tools_pv_probe(int launches, int wgs, int iters, unsigned int* errors_dev, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(launches >= 1 && wgs >= 1 && iters >= 1 && errors_dev, "dd_tools_pv_probe: bad arguments");
  static uint32_t salt = 23;
  for (int i = 0; i < launches; ++i) {
    k_pv_probe<<<wgs, 256, 30720, st>>>(iters, salt++, errors_dev);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// Packed-FP32 fault, the write-after-read reading (round 5).  In k_pv_probe's gfx950 code every failing group has the
// shape `v_pk_fma_f32 ..., v[40:41], ...` x 8 immediately followed by `ds_read_b128 v[40:43]` INTO THE SAME REGISTERS with no wait between: if a
// multi-pass packed op still has operand reads outstanding (its issue delayed by the MFMA neighbour's use of the SIMD) when a fast LDS return
// writes the registers, it reads the NEXT key's probabilities.  This probe fixes the registers by hand and runs three forms of the same loop on
// identical data (a wave's 16 keys, probabilities from LDS, V values in registers), each next to scalar v_fma_f32 sums of the same operands:
//   variant 0  the compiler's shape: the re-load overwrites v[40:43] right behind the packed ops that read them
//   variant 1  the same with `s_nop 7` x 2 between the last packed op and the re-load
//   variant 2  the re-load goes to the OTHER register quad (v[40:43] / v[44:47] alternate): no write-after-read on a packed op's operands
// errors_dev[variant] += lanes whose packed sums differ from the scalar sums.  A fault that is this hazard shows in variant 0 only.
template <int VARIANT>
__global__ __launch_bounds__(256) void k_pk_war_probe(int iters, uint32_t salt, unsigned int* errors) {
  extern __shared__ __align__(16) float pf[];
  float* p_sh = pf + 3072;                             // [64 keys][8 rows], as in k_pv_probe
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t me = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u) ^ salt;
  dd_f32x2 va = {0.5f + (float)(me & 255u) * (1.0f / 256.0f), -0.25f + (float)((me >> 8) & 127u) * (1.0f / 128.0f)};
  dd_f32x2 vb = {1.0f - (float)((me >> 15) & 63u) * (1.0f / 64.0f), 0.125f + (float)((me >> 21) & 31u) * (1.0f / 32.0f)};
  unsigned int bad = 0;
  for (int it = 0; it < iters; ++it) {
    for (int r = wave; r < 8; r += 4) p_sh[lane * 8 + r] = (float)((lane * 8 + r + it * 13 + (int)(salt & 31u)) & 511) * (1.0f / 512.0f);
    __syncthreads();
    // packed: a[2 r + w] += p[key][r] * (w ? vb : va) for r = 0..7 over the wave's 16 keys; a key's eight probabilities = two 16-byte units
    // (rows 0-3 / 4-7), eight packed ops per unit as in the tile pass
    dd_f32x2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (dd_f32x2){0.f, 0.f};
    const uint32_t addr = (uint32_t)(uintptr_t)(p_sh + wave * 16 * 8);      // LDS byte address of the wave's first key
    if constexpr (VARIANT == 0) {
      asm volatile(
          "ds_read_b128 v[40:43], %18\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:16\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:32\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:48\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:64\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:80\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:96\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:112\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:128\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:144\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:160\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:176\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:192\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:208\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:224\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:240\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:256\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:272\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:288\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:304\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:320\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:336\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:352\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:368\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:384\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:400\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:416\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:432\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:448\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:464\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:480\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "ds_read_b128 v[40:43], %18 offset:496\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
          : "v"(va), "v"(vb), "v"(addr)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory");
    } else if constexpr (VARIANT == 1) {
      asm volatile(
          "ds_read_b128 v[40:43], %18\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:16\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:32\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:48\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:64\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:80\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:96\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:112\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:128\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:144\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:160\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:176\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:192\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:208\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:224\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:240\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:256\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:272\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:288\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:304\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:320\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:336\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:352\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:368\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:384\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:400\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:416\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:432\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:448\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:464\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:480\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:496\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[40:41], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[40:41], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[40:41], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[40:41], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[42:43], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[42:43], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[42:43], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[42:43], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
          : "v"(va), "v"(vb), "v"(addr)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory");
    } else if constexpr (VARIANT == 2) {
      asm volatile(
          "ds_read_b128 v[40:43], %18\n"
          "ds_read_b128 v[44:47], %18 offset:16\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[48:51], %18 offset:32\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[52:55], %18 offset:48\n"
          "v_pk_fma_f32 %8, v[44:45], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[44:45], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[44:45], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[44:45], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[46:47], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[46:47], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[46:47], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[46:47], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[56:59], %18 offset:64\n"
          "v_pk_fma_f32 %0, v[48:49], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[48:49], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[48:49], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[48:49], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[50:51], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[50:51], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[50:51], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[50:51], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[60:63], %18 offset:80\n"
          "v_pk_fma_f32 %8, v[52:53], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[52:53], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[52:53], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[52:53], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[54:55], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[54:55], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[54:55], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[54:55], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[40:43], %18 offset:96\n"
          "v_pk_fma_f32 %0, v[56:57], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[56:57], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[56:57], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[56:57], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[58:59], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[58:59], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[58:59], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[58:59], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[44:47], %18 offset:112\n"
          "v_pk_fma_f32 %8, v[60:61], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[60:61], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[60:61], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[60:61], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[62:63], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[62:63], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[62:63], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[62:63], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[48:51], %18 offset:128\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[52:55], %18 offset:144\n"
          "v_pk_fma_f32 %8, v[44:45], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[44:45], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[44:45], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[44:45], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[46:47], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[46:47], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[46:47], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[46:47], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[56:59], %18 offset:160\n"
          "v_pk_fma_f32 %0, v[48:49], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[48:49], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[48:49], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[48:49], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[50:51], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[50:51], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[50:51], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[50:51], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[60:63], %18 offset:176\n"
          "v_pk_fma_f32 %8, v[52:53], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[52:53], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[52:53], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[52:53], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[54:55], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[54:55], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[54:55], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[54:55], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[40:43], %18 offset:192\n"
          "v_pk_fma_f32 %0, v[56:57], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[56:57], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[56:57], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[56:57], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[58:59], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[58:59], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[58:59], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[58:59], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[44:47], %18 offset:208\n"
          "v_pk_fma_f32 %8, v[60:61], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[60:61], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[60:61], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[60:61], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[62:63], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[62:63], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[62:63], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[62:63], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[48:51], %18 offset:224\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[52:55], %18 offset:240\n"
          "v_pk_fma_f32 %8, v[44:45], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[44:45], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[44:45], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[44:45], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[46:47], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[46:47], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[46:47], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[46:47], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[56:59], %18 offset:256\n"
          "v_pk_fma_f32 %0, v[48:49], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[48:49], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[48:49], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[48:49], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[50:51], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[50:51], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[50:51], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[50:51], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[60:63], %18 offset:272\n"
          "v_pk_fma_f32 %8, v[52:53], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[52:53], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[52:53], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[52:53], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[54:55], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[54:55], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[54:55], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[54:55], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[40:43], %18 offset:288\n"
          "v_pk_fma_f32 %0, v[56:57], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[56:57], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[56:57], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[56:57], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[58:59], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[58:59], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[58:59], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[58:59], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[44:47], %18 offset:304\n"
          "v_pk_fma_f32 %8, v[60:61], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[60:61], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[60:61], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[60:61], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[62:63], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[62:63], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[62:63], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[62:63], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[48:51], %18 offset:320\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[52:55], %18 offset:336\n"
          "v_pk_fma_f32 %8, v[44:45], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[44:45], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[44:45], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[44:45], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[46:47], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[46:47], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[46:47], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[46:47], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[56:59], %18 offset:352\n"
          "v_pk_fma_f32 %0, v[48:49], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[48:49], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[48:49], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[48:49], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[50:51], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[50:51], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[50:51], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[50:51], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[60:63], %18 offset:368\n"
          "v_pk_fma_f32 %8, v[52:53], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[52:53], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[52:53], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[52:53], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[54:55], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[54:55], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[54:55], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[54:55], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[40:43], %18 offset:384\n"
          "v_pk_fma_f32 %0, v[56:57], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[56:57], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[56:57], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[56:57], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[58:59], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[58:59], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[58:59], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[58:59], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[44:47], %18 offset:400\n"
          "v_pk_fma_f32 %8, v[60:61], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[60:61], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[60:61], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[60:61], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[62:63], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[62:63], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[62:63], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[62:63], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[48:51], %18 offset:416\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[52:55], %18 offset:432\n"
          "v_pk_fma_f32 %8, v[44:45], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[44:45], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[44:45], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[44:45], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[46:47], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[46:47], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[46:47], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[46:47], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[56:59], %18 offset:448\n"
          "v_pk_fma_f32 %0, v[48:49], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[48:49], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[48:49], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[48:49], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[50:51], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[50:51], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[50:51], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[50:51], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[60:63], %18 offset:464\n"
          "v_pk_fma_f32 %8, v[52:53], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[52:53], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[52:53], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[52:53], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[54:55], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[54:55], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[54:55], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[54:55], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[40:43], %18 offset:480\n"
          "v_pk_fma_f32 %0, v[56:57], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[56:57], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[56:57], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[56:57], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[58:59], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[58:59], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[58:59], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[58:59], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "ds_read_b128 v[44:47], %18 offset:496\n"
          "v_pk_fma_f32 %8, v[60:61], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[60:61], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[60:61], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[60:61], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[62:63], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[62:63], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[62:63], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[62:63], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(1)\n"
          "v_pk_fma_f32 %0, v[40:41], %16, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, v[40:41], %17, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %2, v[40:41], %16, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %3, v[40:41], %17, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, v[42:43], %16, %4 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %5, v[42:43], %17, %5 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %6, v[42:43], %16, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %7, v[42:43], %17, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, v[44:45], %16, %8 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %9, v[44:45], %17, %9 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %10, v[44:45], %16, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %11, v[44:45], %17, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %12, v[46:47], %16, %12 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %13, v[46:47], %17, %13 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %14, v[46:47], %16, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %15, v[46:47], %17, %15 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
          : "v"(va), "v"(vb), "v"(addr)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory");
    } else if constexpr (VARIANT == 3) {
      asm volatile(
          "ds_read_b128 v[40:43], %18\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:16\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:32\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:48\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:64\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:80\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:96\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:112\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:128\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:144\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:160\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:176\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:192\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:208\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:224\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:240\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:256\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:272\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:288\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:304\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:320\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:336\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:352\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:368\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:384\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:400\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:416\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:432\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:448\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:464\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:480\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "ds_read_b128 v[40:43], %18 offset:496\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
          : "v"(va), "v"(vb), "v"(addr)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory");
    } else if constexpr (VARIANT == 4) {
      asm volatile(
          "ds_read_b128 v[40:43], %18\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:16\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:32\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:48\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:64\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:80\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:96\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:112\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:128\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:144\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:160\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:176\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:192\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:208\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:224\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:240\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:256\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:272\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:288\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:304\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:320\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:336\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:352\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:368\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:384\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:400\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:416\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:432\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:448\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:464\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:480\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[40:41], %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %7, %17, v[40:41], %7 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          "ds_read_b128 v[40:43], %18 offset:496\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_mov_b32 v40, v43\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[40:41], %14 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %15, %17, v[40:41], %15 op_sel_hi:[1,0,1]\n"
          "s_nop 7\n"
          "s_nop 7\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
          : "v"(va), "v"(vb), "v"(addr)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory");
    } else {
      asm volatile(
          "ds_read_b128 v[40:43], %18\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:16\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:32\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:48\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:64\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:80\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:96\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:112\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:128\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:144\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:160\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:176\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:192\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:208\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:224\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:240\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:256\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:272\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:288\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:304\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:320\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:336\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:352\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:368\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:384\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:400\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:416\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:432\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:448\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:464\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:480\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %0, %16, v[40:41], %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %1, %17, v[40:41], %1 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %16, v[40:41], %2 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %3, %17, v[40:41], %3 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %4, %16, v[42:43], %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %5, %17, v[42:43], %5 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %16, v[42:43], %6 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %7, %17, v[42:43], %7 op_sel:[0,1,0]\n"
          "ds_read_b128 v[40:43], %18 offset:496\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_pk_fma_f32 %8, %16, v[40:41], %8 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %9, %17, v[40:41], %9 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %10, %16, v[40:41], %10 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %11, %17, v[40:41], %11 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %12, %16, v[42:43], %12 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %13, %17, v[42:43], %13 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %14, %16, v[42:43], %14 op_sel:[0,1,0]\n"
          "v_pk_fma_f32 %15, %17, v[42:43], %15 op_sel:[0,1,0]\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
          : "v"(va), "v"(vb), "v"(addr)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory");
    }
    // scalar reference: the same sums, same order per accumulator
    float ref[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) ref[i][0] = ref[i][1] = 0.f;
#pragma unroll 2
    for (int key = 0; key < 16; ++key) {
      const float* pr = &p_sh[(wave * 16 + key) * 8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float pv = pr[r];
        asm volatile("" : "+v"(pv));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * r][0]) : "v"(pv), "v"(va.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * r][1]) : "v"(pv), "v"(va.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * r + 1][0]) : "v"(pv), "v"(vb.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * r + 1][1]) : "v"(pv), "v"(vb.y));
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      bad += (__float_as_uint(a[i].x) != __float_as_uint(ref[i][0])) | (__float_as_uint(a[i].y) != __float_as_uint(ref[i][1]));
    va = va * 0.9995f + 0.0003f;
    vb = vb * 1.0002f - 0.0001f;
    __syncthreads();
  }
  if (bad) atomicAdd(errors + VARIANT, bad);
}
// errors_dev[0..5] += (lane, row) results of variants 0..5 (tools/gen_pk_war_probe.py: 0-2 the LDS-fed pair as src0; 3-5 hipcc's own group
// replicated: as compiled / with sixteen wait states before the re-load / without its v_mov) that differ from the scalar sums.
extern "C" int dd_tools_pk_war_probe(int launches, int wgs, int iters, unsigned int* errors_dev, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(launches >= 1 && wgs >= 1 && iters >= 1 && errors_dev, "dd_tools_pk_war_probe: bad arguments");
  static uint32_t salt = 31;
  for (int i = 0; i < launches; ++i) {
    k_pk_war_probe<0><<<wgs, 256, 30720, st>>>(iters, salt, errors_dev);
    k_pk_war_probe<1><<<wgs, 256, 30720, st>>>(iters, salt, errors_dev);
    k_pk_war_probe<2><<<wgs, 256, 30720, st>>>(iters, salt, errors_dev);
    k_pk_war_probe<3><<<wgs, 256, 30720, st>>>(iters, salt, errors_dev);
    k_pk_war_probe<4><<<wgs, 256, 30720, st>>>(iters, salt, errors_dev);
    k_pk_war_probe<5><<<wgs, 256, 30720, st>>>(iters, salt++, errors_dev);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// The same question for operands returned by GLOBAL loads (the finishing kernel of the slice GEMVs adds eight 16-byte partial-sum loads per thread
// with v_pk_add_f32): packed sums next to the same sums as scalar v_add_f32, bit for bit.  errors_dev[0] += lanes that differ.
__global__ __launch_bounds__(256) void k_pkadd_gload_probe(const f32x4_t* __restrict__ buf, size_t n16, int iters, uint32_t salt, unsigned int* errors) {
  extern __shared__ __align__(16) uint32_t pl[];
  pl[threadIdx.x] = salt;
  unsigned int bad = 0;
  size_t base = ((size_t)(blockIdx.x * 256u + threadIdx.x) * 2654435761u + salt) % n16;
  for (int it = 0; it < iters; ++it) {
    f32x4_t v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = buf[(base + (size_t)q * 4099u * 64u) % n16];
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; q += 2) acc = acc + (v[q] + v[q + 1]);          // the finishing kernel's pairwise order
    float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      float t0 = v[q].x, t1 = v[q].y, t2 = v[q].z, t3 = v[q].w;
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(t0) : "v"(v[q + 1].x));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(t1) : "v"(v[q + 1].y));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(t2) : "v"(v[q + 1].z));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(t3) : "v"(v[q + 1].w));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[0]) : "v"(t0));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[1]) : "v"(t1));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[2]) : "v"(t2));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[3]) : "v"(t3));
    }
    bad += (__float_as_uint(acc.x) != __float_as_uint(r[0])) | (__float_as_uint(acc.y) != __float_as_uint(r[1])) |
           (__float_as_uint(acc.z) != __float_as_uint(r[2])) | (__float_as_uint(acc.w) != __float_as_uint(r[3]));
    base = (base + 7919u * 64u) % n16;
  }
  if (bad) atomicAdd(errors, bad);
}
__global__ void k_fill_floats(float* buf, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    buf[i] = (float)((uint32_t)(i * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
}
extern "C" int dd_tools_pkadd_gload_probe(int launches, int wgs, int iters, unsigned int* errors_dev, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(launches >= 1 && wgs >= 1 && iters >= 1 && errors_dev, "dd_tools_pkadd_gload_probe: bad arguments");
  static f32x4_t* buf = nullptr;
  const size_t n16 = (size_t)16 << 20;                    // 256 MiB of floats
  if (!buf) {
    DD_HIP(hipMalloc((void**)&buf, n16 * 16));
    k_fill_floats<<<4096, 256, 0, st>>>((float*)buf, n16 * 4);
    DD_CHECK_LAUNCH();
    DD_HIP(hipStreamSynchronize(st));
  }
  static uint32_t salt = 29;
  for (int i = 0; i < launches; ++i) {
    k_pkadd_gload_probe<<<wgs, 256, 4096, st>>>(buf, n16, iters, salt++, errors_dev);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// ----------------------------------------------------------------------------------------------
// The sampler's co-residency fault, shrunk to its victim phase (second half; written at the end of round 4, to be RUN by the next
// one: the round's GPU minutes were spent).  A workgroup of the sampler's shape — 1,024 threads, `lds_bytes` of dynamic LDS with the 624-word
// generator block at the sampler's offset — does nothing but the mt19937 regeneration of dd_dropout.hip mt_twist_block (three block-parallel
// sweeps and the last word, each `barrier; read three words; barrier; write`) `iters` times over, and after each regeneration wave 0 ALONE recomputes
// the same 624 words from a copy of the old state, in order, 64 at a time, with no block barrier (a wave's LDS operations execute in order), and
// all threads compare.  out[0] counts differing words, out[1..7] hold the first difference (workgroup, iteration, word, got, want, and the words
// at index - 1 / + 1 as computed by the sweeps).  Run it on a stream of its own beside a group taking rider steps (tools/sampler_repro.py
// twist_probe): differences there and none alone would put the fault between the barriers and the LDS of this phase, whatever else the sampler does.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tw_mix(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__global__ __launch_bounds__(1024) void k_twist_probe(uint32_t seed, int iters, int mt_word_offset, unsigned int* out) {
  extern __shared__ __align__(16) uint32_t tw_lds[];
  uint32_t* mt = tw_lds + mt_word_offset;       // the block under test, where the sampler keeps it
  uint32_t* old_s = tw_lds;                     // [624] copy of the state before the regeneration
  uint32_t* chk = tw_lds + 640;                 // [624] wave 0's in-order recomputation
  const int tid = threadIdx.x;
  for (int i = tid; i < 624; i += 1024) {
    uint32_t x = seed ^ (uint32_t)(blockIdx.x * 0x9e3779b9u) ^ (uint32_t)(i * 0x85ebca6bu);
    x ^= x >> 15, x *= 0x2c1b3c6du, x ^= x >> 12, x *= 0x297a2d39u, x ^= x >> 15;
    mt[i] = x;
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int i = tid; i < 624; i += 1024) old_s[i] = mt[i];
    // --- the phase under test: mt_twist_block as the sampler runs it ---
    const int segs[4][2] = {{0, 227}, {227, 454}, {454, 623}, {623, 624}};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      __syncthreads();
      uint32_t nv = 0;
      const int i = segs[s][0] + tid;
      const bool act = i < segs[s][1];
      if (act) nv = tw_mix(mt[i], mt[(i + 1) % 624], mt[(i + 397) % 624]);
      __syncthreads();
      if (act) mt[i] = nv;
    }
    __syncthreads();
    // --- wave 0's recomputation, in order ---
    if (tid < 64) {
      for (int b = 0; b < 624; b += 64) {
        const int i = b + tid;
        if (i < 623) {
          const uint32_t c = i < 227 ? old_s[i + 397] : chk[i - 227];
          chk[i] = tw_mix(old_s[i], old_s[i + 1], c);
        }
        // the next 64 words read what OTHER lanes of this wave just wrote: keep the compiler from moving those loads above these stores
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (tid == 0) chk[623] = tw_mix(old_s[623], chk[0], chk[396]);
    }
    __syncthreads();
    for (int i = tid; i < 624; i += 1024) {
      if (mt[i] != chk[i]) {
        if (atomicAdd(&out[0], 1u) == 0u) {
          out[1] = blockIdx.x, out[2] = (unsigned)it, out[3] = (unsigned)i, out[4] = mt[i], out[5] = chk[i];
          out[6] = mt[(i + 623) % 624], out[7] = mt[(i + 1) % 624];
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < 624; i += 1024) mt[i] = chk[i];      // continue from the reference state
    __syncthreads();
  }
}
// Barrier probe (round 5): what round 5's dumps of the 1,024-thread sampler show — waves of one workgroup out of step across s_barrier —
// looked for directly.  Workgroups of NT threads with `lds_bytes` of dynamic LDS; per iteration every thread stores the iteration number to its own
// word of a block at the sampler's generator offset, s_barrier, every thread reads the words of ALL other waves' lanes with its lane number and
// compares, s_barrier; between iterations a wave sleeps a wave- and iteration-dependent few hundred cycles and stores a byte to global memory, so that
// the waves reach the barrier at different times, as the sampler's do.  out[0] += words read that were not this iteration's; out[1] += of those, words
// one iteration OLD (the writer's store had not landed, or the reader passed the barrier early); out[2] += words of a LATER iteration (the writer
// passed two barriers the reader had not); out[3..7] = first event: workgroup, iteration, reader wave, writer wave, value read.
template <int NT>
__global__ __launch_bounds__(NT) void k_barrier_probe(int iters, int mt_word_offset, uint32_t salt, unsigned int* out, unsigned char* sink) {
  extern __shared__ __align__(16) uint32_t bp_lds[];
  volatile uint32_t* blk = bp_lds + mt_word_offset;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NT / 64;
  blk[tid] = 0;
  __syncthreads();
  for (int it = 1; it <= iters; ++it) {
    blk[tid] = (uint32_t)it;
    __syncthreads();
    unsigned bad = 0, old1 = 0, ahead = 0;
    int w_bad = -1;
    uint32_t v_bad = 0;
#pragma unroll 4
    for (int w = 1; w < NW; ++w) {
      const int ww = (wave + w) % NW;
      const uint32_t v = blk[ww * 64 + lane];
      if (v != (uint32_t)it) {
        bad++, old1 += v == (uint32_t)(it - 1), ahead += v > (uint32_t)it;
        if (w_bad < 0) w_bad = ww, v_bad = v;
      }
    }
    if (bad) {
      atomicAdd(&out[1], old1);
      atomicAdd(&out[2], ahead);
      if (atomicAdd(&out[0], bad) == 0u) out[3] = blockIdx.x, out[4] = (unsigned)it, out[5] = (unsigned)wave, out[6] = (unsigned)w_bad, out[7] = v_bad;
    }
    __syncthreads();
    const int nap = (int)(((uint32_t)wave * 2654435761u + (uint32_t)it * 40503u + salt) >> 28);      // 0..15
    for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(8);
    sink[(size_t)blockIdx.x * NT + tid] = (unsigned char)it;
  }
}
extern "C" int dd_tools_barrier_probe(int launches, int wgs, int threads, int iters, int lds_bytes, unsigned int* out_dev, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  const int mt_off = (8192 * 4 * 2 + 8192) / 4;
  DD_REQUIRE(launches >= 1 && wgs >= 1 && wgs <= 4096 && iters >= 1 && out_dev && (threads == 1024 || threads == 512 || threads == 256) &&
                 lds_bytes >= (mt_off + 1024) * 4 && lds_bytes <= 156 * 1024,
             "dd_tools_barrier_probe: bad arguments (threads 256 / 512 / 1024, lds_bytes %d .. %d)", (mt_off + 1024) * 4, 156 * 1024);
  static unsigned char* sink = nullptr;
  static bool attr = false;
  if (!attr) {
    DD_HIP(hipMalloc((void**)&sink, (size_t)4096 * 1024));
    DD_HIP(hipFuncSetAttribute((const void*)k_barrier_probe<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    DD_HIP(hipFuncSetAttribute((const void*)k_barrier_probe<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    DD_HIP(hipFuncSetAttribute((const void*)k_barrier_probe<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    attr = true;
  }
  static uint32_t salt = 1;
  for (int i = 0; i < launches; ++i) {
    if (threads == 1024) k_barrier_probe<1024><<<wgs, 1024, lds_bytes, st>>>(iters, mt_off, salt++, out_dev, sink);
    else if (threads == 512) k_barrier_probe<512><<<wgs, 512, lds_bytes, st>>>(iters, mt_off, salt++, out_dev, sink);
    else k_barrier_probe<256><<<wgs, 256, lds_bytes, st>>>(iters, mt_off, salt++, out_dev, sink);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}
extern "C" int dd_tools_twist_probe(int launches, int wgs, int iters, int lds_bytes, unsigned int* out_dev, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  const int mt_off = (8192 * 4 * 2 + 8192) / 4;                  // the sampler's layout: after its two float arrays and its byte array
  DD_REQUIRE(launches >= 1 && wgs >= 1 && iters >= 1 && out_dev && lds_bytes >= (mt_off + 624 + 8) * 4 && lds_bytes <= 156 * 1024,
             "dd_tools_twist_probe: bad arguments (lds_bytes %d .. %d)", (mt_off + 632) * 4, 156 * 1024);
  static bool attr = false;
  if (!attr) {
    DD_HIP(hipFuncSetAttribute((const void*)k_twist_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    attr = true;
  }
  static uint32_t salt = 17;
  for (int i = 0; i < launches; ++i) {
    k_twist_probe<<<wgs, 1024, lds_bytes, st>>>(salt++, iters, mt_off, out_dev);
    DD_CHECK_LAUNCH();
  }
  return DD_OK;
}

// ---- round 6: a stream that runs on a subset of the CUs, and a probe of where its workgroups land (tools/cu_partition_lab.py) ------------
// hipExtStreamCreateWithCUMask: bit i of the 256-bit mask enables CU i in the driver's numbering; how that numbering maps onto the eight
// XCDs is not documented here, so the probe reports what a mask really gives: every workgroup writes its XCC_ID and HW_ID and then holds
// its CU for `hold` rounds of s_sleep, so that a grid of a few workgroups per CU spreads over every CU the stream may use.
extern "C" int dd_tools_stream_create_cu_mask(const uint32_t* mask, int words, void** stream_out) {
  DD_REQUIRE(mask && words >= 1 && words <= 8 && stream_out, "dd_tools_stream_create_cu_mask: bad arguments");
  hipStream_t st = nullptr;
  DD_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
  *stream_out = (void*)st;
  return DD_OK;
}
extern "C" int dd_tools_stream_destroy(void* stream_) {
  DD_HIP(hipStreamDestroy((hipStream_t)stream_));
  return DD_OK;
}
__global__ __launch_bounds__(64) void k_cu_probe(uint32_t* out, int hold) {
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));       // XCC_ID
    out[2 * blockIdx.x + 1] = (uint32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID: CU / SH / SE ids
  }
  for (int i = 0; i < hold; ++i) __builtin_amdgcn_s_sleep(64);
}
extern "C" int dd_tools_cu_probe(uint32_t* out_dev, int wgs, int hold, void* stream_) {
  DD_REQUIRE(out_dev && wgs >= 1 && hold >= 0, "dd_tools_cu_probe: bad arguments");
  k_cu_probe<<<wgs, 64, 0, (hipStream_t)stream_>>>(out_dev, hold);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

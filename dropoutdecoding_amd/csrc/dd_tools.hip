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
// between the stages on the caller's stream); the product switches
// (8, 11, 13-16) are forwarded to dd_set_tuning.  Every call bumps the graph-key epoch: steps captured under other settings are not replayed.
extern int g_exp_G[4];
extern int g_attn16_tpw, g_attn16_full, g_finish4;
void dd_engine_set_pairs(int on);
void dd_engine_set_branches(int n);
void dd_engine_set_rider(int on);
void dd_engine_set_ride_beside(int on);
void dd_engine_set_rider_branches(int n);
void dd_engine_set_half_planes(int on);
void dd_engine_set_half_planes_first(int on);
void dd_engine_set_rider_staged(int on);
extern int g_exp_U9;
extern "C" int dd_tools_set_tuning(int key, int value) {
  dd_engine_bump_epoch();
  if (key == 8 || key == 11 || (key >= 13 && key <= 16)) return dd_set_tuning(key, value);
  DD_REQUIRE(key == 0 || key == 1 || key == 2 || key == 4 || key == 9 || key == 10 || key == 12 || (key >= 17 && key <= 19) || (key >= 21 && key <= 24) || (key >= 26 && key <= 31) || key == 33,
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


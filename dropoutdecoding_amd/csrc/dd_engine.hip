// The LM engine behind dd_lm_*: owns packed weights, the shared read-only prefix KV cache, per-member new-row
// scratch, and enqueues a whole ensemble decode step (un-masked pass -> keep set -> masks -> packed K-member
// sweep -> vote -> commit) on one HIP stream with no host synchronisation.
// Replaces the reference's decode branch of forward(): models/llava.py:254-376 (see include/dropdec.h).
#include <math.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

#include "dd_lm_kernels.h"

// `gate` arguments: &state->done of the sequence — a finished sequence's look-ahead steps leave rng, masks, keep set, argmax
// and vote untouched (HF stops at EOS; here the stop is device-side because steps are enqueued ahead)
int dd_overlap_keep_from_argmax(const int32_t* argmax_dev, const int32_t* topk_ids, int L, int k, uint8_t* keep,
                                const int32_t* gate, hipStream_t st);
int dd_sample_masks_impl(const float* epi, int L, const double* mprobs, int K, const uint8_t* keep, int mode,
                         int rng_mode, const float* uniforms, uint32_t* rng_state, uint8_t* drop, int32_t* n_drop,
                         int32_t* idx, uint8_t* drop_bits, const int32_t* gate, hipStream_t st, const uint32_t* rng_in = nullptr,
                         bool empty_keep = false);
int dd_kl_keep_impl(const float* step_logits, const float* image_logits, int L, int V, int ld, float percent, uint8_t* keep,
                    float* kl_ws, const int32_t* gate, hipStream_t st);
int dd_spec_check(const uint8_t* keep, const uint8_t* drop_bits, int L, int K, const int32_t* done, int32_t* ok_out,
                  int keep_matters, hipStream_t st, int32_t* host_note = nullptr);
int dd_copy_row_gated(const float* src, float* dst, int n, const int32_t* gate, hipStream_t st);
int dd_argmax_rows_gated(const float* x, int R, int V, int ld, int32_t* out, const int32_t* gate, hipStream_t st);
int dd_vote_gated(const int32_t* ids, int K, int32_t* out2, const int32_t* gate, hipStream_t st);
uint32_t* dd_rng_state_ptr(dd_rng* r);
unsigned long long dd_rng_serial(dd_rng* r);
int dd_argmax_rows_lanes(const float* const* x, int32_t* const* out, const int32_t* const* gates, int n, int R, int V, int ld,
                         hipStream_t st);
int dd_vote_lanes(const int32_t* const* ids, int32_t* const* out2, const int32_t* const* gates, int n, int K, hipStream_t st);
static unsigned long long g_lm_serial = 0;   // handles are identified in graph keys by a serial that is never reused

#include "dd_engine_internal.h"

template <typename T>
static int dalloc(dd_lm* h, T** p, size_t n) {
  void* q = nullptr;
  size_t b = n * sizeof(T);
  if (b == 0) b = 16;
  hipError_t e = hipMalloc(&q, b);
  if (e != hipSuccess) {
    dd_set_error("hipMalloc(%zu bytes) -> %s", b, hipGetErrorString(e));
    return DD_ENOMEM;
  }
  (void)hipMemset(q, 0, b);
  h->allocs.push_back(q);
  h->bytes += b;
  *p = (T*)q;
  return DD_OK;
}
#define DA(ptr, n)                          \
  do {                                      \
    int rc__ = dalloc(h, &(ptr), (size_t)(n)); \
    if (rc__ != DD_OK) {                    \
      dd_lm_destroy(h);                     \
      return rc__;                          \
    }                                       \
  } while (0)
#define RC(expr)              \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != DD_OK) return rc__; \
  } while (0)

void dd_engine_bump_epoch();
// member_logits for K > 16: one re-allocation to DD_MAX_MEMBERS rows.  Steps already enqueued may still read the old buffer, so the
// device is drained first and the old buffer stays in the handle's list until destroy (2 MB); captured graphs hold the old pointer:
// the tuning epoch moves, every cached graph of the process is re-captured on its next use.
static int ensure_member_rows(dd_lm* h, int K) {
  if (!h || K <= h->member_rows_cap) return DD_OK;
  DD_REQUIRE(K <= MAX_MEMBERS, "K=%d members (1..%d)", K, MAX_MEMBERS);
  DD_HIP(hipDeviceSynchronize());
  float* p = nullptr;
  RC(dalloc(h, &p, (size_t)MAX_MEMBERS * h->Vpad));
  h->member_logits = p;
  h->member_rows_cap = MAX_MEMBERS;
  dd_engine_bump_epoch();
  return DD_OK;
}

extern "C" int dd_lm_destroy(dd_lm* h) {
  if (!h) return DD_OK;
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (int i = 0; i < 3; ++i) {
    if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    if (h->side[i]) (void)hipStreamDestroy(h->side[i]);
  }
  if (h->tok_host) (void)hipHostFree(h->tok_host);
  for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
  delete h;
  return DD_OK;
}

extern "C" size_t dd_lm_device_bytes(const dd_lm* h) { return h ? h->bytes : 0; }

static int lm_create_impl(const dd_lm_config* c, dd_lm* parent, dd_lm** out) {
  DD_REQUIRE(c && out, "dd_lm_create: null argument");
  DD_REQUIRE(c->head_dim == 128, "dd_lm_create: head_dim must be 128 (got %d)", c->head_dim);
  DD_REQUIRE(c->hidden_size % 256 == 0 && c->intermediate_size % 256 == 0,
             "dd_lm_create: hidden (%d) and intermediate (%d) sizes must be multiples of 256", c->hidden_size,
             c->intermediate_size);
  DD_REQUIRE(c->num_heads % c->num_kv_heads == 0, "dd_lm_create: heads %% kv_heads != 0");
  int G = c->num_heads / c->num_kv_heads;
  DD_REQUIRE(G == 1 || G == 2 || G == 4, "dd_lm_create: GQA group %d unsupported", G);
  DD_REQUIRE((c->num_heads * 128) % 256 == 0, "dd_lm_create: num_heads*128 must be a multiple of 256");
  DD_REQUIRE(c->max_seq >= 2 && c->max_visual >= 1 && c->max_visual <= 8192, "dd_lm_create: bad max_seq/max_visual");
  DD_REQUIRE(c->max_seq <= 160 * 64, "dd_lm_create: max_seq %d exceeds the decode attention's %d key tiles of 64", c->max_seq, 160);
  DD_REQUIRE(c->k_top >= 1 && c->k_top <= DD_MAX_TOPK, "dd_lm_create: k_top out of range");
  DD_REQUIRE(c->mask_mode >= 0 && c->mask_mode <= 5, "dd_lm_create: mask_mode");
  DD_REQUIRE(c->vote_on >= 0 && c->vote_on <= 2, "dd_lm_create: vote_on");
  dd_lm* h = new dd_lm();
  h->serial = ++g_lm_serial;
  h->cfg = *c;
  h->d = c->hidden_size, h->dff = c->intermediate_size, h->V = c->vocab_size, h->Vpad = (c->vocab_size + 15) / 16 * 16;
  h->H = c->num_heads, h->Hkv = c->num_kv_heads, h->q_dim = h->H * 128, h->kv_dim = h->Hkv * 128;
  h->Lyr = c->num_layers, h->T_cap = (c->max_seq + 63) / 64 * 64, h->Lmax = c->max_visual;
  h->S_d = h->d / 32, h->S_q = h->q_dim / 32, h->S_ff = h->dff / 32;
  h->q_tiles = h->q_dim / 16, h->k_tiles = h->kv_dim / 16, h->qkv_tiles = (h->q_dim + 2 * h->kv_dim) / 16;
  const int d = h->d, dff = h->dff, T = h->T_cap;
  h->tp_world = c->reserved[0] > 1 ? c->reserved[0] : 1;
  h->tp_rank = h->tp_world > 1 ? c->reserved[1] : 0;
  if (h->tp_world > 1 && (h->tp_rank < 0 || h->tp_rank >= h->tp_world || h->tp_world > 8 || c->weight_format == 1 || parent)) {
    dd_set_error("dd_lm_create: tensor-parallel shard %d of %d (2..8 ranks, bf16 / fp16 weights, not a lane)", c->reserved[1], c->reserved[0]);
    delete h;
    return DD_EINVAL;
  }
  h->fp8 = c->weight_format == 1 ? 1 : 0;
  h->wf = c->weight_format == 2 ? 1 : 0;
  h->kv16 = c->kv_format == 1 ? 1 : 0;
  DD_REQUIRE(c->kv_format == 0 || c->kv_format == 1, "dd_lm_create: unknown KV cache format %d", c->kv_format);
  DD_REQUIRE(c->weight_format >= 0 && c->weight_format <= 2, "dd_lm_create: unknown weight format %d", c->weight_format);
  const int wdiv = h->fp8 ? 2 : 1;   // u32x4 units per tile row: S*64 (bf16) or (S/2)*64 (fp8)
  if (parent) {
    // a lane: another sequence over the SAME weights (its own KV cache, state and scratch)
    const dd_lm_config& pc = parent->cfg;
    if (!(pc.vocab_size == c->vocab_size && pc.hidden_size == c->hidden_size && pc.intermediate_size == c->intermediate_size &&
          pc.num_layers == c->num_layers && pc.num_heads == c->num_heads && pc.num_kv_heads == c->num_kv_heads &&
          pc.weight_format == c->weight_format && pc.rope_theta == c->rope_theta && parent->wsrc == nullptr)) {
      dd_set_error("dd_lm_create_shared: the lane's model dimensions differ from the weight owner's (or the owner is itself a lane)");
      delete h;
      return DD_EINVAL;
    }
    h->wsrc = parent;
    h->lw = parent->lw;
    h->lm_head = parent->lm_head, h->s_lm = parent->s_lm, h->deq_tmp = parent->deq_tmp;
    h->final_norm = parent->final_norm, h->embed = parent->embed;
  } else {
  h->lw.resize(h->Lyr);
  for (int l = 0; l < h->Lyr; ++l) {
    LayerW& w = h->lw[l];
    DA(w.wqkv, (size_t)h->qkv_tiles * h->S_d * 64 / wdiv);
    DA(w.wo, (size_t)(d / 16) * h->S_q * 64 / wdiv);
    DA(w.wgu, (size_t)(2 * dff / 16) * h->S_d * 64 / wdiv);
    DA(w.wdown, (size_t)(d / 16) * h->S_ff * 64 / wdiv);
    if (h->fp8) {
      DA(w.s_qkv, (size_t)h->qkv_tiles * 16);
      DA(w.s_o, (size_t)d);
      DA(w.s_gu, (size_t)2 * dff);
      DA(w.s_down, (size_t)d);
    }
    DA(w.norm1, d);
    DA(w.norm2, d);
  }
  DA(h->lm_head, (size_t)(h->Vpad / 16) * h->S_d * 64 / wdiv);
  if (h->fp8) {
    DA(h->s_lm, (size_t)h->Vpad);
    size_t mx = (size_t)h->qkv_tiles * h->S_d;
    if ((size_t)(2 * dff / 16) * h->S_d > mx) mx = (size_t)(2 * dff / 16) * h->S_d;
    if ((size_t)(d / 16) * h->S_ff > mx) mx = (size_t)(d / 16) * h->S_ff;
    if ((size_t)(h->Vpad / 16) * h->S_d > mx) mx = (size_t)(h->Vpad / 16) * h->S_d;
    DA(h->deq_tmp, mx * 64);
  }
  DA(h->final_norm, d);
  DA(h->embed, (size_t)h->V * d);
  }
  DA(h->rope_cos, (size_t)T * 64);
  DA(h->rope_sin, (size_t)T * 64);
  h->lsk = (size_t)h->Hkv * 32 * T * 4 / (h->kv16 ? 2 : 1);     // layer strides in floats of the storage (fp16: half)
  h->lsv = (size_t)h->Hkv * T * 128 / (h->kv16 ? 2 : 1);
  DA(h->kc, h->lsk * h->Lyr);
  DA(h->vc, h->lsv * h->Lyr);
  // decode scratch
  DA(h->xa, GROUP_ROWS * (size_t)d);         // up to 64 rows: a member pass of eight sequences (dd_lm_group_step)
  DA(h->qbuf, GROUP_ROWS * (size_t)h->q_dim);
  DA(h->knew, (size_t)h->Lyr * KV_ROWS * h->kv_dim);
  DA(h->vnew, (size_t)h->Lyr * KV_ROWS * h->kv_dim);
  DA(h->ssq_a, (size_t)(d / 16) * GROUP_ROWS);
  DA(h->ssq_b, (size_t)(d / 16) * GROUP_ROWS);
  int max_splits = T / 64;
  DA(h->part_o, (size_t)h->Hkv * max_splits * GROUP_ROWS * G * 128);
  DA(h->part_ml, (size_t)h->Hkv * max_splits * GROUP_ROWS * G * 2);
  DA(h->part_o_ride, (size_t)h->Hkv * max_splits * 16 * G * 128);
  DA(h->part_ml_ride, (size_t)h->Hkv * max_splits * 16 * G * 2);
  DA(h->hidden, (size_t)MAX_MEMBERS * d);
  DA(h->spec_ok, 4);
  DA(h->rng_backup, 640);
  {
    // slice partials: 8 slices x tiles x 4 planes x 128 floats for qkv / o / down, 4 slice pairs for gate/up
    size_t t8 = (size_t)h->qkv_tiles > (size_t)d / 16 ? (size_t)h->qkv_tiles : (size_t)d / 16;
    // (eight planes for a 64-row pass: single slices for every matrix)
    size_t nfl = 8 * t8 * GROUP_PLANES * 128, gu = (size_t)8 * (2 * dff / 16) * GROUP_PLANES * 128, lmh = (size_t)8 * (h->Vpad / 16) * GROUP_PLANES * 128;
    if (lmh > gu) gu = lmh;                                   // lm_head: 66 MB at V = 32064
    h->gemv_part_floats = nfl > gu ? nfl : gu;
    if (h->gemv_part_floats) DA(h->gemv_part, h->gemv_part_floats + GROUP_ROWS + 8);   // + rstd of the operand rows
  }
  DA(h->xop_d, (size_t)h->S_d * 64 * GROUP_PLANES);     // operand planes (8 rows each)
  DA(h->xop_q, (size_t)h->S_q * 64 * GROUP_PLANES);
  DA(h->xop_ff, (size_t)h->S_ff * 64 * GROUP_PLANES);
  DA(h->base_logits, h->Vpad);
  DA(h->grp_logits, (size_t)GROUP_MAX_LANES * h->Vpad);
  DA(h->grp_argmax, GROUP_MAX_LANES);
  DA(h->base_next, h->Vpad);
  DA(h->argmax_next, 4);
  DA(h->chunk_states, 32);
  DA(h->chunk_k, (size_t)32 * h->kv_dim);
  DA(h->chunk_v, (size_t)32 * h->kv_dim);
  // (the one per-sequence buffer whose size follows DD_MAX_MEMBERS: 8.2 MB at 64 members and V = 32064 — 0.5 GB over 64 lanes —, so it
  // starts at 16 rows, the reference's own lists have 3-8 entries, and grows once when a longer list arrives: ensure_member_rows)
  h->member_rows_cap = 16;
  DA(h->member_logits, (size_t)h->member_rows_cap * h->Vpad);
  DA(h->last_logits, h->Vpad);
  DA(h->last_hidden, d);
  DA(h->argmax_base, 4);
  DA(h->member_tok, MAX_MEMBERS);
  DA(h->member_vote, MAX_MEMBERS);
  DA(h->tokens, MAX_NEW_TOKENS);
  DA(h->keep, h->Lmax);
  DA(h->drop, (size_t)MAX_MEMBERS * h->Lmax);
  DA(h->drop_bits, (size_t)(MAX_MEMBERS / 8) * h->Lmax);
  DA(h->leak_bits, h->Lmax);
  DA(h->n_drop, MAX_MEMBERS);
  DA(h->state, 1);
  // prefill scratch
  DA(h->px, (size_t)T * d);
  DA(h->pq, (size_t)T * (h->q_dim > d ? h->q_dim : d));   // q rows of the prefill; also the final-normed rows [n][d] of prefill_head
  size_t p1 = (size_t)T * (d > h->q_dim ? d : h->q_dim);
  DA(h->p1_hi, p1);
  DA(h->p1_lo, p1);
  DA(h->p2_hi, (size_t)T * dff);
  DA(h->p2_lo, (size_t)T * dff);
  DA(h->image_logits, (size_t)(h->Lmax + 1) * h->Vpad);
  DA(h->row_index, h->Lmax + 1);
  DA(h->epi, h->Lmax);
  DA(h->kl_ws, h->Lmax);
  DA(h->alea, h->Lmax);
  DA(h->var, h->Lmax);
  DA(h->scalars, 4);
  DA(h->topk_vals, (size_t)h->Lmax * DD_MAX_TOPK);
  DA(h->topk_ids, (size_t)h->Lmax * DD_MAX_TOPK);
  h->unc_ws_bytes = dd_uncertainty_workspace_bytes(h->Lmax + 1, h->V);      // (+ 1: the lm_head GEMM also runs the last position: its block partials land there too)
  char* ws;
  DA(ws, h->unc_ws_bytes);
  h->unc_ws = ws;
  DA(h->kv_sums, (size_t)h->Lyr * 2);
  DA(h->seq_tab, 1);
  // RoPE table: inv_freq exactly as HF computes it (fp32 pow and reciprocal), cos/sin on the device
  std::vector<float> inv(64);
  for (int i = 0; i < 64; ++i) inv[i] = 1.0f / powf(c->rope_theta, (float)(2 * i) / 128.0f);
  float* inv_dev;
  DA(inv_dev, 64);
  if (hipMemcpy(inv_dev, inv.data(), 64 * 4, hipMemcpyHostToDevice) != hipSuccess ||
      ddk_rope_table(h->rope_cos, h->rope_sin, T, inv_dev, nullptr) != DD_OK || hipDeviceSynchronize() != hipSuccess) {
    dd_set_error("dd_lm_create: rope table init failed");
    dd_lm_destroy(h);
    return DD_EHIP;
  }
  (void)hipEventCreate(&h->ev0);
  (void)hipEventCreate(&h->ev1);
  if (hipHostMalloc((void**)&h->tok_host, (MAX_NEW_TOKENS + 1 + 4) * sizeof(int32_t), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&h->tok_host_dev, h->tok_host, 0) != hipSuccess) {
    dd_set_error("dd_lm_create: pinned token mirror allocation failed");
    dd_lm_destroy(h);
    return DD_EHIP;
  }
  h->tok_host[0] = 0;
  for (int i = 0; i < 4; ++i) h->tok_host[MAX_NEW_TOKENS + 1 + i] = 0;   // [seq, ok] of the host-decided speculative step
  *out = h;
  return DD_OK;
}

extern "C" int dd_lm_create(const dd_lm_config* c, dd_lm** out) { return lm_create_impl(c, nullptr, out); }

// A further sequence ("lane") over the weights of `weights_from`: own KV cache, state, scratch and token mirror; the
// weight owner must outlive it.  Lanes exist so that dd_lm_group_step can run the base passes of several sequences as
// ONE sweep over the weights.
extern "C" int dd_lm_create_shared(const dd_lm_config* c, dd_lm* weights_from, dd_lm** out) {
  DD_REQUIRE(weights_from, "dd_lm_create_shared: null weight owner");
  return lm_create_impl(c, weights_from, out);
}

// -----------------------------------------------------------------------------------------------
// weights
// -----------------------------------------------------------------------------------------------
extern "C" int dd_lm_load_tensor(dd_lm* h, int id, int layer, const uint16_t* src, int rows, int cols, int on_device) {
  DD_REQUIRE(h && src, "dd_lm_load_tensor: null argument");
  DD_REQUIRE(!h->wsrc, "dd_lm_load_tensor: this handle borrows its weights (dd_lm_create_shared); load into the owner");
  DD_REQUIRE(id >= DD_T_EMBED && id <= DD_T_LM_HEAD, "dd_lm_load_tensor: unknown tensor id %d", id);
  bool per_layer = !(id == DD_T_EMBED || id == DD_T_FINAL_NORM || id == DD_T_LM_HEAD);
  DD_REQUIRE(!per_layer || (layer >= 0 && layer < h->Lyr), "dd_lm_load_tensor: layer %d out of range", layer);
  const int d = h->d, dff = h->dff;
  int er = 0, ec = 0;  // expected shape
  switch (id) {
    case DD_T_EMBED: er = h->V, ec = d; break;
    case DD_T_LM_HEAD: er = h->V, ec = d; break;
    case DD_T_ATTN_NORM: case DD_T_MLP_NORM: case DD_T_FINAL_NORM: er = 1, ec = d; break;
    case DD_T_WQ: er = h->q_dim, ec = d; break;
    case DD_T_WK: case DD_T_WV: er = h->kv_dim, ec = d; break;
    case DD_T_WO: er = d, ec = h->q_dim; break;
    case DD_T_WGATE: case DD_T_WUP: er = dff, ec = d; break;
    case DD_T_WDOWN: er = d, ec = dff; break;
  }
  DD_REQUIRE((size_t)rows * cols == (size_t)er * ec && (er == 1 || (rows == er && cols == ec)),
             "dd_lm_load_tensor: tensor %d expects %d x %d, got %d x %d", id, er, ec, rows, cols);
  DD_REQUIRE(!h->fp8 || id == DD_T_EMBED || id == DD_T_ATTN_NORM || id == DD_T_MLP_NORM || id == DD_T_FINAL_NORM,
             "dd_lm_load_tensor: this engine stores fp8 weights; load matrices with dd_lm_load_tensor_fp8");
  size_t n = (size_t)rows * cols;
  const uint16_t* dev = src;
  uint16_t* staging = nullptr;
  if (!on_device) {
    DD_HIP(hipMalloc((void**)&staging, n * 2));
    hipError_t e = hipMemcpy(staging, src, n * 2, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(staging);
      dd_set_error("dd_lm_load_tensor: H2D copy failed: %s", hipGetErrorString(e));
      return DD_EHIP;
    }
    dev = staging;
  }
  int rc = DD_OK;
  LayerW* w = per_layer ? &h->lw[layer] : nullptr;
  switch (id) {
    case DD_T_EMBED: rc = hipMemcpy(h->embed, dev, n * 2, hipMemcpyDeviceToDevice) == hipSuccess ? DD_OK : DD_EHIP; break;
    case DD_T_ATTN_NORM: rc = ddk_bf16_to_f32(dev, w->norm1, d, nullptr, h->wf); break;
    case DD_T_MLP_NORM: rc = ddk_bf16_to_f32(dev, w->norm2, d, nullptr, h->wf); break;
    case DD_T_FINAL_NORM: rc = ddk_bf16_to_f32(dev, h->final_norm, d, nullptr, h->wf); break;
    case DD_T_WQ: rc = ddk_pack_weight(dev, rows, cols, w->wqkv, 0, 1, PACK_ROPE, h->q_tiles, nullptr); break;
    case DD_T_WK: rc = ddk_pack_weight(dev, rows, cols, w->wqkv, h->q_tiles, 1, PACK_ROPE, h->k_tiles, nullptr); break;
    case DD_T_WV: rc = ddk_pack_weight(dev, rows, cols, w->wqkv, h->q_tiles + h->k_tiles, 1, PACK_PLAIN, h->k_tiles, nullptr); break;
    case DD_T_WO: rc = ddk_pack_weight(dev, rows, cols, w->wo, 0, 1, PACK_PLAIN, d / 16, nullptr); break;
    case DD_T_WGATE: rc = ddk_pack_weight(dev, rows, cols, w->wgu, 0, 2, PACK_PLAIN, dff / 16, nullptr); break;
    case DD_T_WUP: rc = ddk_pack_weight(dev, rows, cols, w->wgu, 1, 2, PACK_PLAIN, dff / 16, nullptr); break;
    case DD_T_WDOWN: rc = ddk_pack_weight(dev, rows, cols, w->wdown, 0, 1, PACK_PLAIN, d / 16, nullptr); break;
    case DD_T_LM_HEAD: rc = ddk_pack_weight(dev, rows, cols, h->lm_head, 0, 1, PACK_PLAIN, h->Vpad / 16, nullptr); break;
  }
  hipError_t e = hipDeviceSynchronize();
  if (staging) (void)hipFree(staging);
  if (rc != DD_OK) return rc;
  DD_HIP(e);
  return DD_OK;
}

extern "C" int dd_lm_load_tensor_fp8(dd_lm* h, int id, int layer, const uint8_t* q, const float* row_scale, int rows,
                                     int cols, int on_device) {
  DD_REQUIRE(h && q && row_scale, "dd_lm_load_tensor_fp8: null argument");
  DD_REQUIRE(!h->wsrc, "dd_lm_load_tensor_fp8: this handle borrows its weights (dd_lm_create_shared); load into the owner");
  DD_REQUIRE(h->fp8, "dd_lm_load_tensor_fp8: the engine was created for bf16 weights (weight_format 0)");
  bool per_layer = id != DD_T_LM_HEAD;
  DD_REQUIRE(id == DD_T_LM_HEAD || (id >= DD_T_WQ && id <= DD_T_WDOWN && id != DD_T_MLP_NORM),
             "dd_lm_load_tensor_fp8: tensor id %d is not a matrix", id);
  DD_REQUIRE(!per_layer || (layer >= 0 && layer < h->Lyr), "dd_lm_load_tensor_fp8: layer %d out of range", layer);
  const int d = h->d, dff = h->dff;
  int er = 0, ec = 0;
  switch (id) {
    case DD_T_LM_HEAD: er = h->V, ec = d; break;
    case DD_T_WQ: er = h->q_dim, ec = d; break;
    case DD_T_WK: case DD_T_WV: er = h->kv_dim, ec = d; break;
    case DD_T_WO: er = d, ec = h->q_dim; break;
    case DD_T_WGATE: case DD_T_WUP: er = dff, ec = d; break;
    case DD_T_WDOWN: er = d, ec = dff; break;
  }
  DD_REQUIRE(rows == er && cols == ec, "dd_lm_load_tensor_fp8: tensor %d expects %d x %d, got %d x %d", id, er, ec, rows, cols);
  const uint8_t* dq = q;
  const float* ds = row_scale;
  uint8_t* sq = nullptr;
  float* ss = nullptr;
  if (!on_device) {
    DD_HIP(hipMalloc((void**)&sq, (size_t)rows * cols));
    DD_HIP(hipMalloc((void**)&ss, (size_t)rows * 4));
    DD_HIP(hipMemcpy(sq, q, (size_t)rows * cols, hipMemcpyHostToDevice));
    DD_HIP(hipMemcpy(ss, row_scale, (size_t)rows * 4, hipMemcpyHostToDevice));
    dq = sq, ds = ss;
  }
  LayerW* w = per_layer ? &h->lw[layer] : nullptr;
  int rc = DD_OK;
  switch (id) {
    case DD_T_WQ: rc = ddk_pack_weight_fp8(dq, ds, rows, cols, w->wqkv, w->s_qkv, 0, 1, PACK_ROPE, h->q_tiles, nullptr); break;
    case DD_T_WK: rc = ddk_pack_weight_fp8(dq, ds, rows, cols, w->wqkv, w->s_qkv, h->q_tiles, 1, PACK_ROPE, h->k_tiles, nullptr); break;
    case DD_T_WV: rc = ddk_pack_weight_fp8(dq, ds, rows, cols, w->wqkv, w->s_qkv, h->q_tiles + h->k_tiles, 1, PACK_PLAIN, h->k_tiles, nullptr); break;
    case DD_T_WO: rc = ddk_pack_weight_fp8(dq, ds, rows, cols, w->wo, w->s_o, 0, 1, PACK_PLAIN, d / 16, nullptr); break;
    case DD_T_WGATE: rc = ddk_pack_weight_fp8(dq, ds, rows, cols, w->wgu, w->s_gu, 0, 2, PACK_PLAIN, dff / 16, nullptr); break;
    case DD_T_WUP: rc = ddk_pack_weight_fp8(dq, ds, rows, cols, w->wgu, w->s_gu, 1, 2, PACK_PLAIN, dff / 16, nullptr); break;
    case DD_T_WDOWN: rc = ddk_pack_weight_fp8(dq, ds, rows, cols, w->wdown, w->s_down, 0, 1, PACK_PLAIN, d / 16, nullptr); break;
    case DD_T_LM_HEAD: rc = ddk_pack_weight_fp8(dq, ds, rows, cols, h->lm_head, h->s_lm, 0, 1, PACK_PLAIN, h->Vpad / 16, nullptr); break;
  }
  hipError_t e = hipDeviceSynchronize();
  if (sq) (void)hipFree(sq);
  if (ss) (void)hipFree(ss);
  if (rc != DD_OK) return rc;
  DD_HIP(e);
  return DD_OK;
}

extern "C" int dd_lm_load_synthetic(dd_lm* h, uint32_t seed, float std) {
  DD_REQUIRE(h && !h->wsrc, "dd_lm_load_synthetic: null handle, or a handle that borrows its weights");
  DD_REQUIRE(h, "dd_lm_load_synthetic: null handle");
  const int d = h->d, dff = h->dff;
  if (h->fp8) {   // random finite e4m3 bytes (|q| <= 240, rms ~ 40) with a constant row scale that gives ~std
    uint32_t s8 = seed * 2654435761u + 7;
    const float sc = std / 40.0f;
    for (int l = 0; l < h->Lyr; ++l) {
      LayerW& w = h->lw[l];
      RC(ddk_fill_synthetic_fp8((uint8_t*)w.wqkv, (size_t)h->qkv_tiles * h->S_d * 512, s8++, nullptr));
      RC(ddk_fill_synthetic_fp8((uint8_t*)w.wo, (size_t)(d / 16) * h->S_q * 512, s8++, nullptr));
      RC(ddk_fill_synthetic_fp8((uint8_t*)w.wgu, (size_t)(2 * dff / 16) * h->S_d * 512, s8++, nullptr));
      RC(ddk_fill_synthetic_fp8((uint8_t*)w.wdown, (size_t)(d / 16) * h->S_ff * 512, s8++, nullptr));
      RC(ddk_fill_const_f32(w.s_qkv, (size_t)h->qkv_tiles * 16, sc, nullptr));
      RC(ddk_fill_const_f32(w.s_o, d, sc, nullptr));
      RC(ddk_fill_const_f32(w.s_gu, (size_t)2 * dff, sc, nullptr));
      RC(ddk_fill_const_f32(w.s_down, d, sc, nullptr));
      RC(ddk_fill_const_f32(w.norm1, d, 1.0f, nullptr));
      RC(ddk_fill_const_f32(w.norm2, d, 1.0f, nullptr));
    }
    RC(ddk_fill_synthetic_fp8((uint8_t*)h->lm_head, (size_t)(h->Vpad / 16) * h->S_d * 512, s8++, nullptr));
    RC(ddk_fill_const_f32(h->s_lm, h->Vpad, sc, nullptr));
    RC(ddk_fill_const_f32(h->final_norm, d, 1.0f, nullptr));
    RC(ddk_fill_synthetic(h->embed, (size_t)h->V * d, s8++, 1.0f, nullptr));
    DD_HIP(hipDeviceSynchronize());
    return DD_OK;
  }
  uint32_t s = seed * 2654435761u + 1;
  for (int l = 0; l < h->Lyr; ++l) {
    LayerW& w = h->lw[l];
    RC(ddk_fill_synthetic((uint16_t*)w.wqkv, (size_t)h->qkv_tiles * h->S_d * 512, s++, std, nullptr, h->wf));
    RC(ddk_fill_synthetic((uint16_t*)w.wo, (size_t)(d / 16) * h->S_q * 512, s++, std, nullptr, h->wf));
    RC(ddk_fill_synthetic((uint16_t*)w.wgu, (size_t)(2 * dff / 16) * h->S_d * 512, s++, std, nullptr, h->wf));
    RC(ddk_fill_synthetic((uint16_t*)w.wdown, (size_t)(d / 16) * h->S_ff * 512, s++, std, nullptr, h->wf));
    RC(ddk_fill_const_f32(w.norm1, d, 1.0f, nullptr));
    RC(ddk_fill_const_f32(w.norm2, d, 1.0f, nullptr));
  }
  RC(ddk_fill_synthetic((uint16_t*)h->lm_head, (size_t)(h->Vpad / 16) * h->S_d * 512, s++, std, nullptr, h->wf));
  RC(ddk_fill_const_f32(h->final_norm, d, 1.0f, nullptr));
  RC(ddk_fill_synthetic(h->embed, (size_t)h->V * d, s++, 1.0f, nullptr, h->wf));
  DD_HIP(hipDeviceSynchronize());
  return DD_OK;
}

// -----------------------------------------------------------------------------------------------
// small state kernels
// -----------------------------------------------------------------------------------------------
__global__ void k_state_after_prefill(DDState* st, int T0, const int32_t* first_tok, int32_t* tokens,
                                      volatile int32_t* mirror) {
  if (threadIdx.x == 0) {
    st->T = T0;
    st->pos = T0;
    st->n_tok = 1;
    st->cur_tok = first_tok[0];
    st->winner = 0;
    st->voted = first_tok[0];
    st->done = dd_is_eos(st, first_tok[0]) ? 1 : 0;   // the eos list itself survives prefills (dd_lm_set_eos)
    tokens[0] = first_tok[0];
    mirror[1] = first_tok[0];
    __threadfence_system();
    mirror[0] = 1;
  }
}
// pos rule: LLaVA-family llava.py:283 (sum(mask)-1 == T; the mask is rebuilt all-ones every step).  InstructBLIP:
// transformers 5.x takes the position from the cache length (== T); under the reference's pinned 4.44,
// prepare_inputs_for_generation derives it from the caller's 2-D mask, which still carries the zeros the last member
// of the previous step left behind, so pos = T - #leaked zeros  (leak_mask == 2).
__global__ __launch_bounds__(256) void k_step_begin(DDState* st, const uint8_t* leak_bits, int L, int mask_positions) {
  __shared__ int cnt[4];
  int c = 0;
  if (mask_positions)
    for (int l = threadIdx.x; l < L; l += 256) c += leak_bits[l] & 1;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) st->pos = st->T - (cnt[0] + cnt[1] + cnt[2] + cnt[3]);
}
__global__ __launch_bounds__(1024) void k_step_end(DDState* st, int K, const int32_t* argmax_base,
                                                   const int32_t* member_tok, const float* base_logits,
                                                   const float* member_logits, int Vpad, float* last_logits,
                                                   int32_t* tokens, const uint8_t* drop_bits, int L, uint8_t* leak_bits,
                                                   int leak, const float* hidden_rows, int d, float* last_hidden,
                                                   volatile int32_t* mirror) {
  if (st->done) return;          // the sequence ended at an EOS: this enqueued-ahead step emits and advances nothing
  int win = K > 0 ? st->winner : 0;
  const float* src = K > 0 ? member_logits + (size_t)win * Vpad : base_logits;
  for (int i = threadIdx.x; i < (Vpad >> 2); i += 1024) ((f32x4_t*)last_logits)[i] = ((const f32x4_t*)src)[i];   // Vpad: a multiple of 16
  if (leak && K > 0)
    for (int l = threadIdx.x; l < L; l += 1024)
      leak_bits[l] = (drop_bits[(size_t)((K - 1) >> 3) * L + l] >> ((K - 1) & 7)) & 1;  // Q2: last member's zeros stay
  __syncthreads();
  if (threadIdx.x == 0) {
    int tok = K > 0 ? member_tok[win] : argmax_base[0];
    int n = st->n_tok;
    if (n < MAX_NEW_TOKENS) {
      tokens[n] = tok;
      mirror[1 + n] = tok;          // host-mapped: visible to a polling host thread without a stream sync
      __threadfence_system();
      mirror[0] = n + 1;
    }
    st->n_tok = n + 1;
    st->cur_tok = tok;
    st->T = st->T + 1;
    if (dd_is_eos(st, tok)) st->done = 1;
  }
}
// rows of the prompt that go through lm_head: the visual span, then the last position
__global__ void k_prefill_rows(int32_t* rows, int span_start, int L, int T0) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < L) rows[i] = span_start + i;
  else if (i == L) rows[L] = T0 - 1;
}
// ---- the same two kernels for several sequences at once (group step): block = sequence
struct StepBeginLanes {
  DDState* st[GROUP_MAX_LANES];
  const uint8_t* leak_bits[GROUP_MAX_LANES];
  int L[GROUP_MAX_LANES], mask_positions[GROUP_MAX_LANES];
};
__global__ __launch_bounds__(256) void k_step_begin_lanes(StepBeginLanes t) {
  __shared__ int cnt[4];
  const int q = blockIdx.x;
  int c = 0;
  if (t.mask_positions[q])
    for (int l = threadIdx.x; l < t.L[q]; l += 256) c += t.leak_bits[q][l] & 1;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) t.st[q]->pos = t.st[q]->T - (cnt[0] + cnt[1] + cnt[2] + cnt[3]);
}
struct StepEndLanes {
  DDState* st[8];
  const int32_t* member_tok[8];
  const float* member_logits[8];
  float* last_logits[8];
  int32_t* tokens[8];
  const uint8_t* drop_bits[8];
  uint8_t* leak_bits[8];
  volatile int32_t* mirror[8];
  int L[8], leak[8];
};
__global__ __launch_bounds__(1024) void k_step_end_lanes(StepEndLanes t, int K, int Vpad) {
  const int q = blockIdx.x;
  DDState* st = t.st[q];
  if (st->done) return;
  const int win = st->winner;
  const float* src = t.member_logits[q] + (size_t)win * Vpad;
  for (int i = threadIdx.x; i < (Vpad >> 2); i += 1024) ((f32x4_t*)t.last_logits[q])[i] = ((const f32x4_t*)src)[i];   // Vpad: a multiple of 16
  if (t.leak[q])
    for (int l = threadIdx.x; l < t.L[q]; l += 1024)
      t.leak_bits[q][l] = (t.drop_bits[q][(size_t)((K - 1) >> 3) * t.L[q] + l] >> ((K - 1) & 7)) & 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tok = t.member_tok[q][win];
    int n = st->n_tok;
    if (n < MAX_NEW_TOKENS) {
      t.tokens[q][n] = tok;
      t.mirror[q][1 + n] = tok;
      __threadfence_system();
      t.mirror[q][0] = n + 1;
    }
    st->n_tok = n + 1;
    st->cur_tok = tok;
    st->T = st->T + 1;
    if (dd_is_eos(st, tok)) st->done = 1;
  }
}
__global__ void k_state_truncate(DDState* st, int T) {
  if (threadIdx.x == 0) {
    st->T = T;
    st->pos = T;
    st->n_tok = 0;
    st->winner = 0;
    st->done = 0;
  }
}
struct EosList {
  int32_t n, ids[DD_MAX_EOS];
};
__global__ void k_set_eos(DDState* st, EosList e) {
  if (threadIdx.x == 0) {
    st->n_eos = e.n;
    for (int i = 0; i < DD_MAX_EOS; ++i) st->eos[i] = e.ids[i];
    st->done = (st->n_tok > 0 && dd_is_eos(st, st->cur_tok)) ? 1 : 0;   // the last emitted token may already be one
  }
}
__global__ void k_set_token(DDState* st, int tok) {
  if (threadIdx.x == 0) st->cur_tok = tok;
}
__global__ void k_set_winner(DDState* st, int winner, const int32_t* tok) {
  if (threadIdx.x == 0 && !st->done) {
    st->winner = winner;
    st->voted = tok[winner];
  }
}
// first-token ensemble: the winner's logits / hidden row become the sequence's, its argmax replaces the greedy token
__global__ __launch_bounds__(1024) void k_first_token_from_member(DDState* st, const int32_t* member_tok,
                                                                  const float* member_logits, int Vpad,
                                                                  float* last_logits, int32_t* tokens,
                                                                  const float* hidden_rows, int d, float* last_hidden,
                                                                  volatile int32_t* mirror) {
  const int win = st->winner;
  for (int i = threadIdx.x; i < Vpad; i += 1024) last_logits[i] = member_logits[(size_t)win * Vpad + i];
  for (int i = threadIdx.x; i < d; i += 1024) last_hidden[i] = hidden_rows[(size_t)win * d + i];
  if (threadIdx.x == 0) {
    int tok = member_tok[win];
    tokens[0] = tok;
    st->cur_tok = tok;
    st->done = dd_is_eos(st, tok) ? 1 : 0;
    mirror[1] = tok;
    __threadfence_system();
    mirror[0] = 1;
  }
}

// -----------------------------------------------------------------------------------------------
// prefill
// -----------------------------------------------------------------------------------------------
// all layers over the T0 prompt rows in h->px (overwritten in place), K/V written into the cache.  drop_plane/drop_bit:
// the member's zero columns of the 2-D attention mask (first-token ensemble), or nullptr for the un-masked pass.
static int prefill_layers(dd_lm* h, int T0, const uint8_t* drop_plane, int drop_bit, int span_start, int span_len,
                          hipStream_t st, int pos0 = 0) {
  const int d = h->d, dff = h->dff;
  // fp8 storage: the prefill GEMM runs on a bf16 expansion of ONE matrix at a time (exact), scales in its epilogue
  auto wsel = [&](GemmArgs& g, u32x4_t* W, float* scale, int n_tiles, int S) -> int {
    if (!h->fp8) {
      g.W = W;
      return DD_OK;
    }
    int rc = ddk_dequant_tiles(W, h->deq_tmp, n_tiles, S, st);
    g.W = h->deq_tmp, g.wscale = scale;
    return rc;
  };
  for (int l = 0; l < h->Lyr; ++l) {
    LayerW& w = h->lw[l];
    RC(ddk_rmsnorm_split(h->px, T0, d, w.norm1, h->cfg.rms_eps, h->p1_hi, h->p1_lo, nullptr, nullptr, st, h->wf));
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.wf = h->wf;
    g.a_hi = h->p1_hi, g.a_lo = h->p1_lo, g.M = T0, g.S = h->S_d, g.n_tiles = h->qkv_tiles;
    RC(wsel(g, w.wqkv, w.s_qkv, h->qkv_tiles, h->S_d));
    g.qbuf = h->pq, g.kc = h->kc + (size_t)l * h->lsk, g.vc = h->vc + (size_t)l * h->lsv, g.T_cap = h->T_cap;
    g.q_tiles = h->q_tiles, g.k_tiles = h->k_tiles, g.q_dim = h->q_dim, g.kv_dim = h->kv_dim, g.pos0 = pos0, g.kv16 = h->kv16;
    g.rope_cos = h->rope_cos, g.rope_sin = h->rope_sin;
    RC(ddk_gemm(EPI_QKV, g, st));
    RC(ddk_attn_prefill(h->pq, g.kc, g.vc, T0, h->T_cap, h->H, h->Hkv, h->p1_hi, h->p1_lo, drop_plane, drop_bit,
                        span_start, span_len, pos0, st, nullptr, h->kv16, h->wf));
    memset(&g, 0, sizeof(g));
    g.wf = h->wf;
    g.a_hi = h->p1_hi, g.a_lo = h->p1_lo, g.M = T0, g.S = h->S_q, g.n_tiles = d / 16, g.out = h->px, g.ldo = d;
    RC(wsel(g, w.wo, w.s_o, d / 16, h->S_q));
    RC(ddk_gemm(EPI_RESID, g, st));
    RC(ddk_rmsnorm_split(h->px, T0, d, w.norm2, h->cfg.rms_eps, h->p1_hi, h->p1_lo, nullptr, nullptr, st, h->wf));
    memset(&g, 0, sizeof(g));
    g.wf = h->wf;
    g.a_hi = h->p1_hi, g.a_lo = h->p1_lo, g.M = T0, g.S = h->S_d, g.n_tiles = 2 * dff / 16;
    RC(wsel(g, w.wgu, w.s_gu, 2 * dff / 16, h->S_d));
    g.o_hi = h->p2_hi, g.o_lo = h->p2_lo, g.ld_planes = dff;
    RC(ddk_gemm(EPI_SILU, g, st));
    memset(&g, 0, sizeof(g));
    g.wf = h->wf;
    g.a_hi = h->p2_hi, g.a_lo = h->p2_lo, g.M = T0, g.S = h->S_ff, g.n_tiles = d / 16, g.out = h->px, g.ldo = d;
    RC(wsel(g, w.wdown, w.s_down, d / 16, h->S_ff));
    RC(ddk_gemm(EPI_RESID, g, st));
  }
  return DD_OK;
}

// final norm + lm_head over `n_rows` rows of h->px selected by row_index (device) -> logits [n_rows][Vpad];
// the normed rows stay in h->pq
float* dd_uncertainty_partials(void* ws, int L_cap, int V, int* n_cb);
int dd_vision_uncertainty_impl(const float* logits, int L, int V, int ld, float* var_tok, float* epi_tok, float* alea_tok, float* scalars3,
                               int k_top, float* topk_vals, int32_t* topk_ids, void* ws, size_t ws_bytes, hipStream_t st, int L_cap,
                               bool partials_ready);
// rowstat: the scorer's softmax block partials, written by the GEMM's epilogue (GemmArgs::rowstat; prefill_tail)
static int prefill_head(dd_lm* h, const int32_t* row_index, int n_rows, float* logits, hipStream_t st,
                        const float* src = nullptr, float* rowstat = nullptr, int rowstat_ld = 0) {
  const int d = h->d;
  RC(ddk_rmsnorm_split(src ? src : h->px, n_rows, d, h->final_norm, h->cfg.rms_eps, h->p1_hi, h->p1_lo, row_index, h->pq, st, h->wf));
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.wf = h->wf;
  g.a_hi = h->p1_hi, g.a_lo = h->p1_lo, g.M = n_rows, g.S = h->S_d, g.n_tiles = h->Vpad / 16;
  if (!h->fp8) {
    g.W = h->lm_head;
  } else {
    RC(ddk_dequant_tiles(h->lm_head, h->deq_tmp, h->Vpad / 16, h->S_d, st));
    g.W = h->deq_tmp, g.wscale = h->s_lm;
  }
  g.out = logits, g.ldo = h->Vpad, g.n_valid = h->V;
  g.rowstat = rowstat, g.rowstat_ld = rowstat_ld;
  return ddk_gemm(EPI_STORE, g, st);
}

static int prefill_tail(dd_lm* h, const float* x_rows, int T0, int span_start, int span_len, hipStream_t st);
int dd_engine_prefill_head(dd_lm* h, const int32_t* row_index, int n_rows, float* logits, hipStream_t st, const float* src) {
  return prefill_head(h, row_index, n_rows, logits, st, src);
}
int dd_engine_prefill_tail(dd_lm* h, const float* x_rows, int T0, int span_start, int span_len, hipStream_t st) {
  return prefill_tail(h, x_rows, T0, span_start, span_len, st);
}
int dd_engine_tp_alloc(dd_lm* h, float** p, size_t floats) { return dalloc(h, p, floats); }
extern "C" int dd_lm_prefill(dd_lm* h, const float* embeds, int T0, int span_start, int span_len, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h && embeds, "dd_lm_prefill: null argument");
  DD_REQUIRE(h->tp_world == 1, "this handle is a tensor-parallel shard: drive it through dd_lm_tp_* (include/dropdec.h)");
  DD_REQUIRE(T0 >= 1 && T0 < h->T_cap, "dd_lm_prefill: T0=%d out of range (KV capacity %d)", T0, h->T_cap);
  // reference llava.py:134-138 raises ValueError on an image-token / feature count mismatch; same contract here
  DD_REQUIRE(span_len >= 1 && span_len <= h->Lmax && span_start >= 0 && span_start + span_len <= T0,
             "dd_lm_prefill: visual span [%d, %d) does not fit the %d input positions (max_visual %d)", span_start,
             span_start + span_len, T0, h->Lmax);
  const int d = h->d;
  DD_HIP(hipMemcpyAsync(h->px, embeds, (size_t)T0 * d * 4, hipMemcpyDeviceToDevice, st));
  RC(prefill_layers(h, T0, nullptr, 0, span_start, span_len, st));
  return prefill_tail(h, h->px, T0, span_start, span_len, st);
}

// What follows the layers of a prefill: lm_head over the visual span + the last position, scorer, first token, state.
// x_rows: the sequence's final residual rows [T0][d] (h->px, or its slice of a batch's rows).
static int prefill_tail(dd_lm* h, const float* x_rows, int T0, int span_start, int span_len, hipStream_t st) {
  const int d = h->d, L = span_len;
  // lm_head over the visual span + the last position only (the reference projects all T0 positions,
  // llava.py:294-305, but consumes just these: llava.py:311-314 and HF's greedy argmax)
  k_prefill_rows<<<(L + 1 + 255) / 256, 256, 0, st>>>(h->row_index, span_start, L, T0);   // no host data: prefill stays asynchronous
  DD_CHECK_LAUNCH();
  // the scorer's row statistics come out of the lm_head GEMM's epilogue as block partials (the logits are read twice afterwards, not three times)
  int n_cb = 0;
  float* bpart = dd_uncertainty_partials(h->unc_ws, h->Lmax + 1, h->V, &n_cb);
  RC(prefill_head(h, h->row_index, L + 1, h->image_logits, st, x_rows, bpart, n_cb));
  DD_HIP(hipMemcpyAsync(h->last_hidden, h->pq + (size_t)L * d, (size_t)d * 4, hipMemcpyDeviceToDevice, st));
  RC(dd_vision_uncertainty_impl(h->image_logits, L, h->V, h->Vpad, h->var, h->epi, h->alea, h->scalars, h->cfg.k_top,
                                h->topk_vals, h->topk_ids, h->unc_ws, h->unc_ws_bytes, st, h->Lmax + 1, true));
  DD_HIP(hipMemcpyAsync(h->last_logits, h->image_logits + (size_t)L * h->Vpad, (size_t)h->Vpad * 4,
                        hipMemcpyDeviceToDevice, st));
  RC(dd_argmax_rows(h->last_logits, 1, h->V, h->Vpad, h->argmax_base, st));
  h->tok_host[0] = 0;
  k_state_after_prefill<<<1, 64, 0, st>>>(h->state, T0, h->argmax_base, h->tokens, h->tok_host_dev);
  DD_CHECK_LAUNCH();
  DD_HIP(hipMemsetAsync(h->leak_bits, 0, h->Lmax, st));
  h->T_host = T0, h->span_start = span_start, h->L = L, h->n_tok_host = 1, h->prefilled = true, h->have_leak = false;
  h->last_K = 0;
  h->steps_since_prefill = 0, h->pend_valid = false;
  return DD_OK;
}

// -----------------------------------------------------------------------------------------------
// Prefill of several sequences at once (lanes over one set of weights): their prompts are laid back to back — every
// sequence padded to a whole number of 128-row blocks — and run through the layers as ONE matrix of n * seq_rows rows, so
// the GEMMs stream each weight matrix once per batch instead of once per sequence and fill the chip with row blocks
// (608 rows alone are five blocks: 0.81 PFLOP/s; thousands of rows: 1.05).  Rows are independent in every kernel of
// the prefill except the attention, which runs per sequence as before; each accumulator tile sees the MFMA sequence of
// the one-sequence prefill, so every sequence's logits, scores, first token and cache are bit-identical to
// dd_lm_prefill on it alone.  fp8 matrices are expanded to bf16 tiles once per BATCH and matrix (round 4; one prefill per sequence
// expanded every matrix per sequence: 2 % of config 5).  Different cache capacities or a batch too small for the large blocks fall back to
// one dd_lm_prefill per sequence.
// -----------------------------------------------------------------------------------------------
extern "C" int dd_lm_prefill_group(dd_lm* const* lanes, int n, const float* const* embeds, const int32_t* T0s,
                                   const int32_t* span_starts, const int32_t* span_lens, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(lanes && embeds && T0s && span_starts && span_lens && n >= 1 && n <= 32, "dd_lm_prefill_group: bad arguments (1..32 sequences)");
  dd_lm* h0 = lanes[0];
  DD_REQUIRE(h0, "dd_lm_prefill_group: null handle");
  dd_lm* owner = h0->wsrc ? h0->wsrc : h0;
  bool batched = n > 1;
  int maxT = 0;
  for (int i = 0; i < n; ++i) {
    dd_lm* q = lanes[i];
    DD_REQUIRE(q && embeds[i], "dd_lm_prefill_group: null argument (sequence %d)", i);
    DD_REQUIRE(q->tp_world == 1, "dd_lm_prefill_group: sequence %d is a tensor-parallel shard", i);
    DD_REQUIRE((q->wsrc ? q->wsrc : q) == owner, "dd_lm_prefill_group: sequence %d does not share the group's weights", i);
    for (int j = 0; j < i; ++j) DD_REQUIRE(lanes[j] != q, "dd_lm_prefill_group: sequence listed twice");
    DD_REQUIRE(T0s[i] >= 1 && T0s[i] < q->T_cap, "dd_lm_prefill_group: sequence %d: T0=%d out of range (KV capacity %d)", i, T0s[i], q->T_cap);
    DD_REQUIRE(span_lens[i] >= 1 && span_lens[i] <= q->Lmax && span_starts[i] >= 0 && span_starts[i] + span_lens[i] <= T0s[i],
               "dd_lm_prefill_group: sequence %d: visual span [%d, %d) does not fit the %d input positions (max_visual %d)", i,
               span_starts[i], span_starts[i] + span_lens[i], T0s[i], q->Lmax);
    batched = batched && q->T_cap == h0->T_cap && q->kv16 == h0->kv16 && q->lsk == h0->lsk && q->lsv == h0->lsv;
    if (T0s[i] > maxT) maxT = T0s[i];
  }
  const int seq_rows = (maxT + 127) / 128 * 128;
  const size_t M = (size_t)n * seq_rows;
  if (!batched || M < 1024) {
    for (int i = 0; i < n; ++i) RC(dd_lm_prefill(lanes[i], embeds[i], T0s[i], span_starts[i], span_lens[i], stream_));
    return DD_OK;
  }
  // batch scratch, owned by the weight owner, grown on demand (a growth waits for the device and releases the smaller blocks:
  // callers that batch the same number of prompts every time — GroupPipeline does — pay it once)
  const int d = h0->d, dff = h0->dff;
  if (owner->pb_rows < M) {
    DD_HIP(hipDeviceSynchronize());                        // nobody may still be using the smaller scratch
    const size_t w1 = (size_t)(d > h0->q_dim ? d : h0->q_dim);
    {
      const size_t old = owner->pb_rows;
      void* olds[6] = {owner->pb_x, owner->pb_q, owner->pb1_hi, owner->pb1_lo, owner->pb2_hi, owner->pb2_lo};
      const size_t old_bytes[6] = {old * d * 4, old * h0->q_dim * 4, old * w1 * 2, old * w1 * 2, old * dff * 2, old * dff * 2};
      for (int i = 0; i < 6; ++i) {
        if (!olds[i]) continue;
        for (auto it = owner->allocs.begin(); it != owner->allocs.end(); ++it)
          if (*it == olds[i]) {
            owner->allocs.erase(it);
            break;
          }
        (void)hipFree(olds[i]);
        owner->bytes -= old_bytes[i] ? old_bytes[i] : 16;
      }
      owner->pb_x = owner->pb_q = nullptr;
      owner->pb1_hi = owner->pb1_lo = owner->pb2_hi = owner->pb2_lo = nullptr;
      owner->pb_rows = 0;
    }
    if (dalloc(owner, &owner->pb_x, M * d) != DD_OK || dalloc(owner, &owner->pb_q, M * h0->q_dim) != DD_OK ||
        dalloc(owner, &owner->pb1_hi, M * w1) != DD_OK || dalloc(owner, &owner->pb1_lo, M * w1) != DD_OK ||
        dalloc(owner, &owner->pb2_hi, M * dff) != DD_OK || dalloc(owner, &owner->pb2_lo, M * dff) != DD_OK) {
      owner->pb_rows = 0;
      return DD_ENOMEM;
    }
    owner->pb_rows = M;
  }
  float *bx = owner->pb_x, *bq = owner->pb_q;
  uint16_t *b1h = owner->pb1_hi, *b1l = owner->pb1_lo, *b2h = owner->pb2_hi, *b2l = owner->pb2_lo;
  {
    SeqTab tab;                                             // per-sequence lengths and cache bases, read by the QKV epilogue
    memset(&tab, 0, sizeof(tab));
    for (int i = 0; i < n; ++i) tab.T[i] = T0s[i], tab.kc[i] = lanes[i]->kc, tab.vc[i] = lanes[i]->vc;
    RC(ddk_put_seq_tab(tab, h0->seq_tab, st));
  }
  DD_HIP(hipMemsetAsync(bx, 0, M * d * 4, st));            // padding rows: zeros (finite everywhere downstream, never stored)
  for (int i = 0; i < n; ++i)
    DD_HIP(hipMemcpyAsync(bx + (size_t)i * seq_rows * d, embeds[i], (size_t)T0s[i] * d * 4, hipMemcpyDeviceToDevice, st));
  auto wsel = [&](GemmArgs& g, u32x4_t* W, float* scale, int n_tiles, int S) -> int {      // as in prefill_layers
    if (!h0->fp8) {
      g.W = W;
      return DD_OK;
    }
    int rc = ddk_dequant_tiles(W, h0->deq_tmp, n_tiles, S, st);
    g.W = h0->deq_tmp, g.wscale = scale;
    return rc;
  };
  for (int l = 0; l < h0->Lyr; ++l) {
    LayerW& w = h0->lw[l];
    RC(ddk_rmsnorm_split(bx, (int)M, d, w.norm1, h0->cfg.rms_eps, b1h, b1l, nullptr, nullptr, st, h0->wf));
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.wf = h0->wf;
    g.a_hi = b1h, g.a_lo = b1l, g.M = (int)M, g.S = h0->S_d, g.n_tiles = h0->qkv_tiles;
    RC(wsel(g, w.wqkv, w.s_qkv, h0->qkv_tiles, h0->S_d));
    g.qbuf = bq, g.T_cap = h0->T_cap, g.q_tiles = h0->q_tiles, g.k_tiles = h0->k_tiles, g.q_dim = h0->q_dim, g.kv_dim = h0->kv_dim;
    g.pos0 = 0, g.kv16 = h0->kv16, g.rope_cos = h0->rope_cos, g.rope_sin = h0->rope_sin;
    g.seq_rows = seq_rows, g.seq_tab = h0->seq_tab, g.seq_off_k = (size_t)l * h0->lsk, g.seq_off_v = (size_t)l * h0->lsv;
    g.kc = lanes[0]->kc + g.seq_off_k, g.vc = lanes[0]->vc + g.seq_off_v;
    RC(ddk_gemm(EPI_QKV, g, st));
    if (ddk_prefill_mfma_enabled()) {                      // causal attention of every sequence over its own cache, one launch
      RC(ddk_attn_prefill_seqs(bq, h0->seq_tab, g.seq_off_k, g.seq_off_v, n, seq_rows, maxT, h0->T_cap, h0->H, h0->Hkv, b1h, b1l, st, h0->kv16,
                               h0->wf));
    } else {
      for (int i = 0; i < n; ++i) {
        const size_t r0 = (size_t)i * seq_rows;
        RC(ddk_attn_prefill(bq + r0 * h0->q_dim, lanes[i]->kc + g.seq_off_k, lanes[i]->vc + g.seq_off_v, T0s[i], h0->T_cap, h0->H, h0->Hkv,
                            b1h + r0 * h0->q_dim, b1l + r0 * h0->q_dim, nullptr, 0, span_starts[i], span_lens[i], 0, st, nullptr, h0->kv16, h0->wf));
      }
    }
    memset(&g, 0, sizeof(g));
    g.wf = h0->wf;
    g.a_hi = b1h, g.a_lo = b1l, g.M = (int)M, g.S = h0->S_q, g.n_tiles = d / 16, g.out = bx, g.ldo = d;
    RC(wsel(g, w.wo, w.s_o, d / 16, h0->S_q));
    RC(ddk_gemm(EPI_RESID, g, st));
    RC(ddk_rmsnorm_split(bx, (int)M, d, w.norm2, h0->cfg.rms_eps, b1h, b1l, nullptr, nullptr, st, h0->wf));
    memset(&g, 0, sizeof(g));
    g.wf = h0->wf;
    g.a_hi = b1h, g.a_lo = b1l, g.M = (int)M, g.S = h0->S_d, g.n_tiles = 2 * dff / 16;
    RC(wsel(g, w.wgu, w.s_gu, 2 * dff / 16, h0->S_d));
    g.o_hi = b2h, g.o_lo = b2l, g.ld_planes = dff;
    RC(ddk_gemm(EPI_SILU, g, st));
    memset(&g, 0, sizeof(g));
    g.wf = h0->wf;
    g.a_hi = b2h, g.a_lo = b2l, g.M = (int)M, g.S = h0->S_ff, g.n_tiles = d / 16, g.out = bx, g.ldo = d;
    RC(wsel(g, w.wdown, w.s_down, d / 16, h0->S_ff));
    RC(ddk_gemm(EPI_RESID, g, st));
  }
  for (int i = 0; i < n; ++i) RC(prefill_tail(lanes[i], bx + (size_t)i * seq_rows * d, T0s[i], span_starts[i], span_lens[i], st));
  return DD_OK;
}

// Cut the sequence back to its first T_keep positions (T_keep >= end of the visual span): everything the prefill derived
// from the image — uncertainty, top-k ids, image logits — stays valid because attention is causal.  Used with
// dd_lm_prefill_extend to answer several questions about ONE image (pope_test/pope_test.py asks 6 per image) without
// re-running the 576 visual positions: the reference re-runs the whole prompt each time (pope_test.py:228-232); the K/V
// rows it recomputes for the shared prefix are the ones already in the cache.
extern "C" int dd_lm_truncate(dd_lm* h, int T_keep, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h && h->prefilled, "dd_lm_truncate: no prefilled sequence");
  DD_REQUIRE(T_keep >= h->span_start + h->L && T_keep >= 1 && T_keep <= h->T_host,
             "dd_lm_truncate: T_keep=%d outside [%d, %d] (end of the visual span .. current length)", T_keep,
             h->span_start + h->L, h->T_host);
  k_state_truncate<<<1, 64, 0, st>>>(h->state, T_keep);
  DD_CHECK_LAUNCH();
  DD_HIP(hipMemsetAsync(h->leak_bits, 0, h->Lmax, st));
  h->tok_host[0] = 0;
  h->T_host = T_keep, h->n_tok_host = 0, h->have_leak = false, h->last_K = 0, h->steps_since_prefill = 0, h->pend_valid = false;
  return DD_OK;
}

// Short chunks (n <= 32 rows: a question's text after a cached image prefix) go through the DECODE kernels: the prefill
// GEMM is built for hundreds of rows and streams the weights at under 1 TB/s when it has a dozen, the 8 / 16 / 32-row
// GEMVs stream them at 4-5 TB/s.  Rows carry their own positions (RoPE in the QKV epilogue), the chunk's K/V rows are
// scattered into the cache before the causal chunk attention (k_attn_prefill with q0), which hands its rows to o_proj
// as packed operand planes.  Same arithmetic as the decode path, i.e. within fp32 rounding of the GEMM path, not
// bit-identical to it.
static int g_extend_rows = 1;   // dd_set_tuning key 11: chunks of <= 32 rows through the decode GEMVs (0: always the GEMM path)
void dd_engine_set_extend_rows(int on) { g_extend_rows = on; }

static int prefill_extend_rows(dd_lm* h, const float* embeds, int n, hipStream_t st) {
  const int d = h->d, dff = h->dff, pos0 = h->T_host;
  const int cap = n <= 8 ? 8 : (n <= 16 ? 16 : 32), ng = cap / 8;
  auto gemv = [&](int epi, GemvArgs& a) -> int {
    a.nb = cap == 8 ? n : 8;
    if (cap == 8) return ddk_gemv(epi, a, st);
    a.n_groups = ng;
    a.part = h->gemv_part, a.part_floats = h->gemv_part_floats;
    return ddk_gemv_groups(epi, a, st);
  };
  RC(ddk_chunk_positions(h->chunk_states, h->state, n, st));
  RC(ddk_pack_embed_rows(embeds, n, cap, d, h->xa, h->lw[0].norm1, h->xop_d, h->ssq_a, d / 16, st, h->wf));
  int ssq_n = 1;
  for (int l = 0; l < h->Lyr; ++l) {
    LayerW& w = h->lw[l];
    float* kc = h->kc + (size_t)l * h->lsk;
    float* vc = h->vc + (size_t)l * h->lsv;
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wqkv, a.S = h->S_d, a.n_tiles = h->qkv_tiles, a.xop = h->xop_d, a.fp8 = h->fp8, a.wscale = w.s_qkv;
    a.ssq_in = h->ssq_a, a.ssq_n = ssq_n, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
    a.qbuf = h->qbuf, a.knew = h->chunk_k, a.vnew = h->chunk_v, a.q_tiles = h->q_tiles, a.k_tiles = h->k_tiles;
    a.q_dim = h->q_dim, a.kv_dim = h->kv_dim, a.rope_cos = h->rope_cos, a.rope_sin = h->rope_sin, a.state = h->chunk_states;
    for (int m = 0; m < cap; ++m) a.state_rows[m] = h->chunk_states + (m < n ? m : 0);
    RC(gemv(EPI_QKV, a));
    RC(ddk_scatter_kv_rows(h->chunk_k, h->chunk_v, n, h->kv_dim, kc, vc, h->T_cap, h->state, st, h->kv16));
    // causal attention of the chunk rows: row i = a single-query decode attention over the cache up to position
    // pos0 + i - 1 (prefix + the chunk rows ahead of it, just scattered) plus its own new key — the fused-base-pass
    // kernel with every "lane" pointing at this one cache; 16 rows per launch
    for (int r0 = 0; r0 < n; r0 += 16) {
      const int nr = n - r0 < 16 ? n - r0 : 16;
      AttnDecodeArgs t;
      memset(&t, 0, sizeof(t));
      t.wf = h->wf;
      t.qbuf = h->qbuf + (size_t)r0 * h->q_dim, t.T_cap = h->T_cap, t.nb = nr, t.n_heads = h->H, t.n_kv = h->Hkv, t.kv16 = h->kv16;
      t.part_o = h->part_o, t.part_ml = h->part_ml, t.xop_out = h->xop_q + (size_t)(r0 / 8) * h->S_q * 64;
      t.knew = h->chunk_k + (size_t)r0 * h->kv_dim, t.vnew = h->chunk_v + (size_t)r0 * h->kv_dim;
      t.n_lanes = nr, t.max_T = pos0 + n;
      for (int m = 0; m < nr; ++m) {
        t.lane_kc[m] = kc, t.lane_vc[m] = vc, t.lane_state[m] = h->chunk_states + r0 + m;
        t.lane_span_start[m] = h->span_start, t.lane_span_len[m] = h->L;
      }
      RC(ddk_attn_decode(t, st));
    }
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wo, a.S = h->S_q, a.n_tiles = d / 16, a.xop = h->xop_q, a.fp8 = h->fp8, a.wscale = w.s_o;
    a.out = h->xa, a.ldo = d, a.normw_next = w.norm2, a.xop_next = h->xop_d, a.S_next = h->S_d, a.ssq_out = h->ssq_b, a.ssq_ld = d / 16;
    RC(gemv(EPI_RESID, a));
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wgu, a.S = h->S_d, a.n_tiles = dff / 16, a.xop = h->xop_d, a.fp8 = h->fp8, a.wscale = w.s_gu;
    a.ssq_in = h->ssq_b, a.ssq_n = d / 16, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
    a.xop_next = h->xop_ff, a.S_next = h->S_ff;
    RC(gemv(EPI_SILU, a));
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wdown, a.S = h->S_ff, a.n_tiles = d / 16, a.xop = h->xop_ff, a.fp8 = h->fp8, a.wscale = w.s_down;
    a.out = h->xa, a.ldo = d, a.normw_next = (l + 1 < h->Lyr) ? h->lw[l + 1].norm1 : h->final_norm;
    a.xop_next = h->xop_d, a.S_next = h->S_d, a.ssq_out = h->ssq_a, a.ssq_ld = d / 16;
    RC(gemv(EPI_RESID, a));
    ssq_n = d / 16;
  }
  // last row -> final norm -> lm_head (one row through the prefill head: its hidden state stays in h->pq)
  k_prefill_rows<<<1, 256, 0, st>>>(h->row_index, 0, 0, n);
  DD_CHECK_LAUNCH();
  return prefill_head(h, h->row_index, 1, h->last_logits, st, h->xa);
}

// Append n more PROMPT positions (fp32 embeddings [n][d]) to a prefilled (or truncated) sequence: a chunked prefill over
// the rows at positions T .. T+n-1 against the cache, then the greedy first token from the last row exactly as
// dd_lm_prefill emits it.  Row for row the arithmetic is that of a full prefill of the longer prompt.
extern "C" int dd_lm_prefill_extend(dd_lm* h, const float* embeds, int n, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h && embeds && h->prefilled, "dd_lm_prefill_extend: null argument or no prefilled sequence");
  DD_REQUIRE(h->tp_world == 1, "this handle is a tensor-parallel shard: drive it through dd_lm_tp_* (include/dropdec.h)");
  DD_REQUIRE(h->n_tok_host <= 1, "dd_lm_prefill_extend: the sequence has already generated tokens (dd_lm_truncate first)");
  DD_REQUIRE(n >= 1 && h->T_host + n < h->T_cap, "dd_lm_prefill_extend: %d more positions do not fit (length %d, capacity %d)",
             n, h->T_host, h->T_cap);
  const int d = h->d, pos0 = h->T_host;
  if (n <= 32 && g_extend_rows) {
    RC(prefill_extend_rows(h, embeds, n, st));
  } else {
    DD_HIP(hipMemcpyAsync(h->px, embeds, (size_t)n * d * 4, hipMemcpyDeviceToDevice, st));
    RC(prefill_layers(h, n, nullptr, 0, h->span_start, h->L, st, pos0));
    k_prefill_rows<<<1, 256, 0, st>>>(h->row_index, 0, 0, n);          // row_index[0] = n - 1: the last new row
    DD_CHECK_LAUNCH();
    RC(prefill_head(h, h->row_index, 1, h->last_logits, st));
  }
  DD_HIP(hipMemcpyAsync(h->last_hidden, h->pq, (size_t)d * 4, hipMemcpyDeviceToDevice, st));
  RC(dd_argmax_rows(h->last_logits, 1, h->V, h->Vpad, h->argmax_base, st));
  h->tok_host[0] = 0;
  k_state_after_prefill<<<1, 64, 0, st>>>(h->state, pos0 + n, h->argmax_base, h->tokens, h->tok_host_dev);
  DD_CHECK_LAUNCH();
  h->T_host = pos0 + n, h->n_tok_host = 1, h->have_leak = false, h->last_K = 0, h->steps_since_prefill = 0, h->pend_valid = false;
  return DD_OK;
}

// the keep set of a step: overlap with the un-masked pass' argmax (models/llava.py:443-482) or, for "epis_kl", the tokens whose
// prefill distribution is closest to the step's (instructblip.py:483-485)
static int step_keep(dd_lm* h, const int32_t* gate, hipStream_t st) {
  if (h->cfg.mask_mode == DD_MASK_IBLIP_KL)
    return dd_kl_keep_impl(h->base_logits, h->image_logits, h->L, h->V, h->Vpad, 0.1f, h->keep, h->kl_ws, gate, st);
  return dd_overlap_keep_from_argmax(h->argmax_base, h->topk_ids, h->L, h->cfg.k_top, h->keep, gate, st);
}

int dd_engine_step_keep(dd_lm* h, const int32_t* gate, hipStream_t st) { return step_keep(h, gate, st); }
int dd_engine_step_begin(dd_lm* h, hipStream_t st) {
  k_step_begin<<<1, 256, 0, st>>>(h->state, h->leak_bits, h->L, h->cfg.leak_mask == 2 ? 1 : 0);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// the ensemble's vote (or mean) over the K member rows: sets state->winner / voted, member_tok[0] for the mean
static int vote_members(dd_lm* h, int K, hipStream_t st) {
  if (h->cfg.vote_on == DD_VOTE_AVERAGE) {
    // select_by_average (llava.py:37-52): member 0's output with its last-token logits replaced by the fp32 mean
    RC(ddk_mean_rows(h->member_logits, K, h->Vpad, h->V, &h->state->done, st));
    RC(dd_argmax_rows_gated(h->member_logits, 1, h->V, h->Vpad, h->member_tok, &h->state->done, st));
    k_set_winner<<<1, 64, 0, st>>>(h->state, 0, h->member_tok);
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
  const int32_t* ids = h->cfg.vote_on == DD_VOTE_HIDDEN ? h->member_vote : h->member_tok;
  return dd_vote_gated(ids, K, &h->state->winner, &h->state->done, st);
}

// Prefill with the ensemble also applied to the FIRST generated token — the reference's `# if True:` toggle at
// models/llava.py:336-337 (the committed POPE numbers, max_new_tokens=1, were most likely produced
// this way).  After the un-masked pass (uncertainty, top-k ids, keep set from its argmax) every member re-runs the
// WHOLE prompt from an empty cache with its zero columns in the attention mask (llava.py:342-359 on the first forward:
// `original_past_key_values` is the empty cache there); the vote picks the member whose logits and KV cache continue.
// Members run last-to-first so the usual winner (member 0) is the one left in the cache; any other winner is re-run.
extern "C" int dd_lm_prefill_ensemble(dd_lm* h, const float* embeds, int T0, int span_start, int span_len,
                                      const double* mprobs, int K, dd_rng* rng, const float* uniforms, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  RC(dd_lm_prefill(h, embeds, T0, span_start, span_len, stream_));
  if (K == 0) return DD_OK;
  DD_REQUIRE(K >= 1 && K <= MAX_MEMBERS && mprobs, "dd_lm_prefill_ensemble: K=%d out of range (1..%d)", K, MAX_MEMBERS);
  RC(ensure_member_rows(h, K));
  DD_REQUIRE(span_start >= 1, "dd_lm_prefill_ensemble: the visual span must not start at position 0 (a fully masked "
                              "first row has no defined attention; the reference toggle exists for LLaVA only)");
  const int d = h->d, L = span_len;
  RC(dd_overlap_keep_from_argmax(h->argmax_base, h->topk_ids, L, h->cfg.k_top, h->keep, nullptr, st));
  int rng_mode = uniforms ? DD_RNG_INJECTED : DD_RNG_MT19937;
  DD_REQUIRE(h->cfg.mask_mode == DD_MASK_IBLIP_QUANTILE || uniforms || rng, "dd_lm_prefill_ensemble: an rng or uniforms is required");
  RC(dd_sample_masks_impl(h->epi, L, mprobs, K, h->keep, h->cfg.mask_mode, rng_mode, uniforms, dd_rng_state_ptr(rng),
                          h->drop, h->n_drop, nullptr, h->drop_bits, nullptr, st));
  const int32_t* last_row = h->row_index + L;     // == T0 - 1 (set by dd_lm_prefill)
  auto member_pass = [&](int k, bool head) -> int {
    DD_HIP(hipMemcpyAsync(h->px, embeds, (size_t)T0 * d * 4, hipMemcpyDeviceToDevice, st));
    RC(prefill_layers(h, T0, h->drop_bits + (size_t)(k >> 3) * L, k & 7, span_start, span_len, st));
    if (!head) return DD_OK;
    RC(prefill_head(h, last_row, 1, h->member_logits + (size_t)k * h->Vpad, st));
    DD_HIP(hipMemcpyAsync(h->hidden + (size_t)k * d, h->pq, (size_t)d * 4, hipMemcpyDeviceToDevice, st));
    RC(dd_argmax_rows(h->member_logits + (size_t)k * h->Vpad, 1, h->V, h->Vpad, h->member_tok + k, st));
    if (h->cfg.vote_on == DD_VOTE_HIDDEN) RC(dd_argmax_rows(h->hidden + (size_t)k * d, 1, d, d, h->member_vote + k, st));
    return DD_OK;
  };
  for (int k = K - 1; k >= 0; --k) RC(member_pass(k, true));
  RC(vote_members(h, K, st));
  int32_t win[2] = {0, 0};
  DD_HIP(hipMemcpyAsync(win, &h->state->winner, 8, hipMemcpyDeviceToHost, st));
  DD_HIP(hipStreamSynchronize(st));
  DD_REQUIRE(win[0] >= 0 && win[0] < K, "dd_lm_prefill_ensemble: vote returned member %d of %d", win[0], K);
  if (win[0] != 0) RC(member_pass(win[0], false));   // leave the winner's K/V in the cache (llava.py:373)
  k_first_token_from_member<<<1, 1024, 0, st>>>(h->state, h->member_tok, h->member_logits, h->Vpad, h->last_logits,
                                                h->tokens, h->hidden, d, h->last_hidden, h->tok_host_dev);
  DD_CHECK_LAUNCH();
  h->last_K = K;
  return DD_OK;
}

// -----------------------------------------------------------------------------------------------
// one packed sweep of nb rows through all layers + lm_head
// -----------------------------------------------------------------------------------------------
// `lanes` (group step): row m of the pass is the base row of sequence lanes[m] — its own token, position, cache, span
// and leak bits; scratch, weights and the per-layer new K/V rows are this handle's (the first lane of the group).
int lm_sweep(dd_lm* h, int nb, const uint8_t* bits, int row0, float* logits_out, hipStream_t st, dd_lm* const* lanes,
             const int32_t* skip_if) {
  const int d = h->d, dff = h->dff;
  // more than 8 lanes: the base rows fill two (up to 16 lanes), four (32) or eight (64) operand planes and go through the
  // grouped GEMV
  const bool wide = lanes && nb > 8;
  const int lane_groups = nb > 32 ? 8 : (nb > 16 ? 4 : 2);
  auto gemv = [&](int epi, GemvArgs& a) -> int {
    a.skip_if = skip_if;
    if (!wide) return ddk_gemv(epi, a, st);
    a.n_groups = lane_groups, a.nb = 8;
    a.part = h->gemv_part, a.part_floats = h->gemv_part_floats;
    return ddk_gemv_groups(epi, a, st);
  };
  if (lanes) {
    EmbedLanes el;
    memset(&el, 0, sizeof(el));
    for (int m = 0; m < nb; ++m) el.state[m] = lanes[m]->state;
    RC(ddk_embed_rows_lanes(h->embed, d, el, wide ? 8 * lane_groups : 8, h->xa, h->lw[0].norm1, h->xop_d, h->ssq_a, d / 16, st, h->wf));
  } else {
    RC(ddk_embed_rows(h->embed, d, h->state, h->xa, h->lw[0].norm1, h->xop_d, h->ssq_a, d / 16, st, skip_if, h->wf));
  }
  int ssq_n = 1;
  for (int l = 0; l < h->Lyr; ++l) {
    LayerW& w = h->lw[l];
    float* knew = h->knew + ((size_t)l * KV_ROWS + row0) * h->kv_dim;
    float* vnew = h->vnew + ((size_t)l * KV_ROWS + row0) * h->kv_dim;
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wqkv, a.S = h->S_d, a.n_tiles = h->qkv_tiles, a.nb = nb, a.xop = h->xop_d;
    a.fp8 = h->fp8, a.wscale = w.s_qkv;
    a.ssq_in = h->ssq_a, a.ssq_n = ssq_n, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
    a.qbuf = h->qbuf, a.knew = knew, a.vnew = vnew, a.q_tiles = h->q_tiles, a.k_tiles = h->k_tiles;
    a.q_dim = h->q_dim, a.kv_dim = h->kv_dim, a.rope_cos = h->rope_cos, a.rope_sin = h->rope_sin, a.state = h->state;
    if (lanes)
      for (int m = 0; m < nb; ++m) a.state_rows[m] = lanes[m]->state;
    RC(gemv(EPI_QKV, a));
    AttnDecodeArgs t;
    memset(&t, 0, sizeof(t));
    t.wf = h->wf;
    t.qbuf = h->qbuf, t.kc = h->kc + (size_t)l * h->lsk, t.vc = h->vc + (size_t)l * h->lsv, t.T_cap = h->T_cap, t.kv16 = h->kv16;
    t.T = h->T_host, t.state = h->state, t.nb = nb, t.n_heads = h->H, t.n_kv = h->Hkv, t.drop_bits = bits;
    t.bit0 = bits ? h->bit0 : 0;
    t.span_start = h->span_start, t.span_len = h->L, t.part_o = h->part_o, t.part_ml = h->part_ml;
    t.knew = knew, t.vnew = vnew, t.xop_out = h->xop_q, t.skip_if = skip_if;
    if (lanes) {
      // one single-query attention per lane over its own cache, 16 rows per launch
      for (int r0 = 0; r0 < nb; r0 += 16) {
        const int nr = nb - r0 < 16 ? nb - r0 : 16;
        AttnDecodeArgs u = t;
        u.qbuf = h->qbuf + (size_t)r0 * h->q_dim, u.nb = nr, u.n_lanes = nr, u.bit0 = 0, u.drop_bits = nullptr, u.max_T = 0;
        u.knew = knew + (size_t)r0 * h->kv_dim, u.vnew = vnew + (size_t)r0 * h->kv_dim;
        u.xop_out = h->xop_q + (size_t)(r0 / 8) * h->S_q * 64;
        for (int m = 0; m < nr; ++m) {
          dd_lm* q = lanes[r0 + m];
          u.lane_kc[m] = q->kc + (size_t)l * q->lsk, u.lane_vc[m] = q->vc + (size_t)l * q->lsv, u.lane_state[m] = q->state;
          u.lane_bits[m] = q->cfg.leak_mask ? q->leak_bits : nullptr;
          u.lane_span_start[m] = q->span_start, u.lane_span_len[m] = q->L;
          if (q->T_host > u.max_T) u.max_T = q->T_host;
        }
        RC(ddk_attn_decode(u, st));
      }
    } else {
      RC(ddk_attn_decode(t, st));
    }
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wo, a.S = h->S_q, a.n_tiles = d / 16, a.nb = nb, a.xop = h->xop_q;
    a.fp8 = h->fp8, a.wscale = w.s_o;
    a.out = h->xa, a.ldo = d, a.normw_next = w.norm2, a.xop_next = h->xop_d, a.S_next = h->S_d, a.ssq_out = h->ssq_b, a.ssq_ld = d / 16;
    RC(gemv(EPI_RESID, a));
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wgu, a.S = h->S_d, a.n_tiles = dff / 16, a.nb = nb, a.xop = h->xop_d;
    a.fp8 = h->fp8, a.wscale = w.s_gu;
    a.ssq_in = h->ssq_b, a.ssq_n = d / 16, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps, a.xop_next = h->xop_ff, a.S_next = h->S_ff;
    RC(gemv(EPI_SILU, a));
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wdown, a.S = h->S_ff, a.n_tiles = d / 16, a.nb = nb, a.xop = h->xop_ff;
    a.fp8 = h->fp8, a.wscale = w.s_down;
    a.out = h->xa, a.ldo = d, a.normw_next = (l + 1 < h->Lyr) ? h->lw[l + 1].norm1 : h->final_norm;
    a.xop_next = h->xop_d, a.S_next = h->S_d, a.ssq_out = h->ssq_a, a.ssq_ld = d / 16;
    RC(gemv(EPI_RESID, a));
    ssq_n = d / 16;
  }
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.wf = h->wf;
  a.W = h->lm_head, a.S = h->S_d, a.n_tiles = h->Vpad / 16, a.nb = nb, a.xop = h->xop_d;
  a.fp8 = h->fp8, a.wscale = h->s_lm;
  a.ssq_in = h->ssq_a, a.ssq_n = ssq_n, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
  a.out = logits_out, a.ldo = h->Vpad, a.n_valid = h->V;
  if (!lanes) a.state = h->state;      // a finished sequence keeps the logits of its EOS step (the group's fused base rows go
  RC(gemv(EPI_STORE, a));              // to scratch and are handed out by k_scatter_base, which checks per sequence)
  return DD_OK;
}

// -----------------------------------------------------------------------------------------------
// decode step, phased
// -----------------------------------------------------------------------------------------------
extern "C" int dd_lm_step_base(dd_lm* h, const double* mprobs, int K, dd_rng* rng, const float* uniforms, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h, "dd_lm_step_base: null handle");
  DD_REQUIRE(h->tp_world == 1, "this handle is a tensor-parallel shard: drive it through dd_lm_tp_* (include/dropdec.h)");
  if (!h->prefilled) {
    dd_set_error("dd_lm_step_base: decode before prefill");
    return DD_ESTATE;
  }
  DD_REQUIRE(K >= 0 && K <= MAX_MEMBERS, "dd_lm_step: K=%d out of range (0..%d)", K, MAX_MEMBERS);
  RC(ensure_member_rows(h, K));
  DD_REQUIRE(K == 0 || mprobs, "dd_lm_step: mprobs required");
  if (h->T_host + 1 >= h->T_cap) {
    dd_set_error("dd_lm_step: KV cache full (%d tokens)", h->T_cap);
    return DD_ESTATE;
  }
  DD_REQUIRE(h->n_tok_host < MAX_NEW_TOKENS, "dd_lm_step: token buffer full");
  // leak_bits are all zero until the first dropout step, so passing them unconditionally is equivalent and keeps the
  // launch arguments identical from step to step (graph replay)
  h->pend_valid = false;       // a row parked by a rider group step is void once the sequence is stepped any other way
  k_step_begin<<<1, 256, 0, st>>>(h->state, h->leak_bits, h->L, h->cfg.leak_mask == 2 ? 1 : 0);
  DD_CHECK_LAUNCH();
  const uint8_t* base_bits = h->cfg.leak_mask ? h->leak_bits : nullptr;
  h->bit0 = 0;
  RC(lm_sweep(h, 1, base_bits, 0, h->base_logits, st));
  const int32_t* gate = &h->state->done;
  RC(dd_argmax_rows_gated(h->base_logits, 1, h->V, h->Vpad, h->argmax_base, gate, st));
  h->last_K = K;
  if (K == 0) return DD_OK;
  RC(step_keep(h, gate, st));
  int rng_mode = uniforms ? DD_RNG_INJECTED : DD_RNG_MT19937;
  DD_REQUIRE(h->cfg.mask_mode == DD_MASK_IBLIP_QUANTILE || uniforms || rng, "dd_lm_step: an rng or uniforms is required");
  RC(dd_sample_masks_impl(h->epi, h->L, mprobs, K, h->keep, h->cfg.mask_mode, rng_mode, uniforms,
                          dd_rng_state_ptr(rng), h->drop, h->n_drop, nullptr, h->drop_bits, gate, st));
  return DD_OK;
}

extern "C" int dd_lm_step_members(dd_lm* h, int m_lo, int m_hi, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h && h->prefilled, "dd_lm_step_members: bad state");
  DD_REQUIRE(h->tp_world == 1, "this handle is a tensor-parallel shard: drive it through dd_lm_tp_* (include/dropdec.h)");
  DD_REQUIRE(m_lo >= 0 && m_lo <= m_hi && m_hi <= h->last_K, "dd_lm_step_members: range [%d,%d) outside K=%d", m_lo,
             m_hi, h->last_K);
  for (int g0 = m_lo; g0 < m_hi;) {
    int g1 = ((g0 >> 3) + 1) * 8;
    if (g1 > m_hi) g1 = m_hi;
    int nb = g1 - g0;
    // rows of this pass are members g0..g1-1: row m reads bit (g0 & 7) + m of plane g0 >> 3
    const uint8_t* bits = h->drop_bits + (size_t)(g0 >> 3) * h->L;
    h->bit0 = g0 & 7;
    RC(lm_sweep(h, nb, bits, g0, h->member_logits + (size_t)g0 * h->Vpad, st));
    RC(dd_argmax_rows_gated(h->member_logits + (size_t)g0 * h->Vpad, nb, h->V, h->Vpad, h->member_tok + g0, &h->state->done, st));
    if (h->cfg.vote_on == DD_VOTE_HIDDEN) {
      float* hid = h->hidden + (size_t)g0 * h->d;
      RC(ddk_final_norm_rows(h->xa, nb, h->d, h->final_norm, h->cfg.rms_eps, hid, st));
      RC(dd_argmax_rows_gated(hid, nb, h->d, h->d, h->member_vote + g0, &h->state->done, st));
    }
    g0 = g1;
  }
  return DD_OK;
}

extern "C" int dd_lm_step_commit(dd_lm* h, int K, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h && h->prefilled && K == h->last_K, "dd_lm_step_commit: bad state (K=%d, expected %d)", K, h ? h->last_K : -1);
  if (K > 0) RC(vote_members(h, K, st));
  RC(ddk_commit_kv(h->commit_k ? h->commit_k : h->knew, h->commit_v ? h->commit_v : h->vnew, h->Lyr, KV_ROWS,
                   h->kv_dim, h->kc, h->vc, h->lsk, h->lsv, h->T_cap, h->state, K > 0 ? 1 : 0, st, h->kv16));
  k_step_end<<<1, 1024, 0, st>>>(h->state, K, h->argmax_base, h->member_tok, h->base_logits, h->member_logits, h->Vpad,
                                 h->last_logits, h->tokens, h->drop_bits, h->L, h->leak_bits, h->cfg.leak_mask, h->hidden,
                                 h->d, h->last_hidden, h->tok_host_dev);
  DD_CHECK_LAUNCH();
  h->T_host += 1;
  h->n_tok_host += 1;
  h->pend_valid = false;       // (the rider step re-parks its ring leaders after its sweeps; every other caller leaves nothing parked)
  if (h->cfg.leak_mask && K > 0) h->have_leak = true;
  return DD_OK;
}

// -----------------------------------------------------------------------------------------------
// one multi-group sweep: the K members of ng = 2 or 4 sequences (group g = rows 8g..8g+K-1) against ONE pass over the
// weights.  Scratch is `h`'s (the group's first lane); logits, new K/V rows and hidden rows land in each sequence's own
// buffers, so that vote and commit run per sequence exactly as after lm_sweep.
// -----------------------------------------------------------------------------------------------
static int g_use_graph = 1;    // dd_set_tuning key 8
// bumped by every tuning call and mixed into every graph key: a step captured under other settings is never replayed
static unsigned long long g_tune_epoch = 0;
void dd_engine_bump_epoch() { ++g_tune_epoch; }
unsigned long long dd_engine_epoch() { return g_tune_epoch; }
static int g_pair_sweeps = 8;  // dd_set_tuning key 9: sequences per member sweep in dd_lm_group_step (0/1: one, 2, 4, 8)
void dd_engine_set_graph(int on) { g_use_graph = on; }
int dd_engine_use_graph() { return g_use_graph; }
void dd_engine_set_pairs(int on) { g_pair_sweeps = on; }
static int g_branches = 2;     // dd_tools_set_tuning key 23: member sweeps of a group step that run concurrently (1..4)
static int g_fp32_fork = 0;    // dd_tools_set_tuning key 37: branches for fp32-cache engines too (round 3 saw lanes differ from their solo runs there)
void dd_engine_set_fp32_fork(int on) { g_fp32_fork = on; }
// dd_tools_set_tuning key 40 (debug; set before the first group step of a handle): the two branches of a classic group step on streams with DISJOINT
// CU masks (branch 0 on a masked stream of its own instead of the caller's): concurrent, but never two branches' workgroups on one CU
static int g_mask_branches = 0;
void dd_engine_set_mask_branches(int on) { g_mask_branches = on; }
// dd_tools_set_tuning key 41 (debug): the attention launches of a multi-group sweep on a CU-masked stream of the sweeping handle's own (fenced by
// events against the sweep's stream) — 1: the first handle that asks gets the lower 128 CUs, the second the upper 128 (the two branches' attentions
// never share a CU with EACH OTHER, but each still meets the other branch's GEMVs); 2: both on the lower 128
static int g_attn_masked = 0;
void dd_engine_set_attn_masked(int mode) { g_attn_masked = mode; }
struct AttnSide {
  hipStream_t st = nullptr;
  hipEvent_t e_in = nullptr, e_out = nullptr;
};
// dd_tools_set_tuning key 42 (debug, with key 40 = 1: branches on disjoint CU halves): bit mask of kernel families of a multi-group sweep that run
// on an UNMASKED stream instead (the whole chip: their workgroups may then share a CU with the other branch's kernels) — 1 qkv GEMV, 2 o_proj,
// 4 gate/up, 8 down, 16 lm_head, 32 embed, 64 the finishing launches (argmax, vote, commit, step end), 128 attention
static int g_unmask = 0;
void dd_engine_set_unmask(int m) { g_unmask = m; }
static std::vector<std::pair<dd_lm*, AttnSide>> g_full_sides;
static int full_side_for(dd_lm* h, AttnSide** out) {
  for (auto& p : g_full_sides)
    if (p.first == h) {
      *out = &p.second;
      return DD_OK;
    }
  AttnSide a;
  DD_HIP(hipStreamCreateWithFlags(&a.st, hipStreamNonBlocking));
  DD_HIP(hipEventCreateWithFlags(&a.e_in, hipEventDisableTiming));
  DD_HIP(hipEventCreateWithFlags(&a.e_out, hipEventDisableTiming));
  g_full_sides.push_back({h, a});
  *out = &g_full_sides.back().second;
  return DD_OK;
}
// run `body(stream)` on the handle's unmasked stream when family `bit` is selected (fenced against `st`), else on `st`
template <typename F>
static int on_family(dd_lm* h, hipStream_t st, int bit, F&& body) {
  if (!(g_unmask & bit)) return body(st);
  AttnSide* fs = nullptr;
  RC(full_side_for(h, &fs));
  DD_HIP(hipEventRecord(fs->e_in, st));
  DD_HIP(hipStreamWaitEvent(fs->st, fs->e_in, 0));
  RC(body(fs->st));
  DD_HIP(hipEventRecord(fs->e_out, fs->st));
  DD_HIP(hipStreamWaitEvent(st, fs->e_out, 0));
  return DD_OK;
}
static std::vector<std::pair<dd_lm*, AttnSide>> g_attn_sides;
static int attn_side_for(dd_lm* h, AttnSide** out) {
  for (auto& p : g_attn_sides)
    if (p.first == h) {
      *out = &p.second;
      return DD_OK;
    }
  AttnSide a;
  uint32_t mask[8];
  const bool upper = g_attn_masked == 1 && (g_attn_sides.size() & 1);
  for (int w = 0; w < 8; ++w) mask[w] = (upper == (w >= 4)) ? 0xFFFFFFFFu : 0u;
  DD_HIP(hipExtStreamCreateWithCUMask(&a.st, 8, mask));
  DD_HIP(hipEventCreateWithFlags(&a.e_in, hipEventDisableTiming));
  DD_HIP(hipEventCreateWithFlags(&a.e_out, hipEventDisableTiming));
  g_attn_sides.push_back({h, a});
  *out = &g_attn_sides.back().second;
  return DD_OK;
}
void dd_engine_set_branches(int n) { g_branches = n < 1 ? 1 : (n > 4 ? 4 : n); }
// the same for the rider form (rings of at least two groups each): 64 lanes 42.5 / 38.2 / 36.9 ms per step with 2 / 3 / 4 branches
// (tools/rider_ab.py), where the classic form gained nothing beyond two; dd_tools_set_tuning key 28
static int g_rider_branches = 4;
void dd_engine_set_rider_branches(int n) { g_rider_branches = n < 1 ? 1 : (n > 4 ? 4 : n); }
static int g_ride_beside = 1;    // dd_tools_set_tuning key 27: the riding rows' attention in the members' launches (0: launches of its own)
void dd_engine_set_ride_beside(int on) { g_ride_beside = on; }

// ---- debug: per-stage checksums of a multi-group sweep (libdropdec_tools.so sets the buffer; tools/race_bisect.py) ----------------------
// trace[(sweep * n_layers + layer) * 16 + stage] += order-independent 32-bit sums of the stage's output (0 embed, 1 q rows, 2 new K rows,
// 3 attention -> o_proj operand, 4 o_proj -> residual rows, 5 gate/up -> down operand, 6 down -> residual rows, 7 down -> next operand,
// 8 / 9 the attention's per-tile statistics / outputs (the tiles every sequence of the sweep has), 10 new V rows)
uint32_t* g_dbg_trace = nullptr;
int g_dbg_trace_cap = 0, g_dbg_sweeps = 0;
int g_dbg_replay = 0;                       // dd_tools_sweep_trace flag: launch every traced attention twice (stages 11 / 12 = the second launch's tiles)
uint32_t* g_dbg_attn = nullptr;            // per-workgroup checksums of the fp32-cache attention tile pass: [sweep][layer][g_dbg_attn_stride]
size_t g_dbg_attn_stride = 0;
__global__ __launch_bounds__(256) void k_dbg_sum(const uint32_t* __restrict__ p, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i] * (uint32_t)(2 * i + 1);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
__global__ __launch_bounds__(256) void k_dbg_sum_tiles(const uint32_t* __restrict__ p, int n_kv, int splits_grid, int row_words, int tiles,
                                                       uint32_t* __restrict__ out) {
  const size_t n = (size_t)n_kv * tiles * row_words;
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t kvh = i / ((size_t)tiles * row_words), rem = i % ((size_t)tiles * row_words);
    acc += p[(kvh * splits_grid + rem / row_words) * row_words + rem % row_words] * (uint32_t)(2 * i + 1);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
static void dbg_sum(int sweep, int n_layers, int layer, int stage, const void* p, size_t bytes, hipStream_t st) {
  if (!g_dbg_trace || sweep >= g_dbg_trace_cap) return;
  k_dbg_sum<<<64, 256, 0, st>>>((const uint32_t*)p, bytes / 4, g_dbg_trace + ((size_t)sweep * n_layers + layer) * 16 + stage);
}

#define RIDER_KV_ROW0 16   // rows of the sweeping handle's new-K/V scratch that hold the riding un-masked rows (0..15: members)
// rider / n_rider (ng == 8 only): up to 8 sequences whose UN-MASKED rows ride in a ninth operand plane of the sweep (row 64 + m =
// sequence rider[m]); their logits go to rows 0.. of h->grp_logits.  See group_step_rider.
// packed (K <= 4, ng == 16): half planes — plane p carries the members of sequences 2 p (rows 0..3) and 2 p + 1 (rows 4..7), so a 64-row
// sweep serves sixteen sequences; per-sequence buffers and attention arguments are indexed by sequence (GemvArgs / AttnDecodeArgs
// half_planes), every row goes through the kernels of the plain pass.
static int lm_sweep_groups(dd_lm* h, dd_lm* const* qs, int ng, int K, hipStream_t st, dd_lm* const* rider = nullptr, int n_rider = 0,
                           bool packed = false) {
  const int d = h->d, dff = h->dff;
  const int planes = n_rider > 0 ? 9 : (packed ? 8 : ng);
  const int hp = packed ? (n_rider > 0 ? 7 : 8) : 0;                  // half planes (with riders: seven of them + two riding planes)
  const int ride_plane0 = packed ? 7 : 8, ride_row0 = 8 * ride_plane0, ride_slot0 = 2 * hp + (ride_plane0 - hp);
  const int rows_g = packed ? 4 : 8;                                  // rows a sequence owns
  auto row0 = [&](int g) -> int { return packed ? 8 * (g >> 1) + 4 * (g & 1) : 8 * g; };
  EmbedLanes el;
  memset(&el, 0, sizeof(el));
  for (int g = 0; g < ng; ++g)
    for (int m = 0; m < K; ++m) el.state[row0(g) + m] = qs[g]->state;
  for (int m = 0; m < n_rider; ++m) el.state[ride_row0 + m] = rider[m]->state;
  RC(on_family(h, st, 32, [&](hipStream_t s2) { return ddk_embed_rows_lanes(h->embed, d, el, 8 * planes, h->xa, h->lw[0].norm1, h->xop_d, h->ssq_a, d / 16, s2, h->wf); }));
  const int dbg_sweep = g_dbg_trace ? g_dbg_sweeps++ : 0;
  dbg_sum(dbg_sweep, h->Lyr, 0, 0, h->xop_d, (size_t)planes * h->S_d * 1024, st);
  int ssq_n = 1;
  for (int l = 0; l < h->Lyr; ++l) {
    LayerW& w = h->lw[l];
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wqkv, a.S = h->S_d, a.n_tiles = h->qkv_tiles, a.nb = K, a.xop = h->xop_d, a.n_groups = planes, a.nb_rider = n_rider, a.half_planes = hp;
    a.fp8 = h->fp8, a.wscale = w.s_qkv;
    a.ssq_in = h->ssq_a, a.ssq_n = ssq_n, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
    a.qbuf = h->qbuf, a.q_tiles = h->q_tiles, a.k_tiles = h->k_tiles;
    a.q_dim = h->q_dim, a.kv_dim = h->kv_dim, a.rope_cos = h->rope_cos, a.rope_sin = h->rope_sin, a.state = qs[0]->state;
    AttnDecodeArgs t;
    memset(&t, 0, sizeof(t));
    t.wf = h->wf;
    t.qbuf = h->qbuf, t.T_cap = h->T_cap, t.nb = K, t.n_heads = h->H, t.n_kv = h->Hkv, t.bit0 = 0, t.kv16 = h->kv16;
    t.part_o = h->part_o, t.part_ml = h->part_ml, t.xop_out = h->xop_q;
    t.n_lanes = packed ? 8 : ng, t.lane_groups = packed ? 8 : ng, t.half_planes = packed ? 1 : 0;
    if (g_dbg_attn && g_dbg_trace && dbg_sweep < g_dbg_trace_cap) t.dbg = g_dbg_attn + ((size_t)dbg_sweep * h->Lyr + l) * g_dbg_attn_stride;
    for (int g = 0; g < ng; ++g) {
      dd_lm* q = qs[g];
      float* kn = q->knew + (size_t)l * KV_ROWS * q->kv_dim;
      float* vn = q->vnew + (size_t)l * KV_ROWS * q->kv_dim;
      a.knew_g[g] = kn, a.vnew_g[g] = vn, t.knew_g[g] = kn, t.vnew_g[g] = vn;
      for (int m = 0; m < rows_g; ++m) a.state_rows[row0(g) + m] = q->state;
      t.lane_kc[g] = q->kc + (size_t)l * q->lsk, t.lane_vc[g] = q->vc + (size_t)l * q->lsv, t.lane_state[g] = q->state;
      t.lane_bits[g] = q->drop_bits, t.lane_span_start[g] = q->span_start, t.lane_span_len[g] = q->L;
      if (q->T_host > t.max_T) t.max_T = q->T_host;
    }
    a.knew = a.knew_g[0], a.vnew = a.vnew_g[0], t.knew = t.knew_g[0], t.vnew = t.vnew_g[0];
    a.part = h->gemv_part, a.part_floats = h->gemv_part_floats;
    float* rk = h->knew + ((size_t)l * KV_ROWS + RIDER_KV_ROW0) * h->kv_dim;
    float* rv = h->vnew + ((size_t)l * KV_ROWS + RIDER_KV_ROW0) * h->kv_dim;
    if (n_rider) {
      a.knew_g[ride_slot0] = rk, a.vnew_g[ride_slot0] = rv;
      if (n_rider > 8) a.knew_g[ride_slot0 + 1] = rk + (size_t)8 * h->kv_dim, a.vnew_g[ride_slot0 + 1] = rv + (size_t)8 * h->kv_dim;
      for (int m = 0; m < n_rider; ++m) a.state_rows[ride_row0 + m] = rider[m]->state;
    }
    RC(on_family(h, st, 1, [&](hipStream_t s2) { return ddk_gemv_groups(EPI_QKV, a, s2); }));
    dbg_sum(dbg_sweep, h->Lyr, l, 1, h->qbuf, (size_t)8 * planes * h->q_dim * 4, st);
    for (int g = 0; g < ng && g_dbg_trace; ++g) dbg_sum(dbg_sweep, h->Lyr, l, 2, a.knew_g[g], (size_t)rows_g * h->kv_dim * 4, st);
    if (n_rider) {
      // the riding rows: one single-query attention per sequence over its own cache (the fused base pass's form), into plane 8
      AttnDecodeArgs u;
      memset(&u, 0, sizeof(u));
      u.wf = h->wf, u.T_cap = h->T_cap, u.kv16 = h->kv16, u.n_heads = h->H, u.n_kv = h->Hkv;
      u.part_o = h->part_o_ride, u.part_ml = h->part_ml_ride;
      u.qbuf = h->qbuf + (size_t)ride_row0 * h->q_dim, u.nb = n_rider, u.n_lanes = n_rider;
      u.knew = rk, u.vnew = rv, u.xop_out = h->xop_q + (size_t)ride_plane0 * h->S_q * 64;
      for (int m = 0; m < n_rider; ++m) {
        dd_lm* q = rider[m];
        u.lane_kc[m] = q->kc + (size_t)l * q->lsk, u.lane_vc[m] = q->vc + (size_t)l * q->lsv, u.lane_state[m] = q->state;
        u.lane_bits[m] = q->cfg.leak_mask ? q->leak_bits : nullptr;
        u.lane_span_start[m] = q->span_start, u.lane_span_len[m] = q->L;
        if (q->T_host > u.max_T) u.max_T = q->T_host;
      }
      if (g_ride_beside || packed) { // beside the members' attention: one partial + one combine launch for both
        RC(ddk_attn_decode_ride(t, u, packed ? 7 : 8, st));
      } else {
        RC(ddk_attn_decode(t, st));
        RC(ddk_attn_decode(u, st));
      }
    } else if (g_attn_masked) {
      AttnSide* as = nullptr;
      RC(attn_side_for(h, &as));
      DD_HIP(hipEventRecord(as->e_in, st));
      DD_HIP(hipStreamWaitEvent(as->st, as->e_in, 0));
      RC(ddk_attn_decode(t, as->st));
      DD_HIP(hipEventRecord(as->e_out, as->st));
      DD_HIP(hipStreamWaitEvent(st, as->e_out, 0));
    } else {
      RC(on_family(h, st, 128, [&](hipStream_t s2) { return ddk_attn_decode(t, s2); }));
    }
    if (g_dbg_trace && dbg_sweep < g_dbg_trace_cap && !n_rider && g_dbg_replay) {
      // (debug) the same attention launched a second time into spare buffers: do two launches over the same inputs agree with each other?
      static std::vector<std::pair<dd_lm*, float*>> spare;
      float* sp = nullptr;
      for (auto& e : spare) sp = e.first == h ? e.second : sp;
      const size_t n_o = (size_t)h->Hkv * (h->T_cap / 64) * GROUP_ROWS * (h->H / h->Hkv) * 128, n_ml = n_o / 64, n_x = (size_t)h->S_q * 64 * GROUP_PLANES * 4;
      if (!sp) {
        DD_HIP(hipMalloc((void**)&sp, (n_o + n_ml + n_x) * 4));
        spare.push_back({h, sp});
      }
      AttnDecodeArgs t2 = t;
      t2.part_o = sp, t2.part_ml = sp + n_o, t2.xop_out = (u32x4_t*)(sp + n_o + n_ml), t2.dbg = nullptr;
      RC(ddk_attn_decode(t2, st));
      int minT2 = 1 << 30;
      for (int g = 0; g < ng; ++g) minT2 = qs[g]->T_host < minT2 ? qs[g]->T_host : minT2;
      const int tiles2 = (minT2 + 63) / 64, sg2 = ddk_attn_grid_tiles(t.max_T, h->T_cap), RT2 = 8 * (packed ? 8 : ng) * (h->H / h->Hkv);
      uint32_t* tr2 = g_dbg_trace + ((size_t)dbg_sweep * h->Lyr + l) * 16;
      k_dbg_sum_tiles<<<64, 256, 0, st>>>((const uint32_t*)t2.part_ml, h->Hkv, sg2, RT2 * 2, tiles2, tr2 + 11);
      k_dbg_sum_tiles<<<64, 256, 0, st>>>((const uint32_t*)t2.part_o, h->Hkv, sg2, RT2 * 128, tiles2, tr2 + 12);
    }
    if (g_dbg_trace && dbg_sweep < g_dbg_trace_cap && !n_rider) {
      int minT = 1 << 30;
      for (int g = 0; g < ng; ++g) minT = qs[g]->T_host < minT ? qs[g]->T_host : minT;
      const int tiles = (minT + 63) / 64, sg = ddk_attn_grid_tiles(t.max_T, h->T_cap), RT = 8 * (packed ? 8 : ng) * (h->H / h->Hkv);
      uint32_t* tr = g_dbg_trace + ((size_t)dbg_sweep * h->Lyr + l) * 16;
      k_dbg_sum_tiles<<<64, 256, 0, st>>>((const uint32_t*)h->part_ml, h->Hkv, sg, RT * 2, tiles, tr + 8);
      k_dbg_sum_tiles<<<64, 256, 0, st>>>((const uint32_t*)h->part_o, h->Hkv, sg, RT * 128, tiles, tr + 9);
      for (int g = 0; g < ng; ++g) dbg_sum(dbg_sweep, h->Lyr, l, 10, t.vnew_g[g], (size_t)rows_g * h->kv_dim * 4, st);
    }
    dbg_sum(dbg_sweep, h->Lyr, l, 3, h->xop_q, (size_t)planes * h->S_q * 1024, st);
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wo, a.S = h->S_q, a.n_tiles = d / 16, a.nb = K, a.xop = h->xop_q, a.n_groups = planes, a.nb_rider = n_rider, a.half_planes = hp;
    a.fp8 = h->fp8, a.wscale = w.s_o;
    a.out = h->xa, a.ldo = d, a.normw_next = w.norm2, a.xop_next = h->xop_d, a.S_next = h->S_d, a.ssq_out = h->ssq_b, a.ssq_ld = d / 16;
    a.part = h->gemv_part, a.part_floats = h->gemv_part_floats;
    RC(on_family(h, st, 2, [&](hipStream_t s2) { return ddk_gemv_groups(EPI_RESID, a, s2); }));
    dbg_sum(dbg_sweep, h->Lyr, l, 4, h->xa, (size_t)8 * planes * d * 4, st);
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wgu, a.S = h->S_d, a.n_tiles = dff / 16, a.nb = K, a.xop = h->xop_d, a.n_groups = planes, a.nb_rider = n_rider, a.half_planes = hp;
    a.fp8 = h->fp8, a.wscale = w.s_gu;
    a.ssq_in = h->ssq_b, a.ssq_n = d / 16, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
    a.xop_next = h->xop_ff, a.S_next = h->S_ff;
    a.part = h->gemv_part, a.part_floats = h->gemv_part_floats;
    RC(on_family(h, st, 4, [&](hipStream_t s2) { return ddk_gemv_groups(EPI_SILU, a, s2); }));
    dbg_sum(dbg_sweep, h->Lyr, l, 5, h->xop_ff, (size_t)planes * h->S_ff * 1024, st);
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = w.wdown, a.S = h->S_ff, a.n_tiles = d / 16, a.nb = K, a.xop = h->xop_ff, a.n_groups = planes, a.nb_rider = n_rider, a.half_planes = hp;
    a.fp8 = h->fp8, a.wscale = w.s_down;
    a.out = h->xa, a.ldo = d, a.normw_next = (l + 1 < h->Lyr) ? h->lw[l + 1].norm1 : h->final_norm;
    a.xop_next = h->xop_d, a.S_next = h->S_d, a.ssq_out = h->ssq_a, a.ssq_ld = d / 16;
    a.part = h->gemv_part, a.part_floats = h->gemv_part_floats;
    RC(on_family(h, st, 8, [&](hipStream_t s2) { return ddk_gemv_groups(EPI_RESID, a, s2); }));
    dbg_sum(dbg_sweep, h->Lyr, l, 6, h->xa, (size_t)8 * planes * d * 4, st);
    dbg_sum(dbg_sweep, h->Lyr, l, 7, h->xop_d, (size_t)planes * h->S_d * 1024, st);
    ssq_n = d / 16;
  }
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.wf = h->wf;
  a.W = h->lm_head, a.S = h->S_d, a.n_tiles = h->Vpad / 16, a.nb = K, a.xop = h->xop_d, a.n_groups = planes, a.nb_rider = n_rider, a.half_planes = hp;
  a.fp8 = h->fp8, a.wscale = h->s_lm;
  a.ssq_in = h->ssq_a, a.ssq_n = ssq_n, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
  for (int g = 0; g < ng; ++g) {
    a.out_g[g] = qs[g]->member_logits;
    for (int m = 0; m < rows_g; ++m) a.state_rows[row0(g) + m] = qs[g]->state;
  }
  a.out = qs[0]->member_logits, a.ldo = h->Vpad, a.n_valid = h->V;
  a.part = h->gemv_part, a.part_floats = h->gemv_part_floats;
  if (n_rider) {
    a.out_g[ride_slot0] = h->grp_logits;
    if (n_rider > 8) a.out_g[ride_slot0 + 1] = h->grp_logits + (size_t)8 * h->Vpad;
    for (int m = 0; m < n_rider; ++m) a.state_rows[ride_row0 + m] = rider[m]->state;
  }
  RC(on_family(h, st, 16, [&](hipStream_t s2) { return ddk_gemv_groups(EPI_STORE, a, s2); }));
  return DD_OK;
}

// what follows a multi-group sweep: member argmax, vote, winner's K/V appended, token emitted — for the ng sequences of
// the sweep with ONE launch per stage (dd_lm_step_commit's work, block = sequence)
// (packed_base >= 0: the sequences are numbers packed_base.. of a half-plane sweep — their rows in h->xa start at 8 (s >> 1) + 4 (s & 1))
int (*dd_engine_group_finish_hook)(dd_lm* const* qs, int ng, int K, hipStream_t st) = nullptr;
static int group_finish(dd_lm* h, dd_lm* const* qs, int ng, int K, hipStream_t st, int packed_base = -1) {
  const int d = h->d;
  const float* lg[8];
  int32_t* tk[8];
  const int32_t* gates[8];
  for (int g = 0; g < ng; ++g) lg[g] = qs[g]->member_logits, tk[g] = qs[g]->member_tok, gates[g] = &qs[g]->state->done;
  RC(dd_argmax_rows_lanes(lg, tk, gates, ng, K, h->V, h->Vpad, st));
  bool plain_vote = true;
  for (int g = 0; g < ng; ++g) {
    dd_lm* q = qs[g];
    if (q->cfg.vote_on == DD_VOTE_HIDDEN) {    // InstructBLIP: argmax over the final-normed hidden state (instructblip.py:125-137)
      const int sq = packed_base + g, xrow = packed_base >= 0 ? 8 * (sq >> 1) + 4 * (sq & 1) : 8 * g;
      RC(ddk_final_norm_rows(h->xa + (size_t)xrow * d, K, d, h->final_norm, h->cfg.rms_eps, q->hidden, st));
      RC(dd_argmax_rows_gated(q->hidden, K, d, d, q->member_vote, &q->state->done, st));
    }
    plain_vote &= q->cfg.vote_on != DD_VOTE_AVERAGE;
  }
  if (!plain_vote) {                            // select_by_average: per sequence, as dd_lm_step_commit does it
    for (int g = 0; g < ng; ++g) {
      RC(dd_lm_step_commit(qs[g], K, st));
      qs[g]->steps_since_prefill++;
    }
    return DD_OK;
  }
  const int32_t* ids[8];
  int32_t* out2[8];
  CommitLanes cl;
  StepEndLanes el;
  memset(&cl, 0, sizeof(cl));
  memset(&el, 0, sizeof(el));
  cl.lsk = h->lsk, cl.lsv = h->lsv, cl.kv16 = h->kv16;
  for (int g = 0; g < ng; ++g) {
    dd_lm* q = qs[g];
    ids[g] = q->cfg.vote_on == DD_VOTE_HIDDEN ? q->member_vote : q->member_tok;
    out2[g] = &q->state->winner;
    cl.knew[g] = q->knew, cl.vnew[g] = q->vnew, cl.kc[g] = q->kc, cl.vc[g] = q->vc, cl.state[g] = q->state;
    el.st[g] = q->state, el.member_tok[g] = q->member_tok, el.member_logits[g] = q->member_logits;
    el.last_logits[g] = q->last_logits, el.tokens[g] = q->tokens, el.drop_bits[g] = q->drop_bits, el.leak_bits[g] = q->leak_bits;
    el.mirror[g] = q->tok_host_dev, el.L[g] = q->L, el.leak[g] = q->cfg.leak_mask ? 1 : 0;
  }
  RC(dd_vote_lanes(ids, out2, gates, ng, K, st));
  RC(ddk_commit_kv_lanes(cl, ng, h->Lyr, KV_ROWS, h->kv_dim, h->T_cap, st));
  k_step_end_lanes<<<ng, 1024, 0, st>>>(el, K, h->Vpad);
  DD_CHECK_LAUNCH();
  for (int g = 0; g < ng; ++g) {
    dd_lm* q = qs[g];
    q->T_host += 1, q->n_tok_host += 1, q->steps_since_prefill++;
    if (q->cfg.leak_mask) q->have_leak = true;
  }
  if (dd_engine_group_finish_hook) RC(dd_engine_group_finish_hook(qs, ng, K, st));   // libdropdec_tools.so: per-step trace records
  return DD_OK;
}

// -----------------------------------------------------------------------------------------------
// group step: one decode step for each of n sequences that share weights, with the n un-masked base passes packed into
// ONE sweep (row m = sequence m).  Each sequence then samples its masks from ITS OWN rng stream and runs its K members
// exactly as dd_lm_decode_step does, so every sequence's tokens, masks and logits are those of a run on its own.
// Per step and sequence the weights are read 1/n + 1 times instead of twice.
// -----------------------------------------------------------------------------------------------
struct ScatterTab {
  float* logits[GROUP_MAX_LANES];
  int32_t* argmax[GROUP_MAX_LANES];
  const DDState* st[GROUP_MAX_LANES];
};
// grid (rows, ROW_COPY_WGS): a logits row is Vpad floats (a multiple of 16, rows of hipMalloc'ed buffers): 16-byte accesses, eight workgroups
// per row (round 6: one workgroup copying a 128-KB row four bytes at a time took 63 us at the end of every rider sweep, in front of the masks)
#define ROW_COPY_WGS 8
__global__ __launch_bounds__(256) void k_scatter_base(const float* grp_logits, const int32_t* grp_argmax, int Vpad, ScatterTab tab) {
  int m = blockIdx.x;
  if (tab.st[m]->done) return;
  const f32x4_t* src = (const f32x4_t*)(grp_logits + (size_t)m * Vpad);
  f32x4_t* dst = (f32x4_t*)tab.logits[m];
  for (int i = blockIdx.y * 256 + threadIdx.x; i < (Vpad >> 2); i += gridDim.y * 256) dst[i] = src[i];
  if (blockIdx.y == 0 && threadIdx.x == 0) tab.argmax[m][0] = grp_argmax[m];
}

// -----------------------------------------------------------------------------------------------
// Rider form of the group step (16, 24, ... 64 sequences in groups of eight, K <= 8, fp16 cache, 16-bit weights of the 7B shapes).
// The un-masked rows of a group step do not need a sweep of their own: a 64-row member sweep streams every weight anyway, and
// eight more rows are a ninth operand plane of the same kernels (dd_gemv.hip try_slices9).  The groups of a branch form a ring
// c0, c1, ... : the sweep of c_j carries the members of c_j AND the un-masked rows of c_(j+1) — whose previous token was voted a
// step ago (or, for the last sweep of the ring, the rows of c0's NEXT step: c0's token of this step was voted in the ring's first
// sweep).  c_(j+1)'s masks are sampled when that sweep ends, its members run next.  c0's rows for the next step are parked in
// base_next / argmax_next and promoted when that step starts; a ring whose c0 has nothing parked (first step after a prefill,
// a changed line-up) starts with the classic fused pass over those sequences.  Every row goes through the kernels of the other
// pass widths (rows bit-identical across widths), every sequence draws its masks from its own stream in the same order: the
// tokens, masks, logits and caches are those of the classic group step and of every sequence decoded alone.
// Per step and sequence the weights are read 1/8 times ... the sweep count of a 32-sequence step drops from five to four.
// -----------------------------------------------------------------------------------------------
static int g_half_planes_first = 1;   // dd_tools_set_tuning key 31: where both apply, half planes (classic form) before the rider form
void dd_engine_set_half_planes_first(int on) { g_half_planes_first = on; }
static int g_half_planes = 1;    // dd_tools_set_tuning key 30: K <= 4 packs two sequences per operand plane (0: one, rows 4..7 empty)
void dd_engine_set_half_planes(int on) { g_half_planes = on; }
// sixteen sequences per 64-row member sweep: K <= 4, fp16 caches, 16-bit weights of the shapes with eight-plane slice kernels for every matrix
static bool half_planes_ok(dd_lm* const* lanes, int n, int K) {
  if (!g_half_planes || K < 1 || K > 4 || g_pair_sweeps < 8 || n < 16) return false;
  dd_lm* h0 = lanes[0];
  if (!h0->kv16 || h0->fp8 || !h0->gemv_part) return false;
  const int qt = h0->qkv_tiles, gt = 2 * h0->dff / 16;
  if (h0->S_d != 128 || h0->S_q != 128 || !(h0->S_ff == 8 * 43 || h0->S_ff == 8 * 56) || qt < 64 || gt < 64 || h0->d / 16 < 64 || h0->Vpad / 16 < 64)
    return false;
  for (int m = 0; m < n; ++m)
    if (lanes[m]->cfg.vote_on == DD_VOTE_AVERAGE) return false;
  return true;
}
static int g_rider = 1;          // dd_tools_set_tuning key 26
static int g_rider_staged = 1;   // dd_tools_set_tuning key 33: the rings' sweeps in stages, late groups' masks sampled between the stages on the caller's stream
void dd_engine_set_rider_staged(int on) { g_rider_staged = on; }
void dd_engine_set_rider(int on) { g_rider = on; }
// branches of the rider form for this call, 0: not applicable
// (*gs: sequences per group — 8, or 14 with half planes (K <= 4): seven planes of two sequences + two riding planes)
static int rider_branches(dd_lm* const* lanes, int n, int K, int* gs_out = nullptr) {
  if (gs_out) *gs_out = 8;
  if (!g_rider || K < 1 || K > 8 || g_pair_sweeps < 8 || n < 16 || n > GROUP_MAX_LANES) return 0;
  const bool hp = half_planes_ok(lanes, n, K);
  const int gs = (hp && n >= 28 && n % 14 == 0) ? 14 : 8;
  if (hp && gs == 8 && g_half_planes_first) return 0;            // K <= 4, not whole groups of fourteen: the classic form with half planes
  if (n % gs) return 0;
  if (gs_out) *gs_out = gs;
  dd_lm* h0 = lanes[0];
  if (!h0->kv16 || !h0->gemv_part) return 0;
  // shapes with nine-plane kernels (dd_gemv.hip try_slices9 / try_slices9_fp8)
  const int qt = h0->qkv_tiles, gt = 2 * h0->dff / 16;
  if (h0->fp8) {                 // fp8 tiles: K = 4096 for qkv / o / gate-up / lm_head, K = 14336 for down (Mistral-7B: BASELINE config 5)
    if (h0->S_d != 128 || h0->S_q != 128 || h0->S_ff != 8 * 56 || qt < 64 || gt < 64 || h0->d / 16 < 64 || h0->Vpad / 16 < 64) return 0;
  } else if (h0->S_d != 128 || h0->S_q != 128 || !(h0->S_ff == 8 * 43 || h0->S_ff == 8 * 56) || (qt % 16) != 0 || qt / 16 * 4 > 256 ||
             gt < 64 || h0->d / 16 < 64 || h0->Vpad / 16 < 64) {
    return 0;
  }
  for (int m = 0; m < n; ++m) {
    const dd_lm* q = lanes[m];
    if (q->cfg.mask_mode != h0->cfg.mask_mode || q->cfg.k_top != h0->cfg.k_top || q->cfg.mask_mode == DD_MASK_IBLIP_KL ||
        q->cfg.vote_on == DD_VOTE_AVERAGE)
      return 0;
  }
  const int groups = n / gs;
  int nbr = (g_rider_branches >= 2 && h0->side[0]) ? (g_rider_branches < groups / 2 ? g_rider_branches : groups / 2) : 1;
  while (nbr > 1 && !h0->side[nbr - 2]) --nbr;
  return nbr < 1 ? 1 : nbr;
}
static bool rider_parked(const dd_lm* q) { return q->pend_valid && q->pend_step == q->steps_since_prefill; }

struct PromoteTab {
  const float* src[GROUP_MAX_LANES];
  float* dst[GROUP_MAX_LANES];
  const int32_t* asrc[GROUP_MAX_LANES];
  int32_t* adst[GROUP_MAX_LANES];
  const DDState* st[GROUP_MAX_LANES];
};
__global__ __launch_bounds__(256) void k_promote_base(PromoteTab tab, int Vpad) {
  const int m = blockIdx.x;
  if (tab.st[m]->done) return;
  const f32x4_t* src = (const f32x4_t*)tab.src[m];
  f32x4_t* dst = (f32x4_t*)tab.dst[m];
  for (int i = blockIdx.y * 256 + threadIdx.x; i < (Vpad >> 2); i += gridDim.y * 256) dst[i] = src[i];
  if (blockIdx.y == 0 && threadIdx.x == 0) tab.adst[m][0] = tab.asrc[m][0];
}

// gs = 8: one sequence per member plane; gs = 14 (K <= 4): half planes — fourteen sequences in seven planes, their partners' un-masked
// rows in planes 7 and 8
static int group_step_rider(dd_lm* const* lanes, int n, const double* mprobs, int K, dd_rng* const* rngs, int nbr, int gs, hipStream_t st) {
  dd_lm* h0 = lanes[0];
  const int groups = n / gs, n_early = gs * nbr;
  const bool packed = gs == 14;
  auto begin_lanes = [&](dd_lm* const* qs, int cnt, hipStream_t s) -> int {
    StepBeginLanes t;
    memset(&t, 0, sizeof(t));
    for (int m = 0; m < cnt; ++m) {
      dd_lm* q = qs[m];
      t.st[m] = q->state, t.leak_bits[m] = q->leak_bits, t.L[m] = q->L, t.mask_positions[m] = q->cfg.leak_mask == 2 ? 1 : 0;
    }
    k_step_begin_lanes<<<cnt, 256, 0, s>>>(t);
    DD_CHECK_LAUNCH();
    return DD_OK;
  };
  auto masks = [&](int m0, int cnt, hipStream_t s) -> int {           // keep sets + masks of lanes m0 .. m0 + cnt - 1 (cnt <= 32)
    MaskLaneArgs ml[32];
    for (int i = 0; i < cnt; ++i) {
      dd_lm* q = lanes[m0 + i];
      ml[i] = {q->epi, q->L, q->keep, q->argmax_base, q->topk_ids, dd_rng_state_ptr(rngs ? rngs[m0 + i] : nullptr), q->drop, q->n_drop,
               q->drop_bits, &q->state->done};
    }
    return dd_sample_masks_lanes(ml, cnt, h0->cfg.k_top, mprobs, K, h0->cfg.mask_mode, s);
  };
  RC(begin_lanes(lanes, n, st));
  h0->bit0 = 0;
  for (int m = 0; m < n; ++m) lanes[m]->last_K = K;
  // the ring leaders' un-masked rows: parked by the previous step, or the classic fused pass over them
  bool parked = true;
  for (int m = 0; m < n_early; ++m) parked &= rider_parked(lanes[m]);
  if (parked) {
    PromoteTab tab;
    memset(&tab, 0, sizeof(tab));
    for (int m = 0; m < n_early; ++m) {
      dd_lm* q = lanes[m];
      tab.src[m] = q->base_next, tab.dst[m] = q->base_logits, tab.asrc[m] = q->argmax_next, tab.adst[m] = q->argmax_base, tab.st[m] = q->state;
    }
    k_promote_base<<<dim3(n_early, ROW_COPY_WGS), 256, 0, st>>>(tab, h0->Vpad);
    DD_CHECK_LAUNCH();
  } else {
    RC(lm_sweep(h0, n_early, nullptr, 0, h0->grp_logits, st, lanes));
    RC(dd_argmax_rows(h0->grp_logits, n_early, h0->V, h0->Vpad, h0->grp_argmax, st));
    ScatterTab tab;
    memset(&tab, 0, sizeof(tab));
    for (int m = 0; m < n_early; ++m) tab.logits[m] = lanes[m]->base_logits, tab.argmax[m] = lanes[m]->argmax_base, tab.st[m] = lanes[m]->state;
    k_scatter_base<<<dim3(n_early, ROW_COPY_WGS), 256, 0, st>>>(h0->grp_logits, h0->grp_argmax, h0->Vpad, tab);
    DD_CHECK_LAUNCH();
  }
  RC(masks(0, n_early, st));
  const bool fork = nbr >= 2;
  if (fork) {
    DD_HIP(hipEventRecord(h0->ev_fork, st));
    for (int i = 0; i + 1 < nbr; ++i) DD_HIP(hipStreamWaitEvent(h0->side[i], h0->ev_fork, 0));
  }
  // Stages: stage j runs the j-th sweep of every ring, one ring per branch; the branches meet after each stage, and the masks of the
  // groups whose un-masked rows just rode are sampled in ONE launch on the caller's stream with nothing beside it — as the masks of the
  // ring leaders (above) and of the classic step are.  (Sampled on the branches, beside the other rings' sweeps, a sequence's tokens
  // differed between two identical runs about once in 2,000 64-lane steps — always a sequence whose masks had been sampled that way;
  // tools/stress_lanes.py.  dd_tools_set_tuning key 33 = 0 restores that form for the comparison.)
  int ring[4][8], klen[4], kmax = 0;
  for (int br = 0; br < nbr; ++br) {
    klen[br] = 0;
    for (int g = br; g < groups; g += nbr) ring[br][klen[br]++] = g;
    if (klen[br] > kmax) kmax = klen[br];
  }
  auto fork_all = [&]() -> int {
    if (!fork) return DD_OK;
    DD_HIP(hipEventRecord(h0->ev_fork, st));
    for (int i = 0; i + 1 < nbr; ++i) DD_HIP(hipStreamWaitEvent(h0->side[i], h0->ev_fork, 0));
    return DD_OK;
  };
  auto join_all = [&]() -> int {
    if (!fork) return DD_OK;
    for (int i = 0; i + 1 < nbr; ++i) {
      DD_HIP(hipEventRecord(h0->ev_join[i], h0->side[i]));
      DD_HIP(hipStreamWaitEvent(st, h0->ev_join[i], 0));
    }
    return DD_OK;
  };
  for (int j = 0; j < kmax; ++j) {
    if (j > 0 && g_rider_staged) RC(fork_all());
    int late[4], n_late = 0;                               // groups whose masks follow this stage
    for (int br = 0; br < nbr; ++br) {
      if (j >= klen[br]) continue;
      const int k = klen[br];
      hipStream_t bs = br ? h0->side[br - 1] : st;
      dd_lm* const* qs = lanes + gs * ring[br][j];
      const int pg = ring[br][(j + 1) % k];                // the group whose un-masked rows ride
      dd_lm* const* rd = lanes + gs * pg;
      const bool ahead = j == k - 1;                       // the ring leader's rows of the NEXT step
      dd_lm* scratch = br ? qs[0] : h0;
      if (ahead) RC(begin_lanes(rd, gs, bs));              // positions of the step ahead (the leader's token was voted in sweep 0)
      RC(lm_sweep_groups(scratch, qs, gs, K, bs, rd, gs, packed));
      if (packed) {
        RC(group_finish(scratch, qs, 8, K, bs, 0));
        RC(group_finish(scratch, qs + 8, gs - 8, K, bs, 8));
      } else {
        RC(group_finish(scratch, qs, 8, K, bs));
      }
      RC(dd_argmax_rows(scratch->grp_logits, gs, h0->V, h0->Vpad, scratch->grp_argmax, bs));
      ScatterTab tab;
      memset(&tab, 0, sizeof(tab));
      for (int m = 0; m < gs; ++m) {
        dd_lm* q = rd[m];
        tab.logits[m] = ahead ? q->base_next : q->base_logits, tab.argmax[m] = ahead ? q->argmax_next : q->argmax_base, tab.st[m] = q->state;
      }
      k_scatter_base<<<dim3(gs, ROW_COPY_WGS), 256, 0, bs>>>(scratch->grp_logits, scratch->grp_argmax, h0->Vpad, tab);
      DD_CHECK_LAUNCH();
      if (!ahead) {
        if (g_rider_staged) late[n_late++] = pg;
        else RC(masks(gs * pg, gs, bs));
      }
    }
    if (g_rider_staged || j == kmax - 1) RC(join_all());
    if (n_late) {                                          // one launch for all of them (at most 32 sequences per launch)
      MaskLaneArgs ml[32];
      int cnt = 0;
      for (int i = 0; i < n_late; ++i)
        for (int m = 0; m < gs; ++m) {
          const int li = gs * late[i] + m;
          dd_lm* q = lanes[li];
          ml[cnt++] = {q->epi, q->L, q->keep, q->argmax_base, q->topk_ids, dd_rng_state_ptr(rngs ? rngs[li] : nullptr), q->drop, q->n_drop,
                       q->drop_bits, &q->state->done};
          if (cnt == 32 || (i == n_late - 1 && m == gs - 1)) {
            RC(dd_sample_masks_lanes(ml, cnt, h0->cfg.k_top, mprobs, K, h0->cfg.mask_mode, st));
            cnt = 0;
          }
        }
    }
  }
  for (int m = 0; m < n; ++m) {
    dd_lm* q = lanes[m];
    q->pend_valid = m < n_early, q->pend_step = q->steps_since_prefill;
  }
  return DD_OK;
}

static int group_step_eager(dd_lm* const* lanes, int n, const double* mprobs, int K, dd_rng* const* rngs, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(lanes && n >= 1 && n <= GROUP_MAX_LANES, "dd_lm_group_step: 1..%d sequences per group (got %d)", GROUP_MAX_LANES, n);
  DD_REQUIRE(K >= 0 && K <= MAX_MEMBERS && (K == 0 || mprobs), "dd_lm_group_step: bad K / mprobs");
  for (int m = 0; lanes && m < n; ++m) RC(ensure_member_rows(lanes[m], K));
  dd_lm* h0 = lanes[0];
  DD_REQUIRE(h0, "dd_lm_group_step: null handle");
  dd_lm* owner = h0->wsrc ? h0->wsrc : h0;
  for (int m = 0; m < n; ++m) {
    dd_lm* q = lanes[m];
    DD_REQUIRE(q, "dd_lm_group_step: null handle");
    DD_REQUIRE(q->tp_world == 1, "dd_lm_group_step: sequence %d is a tensor-parallel shard", m);
    DD_REQUIRE((q->wsrc ? q->wsrc : q) == owner, "dd_lm_group_step: sequence %d does not share the group's weights", m);
    DD_REQUIRE(q->T_cap == h0->T_cap && q->Vpad == h0->Vpad && q->kv16 == h0->kv16,
               "dd_lm_group_step: sequence %d has a different KV capacity or cache format", m);
    for (int j = 0; j < m; ++j) DD_REQUIRE(lanes[j] != q, "dd_lm_group_step: sequence listed twice");
    if (!q->prefilled) {
      dd_set_error("dd_lm_group_step: sequence %d: decode before prefill", m);
      return DD_ESTATE;
    }
    if (q->T_host + 1 >= q->T_cap) {
      dd_set_error("dd_lm_group_step: sequence %d: KV cache full (%d tokens)", m, q->T_cap);
      return DD_ESTATE;
    }
    DD_REQUIRE(q->n_tok_host < MAX_NEW_TOKENS, "dd_lm_group_step: token buffer full");
    DD_REQUIRE(K == 0 || q->cfg.mask_mode == DD_MASK_IBLIP_QUANTILE || (rngs && rngs[m]), "dd_lm_group_step: sequence %d needs an rng", m);
  }
  int rider_gs = 8;
  if (const int nbr = rider_branches(lanes, n, K, &rider_gs)) return group_step_rider(lanes, n, mprobs, K, rngs, nbr, rider_gs, st);
  {
    StepBeginLanes t;
    memset(&t, 0, sizeof(t));
    for (int m = 0; m < n; ++m) {
      dd_lm* q = lanes[m];
      t.st[m] = q->state, t.leak_bits[m] = q->leak_bits, t.L[m] = q->L, t.mask_positions[m] = q->cfg.leak_mask == 2 ? 1 : 0;
    }
    k_step_begin_lanes<<<n, 256, 0, st>>>(t);
    DD_CHECK_LAUNCH();
  }
  h0->bit0 = 0;
  RC(lm_sweep(h0, n, nullptr, 0, h0->grp_logits, st, lanes));
  RC(dd_argmax_rows(h0->grp_logits, n, h0->V, h0->Vpad, h0->grp_argmax, st));
  // hand every sequence its base row (logits + argmax)
  {
    ScatterTab tab;
    memset(&tab, 0, sizeof(tab));
    for (int m = 0; m < n; ++m) tab.logits[m] = lanes[m]->base_logits, tab.argmax[m] = lanes[m]->argmax_base, tab.st[m] = lanes[m]->state;
    k_scatter_base<<<dim3(n, ROW_COPY_WGS), 256, 0, st>>>(h0->grp_logits, h0->grp_argmax, h0->Vpad, tab);
    DD_CHECK_LAUNCH();
  }
  // masks of every sequence first (each from its own rng stream), then the members: two sequences per 16-row sweep where
  // possible (bf16 weights, K <= 8), the 8-row sweep otherwise
  bool same_rule = true;
  for (int m = 0; m < n; ++m) {
    lanes[m]->last_K = K;
    same_rule &= lanes[m]->cfg.mask_mode == h0->cfg.mask_mode && lanes[m]->cfg.k_top == h0->cfg.k_top &&
                 h0->cfg.mask_mode != DD_MASK_IBLIP_KL;        // the fused keep + mask launch computes the overlap keep set
  }
  if (K > 0 && same_rule) {          // keep sets + masks of all sequences: one launch, one workgroup per sequence
    MaskLaneArgs ml[GROUP_MAX_LANES];
    for (int m = 0; m < n; ++m) {
      dd_lm* q = lanes[m];
      ml[m] = {q->epi, q->L, q->keep, q->argmax_base, q->topk_ids, dd_rng_state_ptr(rngs ? rngs[m] : nullptr), q->drop, q->n_drop,
               q->drop_bits, &q->state->done};
    }
    for (int m0 = 0; m0 < n; m0 += 32)       // the sampler's pointer table travels by value: 32 sequences per launch
      RC(dd_sample_masks_lanes(ml + m0, n - m0 < 32 ? n - m0 : 32, h0->cfg.k_top, mprobs, K, h0->cfg.mask_mode, st));
  } else if (K > 0) {
    for (int m = 0; m < n; ++m) {
      dd_lm* q = lanes[m];
      RC(step_keep(q, &q->state->done, st));
      RC(dd_sample_masks_impl(q->epi, q->L, mprobs, K, q->keep, q->cfg.mask_mode, DD_RNG_MT19937, nullptr,
                              dd_rng_state_ptr(rngs ? rngs[m] : nullptr), q->drop, q->n_drop, nullptr, q->drop_bits,
                              &q->state->done, st));
    }
  }
  const bool multi = g_pair_sweeps && K > 0 && K <= 8;
  const bool pack16 = multi && half_planes_ok(lanes, n, K);      // K <= 4: sixteen sequences per 64-row sweep (two per operand plane)
  auto width = [&](int m) -> int {     // sequences of the member sweep that starts at lane m
    const int left = n - m;
    if (pack16 && left >= 16) return 16;
    return !multi ? 1 : (left >= 8 && g_pair_sweeps >= 8 ? 8 : (left >= 4 && g_pair_sweeps >= 4 ? 4 : (left >= 2 ? 2 : 1)));
  };
  // The member sweeps of a group step are independent of each other, and each is a chain of dependent launches in which every
  // matrix ends in a finishing / combine kernel that streams nothing (a fifth of a 64-row sweep).  Dealt over two (up to four)
  // streams — a side branch works on the scratch of ITS first lane, results unchanged — one sweep's small kernels and launch
  // boundaries overlap another's weight streaming: 26.4 -> 22.5 ms per 32-lane step with two branches (tools/branch_ab.py).
  int n_multi = 0;
  for (int m = 0; m < n; m += width(m)) n_multi += width(m) > 1 ? 1 : 0;
  const int nbr = g_branches < n_multi ? g_branches : n_multi;       // branches in use: the caller's stream + nbr - 1 side streams
  // (fp32-cache engines ran one branch in round 3: beside another branch's GEMVs their VALU attention kernel gave different bits — its packed
  // FP32 multiply-adds, see build.py NO_PACKED_FP32 and DESIGN.md "Determinism".  Key 37 = -1 restores the single branch for the A/B.)
  const bool fork = nbr >= 2 && (h0->kv16 || g_fp32_fork >= 0) && h0->side[nbr - 2] != nullptr;
  const bool masked0 = fork && g_mask_branches && nbr == 2 && h0->side[2];     // (debug) branch 0 on a CU-masked stream of its own
  if (fork) {
    DD_HIP(hipEventRecord(h0->ev_fork, st));
    for (int i = 0; i + 1 < nbr; ++i) DD_HIP(hipStreamWaitEvent(h0->side[i], h0->ev_fork, 0));
    if (masked0) DD_HIP(hipStreamWaitEvent(h0->side[2], h0->ev_fork, 0));
  }
  int i_multi = 0;
  for (int m = 0; m < n; ++m) {
    dd_lm* q = lanes[m];
    if (K > 0) {
      const int ng = width(m);
      if (ng > 1) {
        const int br = fork ? i_multi % nbr : 0;               // branch 0 = the caller's stream on the leader's scratch
        dd_lm* scratch = br ? lanes[m] : h0;
        hipStream_t bs = br ? h0->side[br - 1] : (masked0 ? h0->side[2] : st);
        if (ng == 16) {
          RC(lm_sweep_groups(scratch, lanes + m, 16, K, bs, nullptr, 0, true));
          RC(group_finish(scratch, lanes + m, 8, K, bs, 0));
          RC(group_finish(scratch, lanes + m + 8, 8, K, bs, 8));
        } else {
          RC(lm_sweep_groups(scratch, lanes + m, ng, K, bs));
          RC(group_finish(scratch, lanes + m, ng, K, bs));
        }
        ++i_multi;
        m += ng - 1;
        continue;
      }
      RC(dd_lm_step_members(q, 0, K, stream_));
    } else {
      // stock greedy: the base row's new K/V (in the leader's scratch, row m) is what gets appended
      q->commit_k = h0->knew + (size_t)m * h0->kv_dim, q->commit_v = h0->vnew + (size_t)m * h0->kv_dim;
    }
    int rc = dd_lm_step_commit(q, K, stream_);
    q->commit_k = q->commit_v = nullptr;
    RC(rc);
    q->steps_since_prefill++;
  }
  if (fork)
    for (int i = 0; i + 1 < nbr; ++i) {
      DD_HIP(hipEventRecord(h0->ev_join[i], h0->side[i]));
      DD_HIP(hipStreamWaitEvent(st, h0->ev_join[i], 0));
    }
  if (masked0) {
    DD_HIP(hipEventRecord(h0->ev_join[2], h0->side[2]));
    DD_HIP(hipStreamWaitEvent(st, h0->ev_join[2], 0));
  }
  return DD_OK;
}

// -----------------------------------------------------------------------------------------------
// Speculative single-sequence step: ONE sweep over the weights for the un-masked row AND the K members.
// The members' masks depend on the un-masked pass only through the keep set (models/llava.py:603, 660: the tokens whose
// top-k ids contain the base argmax are restored to 1).  The masks are sampled for an EMPTY keep set before the sweep (same
// draws, same order), the K members ride in rows 0..K-1 and the un-masked row in row 8 of a 16-row pass (two operand planes:
// the kernels of the lanes path, rows bit-identical to the 8-row kernels), and afterwards the real keep set is compared with
// what the members dropped: if no kept token was dropped by any member the speculative masks ARE the reference's masks and
// the step is complete after one sweep; otherwise the masks are re-sampled from the saved rng state with the real keep set
// and the members re-run (the classic second sweep) — its kernels are enqueued either way and return at once when the
// device-side flag says the speculation held.  Results are those of the two-sweep step in every case.
// -----------------------------------------------------------------------------------------------
static int g_speculate = 2;   // dd_set_tuning key 14: process default of the policy — 0 never, 1 always, 2 adaptive
void dd_engine_set_speculate(int mode) { g_speculate = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
static int spec_mode_of(const dd_lm* h) { return h->spec_mode >= 0 ? h->spec_mode : g_speculate; }
// Break-even of the speculative step (LLaVA-1.5-7B shapes, K = 8, measured: tests/test_gpu_full_size_configs.py): it costs one
// 16-row sweep when the masks stand (4.0 ms) and that sweep plus the 8-row re-run when they do not (6.85 ms), against a 1-row and
// an 8-row sweep for the plain step (6.12 ms): 4.0 + (1 - h) * 2.85 = 6.12 at h = 0.25.  On a checkpoint whose keep sets
// (models/llava.py:443-482) are rarely empty the share that holds falls below that and the plain step is the faster one.  The
// running share moves by 1/16 per step: misses come in runs (they depend on the token being said), and a policy that backs off
// after five of them costs a workload with a 2/3 hit rate 7 % (measured with 1/8 and a threshold of 0.35).
#define SPEC_BREAK_EVEN 0.25f
#define SPEC_RATE_WEIGHT 0.0625f
#define SPEC_COOLDOWN_STEPS 32
#define SPEC_PROBE_RATE 0.33f     // rate a probe phase starts from: five misses in a row end it

static int lm_sweep_spec(dd_lm* h, int K, hipStream_t st) {
  const int d = h->d, dff = h->dff;
  EmbedLanes el;
  memset(&el, 0, sizeof(el));
  for (int m = 0; m < K; ++m) el.state[m] = h->state;
  el.state[8] = h->state;                                     // row 8: the un-masked row
  RC(ddk_embed_rows_lanes(h->embed, d, el, 16, h->xa, h->lw[0].norm1, h->xop_d, h->ssq_a, d / 16, st, h->wf));
  int ssq_n = 1;
  auto common = [&](GemvArgs& a) {
    a.nb = K, a.n_groups = 2, a.fp8 = h->fp8, a.state = h->state, a.wf = h->wf;
    a.part = h->gemv_part, a.part_floats = h->gemv_part_floats;
  };
  for (int l = 0; l < h->Lyr; ++l) {
    LayerW& w = h->lw[l];
    float* kn = h->knew + (size_t)l * KV_ROWS * h->kv_dim;
    float* vn = h->vnew + (size_t)l * KV_ROWS * h->kv_dim;
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    common(a);
    a.W = w.wqkv, a.S = h->S_d, a.n_tiles = h->qkv_tiles, a.xop = h->xop_d, a.wscale = w.s_qkv;
    a.ssq_in = h->ssq_a, a.ssq_n = ssq_n, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
    a.qbuf = h->qbuf, a.q_tiles = h->q_tiles, a.k_tiles = h->k_tiles;
    a.q_dim = h->q_dim, a.kv_dim = h->kv_dim, a.rope_cos = h->rope_cos, a.rope_sin = h->rope_sin;
    a.knew_g[0] = kn, a.vnew_g[0] = vn, a.knew_g[1] = kn + (size_t)8 * h->kv_dim, a.vnew_g[1] = vn + (size_t)8 * h->kv_dim;
    a.knew = kn, a.vnew = vn;
    RC(ddk_gemv_groups(EPI_QKV, a, st));
    AttnDecodeArgs t;
    memset(&t, 0, sizeof(t));
    t.wf = h->wf;
    t.qbuf = h->qbuf, t.T_cap = h->T_cap, t.nb = K, t.n_heads = h->H, t.n_kv = h->Hkv, t.bit0 = 0, t.kv16 = h->kv16;
    t.part_o = h->part_o, t.part_ml = h->part_ml, t.xop_out = h->xop_q;
    t.n_lanes = 2, t.lane_groups = 2, t.max_T = h->T_host;
    for (int g = 0; g < 2; ++g) {
      t.knew_g[g] = a.knew_g[g], t.vnew_g[g] = a.vnew_g[g];
      t.lane_kc[g] = h->kc + (size_t)l * h->lsk, t.lane_vc[g] = h->vc + (size_t)l * h->lsv, t.lane_state[g] = h->state;
      t.lane_span_start[g] = h->span_start, t.lane_span_len[g] = h->L;
    }
    t.lane_bits[0] = h->drop_bits;                              // members: bit m of plane 0
    t.lane_bits[1] = h->cfg.leak_mask ? h->leak_bits : nullptr;  // un-masked row: InstructBLIP's leaked zeros (bit 0) or nothing
    t.knew = t.knew_g[0], t.vnew = t.vnew_g[0];
    RC(ddk_attn_decode(t, st));
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    common(a);
    a.W = w.wo, a.S = h->S_q, a.n_tiles = d / 16, a.xop = h->xop_q, a.wscale = w.s_o;
    a.out = h->xa, a.ldo = d, a.normw_next = w.norm2, a.xop_next = h->xop_d, a.S_next = h->S_d, a.ssq_out = h->ssq_b, a.ssq_ld = d / 16;
    RC(ddk_gemv_groups(EPI_RESID, a, st));
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    common(a);
    a.W = w.wgu, a.S = h->S_d, a.n_tiles = dff / 16, a.xop = h->xop_d, a.wscale = w.s_gu;
    a.ssq_in = h->ssq_b, a.ssq_n = d / 16, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
    a.xop_next = h->xop_ff, a.S_next = h->S_ff;
    RC(ddk_gemv_groups(EPI_SILU, a, st));
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    common(a);
    a.W = w.wdown, a.S = h->S_ff, a.n_tiles = d / 16, a.xop = h->xop_ff, a.wscale = w.s_down;
    a.out = h->xa, a.ldo = d, a.normw_next = (l + 1 < h->Lyr) ? h->lw[l + 1].norm1 : h->final_norm;
    a.xop_next = h->xop_d, a.S_next = h->S_d, a.ssq_out = h->ssq_a, a.ssq_ld = d / 16;
    RC(ddk_gemv_groups(EPI_RESID, a, st));
    ssq_n = d / 16;
  }
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.wf = h->wf;
  common(a);
  a.W = h->lm_head, a.S = h->S_d, a.n_tiles = h->Vpad / 16, a.xop = h->xop_d, a.wscale = h->s_lm;
  a.ssq_in = h->ssq_a, a.ssq_n = ssq_n, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
  a.out_g[0] = h->member_logits, a.out_g[1] = h->grp_logits;   // row 0 of the second group = the un-masked row's logits
  a.out = h->member_logits, a.ldo = h->Vpad, a.n_valid = h->V;
  RC(ddk_gemv_groups(EPI_STORE, a, st));
  return DD_OK;
}

// The step in three phases.  A: masks for an empty keep set, the combined sweep, the real keep set and the check.  B: the
// fallback (every kernel returns at once when the device flag says the speculation held).  C: member argmax, vote, commit.
// dd_lm_decode_step enqueues A, B, C back to back (one graph: the host never waits, B's ~350 launches cost ≈0.3 ms even when
// they have nothing to do); dd_lm_decode_step_sync enqueues A with a note to the host and B only when the note says so.
static int spec_phase_a(dd_lm* h, const double* mprobs, int K, dd_rng* rng, const float* uniforms, hipStream_t st, bool note) {
  const int32_t* gate = &h->state->done;
  k_step_begin<<<1, 256, 0, st>>>(h->state, h->leak_bits, h->L, h->cfg.leak_mask == 2 ? 1 : 0);
  DD_CHECK_LAUNCH();
  const int mode = h->cfg.mask_mode;
  const int rng_mode = uniforms ? DD_RNG_INJECTED : DD_RNG_MT19937;
  const bool draws = mode != DD_MASK_IBLIP_QUANTILE && !uniforms;
  DD_REQUIRE(mode == DD_MASK_IBLIP_QUANTILE || uniforms || rng, "dd_lm_step: an rng or uniforms is required");
  uint32_t* rs = dd_rng_state_ptr(rng);
  if (draws) DD_HIP(hipMemcpyAsync(h->rng_backup, rs, 625 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  // 1. masks for an empty keep set (the stream advances by the step's K * L draws here, once)
  RC(dd_sample_masks_impl(h->epi, h->L, mprobs, K, nullptr, mode, rng_mode, uniforms, rs, h->drop, h->n_drop, nullptr,
                          h->drop_bits, gate, st, nullptr, true));
  // 2. the combined sweep
  h->bit0 = 0;
  RC(lm_sweep_spec(h, K, st));
  RC(dd_copy_row_gated(h->grp_logits, h->base_logits, h->Vpad, gate, st));
  RC(dd_argmax_rows_gated(h->base_logits, 1, h->V, h->Vpad, h->argmax_base, gate, st));
  // 3. the real keep set; did any member drop one of its tokens?
  RC(step_keep(h, gate, st));
  const int keep_matters = (mode == DD_MASK_NEXT_NO_OVERLAP || mode == DD_MASK_LLAVA_CUMULATIVE_NO_OVERLAP) ? 0 : 1;
  RC(dd_spec_check(h->keep, h->drop_bits, h->L, K, gate, h->spec_ok, keep_matters, st,
                   note ? h->tok_host_dev + MAX_NEW_TOKENS + 1 : nullptr));
  return DD_OK;
}
static int spec_phase_b(dd_lm* h, const double* mprobs, int K, dd_rng* rng, const float* uniforms, hipStream_t st) {
  const int mode = h->cfg.mask_mode;
  const int rng_mode = uniforms ? DD_RNG_INJECTED : DD_RNG_MT19937;
  const bool draws = mode != DD_MASK_IBLIP_QUANTILE && !uniforms;
  const int keep_matters = (mode == DD_MASK_NEXT_NO_OVERLAP || mode == DD_MASK_LLAVA_CUMULATIVE_NO_OVERLAP) ? 0 : 1;
  // 4. fallback (returns at once when the speculation held): the reference's masks from the same draws, members re-run
  if (keep_matters) {
    RC(dd_sample_masks_impl(h->epi, h->L, mprobs, K, h->keep, mode, rng_mode, uniforms, dd_rng_state_ptr(rng), h->drop, h->n_drop, nullptr,
                            h->drop_bits, h->spec_ok, st, draws ? h->rng_backup : nullptr, false));
    RC(lm_sweep(h, K, h->drop_bits, 0, h->member_logits, st, nullptr, h->spec_ok));
  }
  return DD_OK;
}
static int spec_phase_c(dd_lm* h, int K, hipStream_t st) {
  const int32_t* gate = &h->state->done;
  // 5. member argmax (+ InstructBLIP's hidden-state argmax), vote, commit — as after dd_lm_step_members
  RC(dd_argmax_rows_gated(h->member_logits, K, h->V, h->Vpad, h->member_tok, gate, st));
  if (h->cfg.vote_on == DD_VOTE_HIDDEN) {
    RC(ddk_final_norm_rows(h->xa, K, h->d, h->final_norm, h->cfg.rms_eps, h->hidden, st));
    RC(dd_argmax_rows_gated(h->hidden, K, h->d, h->d, h->member_vote, gate, st));
  }
  h->last_K = K;
  return dd_lm_step_commit(h, K, st);
}
static int decode_step_spec(dd_lm* h, const double* mprobs, int K, dd_rng* rng, const float* uniforms, hipStream_t st) {
  RC(spec_phase_a(h, mprobs, K, rng, uniforms, st, false));
  RC(spec_phase_b(h, mprobs, K, rng, uniforms, st));
  return spec_phase_c(h, K, st);
}

static int decode_step_eager(dd_lm* h, const double* mprobs, int K, dd_rng* rng, const float* uniforms, void* stream, bool speculate) {
  if (speculate && K >= 1 && K <= 8 && h && h->prefilled && mprobs && h->T_host + 1 < h->T_cap && h->n_tok_host < MAX_NEW_TOKENS)
    return decode_step_spec(h, mprobs, K, rng, uniforms, (hipStream_t)stream);
  RC(dd_lm_step_base(h, mprobs, K, rng, uniforms, stream));
  if (K > 0) RC(dd_lm_step_members(h, 0, K, stream));
  return dd_lm_step_commit(h, K, stream);
}



// One whole ensemble step = ~340 kernel launches.  After the first (eager) step of a sequence the step is captured
// into a hipGraph and replayed: every launch argument is step-invariant (lengths, positions, tokens and the vote live
// in device memory), except the number of 64-key attention tiles the launch is shaped for (rounded up to 4, so it changes
// every 256 tokens), which is part of the cache key together with K, the
// probabilities and the rng.  Host cost per step drops from ~3.4 ms of launches to one hipGraphLaunch.
static int decode_step_queued(dd_lm* h, const double* mprobs, int K, dd_rng* rng, const float* uniforms, void* stream, bool speculate) {
  DD_REQUIRE(h, "dd_lm_decode_step: null handle");
  DD_REQUIRE(h->tp_world == 1, "this handle is a tensor-parallel shard: drive it through dd_lm_tp_* (include/dropdec.h)");
  hipStream_t st = (hipStream_t)stream;
  const bool graphable = g_use_graph && !uniforms && h->prefilled && h->steps_since_prefill >= 1 && st != nullptr;
  if (!graphable) {
    int rc = decode_step_eager(h, mprobs, K, rng, uniforms, stream, speculate);
    if (rc == DD_OK) h->steps_since_prefill++;
    return rc;
  }
  if (!h->prefilled) {
    dd_set_error("dd_lm_decode_step: decode before prefill");
    return DD_ESTATE;
  }
  DD_REQUIRE(K >= 0 && K <= MAX_MEMBERS && (K == 0 || mprobs), "dd_lm_step: bad K / mprobs");
  RC(ensure_member_rows(h, K));
  if (h->T_host + 1 >= h->T_cap) {
    dd_set_error("dd_lm_step: KV cache full (%d tokens)", h->T_cap);
    return DD_ESTATE;
  }
  DD_REQUIRE(h->n_tok_host < MAX_NEW_TOKENS, "dd_lm_step: token buffer full");
  unsigned long long key = 1469598103934665603ull ^ (g_tune_epoch * 0x9e3779b97f4a7c15ull);
  auto mix = [&](unsigned long long v) { key = (key ^ v) * 1099511628211ull; };
  mix((unsigned long long)K);
  for (int k = 0; k < K; ++k) {
    unsigned long long bits;
    memcpy(&bits, &mprobs[k], 8);
    mix(bits);
  }
  mix((unsigned long long)ddk_attn_grid_tiles(h->T_host, h->T_cap));
  mix(((unsigned long long)h->L << 32) | (unsigned)h->span_start);   // launch arguments fixed by the last prefill
  mix(dd_rng_serial(rng));
  mix((unsigned long long)(uintptr_t)st);
  mix(speculate ? 1ull : 0ull);
  for (auto& g : h->graphs)
    if (g.key == key) {
      DD_HIP(hipGraphLaunch(g.exec, st));
      h->last_K = K, h->T_host += 1, h->n_tok_host += 1, h->steps_since_prefill++;
      if (h->cfg.leak_mask && K > 0) h->have_leak = true;
      return DD_OK;
    }
  // miss: capture this step, then launch it
  const int sT = h->T_host, sN = h->n_tok_host, sK = h->last_K, sB = h->bit0;
  const bool sL = h->have_leak;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    int rc = decode_step_eager(h, mprobs, K, rng, uniforms, stream, speculate);
    if (rc == DD_OK) h->steps_since_prefill++;
    return rc;
  }
  int rc = decode_step_eager(h, mprobs, K, rng, nullptr, stream, speculate);
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(st, &graph);
  hipGraphExec_t exec = nullptr;
  if (rc == DD_OK && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
    (void)hipGraphDestroy(graph);
    if (h->graphs.size() >= 12) {
      (void)hipGraphExecDestroy(h->graphs.front().exec);
      h->graphs.erase(h->graphs.begin());
    }
    h->graphs.push_back({key, exec});
    DD_HIP(hipGraphLaunch(exec, st));
    h->steps_since_prefill++;
    return DD_OK;          // host mirrors were advanced by the captured call
  }
  // capture failed: nothing was executed; restore the host mirrors and run the step eagerly
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipGetLastError();
  h->T_host = sT, h->n_tok_host = sN, h->last_K = sK, h->bit0 = sB, h->have_leak = sL;
  rc = decode_step_eager(h, mprobs, K, rng, uniforms, stream, speculate);
  if (rc == DD_OK) h->steps_since_prefill++;
  return rc;
}
extern "C" int dd_lm_decode_step(dd_lm* h, const double* mprobs, int K, dd_rng* rng, const float* uniforms, void* stream) {
  if (h && K > 16 && K <= MAX_MEMBERS) RC(ensure_member_rows(h, K));      // (before any capture starts)
  DD_REQUIRE(h, "dd_lm_decode_step: null handle");
  // queued steps never tell the host how their check went: "adaptive" speculates here like "always"
  return decode_step_queued(h, mprobs, K, rng, uniforms, stream, spec_mode_of(h) != 0);
}

// Launch `body`'s kernels on `st`: from the handle's graph cache when `key` is there, else captured now (and cached), else eagerly.
template <typename F>
static int replay_or_capture(dd_lm* h, unsigned long long key, hipStream_t st, bool use_graph, F&& body) {
  if (!use_graph) return body();
  for (auto& g : h->graphs)
    if (g.key == key) {
      DD_HIP(hipGraphLaunch(g.exec, st));
      return DD_OK;
    }
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return body();
  }
  int rc = body();
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(st, &graph);
  hipGraphExec_t exec = nullptr;
  if (rc == DD_OK && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
    (void)hipGraphDestroy(graph);
    if (h->graphs.size() >= 24) {
      (void)hipGraphExecDestroy(h->graphs.front().exec);
      h->graphs.erase(h->graphs.begin());
    }
    h->graphs.push_back({key, exec});
    DD_HIP(hipGraphLaunch(exec, st));
    return DD_OK;
  }
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipGetLastError();
  if (rc != DD_OK) return rc;
  return body();             // nothing was executed by the failed capture
}

// One ensemble step of ONE sequence with the fallback decided by the host: phase A (the combined sweep and the check) is launched,
// the host waits for the check's note in pinned memory (the only wait: a few microseconds after A's last kernel), and the members'
// re-run is launched only when the speculation failed — instead of ~350 launches that find nothing to do (≈0.3 ms of a 4 ms step).
// Results are those of dd_lm_decode_step in every case; *held (optional) reports what the check said.  Falls back to
// dd_lm_decode_step where the speculative step does not apply (K = 0 or > 8, injected uniforms, speculation switched off).
extern "C" int dd_lm_decode_step_sync(dd_lm* h, const double* mprobs, int K, dd_rng* rng, void* stream, int* held) {
  if (h && K > 16 && K <= MAX_MEMBERS) RC(ensure_member_rows(h, K));
  DD_REQUIRE(h, "dd_lm_decode_step_sync: null handle");
  DD_REQUIRE(h->tp_world == 1, "this handle is a tensor-parallel shard: drive it through dd_lm_tp_* (include/dropdec.h)");
  hipStream_t st = (hipStream_t)stream;
  if (held) *held = -1;
  const int mode = spec_mode_of(h);
  if (!(mode && K >= 1 && K <= 8 && mprobs && st != nullptr)) return decode_step_queued(h, mprobs, K, rng, nullptr, stream, mode != 0);
  if (mode == 2 && h->spec_cooldown > 0) {
    // adaptive: too few of the recent speculative steps held — the plain two-sweep step (queued, no host wait) for a while
    int rc = decode_step_queued(h, mprobs, K, rng, nullptr, stream, false);
    if (rc == DD_OK) {
      h->spec_cooldown--;
      h->spec_n[2]++;
    }
    return rc;
  }
  if (!h->prefilled) {
    dd_set_error("dd_lm_decode_step_sync: decode before prefill");
    return DD_ESTATE;
  }
  if (h->T_host + 1 >= h->T_cap) {
    dd_set_error("dd_lm_step: KV cache full (%d tokens)", h->T_cap);
    return DD_ESTATE;
  }
  DD_REQUIRE(h->n_tok_host < MAX_NEW_TOKENS, "dd_lm_step: token buffer full");
  DD_REQUIRE(h->cfg.mask_mode == DD_MASK_IBLIP_QUANTILE || rng, "dd_lm_step: an rng is required");
  const bool use_graph = g_use_graph && h->steps_since_prefill >= 1;
  unsigned long long key = 1469598103934665603ull ^ (g_tune_epoch * 0x9e3779b97f4a7c15ull);
  auto mix = [&](unsigned long long v) { key = (key ^ v) * 1099511628211ull; };
  mix(0x73796e63ull);
  mix((unsigned long long)K);
  for (int k = 0; k < K; ++k) {
    unsigned long long bits;
    memcpy(&bits, &mprobs[k], 8);
    mix(bits);
  }
  mix((unsigned long long)ddk_attn_grid_tiles(h->T_host, h->T_cap));
  mix(((unsigned long long)h->L << 32) | (unsigned)h->span_start);
  mix(dd_rng_serial(rng));
  mix((unsigned long long)(uintptr_t)st);
  const int sT = h->T_host, sN = h->n_tok_host;
  volatile int32_t* note = h->tok_host + MAX_NEW_TOKENS + 1;
  const int expect = ++h->spec_seq_host;
  int rc = replay_or_capture(h, key ^ 0xA1ull, st, use_graph, [&]() { return spec_phase_a(h, mprobs, K, rng, nullptr, st, true); });
  if (rc != DD_OK) {
    h->spec_seq_host--;
    return rc;
  }
  // the one wait of the step: the check's note (sequence number, then the verdict)
  const auto t_wait = std::chrono::steady_clock::now();
  for (unsigned long long spins = 0; note[0] != expect; ++spins) {
    if ((spins & 0xFFFFF) == 0xFFFFF) {
      if (hipStreamQuery(st) == hipSuccess && note[0] != expect) {
        dd_set_error("dd_lm_decode_step_sync: the stream drained without the speculation check reporting (expected note %d, saw %d)", expect, (int)note[0]);
        return DD_EHIP;
      }
      if (std::chrono::steady_clock::now() - t_wait > std::chrono::seconds(300)) {
        dd_set_error("dd_lm_decode_step_sync: no verdict from the speculation check after 300 s (note %d, expected %d)", (int)note[0], expect);
        return DD_EHIP;
      }
    }
    __builtin_ia32_pause();
  }
  const int ok = note[1];
  if (held) *held = ok;
  h->spec_n[ok ? 0 : 1]++;
  h->spec_rate = (1.0f - SPEC_RATE_WEIGHT) * h->spec_rate + (ok ? SPEC_RATE_WEIGHT : 0.0f);
  if (mode == 2 && h->spec_rate < SPEC_BREAK_EVEN) {
    h->spec_cooldown = SPEC_COOLDOWN_STEPS, h->spec_rate = SPEC_PROBE_RATE;
    h->spec_n[3]++;
  }
  if (!ok) RC(replay_or_capture(h, key ^ 0xB2ull, st, use_graph, [&]() { return spec_phase_b(h, mprobs, K, rng, nullptr, st); }));
  h->last_K = K;
  rc = replay_or_capture(h, key ^ 0xC3ull, st, use_graph, [&]() { return spec_phase_c(h, K, st); });
  // host mirrors: exactly one step, whether the phases ran on the host now (eager / capture) or were replayed
  h->T_host = sT, h->n_tok_host = sN, h->bit0 = 0;
  if (rc != DD_OK) return rc;
  h->last_K = K, h->T_host = sT + 1, h->n_tok_host = sN + 1, h->steps_since_prefill++;
  if (h->cfg.leak_mask) h->have_leak = true;
  return DD_OK;
}

// the second branch's stream and its fork / join events, created outside any capture
static int group_side_stream(dd_lm* h0) {
  const int want = g_branches > g_rider_branches ? g_branches : g_rider_branches;
  if (want < 2) return DD_OK;
  if (!h0->ev_fork) DD_HIP(hipEventCreateWithFlags(&h0->ev_fork, hipEventDisableTiming));
  for (int i = 0; i + 1 < want; ++i)
    if (!h0->side[i]) {
      if (g_mask_branches) {
        uint32_t mask[8];                                   // 256 CUs: side[0] the upper 128 mask bits, the others the lower 128
        for (int w = 0; w < 8; ++w) mask[w] = ((i == 0) == (w >= 4)) ? 0xFFFFFFFFu : 0u;
        DD_HIP(hipExtStreamCreateWithCUMask(&h0->side[i], 8, mask));
      } else {
        DD_HIP(hipStreamCreateWithFlags(&h0->side[i], hipStreamNonBlocking));
      }
      DD_HIP(hipEventCreateWithFlags(&h0->ev_join[i], hipEventDisableTiming));
    }
  return DD_OK;
}

// Replays the whole group step (n + 1 sweeps, ~1600 launches) from a hipGraph when nothing but device-side state has
// changed since it was captured; the cache lives in the first lane.
extern "C" int dd_lm_group_step(dd_lm* const* lanes, int n, const double* mprobs, int K, dd_rng* const* rngs, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  if (K > 16 && K <= MAX_MEMBERS)                    // (before any capture starts: growing the members' logits drains the device)
    for (int m = 0; lanes && m < n && m < GROUP_MAX_LANES; ++m) RC(ensure_member_rows(lanes[m], K));
  bool graphable = g_use_graph && st != nullptr && lanes && n >= 1 && n <= GROUP_MAX_LANES && lanes[0] && (K == 0 || mprobs);
  for (int m = 0; graphable && m < n; ++m)
    graphable = lanes[m] && lanes[m]->prefilled && lanes[m]->steps_since_prefill >= 1 &&
                lanes[m]->T_host + 1 < lanes[m]->T_cap && lanes[m]->n_tok_host < MAX_NEW_TOKENS;
  if (!graphable) {
    if (getenv("DD_DEBUG")) fprintf(stderr, "[dropdec] group step not graphable (eager)\n");
    if (lanes && n >= 1 && lanes[0]) RC(group_side_stream(lanes[0]));
    return group_step_eager(lanes, n, mprobs, K, rngs, stream_);
  }
  dd_lm* h0 = lanes[0];
  RC(group_side_stream(h0));
  unsigned long long key = 1469598103934665603ull ^ (g_tune_epoch * 0x9e3779b97f4a7c15ull);
  auto mix = [&](unsigned long long v) { key = (key ^ v) * 1099511628211ull; };
  mix(0x67726f7570ull + (unsigned long long)n + ((unsigned long long)g_branches << 40) + ((unsigned long long)g_rider_branches << 44));
  mix((unsigned long long)K);
  for (int k = 0; k < K; ++k) {
    unsigned long long bits;
    memcpy(&bits, &mprobs[k], 8);
    mix(bits);
  }
  int max_T = 0;
  for (int m = 0; m < n; ++m) {
    dd_lm* q = lanes[m];
    mix(q->serial);
    mix(dd_rng_serial(rngs ? rngs[m] : nullptr));
    mix((unsigned long long)ddk_attn_grid_tiles(q->T_host, q->T_cap));
    mix(((unsigned long long)q->L << 32) | (unsigned)q->span_start);
    if (q->T_host > max_T) max_T = q->T_host;
  }
  mix((unsigned long long)ddk_attn_grid_tiles(max_T, h0->T_cap));
  mix((unsigned long long)(uintptr_t)st);
  // rider form: which ring leaders have their un-masked rows parked, and the grids of the rows computed a step ahead
  int rider_gs = 8;
  const int rider_nbr = rider_branches(lanes, n, K, &rider_gs);
  mix(0x7269646572ull + (unsigned long long)rider_nbr + ((unsigned long long)rider_gs << 8));
  for (int m = 0; m < rider_gs * rider_nbr; ++m) {
    mix(rider_parked(lanes[m]) ? 1ull : 2ull);
    mix((unsigned long long)ddk_attn_grid_tiles(lanes[m]->T_host + 1, lanes[m]->T_cap));
  }
  auto advance = [&]() {
    for (int m = 0; m < n; ++m) {
      dd_lm* q = lanes[m];
      q->last_K = K, q->T_host += 1, q->n_tok_host += 1, q->steps_since_prefill++;
      if (q->cfg.leak_mask && K > 0) q->have_leak = true;
      q->pend_valid = m < rider_gs * rider_nbr, q->pend_step = q->steps_since_prefill;
    }
  };
  for (auto& g : h0->graphs)
    if (g.key == key) {
      DD_HIP(hipGraphLaunch(g.exec, st));
      advance();
      return DD_OK;
    }
  if (getenv("DD_DEBUG")) fprintf(stderr, "[dropdec] group step graph miss: capturing (cache holds %zu)\n", h0->graphs.size());
  struct Saved {
    int T, N, K, S, pend_step;
    bool leak, pend_valid;
  } sv[GROUP_MAX_LANES];
  for (int m = 0; m < n; ++m)
    sv[m] = {lanes[m]->T_host, lanes[m]->n_tok_host, lanes[m]->last_K, lanes[m]->steps_since_prefill, lanes[m]->pend_step,
             lanes[m]->have_leak, lanes[m]->pend_valid};
  auto restore = [&]() {
    for (int m = 0; m < n; ++m) {
      dd_lm* q = lanes[m];
      q->T_host = sv[m].T, q->n_tok_host = sv[m].N, q->last_K = sv[m].K, q->steps_since_prefill = sv[m].S, q->have_leak = sv[m].leak;
      q->pend_valid = sv[m].pend_valid, q->pend_step = sv[m].pend_step;
    }
  };
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return group_step_eager(lanes, n, mprobs, K, rngs, stream_);
  }
  int rc = group_step_eager(lanes, n, mprobs, K, rngs, stream_);
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(st, &graph);
  hipGraphExec_t exec = nullptr;
  if (rc == DD_OK && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
    (void)hipGraphDestroy(graph);
    if (h0->graphs.size() >= 12) {
      (void)hipGraphExecDestroy(h0->graphs.front().exec);
      h0->graphs.erase(h0->graphs.begin());
    }
    h0->graphs.push_back({key, exec});
    DD_HIP(hipGraphLaunch(exec, st));
    return DD_OK;          // host mirrors were advanced by the captured call
  }
  if (getenv("DD_DEBUG")) fprintf(stderr, "[dropdec] group step capture failed: rc=%d end_capture=%s graph=%p (%s)\n", rc, hipGetErrorString(e), (void*)graph, dd_last_error());
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipGetLastError();
  restore();               // nothing was executed
  return group_step_eager(lanes, n, mprobs, K, rngs, stream_);
}

extern "C" int dd_lm_set_speculation(dd_lm* h, int mode) {
  DD_REQUIRE(h && mode >= -1 && mode <= 2, "dd_lm_set_speculation: mode %d (-1 process default, 0 never, 1 always, 2 adaptive)", mode);
  h->spec_mode = mode;
  h->spec_rate = 1.0f, h->spec_cooldown = 0;
  return DD_OK;
}
extern "C" int dd_lm_spec_stats(dd_lm* h, int64_t* out4, int reset) {
  DD_REQUIRE(h && out4, "dd_lm_spec_stats: null argument");
  for (int i = 0; i < 4; ++i) out4[i] = h->spec_n[i];
  if (reset)
    for (int i = 0; i < 4; ++i) h->spec_n[i] = 0;
  return DD_OK;
}

extern "C" size_t dd_lm_xchg_stride(const dd_lm* h) {
  return h ? (size_t)h->Vpad + (size_t)h->Lyr * 2 * h->kv_dim : 0;
}

// ---- K-shard exchange -------------------------------------------------------------
__global__ void k_xchg_export_ids(const int32_t* tok, const int32_t* vote, int lo, int hi, int K, int32_t* out) {
  int m = threadIdx.x;
  if (m < K) {
    bool mine = m >= lo && m < hi;
    out[2 * m] = mine ? tok[m] : 0;
    out[2 * m + 1] = mine ? vote[m] : 0;
  }
}
__global__ void k_xchg_import_ids(const int32_t* in, int K, int32_t* tok, int32_t* vote) {
  int m = threadIdx.x;
  if (m < K) {
    tok[m] = in[2 * m];
    vote[m] = in[2 * m + 1];
  }
}
// rec = [logits Vpad][layer][k|v][kv_dim]; exported by the rank that ran the winner, zeros elsewhere
__global__ __launch_bounds__(256) void k_xchg_winner(const DDState* st, int lo, int hi, float* member_logits, float* knew,
                                                     float* vnew, int Vpad, int kv_dim, int n_layers, int rows_per_layer,
                                                     float* rec, int import) {
  int win = st->winner;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)Vpad + (size_t)n_layers * 2 * kv_dim;
  if (i >= total) return;
  float* slot;
  if (i < (size_t)Vpad) {
    slot = member_logits + (size_t)win * Vpad + i;
  } else {
    size_t j = i - Vpad;
    int layer = (int)(j / (2 * (size_t)kv_dim));
    int r = (int)(j % (2 * (size_t)kv_dim));
    float* base = (r < kv_dim) ? knew : vnew;
    slot = base + ((size_t)layer * rows_per_layer + win) * kv_dim + (r % kv_dim);
  }
  if (import) *slot = rec[i];
  else rec[i] = (win >= lo && win < hi) ? *slot : 0.f;
}

extern "C" int dd_lm_xchg_export_ids(dd_lm* h, int m_lo, int m_hi, int32_t* ids, void* stream_) {
  DD_REQUIRE(h && ids && h->last_K >= 1, "dd_lm_xchg_export_ids: bad state");
  DD_REQUIRE(h->cfg.vote_on != DD_VOTE_AVERAGE, "dd_lm_xchg_*: the mean over members needs every member's logits on one rank");
  const int32_t* vote = h->cfg.vote_on == DD_VOTE_HIDDEN ? h->member_vote : h->member_tok;
  k_xchg_export_ids<<<1, 64, 0, (hipStream_t)stream_>>>(h->member_tok, vote, m_lo, m_hi, h->last_K, ids);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
extern "C" int dd_lm_xchg_import_ids(dd_lm* h, const int32_t* ids, void* stream_) {
  DD_REQUIRE(h && ids && h->last_K >= 1, "dd_lm_xchg_import_ids: bad state");
  k_xchg_import_ids<<<1, 64, 0, (hipStream_t)stream_>>>(ids, h->last_K, h->member_tok, h->member_vote);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
static int xchg_winner(dd_lm* h, int lo, int hi, float* rec, int import, hipStream_t st) {
  size_t total = dd_lm_xchg_stride(h);
  k_xchg_winner<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(h->state, lo, hi, h->member_logits, h->knew, h->vnew,
                                                                h->Vpad, h->kv_dim, h->Lyr, KV_ROWS, rec, import);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
extern "C" int dd_lm_xchg_export_winner(dd_lm* h, int m_lo, int m_hi, float* rec, void* stream_) {
  DD_REQUIRE(h && rec && h->last_K >= 1, "dd_lm_xchg_export_winner: bad state");
  const int32_t* ids = h->cfg.vote_on == DD_VOTE_HIDDEN ? h->member_vote : h->member_tok;
  RC(dd_vote_gated(ids, h->last_K, &h->state->winner, &h->state->done, (hipStream_t)stream_));
  return xchg_winner(h, m_lo, m_hi, rec, 0, (hipStream_t)stream_);
}
extern "C" int dd_lm_xchg_import_winner(dd_lm* h, const float* rec, void* stream_) {
  DD_REQUIRE(h && rec && h->last_K >= 1, "dd_lm_xchg_import_winner: bad state");
  return xchg_winner(h, 0, 0, (float*)rec, 1, (hipStream_t)stream_);
}

// HF's greedy loop ends a sequence at the first EOS id (chair_test.py:341-346 relies on it).  The ids live in the
// sequence's device state: the step that emits one marks the sequence done, and steps enqueued beyond it change nothing
// that persists — in particular they draw nothing from the rng stream, which the reference continues into the next image
// (models/llava.py:16-20, :650).  The list survives prefills; n = 0 clears it.
extern "C" int dd_lm_set_eos(dd_lm* h, const int32_t* eos_ids_host, int n, void* stream_) {
  DD_REQUIRE(h && n >= 0 && n <= DD_MAX_EOS && (n == 0 || eos_ids_host), "dd_lm_set_eos: 0..%d ids", DD_MAX_EOS);
  EosList e;
  memset(&e, 0, sizeof(e));
  e.n = n;
  for (int i = 0; i < n; ++i) e.ids[i] = eos_ids_host[i];
  k_set_eos<<<1, 64, 0, (hipStream_t)stream_>>>(h->state, e);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

extern "C" int dd_lm_set_next_token(dd_lm* h, int32_t token, void* stream_) {
  DD_REQUIRE(h && token >= 0 && token < h->V, "dd_lm_set_next_token: bad token %d", token);
  k_set_token<<<1, 64, 0, (hipStream_t)stream_>>>(h->state, token);
  DD_CHECK_LAUNCH();
  h->pend_valid = false;
  return DD_OK;
}

// -----------------------------------------------------------------------------------------------
// read-backs
// -----------------------------------------------------------------------------------------------
extern "C" int dd_lm_get(dd_lm* h, int what, void* dst, size_t bytes, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h && dst, "dd_lm_get: null argument");
  DD_HIP(hipStreamSynchronize(st));
  // copies go through `st`, not through the legacy stream: a synchronous hipMemcpy would order itself after every blocking stream
  // of the process — including one another host thread is capturing a graph on (a second decode pipeline), which HIP refuses
  auto d2h = [&](void* to, const void* from, size_t n) -> hipError_t {
    hipError_t e = hipMemcpyAsync(to, from, n, hipMemcpyDeviceToHost, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
  };
  if (h->prefilled) {
    // the host mirrors count ENQUEUED steps; steps enqueued beyond an EOS did not advance the sequence (DDState::done)
    DDState ds;
    DD_HIP(d2h(&ds, h->state, sizeof(ds)));
    h->T_host = ds.T, h->n_tok_host = ds.n_tok < MAX_NEW_TOKENS ? ds.n_tok : MAX_NEW_TOKENS;
  }
  const void* src = nullptr;
  size_t avail = 0;
  const int K = h->last_K, L = h->L;
  switch (what) {
    case DD_GET_TOKENS: src = h->tokens, avail = (size_t)h->n_tok_host * 4; break;
    case DD_GET_LOGITS: src = h->last_logits, avail = (size_t)h->V * 4; break;
    case DD_GET_EPI: src = h->epi, avail = (size_t)L * 4; break;
    case DD_GET_ALEA: src = h->alea, avail = (size_t)L * 4; break;
    case DD_GET_VAR: src = h->var, avail = (size_t)L * 4; break;
    case DD_GET_UNCERT_SCALARS: src = h->scalars, avail = 12; break;
    case DD_GET_TOPK_IDS: src = h->topk_ids, avail = (size_t)L * h->cfg.k_top * 4; break;
    case DD_GET_TOPK_VALS: src = h->topk_vals, avail = (size_t)L * h->cfg.k_top * 4; break;
    case DD_GET_DROP: src = h->drop, avail = (size_t)K * L; break;
    case DD_GET_N_DROP: src = h->n_drop, avail = (size_t)K * 4; break;
    case DD_GET_MEMBER_ARGMAX: src = h->cfg.vote_on == DD_VOTE_HIDDEN ? h->member_vote : h->member_tok, avail = (size_t)K * 4; break;
    case DD_GET_WINNER: src = &h->state->winner, avail = 8; break;
    case DD_GET_BASE_LOGITS: src = h->base_logits, avail = (size_t)h->V * 4; break;
    case DD_GET_KEEP: src = h->keep, avail = (size_t)L; break;
    case DD_GET_SEQ_LEN: src = &h->state->T, avail = 4; break;
    case DD_GET_HIDDEN: src = h->last_hidden, avail = (size_t)h->d * 4; break;
    case DD_GET_SPEC_OK: src = h->spec_ok, avail = 4; break;
    case DD_GET_IMAGE_LOGITS: {
      DD_REQUIRE(bytes <= (size_t)L * h->V * 4, "dd_lm_get: image logits: at most %zu bytes", (size_t)L * h->V * 4);
      size_t rows = bytes / ((size_t)h->V * 4);
      DD_HIP(hipMemcpy2DAsync(dst, (size_t)h->V * 4, h->image_logits, (size_t)h->Vpad * 4, (size_t)h->V * 4, rows,
                              hipMemcpyDeviceToHost, st));
      DD_HIP(hipStreamSynchronize(st));
      return DD_OK;
    }
    case DD_GET_KV_SUMS:
      RC(ddk_kv_sums(h->kc, h->vc, h->Lyr, h->lsk, h->lsv, h->Hkv, h->T_cap, h->T_host, h->kv_sums, st, h->kv16));
      DD_HIP(hipStreamSynchronize(st));
      src = h->kv_sums, avail = (size_t)h->Lyr * 16;
      break;
    default: DD_REQUIRE(false, "dd_lm_get: unknown item %d", what);
  }
  DD_REQUIRE(bytes <= avail, "dd_lm_get(%d): asked for %zu bytes, only %zu available", what, bytes, avail);
  DD_HIP(d2h(dst, src, bytes));
  return DD_OK;
}

extern "C" double dd_lm_step_algorithmic_bytes(const dd_lm* h, int K) {
  if (!h) return 0;
  // Bytes_tok = sweeps * W_lm + sweeps * T * kv_tok, sweeps = 2 for dropout steps (base + packed
  // members), 1 for the stock greedy step.  W_lm in bf16; kv_tok at the cache's storage width (fp32 here).
  double params = (double)h->Lyr * ((double)(h->q_dim + 2 * h->kv_dim) * h->d + (double)h->d * h->q_dim + 3.0 * h->d * h->dff) +
                  (double)h->V * h->d;
  double w = params * (h->fp8 ? 1.0 : 2.0);
  double kv_tok = (double)h->Lyr * 2 * h->kv_dim * (h->kv16 ? 2.0 : 4.0);
  int sweeps = K > 0 ? 1 + (K + 7) / 8 : 1;
  return sweeps * (w + (double)h->T_host * kv_tok);
}

// Switches of the product library (the measurement hooks and the experiment knobs live in libdropdec_tools.so: dd_tools.hip):
// 8 = replay decode steps from hipGraphs (default 1), 11 = short prompt chunks through the decode GEMVs (default 1),
// 13 = slice-resident 16 / 32 / 64-row GEMVs (default 1; 0: the K-split-over-waves kernels, same bits), 14 = process default of the
// speculation policy (dd_lm_set_speculation), 15 / 16 = block order / big-block threshold of the prefill GEMM (same bits),
// 20 = form of that big block (1: LDS-DMA 160 x 512, the default; 0: the register-staged 128 x 512 block of round 3; same bits).
extern "C" int dd_set_tuning(int key, int value) {
  dd_engine_bump_epoch();
  DD_REQUIRE(key == 8 || key == 11 || (key >= 13 && key <= 16) || key == 20, "dd_set_tuning: unknown key %d (8, 11, 13, 14, 15, 16, 20)", key);
  if (key == 8) dd_engine_set_graph(value);
  else if (key == 11) dd_engine_set_extend_rows(value);
  else if (key == 13) ddk_set_gemv_slices(value);
  else if (key == 14) dd_engine_set_speculate(value);
  else if (key == 15) ddk_set_gemm_xcd_order(value);
  else if (key == 20) ddk_set_gemm_dma(value);
  else ddk_set_gemm_big_rows(value);
  return DD_OK;
}

// Non-blocking look at the tokens emitted so far (no stream synchronisation): the step kernels mirror every token into
// pinned host memory.  Returns the number of tokens copied to dst (<= max_tokens).  Lets generate() stop enqueueing
// steps as soon as an EOS shows up instead of discovering it a whole chunk later.
extern "C" int dd_lm_peek_tokens(dd_lm* h, int32_t* dst, int max_tokens) {
  if (!h || !dst || max_tokens <= 0) return 0;
  volatile int32_t* m = h->tok_host;
  int n = m[0];
  if (n > max_tokens) n = max_tokens;
  if (n > MAX_NEW_TOKENS) n = MAX_NEW_TOKENS;
  for (int i = 0; i < n; ++i) dst[i] = m[1 + i];
  return n;
}

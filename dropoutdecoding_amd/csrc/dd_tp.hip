// Tensor-parallel decode and prefill: the LM's matrices sharded over `world` ranks so that ONE
// sequence's step streams 1 / world of the weights per GPU — the only route to a single-stream multi-GPU speed-up (sharding
// the K members leaves every rank streaming all of W_lm twice per token).  Nothing in the reference to mirror: it runs the K
// members sequentially in one process (models/llava.py:342-359); this is the Megatron split on head / tile boundaries:
//
//   q / k / v   column-parallel by kv-head group: a rank projects, ropes, caches and attends its own heads       no exchange
//   o_proj      row-parallel over the rank's head columns  -> partial [rows][d]                                  SEAM 1
//   gate / up   column-parallel over d_ff / world                                                                no exchange
//   down_proj   row-parallel over the rank's d_ff slice    -> partial [rows][d]                                  SEAM 2
//   embeddings, norms, lm_head, scorer, masks, vote: replicated (every rank holds the same residual stream after a seam, so
//   every rank samples the same masks from its copy of the rng stream and votes the same winner — no exchange)
//
// A rank is an ordinary dd_lm created with its LOCAL head counts / d_ff (dd_lm_config.reserved = {world, rank}); what is new is
// the seam.  The row-parallel GEMV (GEMM in the prefill) stops at its plain product (EPI_STORE into slot `rank` of a gather
// buffer [world][rows][d]), the slots are exchanged, and every rank adds them IN RANK ORDER and runs the epilogue the
// un-sharded kernel fuses (residual, next operand, sum of squares: k_tp_finish / k_tp_add_rows).  Results are therefore
// deterministic for a given world size; against the un-sharded engine they differ by fp32 reassociation (a K-slice's MFMA
// chain cannot be cut at a rank boundary without changing its rounding), i.e. logits agree to ~1e-6 relative and tokens are
// equal wherever the top-2 margin exceeds that.  world = 1 through this path is bit-identical to the un-sharded engine.
//
// Two drivers share one code path (`ranks[0..n)`):
//   linked      n = world handles of ONE process on one device (tests, and what a single GPU can show): they share rank 0's
//               gather buffer, the phases of all ranks are issued in lock step on one stream, a seam moves nothing;
//   distributed n = 1 handle per process, one process per GPU: at a seam the engine calls the exchange the host registered
//               (dd_lm_tp_set_exchange: an all-gather of the slots over torch.distributed — RCCL over xGMI).
#include "dd_engine_internal.h"

#define RC(expr)                    \
  do {                              \
    int rc__ = (expr);              \
    if (rc__ != DD_OK) return rc__; \
  } while (0)

int dd_sample_masks_impl(const float* epi, int L, const double* mprobs, int K, const uint8_t* keep, int mode, int rng_mode,
                         const float* uniforms, uint32_t* rng_state, uint8_t* drop, int32_t* n_drop, int32_t* idx, uint8_t* drop_bits,
                         const int32_t* gate, hipStream_t st, const uint32_t* rng_in = nullptr, bool empty_keep = false);
int dd_argmax_rows_gated(const float* x, int R, int V, int ld, int32_t* out, const int32_t* gate, hipStream_t st);
uint32_t* dd_rng_state_ptr(dd_rng* r);

static int tp_check(dd_lm* const* R, int n, const char* who) {
  DD_REQUIRE(R && n >= 1 && R[0], "%s: null argument", who);
  const int W = R[0]->tp_world;
  DD_REQUIRE(n == W || n == 1, "%s: pass all %d linked ranks, or this process's one rank", who, W);
  for (int r = 0; r < n; ++r) {
    DD_REQUIRE(R[r] && R[r]->tp_world == W && R[r]->d == R[0]->d && R[r]->Lyr == R[0]->Lyr && R[r]->V == R[0]->V,
               "%s: rank %d is not a shard of the same model", who, r);
    DD_REQUIRE(n == 1 || R[r]->tp_rank == r, "%s: linked ranks must be passed in rank order", who);
    DD_REQUIRE(R[r]->tp_gather, "%s: rank %d has no gather buffer (dd_lm_tp_link / dd_lm_tp_set_exchange first)", who, r);
    DD_REQUIRE(n == 1 || R[r]->tp_gather == R[0]->tp_gather, "%s: the ranks are not linked to each other", who);
  }
  DD_REQUIRE(n == W || W == 1 || R[0]->tp_exchange, "%s: one rank of %d and no exchange registered", who, W);
  return DD_OK;
}

// slots are written: linked ranks wrote them in place; a lone rank of several asks the host to all-gather them
static int tp_seam(dd_lm* const* R, int n, int rows, hipStream_t st) {
  dd_lm* h = R[0];
  if (n == h->tp_world && !(n == 1 && h->tp_exchange)) return DD_OK;     // (a lone rank of ONE with an exchange registered calls it too:
  int rc = h->tp_exchange(h->tp_ctx, rows, (void*)st);                    //  the world-1 run of the distributed driver exercises the collective)
  if (rc != 0) {
    dd_set_error("tensor-parallel exchange callback failed (rc=%d)", rc);
    return DD_EHIP;
  }
  return DD_OK;
}

extern "C" int dd_lm_tp_link(dd_lm* const* ranks, int world, int rows_cap) {
  DD_REQUIRE(ranks && world >= 1 && world <= 8 && ranks[0], "dd_lm_tp_link: 1..8 ranks");
  dd_lm* h0 = ranks[0];
  if (rows_cap < h0->T_cap) rows_cap = h0->T_cap;
  for (int r = 0; r < world; ++r)
    DD_REQUIRE(ranks[r] && ranks[r]->tp_world == world && ranks[r]->tp_rank == r && ranks[r]->d == h0->d,
               "dd_lm_tp_link: handle %d is not rank %d of %d of this model", r, r, world);
  const size_t floats = (size_t)world * rows_cap * h0->d;
  if (h0->tp_gather_floats < floats) {
    RC(dd_engine_tp_alloc(h0, &h0->tp_gather, floats));
    h0->tp_gather_floats = floats;
  }
  for (int r = 1; r < world; ++r) ranks[r]->tp_gather = h0->tp_gather, ranks[r]->tp_gather_floats = h0->tp_gather_floats;
  return DD_OK;
}

// One rank per process: gather_dev [world][rows_cap][d] fp32 is the caller's (a torch tensor); at every seam the engine has
// written this rank's slot (slot stride = rows * d floats for the seam's `rows`) and calls exchange(ctx, rows, stream), which
// must all-gather the slots across the ranks in place, ordered on `stream`.
extern "C" int dd_lm_tp_set_exchange(dd_lm* h, float* gather_dev, size_t gather_floats, int (*exchange)(void*, int, void*), void* ctx) {
  DD_REQUIRE(h && gather_dev, "dd_lm_tp_set_exchange: null argument");
  // a prefill may bring up to T_cap - 1 rows to a seam: a buffer that only fits the decode rows would fail there, long after this call
  DD_REQUIRE(gather_floats >= (size_t)h->tp_world * h->T_cap * h->d,
             "dd_lm_tp_set_exchange: the gather buffer holds %zu floats, %d ranks x %d rows (the KV capacity) x %d need %zu", gather_floats,
             h->tp_world, h->T_cap, h->d, (size_t)h->tp_world * h->T_cap * h->d);
  DD_REQUIRE(h->tp_world == 1 || exchange, "dd_lm_tp_set_exchange: a rank of %d needs an exchange", h->tp_world);
  h->tp_gather = gather_dev, h->tp_gather_floats = gather_floats, h->tp_exchange = exchange, h->tp_ctx = ctx;
  return DD_OK;
}

// ---- prefill: all layers over the T0 prompt rows (every rank's h->px holds the same rows) ---------------------------------
static int tp_prefill_layers(dd_lm* const* R, int n, int T0, hipStream_t st) {
  dd_lm* h0 = R[0];
  const int W = h0->tp_world, d = h0->d;
  const size_t slot = (size_t)T0 * d;
  DD_REQUIRE((size_t)W * slot <= h0->tp_gather_floats, "tensor-parallel prefill: %d rows do not fit the gather buffer", T0);
  for (int l = 0; l < h0->Lyr; ++l) {
    for (int r = 0; r < n; ++r) {
      dd_lm* h = R[r];
      LayerW& w = h->lw[l];
      RC(ddk_rmsnorm_split(h->px, T0, d, w.norm1, h->cfg.rms_eps, h->p1_hi, h->p1_lo, nullptr, nullptr, st, h->wf));
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.wf = h->wf;
      g.a_hi = h->p1_hi, g.a_lo = h->p1_lo, g.M = T0, g.S = h->S_d, g.n_tiles = h->qkv_tiles, g.W = w.wqkv;
      g.qbuf = h->pq, g.kc = h->kc + (size_t)l * h->lsk, g.vc = h->vc + (size_t)l * h->lsv, g.T_cap = h->T_cap;
      g.q_tiles = h->q_tiles, g.k_tiles = h->k_tiles, g.q_dim = h->q_dim, g.kv_dim = h->kv_dim, g.pos0 = 0, g.kv16 = h->kv16;
      g.rope_cos = h->rope_cos, g.rope_sin = h->rope_sin;
      RC(ddk_gemm(EPI_QKV, g, st));
      RC(ddk_attn_prefill(h->pq, g.kc, g.vc, T0, h->T_cap, h->H, h->Hkv, h->p1_hi, h->p1_lo, nullptr, 0, 0, 0, 0, st, nullptr, h->kv16, h->wf));
      memset(&g, 0, sizeof(g));
      g.wf = h->wf;
      g.a_hi = h->p1_hi, g.a_lo = h->p1_lo, g.M = T0, g.S = h->S_q, g.n_tiles = d / 16, g.W = w.wo;
      g.out = h->tp_gather + (size_t)h->tp_rank * slot, g.ldo = d, g.n_valid = d;       // this rank's partial o_proj product
      RC(ddk_gemm(EPI_STORE, g, st));
    }
    RC(tp_seam(R, n, T0, st));
    // (every linked rank reads all slots before any of them writes its next partial into the shared buffer)
    for (int r = 0; r < n; ++r) RC(ddk_tp_add_rows(R[r]->px, R[r]->tp_gather, W, slot, st));
    for (int r = 0; r < n; ++r) {
      dd_lm* h = R[r];
      LayerW& w = h->lw[l];
      RC(ddk_rmsnorm_split(h->px, T0, d, w.norm2, h->cfg.rms_eps, h->p1_hi, h->p1_lo, nullptr, nullptr, st, h->wf));
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.wf = h->wf;
      g.a_hi = h->p1_hi, g.a_lo = h->p1_lo, g.M = T0, g.S = h->S_d, g.n_tiles = 2 * h->dff / 16, g.W = w.wgu;
      g.o_hi = h->p2_hi, g.o_lo = h->p2_lo, g.ld_planes = h->dff;
      RC(ddk_gemm(EPI_SILU, g, st));
      memset(&g, 0, sizeof(g));
      g.wf = h->wf;
      g.a_hi = h->p2_hi, g.a_lo = h->p2_lo, g.M = T0, g.S = h->S_ff, g.n_tiles = d / 16, g.W = w.wdown;
      g.out = h->tp_gather + (size_t)h->tp_rank * slot, g.ldo = d, g.n_valid = d;
      RC(ddk_gemm(EPI_STORE, g, st));
    }
    RC(tp_seam(R, n, T0, st));
    for (int r = 0; r < n; ++r) RC(ddk_tp_add_rows(R[r]->px, R[r]->tp_gather, W, slot, st));
  }
  return DD_OK;
}

// dd_lm_prefill for a sharded model: embeds_dev [T0][d] fp32 (the same on every rank), visual span as there.
extern "C" int dd_lm_tp_prefill(dd_lm* const* ranks, int n, const float* embeds, int T0, int span_start, int span_len, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  RC(tp_check(ranks, n, "dd_lm_tp_prefill"));
  dd_lm* h0 = ranks[0];
  DD_REQUIRE(embeds, "dd_lm_tp_prefill: null argument");
  DD_REQUIRE(T0 >= 1 && T0 < h0->T_cap, "dd_lm_tp_prefill: T0=%d out of range (KV capacity %d)", T0, h0->T_cap);
  DD_REQUIRE(span_len >= 1 && span_len <= h0->Lmax && span_start >= 0 && span_start + span_len <= T0,
             "dd_lm_tp_prefill: visual span [%d, %d) does not fit the %d input positions (max_visual %d)", span_start,
             span_start + span_len, T0, h0->Lmax);
  for (int r = 0; r < n; ++r)
    DD_HIP(hipMemcpyAsync(ranks[r]->px, embeds, (size_t)T0 * h0->d * 4, hipMemcpyDeviceToDevice, st));
  RC(tp_prefill_layers(ranks, n, T0, st));
  for (int r = 0; r < n; ++r) RC(dd_engine_prefill_tail(ranks[r], ranks[r]->px, T0, span_start, span_len, st));
  return DD_OK;
}

// ---- one packed sweep of nb <= 8 rows: lm_sweep's single-sequence form with the two seams per layer ------------------------
// bits[r] / logits[r]: rank r's drop-bit plane (members) or leak bits (un-masked row) and where its logits go
static int tp_sweep(dd_lm* const* R, int n, int nb, const uint8_t* const* bits, int row0, float* const* logits, hipStream_t st) {
  dd_lm* h0 = R[0];
  const int W = h0->tp_world, d = h0->d;
  const size_t slot = (size_t)nb * d;
  for (int r = 0; r < n; ++r) {
    dd_lm* h = R[r];
    RC(ddk_embed_rows(h->embed, d, h->state, h->xa, h->lw[0].norm1, h->xop_d, h->ssq_a, d / 16, st, nullptr, h->wf));
  }
  int ssq_n = 1;
  for (int l = 0; l < h0->Lyr; ++l) {
    for (int r = 0; r < n; ++r) {
      dd_lm* h = R[r];
      LayerW& w = h->lw[l];
      float* knew = h->knew + ((size_t)l * KV_ROWS_PER_LAYER + row0) * h->kv_dim;
      float* vnew = h->vnew + ((size_t)l * KV_ROWS_PER_LAYER + row0) * h->kv_dim;
      GemvArgs a;
      memset(&a, 0, sizeof(a));
      a.wf = h->wf;
      a.W = w.wqkv, a.S = h->S_d, a.n_tiles = h->qkv_tiles, a.nb = nb, a.xop = h->xop_d;
      a.ssq_in = h->ssq_a, a.ssq_n = ssq_n, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
      a.qbuf = h->qbuf, a.knew = knew, a.vnew = vnew, a.q_tiles = h->q_tiles, a.k_tiles = h->k_tiles;
      a.q_dim = h->q_dim, a.kv_dim = h->kv_dim, a.rope_cos = h->rope_cos, a.rope_sin = h->rope_sin, a.state = h->state;
      RC(ddk_gemv(EPI_QKV, a, st));
      AttnDecodeArgs t;
      memset(&t, 0, sizeof(t));
      t.wf = h->wf;
      t.qbuf = h->qbuf, t.kc = h->kc + (size_t)l * h->lsk, t.vc = h->vc + (size_t)l * h->lsv, t.T_cap = h->T_cap, t.kv16 = h->kv16;
      t.T = h->T_host, t.state = h->state, t.nb = nb, t.n_heads = h->H, t.n_kv = h->Hkv, t.drop_bits = bits[r];
      t.bit0 = bits[r] ? h->bit0 : 0;
      t.span_start = h->span_start, t.span_len = h->L, t.part_o = h->part_o, t.part_ml = h->part_ml;
      t.knew = knew, t.vnew = vnew, t.xop_out = h->xop_q;
      RC(ddk_attn_decode(t, st));
      memset(&a, 0, sizeof(a));
      a.wf = h->wf;
      a.W = w.wo, a.S = h->S_q, a.n_tiles = d / 16, a.nb = nb, a.xop = h->xop_q;
      a.out = h->tp_gather + (size_t)h->tp_rank * slot, a.ldo = d, a.n_valid = d;          // partial o_proj product
      RC(ddk_gemv(EPI_STORE, a, st));
    }
    RC(tp_seam(R, n, nb, st));
    // (every linked rank reads all slots before any of them writes its next partial into the shared buffer)
    for (int r = 0; r < n; ++r)
      RC(ddk_tp_finish(R[r]->tp_gather, W, slot, nb, R[r]->xa, d, R[r]->lw[l].norm2, R[r]->xop_d, R[r]->ssq_b, d / 16, R[r]->wf, st));
    for (int r = 0; r < n; ++r) {
      dd_lm* h = R[r];
      LayerW& w = h->lw[l];
      GemvArgs a;
      memset(&a, 0, sizeof(a));
      a.wf = h->wf;
      a.W = w.wgu, a.S = h->S_d, a.n_tiles = h->dff / 16, a.nb = nb, a.xop = h->xop_d;
      a.ssq_in = h->ssq_b, a.ssq_n = d / 16, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps, a.xop_next = h->xop_ff, a.S_next = h->S_ff;
      RC(ddk_gemv(EPI_SILU, a, st));
      memset(&a, 0, sizeof(a));
      a.wf = h->wf;
      a.W = w.wdown, a.S = h->S_ff, a.n_tiles = d / 16, a.nb = nb, a.xop = h->xop_ff;
      a.out = h->tp_gather + (size_t)h->tp_rank * slot, a.ldo = d, a.n_valid = d;
      RC(ddk_gemv(EPI_STORE, a, st));
    }
    RC(tp_seam(R, n, nb, st));
    for (int r = 0; r < n; ++r) {
      dd_lm* h = R[r];
      const float* nw = (l + 1 < h->Lyr) ? h->lw[l + 1].norm1 : h->final_norm;
      RC(ddk_tp_finish(h->tp_gather, W, slot, nb, h->xa, d, nw, h->xop_d, h->ssq_a, d / 16, h->wf, st));
    }
    ssq_n = d / 16;
  }
  for (int r = 0; r < n; ++r) {          // lm_head: replicated (2 % of the weights), every rank holds the full logits
    dd_lm* h = R[r];
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    a.wf = h->wf;
    a.W = h->lm_head, a.S = h->S_d, a.n_tiles = h->Vpad / 16, a.nb = nb, a.xop = h->xop_d;
    a.ssq_in = h->ssq_a, a.ssq_n = ssq_n, a.ssq_ld = d / 16, a.inv_k = 1.0f / d, a.eps = h->cfg.rms_eps;
    a.out = logits[r], a.ldo = h->Vpad, a.n_valid = h->V, a.state = h->state;
    RC(ddk_gemv(EPI_STORE, a, st));
  }
  return DD_OK;
}

// dd_lm_decode_step for a sharded model (1 <= K <= 8, or K = 0: stock greedy): un-masked sweep, keep set, masks, packed member
// sweep, vote, commit — the two-sweep form, every small kernel replicated per rank.  rngs[r]: rank r's copy of the stream.
static int tp_step_body(dd_lm* const* ranks, int n, const double* mprobs, int K, dd_rng* const* rngs, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  const uint8_t* bits[8];
  float* logits[8];
  for (int r = 0; r < n; ++r) {
    dd_lm* h = ranks[r];
    if (!h->prefilled) {
      dd_set_error("dd_lm_tp_decode_step: decode before prefill");
      return DD_ESTATE;
    }
    if (h->T_host + 1 >= h->T_cap) {
      dd_set_error("dd_lm_tp_decode_step: KV cache full (%d tokens)", h->T_cap);
      return DD_ESTATE;
    }
    DD_REQUIRE(K == 0 || h->cfg.mask_mode == DD_MASK_IBLIP_QUANTILE || (rngs && rngs[r]), "dd_lm_tp_decode_step: rank %d needs its rng", r);
    RC(dd_engine_step_begin(h, st));
    h->bit0 = 0;
    bits[r] = h->cfg.leak_mask ? h->leak_bits : nullptr;
    logits[r] = h->base_logits;
  }
  RC(tp_sweep(ranks, n, 1, bits, 0, logits, st));
  for (int r = 0; r < n; ++r) {
    dd_lm* h = ranks[r];
    const int32_t* gate = &h->state->done;
    RC(dd_argmax_rows_gated(h->base_logits, 1, h->V, h->Vpad, h->argmax_base, gate, st));
    h->last_K = K;
    if (K == 0) continue;
    RC(dd_engine_step_keep(h, gate, st));
    RC(dd_sample_masks_impl(h->epi, h->L, mprobs, K, h->keep, h->cfg.mask_mode, DD_RNG_MT19937, nullptr,
                            dd_rng_state_ptr(rngs ? rngs[r] : nullptr), h->drop, h->n_drop, nullptr, h->drop_bits, gate, st));
    bits[r] = h->drop_bits, logits[r] = h->member_logits;
  }
  if (K > 0) {
    RC(tp_sweep(ranks, n, K, bits, 0, logits, st));
    for (int r = 0; r < n; ++r) {
      dd_lm* h = ranks[r];
      RC(dd_argmax_rows_gated(h->member_logits, K, h->V, h->Vpad, h->member_tok, &h->state->done, st));
      if (h->cfg.vote_on == DD_VOTE_HIDDEN) {
        RC(ddk_final_norm_rows(h->xa, K, h->d, h->final_norm, h->cfg.rms_eps, h->hidden, st));
        RC(dd_argmax_rows_gated(h->hidden, K, h->d, h->d, h->member_vote, &h->state->done, st));
      }
    }
  }
  for (int r = 0; r < n; ++r) {
    RC(dd_lm_step_commit(ranks[r], K, stream_));
    ranks[r]->steps_since_prefill++;
  }
  return DD_OK;
}

// Linked ranks issue no host callback, so their whole step (2 sweeps x world ranks: ~3,000 launches at 8 ranks and 32 layers)
// replays from a hipGraph like dd_lm_decode_step's; a distributed rank's exchange is a host call at every seam and stays eager.
int dd_engine_use_graph();
unsigned long long dd_rng_serial(dd_rng* r);
extern "C" int dd_lm_tp_decode_step(dd_lm* const* ranks, int n, const double* mprobs, int K, dd_rng* const* rngs, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  RC(tp_check(ranks, n, "dd_lm_tp_decode_step"));
  DD_REQUIRE(K >= 0 && K <= 8 && (K == 0 || mprobs), "dd_lm_tp_decode_step: K=%d (0..8)", K);
  dd_lm* h0 = ranks[0];
  bool graphable = dd_engine_use_graph() && st != nullptr && n == h0->tp_world && !h0->tp_exchange;   // a host exchange cannot be replayed
  for (int r = 0; graphable && r < n; ++r)
    graphable = ranks[r]->prefilled && ranks[r]->steps_since_prefill >= 1 && ranks[r]->T_host + 1 < ranks[r]->T_cap &&
                ranks[r]->n_tok_host < MAX_NEW_TOKENS;
  if (!graphable) return tp_step_body(ranks, n, mprobs, K, rngs, stream_);
  unsigned long long key = 1469598103934665603ull ^ (dd_engine_epoch() * 0x9e3779b97f4a7c15ull);
  auto mix = [&](unsigned long long v) { key = (key ^ v) * 1099511628211ull; };
  mix(0x7470ull + (unsigned long long)n);
  mix((unsigned long long)K);
  for (int k = 0; k < K; ++k) {
    unsigned long long b;
    memcpy(&b, &mprobs[k], 8);
    mix(b);
  }
  for (int r = 0; r < n; ++r) {
    mix(ranks[r]->serial);
    mix(dd_rng_serial(rngs ? rngs[r] : nullptr));
  }
  mix((unsigned long long)ddk_attn_grid_tiles(h0->T_host, h0->T_cap));
  mix(((unsigned long long)h0->L << 32) | (unsigned)h0->span_start);
  mix((unsigned long long)(uintptr_t)st);
  auto advance = [&]() {
    for (int r = 0; r < n; ++r) {
      dd_lm* q = ranks[r];
      q->last_K = K, q->T_host += 1, q->n_tok_host += 1, q->steps_since_prefill++;
      if (q->cfg.leak_mask && K > 0) q->have_leak = true;
    }
  };
  for (auto& g : h0->graphs)
    if (g.key == key) {
      DD_HIP(hipGraphLaunch(g.exec, st));
      advance();
      return DD_OK;
    }
  struct Saved {
    int T, N, K, S;
    bool leak;
  } sv[8];
  for (int r = 0; r < n; ++r) sv[r] = {ranks[r]->T_host, ranks[r]->n_tok_host, ranks[r]->last_K, ranks[r]->steps_since_prefill, ranks[r]->have_leak};
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return tp_step_body(ranks, n, mprobs, K, rngs, stream_);
  }
  int rc = tp_step_body(ranks, n, mprobs, K, rngs, stream_);
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
    return DD_OK;          // the captured call advanced the host mirrors
  }
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipGetLastError();
  for (int r = 0; r < n; ++r) {      // nothing was executed
    dd_lm* q = ranks[r];
    q->T_host = sv[r].T, q->n_tok_host = sv[r].N, q->last_K = sv[r].K, q->steps_since_prefill = sv[r].S, q->have_leak = sv[r].leak;
  }
  if (rc != DD_OK) return rc;
  return tp_step_body(ranks, n, mprobs, K, rngs, stream_);
}

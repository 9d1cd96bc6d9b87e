/*
 * dropdec.h — C-ABI of the MI355X-native Dropout-Decoding hot path (libdropdec.so).
 *
 * The reference (kigb/DropoutDecoding) is pure Python: its "FFI" for this path is the set of
 * Python methods its forward() overrides call.  Each entry point below names the reference
 * function (file:line in the reference tree) it replaces.  The Python host side
 * (dropoutdecoding_amd/) binds these with ctypes; INTEGRATION.md shows the stub a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative DD_E* code and never throws;
 *     dd_last_error() returns a human-readable message for the calling thread's last failure;
 *   - pointers named *_dev are device (HBM) pointers owned by the caller unless stated
 *     otherwise; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - nothing here allocates device memory except dd_lm_create/dd_lm_load_* (and
 *     dd_rng_create), and nothing synchronises the device except the dd_*_read/_get copies
 *     that say so;
 *   - integer results (token ids, top-k ids, mask flags, winner index) are bit-exact with the
 *     reference's CPU path on identical inputs; floating-point results agree to the tolerances
 *     stated in tests/ (DESIGN.md "Numerics").
 */
#ifndef DROPDEC_H
#define DROPDEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD_OK 0
#define DD_EINVAL (-1)     /* bad argument / unsupported shape */
#define DD_EHIP (-2)       /* a HIP runtime call failed */
#define DD_ENOMEM (-3)
#define DD_ESTATE (-4)     /* call sequence error (e.g. decode before prefill) */

#define DD_MAX_MEMBERS_PER_PASS 8   /* ensemble members packed into one weight sweep */
#define DD_MAX_MEMBERS 64           /* K = len(settings['voting_numbers']) per decode step: 1..64 (ceil(K / 8) packed sweeps; the mask
                                     * sampler's per-member table and the per-layer rows of new K / V hold 64); the reference accepts any
                                     * list length (models/llava.py:340) but ships 3 and 4, BASELINE uses 8 */
#define DD_MAX_TOPK 16

/* mask-sampler modes: which family's get_image_attention_mask() semantics to follow */
#define DD_MASK_LLAVA_CUMULATIVE 0  /* models/llava.py:589-662, mask not reset between members (:344) */
#define DD_MASK_NEXT_RESET 1        /* models/llavanext.py:779-808, reset at :546                       */
#define DD_MASK_NEXT_NO_OVERLAP 2   /* models/llavanext.py:809-829 ("epis_no_overlap", no keep-restore)  */
#define DD_MASK_IBLIP_QUANTILE 3    /* models/instructblip.py:447-460, deterministic quantile, reset :121 */
#define DD_MASK_LLAVA_CUMULATIVE_NO_OVERLAP 4 /* models/llava.py:663-683 ("epis_no_overlap") at the :344 call site:
                                               * cumulative like mode 0, no keep-restore (dormant in the reference) */

#define DD_MASK_IBLIP_KL 5          /* models/instructblip.py:464-485 ("epis_kl", the commented call at :123; dormant): stochastic rule
                                     * and reset like mode 1, but the tokens restored are the 10 % with the lowest
                                     * KL(step || token) — keep flags from dd_kl_keep — instead of the overlap keep set */

/* where the dropout uniforms come from (models/llava.py:650 `torch.rand_like`) */
#define DD_RNG_INJECTED 0           /* caller supplies uniforms[K][L] (parity tests)                     */
#define DD_RNG_MT19937 1            /* a dd_rng kept in device memory: the torch-CPU mt19937 stream (dd_rng_create) or
                                     * the torch-GPU Philox stream (dd_rng_create_philox) — the handle knows which  */
#define DD_RNG_DEVICE DD_RNG_MT19937

/* what the ensemble votes on */
#define DD_VOTE_LOGITS 0            /* models/llava.py:27, models/llavanext.py:31                         */
#define DD_VOTE_HIDDEN 1            /* models/instructblip.py:125-137 (argmax over the final hidden state) */
#define DD_VOTE_AVERAGE 2           /* models/llava.py:37-52 select_by_average (dormant): logits := fp32 mean over the
                                     * members in list order, token = argmax(mean), member 0's KV is kept */

int dd_version(void);
const char* dd_last_error(void);
/* compiled-for architecture string, e.g. "gfx950" */
const char* dd_arch(void);

/* ------------------------------------------------------------------------------------------
 * RNG: torch's CPU default generator restated on the device.
 * Replaces torch.manual_seed(seed) (models/llava.py:16-20, llavanext.py:18-21,
 * instructblip.py:17-21) + the stream torch.rand_like() consumes (models/llava.py:650).
 * State = 625 uint32 words in device memory (624 mt words + read index).
 * ------------------------------------------------------------------------------------------ */
typedef struct dd_rng dd_rng;
int dd_rng_create(uint32_t seed, dd_rng** out);
int dd_rng_destroy(dd_rng* r);
int dd_rng_seed(dd_rng* r, uint32_t seed, void* stream);
/* out_dev[n] = next n float32 uniforms of the stream, (x & 0xFFFFFF) * 2^-24 */
int dd_rng_uniform(dd_rng* r, float* out_dev, int n, void* stream);
/* torch's GPU default generator restated: what the reference draws when the model sits on a GPU
 * (models/llava.py:650 with epis_uncert on the device; seed from models/llava.py:16-20).
 * Philox4x32-10, key = seed, 64-bit offset advanced by 4 per rand_like; one rand_like over
 * n <= 524288 float32 elements gives element i = first word of counter (offset/4, subsequence i)
 * mapped by fma(x, 2^-32, 2^-32) with 1.0 folded to 0.0 (ATen's elementwise random kernel over
 * rocRAND's uniform).  The state lives in the same 625-word device block (words 0..3, tag in
 * word 624), so every entry point that takes a dd_rng accepts it and graph replays advance it on
 * the device.  dd_rng_seed on such a handle = torch.manual_seed (offset back to 0);
 * dd_rng_uniform = one torch.rand(n, device="cuda").  `offset` must be a multiple of 4. */
int dd_rng_create_philox(unsigned long long seed, unsigned long long offset, dd_rng** out);

/* ------------------------------------------------------------------------------------------
 * Per-visual-token uncertainty scorer.
 * Replaces calculate_vision_uncertainty(logits) (models/llava.py:710-756; copies at
 * llavanext.py:878-924, instructblip.py:511-557) and get_topk_token_id (llava.py:428-441).
 *   logits_dev [L][ld] fp32 (ld >= V; columns >= V ignored)
 *   var_tok/epi_tok/alea_tok [L] fp32; scalars3 = {mean var, mean epi, mean alea}
 *   topk_vals [L][k] fp32 / topk_ids [L][k] int32 (either may be NULL), descending, ties ->
 *   lowest id first
 *   workspace_dev: at least dd_uncertainty_workspace_bytes(L, V) bytes
 * Softmax statistics: {max, sum exp(x - max)} per row and 64-column block, combined per row in a
 * fixed order — inside dd_lm_prefill those block statistics come out of the lm_head GEMM's epilogue
 * (the softmax reduction fused into the unembedding product: the [L][V] logits are then read twice,
 * for the column mean and for epi / alea / var + top-k); this stand-alone entry point computes them
 * from the stored logits with the same arithmetic, so both give the same bits.
 * ------------------------------------------------------------------------------------------ */
size_t dd_uncertainty_workspace_bytes(int L, int V);
int dd_vision_uncertainty(const float* logits_dev, int L, int V, int ld,
                          float* var_tok_dev, float* epi_tok_dev, float* alea_tok_dev, float* scalars3_dev,
                          int k_top, float* topk_vals_dev, int32_t* topk_ids_dev,
                          void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Keep set. Replaces get_overlap_image_tokens (models/llava.py:443-482).
 *   keep_dev[l] = 1 iff argmax(step_logits) is among topk_ids[l][0..k)
 *   argmax_dev (optional) receives the argmax (first index among equal maxima)
 * ------------------------------------------------------------------------------------------ */
int dd_overlap_keep(const float* step_logits_dev, int V, const int32_t* topk_ids_dev, int L, int k,
                    uint8_t* keep_dev, int32_t* argmax_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * "epis_kl" keep set. Replaces lowest_percent_kl_indices (models/instructblip.py:559-578; the same function at llava.py:758):
 *   kl_dev[l] = sum_v softmax(step_logits)[v] * (log softmax(step_logits)[v] - log_softmax(image_logits[l])[v]),
 *   keep_dev[l] = 1 for the int(0.1 * L) smallest (value, then index).  image_logits_dev [L][ld] fp32 (the prefill logits over
 *   the visual span, models/llava.py:412-426), kl_dev [L] fp32 scratch / output.
 * ------------------------------------------------------------------------------------------ */
int dd_kl_keep(const float* step_logits_dev, const float* image_logits_dev, int L, int V, int ld, uint8_t* keep_dev,
               float* kl_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Uncertainty-guided visual-token dropout masks for all K members of one step.
 * Replaces get_image_attention_mask(method="epis"/"epis_no_overlap") for the three families
 * (models/llava.py:589-683, llavanext.py:779-829, instructblip.py:447-460) including the
 * per-family reset / cumulative / keep-restore behaviour of the calling loop
 * (llava.py:342-346, llavanext.py:544-551, instructblip.py:119-122).
 *   epi_dev [L] fp32; mprobs_host [K] doubles = settings['voting_numbers'] (models/config.py:2)
 *   keep_dev [L] u8 (from dd_overlap_keep); ignored for the two *_NO_OVERLAP modes
 *   rng_mode DD_RNG_INJECTED: uniforms_dev [K][L] fp32;  DD_RNG_MT19937: rng != NULL, the
 *   stream advances by K*L draws (member-major), exactly like K rand_like(epi) calls
 *   drop_dev [K][L] u8: 1 = attention mask set to 0 for that member
 *   n_drop_dev [K] int32 (= the reference's masked_numbers, llava.py:661-662)
 *   idx_dev (optional) [K][L] int32: ascending indices of dropped tokens, -1 padded
 * K <= 64, L <= 8192.
 * ------------------------------------------------------------------------------------------ */
int dd_sample_masks(const float* epi_dev, int L, const double* mprobs_host, int K,
                    const uint8_t* keep_dev, int mode, int rng_mode, const float* uniforms_dev, dd_rng* rng,
                    uint8_t* drop_dev, int32_t* n_drop_dev, int32_t* idx_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Majority vote. Replaces select_by_vote (models/llava.py:22-36, llavanext.py:26-39).
 *   argmax_ids_dev [K] int32 -> out2_dev = {winner member index, majority token id}
 *   ties: the id inserted first (lowest member index) wins; winner = first member with it.
 * ------------------------------------------------------------------------------------------ */
int dd_vote(const int32_t* argmax_ids_dev, int K, int32_t* out2_dev, void* stream);

/* rows [R][ld] fp32 -> argmax_dev [R] int32 over the first V columns (torch.argmax semantics) */
int dd_argmax_rows(const float* x_dev, int R, int V, int ld, int32_t* argmax_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * The language-model engine: the K-way masked-context decode step.
 * Replaces the third-party LM the reference calls 1+K times per token
 * (models/llava.py:294-303,350-359; llavanext.py:505-514,553-562; instructblip.py:68-82,125-140)
 * together with the KV deep copies (llava.py:292,343), the per-step mask/position rebuild
 * (llava.py:254-283), the ensemble loop (llava.py:338-359), the vote and the return of the
 * winner's logits + cache (llava.py:361-376).
 * Weights are bf16, activations/accumulation fp32, KV cache fp32 (DESIGN.md "Numerics").
 * ------------------------------------------------------------------------------------------ */
typedef struct dd_lm_config {
  int32_t vocab_size;        /* V (real)                                   */
  int32_t hidden_size;       /* d, multiple of 256                          */
  int32_t intermediate_size; /* d_ff, multiple of 256                       */
  int32_t num_layers;
  int32_t num_heads;
  int32_t num_kv_heads;
  int32_t head_dim;          /* must be 128                                 */
  float rms_eps;
  float rope_theta;
  int32_t max_seq;           /* KV capacity in tokens                       */
  int32_t max_visual;        /* max visual-span length L                    */
  int32_t k_top;             /* 5 (LLaVA-1.5) or 10 (NeXT, InstructBLIP)    */
  int32_t mask_mode;         /* DD_MASK_* (0..5)                            */
  int32_t vote_on;           /* DD_VOTE_*                                   */
  int32_t leak_mask;         /* InstructBLIP Q2: 1 = the un-masked pass sees the last member's zeros (positions from the
                                cache length, transformers 5.x); 2 = additionally position = T - #zeros (the 4.44 rule) */
  int32_t weight_format;     /* 0 = bf16 matrices, 1 = OCP fp8 e4m3fn matrices + per-output-row fp32 scales (BASELINE config 5),
                                2 = fp16 matrices / embeddings / norm vectors: fp16-native checkpoints (every model the reference
                                loads: chair_test/chair_test.py:189-213, torch_dtype=float16) stay EXACT instead of losing three
                                mantissa bits in a cast to bf16; activations are then split hi + lo in fp16 (~22 mantissa bits) and
                                every weight product runs on v_mfma_f32_16x16x32_f16 */
  int32_t kv_format;         /* KV cache storage: 0 = fp32 (default), 1 = fp16 — the width the reference keeps its cache in
                                (chair_test/chair_test.py:189-213: torch_dtype=float16): half the attention bytes; K/V are
                                rounded to nearest-even when they enter the cache, attention arithmetic stays fp32 */
  int32_t reserved[3];       /* [0], [1]: tensor-parallel shard — world (2..8; 0 or 1: not sharded) and rank (dd_lm_tp_*): the head
                                counts and intermediate_size above are then the rank's LOCAL ones; [2]: 0 */
} dd_lm_config;

typedef struct dd_lm dd_lm;

/* tensor ids for dd_lm_load_tensor (HF parameter names in comments) */
#define DD_T_EMBED 0      /* model.embed_tokens.weight            [V][d]      */
#define DD_T_ATTN_NORM 1  /* layers.i.input_layernorm.weight      [d]         */
#define DD_T_WQ 2         /* layers.i.self_attn.q_proj.weight     [H*128][d]  */
#define DD_T_WK 3         /* layers.i.self_attn.k_proj.weight     [Hkv*128][d]*/
#define DD_T_WV 4         /* layers.i.self_attn.v_proj.weight                 */
#define DD_T_WO 5         /* layers.i.self_attn.o_proj.weight     [d][H*128]  */
#define DD_T_MLP_NORM 6   /* layers.i.post_attention_layernorm.weight         */
#define DD_T_WGATE 7      /* layers.i.mlp.gate_proj.weight        [d_ff][d]   */
#define DD_T_WUP 8        /* layers.i.mlp.up_proj.weight                      */
#define DD_T_WDOWN 9      /* layers.i.mlp.down_proj.weight        [d][d_ff]   */
#define DD_T_FINAL_NORM 10/* model.norm.weight                                */
#define DD_T_LM_HEAD 11   /* lm_head.weight                       [V][d]      */

int dd_lm_create(const dd_lm_config* cfg, dd_lm** out);
int dd_lm_destroy(dd_lm* h);
/* bytes of device memory the handle holds (weights + KV + scratch) */
size_t dd_lm_device_bytes(const dd_lm* h);

/* Copy one HF-layout 16-bit tensor (row-major, `rows` x `cols`) into the engine and re-tile it
 * for MFMA streaming: bf16 bits for engines of weight_format 0 (and the embedding / norm vectors of fp8 engines), fp16 bits
 * for weight_format 2.  src may be a host or a device pointer (src_on_device).  Synchronous. */
int dd_lm_load_tensor(dd_lm* h, int tensor_id, int layer, const uint16_t* src_bf16, int rows, int cols,
                      int src_on_device);
/* fp8 engines (weight_format 1): one matrix as OCP e4m3fn bytes [rows][cols] plus row_scale[rows]; W = scale * q.
 * Quantisation policy is the caller's (dropoutdecoding_amd.lm.quantize_fp8 uses per-row absmax / 448); the kernels
 * expand fp8 -> bf16 exactly in registers and apply the scale in the epilogue.  Embedding and norm vectors still go
 * through dd_lm_load_tensor. */
int dd_lm_load_tensor_fp8(dd_lm* h, int tensor_id, int layer, const uint8_t* q_e4m3, const float* row_scale, int rows,
                          int cols, int src_on_device);

/* Fill every weight with a deterministic pseudo-random bf16 pattern of the given scale
 * (synthetic-weights benchmark mode; no host traffic). */
int dd_lm_load_synthetic(dd_lm* h, uint32_t seed, float std);

/* Prefill: full causal pass over T0 input embeddings (fp32, row-major [T0][d], device),
 * lm_head over the visual span + last position, top-k ids, uncertainty scorer.
 * Replaces the prefill branch of forward() (models/llava.py:285-314).  Resets the sequence. */
int dd_lm_prefill(dd_lm* h, const float* embeds_dev, int T0, int span_start, int span_len, void* stream);

/* dd_lm_prefill for n (<= 32) sequences that share one set of weights (dd_lm_create_shared), in one pass over the weights:
 * the prompts run through the layers as one matrix (rows of sequence i: embeds_dev[i], [T0[i]][d] fp32 on the device).
 * Every sequence ends up exactly as dd_lm_prefill would leave it (bit-identical logits, scores, first token, cache).
 * The reference has no counterpart (it prefills one image per process: models/llava.py:285-314); this is the prefill side
 * of the throughput mode (dd_lm_group_step). */
int dd_lm_prefill_group(dd_lm* const* lanes, int n, const float* const* embeds_dev, const int32_t* T0, const int32_t* span_start,
                        const int32_t* span_len, void* stream);

/* Prefix reuse (several prompts over one image, e.g. the 6 POPE questions per image, pope_test/pope_test.py:215-241).
 * dd_lm_truncate cuts the sequence back to its first T_keep positions (>= end of the visual span; the image-derived
 * uncertainty / top-k stay valid: attention is causal); dd_lm_prefill_extend appends n more prompt positions (fp32
 * embeddings [n][d], device) with a chunked prefill against the cache and emits the greedy first token like
 * dd_lm_prefill.  prefill(P) + extend(tail) == prefill(P ‖ tail) row for row. */
int dd_lm_truncate(dd_lm* h, int T_keep, void* stream);
int dd_lm_prefill_extend(dd_lm* h, const float* embeds_dev, int n, void* stream);

/* Prefill with the ensemble ALSO applied to the first generated token: the reference's `# if True:` toggle at
 * models/llava.py:336-337 (with it, the first forward runs llava.py:342-359 on the whole prompt: every member starts
 * from the empty cache and masks its columns for all query rows).  Same arguments as dd_lm_prefill plus the step's
 * (mprobs, K, rng | uniforms).  K == 0 is dd_lm_prefill.  Requires span_start >= 1.  Afterwards dd_lm_get(DD_GET_DROP,
 * ..N_DROP, ..MEMBER_ARGMAX, ..WINNER) describe this first step. */
int dd_lm_prefill_ensemble(dd_lm* h, const float* embeds_dev, int T0, int span_start, int span_len, const double* mprobs,
                           int K, dd_rng* rng, const float* uniforms_dev, void* stream);

/* Single-sequence steps with 1 <= K <= 8 run SPECULATIVELY (policy: dd_lm_set_speculation): the K members share ONE
 * sweep over the weights with the un-masked pass, their masks sampled for an empty keep set from the same draws; when the
 * real keep set (models/llava.py:603, 660) would have restored a token some member dropped, the masks are re-sampled from
 * the saved rng state and the members re-run.  Every result (tokens, masks, logits, KV rows, rng stream) is that of the
 * two-sweep step, bit for bit; DD_GET_SPEC_OK tells which way the last step went. */
/* One ensemble decode step, enqueued on `stream` without host synchronisation:
 * embed(last token) -> un-masked pass -> argmax -> keep set -> K masks -> K masked members in
 * one packed sweep -> vote -> winner's logits/argmax + KV row committed.
 * mprobs_host[K] = settings['voting_numbers'] read at this step (models/llava.py:340).
 * K = 0 runs the stock greedy step (`--original`).  rng may be NULL for IBLIP/injected modes;
 * uniforms_dev (optional, [K][L]) overrides the rng for this step (parity tests). */
int dd_lm_decode_step(dd_lm* h, const double* mprobs_host, int K, dd_rng* rng, const float* uniforms_dev,
                      void* stream);

/* dd_lm_decode_step for ONE sequence with the fallback of the speculative step decided by the host (the reference's
 * loop, models/llava.py:326-359, one image at a time): the combined sweep (un-masked row + K members under masks drawn
 * for an empty keep set) and the check are launched, the calling thread waits for the check's verdict in pinned memory
 * (the only wait of the step, microseconds after the sweep's last kernel), and the members' re-run with the real keep
 * set is launched only when the verdict asks for it — dd_lm_decode_step enqueues that re-run unconditionally (its ~350
 * kernels return at once when not needed: about 0.3 ms of a 4 ms step).  Every result equals dd_lm_decode_step's.
 * *held (may be NULL): 1 the speculative members stood, 0 they were re-run, -1 the call went through dd_lm_decode_step
 * (K = 0 or > 8, speculation switched off).  The steps of one sequence must not be mixed between the two calls while
 * earlier ones are still in flight on different streams. */
int dd_lm_decode_step_sync(dd_lm* h, const double* mprobs_host, int K, dd_rng* rng, void* stream, int* held);

/* When to speculate (see above; results never depend on it).  mode: 0 never (always the un-masked sweep, then the member
 * sweep — the reference's own order, models/llava.py:294-359), 1 always, 2 adaptive, -1 (default) the process default
 * (dd_set_tuning key 14, itself 2 by default).  Adaptive: dd_lm_decode_step_sync sees every check's verdict; while the
 * running share of speculative steps that held is below the break-even (about one in four: a failed speculation costs a
 * 16-row sweep plus the 8-row re-run, more than the plain step's 1-row + 8-row sweeps) it issues plain two-sweep steps and
 * re-probes every 32 steps.  The share depends on the checkpoint: the speculation holds when the step's keep set
 * (models/llava.py:443-482) is empty or untouched by every member's drops.  dd_lm_decode_step (queued, the host never
 * learns the verdicts) speculates under modes 1 and 2 alike.
 * dd_lm_spec_stats: out4 = {speculative steps that held, that were re-run, plain two-sweep steps issued by the adaptive
 * policy, times the policy switched speculation off} since creation or the last reset. */
int dd_lm_set_speculation(dd_lm* h, int mode);
int dd_lm_spec_stats(dd_lm* h, int64_t* out4, int reset);

/* The same step in phases, for sharding the K members over ranks (nothing in the reference to
 * mirror: its K members run sequentially in one process, models/llava.py:342-359):
 *   dd_lm_step_base     un-masked pass + keep set + masks for ALL K members (every rank, identical, no comm);
 *   dd_lm_step_members  members [m_lo, m_hi) in one packed sweep (m_lo..m_hi within one group of 8);
 *   dd_lm_xchg_export_ids    ids_dev[2*K] int32 <- {token argmax, vote id} of the members this rank ran, 0 elsewhere;
 *                            the caller all-reduces (sum) ids_dev across ranks;
 *   dd_lm_xchg_import_ids    all K ids back into the engine;
 *   dd_lm_xchg_export_winner votes on the device, then rec_dev[dd_lm_xchg_stride] fp32 <- the winner's
 *                            {logits[V_pad], new KV rows [layers][2][kv_dim]} if this rank ran it, zeros otherwise;
 *                            the caller all-reduces (sum) rec_dev: a broadcast from a data-dependent root with no
 *                            host round trip;
 *   dd_lm_xchg_import_winner the record into the winner's slots;
 *   dd_lm_step_commit   vote (same result on every rank), append the winner's KV row, emit the token. */
int dd_lm_step_base(dd_lm* h, const double* mprobs_host, int K, dd_rng* rng, const float* uniforms_dev, void* stream);
int dd_lm_step_members(dd_lm* h, int m_lo, int m_hi, void* stream);
int dd_lm_step_commit(dd_lm* h, int K, void* stream);
size_t dd_lm_xchg_stride(const dd_lm* h);   /* floats per winner record */
int dd_lm_xchg_export_ids(dd_lm* h, int m_lo, int m_hi, int32_t* ids_dev, void* stream);
int dd_lm_xchg_import_ids(dd_lm* h, const int32_t* ids_dev, void* stream);
int dd_lm_xchg_export_winner(dd_lm* h, int m_lo, int m_hi, float* rec_dev, void* stream);
int dd_lm_xchg_import_winner(dd_lm* h, const float* rec_dev, void* stream);

/* Read-backs (synchronise `stream` first). what: */
#define DD_GET_TOKENS 0        /* int32 [n_generated]   tokens emitted so far (incl. the prefill's greedy token) */
#define DD_GET_LOGITS 1        /* fp32  [V]             logits returned by the last forward                      */
#define DD_GET_EPI 2           /* fp32  [L]             epis_uncert_per_token                                     */
#define DD_GET_ALEA 3          /* fp32  [L]                                                                        */
#define DD_GET_VAR 4           /* fp32  [L]                                                                        */
#define DD_GET_UNCERT_SCALARS 5/* fp32  [3]             variance, epis_uncert, alea_uncert                         */
#define DD_GET_TOPK_IDS 6      /* int32 [L][k_top]                                                                 */
#define DD_GET_TOPK_VALS 7     /* fp32  [L][k_top]                                                                 */
#define DD_GET_DROP 8          /* u8    [K][L]          last step's drop flags                                     */
#define DD_GET_N_DROP 9        /* int32 [K]             last step's masked_numbers                                 */
#define DD_GET_MEMBER_ARGMAX 10/* int32 [K]                                                                        */
#define DD_GET_WINNER 11       /* int32 [2]             {winner index, voted id}                                   */
#define DD_GET_BASE_LOGITS 12  /* fp32  [V]             un-masked pass logits of the last step                     */
#define DD_GET_IMAGE_LOGITS 13 /* fp32  [L][V]          prefill logits over the visual span                        */
#define DD_GET_KEEP 14         /* u8    [L]                                                                        */
#define DD_GET_KV_SUMS 15      /* fp64  [layers][2]     sum of K and of V entries over the committed cache         */
#define DD_GET_SEQ_LEN 16      /* int32 [1]             committed KV length                                        */
#define DD_GET_HIDDEN 17       /* fp32  [d]             final-normed hidden state behind DD_GET_LOGITS             */
#define DD_GET_SPEC_OK 18      /* int32 [1]             last single-sequence step: 1 = the members' masks sampled for an empty keep set
                                                        stood (one sweep), 0 = they were re-sampled and the members re-run       */
int dd_lm_get(dd_lm* h, int what, void* dst_host, size_t bytes, void* stream);

/* Non-blocking: copy the tokens emitted so far (mirrored by the step kernels into pinned host memory) to dst and return
 * how many; does not synchronise any stream. */
int dd_lm_peek_tokens(dd_lm* h, int32_t* dst_host, int max_tokens);

/* End of sequence, device-side.  HF's greedy loop (third-party GenerationMixin._sample, the caller of the reference's
 * forward(); chair_test/chair_test.py:341-346) stops at the first EOS id, and the reference's global rng stream then
 * continues into the next image (models/llava.py:16-20, :650).  Decode steps are enqueued here without host
 * synchronisation, possibly several beyond the step that will emit the EOS, so the stop has to live on the device: the
 * step that emits one of these ids marks the sequence finished, and every later enqueued step of that sequence is a no-op
 * for all persistent state — it draws nothing from the rng, leaves masks / keep set / vote / logits / tokens / KV length as
 * the EOS step left them — until the next dd_lm_prefill / dd_lm_truncate.  eos_ids_host: up to 8 ids (host memory);
 * n = 0 clears the list (never stop).  The list survives prefills.  dd_lm_get re-reads the committed length, so host-side
 * bookkeeping of enqueued steps never leaks into results. */
int dd_lm_set_eos(dd_lm* h, const int32_t* eos_ids_host, int n, void* stream);

/* Force the next decode step's input token (default: the last emitted token). */
int dd_lm_set_next_token(dd_lm* h, int32_t token, void* stream);

/* Roofline bookkeeping: algorithmic HBM bytes of one decode step at the current length
 * (2 * W_lm + 2 * T * kv_tok for dropout steps) */
double dd_lm_step_algorithmic_bytes(const dd_lm* h, int K);

/* ------------------------------------------------------------------------------------------
 * Vision front-end: CLIP-style ViT tower (+ optional LLaVA 2-layer projector) on own kernels.
 * Replaces what the reference runs through third-party modules at models/llava.py:233-246
 * (vision_tower(...).hidden_states[vision_feature_layer][:, 1:] -> multi_modal_projector) and, per tile,
 * models/llavanext.py:409-417.  bf16 weights, fp32 activations (hi/lo bf16 planes through the MFMA GEMM).
 * num_layers = how many encoder layers to RUN: vision_feature_layer = -2 on a 24-layer tower -> 23.
 * ------------------------------------------------------------------------------------------ */
typedef struct dd_vit_config {
  int32_t image_size, patch_size;     /* 336, 14 */
  int32_t hidden_size;                /* 1024 (CLIP ViT-L) or 1408 (EVA ViT-g): multiple of 64; head_dim 64 or 88 */
  int32_t intermediate_size;          /* 4096 */
  int32_t num_layers;                 /* encoder layers to run */
  int32_t num_heads;                  /* 16 */
  int32_t proj_dim;                   /* 0 = return raw features [P][hidden]; else LLaVA projector output width */
  int32_t act;                        /* MLP activation: 0 quick_gelu (CLIP), 1 gelu(erf) */
  float ln_eps;                       /* 1e-5 (CLIP), 1e-6 (EVA) */
  int32_t flags;                      /* DD_VIT_*: 0 = CLIP (pre-LayerNorm, class token dropped, no post-LayerNorm) */
  int32_t reserved[6];
} dd_vit_config;
#define DD_VIT_NO_PRE_LN 1    /* the tower has no pre-LayerNorm (EVA ViT-g of InstructBLIP) */
#define DD_VIT_POST_LN 2      /* apply post_layernorm to the returned tokens (InstructBlipVisionModel.last_hidden_state) */
#define DD_VIT_KEEP_CLASS 4   /* return all P + 1 tokens, class token first (the Q-Former cross-attends to all 257) */
typedef struct dd_vit dd_vit;

/* tensor ids for dd_vit_load_tensor (HF CLIPVisionModel / LlavaMultiModalProjector names) */
#define DD_VT_PATCH 0      /* embeddings.patch_embedding.weight [hidden][3*p*p] flattened, zero-padded to a multiple of 64 columns */
#define DD_VT_CLASS 1      /* embeddings.class_embedding [hidden]                      */
#define DD_VT_POS 2        /* embeddings.position_embedding.weight [P+1][hidden]        */
#define DD_VT_PRE_LN_W 3   /* pre_layrnorm.weight */
#define DD_VT_PRE_LN_B 4
#define DD_VT_LN1_W 5      /* encoder.layers.i.layer_norm1.weight */
#define DD_VT_LN1_B 6
#define DD_VT_WQ 7         /* self_attn.q_proj.weight [hidden][hidden] */
#define DD_VT_WK 8
#define DD_VT_WV 9
#define DD_VT_BQ 10        /* self_attn.q_proj.bias */
#define DD_VT_BK 11
#define DD_VT_BV 12
#define DD_VT_WO 13        /* self_attn.out_proj.weight */
#define DD_VT_BO 14
#define DD_VT_LN2_W 15
#define DD_VT_LN2_B 16
#define DD_VT_FC1_W 17     /* mlp.fc1.weight [intermediate][hidden] */
#define DD_VT_FC1_B 18
#define DD_VT_FC2_W 19     /* mlp.fc2.weight [hidden][intermediate] */
#define DD_VT_FC2_B 20
#define DD_VT_PROJ1_W 21   /* multi_modal_projector.linear_1.weight [proj][hidden] */
#define DD_VT_PROJ1_B 22
#define DD_VT_PROJ2_W 23   /* multi_modal_projector.linear_2.weight [proj][proj]   */
#define DD_VT_PROJ2_B 24
#define DD_VT_PATCH_B 25   /* embeddings.patch_embedding.bias [hidden] (EVA; CLIP's patch conv has none) */
#define DD_VT_POST_LN_W 26 /* post_layernorm.weight */
#define DD_VT_POST_LN_B 27

int dd_vit_create(const dd_vit_config* cfg, dd_vit** out);
int dd_vit_destroy(dd_vit* h);
int dd_vit_load_tensor(dd_vit* h, int tensor_id, int layer, const uint16_t* src_bf16, int rows, int cols, int src_on_device);
/* pixels_dev [n_images][3][H][W] fp32 (already normalised) -> out_dev [n_images][P][proj_dim or hidden] fp32
 * ([n_images][P + 1][hidden] with DD_VIT_KEEP_CLASS).  Several images per call run through the tower as one matrix, 16 at a
 * time (each image's tokens padded to whole 128-row blocks, one attention launch over the images): the same values as one call
 * per image, at 2.0 instead of 5.3 ms per image for CLIP-ViT-L/14-336 + projector. */
int dd_vit_forward(dd_vit* h, const float* pixels_dev, int n_images, float* out_dev, void* stream);

/* ---- InstructBLIP Q-Former + language projection on own kernels ---------------------------------------------------------
 * Replaces the third-party modules reference models/instructblip.py:613-633 calls (`self.qformer(input_ids=qformer_input_ids,
 * attention_mask=..., query_embeds=query_tokens, encoder_hidden_states=image_embeds, ...)` then
 * `self.language_projection(query_output[:, :Q])`): a BERT-style post-LayerNorm encoder over [query tokens ; instruction
 * tokens] whose query rows cross-attend to the vision tower's output every `cross_attention_frequency` layers and whose
 * query / instruction rows run separate feed-forward blocks.  One sequence per call (the reference's batch is 1); padded
 * instruction tokens are dropped by the caller (masked keys contribute nothing, so the query rows are unchanged). */
typedef struct dd_qformer_config {
  int hidden_size;           /* 768 */
  int num_heads;             /* 12 (head_dim must be 64) */
  int num_layers;            /* 12 */
  int intermediate_size;     /* 3072 */
  int encoder_hidden_size;   /* 1408 (EVA ViT-g) */
  int cross_attention_frequency; /* 2: layers 0, 2, 4, ... cross-attend */
  int num_query_tokens;      /* 32 */
  int vocab_size;            /* 30523 */
  int max_position_embeddings; /* 512 */
  int proj_dim;              /* language_projection width: 4096 */
  int max_text_tokens;       /* capacity for instruction tokens per call */
  int max_encoder_tokens;    /* capacity for vision tokens per call (257) */
  float ln_eps;              /* 1e-12 */
} dd_qformer_config;
typedef struct dd_qformer dd_qformer;
/* tensor ids for dd_qformer_load_tensor (HF InstructBlipQFormerModel parameter names) */
#define DD_QF_WORD_EMB 0   /* embeddings.word_embeddings.weight [vocab][d] */
#define DD_QF_POS_EMB 1    /* embeddings.position_embeddings.weight [max_pos][d] */
#define DD_QF_EMB_LN_W 2   /* embeddings.layernorm */
#define DD_QF_EMB_LN_B 3
#define DD_QF_QUERY_TOKENS 4 /* query_tokens [Q][d] */
#define DD_QF_PROJ_W 5     /* language_projection.weight [proj][d] */
#define DD_QF_PROJ_B 6
/* per layer (encoder.layer.N.) */
#define DD_QF_SA_WQ 10     /* attention.attention.query / key / value */
#define DD_QF_SA_WK 11
#define DD_QF_SA_WV 12
#define DD_QF_SA_BQ 13
#define DD_QF_SA_BK 14
#define DD_QF_SA_BV 15
#define DD_QF_SA_WO 16     /* attention.output.dense */
#define DD_QF_SA_BO 17
#define DD_QF_SA_LN_W 18   /* attention.output.LayerNorm */
#define DD_QF_SA_LN_B 19
#define DD_QF_CA_WQ 20     /* crossattention.attention.query [d][d]; key / value [d][encoder_hidden] (cross layers only) */
#define DD_QF_CA_BQ 21
#define DD_QF_CA_WK 22
#define DD_QF_CA_BK 23
#define DD_QF_CA_WV 24
#define DD_QF_CA_BV 25
#define DD_QF_CA_WO 26     /* crossattention.output.dense */
#define DD_QF_CA_BO 27
#define DD_QF_CA_LN_W 28
#define DD_QF_CA_LN_B 29
#define DD_QF_FFQ_W1 30    /* intermediate_query.dense / output_query.dense + LayerNorm (query rows) */
#define DD_QF_FFQ_B1 31
#define DD_QF_FFQ_W2 32
#define DD_QF_FFQ_B2 33
#define DD_QF_FFQ_LN_W 34
#define DD_QF_FFQ_LN_B 35
#define DD_QF_FFT_W1 36    /* intermediate.dense / output.dense + LayerNorm (instruction rows) */
#define DD_QF_FFT_B1 37
#define DD_QF_FFT_W2 38
#define DD_QF_FFT_B2 39
#define DD_QF_FFT_LN_W 40
#define DD_QF_FFT_LN_B 41

int dd_qformer_create(const dd_qformer_config* cfg, dd_qformer** out);
int dd_qformer_destroy(dd_qformer* h);
int dd_qformer_load_tensor(dd_qformer* h, int tensor_id, int layer, const uint16_t* src_bf16, int rows, int cols, int src_on_device);
/* text_ids_dev [n_text] int32 instruction token ids (n_text may be 0); enc_dev [n_enc][encoder_hidden] fp32 vision tokens ->
 * out_dev [num_query_tokens][proj_dim] fp32: the Q visual embeddings the LM sees at positions 0..Q-1.
 * hidden_out_dev (optional) receives the Q-Former's last hidden state rows [num_query_tokens + n_text][hidden]. */
int dd_qformer_forward(dd_qformer* h, const int32_t* text_ids_dev, int n_text, const float* enc_dev, int n_enc, float* out_dev,
                       float* hidden_out_dev, void* stream);

/* Switches of the library (process-wide; every setting produces the same results):
 *   8  replay decode steps from hipGraphs (default 1; 0: launch every kernel of a step),
 *   11 short prompt chunks (dd_lm_prefill_extend, <= 32 rows) through the decode GEMVs (default 1; 0: the prefill GEMMs),
 *   13 slice-resident 16 / 32 / 64-row GEMVs (default 1; 0: the K-split-over-waves kernels, same bits),
 *   14 process default of the speculation policy of single-sequence steps (dd_lm_set_speculation: 0 never, 1 always,
 *      2 adaptive = default),
 *   15 XCD-aware block order of the prefill GEMM (default 1; same bits), 16 rows from which the prefill GEMM uses its
 *      big LDS-staged block (default 1024; 0: never; same bits), 20 form of that block (default 1: 160 x 512, operand
 *      fragments by LDS-DMA; 0: round 3's register-staged 128 x 512 block; same bits).
 * Kernel-variant experiment knobs and the timing hooks of bench.py / tools/ are not part of this library: they live in
 * libdropdec_tools.so (include/dropdec_tools.h). */
int dd_set_tuning(int key, int value);

/* ---- several sequences over one set of weights -------------------------------------------------------------------------
 * The reference decodes one image at a time (batch 1, chair_test.py:270-346) and, for 500 images, shards them over
 * processes; every process has its own torch generator.  A "lane" is that: a further sequence with its own KV cache,
 * state, token mirror and (caller-side) dd_rng, over the weights of an existing handle.  dd_lm_group_step advances n
 * lanes by one token each; results per lane are bit-identical to dd_lm_decode_step on that lane alone, but the n
 * un-masked base passes (models/llava.py:294-303 for each image) run as ONE sweep, so the weights are streamed
 * 1/n + ceil(K/8) times per sequence and token instead of 1 + ceil(K/8) times.
 *   dd_lm_create_shared: same cfg dimensions as the owner; max_seq must equal the owner's for lanes grouped together.
 *   dd_lm_group_step: lanes[0..n), n <= 64, all prefilled; rngs[m] is lane m's stream (may be NULL for InstructBLIP's
 *   deterministic masks or K == 0).  The owner may itself be one of the lanes. */
int dd_lm_create_shared(const dd_lm_config* cfg, dd_lm* weights_from, dd_lm** out);
int dd_lm_group_step(dd_lm* const* lanes, int n, const double* mprobs, int K, dd_rng* const* rngs, void* stream);

/* ---- tensor-parallel decode: ONE sequence over several GPUs ------------------------------------------------------------
 * Nothing in the reference to mirror (it runs the K members sequentially in one process, models/llava.py:342-359; its
 * multi-GPU shape is independent jobs, scripts/run_main_experiments.py:81-86).  Sharding the K members leaves every rank
 * streaming all of W_lm twice per token (dd_lm_step_* / dist.KShardDecoder); sharding the WEIGHTS divides the bytes: q/k/v
 * and gate/up column-parallel (by kv-head group / d_ff slice), o_proj and down_proj row-parallel with one exchange each
 * per layer, everything else replicated.  A rank is a dd_lm created with its LOCAL num_heads / num_kv_heads /
 * intermediate_size (padded with zero rows / columns to a multiple of 256) and cfg.reserved = {world, rank}, loaded with
 * its slices through dd_lm_load_tensor; it only accepts the calls below (+ dd_lm_get, dd_lm_set_eos, dd_lm_peek_tokens, ...).
 *   dd_lm_tp_link          all `world` ranks live in ONE process on one device (tests; what one GPU can show): they share a
 *                          gather buffer [world][rows][d] and the calls below take all of them, in rank order;
 *   dd_lm_tp_set_exchange  one rank per process (one per GPU): gather_dev [world][rows_cap][d] fp32 belongs to the caller;
 *                          at each of the 2 seams per layer the engine writes this rank's slot (slot stride rows * d floats)
 *                          and calls exchange(ctx, rows, stream), which all-gathers the slots in place, ordered on `stream`
 *                          (torch.distributed over RCCL / xGMI: dropoutdecoding_amd/dist.py TensorParallelRank); the calls
 *                          below then take that one handle (n = 1).
 *   dd_lm_tp_prefill       = dd_lm_prefill; dd_lm_tp_decode_step = dd_lm_decode_step (two-sweep form, 0 <= K <= 8; rngs[i]:
 *                          the i-th passed rank's copy of the stream — every rank draws the same masks, no exchange).
 * The slots are added in rank order: results are deterministic for a given world size, and linked and distributed runs of
 * the same world size agree bit for bit; against the un-sharded engine they differ by fp32 reassociation (logits ~1e-6
 * relative), world = 1 not at all. */
int dd_lm_tp_link(dd_lm* const* ranks, int world, int rows_cap);
int dd_lm_tp_set_exchange(dd_lm* h, float* gather_dev, size_t gather_floats, int (*exchange)(void* ctx, int rows, void* stream),
                          void* ctx);
int dd_lm_tp_prefill(dd_lm* const* ranks, int n, const float* embeds_dev, int T0, int span_start, int span_len, void* stream);
int dd_lm_tp_decode_step(dd_lm* const* ranks, int n, const double* mprobs_host, int K, dd_rng* const* rngs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DROPDEC_H */

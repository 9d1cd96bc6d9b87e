// Engine internals shared by dd_engine.hip and the measurement hooks in dd_tools.hip (libdropdec_tools.so only).
#pragma once
#include <vector>

#include "dd_lm_kernels.h"

#define MAX_MEMBERS DD_MAX_MEMBERS
#define GROUP_ROWS 72     // rows of the widest decode pass: the members of eight sequences + eight un-masked rows riding along (nine operand planes)
#define GROUP_PLANES (GROUP_ROWS / 8)
#define GROUP_MAX_LANES 64
#define KV_ROWS 64        // new K/V rows kept per layer: 16 members, or the base rows of up to 64 lanes (group step)
#define MAX_NEW_TOKENS 8192

struct LayerW {
  u32x4_t *wqkv, *wo, *wgu, *wdown;
  float *norm1, *norm2;
  float *s_qkv = nullptr, *s_o = nullptr, *s_gu = nullptr, *s_down = nullptr;   // fp8: per-row scales, packed order
};

struct dd_lm {
  dd_lm_config cfg;
  int d, dff, V, Vpad, H, Hkv, q_dim, kv_dim, Lyr, T_cap, Lmax;
  int S_d, S_q, S_ff, qkv_tiles, q_tiles, k_tiles;
  std::vector<void*> allocs;
  size_t bytes = 0;
  unsigned long long serial = 0;
  dd_lm* wsrc = nullptr;       // lane created by dd_lm_create_shared: weights (and rope tables) belong to this handle
  float* grp_logits = nullptr; // [GROUP_MAX_LANES][Vpad] base-pass logits of a group step (this handle is the group's first lane)
  int32_t* grp_argmax = nullptr;
  // rider (group_step_rider): the un-masked row of this sequence's NEXT step, computed ahead while it rode in another group's member
  // sweep — logits / argmax parked here until the step they belong to starts; valid while pend_step == steps_since_prefill
  float* base_next = nullptr;
  int32_t* argmax_next = nullptr;
  bool pend_valid = false;
  int pend_step = -1;
  DDState* chunk_states = nullptr;   // [32] positions of the rows of a short prompt chunk (dd_lm_prefill_extend)
  float *chunk_k = nullptr, *chunk_v = nullptr;   // [32][kv_dim] roped K / V rows of the chunk, one layer at a time
  const float *commit_k = nullptr, *commit_v = nullptr;   // K == 0 group step: this lane's base row in the leader's scratch
  // weights
  std::vector<LayerW> lw;
  u32x4_t* lm_head = nullptr;
  float* s_lm = nullptr;
  int fp8 = 0;                 // weight storage: 0 bf16, 1 OCP e4m3fn + per-row scales
  int wf = 0;                  // 16-bit weight / operand type: 0 bf16, 1 fp16 (weight_format 2: fp16 checkpoints stay exact)
  int kv16 = 0;                // KV cache storage: 0 fp32, 1 fp16 (the reference's cache width; layouts in dd_lm_kernels.h)
  u32x4_t* deq_tmp = nullptr;  // fp8: bf16 tiles of ONE matrix for the prefill GEMM
  float* final_norm = nullptr;
  uint16_t* embed = nullptr;
  float *rope_cos = nullptr, *rope_sin = nullptr;
  // kv
  float *kc = nullptr, *vc = nullptr;
  size_t lsk = 0, lsv = 0;
  // decode scratch
  float *xa, *qbuf, *knew, *vnew, *ssq_a, *ssq_b, *part_o, *part_ml, *hidden;
  int32_t* spec_ok = nullptr;   // speculative step: 1 = the members of the combined sweep stand, 0 = re-run them (device flag)
  uint32_t* rng_backup = nullptr;   // mt19937 state before the speculative draws (the re-run repeats exactly these)
  float* gemv_part = nullptr;   // partial sums of the slice-resident 16 / 32-row GEMVs (dd_gemv_slices.h)
  size_t gemv_part_floats = 0;
  u32x4_t *xop_d, *xop_q, *xop_ff;
  float *base_logits, *member_logits, *last_logits, *last_hidden;
  int member_rows_cap = 0;    // rows member_logits holds: 16 at creation, DD_MAX_MEMBERS from the first step with K > 16 on (ensure_member_rows)
  int32_t *argmax_base, *member_tok, *member_vote, *tokens;
  uint8_t *keep, *drop, *drop_bits, *leak_bits;
  int32_t* n_drop;
  DDState* state;
  // prefill scratch
  float *px, *pq, *image_logits;
  uint16_t *p1_hi, *p1_lo, *p2_hi, *p2_lo;
  // scratch of dd_lm_prefill_group (weight owner only): n * seq_rows rows of residual, q, and the two operand plane pairs
  SeqTab* seq_tab = nullptr;    // device table of a dd_lm_prefill_group call led by this handle
  size_t pb_rows = 0;
  float *pb_x = nullptr, *pb_q = nullptr;
  uint16_t *pb1_hi = nullptr, *pb1_lo = nullptr, *pb2_hi = nullptr, *pb2_lo = nullptr;
  int32_t* row_index;
  float *epi, *alea, *var, *scalars, *topk_vals, *kl_ws;
  int32_t* topk_ids;
  void* unc_ws;
  size_t unc_ws_bytes;
  double* kv_sums;
  // host mirrors
  int T_host = 0, span_start = 0, L = 0, n_tok_host = 0, last_K = 0;
  bool prefilled = false;
  bool have_leak = false;
  int bit0 = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // hipGraph cache of whole decode steps (key: K, probabilities, key-tile count, rng)
  struct GraphEntry {
    unsigned long long key;
    hipGraphExec_t exec;
  };
  std::vector<GraphEntry> graphs;
  int steps_since_prefill = 0;
  // host-visible token mirror (pinned, device-mapped): the decode loop can watch for EOS without synchronising
  int32_t* tok_host = nullptr;       // host pointer: [0] = count, [1..] = tokens
  int32_t* tok_host_dev = nullptr;   // the same memory as seen from the device
  int spec_seq_host = 0;             // speculation checks announced to the host so far (dd_lm_decode_step_sync)
  // speculation policy of this sequence (dd_lm_set_speculation): -1 the process default, 0 never, 1 always, 2 adaptive —
  // dd_lm_decode_step_sync learns every check's verdict and stops speculating while too few of them hold
  int spec_mode = -1;
  float spec_rate = 1.0f;            // running share of speculative steps that held (weight 1/8 per step)
  int spec_cooldown = 0;             // adaptive: two-sweep steps left before speculation is tried again
  long long spec_n[4] = {0, 0, 0, 0};   // speculative steps that held / were re-run, two-sweep steps, switches to two-sweep
  // further branches of a group step (the group's first lane owns them): the member sweeps are dealt over the caller's stream
  // and these, and run concurrently
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  float *part_o_ride = nullptr, *part_ml_ride = nullptr;   // flash-decoding partials of the riding rows (their attention shares a launch with the members')
  // tensor-parallel shard (dd_tp.hip): this handle holds the q/k/v/gate/up columns and the o/down rows of rank tp_rank of
  // tp_world (its dims above are the LOCAL ones); the row-parallel matrices write partial sums into this rank's slot of the
  // gather buffer [tp_world][rows][d] and every rank adds the slots in rank order
  int tp_world = 1, tp_rank = 0;
  float* tp_gather = nullptr;        // linked ranks of one process share rank 0's buffer; one rank per process: its own
  size_t tp_gather_floats = 0;
  int (*tp_exchange)(void* ctx, int rows, void* stream) = nullptr;   // one rank per process: all-gather of the slots (torch.distributed)
  void* tp_ctx = nullptr;
};

#define KV_ROWS_PER_LAYER KV_ROWS
// pieces of dd_engine.hip that the tensor-parallel driver (dd_tp.hip) reuses
int dd_engine_prefill_head(dd_lm* h, const int32_t* row_index, int n_rows, float* logits, hipStream_t st, const float* src);
int dd_engine_prefill_tail(dd_lm* h, const float* x_rows, int T0, int span_start, int span_len, hipStream_t st);
int dd_engine_step_keep(dd_lm* h, const int32_t* gate, hipStream_t st);
int dd_engine_step_begin(dd_lm* h, hipStream_t st);          // k_step_begin: the step's RoPE position
void dd_engine_bump_epoch();                 // every tuning call: graph keys carry the epoch, a step captured under other settings is not replayed
unsigned long long dd_engine_epoch();
int dd_engine_tp_alloc(dd_lm* h, float** p, size_t floats);  // device memory owned (and released) by the handle
// set by libdropdec_tools.so (dd_tools_trace_*): called at the end of every group_finish, on the stream of that sweep
extern int (*dd_engine_group_finish_hook)(dd_lm* const* qs, int ng, int K, hipStream_t st);
// one packed sweep of nb rows through all layers + lm_head (dd_engine.hip)
int lm_sweep(dd_lm* h, int nb, const uint8_t* bits, int row0, float* logits_out, hipStream_t st, dd_lm* const* lanes = nullptr,
             const int32_t* skip_if = nullptr);

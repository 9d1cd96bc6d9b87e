// Internal launch interface of the LM kernels (dd_lm_kernels.hip) used by dd_engine.hip.
#pragma once
#include "dd_common.h"

// Device-resident sequence state: kernels read lengths/positions from here so that a whole
// decode step is enqueued without the host knowing anything but the (deterministic) length.
struct DDState {
  int32_t T;        // committed KV length == number of cached keys every row attends to
  int32_t pos;      // RoPE position of the token being decoded
  int32_t n_tok;    // tokens emitted so far
  int32_t cur_tok;  // input token of the step being decoded
  int32_t winner;   // last vote: member index
  int32_t voted;    // last vote: id
  // HF's greedy loop stops at EOS; steps are enqueued ahead of the host's knowledge of the tokens, so the
  // stop lives here: the step that emits an EOS id sets `done`, and every later enqueued step of the sequence is a no-op
  // for everything that persists (rng stream, masks, vote, KV append, tokens, lengths) until the next prefill.
  int32_t done;
  int32_t n_eos;
  int32_t eos[8];
};
#define DD_MAX_EOS 8
__host__ __device__ inline bool dd_is_eos(const DDState* st, int tok) {
  bool hit = false;
  for (int i = 0; i < DD_MAX_EOS; ++i) hit |= (i < st->n_eos && st->eos[i] == tok);
  return hit;
}

// ---- KV cache layouts -------------------------------------------------------------------------
// fp32 (kv16 = 0):  K [kv_head][d/4][T_cap][4]  (keys on lanes: a wave's 16-byte loads of one d-chunk are 1 KiB contiguous),
//                   V [kv_head][T_cap][128]
// fp16 (kv16 = 1, the width the reference keeps its cache in — chair_test/chair_test.py:189-213 loads every model with
//                   torch_dtype=float16):  K [kv_head][d/8][T_cap][8] halves (16 bytes per key and chunk),
//                   V [kv_head][T_cap/8][128][8] halves: a 16-byte load = 8 consecutive keys of one d — with K's chunks the two
//                   A operands of v_mfma_f32_16x16x32_f16 as they lie (S^T = K . Q^T: 16 keys x 32 d; O^T = V^T . P^T: 16 d x 32 keys).
// Pointers stay `float*` in the interfaces; layer strides are in floats of the actual storage (half the element count for
// fp16).  Element offsets (in elements of the storage type) of key t, dimension d of kv head `kvh`:
__host__ __device__ inline size_t dd_k32(int kvh, int d, int t, int Tc) { return (((size_t)kvh * 32 + (d >> 2)) * Tc + t) * 4 + (d & 3); }
__host__ __device__ inline size_t dd_v32(int kvh, int d, int t, int Tc) { return ((size_t)kvh * Tc + t) * 128 + d; }
__host__ __device__ inline size_t dd_k16(int kvh, int d, int t, int Tc) { return (((size_t)kvh * 16 + (d >> 3)) * Tc + t) * 8 + (d & 7); }
__host__ __device__ inline size_t dd_v16(int kvh, int d, int t, int Tc) {
  return (((size_t)kvh * (Tc >> 3) + (t >> 3)) * 128 + d) * 8 + (t & 7);
}
typedef _Float16 dd_half;
__device__ __forceinline__ void dd_kv_store(float* kc, float* vc, int kv16, int kvh, int d, int t, int Tc, bool is_k, float v) {
  if (kv16) {
    if (is_k) ((dd_half*)kc)[dd_k16(kvh, d, t, Tc)] = (dd_half)v;      // round to nearest even, as a cast to torch.float16 does
    else ((dd_half*)vc)[dd_v16(kvh, d, t, Tc)] = (dd_half)v;
  } else {
    if (is_k) kc[dd_k32(kvh, d, t, Tc)] = v;
    else vc[dd_v32(kvh, d, t, Tc)] = v;
  }
}
__device__ __forceinline__ float dd_kv_load(const float* kc, const float* vc, int kv16, int kvh, int d, int t, int Tc, bool is_k) {
  if (kv16) return is_k ? (float)((const dd_half*)kc)[dd_k16(kvh, d, t, Tc)] : (float)((const dd_half*)vc)[dd_v16(kvh, d, t, Tc)];
  return is_k ? kc[dd_k32(kvh, d, t, Tc)] : vc[dd_v32(kvh, d, t, Tc)];
}

// ---- weight layout ------------------------------------------------------------------------
// W[N][K] bf16 (HF: out_features x in_features) is stored as 16x32 MFMA operand tiles:
//   tile (nt, ks) = 64 lanes x 16 bytes at ((nt * S + ks) * 64 + lane), S = K/32,
//   lane = (h << 4) | r holds W[row(nt, r)][ks*32 + 8h + 0..7]
// so one wave instruction streams 1 KiB contiguous and feeds v_mfma_f32_16x16x32_bf16 directly.
// row(nt, r) = nt*16 + r except inside rotary heads (PACK_ROPE), where tile tt of a head holds
// rows {8tt..8tt+7} and {64+8tt..64+8tt+7} so that both halves of a rotate_half pair meet in one tile.
#define PACK_PLAIN 0
#define PACK_ROPE 1

int ddk_pack_weight(const uint16_t* src_dev, int rows, int cols, u32x4_t* dst, int dst_tile0, int tile_stride,
                    int pack_mode, int n_src_tiles, hipStream_t st);
int ddk_fill_synthetic(uint16_t* dst_bf16, size_t n, uint32_t seed, float std, hipStream_t st, int wf = 0);   // wf 1: fp16 bits
int ddk_fill_const_f32(float* dst, size_t n, float v, hipStream_t st);
int ddk_bf16_to_f32(const uint16_t* src, float* dst, int n, hipStream_t st, int wf = 0);
int ddk_rope_table(float* cos_t, float* sin_t, int max_seq, const float* inv_freq_dev, hipStream_t st);

// ---- decode (NB <= 8 rows against one weight sweep) ------------------------------------------
// The x operand of every decode GEMV arrives PACKED from its producer: [S][64 lanes][8 bf16] where
// lane (h<<4)|c holds, for k = ks*32 + 8h + 0..7, the hi bf16 part of row c (c < 8) or the lo part of
// row c-8 (c >= 8) of a fp32 activation: x = hi + lo carries ~16 mantissa bits through the bf16 MFMA.
// RMSNorm is folded: the producer packs z = norm_weight * x, the consumer multiplies its outputs by
// rstd(row) = 1/sqrt(mean(x^2)+eps), assembled from the producer's per-workgroup sums of squares.
#define EPI_STORE 0   // out[m][n] = y
#define EPI_RESID 1   // x[m][n] += y; packs z = normw_next * x for the next GEMV; emits sum-of-squares slots
#define EPI_SILU 2    // tile pair (gate, up): xop_next <- split(silu(g) * u)
#define EPI_QKV 3     // rotary q/k + v scattered to qbuf / new-row KV scratch
#define EPI_ACT 5     // (prefill GEMM only) y = act(y + bias) -> packed hi/lo planes; act 0 quick_gelu, 1 gelu(erf), 2 none
#define EPI_QKV_VIT 6 // (prefill GEMM only) ViT q/k/v with bias: q*scale -> qbuf, k -> K^T tiles, v -> rows

// FP8 weight storage (OCP e4m3fn, per-output-row fp32 scales): tile (nt, ks2) = 64 lanes x 16 bytes at
// ((nt*S2 + ks2)*64 + lane), S2 = K/64; lane (h<<4)|r holds the 8 fp8 of W[row][ks2*64 + 8h..] followed by the 8 fp8
// of W[row][ks2*64 + 32 + 8h..]: one 1 KiB wave load feeds TWO bf16 MFMA k-steps after an exact in-register
// fp8 -> bf16 conversion (v_cvt_pk_f32_fp8 + v_perm).  Scales are stored in packed-row order and applied in the
// epilogue: y[n] = scale[n] * sum_k q[n][k] x[k].
int ddk_pack_weight_fp8(const uint8_t* src_dev, const float* row_scale_dev, int rows, int cols, u32x4_t* dst,
                        float* dst_scale, int dst_tile0, int tile_stride, int pack_mode, int n_src_tiles, hipStream_t st);
int ddk_dequant_tiles(const u32x4_t* src_fp8, u32x4_t* dst_bf16, int n_tiles, int S, hipStream_t st);
int ddk_fill_synthetic_fp8(uint8_t* dst, size_t n, uint32_t seed, hipStream_t st);

struct GemvArgs {
  const u32x4_t* W;
  int fp8;              // 1: W holds fp8 tiles (S2 = S/2 per tile row), wscale required
  const float* wscale;  // [tiles*16] per-row scales in packed-row order (fp8) or nullptr
  int S;                // K / 32
  int n_tiles;          // number of 16-row tiles (EPI_SILU: number of gate/up PAIRS)
  int nb;               // live rows (1..8)
  const u32x4_t* xop;   // [S][64]
  const float* ssq_in;  // [8][ssq_ld] partial sums of squares of the un-normalised input (ssq_n live per row), or nullptr
  int ssq_n;
  int ssq_ld;           // row pitch of ssq_in / ssq_out (a multiple of 4: rows are read with 16-byte loads)
  float inv_k;          // 1 / K
  float eps;
  // epilogue
  float* out;           // EPI_STORE: [nb][ldo]; EPI_RESID: x [8][N]
  int ldo;
  int n_valid;          // EPI_STORE: columns >= n_valid are not written
  const float* normw_next;  // EPI_RESID: [N]
  u32x4_t* xop_next;    // EPI_RESID / EPI_SILU: packed operand for the next GEMV
  float* ssq_out;       // EPI_RESID: [8][ssq_ld], column blockIdx.x
  // EPI_QKV
  float* qbuf;          // [8][q_dim]
  float* knew;          // [8][kv_dim] rows of this pass, this layer
  float* vnew;
  int q_tiles, k_tiles; // tile counts of the q and k blocks
  int q_dim, kv_dim;
  const float* rope_cos; // [max_seq][64]
  const float* rope_sin;
  const DDState* state;
  const DDState* state_rows[72];  // lanes: row m takes its position from state_rows[m] (null entries: `state`)
  // multi-group passes (ddk_gemv_groups): rows 8g..8g+7 are group g (2, 4, 8 or 9 groups) — group g's operand plane is plane g
  // of `xop` (and of xop_next: plane stride S_next * 64 u32x4); its logits / new K/V rows may live in another sequence's
  // buffers
  int n_groups;
  int nb_rider;         // n_groups == 9: live rows of plane 8 (the riding un-masked rows: one per sequence, not one per member); 0: nb
  // half_planes = H > 0 (K <= 4): the first H planes carry the members of TWO sequences each — rows 0..3 sequence 2 g, rows 4..7 sequence
  // 2 g + 1, nb (<= 4) live rows each; H = 8: all planes of a 64-row pass (sixteen sequences), H = 7 with n_groups == 9: fourteen
  // sequences + TWO riding planes (7 and 8: nb_rider un-masked rows, one per sequence).  out_g / knew_g / vnew_g are indexed by slot:
  // sequence s for the half planes, 2 H + (plane - H) for the others; rows by member (riding planes: by row)
  int half_planes;
  __host__ __device__ bool row_live(int plane, int ml) const {
    if (plane < half_planes) return (ml & 3) < nb;
    const int ride0 = half_planes ? half_planes : 8;                  // first riding plane
    if (nb_rider && plane >= ride0) return ml < nb_rider - 8 * (plane - ride0);
    return ml < nb;
  }
  __host__ __device__ int slot(int plane, int ml) const { return plane < half_planes ? 2 * plane + (ml >> 2) : 2 * half_planes + (plane - half_planes); }
  __host__ __device__ int slot_row(int plane, int ml) const { return plane < half_planes ? (ml & 3) : ml; }
  float* out_g[16];     // EPI_STORE rows of group g (null: out + 8 g * ldo)
  float* knew_g[16];    // EPI_QKV rows of group g (null: knew + 8 g * kv_dim)
  float* vnew_g[16];
  int S_next;           // K / 32 of the GEMV that consumes xop_next
  int wf;                   // 16-bit type of W and of the packed operands: 0 bf16, 1 fp16 (engines created with weight_format 2)
  const int32_t* skip_if;   // optional (k_gemv): *skip_if != 0 -> the launch returns at once (fallback sweep of a speculative step)
  float* part;          // scratch for the slice-resident path (dd_gemv_slices.h): partial sums, or nullptr (then k_gemv_groups runs)
  size_t part_floats;   // capacity; 72 more floats behind it hold rstd of the operand rows
};
int ddk_gemv(int epi, const GemvArgs& a, hipStream_t st);
int ddk_gemv_groups(int epi, const GemvArgs& a, hipStream_t st);   // the same for a.n_groups (2, 4, 8 or 9) groups of up to 8 rows
// tensor-parallel seams (dd_tp.hip): slots of `gather` [W][...] added in rank order
int ddk_tp_finish(const float* gather, int W, size_t slot_floats, int nb, float* x, int N, const float* normw, u32x4_t* xop_next,
                  float* ssq_out, int ssq_ld, int wf, hipStream_t st);      // decode rows: + k_gemv's EPI_RESID epilogue
int ddk_tp_add_rows(float* x, const float* gather, int W, size_t n, hipStream_t st);   // prefill rows: x += sum of the slots
void ddk_set_tuning(int key, int value);
void ddk_set_gemv_slices(int on);
void ddk_set_gemm_big_rows(int rows);  // dd_set_tuning key 16: rows from which the prefill GEMM uses the 128 x 512 LDS-staged block (0: never; same bits)
void ddk_set_gemm_dma(int on);         // dd_set_tuning key 20: LDS-DMA 160 x 512 block of the prefill GEMM (default on; 0: the register-staged 128 x 512 block; same bits)
void ddk_set_gemm_xcd_order(int on);   // dd_set_tuning key 15: XCD-aware block order of the prefill GEMM (default on; same bits)
void ddk_set_slices_only(int on);
void ddk_set_attn_split(int v);
void ddk_set_prefill_mfma(int on);
int ddk_prefill_mfma_enabled();
int ddk_attn_vit_mfma(const float* q, const float* kt, const float* v, int T, int Tc, int n_heads, uint16_t* o_hi, uint16_t* o_lo,
                      hipStream_t st, int head_pitch = 64, int Tk = 0, float scaling = 1.0f);
// bidirectional attention on the matrix cores: head pitch 64 or 96; Tk > 0: T queries against Tk keys (cross-attention);
// scaling multiplies the scores (1 when q arrives pre-scaled)

struct AttnDecodeArgs {
  const float* qbuf;     // [8][q_dim] roped
  const float* kc;       // this layer: [n_kv][32][T_cap][4]
  const float* vc;       // this layer: [n_kv][T_cap][128]
  int T_cap;
  int kv16;              // 1: the cache holds fp16 (layouts above); the rows' own new K/V (knew / vnew) are fp32 either way
  int T;                 // host copy of the prefix length (grid sizing; also the length when `state` is null)
  const DDState* state;  // when set, kernels read the prefix length from the device (graph replays keep advancing)
  int nb, n_heads, n_kv;
  const uint8_t* drop_bits;  // [span_len] bit (bit0 + m) = row m drops that visual token, or nullptr
  int bit0;
  int span_start, span_len;
  float* part_o;         // [n_kv][splits][R][128]
  float* part_ml;        // [n_kv][splits][R][2]
  // combine
  const float* knew;     // [8][kv_dim] roped new keys of this layer (rows of this pass)
  const float* vnew;
  u32x4_t* xop_out;      // packed hi/lo operand for o_proj, [q_dim/32][64]
  int wf;                  // 16-bit type of the packed operand written for o_proj (0 bf16, 1 fp16)
  const int32_t* skip_if;  // optional: *skip_if != 0 -> both kernels return at once (fallback sweep of a speculative step)
  // lanes (n_lanes > 0): row m of the pass belongs to sequence m — its own cache, length, span and (un-shifted) bits.
  // Used by the fused base pass of a group of sequences; kc/vc/state/drop_bits/span_* above are ignored then.
  int n_lanes;
  int lane_groups;       // 0: lane m = row m (fused base pass of up to 16 sequences).  2 / 4 / 8: a 16- / 32- / 64-row pass of that many sequences — rows
                         // 8g..8g+7 are members of lane g (each group reads its own cache with its own drop bits, bit = row & 7)
  const float* knew_g[16];  // lane_groups > 0: new K/V rows of group g (half_planes: of sequence s, rows = members)
  const float* vnew_g[16];
  int half_planes;          // lane_groups == 8, K <= 4: plane g carries sequences 2 g (rows 0..3) and 2 g + 1 (rows 4..7); the lane_* arrays, knew_g
                            // and vnew_g are indexed by sequence (16), a member's drop bit is its index within its sequence
  uint32_t* dbg;         // debug (fp32-cache kernel only; libdropdec_tools.so sets it): per-workgroup checksums of what the tile pass loaded and
                         // exchanged through LDS: [wg][8] = K registers, V registers, q rows, scores read, p written, p read, outputs read, 0
  int max_T;             // host: largest prefix length among the lanes (grid sizing)
  int splits_stride, tiles_per_wg;   // set by the launchers of k_attn_partial16: tile stride of the partial buffers, key tiles per workgroup
  const float* lane_kc[16];
  const float* lane_vc[16];
  const DDState* lane_state[16];
  const uint8_t* lane_bits[16];
  int lane_span_start[16], lane_span_len[16];
};
int ddk_attn_decode(const AttnDecodeArgs& a, hipStream_t st);
int ddk_attn_decode_ride(const AttnDecodeArgs& a, const AttnDecodeArgs& u, int planes_m, hipStream_t st);   // member pass (planes_m planes) + riding rows, one launch
int ddk_attn_grid_tiles(int T, int T_cap);   // tiles the decode attention is launched with for a prefix of T keys

// ---- prefill (M rows) -------------------------------------------------------------------------
int ddk_rmsnorm_split(const float* x, int M, int d, const float* w, float eps, uint16_t* hi, uint16_t* lo,
                      const int32_t* row_index, float* normed_out, hipStream_t st, int wf = 0);

struct GemmArgs {
  const uint16_t* a_hi;  // [M][K] bf16
  const uint16_t* a_lo;
  int M, S;
  const u32x4_t* W;
  const float* wscale;   // per-column (output row of W) scales in packed-row order, or nullptr
  const float* bias;     // per-column bias (natural order) added after the scale, or nullptr
  int act;               // EPI_ACT: 0 quick_gelu, 1 gelu(erf), 2 identity
  int vit_hidden, vit_head_dim, vit_head_pad;   // EPI_QKV_VIT (head_pad: pitch of a head in q / K^T / V; 0 = head_dim)
  int grid_y, xcd_order; // set by the launcher: row blocks of the grid; XCD-aware block order on / off
  // EPI_QKV_VIT over the tokens of several images back to back (dd_vit_forward): vit_img_rows > 0 = rows per image (a multiple of
  // 128); row r is token r % vit_img_rows of image r / vit_img_rows, live below vit_T, and writes that image's K^T / V block
  int vit_img_rows, vit_T;
  size_t vit_k_stride, vit_v_stride;   // floats between the K^T (V) blocks of consecutive images
  int vit_col0;          // EPI_QKV_VIT: output column c of W counts as column c + vit_col0 of a fused [q | k | v] projection
                         // (a [k | v] weight of a cross-attention runs with vit_col0 = hidden)
  float vit_qscale;
  int n_tiles;           // 16-col tiles (EPI_SILU: gate/up tiles interleaved, n_tiles = 2 * d_ff/16)
  float* out;            // EPI_STORE [M][ldo] / EPI_RESID x[M][ldo]
  int ldo, n_valid;
  // EPI_STORE, optional (the lm_head over the visual span: the scorer's softmax statistics fused into the GEMM, round 5): per row and 64-column
  // block {max, sum exp(x - max)} -> rowstat[(row * rowstat_ld + block) * 2 + {0, 1}] (dd_row_block_stats: ONE definition of the arithmetic, shared
  // with the stand-alone scorer's k_row_partials, so that the fused and the stand-alone paths give the same bits)
  float* rowstat;
  int rowstat_ld;
  uint16_t* o_hi;        // EPI_SILU planes [M][ld_planes]
  uint16_t* o_lo;
  int ld_planes;
  // EPI_QKV
  float* qbuf;           // [M][q_dim]
  float* kc;             // layer K cache
  float* vc;
  int T_cap, q_tiles, k_tiles, q_dim, kv_dim, pos0;
  int kv16;              // EPI_QKV: the cache holds fp16
  int wf;                // 16-bit type of W, the A planes and the planes written by the epilogue (0 bf16, 1 fp16)
  const float* rope_cos;
  const float* rope_sin;
  // EPI_QKV over the prompts of several sequences laid back to back (dd_lm_prefill_group): seq_rows > 0 = rows per sequence
  // (a multiple of 128: a row block never straddles two sequences); row r belongs to sequence r / seq_rows, is its position
  // pos0 + r % seq_rows, is live below its sequence's length and writes that sequence's cache.  The per-sequence table lives in
  // device memory (arrays inside this by-value struct would be indexed per lane and end up in scratch)
  int seq_rows;
  const struct SeqTab* seq_tab;
  size_t seq_off_k, seq_off_v;   // this layer's offset into every sequence's K / V cache (floats of the storage)
};
struct SeqTab {
  int32_t T[32];
  float* kc[32];
  float* vc[32];
};
int ddk_gemm(int epi, const GemmArgs& a, hipStream_t st);

int ddk_attn_prefill(const float* qbuf, const float* kc, const float* vc, int T, int T_cap, int n_heads, int n_kv,
                     uint16_t* o_hi, uint16_t* o_lo, const uint8_t* drop_plane, int drop_bit, int span_start,
                     int span_len, int q0, hipStream_t st, u32x4_t* xop_out = nullptr, int kv16 = 0, int wf = 0);   // q0 = position of query row 0
// xop_out: write the rows as packed decode-GEMV operand planes (row-major [32 rows]) instead of the GEMM's A planes
struct SeqTab;
int ddk_put_seq_tab(const SeqTab& tab, SeqTab* dev, hipStream_t st);   // host table -> device, by value through a launch
int ddk_attn_vit_mfma_batch(const float* q, const SeqTab* tab, int n, int img_rows, int T, int Tc, int n_heads, uint16_t* o_hi, uint16_t* o_lo,
                            hipStream_t st, int head_pitch);   // ddk_attn_vit_mfma for n images in one launch (q / planes: img_rows rows per image)
int ddk_attn_prefill_seqs(const float* qbuf, const SeqTab* tab, size_t off_k, size_t off_v, int n, int seq_rows, int max_T, int T_cap,
                          int n_heads, int n_kv, uint16_t* o_hi, uint16_t* o_lo, hipStream_t st, int kv16, int wf);   // n sequences, one launch
int ddk_pack_embed_rows(const float* rows, int n, int rows_cap, int d, float* x, const float* normw, u32x4_t* xop, float* ssq,
                        int ssq_ld, hipStream_t st, int wf = 0);
int ddk_chunk_positions(DDState* rows, const DDState* base, int n, hipStream_t st);
int ddk_scatter_kv_rows(const float* kr, const float* vr, int n, int kv_dim, float* kc, float* vc, int T_cap, const DDState* base,
                        hipStream_t st, int kv16 = 0);
int ddk_mean_rows(float* rows, int K, int ld, int n, const int32_t* gate, hipStream_t st);

// ---- small glue -------------------------------------------------------------------------------
// x[0..8)[d] <- embed[cur_tok] (all rows equal), xop <- split(normw * x), ssq slot 0
int ddk_embed_rows(const uint16_t* embed, int d, const DDState* state, float* x, const float* normw, u32x4_t* xop,
                   float* ssq, int ssq_ld, hipStream_t st, const int32_t* skip_if = nullptr, int wf = 0);
struct EmbedLanes {
  const DDState* state[72];   // row m embeds the current token of this sequence (null: row unused)
};
int ddk_embed_rows_lanes(const uint16_t* embed, int d, const EmbedLanes& lanes, int rows, float* x, const float* normw,
                         u32x4_t* xop, float* ssq, int ssq_ld, hipStream_t st, int wf = 0);   // rows = 8, 16, 32, 64 or 72
int ddk_embed_tokens(const uint16_t* embed, int d, const int32_t* tokens, int n, float* x, hipStream_t st, int wf = 0);
struct CommitLanes {       // winners of up to 8 sequences appended to their caches in one launch
  const float* knew[8];
  const float* vnew[8];
  float* kc[8];
  float* vc[8];
  const DDState* state[8];
  size_t lsk, lsv;
  int kv16;
};
int ddk_commit_kv_lanes(const CommitLanes& t, int n, int n_layers, int rows_per_layer, int kv_dim, int T_cap, hipStream_t st);
int ddk_commit_kv(const float* knew, const float* vnew, int n_layers, int rows_per_layer, int kv_dim, float* kc,
                  float* vc, size_t layer_stride_k, size_t layer_stride_v, int T_cap, const DDState* state,
                  int use_winner, hipStream_t st, int kv16 = 0);
int ddk_final_norm_rows(const float* x, int rows, int d, const float* w, float eps, float* out, hipStream_t st);
int ddk_kv_sums(const float* kc, const float* vc, int n_layers, size_t lsk, size_t lsv, int n_kv, int T_cap, int T,
                double* out, hipStream_t st, int kv16 = 0);

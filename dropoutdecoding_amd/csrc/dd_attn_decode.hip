// Decode attention of the K-way masked-context step (gfx950, wave64): single-query attention of all rows of a pass (ensemble
// members x GQA group, or the rows of several sequences) over the shared prefix cache, per-member drop bits, flash-decoding
// partial + combine.  fp16 cache: on the matrix cores (k_attn_partial16); fp32 cache: VALU (k_attn_partial).
// Reference anchors: the attention inside the LM forward the reference calls at models/llava.py:294-303,350-359, with the 2-D
// mask of models/llava.py:346-349.
#include <type_traits>

#include "dd_lm_kernels.h"
#include "dd_lm_device.h"

// ===============================================================================================
// decode attention: partial (one wave per kv head x 64-key split) + combine
// ===============================================================================================
#define ATT_MAX_SPLITS 160

// One workgroup (4 waves) per (kv head, 64-key tile).  Every wave requests its share of the K tile (8 of the 32
// 16-byte d-chunks, keys on lanes) AND of the V tile (16 keys, two per instruction) before anything else, so the
// whole 64 KiB tile is in flight at once and the kernel pays one memory latency, not 64.  Scores are reduced over
// the four waves through LDS, softmax statistics are per tile (flash-decoding), P.V partials are reduced the same way.
// GH = q heads of the GQA group handled by one workgroup (blockIdx.z picks the slice): 32 rows per workgroup (8 members
// x 4 heads) need 122 KiB of LDS and 156 VGPRs, i.e. one workgroup per CU; two slices of 16 rows run two per CU and
// read the K/V tile twice through L2.
// ML == 1 (lanes): NBT == 1 and GH == G; blockIdx.z is the ROW of the pass = the sequence whose cache this workgroup
// reads; results go to the 8-rows-per-head layout the 8-row combine reads.
// ML == 2 (groups): NBT == 8; blockIdx.z = group * (G / GH) + GQA slice; group g's 8 rows (members of sequence g) read
// that sequence's cache and drop bits; results go to an (8 * a.lane_groups)-rows-per-head layout (row = 8 * group + member).
// (fp32 cache; the fp16 cache goes through k_attn_partial16 below)
int g_attn32_nopk = 0;         // dd_tools_set_tuning key 43 (experiment): the fp32 tile pass with scalar instead of packed FP32 multiply-adds
int g_attn32_lds_pad = 0;      // dd_tools_set_tuning key 39 (debug): bytes added to the fp32 attention kernel's dynamic LDS request
template <int NBT, int G, int GH, int ML = 0, int DBG = 0>      // DBG: per-workgroup checksums into a.dbg (libdropdec_tools.so's race bisect; more registers)
__global__ __launch_bounds__(256) void k_attn_partial(AttnDecodeArgs a) {
  constexpr int R = NBT * GH;        // rows of this workgroup
  const int lane_rows = a.n_lanes > 8 ? 16 : 8;   // ML == 1: rows per q head in the buffers (what the combine is built for)
  const int RT = ML == 1 ? lane_rows * G : (ML == 2 ? 8 * a.lane_groups * G : NBT * G);   // rows per kv head in the partial buffers
  // ML == 2 with NBT < 8: the 8 members of a group are split over 8 / NBT workgroups (member offset mo); each re-reads
  // the K/V tile through L2 but carries 1 / (8 / NBT) of the LDS and VALU work, which is what bounds the 8-row variant
  constexpr int MSPLIT = ML == 2 ? 8 / NBT : 1;
  const int zz = ML == 2 ? blockIdx.z % ((G / GH) * MSPLIT) : 0;
  const int g0 = ML == 1 ? 0 : (ML == 2 ? (zz / MSPLIT) * GH : blockIdx.z * GH);
  const int mo = ML == 2 ? (zz % MSPLIT) * NBT : 0;
  const int lane_row = ML == 1 ? blockIdx.z : (ML == 2 ? blockIdx.z / ((G / GH) * MSPLIT) : 0);
  // partial-buffer row of this workgroup's row r (= local head r / NBT, member r % NBT)
  auto buf_row = [&](int r) -> int {
    if (ML == 1) return r * lane_rows + lane_row;
    if (ML == 2) return (g0 + r / NBT) * 8 * a.lane_groups + lane_row * 8 + mo + r % NBT;
    return g0 * NBT + r;
  };
  extern __shared__ __align__(16) float att_sh[];
  float* q_sh = att_sh;                       // [R][128]
  float* s_part = q_sh + R * HEAD_DIM;        // [4][R][64]
  float* p_sh = s_part + 4 * R * ATT_SPLIT;   // [64][R]
  float* o_part = p_sh + ATT_SPLIT * R;       // [4][R][128]
  if (a.skip_if && *a.skip_if) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kvh = blockIdx.x, split = blockIdx.y;
  const int T = ML ? a.lane_state[lane_row]->T : (a.state ? a.state->T : a.T), t0 = split * ATT_SPLIT;
  if (t0 >= T) return;   // shorter lane / stale graph: this tile does not exist (the combine skips it as well)
  const float* kc_l = ML ? a.lane_kc[lane_row] : a.kc;
  const float* vc_l = ML ? a.lane_vc[lane_row] : a.vc;
  const uint8_t* bits_l = ML ? a.lane_bits[lane_row] : a.drop_bits;
  const int span0 = ML ? a.lane_span_start[lane_row] : a.span_start, spanL = ML ? a.lane_span_len[lane_row] : a.span_len;
  const int q_dim = a.n_heads * HEAD_DIM;
  const int nkeys = min(ATT_SPLIT, T - t0);
  const int half = lane >> 5, dq = lane & 31;

  // 1. all K / V requests of this wave (addresses clamped to the last live key; dead keys get p = 0)
  const int kt = t0 + min(lane, nkeys - 1);
  f32x4_t k4[8], v4[8];
  {
    const float* kbase = kc_l + (((size_t)kvh * 32 + wave * 8) * a.T_cap + kt) * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) k4[i] = *(const f32x4_t*)(kbase + (size_t)i * a.T_cap * 4);
    const float* vbase = vc_l + ((size_t)kvh * a.T_cap + t0) * HEAD_DIM + dq * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int key = min(wave * 16 + 2 * j + half, nkeys - 1);
      v4[j] = *(const f32x4_t*)(vbase + (size_t)key * HEAD_DIM);
    }
  }
  uint32_t bits = 0;
  if (bits_l && lane < nkeys) {
    int ka = t0 + lane;
    if (ka >= span0 && ka < span0 + spanL) bits = bits_l[ka - span0];
  }
  uint32_t dsum[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};       // debug checksums (a.dbg)
  auto dacc = [&](int slot, float v, int salt) { dsum[slot] += __float_as_uint(v) * (uint32_t)(2 * salt + 1); };
  if constexpr (DBG == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dacc(0, k4[i].x, i * 4), dacc(0, k4[i].y, i * 4 + 1), dacc(0, k4[i].z, i * 4 + 2), dacc(0, k4[i].w, i * 4 + 3);
      dacc(1, v4[i].x, i * 4), dacc(1, v4[i].y, i * 4 + 1), dacc(1, v4[i].z, i * 4 + 2), dacc(1, v4[i].w, i * 4 + 3);
    }
  }
  // 2. q rows (r = g*NBT + m) into LDS
  for (int i = tid; i < R * HEAD_DIM; i += 256) {
    int r = i / HEAD_DIM, d = i % HEAD_DIM, g = g0 + r / NBT;
    int m = ML == 1 ? lane_row : mo + r % NBT;            // row within its group (live if < nb)
    int qrow = ML == 2 ? lane_row * 8 + m : m;            // row of the pass
    q_sh[i] = (m < a.nb) ? a.qbuf[(size_t)qrow * q_dim + (kvh * G + g) * HEAD_DIM + d] : 0.f;
  }
  __syncthreads();
  // 3. partial scores over this wave's 32 d values
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float sp = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4_t q4 = *(const f32x4_t*)&q_sh[r * HEAD_DIM + (wave * 8 + i) * 4];
      if constexpr (DBG == 1) dacc(2, q4.x + q4.y + q4.z + q4.w, r * 8 + i);
      sp += q4.x * k4[i].x + q4.y * k4[i].y + q4.z * k4[i].z + q4.w * k4[i].w;
    }
    s_part[(wave * R + r) * ATT_SPLIT + lane] = sp;
  }
  __syncthreads();
  // 4. softmax statistics of the tile: wave w owns rows r = w, w+4, ...
  const float scaling = 0.08838834764831845f;  // head_dim ** -0.5
  for (int r = wave; r < R; r += 4) {
    int m = ML == 1 ? 0 : mo + r % NBT;  // lanes: bit 0 of the sequence's own (leak) bits; groups: the member's bit
    float sv = (s_part[(0 * R + r) * ATT_SPLIT + lane] + s_part[(1 * R + r) * ATT_SPLIT + lane]) +
               (s_part[(2 * R + r) * ATT_SPLIT + lane] + s_part[(3 * R + r) * ATT_SPLIT + lane]);
    if constexpr (DBG == 1) dacc(3, sv, r);
    sv *= scaling;
    if (lane >= nkeys || ((bits >> (a.bit0 + m)) & 1u)) sv = -INFINITY;  // zero in the 2-D mask: weight exactly 0
    float mx = dd_wave_max(sv);
    float p = (sv == -INFINITY) ? 0.f : expf(sv - mx);
    float l = dd_wave_sum(p);
    if constexpr (DBG == 1) dacc(4, p, r);
    p_sh[lane * R + r] = p;
    if (lane == 0) {
      float* ml = a.part_ml + (((size_t)kvh * gridDim.y + split) * RT + buf_row(r)) * 2;
      ml[0] = mx;
      ml[1] = l;
    }
  }
  __syncthreads();
  // 5. P.V over this wave's 16 keys (lanes 0-31 even keys, 32-63 odd keys; 4 consecutive d per lane)
  f32x4_t acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int key = wave * 16 + 2 * j + half;
    const float* pr = &p_sh[min(key, ATT_SPLIT - 1) * R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (DBG == 1 && dq == 0) dacc(5, pr[r], (wave * 8 + j) * R + r);
      if constexpr (DBG == 2) {              // (experiment, dd_tools_set_tuning key 43) the same fused multiply-adds as SCALAR instructions: no v_pk_fma_f32
        const float pv = pr[r];
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[r].x) : "v"(pv), "v"(v4[j].x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[r].y) : "v"(pv), "v"(v4[j].y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[r].z) : "v"(pv), "v"(v4[j].z));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[r].w) : "v"(pv), "v"(v4[j].w));
      } else {
        acc[r] += pr[r] * v4[j];   // p = 0 for dead / dropped keys
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    f32x4_t o = acc[r];
    if constexpr (DBG == 2) {
      float t0 = __shfl_xor(o.x, 32), t1 = __shfl_xor(o.y, 32), t2 = __shfl_xor(o.z, 32), t3 = __shfl_xor(o.w, 32);
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(o.x) : "v"(t0));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(o.y) : "v"(t1));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(o.z) : "v"(t2));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(o.w) : "v"(t3));
    } else {
    o.x += __shfl_xor(o.x, 32);
    o.y += __shfl_xor(o.y, 32);
    o.z += __shfl_xor(o.z, 32);
    o.w += __shfl_xor(o.w, 32);
    }
    if (half == 0) *(f32x4_t*)&o_part[(wave * R + r) * HEAD_DIM + dq * 4] = o;
  }
  __syncthreads();
  // 6. fixed-order sum over the four waves
  for (int i = tid; i < R * HEAD_DIM; i += 256) {
    float o = (o_part[i] + o_part[R * HEAD_DIM + i]) + (o_part[2 * R * HEAD_DIM + i] + o_part[3 * R * HEAD_DIM + i]);
    if constexpr (DBG == 1) dacc(6, o, i);
    if (ML) {
      int r = i / HEAD_DIM, dd = i % HEAD_DIM;
      a.part_o[(((size_t)kvh * gridDim.y + split) * RT + buf_row(r)) * HEAD_DIM + dd] = o;
    } else {
      a.part_o[(((size_t)kvh * gridDim.y + split) * RT + g0 * NBT) * HEAD_DIM + i] = o;
    }
  }
  if (DBG == 1 && a.dbg) {
    uint32_t* slot = a.dbg + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) << 3);
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      uint32_t v = dsum[c];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) atomicAdd(slot + c, v);
    }
  }
}

// Merge of the tiles of one (head, row) + the row's own new key / value (each ensemble member attends to the shared prefix + ITS
// OWN new token) + hi/lo packing for o_proj: 128 threads (thread = output dimension d), shared by the stand-alone kernel
// (tiles from the partial buffers in memory) and by the all-tiles form of k_attn_partial16 (tiles still in LDS) — ONE body, with
// explicit fused multiply-adds, so that a row's bits do not depend on which of the two produced it.
// ld_ml(tile) -> (max, sum) of the tile for this row; ld_o(tile) -> its un-normalised output at dimension d.
// sh: 2 + ATT_MAX_SPLITS + 2 + 2 floats of shared memory for this group of 128 threads.
// PRE (round 6; the stand-alone merge over the partial buffers in memory): the outputs of the first ATT_COMB_PRE tiles are requested at the top,
// with every other load of the block, instead of behind the two barriers — one memory round trip per launch instead of two.  The multiply-adds
// and their order are unchanged: the same bits (tests/test_gpu_single_stream_attn.py; dd_tools_set_tuning key 55 = 0 restores the late loads).
// Worth 0.3 % of a two-sweep single-sequence step, 3 % of its speculative step (profiles/r06_lab/attn_one_launch_ab.log).
#define ATT_COMB_SH (2 + ATT_MAX_SPLITS + 2 + 2)
#define ATT_COMB_PRE 16
template <int PRE = 0, typename LD_ML, typename LD_O>
__device__ __forceinline__ void attn_combine_core(const AttnDecodeArgs& a, int head, int kvh, int m, bool wide, int d, int splits, float* sh,
                                                  LD_ML ld_ml, LD_O ld_o, bool store = true) {
  float* red = sh;
  float* w_sh = sh + 2;
  float* mx_sh = sh + 2 + ATT_MAX_SPLITS;
  float* den_sh = mx_sh + 2;
  const int lane = d & 63, wv = d >> 6;
  const int q_dim = a.n_heads * HEAD_DIM, kv_dim = a.n_kv * HEAD_DIM;
  const float scaling = 0.08838834764831845f;
  const int grp = wide ? (a.half_planes ? m >> 2 : m >> 3) : 0;                  // group (half planes: sequence) of the row
  const int mrow = a.half_planes ? (m & 3) : (m & 7);
  const float* knew_r = (wide && a.knew_g[grp]) ? a.knew_g[grp] + (size_t)mrow * kv_dim : a.knew + (size_t)m * kv_dim;
  const float* vnew_r = (wide && a.vnew_g[grp]) ? a.vnew_g[grp] + (size_t)mrow * kv_dim : a.vnew + (size_t)m * kv_dim;
  // every load of this block is issued here, before the first dependent use
  float qd = a.qbuf[(size_t)m * q_dim + head * HEAD_DIM + d];
  float kd = knew_r[kvh * HEAD_DIM + d];
  float vd = vnew_r[kvh * HEAD_DIM + d];
  float ms0 = -INFINITY, ls0 = 0.f, ms1 = -INFINITY, ls1 = 0.f;   // two tiles per thread: up to 256 tiles
  if (d < splits) ld_ml(d, ms0, ls0);
  if (d + 128 < splits) ld_ml(d + 128, ms1, ls1);
  float opre[PRE ? PRE : 1];
  if constexpr (PRE > 0) {
#pragma unroll
    for (int sp = 0; sp < PRE; ++sp) opre[sp] = sp < splits ? ld_o(sp) : 0.f;
  }
  float part = dd_wave_sum(qd * kd);
  float mloc = dd_wave_max(fmaxf(ms0, ms1));
  if (lane == 0) { red[wv] = part; mx_sh[wv] = mloc; }
  __syncthreads();
  float s_self = (red[0] + red[1]) * scaling;
  float M = fmaxf(s_self, fmaxf(mx_sh[0], mx_sh[1]));
  float w0 = (ms0 == -INFINITY) ? 0.f : expf(ms0 - M), w1 = (ms1 == -INFINITY) ? 0.f : expf(ms1 - M);
  if (d < splits) w_sh[d] = w0;
  if (d + 128 < splits) w_sh[d + 128] = w1;
  float dl = dd_wave_sum(__builtin_fmaf(w1, ls1, w0 * ls0));
  if (lane == 0) den_sh[wv] = dl;
  __syncthreads();
  float w_self = expf(s_self - M);
  float den = w_self + (den_sh[0] + den_sh[1]);
  float num = w_self * vd;
  if constexpr (PRE > 0) {
#pragma unroll
    for (int sp = 0; sp < PRE; ++sp)
      if (sp < splits) num = __builtin_fmaf(w_sh[sp], opre[sp], num);
    for (int sp = PRE; sp < splits; ++sp) num = __builtin_fmaf(w_sh[sp], ld_o(sp), num);
  } else {
    for (int sp = 0; sp < splits; ++sp) num = __builtin_fmaf(w_sh[sp], ld_o(sp), num);
  }
  if (!store) return;
  if (wide) xop_store16(a.xop_out, head * HEAD_DIM + d, m, num / den, q_dim >> 5, a.wf);
  else xop_store(a.xop_out, head * HEAD_DIM + d, m, num / den, a.wf);
}

// The same tile pass for the fp16 cache on the matrix cores.  The VALU form above costs ~1,100 vector instructions per wave and
// tile at 8 rows — the grouped decode attention ran at 2.3 TB/s of K/V bytes, bound by them, not by HBM.  Here
//   S^T = K . Q^T :  A = the 16-byte chunks of K as the cache stores them (16 keys x 32 d per fragment), B = the rows' q split
//                    into fp16 hi + lo columns (8 rows x {hi, lo} = the 16 columns; 2^-22 relative), wave w takes keys 16w..16w+15;
//   O^T = V^T . P^T: A = the cache's octets of V (16 d x 32 keys per fragment), B = the tile's probabilities as fp16 hi + lo,
//                    wave w takes output dimensions 32w..32w+31, so no cross-wave reduction of the outputs is needed.
// Tile softmax statistics, masks, buffers and row maps are those of k_attn_partial; all variants (rows alone, members of one
// sequence, lanes, groups) go through this one body, so a row's bits do not depend on the pass it rides in.
// FULL: the workgroup takes ALL tiles of its (kv head, sequence) — contexts of up to ATT_FULL_TILES tiles —, keeps the tiles'
// statistics and outputs in LDS instead of the partial buffers and runs the merge itself (attn_combine_core, the body the
// stand-alone k_attn_combine runs over the buffers): no partial-buffer round trip, no second launch, the same bits.
#define ATT_FULL_TILES 12
// PF = 0: ONE register set for a tile's K / V fragments instead of two (the next tile is requested only when this one is done): 96 instead of
// 144 VGPRs, so that a workgroup fits on a CU BESIDE a nine-plane slice GEMV workgroup (2 x 200-208 VGPRs per SIMD lane, 144 KiB of LDS) —
// the rider sweeps' attention then overlaps the other branches' qkv / o_proj / down streams instead of waiting for their CUs (round 4: 64-lane
// step 36.4 -> 35.0 ms).  Same arithmetic, only the load timing differs: the same bits.
// sh: ATTN16_SH_FLOATS(NBT * GH) floats of the caller's static LDS (q rows, tile probabilities, per-wave maxima / sums) — passed in so that the
// two bodies of the rider kernel, of which a workgroup runs one, share one allocation: with a buffer each the kernel asked for 19.5 KiB at
// GQA 4 and could not sit beside a 144-KiB slice GEMV workgroup.
#define ATTN16_SH_FLOATS(R_) ((((R_) + 7) / 8 * 8) * ((HEAD_DIM + 4) + (ATT_SPLIT + 4) + 8))
template <int NBT, int G, int GH, int ML, int FULL = 0, int PF = 1>
__device__ __forceinline__ void attn_partial16_body(const AttnDecodeArgs& a, const int bx, const int by, const int bz, float* const sh) {
  constexpr int R = NBT * GH, RB = (R + 7) / 8, RP = RB * 8;
  extern __shared__ __align__(16) float full_sh[];        // FULL: [tiles][RP][HEAD_DIM] outputs, then [tiles][RP][2] statistics
  float* const fo_sh = full_sh;
  float* const fml_sh = full_sh + (size_t)ATT_FULL_TILES * RP * HEAD_DIM;
  __shared__ float comb_sh[FULL ? 2 : 1][FULL ? ATT_COMB_SH : 1];
  const int lane_rows = a.n_lanes > 8 ? 16 : 8;
  const int RT = ML == 1 ? lane_rows * G : (ML == 2 ? 8 * a.lane_groups * G : NBT * G);
  constexpr int MSPLIT = ML == 2 ? 8 / NBT : 1;
  const int zz = ML == 2 ? bz % ((G / GH) * MSPLIT) : 0;
  const int g0 = ML == 1 ? 0 : (ML == 2 ? (zz / MSPLIT) * GH : bz * GH);
  const int mo = ML == 2 ? (zz % MSPLIT) * NBT : 0;
  const int lane_row = ML == 1 ? bz : (ML == 2 ? bz / ((G / GH) * MSPLIT) : 0);
  // half planes (ML == 2, NBT == 4): the workgroup's four rows are the members of sequence 2 * plane + (mo >> 2)
  const int seq_row = (ML == 2 && a.half_planes) ? 2 * lane_row + (mo >> 2) : lane_row;
  const int mo_bit = (ML == 2 && a.half_planes) ? 0 : mo;      // member index of row 0 (drop bit, liveness)
  auto buf_row = [&](int r) -> int {
    if (ML == 1) return r * lane_rows + lane_row;
    if (ML == 2) return (g0 + r / NBT) * 8 * a.lane_groups + lane_row * 8 + mo + r % NBT;
    return g0 * NBT + r;
  };
  float (*const q_sh)[HEAD_DIM + 4] = (float (*)[HEAD_DIM + 4])sh;
  float (*const p_sh)[ATT_SPLIT + 4] = (float (*)[ATT_SPLIT + 4])(sh + RP * (HEAD_DIM + 4));
  float (*const red_sh)[4][RP] = (float (*)[4][RP])(sh + RP * (HEAD_DIM + 4) + RP * (ATT_SPLIT + 4));
  if (a.skip_if && *a.skip_if) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, h4 = lane >> 4;
  const int kvh = bx;
  const int T = ML ? a.lane_state[seq_row]->T : (a.state ? a.state->T : a.T);
  // this workgroup's key tiles: tiles_per_wg consecutive ones (the launcher sizes the grid for ONE round of workgroups: with a
  // tile per workgroup the 8-sequence pass had 1.25 rounds, a quarter-full second one)
  const int tpw = FULL ? ATT_FULL_TILES : (a.tiles_per_wg > 0 ? a.tiles_per_wg : 1);
  const int split0 = by * tpw, n_live = (T + ATT_SPLIT - 1) / ATT_SPLIT;
  const int split1 = min(min(split0 + tpw, FULL ? ATT_FULL_TILES : a.splits_stride), n_live);
  if (split0 >= split1) return;   // shorter lane / stale graph: these tiles do not exist (the combine skips them as well)
  const dd_half* kc_l = (const dd_half*)(ML ? a.lane_kc[seq_row] : a.kc);
  const dd_half* vc_l = (const dd_half*)(ML ? a.lane_vc[seq_row] : a.vc);
  const uint8_t* bits_l = ML ? a.lane_bits[seq_row] : a.drop_bits;
  const int span0 = ML ? a.lane_span_start[seq_row] : a.span_start, spanL = ML ? a.lane_span_len[seq_row] : a.span_len;
  const int q_dim = a.n_heads * HEAD_DIM;

  // every K / V request of a tile at once (dead keys: the last live key's chunk / whatever the octet holds — cache memory is
  // zero-initialised and only ever holds finite values; their probabilities are exactly 0)
  // member index of this lane's rows within their sequence (8 blk is a multiple of NBT: the same for every row block)
  const int m_lane = ML == 1 ? 0 : mo_bit + (c & 7) % NBT;
  auto load_tile = [&](int split, u32x4_t (&kf)[4], u32x4_t (&vf)[2][2], uint32_t& kmask) {
    const int t0 = split * ATT_SPLIT, nkeys = min(ATT_SPLIT, T - t0);
    const int key = t0 + min(16 * wave + c, nkeys - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const u32x4_t*)(kc_l + (((size_t)kvh * 16 + 4 * ks + h4) * a.T_cap + key) * 8);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
        vf[dt][k2] = *(const u32x4_t*)(vc_l + (((size_t)kvh * (a.T_cap >> 3) + (t0 >> 3) + 4 * k2 + h4) * HEAD_DIM + 32 * wave + 16 * dt + c) * 8);
    kmask = 0u;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {              // drop flags of this lane's keys 16 w + 4 h4 + reg for its member: bit reg
      const int kk = 16 * wave + 4 * h4 + reg, ka = t0 + kk;
      const uint32_t bits = (bits_l && kk < nkeys && ka >= span0 && ka < span0 + spanL) ? bits_l[ka - span0] : 0u;
      kmask |= ((bits >> (a.bit0 + m_lane)) & 1u) << reg;
    }
  };
  u32x4_t kf[PF ? 2 : 1][4], vf[PF ? 2 : 1][2][2];
  uint32_t kbits[PF ? 2 : 1];
  load_tile(split0, kf[0], vf[0], kbits[0]);
  // q rows (r = g * NBT + m) into LDS, zero for rows past R or past the live members
  for (int i = tid; i < RP * HEAD_DIM; i += 256) {
    int r = i / HEAD_DIM, d = i % HEAD_DIM, g = g0 + r / NBT;
    int m = ML == 1 ? lane_row : mo + r % NBT;
    int qrow = ML == 2 ? lane_row * 8 + m : m;
    const int mlive = ML == 2 ? mo_bit + r % NBT : m;      // member index within its sequence
    q_sh[r][d] = (r < R && mlive < a.nb) ? a.qbuf[(size_t)qrow * q_dim + (kvh * G + g) * HEAD_DIM + d] : 0.f;
  }
  __syncthreads();
  // fp32 x[8] -> this lane's B column: the fp16 hi part (columns 0-7) or the lo part (columns 8-15) of row c & 7
  auto split_col = [&](const float* x) -> u32x4_t {
    dd_f16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dd_half hi = (dd_half)x[j];
      float rem = x[j] - (float)hi;
      o[j] = c < 8 ? hi : (dd_half)rem;
    }
    return __builtin_bit_cast(u32x4_t, o);
  };
  // the rows' Q^T columns do not depend on the tile
  u32x4_t qb[RB][4];
#pragma unroll
  for (int blk = 0; blk < RB; ++blk) {
    const float* qr = &q_sh[8 * blk + (c & 7)][8 * h4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
      *(f32x4_t*)&x[0] = *(const f32x4_t*)(qr + 32 * ks);
      *(f32x4_t*)&x[4] = *(const f32x4_t*)(qr + 32 * ks + 4);
      qb[blk][ks] = split_col(x);
    }
  }
  const float scaling = 0.08838834764831845f;  // head_dim ** -0.5

  auto tile = [&](int split, const u32x4_t (&kfr)[4], const u32x4_t (&vfr)[2][2], const uint32_t kb) {
    const int t0 = split * ATT_SPLIT, nkeys = min(ATT_SPLIT, T - t0);
    float pmax[RB], sv[RB][4];
#pragma unroll
    for (int blk = 0; blk < RB; ++blk) {
      // S^T of this wave's 16 keys against block blk's 8 rows
      f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dd_f16x8_t, kfr[ks]), __builtin_bit_cast(dd_f16x8_t, qb[blk][ks]), sacc, 0, 0, 0);
      // hi + lo columns; mask; the wave's maximum per row (lanes c < 8 carry row 8 blk + c, keys 16 w + 4 h4 + reg)
      const int row = 8 * blk + (c & 7);
      float mx = -INFINITY;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float s = (sacc[reg] + __shfl_down(sacc[reg], 8)) * scaling;
        const int kk = 16 * wave + 4 * h4 + reg;
        if (kk >= nkeys || ((kb >> reg) & 1u)) s = -INFINITY;   // zero in the 2-D mask: weight exactly 0
        sv[blk][reg] = s;
        mx = fmaxf(mx, s);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (h4 == 0 && c < 8) red_sh[0][wave][row] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < RB; ++blk) {
      const int row = 8 * blk + (c & 7);
      const float M = fmaxf(fmaxf(red_sh[0][0][row], red_sh[0][1][row]), fmaxf(red_sh[0][2][row], red_sh[0][3][row]));
      pmax[blk] = M;
      float l = 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float p = (sv[blk][reg] == -INFINITY) ? 0.f : expf(sv[blk][reg] - M);
        l += p;
        if (c < 8) p_sh[row][16 * wave + 4 * h4 + reg] = p;
      }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      if (h4 == 0 && c < 8) red_sh[1][wave][row] = l;
    }
    __syncthreads();
    if (wave == 0 && h4 == 0 && c < 8) {
#pragma unroll
      for (int blk = 0; blk < RB; ++blk) {
        const int row = 8 * blk + c;
        if (row < R) {
          float* ml = FULL ? fml_sh + ((size_t)split * RP + row) * 2
                           : a.part_ml + (((size_t)kvh * a.splits_stride + split) * RT + buf_row(row)) * 2;
          ml[0] = pmax[blk];
          ml[1] = (red_sh[1][0][row] + red_sh[1][1][row]) + (red_sh[1][2][row] + red_sh[1][3][row]);
        }
      }
    }
    // O^T of this wave's 32 output dimensions: P^T columns from LDS (all 64 keys), V^T fragments from the registers
#pragma unroll
    for (int blk = 0; blk < RB; ++blk) {
      u32x4_t pb[2];
      const float* pr = &p_sh[8 * blk + (c & 7)][8 * h4];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        float x[8];
        *(f32x4_t*)&x[0] = *(const f32x4_t*)(pr + 32 * k2);
        *(f32x4_t*)&x[4] = *(const f32x4_t*)(pr + 32 * k2 + 4);
        pb[k2] = split_col(x);
      }
      const int row = 8 * blk + c;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4_t oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
          oacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dd_f16x8_t, vfr[dt][k2]), __builtin_bit_cast(dd_f16x8_t, pb[k2]), oacc, 0, 0, 0);
        f32x4_t o;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) o[reg] = oacc[reg] + __shfl_down(oacc[reg], 8);
        if (c < 8 && row < R) {
          if constexpr (FULL) *(f32x4_t*)&fo_sh[((size_t)split * RP + row) * HEAD_DIM + 32 * wave + 16 * dt + 4 * h4] = o;
          else *(f32x4_t*)&a.part_o[(((size_t)kvh * a.splits_stride + split) * RT + buf_row(row)) * HEAD_DIM + 32 * wave + 16 * dt + 4 * h4] = o;
        }
      }
    }
  };
  if constexpr (!PF) {
    for (int sp = split0; sp < split1; ++sp) {
      if (sp > split0) {
        __syncthreads();                                  // p_sh / red_sh are reused
        load_tile(sp, kf[0], vf[0], kbits[0]);
      }
      tile(sp, kf[0], vf[0], kbits[0]);
    }
  } else {
    // tiles in pairs with the other register set prefetching: the next tile's loads are in flight while this one is computed
    for (int sp = split0; sp < split1; sp += 2) {
      if (sp + 1 < split1) load_tile(sp + 1, kf[PF ? 1 : 0], vf[PF ? 1 : 0], kbits[PF ? 1 : 0]);
      tile(sp, kf[0], vf[0], kbits[0]);
      if (sp + 1 < split1) {
        __syncthreads();                                  // p_sh / red_sh are reused
        if (sp + 2 < split1) load_tile(sp + 2, kf[0], vf[0], kbits[0]);
        tile(sp + 1, kf[PF ? 1 : 0], vf[PF ? 1 : 0], kbits[PF ? 1 : 0]);
        if (sp + 2 < split1) __syncthreads();
      }
    }
  }
  if constexpr (FULL) {
    // the merge, two rows at a time (threads 0-127 / 128-255 each run the 128-thread core on a row of their own)
    __syncthreads();
    const int half = tid >> 7, d = tid & 127;
    const bool wide = ML == 2 || (ML == 1 && a.n_lanes > 8);
#pragma unroll 1
    for (int it = 0; 2 * it < R; ++it) {
      const int rr = 2 * it + half, r = rr < R ? rr : R - 1;
      const int g = g0 + r / NBT;
      const int m = ML == 2 ? lane_row * 8 + mo + r % NBT : (ML == 1 ? lane_row : r % NBT);
      const bool store = rr < R && (ML != 0 || m < a.nb);
      attn_combine_core(
          a, kvh * G + g, kvh, m, wide, d, n_live, comb_sh[half],
          [&](int t, float& mx, float& l) { mx = fml_sh[((size_t)t * RP + r) * 2], l = fml_sh[((size_t)t * RP + r) * 2 + 1]; },
          [&](int t) -> float { return fo_sh[((size_t)t * RP + r) * HEAD_DIM + d]; }, store);
      __syncthreads();                                  // the core's scratch is reused by the next pair of rows
    }
  }
}

template <int NBT, int G, int GH, int ML, int FULL = 0>
__global__ __launch_bounds__(256) void k_attn_partial16(AttnDecodeArgs a) {
  __shared__ __align__(16) float sh[ATTN16_SH_FLOATS(NBT * GH)];
  attn_partial16_body<NBT, G, GH, ML, FULL>(a, blockIdx.x, blockIdx.y, blockIdx.z, sh);
}
// Rider sweeps (dd_engine.hip group_step_rider): the members of eight sequences (a: the groups form, workgroups z < za) AND the
// riding un-masked rows of up to eight other sequences (u: the lanes form, one row per sequence) in ONE launch — the two are
// independent and each alone leaves the chip half idle; the bodies are the ones above, so every row keeps its bits.
// One register set (PF = 0) under a register cap wherever the rows fit: MHA 94 VGPRs (cap 96), GQA with up to 16 rows per workgroup 119 (cap 128) —
// beside the nine-plane GEMVs' 2 x 200-208 (bf16) or 2 x 176-184 (fp8) per SIMD lane; 32 rows per workgroup keep two sets.
template <int G, int GH, int NBT, int PF = ((G == 1 || NBT * GH <= 16) ? 0 : 1)>
__global__ __launch_bounds__(256, PF ? 1 : (G == 1 ? 5 : 4)) void k_attn_partial16_ride(AttnDecodeArgs a, AttnDecodeArgs u, int za) {
  __shared__ __align__(16) float sh[ATTN16_SH_FLOATS(NBT * GH > G ? NBT * GH : G)];
  if ((int)blockIdx.z < za) attn_partial16_body<NBT, G, GH, 2, 0, PF>(a, blockIdx.x, blockIdx.y, blockIdx.z, sh);
  else attn_partial16_body<1, G, G, 1, 0, PF>(u, blockIdx.x, blockIdx.y, blockIdx.z - za, sh);
}
int g_attn16_ride_pf = 0;        // dd_tools_set_tuning key 47: 1 = the two-register-set (round-3) form for every rider sweep

// grid (n_heads, nb), block 128 (thread = d): the merge above over the partial buffers in memory
template <int NBT, int G, int PRE = 0>
__device__ __forceinline__ void attn_combine_body(const AttnDecodeArgs& a, int splits_grid, const int bx, const int by) {
  constexpr int R = NBT * G;
  __shared__ float sh[ATT_COMB_SH];
  if (a.skip_if && *a.skip_if) return;
  const int head = bx, m = by, d = threadIdx.x, kvh = head / G, g = head % G;
  const int r = g * NBT + m;
  // `splits_grid` is the stride of the partial buffers (tiles the partial kernel was launched with); a lane's row only
  // has the tiles of its own, possibly shorter, sequence
  const int splits = a.n_lanes ? (a.lane_state[a.lane_groups ? (a.half_planes ? m >> 2 : m >> 3) : m]->T + ATT_SPLIT - 1) / ATT_SPLIT
                               : (a.state ? (a.state->T + ATT_SPLIT - 1) / ATT_SPLIT : splits_grid);
  const float* mlb = a.part_ml + ((size_t)kvh * splits_grid * R + r) * 2;
  const size_t ml_stride = (size_t)R * 2;
  const float* po = a.part_o + ((size_t)kvh * splits_grid * R + r) * HEAD_DIM + d;
  const size_t o_stride = (size_t)R * HEAD_DIM;
  attn_combine_core<PRE>(
      a, head, kvh, m, NBT > 8, d, splits, sh, [&](int t, float& mx, float& l) { mx = mlb[t * ml_stride], l = mlb[t * ml_stride + 1]; },
      [&](int t) -> float { return po[(size_t)t * o_stride]; });
}
template <int NBT, int G, int PRE = 0>
__global__ __launch_bounds__(HEAD_DIM) void k_attn_combine(AttnDecodeArgs a, int splits_grid) {
  attn_combine_body<NBT, G, PRE>(a, splits_grid, blockIdx.x, blockIdx.y);
}
int g_attn_comb_pre = 1;         // dd_tools_set_tuning key 55: 0 = the merge launches request the tiles' outputs behind their barriers (until round 6)
template <int NBT, int G>
static void launch_combine(const AttnDecodeArgs& a, int rows, int splits, hipStream_t st) {
  if (g_attn_comb_pre) k_attn_combine<NBT, G, ATT_COMB_PRE><<<dim3(a.n_heads, rows), HEAD_DIM, 0, st>>>(a, splits);
  else k_attn_combine<NBT, G><<<dim3(a.n_heads, rows), HEAD_DIM, 0, st>>>(a, splits);
}
// the merges of a rider sweep in one launch: rows 0..63 the members' (a), rows 64.. the riding rows' (u)
// (rows_a = 8 x the member planes; RU = rows per head of the riding rows' partial buffers: 8, or 16 with more than eight of them)
template <int G, int RU, int PRE = 0>
__global__ __launch_bounds__(HEAD_DIM) void k_attn_combine_ride(AttnDecodeArgs a, AttnDecodeArgs u, int splits_a, int splits_u, int rows_a) {
  if ((int)blockIdx.y < rows_a) attn_combine_body<64, G, PRE>(a, splits_a, blockIdx.x, blockIdx.y);
  else attn_combine_body<RU, G, PRE>(u, splits_u, blockIdx.x, blockIdx.y - rows_a);
}

// Key tiles the partial kernel is LAUNCHED with: the live count rounded up to a multiple of 4 (workgroups of tiles past
// the sequence end return at once), so that the launch shape — and with it a captured hipGraph — stays valid for 256
// more tokens instead of 64.  The combine takes the live count from the device-side length.
int ddk_attn_grid_tiles(int T, int T_cap) {
  int tiles = (T + ATT_SPLIT - 1) / ATT_SPLIT;
  int up = (tiles + 3) / 4 * 4, cap = T_cap / ATT_SPLIT;
  return up < cap ? up : (cap > tiles ? cap : tiles);
}

// k_attn_partial16: key tiles per workgroup (the next tile's loads travel while the current one is computed)
int g_attn16_tpw = 0;   // dd_set_tuning key 21: key tiles per workgroup of the fp16-cache decode attention (0: sized for one round)
static void attn16_grid(AttnDecodeArgs& b, int splits, int wg_per_tile) {
  b.splits_stride = splits;
  // up to 4 tiles per workgroup while at least ~1,000 workgroups remain (measured on the 32-lane step: 30.1 / 29.5 / 29.3 ms with
  // 1 / 2 / 4 tiles; a single sequence's 384 tile-workgroups stay one tile each)
  int tpw = (int)((long)wg_per_tile * splits / 1024);
  tpw = tpw < 1 ? 1 : (tpw > 4 ? 4 : tpw);
  b.tiles_per_wg = g_attn16_tpw > 0 ? g_attn16_tpw : tpw;
}
// all-tiles form (FULL) of k_attn_partial16: contexts of up to ATT_FULL_TILES tiles and enough (kv head, sequence) workgroups
int g_attn16_full = 1;   // dd_set_tuning key 22
// fp16 cache, GQA: all q heads of a kv group in ONE workgroup (up to 32 rows: four MFMA row blocks) read every K / V tile once instead of
// once per GQA slice of 16 rows — the same rows through the same blocks of eight, so the same bits.  dd_tools_set_tuning key 38.
int g_attn16_gh_all = 0;
static bool attn16_full_ok(int splits, int wgs) { return g_attn16_full && splits <= ATT_FULL_TILES && wgs >= 128; }
template <int NBT, int G, int GH, int ML>
static int launch_attn16_full(const AttnDecodeArgs& a, dim3 grid, hipStream_t st) {
  constexpr int R = NBT * GH, RP = (R + 7) / 8 * 8;
  constexpr size_t smem = (size_t)ATT_FULL_TILES * RP * (HEAD_DIM + 2) * sizeof(float);
  static bool attr = false;
  if (!attr && smem > 32 * 1024) {
    DD_HIP(hipFuncSetAttribute((const void*)k_attn_partial16<NBT, G, GH, ML, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  k_attn_partial16<NBT, G, GH, ML, 1><<<grid, 256, smem, st>>>(a);
  return DD_OK;
}
template <int NBT, int G>
static int launch_attn(const AttnDecodeArgs& a, hipStream_t st) {
  constexpr int GH = (NBT * G > 16) ? 2 : G;     // at most 16 rows per workgroup
  constexpr int R = NBT * GH;
  int splits = ddk_attn_grid_tiles(a.T, a.T_cap);
  DD_REQUIRE(splits >= 1 && splits <= ATT_MAX_SPLITS, "attn: %d key tiles unsupported (1..%d)", splits, ATT_MAX_SPLITS);
  size_t smem = (size_t)(R * HEAD_DIM + 4 * R * ATT_SPLIT + ATT_SPLIT * R + 4 * R * HEAD_DIM) * sizeof(float);
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    DD_HIP(hipFuncSetAttribute((const void*)k_attn_partial<NBT, G, GH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  if (a.kv16) {
    AttnDecodeArgs b = a;
    constexpr int GHALL = (NBT * G > 32) ? 2 : G;
    if (GHALL != GH && g_attn16_gh_all) {
      attn16_grid(b, splits, a.n_kv * (G / GHALL));
      k_attn_partial16<NBT, G, GHALL, 0><<<dim3(a.n_kv, (splits + b.tiles_per_wg - 1) / b.tiles_per_wg, G / GHALL), 256, 0, st>>>(b);
    } else {
      attn16_grid(b, splits, a.n_kv * (G / GH));
      k_attn_partial16<NBT, G, GH, 0><<<dim3(a.n_kv, (splits + b.tiles_per_wg - 1) / b.tiles_per_wg, G / GH), 256, 0, st>>>(b);
    }
  } else {
    k_attn_partial<NBT, G, GH><<<dim3(a.n_kv, splits, G / GH), 256, smem, st>>>(a);
  }
  launch_combine<NBT, G>(a, a.nb, splits, st);
  return DD_OK;
}

// fused base pass of up to 8 sequences: one single-query attention per row, each over its own cache
template <int G>
static int launch_attn_lanes(const AttnDecodeArgs& a, hipStream_t st) {
  constexpr int R = G;
  int splits = ddk_attn_grid_tiles(a.max_T, a.T_cap);
  DD_REQUIRE(splits >= 1 && splits <= ATT_MAX_SPLITS, "attn: %d key tiles unsupported (1..%d)", splits, ATT_MAX_SPLITS);
  size_t smem = (size_t)(R * HEAD_DIM + 4 * R * ATT_SPLIT + ATT_SPLIT * R + 4 * R * HEAD_DIM) * sizeof(float);
  if (a.kv16 && attn16_full_ok(splits, a.n_kv * a.n_lanes)) {      // all tiles per workgroup, merge included: no combine launch
    return launch_attn16_full<1, G, G, 1>(a, dim3(a.n_kv, 1, a.n_lanes), st);
  }
  if (a.kv16) {
    AttnDecodeArgs b = a;
    attn16_grid(b, splits, a.n_kv * a.n_lanes);
    k_attn_partial16<1, G, G, 1><<<dim3(a.n_kv, (splits + b.tiles_per_wg - 1) / b.tiles_per_wg, a.n_lanes), 256, 0, st>>>(b);
  } else {
    k_attn_partial<1, G, G, 1><<<dim3(a.n_kv, splits, a.n_lanes), 256, smem, st>>>(a);
  }
  if (a.n_lanes > 8) launch_combine<16, G>(a, a.nb, splits, st);
  else launch_combine<8, G>(a, a.nb, splits, st);
  return DD_OK;
}

// multi-group pass: members of NG sequences (8 rows each), every group over its own cache
static int g_attn_msplit = 1;   // workgroups per group of 8 members in the grouped decode attention (1, 2 or 4; dd_set_tuning key 10)
void ddk_set_attn_split(int v) { g_attn_msplit = v; }

template <int G, int NG, int NBT>
static int launch_attn_groups_n(const AttnDecodeArgs& a, hipStream_t st) {
  constexpr int GH = (NBT * G > 16) ? 2 : G;
  constexpr int R = NBT * GH;
  int splits = ddk_attn_grid_tiles(a.max_T, a.T_cap);
  DD_REQUIRE(splits >= 1 && splits <= ATT_MAX_SPLITS, "attn: %d key tiles unsupported (1..%d)", splits, ATT_MAX_SPLITS);
  size_t smem = (size_t)(R * HEAD_DIM + 4 * R * ATT_SPLIT + ATT_SPLIT * R + 4 * R * HEAD_DIM) * sizeof(float);
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    DD_HIP(hipFuncSetAttribute((const void*)k_attn_partial<NBT, G, GH, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  if (a.kv16 && NBT == 8 && g_attn16_full >= 2 && attn16_full_ok(splits, a.n_kv * NG * (G / GH))) {   // key 22 = 2: measured no faster (33 vs 26 + 6 us: one workgroup per CU walks ten tiles in a row)
    return launch_attn16_full<NBT, G, GH, 2>(a, dim3(a.n_kv, 1, NG * (G / GH)), st);
  }
  if (a.kv16) {
    AttnDecodeArgs b = a;
    constexpr int GHALL = (NBT * G > 32) ? 2 : G;
    if (GHALL != GH && g_attn16_gh_all) {
      attn16_grid(b, splits, a.n_kv * NG * (G / GHALL) * (8 / NBT));
      k_attn_partial16<NBT, G, GHALL, 2><<<dim3(a.n_kv, (splits + b.tiles_per_wg - 1) / b.tiles_per_wg, NG * (G / GHALL) * (8 / NBT)), 256, 0, st>>>(b);
    } else {
      attn16_grid(b, splits, a.n_kv * NG * (G / GH) * (8 / NBT));
      k_attn_partial16<NBT, G, GH, 2><<<dim3(a.n_kv, (splits + b.tiles_per_wg - 1) / b.tiles_per_wg, NG * (G / GH) * (8 / NBT)), 256, 0, st>>>(b);
    }
  } else {
    bool traced = false;
    if constexpr (G == 1 && NBT == 8) {              // (the traced instantiation exists for the MHA shape the race bisect runs)
      if (a.dbg) {
        k_attn_partial<NBT, G, GH, 2, 1><<<dim3(a.n_kv, splits, NG * (G / GH) * (8 / NBT)), 256, smem, st>>>(a);
        traced = true;
      } else if (g_attn32_nopk) {
        k_attn_partial<NBT, G, GH, 2, 2><<<dim3(a.n_kv, splits, NG * (G / GH) * (8 / NBT)), 256, smem, st>>>(a);
        traced = true;
      }
    }
    if (traced) {
    } else if (g_attn32_lds_pad > 0) {
      DD_HIP(hipFuncSetAttribute((const void*)k_attn_partial<NBT, G, GH, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(smem + g_attn32_lds_pad)));
      k_attn_partial<NBT, G, GH, 2><<<dim3(a.n_kv, splits, NG * (G / GH) * (8 / NBT)), 256, smem + g_attn32_lds_pad, st>>>(a);
    } else
    k_attn_partial<NBT, G, GH, 2><<<dim3(a.n_kv, splits, NG * (G / GH) * (8 / NBT)), 256, smem, st>>>(a);
  }
  launch_combine<8 * NG, G>(a, 8 * NG, splits, st);
  return DD_OK;
}
// multi-group pass: members of NG sequences (8 rows each), every group over its own cache
template <int G, int NG>
static int launch_attn_groups(const AttnDecodeArgs& a, hipStream_t st) {
  if (a.half_planes) return launch_attn_groups_n<G, NG, 4>(a, st);      // two workgroups per plane, one per sequence of the pair
  if (g_attn_msplit == 4) return launch_attn_groups_n<G, NG, 2>(a, st);
  if (g_attn_msplit == 2) return launch_attn_groups_n<G, NG, 4>(a, st);
  return launch_attn_groups_n<G, NG, 8>(a, st);
}

template <int G, int NBT, int GH>
static int launch_attn_ride_gh(const AttnDecodeArgs& a, const AttnDecodeArgs& u, int planes_m, hipStream_t st) {
  const int splits_a = ddk_attn_grid_tiles(a.max_T, a.T_cap), splits_u = ddk_attn_grid_tiles(u.max_T, u.T_cap);
  DD_REQUIRE(splits_a >= 1 && splits_a <= ATT_MAX_SPLITS && splits_u >= 1 && splits_u <= ATT_MAX_SPLITS,
             "attn: %d / %d key tiles unsupported (1..%d)", splits_a, splits_u, ATT_MAX_SPLITS);
  AttnDecodeArgs b = a, v = u;
  attn16_grid(b, splits_a, a.n_kv * planes_m * (G / GH) * (8 / NBT));
  attn16_grid(v, splits_u, u.n_kv * u.n_lanes);
  const int ya = (splits_a + b.tiles_per_wg - 1) / b.tiles_per_wg, yu = (splits_u + v.tiles_per_wg - 1) / v.tiles_per_wg;
  const int za = planes_m * (G / GH) * (8 / NBT);
  if (g_attn16_ride_pf) k_attn_partial16_ride<G, GH, NBT, 1><<<dim3(a.n_kv, ya > yu ? ya : yu, za + u.n_lanes), 256, 0, st>>>(b, v, za);
  else k_attn_partial16_ride<G, GH, NBT><<<dim3(a.n_kv, ya > yu ? ya : yu, za + u.n_lanes), 256, 0, st>>>(b, v, za);
  const dim3 cgrid(a.n_heads, 8 * planes_m + u.nb);
  if (g_attn_comb_pre) {
    if (u.n_lanes > 8) k_attn_combine_ride<G, 16, ATT_COMB_PRE><<<cgrid, HEAD_DIM, 0, st>>>(a, u, splits_a, splits_u, 8 * planes_m);
    else k_attn_combine_ride<G, 8, ATT_COMB_PRE><<<cgrid, HEAD_DIM, 0, st>>>(a, u, splits_a, splits_u, 8 * planes_m);
  } else {
    if (u.n_lanes > 8) k_attn_combine_ride<G, 16><<<cgrid, HEAD_DIM, 0, st>>>(a, u, splits_a, splits_u, 8 * planes_m);
    else k_attn_combine_ride<G, 8><<<cgrid, HEAD_DIM, 0, st>>>(a, u, splits_a, splits_u, 8 * planes_m);
  }
  return DD_OK;
}
template <int G, int NBT>
static int launch_attn_ride(const AttnDecodeArgs& a, const AttnDecodeArgs& u, int planes_m, hipStream_t st) {
  constexpr int GH16 = (NBT * G > 16) ? 2 : G, GHALL = (NBT * G > 32) ? 2 : G;
  if constexpr (GHALL != GH16) {
    if (g_attn16_gh_all) return launch_attn_ride_gh<G, NBT, GHALL>(a, u, planes_m, st);
  }
  return launch_attn_ride_gh<G, NBT, GH16>(a, u, planes_m, st);
}
// a: the member pass of a rider sweep (lane_groups == 8: eight sequences, one per plane — or, with half planes, fourteen in planes_m = 7
// planes), u: un-masked rows of up to sixteen sequences (lanes form, own partial buffers), both over fp16 caches: the two attentions of
// the sweep in one partial + one combine launch
int ddk_attn_decode_ride(const AttnDecodeArgs& a, const AttnDecodeArgs& u, int planes_m, hipStream_t st) {
  DD_REQUIRE(a.kv16 && u.kv16 && a.lane_groups == 8 && a.n_lanes == 8 && a.nb >= 1 && a.nb <= 8 && u.n_lanes >= 1 && u.n_lanes <= 16 &&
                 u.nb == u.n_lanes && !u.lane_groups && a.n_heads == u.n_heads && a.n_kv == u.n_kv && a.part_o != u.part_o && a.part_ml != u.part_ml &&
                 (a.half_planes ? (planes_m == 7 && a.nb <= 4) : planes_m == 8),
             "attn_ride: a member pass of eight sequences (or fourteen in seven half planes) + up to sixteen riding rows, fp16 caches, separate partial buffers");
  const int G = a.n_heads / a.n_kv;
  DD_REQUIRE(a.n_heads % a.n_kv == 0 && (G == 1 || G == 2 || G == 4), "attn_ride: GQA group %d unsupported (1, 2, 4)", G);
  int rc;
  if (a.half_planes) rc = G == 1 ? launch_attn_ride<1, 4>(a, u, planes_m, st) : (G == 2 ? launch_attn_ride<2, 4>(a, u, planes_m, st) : launch_attn_ride<4, 4>(a, u, planes_m, st));
  else rc = G == 1 ? launch_attn_ride<1, 8>(a, u, planes_m, st) : (G == 2 ? launch_attn_ride<2, 8>(a, u, planes_m, st) : launch_attn_ride<4, 8>(a, u, planes_m, st));
  if (rc != DD_OK) return rc;
  DD_CHECK_LAUNCH();
  return DD_OK;
}

int ddk_attn_decode(const AttnDecodeArgs& a, hipStream_t st) {
  DD_REQUIRE(a.n_heads % a.n_kv == 0, "attn: heads %d not a multiple of kv heads %d", a.n_heads, a.n_kv);
  int G = a.n_heads / a.n_kv;
  DD_REQUIRE(G == 1 || G == 2 || G == 4, "attn: GQA group %d unsupported (1, 2, 4)", G);
  int rc = DD_OK;                    // a launcher that refuses (too many key tiles, attribute failure) launches nothing
  if (a.n_lanes > 0 && a.lane_groups) {
    DD_REQUIRE((a.lane_groups == 2 || a.lane_groups == 4 || a.lane_groups == 8) && a.n_lanes == a.lane_groups && a.nb >= 1 && a.nb <= 8,
               "attn: a multi-group pass takes 2, 4 or 8 sequences of up to 8 members");
    DD_REQUIRE(!a.half_planes || (a.kv16 && a.lane_groups == 8 && a.nb <= 4), "attn: half planes take eight planes over fp16 caches, K <= 4");
    if (a.lane_groups == 8) {
      if (G == 1) rc = launch_attn_groups<1, 8>(a, st);
      else if (G == 2) rc = launch_attn_groups<2, 8>(a, st);
      else rc = launch_attn_groups<4, 8>(a, st);
    } else if (a.lane_groups == 2) {
      if (G == 1) rc = launch_attn_groups<1, 2>(a, st);
      else if (G == 2) rc = launch_attn_groups<2, 2>(a, st);
      else rc = launch_attn_groups<4, 2>(a, st);
    } else {
      if (G == 1) rc = launch_attn_groups<1, 4>(a, st);
      else if (G == 2) rc = launch_attn_groups<2, 4>(a, st);
      else rc = launch_attn_groups<4, 4>(a, st);
    }
    if (rc != DD_OK) return rc;
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
  if (a.n_lanes > 0) {
    DD_REQUIRE(a.n_lanes <= 16 && a.nb == a.n_lanes, "attn: %d lanes for %d rows", a.n_lanes, a.nb);
    if (G == 1) rc = launch_attn_lanes<1>(a, st);
    else if (G == 2) rc = launch_attn_lanes<2>(a, st);
    else rc = launch_attn_lanes<4>(a, st);
    if (rc != DD_OK) return rc;
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
  bool one = a.nb == 1;
  if (G == 1) rc = one ? launch_attn<1, 1>(a, st) : launch_attn<8, 1>(a, st);
  else if (G == 2) rc = one ? launch_attn<1, 2>(a, st) : launch_attn<8, 2>(a, st);
  else rc = one ? launch_attn<1, 4>(a, st) : launch_attn<8, 4>(a, st);
  if (rc != DD_OK) return rc;
  DD_CHECK_LAUNCH();
  return DD_OK;
}


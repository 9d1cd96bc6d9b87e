// Slice-resident decode GEMV for 16 / 32 rows (NG = 2 / 4 operand planes of 8 rows each), bf16 weights.
//
// Why: with several operand planes the K-split-over-waves kernel (k_gemv_groups) reads NG KiB of packed operand from L2
// for every KiB of weights it streams from HBM, once per workgroup — 2-4x the weight bytes per launch — and the CU's
// L2 -> L1 path, not HBM, sets its time (measured with tools/gemv_lab: dropping those reads takes the 32-row gate/up
// GEMV from 45 to 34 us, o_proj from 16 to 8, down from 42 to 18).
//
// How: K is cut into the SAME 8 interleaved slices k_gemv uses (slice q = k-steps q, q+8, ...; k_gemv's wave q).  A
// workgroup takes ONE slice of all NG operand planes into LDS (read once from L2) and its 8 waves each own whole weight
// tiles over that slice: a wave streams its tiles' k-steps straight to registers through a ring of U requests and takes the
// B operands from LDS (ds_read_b128, conflict-free, lgkmcnt — so the weight queue depth is not tied to operand registers).
// The slice's accumulators are the ones wave q of k_gemv would hold for that tile: same k order, same MFMA chain.  They
// are written out as partial sums (hi + lo columns folded: D[m] + D[m+8], the first add of k_gemv's reduction) and a
// finishing kernel (k_gemv_finish in dd_lm_kernels.hip) adds the slices in k_gemv's order — pairs of waves first:
// y = sum_p ((hi+lo)(2p) + (hi+lo)(2p+1)) — and runs the epilogue: the arithmetic of k_gemv / k_gemv_groups bit for bit.
// CH = 2: a workgroup holds a PAIR of slices (2q, 2q+1) and adds the pair itself — half the partial-sum traffic, one
// workgroup per CU (128 KiB of operands at four planes).
//
// Long K (slice larger than the LDS chunk): the slice is staged in chunks of CS k-steps; the next chunk's operand pieces
// are requested into registers at the START of the current chunk (they return ahead of the weight requests issued after
// them) and committed to LDS between two barriers, so the weight ring never drains.  Then every wave owns exactly one
// tile group (its accumulators live across the chunks).
#pragma once
#include "dd_common.h"
#include "dd_lm_device.h"

struct SliceArgs {
  const u32x4_t* W;     // packed weight tiles [n_tiles][S][64]
  const u32x4_t* xop;   // NG operand planes [NG][S][64]
  float* part;          // partial sums [8 / CH][n_tiles][NG][128]
  int S;                // K / 32
  int n_groups;         // tile groups of TW tiles
  int G;                // workgroups per slice (grid = 8 * G)
  // folded RMSNorm of the operand rows: workgroup 0 assembles rstd(row) from the producer's sum-of-squares slots while the
  // others stream, so that the finishing kernel starts with one 4-byte load instead of a reduction (ssq_in null: no norm)
  const float* ssq_in;
  int ssq_n, ssq_ld;
  float inv_k, eps;
  float* rstd_out;      // [8 * NG]
  // halves == 2: the launch covers 2 * NG planes as two half passes of NG planes each over the SAME weight tiles: workgroup
  // ids b and b + 8 (same XCD under round-robin dispatch, scheduled back to back) run the two halves of one (slice, group), so
  // the second reads the tiles from that XCD's L2 — HBM sees the weights once, each CU carries NG planes of MFMA / LDS work.
  // Partial sums are laid out for 2 * NG planes (plane = half * NG + h).  grid = 2 * (8 / CH) * G, a multiple of 16.
  int halves;
  // 1: weight tiles with the default cache policy instead of non-temporal loads (A/B, dd_tools_set_tuning key 36: beside other branches that
  // stream the same matrix moments later a tile may be served from the Infinity Cache)
  int temporal;
  // 1 (the product since round 5's last day): the launch carries one workgroup more per half (blockIdx.x >= the streaming workgroups' count) that
  // does nothing but the rows' rstd.  Until then workgroup 0 did it BEFORE its own weight stream: the launch ended 4-6 us late, waiting for that
  // one workgroup (tools/seq_lab.hip: gate/up at 72 rows 48.1 -> 43.5 us).  0: the old placement (A/B, dd_tools_set_tuning key 50).  Same bits.
  int rstd_wg;
};
// (temporal bit 3, value 8: TIMING EXPERIMENT — k_gemv_slices_seq writes a tile pair's partial sums as soon as the pair is complete, as it did until
// round 5's last day, instead of at the end of the wave's stream)
// (temporal bit 1, value 2: TIMING EXPERIMENT of the tools library — the kernels skip the operand planes' staging and compute on whatever the LDS
// holds; how much of a launch the blocking stage-in costs.  Bit 2, value 4: the slice-pair kernels write no partial sums.  Never set by the product.)
// The timing experiments (bits 1..3: garbage results, or the round-5 store placement) exist only in libdropdec_tools.so's build of this
// header (build.py TOOLS_VARIANTS: -DDD_TIMING_EXPERIMENTS); in the product library the branches are compiled out.
#ifdef DD_TIMING_EXPERIMENTS
#define DD_TEXP(a_, bit_) ((a_).temporal & (bit_))
#else
#define DD_TEXP(a_, bit_) 0
#endif
__device__ __forceinline__ u32x4_t dd_ldw(int temporal, const u32x4_t* p) { return (temporal & 1) ? *p : __builtin_nontemporal_load(p); }

// rstd(row) = 1 / sqrt(mean(x^2) + eps) from per-workgroup partial sums of squares; wave w of the calling workgroup
// (8 waves) assembles rows w, w + 8, ...  ONE definition for every kernel that needs it: the sum order is part of the result.
template <int NG>
__device__ __forceinline__ void dd_rows_rstd(const float* ssq_in, int ssq_n, int ssq_ld, float inv_k, float eps, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4_t sv[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    sv[g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (4 * lane < ssq_n) sv[g] = *(const f32x4_t*)(ssq_in + (size_t)(wave + 8 * g) * ssq_ld + 4 * lane);
  }
  const int i0 = 4 * lane;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float v = 0.f;
    if (i0 < ssq_n) v += sv[g].x;
    if (i0 + 1 < ssq_n) v += sv[g].y;
    if (i0 + 2 < ssq_n) v += sv[g].z;
    if (i0 + 3 < ssq_n) v += sv[g].w;
    for (int i = lane + 256; i < ssq_n; i += 64) v += ssq_in[(size_t)(wave + 8 * g) * ssq_ld + i];
    v = dd_wave_sum(v);
    if (lane == 0) out[wave + 8 * g] = 1.0f / sqrtf(v * inv_k + eps);
  }
}

// 128 floats per (slice, tile, plane): element (n = output row of the tile, m = row of the plane) at
// ((n >> 2) * 8 + m) * 4 + (n & 3)
__device__ __forceinline__ int dd_part_index(int n, int m) { return (((n >> 2) * 8 + m) << 2) + (n & 3); }

// grid = (8 / CH) * G workgroups of 512 threads; dynamic LDS = CH * min(SPW, CS) * NG KiB
// EPI_TAG: the epilogue of the finishing kernel that follows (unused here: it only keeps the instantiations of the four
// matrices apart, so that a kernel trace names each of them)
// PROG = 1 (round 5; whole-slice kernels only): progressive stage-in, as in k_gemv_slices_seq below — only the first ring block's pieces are staged
// before the weight stream starts, the following blocks' pieces are requested while the wave's FIRST tile group consumes the block before them.
template <int TW, int NG, int U, int SPW, int CS, int CH = 1, int WF = 0, int EPI_TAG = 0, int PROG = 0>
__global__ __launch_bounds__(512) void k_gemv_slices(SliceArgs a) {
  constexpr int NCH = (SPW + CS - 1) / CS;             // LDS chunks per slice
  constexpr int PW = (CH * CS * NG + 7) / 8;           // operand pieces (1 KiB) per wave and chunk
  static_assert(SPW % U == 0 || NCH > 1 || SPW < U, "ring depth must divide the slice");
  static_assert(CH == 1 || NCH == 1, "slice pairs only when a whole slice fits the LDS chunk");
  extern __shared__ __align__(16) u32x4_t xs[];        // [CH][CS][NG][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NQ = 8 / CH;                           // slice sets
  int bid = blockIdx.x, half = 0;
  if (a.halves == 2) {
    half = (bid >> 3) & 1;
    bid = ((bid >> 4) << 3) | (bid & 7);
  }
  const int qs = bid % NQ, j = bid / NQ;
  const int q = qs * CH;                               // first slice of this workgroup
  const size_t xplane = (size_t)a.S * 64;
  const u32x4_t* const xop = a.xop + (size_t)half * NG * xplane;
  const int ng_all = a.halves == 2 ? 2 * NG : NG, plane0 = half * NG;
  const int n_tiles = a.n_groups * TW;
  {
    const int work_blocks = (a.halves == 2 ? 2 : 1) * NQ * a.G;
    if ((int)blockIdx.x >= work_blocks) {              // SliceArgs::rstd_wg: the workgroup(s) behind the streaming ones — a half's rows' rstd
      const int hh = blockIdx.x - work_blocks;
      if (a.ssq_in) dd_rows_rstd<NG>(a.ssq_in + (size_t)hh * 8 * NG * a.ssq_ld, a.ssq_n, a.ssq_ld, a.inv_k, a.eps, a.rstd_out + hh * 8 * NG);
      return;
    }
  }
  if (!a.rstd_wg && a.ssq_in && (blockIdx.x == 0 || (a.halves == 2 && blockIdx.x == 8)))   // (old placement) the first workgroup of each half
    dd_rows_rstd<NG>(a.ssq_in + (size_t)half * 8 * NG * a.ssq_ld, a.ssq_n, a.ssq_ld, a.inv_k, a.eps, a.rstd_out + half * 8 * NG);

  u32x4_t xv[PW];
  auto stage_issue = [&](int c0, int n) {              // chunk = slice steps c0 .. c0+n-1: piece p = (slice, step, plane)
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      int p = wave + 8 * i;
      int pc = p < CH * n * NG ? p : 0;
      int ch = pc / (n * NG), r = pc % (n * NG);
      xv[i] = xop[(size_t)(q + ch + 8 * (c0 + r / NG)) * 64 + (r % NG) * xplane + lane];
    }
  };
  auto stage_commit = [&](int n) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      int p = wave + 8 * i;
      if (p < CH * n * NG) xs[(size_t)p * 64 + lane] = xv[i];
    }
  };
  auto fold = [&](f32x4_t v) -> f32x4_t {              // D[n][m] + D[n][m + 8]: hi + lo column of row m (lanes c < 8 of 16)
    v.x += __shfl_down(v.x, 8);
    v.y += __shfl_down(v.y, 8);
    v.z += __shfl_down(v.z, 8);
    v.w += __shfl_down(v.w, 8);
    return v;
  };
  auto store_partials = [&](int g, f32x4_t (&sum)[TW][NG]) {     // sum: already folded
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int h = 0; h < NG; ++h)
        if ((lane & 8) == 0) {
          const int l32 = (lane >> 4) * 8 + (lane & 7);
          *(f32x4_t*)&a.part[((((size_t)qs * n_tiles + (size_t)g * TW + t) * ng_all + plane0 + h) << 7) + l32 * 4] = sum[t][h];
        }
  };
  auto mfma = [](u32x4_t w, u32x4_t b, f32x4_t c) -> f32x4_t { return dd_mfma16<WF>(w, b, c); };
  const size_t wstep = 8 * 64;                         // one slice step = 8 k-steps of the tile row

  if constexpr (NCH == 1) {
    // ---- whole slice (or slice pair) resident: a wave walks its tile groups g = j + G * (wave + 8 i); per group it streams
    // CH * SPW steps (slice q, then slice q + 1); the request ring runs on across slice and group boundaries
    constexpr int NS = CH * SPW;                       // steps per group
    constexpr int UU = SPW < U ? SPW : U;
    constexpr int NB = NS / UU;
    static_assert(SPW % UU == 0, "ring depth must divide the slice");
    int g = j + a.G * wave;
    const bool any = g < a.n_groups;
    // weight address of step s (0 .. NS-1) of a group: slice q + s / SPW, slice step s % SPW
    auto woff = [&](int s) -> size_t { return (size_t)(s / SPW) * 64 + (size_t)(s % SPW) * wstep; };
    u32x4_t w[TW][UU];
    if (any) {
#pragma unroll
      for (int u = 0; u < UU; ++u)
#pragma unroll
        for (int t = 0; t < TW; ++t)
          w[t][u] = dd_ldw(a.temporal, a.W + ((size_t)(g * TW + t) * a.S + q) * 64 + lane + woff(u));
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PWB = (UU * NG + 7) / 8;             // operand pieces per wave and ring block (progressive stage-in)
    u32x4_t xb[PROG ? PWB : 1];
    auto issue_blk = [&](int b) {                      // pieces of steps b UU .. b UU + UU - 1 of the NS-step space (slice q + s / SPW, step s % SPW)
#pragma unroll
      for (int i = 0; i < PWB; ++i) {
        const int p = wave + 8 * i, pc = p < UU * NG ? p : 0;
        const int sidx = b * UU + pc / NG;
        xb[PROG ? i : 0] = xop[(size_t)(q + sidx / SPW + 8 * (sidx % SPW)) * 64 + (pc % NG) * xplane + lane];
      }
    };
    auto commit_blk = [&](int b) {
#pragma unroll
      for (int i = 0; i < PWB; ++i) {
        const int p = wave + 8 * i;
        if (p < UU * NG) xs[(size_t)(b * UU * NG + p) * 64 + lane] = xb[PROG ? i : 0];
      }
    };
    if constexpr (PROG) {
      if (!DD_TEXP(a, 2)) {
        issue_blk(0);
        commit_blk(0);
      }
      __syncthreads();
      if (!any) {                                      // no tile group: this wave only stages its share of the following blocks
#pragma unroll
        for (int b = 1; b < NB; ++b) {
          if (!DD_TEXP(a, 2)) {
            issue_blk(b);
            commit_blk(b);
          }
          __syncthreads();
        }
        return;
      }
    } else {
      if (!DD_TEXP(a, 2)) {
        stage_issue(0, SPW);
        stage_commit(SPW);
      }
      __syncthreads();
      if (!any) return;
    }
    bool first_group = PROG != 0;
    while (true) {
      const int gn = g + a.G * 8;
      const bool has_next = gn < a.n_groups;
      const u32x4_t* wp[TW];
      const u32x4_t* wn[TW];
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        wp[t] = a.W + ((size_t)(g * TW + t) * a.S + q) * 64 + lane;
        wn[t] = a.W + ((size_t)((has_next ? gn : g) * TW + t) * a.S + q) * 64 + lane;
      }
      f32x4_t acc[TW][NG], sum[TW][NG];
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int h = 0; h < NG; ++h) acc[t][h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      auto block = [&](int blk, bool request_next_group) {
#pragma unroll
        for (int u = 0; u < UU; ++u) {
          const int s = blk * UU + u;
          u32x4_t b[NG];
#pragma unroll
          for (int h = 0; h < NG; ++h) b[h] = xs[(size_t)(s * NG + h) * 64 + lane];
#pragma unroll
          for (int t = 0; t < TW; ++t) {
#pragma unroll
            for (int h = 0; h < NG; ++h) acc[t][h] = mfma(w[t][u], b[h], acc[t][h]);
            if (blk + 1 < NB) w[t][u] = dd_ldw(a.temporal, wp[t] + woff(s + UU));
            else if (request_next_group) w[t][u] = dd_ldw(a.temporal, wn[t] + woff(u));
          }
          __builtin_amdgcn_sched_barrier(0);           // keep consume -> re-request order
        }
        if ((blk + 1) * UU % SPW == 0) {               // a slice's chain is complete: fold it into the group's sum
          const int ch = (blk + 1) * UU / SPW - 1;
#pragma unroll
          for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int h = 0; h < NG; ++h) {
              f32x4_t f = fold(acc[t][h]);
              if (ch == 0) sum[t][h] = f;
              else sum[t][h] = sum[t][h] + f;          // (hi+lo)(2p) + (hi+lo)(2p+1)
              acc[t][h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
        }
      };
      auto pre = [&](int blk) {
        if constexpr (PROG) {
          if (first_group && blk + 1 < NB && !DD_TEXP(a, 2)) issue_blk(blk + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto post = [&](int blk) {
        if constexpr (PROG) {
          if (first_group && blk + 1 < NB) {           // workgroup-uniform: every wave with a tile group is in its first one
            if (!DD_TEXP(a, 2)) commit_blk(blk + 1);
            __syncthreads();
          }
        }
      };
      if (has_next) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          pre(blk);
          block(blk, true);
          post(blk);
        }
      } else {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          pre(blk);
          block(blk, false);
          post(blk);
        }
      }
      first_group = false;
      store_partials(g, sum);
      if (!has_next) break;
      g = gn;
    }
  } else {
    // ---- chunked slice: one tile group per wave, accumulators live across the chunks
    static_assert(CS % U == 0, "chunk must be a whole number of ring blocks");
    const int g = j + a.G * wave;
    const bool live = g < a.n_groups;
    const int gg = live ? g : 0;
    const u32x4_t* wp[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) wp[t] = a.W + ((size_t)(gg * TW + t) * a.S + q) * 64 + lane;
    stage_issue(0, CS);
    u32x4_t w[TW][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TW; ++t) w[t][u] = dd_ldw(a.temporal, wp[t] + (size_t)u * wstep);
    __builtin_amdgcn_sched_barrier(0);
    stage_commit(CS);
    __syncthreads();
    f32x4_t acc[TW][NG];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int h = 0; h < NG; ++h) acc[t][h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      constexpr int dummy = 0;
      (void)dummy;
      const int c0 = k * CS;
      const int n = (SPW - c0) < CS ? (SPW - c0) : CS;               // steps of this chunk (compile-time after unrolling)
      const int n_next = (SPW - c0 - CS) < CS ? (SPW - c0 - CS) : CS;
      if (k + 1 < NCH) stage_issue(c0 + CS, n_next);                  // requested ahead of the ring's later requests
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cc = 0; cc < CS; ++cc) {
        if (cc < n) {
          const int c = c0 + cc, u = c % U;
          u32x4_t b[NG];
#pragma unroll
          for (int h = 0; h < NG; ++h) b[h] = xs[(size_t)(cc * NG + h) * 64 + lane];
#pragma unroll
          for (int t = 0; t < TW; ++t) {
#pragma unroll
            for (int h = 0; h < NG; ++h) acc[t][h] = mfma(w[t][u], b[h], acc[t][h]);
            if (c + U < SPW) w[t][u] = dd_ldw(a.temporal, wp[t] + (size_t)(c + U) * wstep);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (k + 1 < NCH) {
        __syncthreads();                 // every wave has read this chunk
        stage_commit(n_next);
        __syncthreads();
      }
    }
    if (live) {
      f32x4_t sum[TW][NG];
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int h = 0; h < NG; ++h) sum[t][h] = fold(acc[t][h]);
      store_partials(g, sum);
    }
  }
}

// Slice PAIRS with one slice resident at a time (the eight-plane pass: a pair does not fit the LDS).  A workgroup owns the pair
// (2 qs, 2 qs + 1) for its tile groups (one tile each, up to MAXG per wave): it takes slice 2 qs into LDS, streams that slice of
// all its tiles keeping each tile's folded sum in registers, swaps slice 2 qs + 1 in, streams again, adds, and writes ONE
// partial sum per pair — half the partial-sum traffic of single slices (gate/up at 64 rows: 22.5 instead of 45 MB written and
// read back) for a second operand load per workgroup.  Same chains, same order: (hi+lo)(2p) + (hi+lo)(2p+1) as the pair kernels.
// grid = 4 * G workgroups of 512 threads; dynamic LDS = SPW * NG KiB; a.part laid out for NP = 4.
// Round 5: the folded sums of TWO tiles share one register set.  After the fold (hi + lo column of a row: lanes c and c + 8 of a 16-lane group)
// only half the lanes of an accumulator carry a result, so the sum of tile 2 s sits in the lanes with bit 3 clear and that of tile 2 s + 1 in the
// lanes with bit 3 set (v + shfl_xor(v, 8) gives hi + lo in the one half and lo + hi — the same bits — in the other): (MAXG + 1) / 2 sets instead
// of MAXG, 36 registers fewer at nine planes and two or three tiles per wave.  With that the nine-plane gate/up kernel fits on a SIMD beside the
// rider sweeps' attention workgroups and has room for eight weight requests in flight.
// (A progressive stage-in of the operand planes was built and measured here in round 5 and removed again: at eight and nine planes the per-block
// barriers of the first tile cost more than the staging they hide — 64-lane step 36.4 -> 37.7 ms, profiles/r05_progressive_stage_in.log; the
// whole-slice kernels above keep it for two and four planes, where it pays.)
// (amdgpu_waves_per_eu(1, 2): the LDS admits one workgroup per CU = two waves per SIMD; without the hint the compiler aims the two-tile forms at
// three waves and spills three registers to private scratch to get under 168 — which build.py refuses)
// NH = slices a workgroup adds up one after the other: 2 = the pairs above (the product); 4 = QUADS — a timing experiment of round 6 (tools
// library only: the sums come out in the order ((s0 + s1) + s2) + s3, which no other pass width produces): half the partial sums again for two more
// operand stage-ins per workgroup (tools key 53).
template <int NG, int U, int SPW, int MAXG, int WF = 0, int EPI_TAG = 0, int NH = 2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_gemv_slices_seq(SliceArgs a) {
  constexpr int NQS = 8 / NH;                          // workgroups per tile group: one per run of NH slices
  static_assert(SPW % U == 0, "ring depth must divide the slice");
  constexpr int PW = (SPW * NG + 7) / 8;               // operand pieces (1 KiB) per wave and slice
  constexpr int NB = SPW / U;
  constexpr int NSET = (MAXG + 1) / 2;                 // register sets of folded sums: two tiles each
  extern __shared__ __align__(16) u32x4_t xs[];        // [SPW][NG][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool hi_half = (lane & 8) != 0;
  const int qs = blockIdx.x % NQS, j = blockIdx.x / NQS;
  const size_t xplane = (size_t)a.S * 64;
  const int n_tiles = a.n_groups;
  if ((int)blockIdx.x >= NQS * a.G) {                    // SliceArgs::rstd_wg: the workgroup behind the streaming ones — the rows' rstd
    if (a.ssq_in) dd_rows_rstd<NG>(a.ssq_in, a.ssq_n, a.ssq_ld, a.inv_k, a.eps, a.rstd_out);
    return;
  }
  if (!a.rstd_wg && blockIdx.x == 0 && a.ssq_in) dd_rows_rstd<NG>(a.ssq_in, a.ssq_n, a.ssq_ld, a.inv_k, a.eps, a.rstd_out);   // (old placement)
  const size_t wstep = 8 * 64;
  int gidx[MAXG];
  int ng = 0;
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    gidx[i] = j + a.G * (wave + 8 * i);
    if (gidx[i] < n_tiles) ng = i + 1;
  }
  auto wptr = [&](int item) -> const u32x4_t* {        // item = slice half * MAXG + group slot (clamped to a live group)
    const int half = item / MAXG, gi = item % MAXG;
    const int g = gidx[gi < ng ? gi : 0] < n_tiles ? gidx[gi < ng ? gi : 0] : 0;
    return a.W + ((size_t)g * a.S + NH * qs + half) * 64 + lane;
  };
  auto stage = [&](int half) {                         // operand slice 2 qs + half -> LDS (all waves)
    u32x4_t xv[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      int p = wave + 8 * i;
      int pc = p < SPW * NG ? p : 0;
      xv[i] = a.xop[(size_t)(NH * qs + half + 8 * (pc / NG)) * 64 + (pc % NG) * xplane + lane];
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      int p = wave + 8 * i;
      if (p < SPW * NG) xs[(size_t)p * 64 + lane] = xv[i];
    }
  };
  auto fold2 = [&](f32x4_t v) -> f32x4_t {             // hi + lo column of a row, in BOTH lanes of the pair (a + b and b + a: the same bits)
    v.x += __shfl_xor(v.x, 8);
    v.y += __shfl_xor(v.y, 8);
    v.z += __shfl_xor(v.z, 8);
    v.w += __shfl_xor(v.w, 8);
    return v;
  };
  f32x4_t sum[NSET][NG];
#pragma unroll
  for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
    for (int h = 0; h < NG; ++h) sum[s_][h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto store_set = [&](int st) {                       // a set's two tiles are complete: each half of the lanes stores its tile
    const int my_gi = 2 * st + (hi_half ? 1 : 0);
    const int my_g = j + a.G * (wave + 8 * my_gi);      // = gidx[my_gi], as arithmetic (a lane-dependent index into gidx[] would put it in scratch)
    if (my_gi < MAXG && my_gi < ng && !DD_TEXP(a, 4)) {      // (temporal bit 2: timing experiment — no partial sums written)
      const int l32 = (lane >> 4) * 8 + (lane & 7);
#pragma unroll
      for (int h = 0; h < NG; ++h) *(f32x4_t*)&a.part[((((size_t)qs * n_tiles + my_g) * NG + h) << 7) + l32 * 4] = sum[st][h];
    }
  };
  // the ring runs over the wave's items (slice half, group) in order; the next item's first U tiles are requested while the
  // current item's last block is consumed — also across the operand swap
  u32x4_t w[U];
  {
    const u32x4_t* p0 = wptr(0);
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = dd_ldw(a.temporal, p0 + (size_t)u * wstep);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (!DD_TEXP(a, 2)) stage(0);
  __syncthreads();
#pragma unroll
  for (int half = 0; half < NH; ++half) {
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
      const int item = half * MAXG + gi;
      const bool live = gi < ng;                       // wave-uniform
      const bool last_item = item == NH * MAXG - 1;
      const u32x4_t* wp = wptr(item);
      const u32x4_t* wn = wptr(last_item ? item : item + 1);
      if (live) {
        f32x4_t acc[NG];
#pragma unroll
        for (int h = 0; h < NG; ++h) acc[h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int s = blk * U + u;
            u32x4_t b[NG];
#pragma unroll
            for (int h = 0; h < NG; ++h) b[h] = xs[(size_t)(s * NG + h) * 64 + lane];
#pragma unroll
            for (int h = 0; h < NG; ++h) acc[h] = dd_mfma16<WF>(w[u], b[h], acc[h]);
            if (blk + 1 < NB) w[u] = dd_ldw(a.temporal, wp + (size_t)(s + U) * wstep);
            else if (!last_item) w[u] = dd_ldw(a.temporal, wn + (size_t)u * wstep);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        const bool mine = ((gi & 1) != 0) == hi_half;  // this half of the lanes keeps tile gi's sum
#pragma unroll
        for (int h = 0; h < NG; ++h) {
          const f32x4_t f = fold2(acc[h]);
          f32x4_t& d = sum[gi >> 1][h];
          const f32x4_t nv = half == 0 ? f : d + f;    // (hi+lo)(2p) + (hi+lo)(2p+1)
          d.x = mine ? nv.x : d.x, d.y = mine ? nv.y : d.y, d.z = mine ? nv.z : d.z, d.w = mine ? nv.w : d.w;
        }
      } else if (!last_item) {                         // a slot this wave does not have: hand the ring to the next item
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = dd_ldw(a.temporal, wn + (size_t)u * wstep);
      }
      if (half == NH - 1 && DD_TEXP(a, 8) && ((gi & 1) || gi == MAXG - 1)) store_set(gi >> 1);   // (timing experiment: the round-5 placement)
    }
    if (half + 1 < NH) {
      // (requesting the second slice's pieces BEFORE this barrier — their latency beside the slower waves' last tile — was measured: no gain,
      // 60 registers; tools/seq_lab.hip, profiles/r05_lab/)
      __syncthreads();                                 // every wave has finished reading slice 2 qs
      if (!DD_TEXP(a, 2)) stage(half + 1);
      __syncthreads();
    }
  }
  // Every partial sum is written HERE, after the wave's last weight request has been consumed: gfx950 counts loads and stores in one counter
  // (vmcnt) that retires in order, so a store issued in mid-stream makes every later weight piece wait for the store's acknowledgement.
  if (!DD_TEXP(a, 8)) {
#pragma unroll
    for (int st = 0; st < NSET; ++st) store_set(st);
  }
}

// The slice-resident GEMV for FP8 weight tiles (weight_format 1: OCP e4m3fn + per-row scales, BASELINE config 5).  One 1 KiB
// weight load = 16 rows x 64 k = TWO bf16 k-steps after an exact in-register expansion (fp8x16_to_bf16), which is done once per
// load and feeds all NG operand planes — the conversion VALU that bounds the 8-row fp8 kernel is amortised over the planes.
// K is cut as the 8-row fp8 kernel cuts it: slice q = 64-k steps q, q + 8, ... (its wave q), each step = bf16 k-steps 2 s and
// 2 s + 1 in that order, so every accumulator chain — and with the pairwise slice order the final sum — is the 8-row kernel's,
// bit for bit; the row scale is applied by the finishing kernel after the sum, as there.
// SPW2 = 64-k steps per slice (K / 512); CH = slices per workgroup (2: the pair is added here, half the partial sums);
// UW = weight loads in flight per wave (a divisor of SPW2).  A wave owns whole tiles g = j + G (wave + 8 i) over its slice(s).
// grid = (8 / CH) * G workgroups of 512 threads; dynamic LDS = CH * 2 * SPW2 * NG KiB.
template <int NG, int SPW2, int CH, int UW, int EPI_TAG = 0, int XPF = 0>
__global__ __launch_bounds__(512) void k_gemv_slices_fp8(SliceArgs a) {
  static_assert(SPW2 % UW == 0, "loads in flight must divide the slice");
  constexpr int NB = SPW2 / UW;                        // weight blocks per (tile, slice)
  constexpr int PIECES = CH * 2 * SPW2 * NG;           // operand pieces (1 KiB) of the workgroup
  constexpr int PW = (PIECES + 7) / 8;
  extern __shared__ __align__(16) u32x4_t xs[];        // [CH][2 * SPW2][NG][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NQ = 8 / CH;
  const int qs = blockIdx.x % NQ, j = blockIdx.x / NQ, q = qs * CH;
  const int S2 = a.S >> 1;
  const size_t xplane = (size_t)a.S * 64;
  const int n_tiles = a.n_groups;
  if ((int)blockIdx.x >= NQ * a.G) {                   // SliceArgs::rstd_wg: the workgroup behind the streaming ones — the rows' rstd
    if (a.ssq_in) dd_rows_rstd<NG>(a.ssq_in, a.ssq_n, a.ssq_ld, a.inv_k, a.eps, a.rstd_out);
    return;
  }
  if (!a.rstd_wg && a.ssq_in && blockIdx.x == 0) dd_rows_rstd<NG>(a.ssq_in, a.ssq_n, a.ssq_ld, a.inv_k, a.eps, a.rstd_out);
  int g = j + a.G * wave;
  const bool any = g < n_tiles;
  // block b (0 .. CH * NB - 1) of tile g: slice q + b / NB, 64-k steps (b % NB) * UW ...; consecutive steps of a slice are 8 apart
  auto wptr = [&](int tile, int b) -> const u32x4_t* {
    return a.W + ((size_t)tile * S2 + q + b / NB + (size_t)8 * (b % NB) * UW) * 64 + lane;
  };
  u32x4_t wc[UW], wn[UW];
  if (any) {
    const u32x4_t* p = wptr(g, 0);
#pragma unroll
    for (int u = 0; u < UW; ++u) wc[u] = dd_ldw(a.temporal, p + (size_t)u * 8 * 64);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    u32x4_t xv[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int p = wave + 8 * i, pc = p < PIECES ? p : 0;
      const int ch = pc / (2 * SPW2 * NG), r = pc % (2 * SPW2 * NG), t = r / NG, h = r % NG;
      xv[i] = a.xop[(size_t)(2 * (q + ch + 8 * (t >> 1)) + (t & 1)) * 64 + h * xplane + lane];
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int p = wave + 8 * i;
      if (p < PIECES) xs[(size_t)p * 64 + lane] = xv[i];
    }
  }
  __syncthreads();
  if (!any) return;
  auto fold = [&](f32x4_t v) -> f32x4_t {              // hi + lo column of a row (lanes c < 8 of 16)
    v.x += __shfl_down(v.x, 8);
    v.y += __shfl_down(v.y, 8);
    v.z += __shfl_down(v.z, 8);
    v.w += __shfl_down(v.w, 8);
    return v;
  };
  while (true) {
    const int gn = g + a.G * 8;
    const bool has_next = gn < n_tiles;
    f32x4_t acc[NG], sum[NG];
#pragma unroll
    for (int b = 0; b < CH * NB; ++b) {
      // request the next block (of this tile, or the first of the wave's next tile) before consuming this one
      if (DD_TEXP(a, 64)) {                            // (timing experiment: the weight stream stops after the first block — LDS reads + MFMAs alone)
#pragma unroll
        for (int u = 0; u < UW; ++u) wn[u] = wc[u];
      } else
      if (b + 1 < CH * NB || has_next) {
        const u32x4_t* p = b + 1 < CH * NB ? wptr(g, b + 1) : wptr(gn, 0);
#pragma unroll
        for (int u = 0; u < UW; ++u) wn[u] = dd_ldw(a.temporal, p + (size_t)u * 8 * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (b % NB == 0) {
#pragma unroll
        for (int h = 0; h < NG; ++h) acc[h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
      const int ch = b / NB, s0 = (b % NB) * UW;
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        u32x4_t k0, k1;
        fp8x16_to_bf16(wc[u], k0, k1);
        const u32x4_t* x0 = xs + ((size_t)(ch * 2 * SPW2 + 2 * (s0 + u)) * NG) * 64 + lane;
        if constexpr (NG >= 8 && XPF > 0) {
          // round 6: the operand fragments through a ring of XPF registers sets, XPF - 1 reads ahead of the MFMA that consumes them.  Left to
          // the compiler a fragment is requested one MFMA (16 cycles) before its use, an LDS read takes several times that, and the
          // nine-plane loop runs at the LDS latency: 18 fragments per 1 KiB of weights.  Fragment i = (plane i >> 1, half i & 1): the
          // MFMA order per accumulator is unchanged.
          u32x4_t fr[XPF];
#pragma unroll
          for (int i = 0; i < XPF - 1; ++i) fr[i] = x0[(size_t)(((i & 1) ? NG : 0) + (i >> 1)) * 64];
          if (DD_TEXP(a, 16)) {                        // (timing experiment, tools library: no further fragment reads — every MFMA multiplies fragment 0)
#pragma unroll
            for (int i = 0; i < 2 * NG; ++i) acc[i >> 1] = dd_mfma16<0>((i & 1) ? k1 : k0, fr[0], acc[i >> 1]);
          } else if (DD_TEXP(a, 32)) {                 // (timing experiment: all 18 fragment reads, but only the first plane's two MFMAs)
#pragma unroll
            for (int i = 0; i < 2 * NG; ++i) {
              if (i + XPF - 1 < 2 * NG) fr[(i + XPF - 1) % XPF] = x0[(size_t)((((i + XPF - 1) & 1) ? NG : 0) + ((i + XPF - 1) >> 1)) * 64];
              if (i < 2) acc[0] = dd_mfma16<0>((i & 1) ? k1 : k0, fr[i % XPF], acc[0]);
              else acc[i >> 1].x += __builtin_bit_cast(float, fr[i % XPF].x);
              __builtin_amdgcn_sched_barrier(0);
            }
          } else
#pragma unroll
          for (int i = 0; i < 2 * NG; ++i) {
            if (i + XPF - 1 < 2 * NG) fr[(i + XPF - 1) % XPF] = x0[(size_t)((((i + XPF - 1) & 1) ? NG : 0) + ((i + XPF - 1) >> 1)) * 64];
            acc[i >> 1] = dd_mfma16<0>((i & 1) ? k1 : k0, fr[i % XPF], acc[i >> 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
        for (int h = 0; h < NG; ++h) {
          acc[h] = dd_mfma16<0>(k0, x0[(size_t)h * 64], acc[h]);
          acc[h] = dd_mfma16<0>(k1, x0[(size_t)(NG + h) * 64], acc[h]);
        }
        }
      }
      if (b % NB == NB - 1) {                          // the slice's chain is complete
#pragma unroll
        for (int h = 0; h < NG; ++h) {
          f32x4_t f = fold(acc[h]);
          if (ch == 0) sum[h] = f;
          else sum[h] = sum[h] + f;                    // (hi+lo)(2p) + (hi+lo)(2p+1)
        }
      }
#pragma unroll
      for (int u = 0; u < UW; ++u) wc[u] = wn[u];
    }
    if ((lane & 8) == 0) {
      const int l32 = (lane >> 4) * 8 + (lane & 7);
#pragma unroll
      for (int h = 0; h < NG; ++h) *(f32x4_t*)&a.part[((((size_t)qs * n_tiles + g) * NG + h) << 7) + l32 * 4] = sum[h];
    }
    if (!has_next) break;
    g = gn;
  }
}

// (Round 6, measured and NOT kept: the same kernel with TWO tiles per wave in lock-step, every operand fragment feeding both tiles' MFMAs —
// half the LDS reads per weight byte.  Bit-identical, 255 VGPRs, and slower: Mistral-7B's gate/up at 72 rows 39.0 -> 42.2 us alone, the 64-lane
// step 44.3 -> 45.6 ms.  The nine-plane fp8 kernel is bound neither by the LDS bandwidth nor — XPF, above: 40.5 -> 38.7 us — much by its latency;
// profiles/r06_lab/fp8_two_tiles_per_wave.log, fp8_fragment_ring.log.)
// The same for LONG K (down_proj of Mistral-7B: K = 14336, 28 steps of 64 k per slice) at four / eight / nine operand planes: a slice of
// all planes does not fit the LDS (28 x 2 x NG KiB), so it is staged in chunks of CS2 steps — the next chunk's operand pieces are
// requested into registers at the start of the current chunk and committed between two barriers, the weight queue runs on across the
// chunk boundary (as k_gemv_slices does for bf16's long K).  One tile per wave (g = j + G wave), single slices: the chain of slice q
// is the 8-row fp8 kernel's wave q (64-k steps q, q + 8, ...; both halves of a step in order), the finishing kernel adds the eight
// slices pairwise and applies the row scale — bit for bit the 8-row kernel.
// grid = 8 * G workgroups of 512 threads; dynamic LDS = 2 * CS2 * NG KiB; UW divides CS2.
template <int NG, int SPW2, int CS2, int UW, int EPI_TAG = 0>
__global__ __launch_bounds__(512) void k_gemv_slices_fp8c(SliceArgs a) {
  static_assert(SPW2 % CS2 == 0 && CS2 % UW == 0, "chunks of whole weight blocks");
  constexpr int NCH = SPW2 / CS2, NBC = CS2 / UW;      // chunks per slice, weight blocks per chunk
  constexpr int PIECES = 2 * CS2 * NG;                 // operand pieces (1 KiB) per chunk
  constexpr int PW = (PIECES + 7) / 8;
  extern __shared__ __align__(16) u32x4_t xs[];        // [2 * CS2][NG][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int S2 = a.S >> 1;
  const size_t xplane = (size_t)a.S * 64;
  const int n_tiles = a.n_groups;
  if ((int)blockIdx.x >= 8 * a.G) {                    // SliceArgs::rstd_wg: the workgroup behind the streaming ones — the rows' rstd
    if (a.ssq_in) dd_rows_rstd<NG>(a.ssq_in, a.ssq_n, a.ssq_ld, a.inv_k, a.eps, a.rstd_out);
    return;
  }
  if (!a.rstd_wg && a.ssq_in && blockIdx.x == 0) dd_rows_rstd<NG>(a.ssq_in, a.ssq_n, a.ssq_ld, a.inv_k, a.eps, a.rstd_out);
  const int g = j + a.G * wave;
  const bool live = g < n_tiles;
  const u32x4_t* const wp = a.W + ((size_t)(live ? g : 0) * S2 + q) * 64 + lane;      // step s of the slice: wp + 8 s tiles
  u32x4_t xv[PW];
  auto stage_issue = [&](int k) {                      // chunk k: steps k CS2 .. k CS2 + CS2 - 1, piece = (32-k half-step t, plane h)
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int p = wave + 8 * i, pc = p < PIECES ? p : 0;
      const int t = pc / NG, h = pc % NG;
      xv[i] = a.xop[(size_t)(2 * (q + 8 * (k * CS2 + (t >> 1))) + (t & 1)) * 64 + h * xplane + lane];
    }
  };
  auto stage_commit = [&]() {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int p = wave + 8 * i;
      if (p < PIECES) xs[(size_t)p * 64 + lane] = xv[i];
    }
  };
  u32x4_t wc[UW], wn[UW];
#pragma unroll
  for (int u = 0; u < UW; ++u) wc[u] = dd_ldw(a.temporal, wp + (size_t)u * 8 * 64);
  __builtin_amdgcn_sched_barrier(0);
  stage_issue(0);
  stage_commit();
  __syncthreads();
  f32x4_t acc[NG];
#pragma unroll
  for (int h = 0; h < NG; ++h) acc[h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    if (k + 1 < NCH) stage_issue(k + 1);               // requested ahead of this chunk's later weight requests
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
      const int s0 = k * CS2 + b * UW;                 // first step of this block
      if (s0 + UW < SPW2) {
#pragma unroll
        for (int u = 0; u < UW; ++u) wn[u] = dd_ldw(a.temporal, wp + (size_t)(s0 + UW + u) * 8 * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        u32x4_t k0, k1;
        fp8x16_to_bf16(wc[u], k0, k1);
        const u32x4_t* x0 = xs + ((size_t)(2 * (b * UW + u)) * NG) * 64 + lane;
#pragma unroll
        for (int h = 0; h < NG; ++h) {
          acc[h] = dd_mfma16<0>(k0, x0[(size_t)h * 64], acc[h]);
          acc[h] = dd_mfma16<0>(k1, x0[(size_t)(NG + h) * 64], acc[h]);
        }
      }
#pragma unroll
      for (int u = 0; u < UW; ++u) wc[u] = wn[u];
    }
    if (k + 1 < NCH) {
      __syncthreads();                                 // every wave has read this chunk
      stage_commit();
      __syncthreads();
    }
  }
#pragma unroll
  for (int h = 0; h < NG; ++h) {                       // hi + lo column of a row (lanes c < 8 of 16); every lane takes part in the shuffle
    f32x4_t v = acc[h];
    v.x += __shfl_down(v.x, 8);
    v.y += __shfl_down(v.y, 8);
    v.z += __shfl_down(v.z, 8);
    v.w += __shfl_down(v.w, 8);
    acc[h] = v;
  }
  if (live && (lane & 8) == 0) {
    const int l32 = (lane >> 4) * 8 + (lane & 7);
#pragma unroll
    for (int h = 0; h < NG; ++h) *(f32x4_t*)&a.part[((((size_t)q * n_tiles + g) * NG + h) << 7) + l32 * 4] = acc[h];
  }
}


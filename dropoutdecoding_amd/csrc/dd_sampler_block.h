// The mask sampler as it was until round 5: ONE 1,024-THREAD WORKGROUP per sequence, the generator block, the uniforms and the running mask in
// LDS, every phase fenced by s_barrier.  libdropdec_tools.so only (dd_dropout.hip includes this file under DD_KEEP_SCRATCH_SAMPLER): the product
// samples with one wave per sequence and no barrier at all (dd_dropout.hip "one wave per sequence"), because this form is the victim of the
// co-residency fault of DESIGN.md 3e — beside the kernels of a rider step its sixteen waves fall out of step across s_barrier about once in
// 50,000 launches (round 5's dumps: stores of two of the four regenerating waves missing for two regenerations in a row, values of the NEXT
// generation visible before a regeneration's entry barrier).  Kept for the unit reproducer (tools/sampler_repro.py):
//   * k_sample_masks_lanes_block    the product kernel of round 4 (no private scratch)
//   * k_sample_masks_lanes_scratch  round 3's form (616 bytes of private scratch per lane)
//   * k_sample_masks_lanes_dbg      the checking form: every thread's word of the generator block mirrored in a register (mt_dbg_check)
#pragma once
#define MASK_THREADS 1024

// Regenerate the 624 words held in LDS. The sequential recurrence new[i] = f(old[i], old[i+1], x[i+397 mod 624])
// only reaches back 227 places, so it runs as three block-parallel sweeps plus the last word.
__device__ void mt_twist_block(uint32_t* mt) {
  const int segs[4][2] = {{0, 227}, {227, 454}, {454, 623}, {623, 624}};
  for (int s = 0; s < 4; ++s) {
    __syncthreads();
    uint32_t nv = 0;
    int i = segs[s][0] + (int)threadIdx.x;
    bool act = i < segs[s][1];
    if (act) nv = mt_mix(mt[i], mt[(i + 1) % MT_N], mt[(i + MT_M) % MT_N]);
    __syncthreads();
    if (act) mt[i] = nv;
  }
  __syncthreads();
}


// Debug context of the tools library's sampler (k_sample_masks_lanes_dbg, DD_KEEP_SCRATCH_SAMPLER builds only): every thread
// below 624 keeps, in a register, the word of the generator block it owns as of the block's last write (launch-start load / regeneration);
// mt_dbg_check compares the block in LDS with those registers and, on the first difference of a run, dumps what it sees to global memory.
// The product kernels instantiate DBG = false: no context, no checks, the code they always had.
struct MtDbg {
  uint32_t shadow;
  uint32_t* buf;               // [0] claim, [1..15] header, [16..] dump (see mt_dbg_check)
  const uint32_t* src;         // the launch-start state in global memory
  const uint32_t* smem32;      // the workgroup's dynamic LDS
  int smem_words, member, twists, checks;
  uint32_t tag;
};
#define MT_DBG_DUMP_WORDS (16 + 5 * 640)
__device__ __noinline__ void mt_dbg_check(const uint32_t* mt, const int* idx_sh, MtDbg& D, int where) {
  __shared__ int claim;
  const int tid = threadIdx.x;
  const uint32_t seen = tid < MT_N ? mt[tid] : 0u;
  const bool bad = tid < MT_N && seen != D.shadow;
  D.checks++;
  if (__syncthreads_or(bad ? 1 : 0)) {
    if (tid == 0) {
      claim = atomicCAS(&D.buf[0], 0u, 1u) == 0u ? 1 : 0;
      atomicAdd(&D.buf[1], 1u);                      // failed checks of the run
    }
    __syncthreads();
    if (claim) {
      uint32_t* b = D.buf;
      if (tid == 0) {
        b[2] = blockIdx.x, b[3] = (uint32_t)D.member, b[4] = (uint32_t)D.twists, b[5] = (uint32_t)where, b[6] = D.tag, b[7] = (uint32_t)*idx_sh;
        b[8] = (uint32_t)D.checks, b[9] = (uint32_t)D.smem_words;
        b[10] = (uint32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_ID: wave / SIMD / CU / SE ids of wave 0
        b[11] = (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));    // XCC_ID
        b[12] = (uint32_t)__builtin_amdgcn_s_getreg((6 << 0) | (0 << 6) | (31 << 11));     // LDS_ALLOC: base / size of the allocation
      }
      if (tid < MT_N) {
        b[16 + tid] = D.shadow;                      // what the owner thread last saw written
        b[16 + 640 + tid] = seen;                    // what the block holds now
        b[16 + 3 * 640 + tid] = D.src[tid];          // the state this launch loaded
      }
      __syncthreads();
      for (int spin = 0; spin < 64; ++spin) __builtin_amdgcn_s_sleep(64);
      __syncthreads();
      if (tid < MT_N) b[16 + 2 * 640 + tid] = mt[tid];                                      // ... and a few microseconds later
      for (int i = tid; i < D.smem_words; i += blockDim.x) b[MT_DBG_DUMP_WORDS + i] = D.smem32[i];
      __threadfence();
    }
    __syncthreads();
  }
  if (tid < MT_N) D.shadow = seen;                   // one report per corruption
}
#define MT_DBG(where) do { if constexpr (DBG) mt_dbg_check(mt, idx_sh, D, where); } while (0)
#define MT_DBG_REFRESH() do { if constexpr (DBG) { if (threadIdx.x < MT_N) D.shadow = mt[threadIdx.x]; D.twists++; } } while (0)

// Fill out[0..n) (LDS or global) with the next n uniforms; generator block in LDS, *idx block-uniform.
template <bool DBG>
__device__ void mt_fill_block(uint32_t* mt, int* idx_sh, float* out, int n, MtDbg& D) {
  __syncthreads();
  if ((uint32_t)*idx_sh == PHILOX_TAG) {   // one rand_like: element t = subsequence t at the current offset; offset += 4
    const unsigned long long off = (unsigned long long)mt[2] | ((unsigned long long)mt[3] << 32);
    const uint32_t k0 = mt[0], k1 = mt[1];
    for (int t = threadIdx.x; t < n; t += blockDim.x) out[t] = philox_uniform(philox_first(k0, k1, off >> 2, (unsigned long long)t));
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long o2 = off + 4ull;
      mt[2] = (uint32_t)o2, mt[3] = (uint32_t)(o2 >> 32);
    }
    __syncthreads();
    return;
  }
  int pos = 0;
  while (pos < n) {
    __syncthreads();
    int idx = *idx_sh;
    if (idx >= MT_N) {
      MT_DBG(1);
      mt_twist_block(mt);
      MT_DBG_REFRESH();
      idx = 0;
    }
    int take = min(n - pos, MT_N - idx);
    for (int t = threadIdx.x; t < take; t += blockDim.x) out[pos + t] = mt_temper_uniform(mt[idx + t]);
    __syncthreads();
    if (threadIdx.x == 0) *idx_sh = idx + take;
    pos += take;
  }
  __syncthreads();
}

__device__ float block_min_max(const float* e, int L, bool want_max, float* sh) {
  float v = want_max ? -INFINITY : INFINITY;
  for (int l = threadIdx.x; l < L; l += blockDim.x) v = want_max ? fmaxf(v, e[l]) : fminf(v, e[l]);
  if (!want_max) v = -v;
  v = block_max_f(v, sh);
  return want_max ? v : -v;
}

// exclusive prefix of `flag` over the block in thread order; returns prefix, *total gets the block total
__device__ int block_exclusive_scan_flag(bool flag, int* sh /*>=17*/, int* total) {
  unsigned long long b = __ballot(flag);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int within = __popcll(b & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) sh[w] = __popcll(b);
  __syncthreads();
  int off = 0, tot = 0;
  for (int j = 0; j < nw; ++j) {
    if (j < w) off += sh[j];
    tot += sh[j];
  }
  *total = tot;
  return off + within;
}

template <bool DBG = false>
__device__ __forceinline__ void sample_masks_body(const MaskConst C, const MaskSeq P, unsigned char* smem, uint32_t* dbg_buf = nullptr, uint32_t dbg_tag = 0,
                                                  int dbg_smem_bytes = 0) {
  float* e = (float*)smem;                       // [Lp]
  float* u = e + MASK_MAX_L;                     // [Lp] uniforms of the current member, or sort buffer
  uint8_t* running = (uint8_t*)(u + MASK_MAX_L);  // [L]
  uint32_t* mt = (uint32_t*)(running + MASK_MAX_L);
  __shared__ float sh_f[16];
  __shared__ int sh_i[17];
  __shared__ int idx_sh_;
  int* const idx_sh = &idx_sh_;
  const int L = P.L, tid = threadIdx.x;
  MtDbg D;
  if constexpr (DBG) {
    D.buf = dbg_buf, D.src = P.rng_in ? P.rng_in : P.rng_state, D.smem32 = (const uint32_t*)smem, D.smem_words = dbg_smem_bytes / 4;
    D.member = -1, D.twists = 0, D.checks = 0, D.tag = dbg_tag, D.shadow = 0;
  }

  for (int l = tid; l < L; l += MASK_THREADS) {
    e[l] = P.epi[l];
    running[l] = 0;
  }
  if (C.rng_mode == DD_RNG_MT19937) {
    const uint32_t* src = P.rng_in ? P.rng_in : P.rng_state;
    for (int i = tid; i < MT_N; i += MASK_THREADS) mt[i] = src[i];
    if (tid == 0) *idx_sh = (int)src[MT_N];
  }
  __syncthreads();
  if constexpr (DBG) {
    if (tid < MT_N) D.shadow = mt[tid];
  }
  float lo = 0.f, hi = 0.f;
  if (C.mode != DD_MASK_IBLIP_QUANTILE) {
    lo = block_min_max(e, L, false, sh_f);  // torch.quantile(e, 0) == min   (llava.py:641)
    hi = block_min_max(e, L, true, sh_f);   // torch.quantile(e, 1) == max   (llava.py:642)
  } else {
    // ascending bitonic sort of e into u (padded with +inf): torch.quantile sorts first
    int Lp = 1;
    while (Lp < L) Lp <<= 1;
    for (int l = tid; l < Lp; l += MASK_THREADS) u[l] = l < L ? e[l] : INFINITY;
    __syncthreads();
    for (int k2 = 2; k2 <= Lp; k2 <<= 1) {
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < Lp; i += MASK_THREADS) {
          int ixj = i ^ j;
          if (ixj > i) {
            float a = u[i], b = u[ixj];
            bool up = ((i & k2) == 0);
            if ((a > b) == up) {
              u[i] = b;
              u[ixj] = a;
            }
          }
        }
        __syncthreads();
      }
    }
  }

  for (int k = 0; k < C.K; ++k) {   // DD_MASK_IBLIP_KL runs the NEXT_RESET rule: its keep flags come from dd_kl_keep instead of the overlap
    if (C.mode != DD_MASK_LLAVA_CUMULATIVE && C.mode != DD_MASK_LLAVA_CUMULATIVE_NO_OVERLAP) {  // reset: llavanext.py:546, instructblip.py:121
      for (int l = tid; l < L; l += MASK_THREADS) running[l] = 0;
    }
    float thr = 0.f;
    if (C.mode == DD_MASK_IBLIP_QUANTILE) {
      // torch.quantile(e, q) with linear interpolation, fp32: rank = q*(n-1); lerp(sorted[floor], sorted[ceil], frac)
      // ATen's lerp: weight < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w), contracted to one fma on the CPU build.
      float rank = C.q[k] * (float)(L - 1);
      float fl = floorf(rank);
      int i0 = (int)fl, i1 = (int)ceilf(rank);
      float w = rank - fl;
      float a = u[i0], b = u[i1], diff = b - a;
      thr = (w < 0.5f) ? fmaf(w, diff, a) : fmaf(-diff, 1.0f - w, b);
    } else if (C.rng_mode == DD_RNG_MT19937) {
      if constexpr (DBG) D.member = k;
      MT_DBG(3);
      mt_fill_block<DBG>(mt, idx_sh, u, L, D);  // one rand_like(e) per member (llava.py:650)
      MT_DBG(4);
    }
    __syncthreads();
    const float scale = C.scale[k];
    const float range = __fsub_rn(hi, lo);
    const bool no_overlap = C.mode == DD_MASK_NEXT_NO_OVERLAP || C.mode == DD_MASK_LLAVA_CUMULATIVE_NO_OVERLAP;
    int total = 0, cnt = 0;
    for (int base = 0; base < L; base += MASK_THREADS) {
      int l = base + tid;
      bool dropped = false;
      if (l < L) {
        bool d;
        if (C.mode == DD_MASK_IBLIP_QUANTILE) {
          d = e[l] >= thr;  // instructblip.py:453
        } else {
          float r = (C.rng_mode == DD_RNG_MT19937) ? u[l] : P.uniforms[(size_t)k * L + l];
          // p = 0.1 + (mprob-0.1)*(clamp(e,lo,hi)-lo)/(hi-lo), every step rounded to fp32 (llava.py:646-647)
          float c = fminf(fmaxf(e[l], lo), hi);
          float p = __fadd_rn(0.1f, __fdiv_rn(__fmul_rn(scale, __fsub_rn(c, lo)), range));
          d = r < p;  // llava.py:653 (NaN p when hi == lo: nothing dropped)
        }
        uint8_t run = running[l] | (d ? 1 : 0);                      // llava.py:654-657, in place
        if (!no_overlap && P.keep && P.keep[l]) run = 0;  // llava.py:660 keep-restore (keep null: an empty keep set)
        running[l] = run;
        P.drop[(size_t)k * L + l] = run;
        dropped = run != 0;
      }
      int off = block_exclusive_scan_flag(dropped, sh_i, &total);
      if (P.idx && dropped) P.idx[(size_t)k * L + cnt + off] = l;
      cnt += total;
    }
    if (P.idx)
      for (int l = cnt + tid; l < L; l += MASK_THREADS) P.idx[(size_t)k * L + l] = -1;
    if (tid == 0) P.n_drop[k] = cnt;  // masked_numbers (llava.py:661-662)
    __syncthreads();
    if (P.drop_bits) {
      for (int l = tid; l < L; l += MASK_THREADS) {
        uint8_t* bp = P.drop_bits + (size_t)(k >> 3) * L + l;
        uint8_t cur = (k & 7) ? *bp : 0;
        *bp = cur | (uint8_t)((running[l] ? 1 : 0) << (k & 7));
      }
    }
    __syncthreads();
    if (C.rng_mode == DD_RNG_MT19937) MT_DBG(5);
  }
  if (C.rng_mode == DD_RNG_MT19937 && !P.rng_in) {
    for (int i = tid; i < MT_N; i += MASK_THREADS) P.rng_state[i] = mt[i];
    if (tid == 0) P.rng_state[MT_N] = (uint32_t)*idx_sh;
  }
}

__global__ __launch_bounds__(MASK_THREADS) void k_sample_masks_lanes_block(MaskLanes M) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int m = blockIdx.x;
  if (M.gate[m] && *M.gate[m]) return;    // this sequence finished (EOS): its stream and masks stay as they are
  const int L = M.L[m], k = M.k_top;
  const int tok = M.argmax[m][0];
  for (int l = threadIdx.x; l < L; l += MASK_THREADS) {
    bool hit = false;
    for (int j = 0; j < k; ++j) hit |= (M.topk[m][(size_t)l * k + j] == tok);
    M.keep[m][l] = hit ? 1 : 0;
  }
  __shared__ float sh_tab[128];
  {  // scale[] and q[] are adjacent in the kernel arguments: 128 floats read straight from the kernarg segment
    const float __attribute__((address_space(4)))* tab = (const float __attribute__((address_space(4)))*)(
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MaskLanes, common.scale));
    if (threadIdx.x < 128) sh_tab[threadIdx.x] = tab[threadIdx.x];
  }
  __syncthreads();                   // keep[] is read back by this same workgroup
  // (uniforms / idx / rng_in are null for lanes — taken from the zeroed common block, not written as literal nullptr: with the
  // constants folded into the inlined body hipcc 7.2's instcombine dies on the dead injected-uniforms load)
  const MaskSeq S = {M.epi[m], L, M.keep[m], M.common.uniforms, M.rng_state[m], M.drop[m], M.n_drop[m], M.common.idx, M.drop_bits[m], M.common.rng_in};
  const MaskConst C = {M.common.K, M.common.mode, M.common.rng_mode, sh_tab, sh_tab + 64};
  sample_masks_body(C, S, smem);
}
// The form this kernel had until round 4, kept in libdropdec_tools.so ONLY (build.py compiles this file a second time with the macro)
// for the A/B of tools/stress_lanes.py and the unit reproducer: the by-value copy below is indexed by the member loop, so the compiler
// keeps it in private scratch (616 bytes per lane).
__global__ __launch_bounds__(MASK_THREADS) void k_sample_masks_lanes_scratch(MaskLanes M) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int m = blockIdx.x;
  if (M.gate[m] && *M.gate[m]) return;
  const int L = M.L[m], k = M.k_top;
  const int tok = M.argmax[m][0];
  for (int l = threadIdx.x; l < L; l += MASK_THREADS) {
    bool hit = false;
    for (int j = 0; j < k; ++j) hit |= (M.topk[m][(size_t)l * k + j] == tok);
    M.keep[m][l] = hit ? 1 : 0;
  }
  __syncthreads();
  MaskParams P = M.common;           // (uniforms, idx, rng_in: null in the common block)
  P.epi = M.epi[m], P.L = L, P.keep = M.keep[m], P.rng_state = M.rng_state[m];
  P.drop = M.drop[m], P.n_drop = M.n_drop[m], P.drop_bits = M.drop_bits[m];
  const MaskSeq S = {P.epi, P.L, P.keep, P.uniforms, P.rng_state, P.drop, P.n_drop, P.idx, P.drop_bits, P.rng_in};
  const MaskConst C = {P.K, P.mode, P.rng_mode, P.scale, P.q};     // tables in the private copy
  sample_masks_body(C, S, smem);
}
// The product kernel with the generator-block checks of mt_dbg_check compiled in (tools key 34 = 2; buffer: dd_dropout_set_sampler_dbg).
__global__ __launch_bounds__(MASK_THREADS) void k_sample_masks_lanes_dbg(MaskLanes M, uint32_t* dbg_buf, uint32_t tag, int smem_bytes) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int m = blockIdx.x;
  if (M.gate[m] && *M.gate[m]) return;
  const int L = M.L[m], k = M.k_top;
  const int tok = M.argmax[m][0];
  for (int l = threadIdx.x; l < L; l += MASK_THREADS) {
    bool hit = false;
    for (int j = 0; j < k; ++j) hit |= (M.topk[m][(size_t)l * k + j] == tok);
    M.keep[m][l] = hit ? 1 : 0;
  }
  __shared__ float sh_tab[128];
  {
    const float __attribute__((address_space(4)))* tab = (const float __attribute__((address_space(4)))*)(
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MaskLanes, common.scale));
    if (threadIdx.x < 128) sh_tab[threadIdx.x] = tab[threadIdx.x];
  }
  __syncthreads();
  const MaskSeq S = {M.epi[m], L, M.keep[m], M.common.uniforms, M.rng_state[m], M.drop[m], M.n_drop[m], M.common.idx, M.drop_bits[m], M.common.rng_in};
  const MaskConst C = {M.common.K, M.common.mode, M.common.rng_mode, sh_tab, sh_tab + 64};
  sample_masks_body<true>(C, S, smem, dbg_buf, tag, smem_bytes);
}
static int g_lanes_sampler_scratch = 0;
static uint32_t* g_sampler_dbg_buf = nullptr;
static uint32_t g_sampler_dbg_tag = 0;
void dd_dropout_set_lanes_sampler_scratch(int on) { g_lanes_sampler_scratch = on; }
void dd_dropout_set_sampler_dbg(uint32_t* buf) { g_sampler_dbg_buf = buf, g_sampler_dbg_tag = 0; }
uint32_t dd_dropout_sampler_dbg_tag() { return g_sampler_dbg_tag; }
size_t dd_dropout_sampler_dbg_words() { return (size_t)MT_DBG_DUMP_WORDS + (size_t)156 * 1024 / 4; }

// fuse_lab: the nine-plane gate/up kernel with its finishing step inside the launch — a measurement tool, not product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops tools/fuse_lab.hip -o tools/fuse_lab
// The 64-lane step is short of bytes per second, and 22 % of its memory traffic is partial sums written by the slice kernels and read back by
// the finishing kernels.  Idea: put the four slice-pair workgroups of a tile group on ONE XCD (workgroups go to XCDs round-robin:
// blockIdx.x % 8) and let the tile's LAST ARRIVER add the four sums — from that XCD's L2, where the other three have just been written —
// in the fixed order ((p0 + p1) + p2) + p3.  No spinning (nobody waits for anybody), no device-scope release (which on this part writes the
// whole L2 back): stores -> s_waitcnt vmcnt(0) -> a counter atomic in the same L2 -> sc1 loads.
// Variants: A product block order, stream + separate finishing kernel;  B XCD order, stream + finishing kernel (block b finishes tile b: the same
// XCD as the tile's partial sums);  C XCD order, finished inside the launch.  Every variant's finished sums are compared with A's, bit for bit.
//   tools/fuse_lab            all variants, timings
//   tools/fuse_lab A|B|C|D|E  one variant only (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../dropoutdecoding_amd/csrc/dd_gemv_slices.h"
void dd_set_error(const char*, ...) {}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct FusedArgs {
  SliceArgs a;
  unsigned* cnt;     // [n_tiles] arrivals per tile (zero before the first launch; the finishing wave resets its tile's counter)
  float* y;          // [n_tiles][NG][128] finished sums
  unsigned* dbg;     // [0] workgroups whose XCC_ID != blockIdx.x % 8, [1] finishing waves
  int fused;         // 0: partial sums only (a separate finishing kernel follows); the tile's last arriver finishes it, reading the other pairs' sums with 1: sc1 loads, 2: sc0 loads, 3: plain loads
  int xcd_map;       // 1: the four slice pairs of a tile group on one XCD (blockIdx.x % 8), 0: the product's order (qs = blockIdx.x & 3)
};
// The product kernel's text (csrc/dd_gemv_slices.h k_gemv_slices_seq) with two changes: the block -> (group, pair) map, and the arrival / finish
// step behind a set's stores.
template <int NG, int U, int SPW, int MAXG, int WF = 0, int EPI_TAG = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_seq_fused(FusedArgs fa) {
  const SliceArgs& a = fa.a;
  static_assert(SPW % U == 0, "ring depth must divide the slice");
  constexpr int PW = (SPW * NG + 7) / 8;               // operand pieces (1 KiB) per wave and slice
  constexpr int NB = SPW / U;
  constexpr int NSET = (MAXG + 1) / 2;                 // register sets of folded sums: two tiles each
  extern __shared__ __align__(16) u32x4_t xs[];        // [SPW][NG][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool hi_half = (lane & 8) != 0;
  int qs = blockIdx.x & 3, j = blockIdx.x >> 2;
  if (fa.xcd_map && (int)blockIdx.x < 4 * a.G) {       // blockIdx.x = xcd + 8 * (4 * jl + qs): the four pairs of group j = 8 jl + xcd share an XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    qs = slot & 3, j = (slot >> 2) * 8 + xcd;
    if (threadIdx.x == 0) {
      const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF;
      if ((int)xcc != xcd) atomicAdd(&fa.dbg[0], 1u);
    }
  }
  const size_t xplane = (size_t)a.S * 64;
  const int n_tiles = a.n_groups;
  if ((int)blockIdx.x >= 4 * a.G) {                    // SliceArgs::rstd_wg: the workgroup behind the streaming ones — the rows' rstd
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
    return a.W + ((size_t)g * a.S + 2 * qs + half) * 64 + lane;
  };
  auto stage = [&](int half) {                         // operand slice 2 qs + half -> LDS (all waves)
    u32x4_t xv[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      int p = wave + 8 * i;
      int pc = p < SPW * NG ? p : 0;
      xv[i] = a.xop[(size_t)(2 * qs + half + 8 * (pc / NG)) * 64 + (pc % NG) * xplane + lane];
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
    if (my_gi < MAXG && my_gi < ng && !(a.temporal & 4)) {      // (temporal bit 2: timing experiment — no partial sums written)
      const int l32 = (lane >> 4) * 8 + (lane & 7);
#pragma unroll
      for (int h = 0; h < NG; ++h) *(f32x4_t*)&a.part[((((size_t)qs * n_tiles + my_g) * NG + h) << 7) + l32 * 4] = sum[st][h];
    }
    if (fa.fused) {
      // arrival: the stores above are in this XCD's L2 once vmcnt is 0 (the L1 writes through); then one lane per tile counts the tile's arrivals
      // (an atomic in the same L2).  The wave that counts the fourth arrival reads the other three pairs' sums from that L2 and finishes the tile.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const bool live = my_gi < MAXG && my_gi < ng;
      unsigned old = 0;
      if (live && (lane & 7) == 0 && (lane >> 4) == 0) old = atomicAdd(&fa.cnt[my_g], 1u);      // lanes 0 (tile 2 st) and 8 (tile 2 st + 1)
      old = __shfl(old, hi_half ? 8 : 0);
      if (live && old == 3) {
        const int l32 = (lane >> 4) * 8 + (lane & 7);
        f32x4_t tot[NG];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          f32x4_t v[NG];
          if (p != qs) {
#pragma unroll
            for (int h = 0; h < NG; ++h) {
              const float* src = &a.part[((((size_t)p * n_tiles + my_g) * NG + h) << 7) + l32 * 4];
              if (fa.fused == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[h]) : "v"(src) : "memory");
              else if (fa.fused == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v[h]) : "v"(src) : "memory");
              else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[h]) : "v"(src) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
#pragma unroll
          for (int h = 0; h < NG; ++h) {
            const f32x4_t x = p == qs ? sum[st][h] : v[h];
            tot[h] = p == 0 ? x : tot[h] + x;          // ((p0 + p1) + p2) + p3 whoever finishes
          }
        }
#pragma unroll
        for (int h = 0; h < NG; ++h) *(f32x4_t*)&fa.y[(((size_t)my_g * NG + h) << 7) + l32 * 4] = tot[h];
        if ((lane & 7) == 0 && (lane >> 4) == 0) {
          fa.cnt[my_g] = 0;
          atomicAdd(&fa.dbg[1], 1u);
        }
      }
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
  if (!(a.temporal & 2)) stage(0);
  __syncthreads();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
      const int item = half * MAXG + gi;
      const bool live = gi < ng;                       // wave-uniform
      const bool last_item = item == 2 * MAXG - 1;
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
      if (half == 1 && (a.temporal & 8) && ((gi & 1) || gi == MAXG - 1)) store_set(gi >> 1);   // (timing experiment: the round-5 placement)
    }
    if (half == 0) {
      // (requesting the second slice's pieces BEFORE this barrier — their latency beside the slower waves' last tile — was measured: no gain,
      // 60 registers; tools/seq_lab.hip, profiles/r05_lab/)
      __syncthreads();                                 // every wave has finished reading slice 2 qs
      if (!(a.temporal & 2)) stage(1);
      __syncthreads();
    }
  }
  // Every partial sum is written HERE, after the wave's last weight request has been consumed: gfx950 counts loads and stores in one counter
  // (vmcnt) that retires in order, so a store issued in mid-stream makes every later weight piece wait for the store's acknowledgement.
  if (!(a.temporal & 8)) {
#pragma unroll
    for (int st = 0; st < NSET; ++st) store_set(st);
  }
}

// the separate finishing step: block b = tile b, thread = (plane, 16-byte column group): ((p0 + p1) + p2) + p3
template <int NG>
__global__ __launch_bounds__(32 * NG) void k_finish_simple(const float* __restrict__ part, float* __restrict__ y, int n_tiles) {
  const int g = blockIdx.x, h = threadIdx.x >> 5, l32 = threadIdx.x & 31;
  f32x4_t v[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) v[p] = *(const f32x4_t*)&part[((((size_t)p * n_tiles + g) * NG + h) << 7) + l32 * 4];
  *(f32x4_t*)&y[(((size_t)g * NG + h) << 7) + l32 * 4] = ((v[0] + v[1]) + v[2]) + v[3];
}

__global__ void k_fill(uint32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u;
    h ^= h >> 15, h *= 2246822519u, h ^= h >> 13, h *= 3266489917u, h ^= h >> 16;
    p[i] = (h & 0x807f807fu) | 0x3c003c00u | ((h >> 3) & 0x01800180u);     // two bf16: random sign and mantissa, four exponents
  }
}

static const int NL = 8, NG = 9;
static u32x4_t *g_w, *g_x;
static float *g_part, *g_y, *g_ssq, *g_rstd;
static unsigned *g_cnt, *g_dbg;
static size_t g_wstride;

// variant: 'A' product order + finishing kernel, 'B' XCD order + finishing kernel, 'C' XCD order, finished inside the launch
static void run(char variant, int n_tiles, int G, std::vector<float>* out, bool quiet = false) {
  auto k = k_seq_fused<NG, 4, 16, 3, 0, 2>;
  const size_t smem = (size_t)16 * NG * 1024;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  FusedArgs fa;
  memset(&fa, 0, sizeof(fa));
  SliceArgs& a = fa.a;
  a.xop = g_x, a.part = g_part, a.S = 128, a.n_groups = n_tiles, a.G = G, a.halves = 1, a.rstd_wg = 1;
  a.ssq_in = g_ssq, a.ssq_n = 256, a.ssq_ld = 256, a.inv_k = 1.f / 4096, a.eps = 1e-5f, a.rstd_out = g_rstd;
  fa.cnt = g_cnt, fa.y = g_y, fa.dbg = g_dbg, fa.fused = variant == 'C' ? 1 : variant == 'D' ? 2 : variant == 'E' ? 3 : 0, fa.xcd_map = variant != 'A';
  CK(hipMemset(g_cnt, 0, 2048 * 4));
  CK(hipMemset(g_dbg, 0, 16));
  CK(hipMemset(g_y, 0xff, (size_t)n_tiles * NG * 512));
  auto launch = [&](int i) {
    fa.a.W = g_w + (size_t)(i % NL) * g_wstride;
    hipLaunchKernelGGL(k, dim3(4 * G + 1), dim3(512), smem, 0, fa);
    if (variant < 'C') hipLaunchKernelGGL(k_finish_simple<NG>, dim3(n_tiles), dim3(32 * NG), 0, 0, g_part, g_y, n_tiles);
  };
  launch(0);
  CK(hipDeviceSynchronize());
  if (out) {
    out->resize((size_t)n_tiles * NG * 128);
    CK(hipMemcpy(out->data(), g_y, out->size() * 4, hipMemcpyDeviceToHost));
  }
  unsigned dbg[4];
  CK(hipMemcpy(dbg, g_dbg, 16, hipMemcpyDeviceToHost));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 48;
  for (int i = 0; i < 6; ++i) launch(i);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch(i + rep);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  if (!quiet)
    printf("%c  %-62s tiles=%4d G=%2d : %7.2f us per GEMV (stream%s); first launch: %u workgroups off their XCD, %u tiles finished in-launch\n", variant,
           variant == 'A' ? "product block order, separate finishing kernel" : variant == 'B' ? "a group's four pairs on one XCD, separate finishing kernel"
           : variant == 'C' ? "a group's four pairs on one XCD, last arriver finishes (sc1 loads)" : variant == 'D' ? "... last arriver finishes (sc0 loads)" : "... last arriver finishes (plain loads)",
           n_tiles, G, best * 1e3 / reps, variant >= 'C' ? "" : " + finish", dbg[0], dbg[1]);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int max_tiles = 1536;
  g_wstride = (size_t)max_tiles * 128 * 64;
  CK(hipMalloc(&g_w, g_wstride * 16 * NL));
  CK(hipMalloc(&g_x, (size_t)NG * 128 * 64 * 16));
  CK(hipMalloc(&g_part, (size_t)4 * max_tiles * NG * 128 * 4));
  CK(hipMalloc(&g_y, (size_t)max_tiles * NG * 128 * 4));
  CK(hipMalloc(&g_ssq, (size_t)72 * 256 * 4));
  CK(hipMalloc(&g_rstd, 72 * 4));
  CK(hipMalloc(&g_cnt, 2048 * 4));
  CK(hipMalloc(&g_dbg, 16));
  CK(hipMemset(g_ssq, 0x3c, (size_t)72 * 256 * 4));
  k_fill<<<2048, 256>>>((uint32_t*)g_w, g_wstride * 4 * NL);
  k_fill<<<256, 256>>>((uint32_t*)g_x, (size_t)NG * 128 * 64 * 4);
  CK(hipDeviceSynchronize());
  if (argc > 1) {                                  // one variant, few launches: for counter passes
    run(argv[1][0], 1376, 64, nullptr);
    return 0;
  }
  for (int n_tiles : {1376, 1536}) {
    std::vector<float> ya, yb, yc, yd, ye, ya58;
    run('A', n_tiles, 64, &ya);
    run('B', n_tiles, 64, &yb);
    run('C', n_tiles, 64, &yc);
    run('D', n_tiles, 64, &yd);
    run('E', n_tiles, 64, &ye);
    if (n_tiles == 1376) run('A', n_tiles, 58, &ya58);
    auto same = [](const std::vector<float>& p, const std::vector<float>& q) { return p.size() == q.size() && !memcmp(p.data(), q.data(), p.size() * 4); };
    printf("   finished sums: B == A %s, C == A %s, D == A %s, E == A %s%s\n", same(ya, yb) ? "yes" : "NO", same(ya, yc) ? "yes" : "NO", same(ya, yd) ? "yes" : "NO", same(ya, ye) ? "yes" : "NO",
           n_tiles == 1376 ? (same(ya, ya58) ? ", A (58 per pair) == A yes" : ", A (58 per pair) == A NO") : "");
  }
  return 0;
}

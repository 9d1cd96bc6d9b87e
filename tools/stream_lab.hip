// stream_lab: what bounds the nine-plane slice GEMV's weight stream?  A measurement tool, not product; needs nothing from this repository.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_lab.hip -o tools/stream_lab && tools/stream_lab
//
// The product's 72-row gate/up GEMV (csrc/dd_gemv_slices.h k_gemv_slices_seq<9, 4, 16, 3>) streams 180 MB of weight tiles at 4.0-4.5 TB/s where a
// plain read reaches 6.2; round 5 measured that neither the operand stage-in, nor the partial-sum stores, nor the ring depth (4 -> 8 requests
// per wave) explains the gap.  This file takes the kernel apart:
//   PAT  — the ORDER in which a wave walks its 1 KiB weight pieces (the bytes and the number of requests are the same in every pattern)
//          0 product: tile-major matrix [tile][128 k-steps][64 lanes]; a wave owns (tile, slice-pair half) items and reads k-steps
//            2 qs + half + 8 s, s = 0..15: 1 KiB pieces 8 KiB apart
//          1 slice-major matrix [tile][8 slices][16 k-steps][64]: the same items, each a contiguous 16 KiB
//          2 wave-linear: every wave reads one contiguous run
//          3 chip-linear: at step n the 2,048 waves of the launch read 2 MiB side by side (what a copy kernel does)
//          4 product layout, both halves of a slice pair interleaved: 2 KiB pieces 8 KiB apart
//   WORK — what a wave does with a piece
//          0 folds it into a register (xor)
//          1 NG ds_read_b128 + NG MFMA 16x16x32 per piece, one piece at a time (sched_barrier after each, as the product does); the ORDER of
//            the reads and products inside a piece is the compiler's — hipcc 7.2 emits them as pairs: two reads, wait, two products, ...
//          2 the same without the sched_barrier (the compiler may overlap pieces)
//          4 as 1 with the order forced (sched_group_barrier): all NG reads of a piece, then its NG products
//          5 one continuous pipeline over (piece, plane): the operand of product t + WIN is requested right after product t is issued
//   U    — weight requests in flight per wave;  WAVES x grid — 8 x 256 (one workgroup per CU, the product) or 4 x 512 / 8 x 512 (two per CU)
// Weights cycle over NL copies of the matrix so nothing is cache-resident.  LDS contents are whatever they are (timing only).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct Args {
  const u32x4_t* W;
  float* out;
  int S;          // k-steps per tile (128 at K = 4096)
  int n_tiles;
  int G;          // workgroups per slice pair (grid = 4 G)
  int nt;         // 1: non-temporal weight loads
};

__device__ __forceinline__ f32x4_t mfma(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4_t ldw(int nt, const u32x4_t* p) { return nt ? __builtin_nontemporal_load(p) : *p; }

template <int U, int PAT, int WORK, int NG, int MAXG, int WAVES, int WIN = 4>
__global__ __launch_bounds__(WAVES * 64) void k_stream(Args a) {
  constexpr int SPW = 16;                       // k-steps per slice
  constexpr int NLOAD = 2 * MAXG * SPW;         // pieces per wave (all items live)
  extern __shared__ __align__(16) u32x4_t xs[];  // [SPW][NG][64] when WORK > 0
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qs = blockIdx.x & 3, j = blockIdx.x >> 2;
  const int b = blockIdx.x;
  const size_t nwaves = (size_t)gridDim.x * WAVES;
  // piece n of this wave -> offset in 1 KiB pieces (n is a compile-time constant wherever this is called: the loops are fully unrolled)
  auto piece = [&](int n) -> size_t {
    if constexpr (PAT == 2) return ((size_t)b * WAVES + wave) * NLOAD + n;
    if constexpr (PAT == 3) return (size_t)n * nwaves + (size_t)b * WAVES + wave;
    if constexpr (PAT == 4) {
      const int gi = n / (2 * SPW), r = n % (2 * SPW);
      const int g = j + a.G * (wave + WAVES * gi);
      return (size_t)g * a.S + 2 * qs + (r & 1) + 8 * (r >> 1);
    }
    const int item = n / SPW, s = n % SPW;
    const int half = item / MAXG, gi = item % MAXG;
    const int g = j + a.G * (wave + WAVES * gi);
    if constexpr (PAT == 0) return (size_t)g * a.S + 2 * qs + half + 8 * s;
    return ((size_t)g * 8 + 2 * qs + half) * SPW + s;     // PAT 1
  };
  u32x4_t w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) w[u] = ldw(a.nt, a.W + piece(u) * 64 + lane);
  if constexpr (WORK > 0) {                     // something in the LDS (bounded values: small bf16 patterns)
    for (int i = threadIdx.x; i < SPW * NG * 64; i += WAVES * 64) xs[i] = (u32x4_t){0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
  }
  f32x4_t acc[NG];
#pragma unroll
  for (int h = 0; h < NG; ++h) acc[h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  u32x4_t fold = (u32x4_t){0, 0, 0, 0};
  if constexpr (WORK == 5) {
    // one continuous pipeline over (piece, plane): the operand of product t + WIN is requested from LDS right after product t is issued
    static_assert((U * NG) % WIN == 0 && NLOAD % U == 0, "the operand ring must close over a block of U pieces");
    constexpr int T = NLOAD * NG, RING = SPW * NG;
    u32x4_t bq[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) bq[i] = xs[(size_t)(i % RING) * 64 + lane];
    auto block = [&](int n0, auto load_c) {
      constexpr bool LOAD = decltype(load_c)::value;
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int h = 0; h < NG; ++h) {
          const int tl = u * NG + h;             // compile-time: the ring slot
          const int t = n0 * NG + tl;
          acc[h] = mfma(w[u], bq[tl % WIN], acc[h]);
          bq[tl % WIN] = xs[(size_t)((t + WIN) % RING) * 64 + lane];      // (the last WIN reads of the launch are never used)
          if (LOAD && h == NG - 1) w[u] = ldw(a.nt, a.W + piece(n0 + u + U) * 64 + lane);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (LOAD && h == NG - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);      // (bounds the region the pipeline solver works on: one piece)
      }
    };
#pragma unroll 1
    for (int n0 = 0; n0 < NLOAD - U; n0 += U) block(n0, std::true_type{});
    block(NLOAD - U, std::false_type{});
  } else if constexpr (WORK == 0) {
#pragma unroll 1
    for (int n0 = 0; n0 < NLOAD; n0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        fold ^= w[u];
        w[u] = ldw(a.nt, a.W + piece(n0 + u + U < NLOAD ? n0 + u + U : n0 + u) * 64 + lane);     // (the last U pieces are read twice: no branch in the ring)
      }
    }
  } else {
#pragma unroll
    for (int n = 0; n < NLOAD; ++n) {
      const int u = n % U, s = n % SPW;
      {
        u32x4_t bb[NG];
#pragma unroll
        for (int h = 0; h < NG; ++h) bb[h] = xs[(size_t)(s * NG + h) * 64 + lane];
#pragma unroll
        for (int h = 0; h < NG; ++h) acc[h] = mfma(w[u], bb[h], acc[h]);
      }
      if (n + U < NLOAD) w[u] = ldw(a.nt, a.W + piece(n + U) * 64 + lane);
      if constexpr (WORK == 4) {                 // every operand of the piece requested before its first product
        __builtin_amdgcn_sched_group_barrier(0x100, NG, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NG, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if constexpr (WORK != 2) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = __builtin_bit_cast(float, fold.x ^ fold.y ^ fold.z ^ fold.w);
#pragma unroll
  for (int h = 0; h < NG; ++h) r += acc[h].x + acc[h].y + acc[h].z + acc[h].w;
  a.out[((size_t)b * WAVES + wave) * 64 + lane] = r;
}

static u32x4_t* g_w = nullptr;
static float* g_out = nullptr;
static size_t g_copy_pieces = 0;   // 1 KiB pieces per matrix copy
static int g_nl = 0;

template <int U, int PAT, int WORK, int NG, int MAXG, int WAVES, int WIN = 4>
static void run(const char* what, int grid, int lds_kib, int nt, int n_tiles) {
  Args a;
  a.S = 128, a.n_tiles = n_tiles, a.G = grid / 4, a.nt = nt, a.out = g_out;
  if ((size_t)a.G * WAVES * MAXG != (size_t)n_tiles) {
    printf("%-58s skipped: G * WAVES * MAXG != tiles\n", what);
    return;
  }
  auto k = k_stream<U, PAT, WORK, NG, MAXG, WAVES, WIN>;
  const size_t lds = (size_t)lds_kib * 1024;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int warm = 6, reps = 48;
  for (int i = 0; i < warm; ++i) {
    a.W = g_w + (size_t)(i % g_nl) * g_copy_pieces * 64;
    hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, 0, a);
  }
  CK(hipDeviceSynchronize());
  float best = 1e30f, tot = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) {
      a.W = g_w + (size_t)((i + rep) % g_nl) * g_copy_pieces * 64;
      hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, 0, a);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms;
    if (ms < best) best = ms;
  }
  const double bytes = (double)n_tiles * 128 * 1024;
  const double us = best * 1e3 / reps;
  int nblk = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)k, WAVES * 64, lds));
  printf("%-58s U=%d grid=%4d x %d waves, lds %3d KiB (%d wg/CU) nt=%d: %7.2f us  %5.2f TB/s  (mean %7.2f us)\n", what, U, grid, WAVES, lds_kib, nblk, nt, us,
         bytes / us * 1e-6, tot * 1e3 / (3 * reps));
  fflush(stdout);
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const int n_tiles = 1536;                      // 64 x 24: every wave of a 2,048-wave launch has three tiles (the product's 1,376: 2.69 on average)
  g_copy_pieces = (size_t)n_tiles * 128;
  g_nl = 8;
  const size_t bytes = g_copy_pieces * 1024 * g_nl;
  CK(hipMalloc(&g_w, bytes));
  CK(hipMemset(g_w, 0x3c, bytes));
  CK(hipMalloc(&g_out, (size_t)4096 * 8 * 64 * 4));
  printf("matrix: %d tiles x 128 KiB = %.1f MB, %d copies\n", n_tiles, g_copy_pieces * 1024 / 1e6, g_nl);
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");

  // ---- A: the order of the pieces, nothing but the loads (one workgroup of 8 waves per CU, as the product) ----
  run<4, 0, 0, 9, 3, 8>("A0 stream only, product order (1 KiB @ 8 KiB)", 256, 144, 1, n_tiles);
  run<4, 1, 0, 9, 3, 8>("A1 stream only, slice-major (16 KiB runs)", 256, 144, 1, n_tiles);
  run<4, 3, 0, 9, 3, 8>("A3 stream only, chip-linear", 256, 144, 1, n_tiles);
  run<8, 0, 0, 9, 3, 8>("A0 product order", 256, 144, 1, n_tiles);
  run<8, 3, 0, 9, 3, 8>("A3 chip-linear", 256, 144, 1, n_tiles);
  run<16, 0, 0, 9, 3, 8>("A0 product order", 256, 144, 1, n_tiles);
  run<4, 0, 0, 9, 3, 8>("A0 product order, plain (temporal) loads", 256, 144, 0, n_tiles);
  run<4, 0, 0, 9, 1, 8>("A0 product order, 768 workgroups of one tile per wave", 768, 48, 1, n_tiles);

  // ---- B: add the operand reads and the matrix products ----
  run<4, 0, 1, 9, 3, 8>("B1 + 9 ds_read + 9 MFMA per piece, compiler's order", 256, 144, 1, n_tiles);
  run<4, 0, 4, 9, 3, 8>("B4 ... all reads of a piece, then its products", 256, 144, 1, n_tiles);
  run<4, 0, 5, 9, 3, 8, 2>("B5 ... continuous pipeline, 2 operands ahead", 256, 144, 1, n_tiles);
  run<4, 0, 5, 9, 3, 8, 4>("B5 ... continuous pipeline, 4 operands ahead", 256, 144, 1, n_tiles);
  run<4, 0, 5, 9, 3, 8, 6>("B5 ... continuous pipeline, 6 operands ahead", 256, 144, 1, n_tiles);
  run<4, 0, 5, 9, 3, 8, 9>("B5 ... continuous pipeline, 9 operands ahead", 256, 144, 1, n_tiles);
  run<8, 0, 1, 9, 3, 8>("B1 compiler's order", 256, 144, 1, n_tiles);
  run<8, 0, 4, 9, 3, 8>("B4 all reads of a piece, then its products", 256, 144, 1, n_tiles);
  run<8, 0, 5, 9, 3, 8, 4>("B5 continuous pipeline, 4 operands ahead", 256, 144, 1, n_tiles);
  run<8, 0, 5, 9, 3, 8, 6>("B5 continuous pipeline, 6 operands ahead", 256, 144, 1, n_tiles);
  run<8, 0, 5, 9, 3, 8, 9>("B5 continuous pipeline, 9 operands ahead", 256, 144, 1, n_tiles);
  run<8, 3, 5, 9, 3, 8, 6>("B5 chip-linear, continuous pipeline, 6 ahead", 256, 144, 1, n_tiles);
  if (!quick) {
    // fewer planes: how the time grows with the work per piece
    run<4, 0, 1, 4, 3, 8>("B1 4 planes, compiler's order", 256, 144, 1, n_tiles);
    run<4, 0, 5, 4, 3, 8, 4>("B5 4 planes, continuous pipeline, 4 ahead", 256, 144, 1, n_tiles);
    run<8, 0, 5, 4, 3, 8, 4>("B5 4 planes, continuous pipeline, 4 ahead", 256, 144, 1, n_tiles);
    run<4, 0, 1, 2, 3, 8>("B1 2 planes, compiler's order", 256, 144, 1, n_tiles);
    run<8, 0, 5, 4, 1, 8, 4>("B5 4 planes, 768 wg, continuous pipeline, 4 ahead", 768, 64, 1, n_tiles);
    run<8, 0, 1, 4, 1, 8>("B1 4 planes, 768 wg, compiler's order", 768, 64, 1, n_tiles);
  }
  return 0;
}

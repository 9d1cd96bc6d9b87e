// gemv_lab: timing-only prototypes of the 32-row (four operand planes) decode GEMV — a measurement tool, not product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemv_lab.hip -o tools/gemv_lab && tools/gemv_lab
// Kernel A replicates k_gemv_groups' main loop (K split over the 8 waves of a workgroup, operand planes from global/L2,
// LDS reduce) with knobs that isolate what bounds it; kernel E is the slice-resident form (operand slice in LDS, a wave
// owns whole tiles over one K slice, partial sums out).  Weights cycle over `NL` copies so nothing is cache-resident.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>

#include "../dropoutdecoding_amd/csrc/dd_gemv_slices.h"   // the product's slice-resident kernel (typedefs come with it)
void dd_set_error(const char*, ...) {}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ f32x4_t mfma(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------
// A: replica of k_gemv_groups.  XMODE 0: operand planes from global (as shipped); 1: no operand loads (b := w);
// 2: operand loads folded onto a 16 KiB window (same request count, L1-resident)
// ------------------------------------------------------------------------------------------------------------
template <int TILES, int NG, int U, int XMODE>
__global__ __launch_bounds__(512) void k_a(const u32x4_t* __restrict__ W, const u32x4_t* __restrict__ X, float* __restrict__ out, int S) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int spw = S / 8;
  const int tile0 = blockIdx.x * TILES;
  f32x4_t acc[TILES][NG];
  const u32x4_t* wp[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[t][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    wp[t] = W + ((size_t)(tile0 + t) * S + wave) * 64 + lane;
  }
  const u32x4_t* xp = X + (size_t)wave * 64 + lane;
  const size_t xplane = (size_t)S * 64;
  int s = 0;
  for (; s + U <= spw; s += U) {
    u32x4_t b[U][NG], w[TILES][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) w[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)(s + u) * 8 * 64);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (XMODE == 0) b[u][g] = xp[(size_t)(s + u) * 8 * 64 + g * xplane];
        else if (XMODE == 2) b[u][g] = xp[(size_t)((s + u) & 1) * 8 * 64 + (g & 1) * xplane];
        else b[u][g] = w[0][u];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[t][g] = mfma(w[t][u], b[u][g], acc[t][g]);
  }
  for (; s < spw; ++s) {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      u32x4_t w = __builtin_nontemporal_load(wp[t] + (size_t)s * 8 * 64);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[t][g] = mfma(w, XMODE == 1 ? w : xp[(size_t)s * 8 * 64 + g * xplane], acc[t][g]);
    }
  }
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int g = 0; g < NG; ++g) *(f32x4_t*)&red[((t * NG + g) * 8 + wave) * 256 + lane * 4] = acc[t][g];
  __syncthreads();
  const int et = threadIdx.x, eg = et >> 7, ml = et & 7, en = (et & 127) >> 3;
  if (et < 128 * NG) {
    float y = 0.f;
    int o = ((en >> 2) * 16 + ml) * 4 + (en & 3);
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const float* r = &red[((t * NG + eg) * 8 + w) * 256];
        y += r[o] + r[o + 32];
      }
    out[((size_t)blockIdx.x * NG + eg) * 128 + (et & 127)] = y;
  }
}

// ------------------------------------------------------------------------------------------------------------
// E: slice-resident.  K is cut into NSL interleaved slices (slice q = k-steps q, q+NSL, ...).  A workgroup holds ONE slice of
// all NG operand planes in LDS (read once from L2) and its waves each own groups of TW tiles: a wave streams its tiles'
// k-steps of that slice straight to registers, U steps requested ahead, takes the B operands from LDS, and writes the
// slice's partial sums.  No barrier after the operand slice has landed.
// grid = NSL * G; WG (q = b % NSL, j = b / NSL); tile groups of slice q: g = j + G * (wave + WAVES * i)
// ------------------------------------------------------------------------------------------------------------
template <int TW, int NG, int U, int WAVES, int XSRC>
__global__ __launch_bounds__(WAVES * 64) void k_e(const u32x4_t* __restrict__ W, const u32x4_t* __restrict__ X, float* __restrict__ out,
                                                  int S, int n_groups, int NSL, int G) {
  extern __shared__ __align__(16) u32x4_t xs[];   // [cs][NG][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x % NSL, j = blockIdx.x / NSL;
  const int cs = (S - q + NSL - 1) / NSL;         // k-steps of this slice
  const size_t xplane = (size_t)S * 64;
  if (XSRC == 0) {
    // operand slice -> LDS: cs * NG pieces of 1 KiB, dealt over the waves, 4 requests in flight per wave
    const int pieces = cs * NG;
    for (int p0 = wave; p0 < pieces; p0 += WAVES * 4) {
      u32x4_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int p = p0 + u * WAVES;
        if (p < pieces) v[u] = X[(size_t)(q + (p / NG) * NSL) * 64 + (p % NG) * xplane + lane];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int p = p0 + u * WAVES;
        if (p < pieces) xs[(size_t)p * 64 + lane] = v[u];
      }
    }
    __syncthreads();
  }
  for (int g = j + G * wave; g < n_groups; g += G * WAVES) {
    f32x4_t acc[TW][NG];
    const u32x4_t* wp[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
#pragma unroll
      for (int h = 0; h < NG; ++h) acc[t][h] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      wp[t] = W + ((size_t)(g * TW + t) * S + q) * 64 + lane;
    }
    const size_t wstep = (size_t)NSL * 64;
    auto bop = [&](int c, int h) -> u32x4_t {
      return XSRC == 0 ? xs[(size_t)(c * NG + h) * 64 + lane] : X[(size_t)(q + c * NSL) * 64 + h * xplane + lane];
    };
    // ring of U requests per tile: slot u is consumed and at once re-requested U steps ahead, so ~U KiB per tile stay in
    // flight for the whole slice (sched_barrier keeps the compiler from sinking the requests down to their uses)
    u32x4_t w[TW][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TW; ++t)
        w[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)(u < cs ? u : cs - 1) * wstep);
    __builtin_amdgcn_sched_barrier(0);
    int c = 0;
    for (; c + 2 * U <= cs; c += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        u32x4_t b[NG];
#pragma unroll
        for (int h = 0; h < NG; ++h) b[h] = bop(c + u, h);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
#pragma unroll
          for (int h = 0; h < NG; ++h) acc[t][h] = mfma(w[t][u], b[h], acc[t][h]);
          w[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)(c + U + u) * wstep);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // drain: steps c .. cs-1 are in (or about to be in) the ring; steps >= c + U still have to be requested
    for (; c < cs; c += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (c + u < cs) {
          u32x4_t b[NG];
#pragma unroll
          for (int h = 0; h < NG; ++h) b[h] = bop(c + u, h);
#pragma unroll
          for (int t = 0; t < TW; ++t) {
#pragma unroll
            for (int h = 0; h < NG; ++h) acc[t][h] = mfma(w[t][u], b[h], acc[t][h]);
            if (c + U + u < cs) w[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)(c + U + u) * wstep);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // partial sums of this slice: hi + lo columns folded (lane c and c + 8 of each 16-lane group), 2 KiB per tile and slice
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int h = 0; h < NG; ++h) {
        f32x4_t v = acc[t][h];
        v.x += __shfl_down(v.x, 8);
        v.y += __shfl_down(v.y, 8);
        v.z += __shfl_down(v.z, 8);
        v.w += __shfl_down(v.w, 8);
        if ((lane & 8) == 0) {
          int l32 = (lane >> 4) * 8 + (lane & 7);
          *(f32x4_t*)&out[((((size_t)q * n_groups + g) * TW + t) * NG + h) * 128 + l32 * 4] = v;
        }
      }
  }
}

struct Shape {
  const char* name;
  int n_tiles, S, tiles_a;   // tiles_a: TILES of the shipped kernel for this matrix
};

static float time_it(hipStream_t st, int iters, const std::function<void(int)>& launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 8; ++i) launch(i);
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms * 1000.f / iters;
}

template <int TILES, int NG, int U, int XMODE>
static void run_a(const Shape& sh, const u32x4_t* W, size_t wstride, int NL, const u32x4_t* X, float* out, hipStream_t st) {
  size_t smem = (size_t)TILES * NG * 8 * 256 * 4;
  CK(hipFuncSetAttribute((const void*)k_a<TILES, NG, U, XMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  int grid = sh.n_tiles / TILES;
  float us = time_it(st, 64, [&](int i) { k_a<TILES, NG, U, XMODE><<<grid, 512, smem, st>>>(W + (size_t)(i % NL) * wstride, X, out, sh.S); });
  double bytes = (double)sh.n_tiles * sh.S * 1024;
  printf("  A  %-8s TILES=%d NG=%d U=%d XMODE=%d grid=%4d lds=%3zuK : %7.2f us  %5.2f TB/s\n", sh.name, TILES, NG, U, XMODE, grid, smem >> 10,
         us, bytes / us * 1e-6);
}

template <int TW, int NG, int U, int WAVES, int XSRC>
static void run_e(const Shape& sh, int NSL, int G, const u32x4_t* W, size_t wstride, int NL, const u32x4_t* X, float* out, hipStream_t st) {
  int cs = (sh.S + NSL - 1) / NSL;
  size_t smem = XSRC == 0 ? (size_t)cs * NG * 1024 : 0;
  if (smem > 160 * 1024) {
    printf("  E  %-8s NSL=%d: slice needs %zuK of LDS, skipped\n", sh.name, NSL, smem >> 10);
    return;
  }
  CK(hipFuncSetAttribute((const void*)k_e<TW, NG, U, WAVES, XSRC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  int n_groups = sh.n_tiles / TW;
  float us = time_it(st, 64, [&](int i) {
    k_e<TW, NG, U, WAVES, XSRC><<<NSL * G, WAVES * 64, smem, st>>>(W + (size_t)(i % NL) * wstride, X, out, sh.S, n_groups, NSL, G);
  });
  double bytes = (double)sh.n_tiles * sh.S * 1024;
  printf("  E  %-8s TW=%d NG=%d U=%d WAVES=%2d XSRC=%d NSL=%2d G=%3d grid=%4d lds=%3zuK groups/wave=%.2f : %7.2f us  %5.2f TB/s\n", sh.name, TW, NG, U,
         WAVES, XSRC, NSL, G, NSL * G, smem >> 10, (double)n_groups / (G * WAVES), us, bytes / us * 1e-6);
}

template <int TW, int NG, int U, int SPW, int CS, int CH = 1>
static void run_s(const Shape& sh, int G, const u32x4_t* W, size_t wstride, int NL, const u32x4_t* X, float* out, hipStream_t st, int halves = 1) {
  size_t smem = (size_t)CH * (CS < SPW ? CS : SPW) * NG * 1024;
  CK(hipFuncSetAttribute((const void*)k_gemv_slices<TW, NG, U, SPW, CS, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  SliceArgs a;
  memset(&a, 0, sizeof(a));
  a.xop = X, a.part = out, a.S = sh.S, a.n_groups = sh.n_tiles / TW, a.G = G, a.halves = halves;
  const int grid = halves * (8 / CH) * G;
  float us = time_it(st, 64, [&](int i) {
    SliceArgs b = a;
    b.W = W + (size_t)(i % NL) * wstride;
    k_gemv_slices<TW, NG, U, SPW, CS, CH><<<grid, 512, smem, st>>>(b);
  });
  double bytes = (double)sh.n_tiles * sh.S * 1024;
  printf("  S  %-8s TW=%d NG=%d U=%2d SPW=%d CS=%d CH=%d halves=%d G=%3d grid=%4d lds=%3zuK : %7.2f us  %5.2f TB/s (%d rows)\n", sh.name, TW, NG, U, SPW, CS,
         CH, halves, G, grid, smem >> 10, us, bytes / us * 1e-6, 8 * NG * halves);
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : "";
  hipStream_t st;
  CK(hipStreamCreate(&st));
  Shape shapes[4] = {{"qkv", 768, 128, 2}, {"o", 256, 128, 1}, {"gateup", 1376, 128, 2}, {"down", 256, 344, 1}};
  const int NL = 6;
  size_t wmax = (size_t)1376 * 128 * 64;   // u32x4 units of the largest matrix
  u32x4_t* W;
  CK(hipMalloc((void**)&W, wmax * 16 * NL));
  {
    // bf16 pattern with small magnitudes (never NaN/Inf): 0x3c00 | low bits
    std::vector<uint32_t> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c003c00u | ((uint32_t)(i * 2654435761u) & 0x007f007fu);
    for (size_t off = 0; off < wmax * 16 * NL; off += h.size() * 4) {
      size_t n = wmax * 16 * NL - off < h.size() * 4 ? wmax * 16 * NL - off : h.size() * 4;
      CK(hipMemcpy((char*)W + off, h.data(), n, hipMemcpyHostToDevice));
    }
  }
  u32x4_t* X;
  CK(hipMalloc((void**)&X, (size_t)344 * 64 * 16 * 8));     // up to 8 operand planes (64 rows)
  CK(hipMemcpy(X, W, (size_t)344 * 64 * 16 * 8, hipMemcpyDeviceToDevice));
  float* out;
  CK(hipMalloc((void**)&out, (size_t)64 << 20));
  for (const Shape& sh : shapes) {
    if (only[0] && strcmp(only, sh.name)) continue;
    size_t wstride = (size_t)sh.n_tiles * sh.S * 64;
    printf("%s: %d tiles x %d k-steps = %.1f MB\n", sh.name, sh.n_tiles, sh.S, (double)sh.n_tiles * sh.S * 1024 / 1e6);
#define A(T, NG, U, XM) run_a<T, NG, U, XM>(sh, W, wstride, NL, X, out, st)
#define E(TW, NG, U, WV, XS, NSL, G) run_e<TW, NG, U, WV, XS>(sh, NSL, G, W, wstride, NL, X, out, st)
#define SL(TW, NG, U, SPW, CS, G) run_s<TW, NG, U, SPW, CS>(sh, G, W, wstride, NL, X, out, st)
#define SH(TW, NG, U, SPW, CS, CH, G, HV) run_s<TW, NG, U, SPW, CS, CH>(sh, G, W, wstride, NL, X, out, st, HV)
    if (!strcmp(sh.name, "qkv")) {
      SH(2, 4, 8, 16, 16, 1, 24, 1);
      SH(1, 8, 8, 16, 16, 1, 24, 1);
      SH(2, 4, 8, 16, 16, 1, 24, 2);
      SH(2, 4, 8, 16, 16, 1, 16, 2);
      SH(1, 4, 8, 16, 16, 2, 32, 2);
      SH(1, 4, 8, 16, 16, 2, 64, 2);
    } else if (!strcmp(sh.name, "o")) {
      SH(1, 4, 8, 16, 16, 1, 32, 1);
      SH(1, 8, 8, 16, 16, 1, 16, 1);
      SH(1, 4, 8, 16, 16, 1, 16, 2);
      SH(1, 4, 8, 16, 16, 1, 32, 2);
      SH(1, 4, 8, 16, 16, 2, 32, 2);
    } else if (!strcmp(sh.name, "gateup")) {
      SH(1, 4, 8, 16, 16, 2, 64, 1);
      SH(1, 8, 8, 16, 16, 1, 32, 1);
      SH(1, 4, 8, 16, 16, 2, 64, 2);
      SH(1, 4, 8, 16, 16, 2, 32, 2);
      SH(1, 4, 8, 16, 16, 2, 128, 2);
      SH(1, 4, 8, 16, 16, 1, 32, 2);
    } else {
      SH(1, 4, 8, 43, 16, 1, 32, 1);
      SH(1, 8, 8, 43, 8, 1, 32, 1);
      SH(1, 4, 8, 43, 16, 1, 32, 2);
    }
  }
  return 0;
}

// seq_lab: the product's nine-plane gate/up kernel (csrc/dd_gemv_slices.h k_gemv_slices_seq<9, 4, 16, 3>) timed alone under controlled conditions —
// a measurement tool, not product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-enable-packed-fp32=0 ... (see build line in tools/README.md) tools/seq_lab.hip -o tools/seq_lab
// tools/stream_lab.hip shows that the kernel's main loop (one 1 KiB weight piece: nine ds_read_b128 + nine MFMA, four requests in flight per wave, one
// workgroup per CU) runs at the speed of a plain read (6.1 TB/s) when nothing else happens.  This driver takes the real kernel and removes /
// changes one ingredient at a time: the operand stage-in (SliceArgs::temporal bit 1), the partial-sum stores (bit 2), the rstd prologue of
// workgroup 0 (ssq_in), the tile count (1,376 = 2.97 tiles per wave over 58 workgroups per slice pair = 232 workgroups on 256 CUs; 1,392: even;
// 1,536 over 64 per pair: 256 workgroups, three tiles per wave), and the data (constant / random bit patterns).
#define DD_TIMING_EXPERIMENTS 1   // the timing-only branches of dd_gemv_slices.h (DD_TEXP) are compiled out of the product library
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

__global__ void k_fill(uint32_t* p, size_t n, int mode) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u;
    h ^= h >> 15, h *= 2246822519u, h ^= h >> 13, h *= 3266489917u, h ^= h >> 16;
    // two bf16: random sign and mantissa, exponent 0x78..0x7b (|x| in [2^-7, 2^-3)): what synthetic weights look like
    uint32_t v = (h & 0x807f807fu) | 0x3c003c00u | ((h >> 3) & 0x01800180u);
    p[i] = mode ? v : 0x3c003c00u;
  }
}

static u32x4_t *g_w, *g_x;
static float *g_part, *g_ssq, *g_rstd;
static const int NL = 8;
static size_t g_wstride;   // u32x4 per matrix copy

template <int NG, int U, int MAXG>
static void run(const char* what, int n_tiles, int G, int temporal, bool ssq, int extra = 0) {
  auto k = k_gemv_slices_seq<NG, U, 16, MAXG, 0, 2>;
  const size_t smem = (size_t)16 * NG * 1024;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  SliceArgs a;
  memset(&a, 0, sizeof(a));
  a.xop = g_x, a.part = g_part, a.S = 128, a.n_groups = n_tiles, a.G = G, a.halves = 1, a.temporal = temporal & ~16, a.rstd_wg = (temporal & 16) ? 0 : 1;
  if (ssq) a.ssq_in = g_ssq, a.ssq_n = 256, a.ssq_ld = 256, a.inv_k = 1.f / 4096, a.eps = 1e-5f, a.rstd_out = g_rstd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 48;
  for (int i = 0; i < 6; ++i) {
    a.W = g_w + (size_t)(i % NL) * g_wstride;
    hipLaunchKernelGGL(k, dim3(4 * G + extra), dim3(512), smem, 0, a);
  }
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) {
      a.W = g_w + (size_t)((i + rep) % NL) * g_wstride;
      hipLaunchKernelGGL(k, dim3(4 * G + extra), dim3(512), smem, 0, a);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double us = best * 1e3 / reps, bytes = (double)n_tiles * 128 * 1024;
  printf("%-64s U=%d tiles=%4d G=%2d (%3d wg) : %7.2f us  %5.2f TB/s\n", what, U, n_tiles, G, 4 * G, us, bytes / us * 1e-6);
  fflush(stdout);
}

int main() {
  const int max_tiles = 1536;
  g_wstride = (size_t)max_tiles * 128 * 64;
  CK(hipMalloc(&g_w, g_wstride * 16 * NL));
  CK(hipMalloc(&g_x, (size_t)9 * 128 * 64 * 16));
  CK(hipMalloc(&g_part, (size_t)4 * max_tiles * 9 * 128 * 4));
  CK(hipMalloc(&g_ssq, (size_t)72 * 256 * 4));
  CK(hipMalloc(&g_rstd, 72 * 4));
  CK(hipMemset(g_ssq, 0x3c, (size_t)72 * 256 * 4));
  for (int mode = 1; mode >= 0; --mode) {
    k_fill<<<2048, 256>>>((uint32_t*)g_w, g_wstride * 4 * NL, mode);
    k_fill<<<256, 256>>>((uint32_t*)g_x, (size_t)9 * 128 * 64 * 4, mode);
    CK(hipDeviceSynchronize());
    printf("---- data: %s\n", mode ? "random bf16 (sign, mantissa, four exponents)" : "constant");
    // temporal bits: 2 no stage-in, 4 no stores, 8 stores where round 5 had them (at each set's completion), 16 (this driver only) rstd in workgroup 0 (SliceArgs::rstd_wg = 0)
    run<9, 4, 3>("round 5: rstd in workgroup 0, stores in mid-stream", 1376, 58, 8 + 16, true);
    run<9, 4, 3>("rstd in a workgroup of its own (233rd), stores in mid-stream", 1376, 58, 8, true, 1);
    run<9, 4, 3>("rstd in workgroup 0, stores at the end", 1376, 58, 16, true);
    run<9, 4, 3>("both: rstd in its own workgroup, stores at the end", 1376, 58, 0, true, 1);
    run<9, 4, 3>("... no rstd at all", 1376, 58, 0, false);
    run<9, 4, 3>("... no rstd, no stage-in", 1376, 58, 2, false);
    run<9, 4, 3>("... no rstd, no stores", 1376, 58, 4, false);
    run<9, 4, 3>("... no rstd, neither", 1376, 58, 6, false);
    run<9, 4, 3>("both, 64 per pair (257 workgroups)", 1376, 64, 0, true, 1);
    run<9, 8, 3>("both, eight requests per wave", 1376, 58, 0, true, 1);
    run<9, 4, 3>("1,536 tiles over 64 per pair, stores at the end, no rstd", 1536, 64, 0, false);
    run<9, 4, 3>("1,536 tiles, neither stage-in nor stores", 1536, 64, 6, false);
  }
  return 0;
}

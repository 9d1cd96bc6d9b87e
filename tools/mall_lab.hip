// mall_lab: is a buffer that one kernel has just written served to the next kernel by the 256 MB Infinity Cache, or by the HBM stacks?
// A measurement tool, not product; needs nothing from this repository.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mall_lab.hip -o tools/mall_lab && tools/mall_lab
// Why: the slice GEMVs hand 84 MB of partial sums per layer and sweep to their finishing kernels through memory.  FETCH_SIZE /
// WRITE_SIZE count what crosses from the L2s to the fabric and cannot tell the memory-side cache from the DRAM behind it; time can.
// For a region of S MB: (hot) write it, read it back at once;  (cold) write it, stream 3 GB of something else, read it;  (re-read) read it twice.
// First: a plain cold read of 1 GB with one 16-byte store per 16 / 8 / 4 loads mixed in (the step reads about eight bytes per byte it writes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ __launch_bounds__(256) void k_write(u32x4_t* p, size_t n, uint32_t tag) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (u32x4_t){tag, (uint32_t)i, tag ^ (uint32_t)i, 1u};
}
template <int NT>
__global__ __launch_bounds__(256) void k_read(const u32x4_t* p, size_t n, uint32_t* out) {
  u32x4_t f = (u32x4_t){0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    u32x4_t a, b, c, d;
    if (NT) a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride), c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
    else a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    f ^= a ^ b ^ c ^ d;
  }
  for (; i < n; i += stride) f ^= p[i];
  if ((f.x ^ f.y ^ f.z ^ f.w) == 0x12345678u) out[0] = 1;
}

// reads with one 16-byte store per EVERY loads (EVERY = 0: none): the step's mix is about eight bytes read per byte written
template <int EVERY>
__global__ __launch_bounds__(256) void k_mix(const u32x4_t* p, size_t n, u32x4_t* q, uint32_t* out) {
  u32x4_t f = (u32x4_t){0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, j = i;
  int c = 0;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const u32x4_t a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride), cc = __builtin_nontemporal_load(p + i + 2 * stride),
                  d = __builtin_nontemporal_load(p + i + 3 * stride);
    f ^= a ^ b ^ cc ^ d;
    if (EVERY && ++c * 4 >= EVERY) {
      q[j] = f;
      j += stride;
      c = 0;
    }
  }
  if ((f.x ^ f.y ^ f.z ^ f.w) == 0x12345678u) out[0] = 1;
}
template <int EVERY>
static void mix(const u32x4_t* big, size_t big_n, u32x4_t* x, uint32_t* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t n = ((size_t)1 << 30) / 16;                  // 1 GB read per launch, a different GB each time
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_mix<EVERY>, dim3(2048), dim3(256), 0, 0, big + (size_t)(rep % 3) * n, n, x, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double rd = 1073.741824, wr = EVERY ? rd / EVERY : 0.0;   // MB
  printf("1 GB read, one 16-byte store per %2d loads: %7.1f us  read %5.2f TB/s + written %5.2f TB/s = %5.2f TB/s\n", EVERY, best * 1e3, rd / (best * 1e3), wr / (best * 1e3),
         (rd + wr) / (best * 1e3));
  (void)big_n;
}

static float timed_read(const u32x4_t* p, size_t n, uint32_t* out, bool nt) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  if (nt) hipLaunchKernelGGL(k_read<1>, dim3(2048), dim3(256), 0, 0, p, n, out);
  else hipLaunchKernelGGL(k_read<0>, dim3(2048), dim3(256), 0, 0, p, n, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms * 1e3f;
}

int main() {
  const size_t big_bytes = (size_t)3 << 30;
  u32x4_t *big, *x;
  uint32_t* out;
  CK(hipMalloc(&big, big_bytes));
  CK(hipMalloc(&x, (size_t)768 << 20));
  CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, big, big_bytes / 16, 7u);
  CK(hipDeviceSynchronize());
  mix<0>(big, big_bytes / 16, x, out);
  mix<16>(big, big_bytes / 16, x, out);
  mix<8>(big, big_bytes / 16, x, out);
  mix<4>(big, big_bytes / 16, x, out);
  printf("%8s | %22s | %22s | %22s | %22s\n", "region", "read right after write", "after 3 GB of other reads", "second read in a row", "hot, non-temporal loads");
  for (int mb : {16, 32, 64, 128, 192, 256, 384, 768}) {
    const size_t n = ((size_t)mb << 20) / 16;
    float best[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, x, n, (uint32_t)rep);
      float t = timed_read(x, n, out, false);                       // hot
      if (t < best[0]) best[0] = t;
      hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, x, n, (uint32_t)rep + 100);
      timed_read(big, big_bytes / 16, out, false);                  // flush
      t = timed_read(x, n, out, false);                             // cold
      if (t < best[1]) best[1] = t;
      t = timed_read(x, n, out, false);                             // read again
      if (t < best[2]) best[2] = t;
      hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, x, n, (uint32_t)rep + 200);
      t = timed_read(x, n, out, true);                              // hot, nt loads
      if (t < best[3]) best[3] = t;
    }
    printf("%5d MB |", mb);
    for (int k = 0; k < 4; ++k) printf(" %8.1f us %6.2f TB/s |", best[k], (double)mb * 1.048576 / best[k]);
    printf("\n");
    fflush(stdout);
  }
  return 0;
}

// Measurement code of round 1 (persistent-sweep sizing): NOT part of libdropdec.so any more.
// Kept as a record; to run it, compile it next to the library sources and bind dd_persist_read_bench by hand.
// Persistent-sweep experiments (measurement hooks, not part of the reference's surface).
//
// Question this file answers with numbers: the decode sweep is ~170 dependent kernel launches whose weight streams each
// pay a fill/drain ramp (DESIGN.md "Measured": 5.5 TB/s asymptote + 2.3 us per launch).  Would ONE resident kernel
// that walks the same phases, separated by grid barriers, with the next phase's first loads issued BEFORE the barrier,
// keep HBM busy across the phase boundaries?
#include "dd_common.h"
#include "../../include/dropdec.h"

#define PB_MAX_PHASES 8
#define RC(expr)              \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != DD_OK) return rc__; \
  } while (0)

struct PersistBenchArgs {
  const u32x4_t* base;        // weights arena
  size_t layer_stride16;      // in 16-byte units
  size_t phase_off16[PB_MAX_PHASES];
  size_t phase_n16[PB_MAX_PHASES];   // 16-byte units in this phase (multiple of 64 * gridDim)
  int n_phases, n_layers;
  unsigned int* barrier;      // zeroed counter
  unsigned int* sink;
  int* err;
};

__device__ __forceinline__ bool pb_grid_barrier(unsigned int* ctr, unsigned int target, int* err) {
  // one arrival per workgroup; spin is bounded so a scheduling surprise cannot hang the box
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) {
        *err = 1;
        ok = false;
        break;
      }
    }
  }
  return ok;
}

// U = 16-byte loads in flight per lane; PREFETCH = issue the next phase's first batch before waiting on the barrier
template <int U, int PREFETCH>
__global__ __launch_bounds__(512) void k_persist_read(PersistBenchArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = gridDim.x * 8;                 // waves in the grid
  const int gw = blockIdx.x * 8 + wave;         // this wave's index: consecutive 1 KiB chunks go to consecutive waves of a WG
  unsigned int acc = 0;
  unsigned int epoch = 0;
  u32x4_t v[U];
  bool have = false;
  for (int l = 0; l < a.n_layers; ++l) {
    for (int p = 0; p < a.n_phases; ++p) {
      const u32x4_t* src = a.base + (size_t)l * a.layer_stride16 + a.phase_off16[p];
      const size_t n_chunks = a.phase_n16[p] / 64;          // 1 KiB per wave per chunk
      // this WG owns a contiguous slab; inside it the 8 waves interleave chunk by chunk
      const size_t per_wg = n_chunks / gridDim.x;
      const size_t c0 = (size_t)blockIdx.x * per_wg;
      size_t c = wave;
      if (have) {                                            // batch issued before the barrier
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].w;
        c += (size_t)U * 8;
        have = false;
      }
      for (; c + (size_t)(U - 1) * 8 < per_wg; c += (size_t)U * 8) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + (c0 + c + (size_t)u * 8) * 64 + lane);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].w;
      }
      for (; c < per_wg; c += 8) acc ^= __builtin_nontemporal_load(src + (c0 + c) * 64 + lane).x;
      // ---- phase boundary ----
      const bool last = (l == a.n_layers - 1) && (p == a.n_phases - 1);
      if (last) break;
      ++epoch;
      if (PREFETCH) {
        int np = p + 1, nl = l;
        if (np == a.n_phases) np = 0, nl = l + 1;
        const u32x4_t* nsrc = a.base + (size_t)nl * a.layer_stride16 + a.phase_off16[np];
        const size_t nper = (a.phase_n16[np] / 64) / gridDim.x;
        const size_t nc0 = (size_t)blockIdx.x * nper;
        if (nper >= (size_t)U * 8) {
          if (wave != 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(nsrc + (nc0 + wave + (size_t)u * 8) * 64 + lane);
          }
          have = true;
        }
        __builtin_amdgcn_s_barrier();
        bool ok = pb_grid_barrier(a.barrier, epoch * gridDim.x, a.err);
        if (have && wave == 0) {
#pragma unroll
          for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(nsrc + (nc0 + wave + (size_t)u * 8) * 64 + lane);
        }
        __builtin_amdgcn_s_barrier();
        (void)ok;
      } else {
        __syncthreads();
        pb_grid_barrier(a.barrier, epoch * gridDim.x, a.err);
        __syncthreads();
      }
      if (*(volatile int*)a.err) return;
    }
  }
  if (acc == 0x9e3779b9u) a.sink[0] = acc;
  (void)nw;
  (void)gw;
}

// one launch per phase (what the engine does today, minus the arithmetic)
template <int U>
__global__ __launch_bounds__(512) void k_phase_read(const u32x4_t* __restrict__ src, size_t n16, unsigned int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t n_chunks = n16 / 64;
  const size_t per_wg = n_chunks / gridDim.x;
  const size_t c0 = (size_t)blockIdx.x * per_wg;
  unsigned int acc = 0;
  u32x4_t v[U];
  size_t c = wave;
  for (; c + (size_t)(U - 1) * 8 < per_wg; c += (size_t)U * 8) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + (c0 + c + (size_t)u * 8) * 64 + lane);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].w;
  }
  for (; c < per_wg; c += 8) acc ^= __builtin_nontemporal_load(src + (c0 + c) * 64 + lane).x;
  if (acc == 0x9e3779b9u) sink[0] = acc;
}

// mode 0: one launch per phase with `grid` workgroups; 1: persistent + barriers; 2: persistent + prefetch across the
// barrier.  phase_bytes[] are per-layer phase sizes laid out back to back; ms_out = time of one sweep of n_layers.
extern "C" int dd_persist_read_bench(const void* buf_dev, size_t layer_stride_bytes, const size_t* phase_bytes, int n_phases,
                                     int n_layers, int mode, int grid, int U, int iters, float* ms_out, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(buf_dev && phase_bytes && ms_out && n_phases >= 1 && n_phases <= PB_MAX_PHASES && n_layers >= 1 && iters >= 1,
             "dd_persist_read_bench: bad arguments");
  DD_REQUIRE(grid >= 1 && (U == 8 || U == 16), "dd_persist_read_bench: grid >= 1, U in {8, 16}");
  PersistBenchArgs a{};
  a.base = (const u32x4_t*)buf_dev;
  a.layer_stride16 = layer_stride_bytes / 16;
  size_t off = 0;
  for (int p = 0; p < n_phases; ++p) {
    a.phase_off16[p] = off / 16;
    a.phase_n16[p] = phase_bytes[p] / 16;
    off += phase_bytes[p];
  }
  DD_REQUIRE(off <= layer_stride_bytes, "dd_persist_read_bench: phases exceed the layer stride");
  a.n_phases = n_phases, a.n_layers = n_layers;
  unsigned int* scratch = nullptr;
  DD_HIP(hipMalloc((void**)&scratch, 64));
  a.barrier = scratch, a.sink = scratch + 4, a.err = (int*)(scratch + 8);
  hipEvent_t e0, e1;
  DD_HIP(hipEventCreate(&e0));
  DD_HIP(hipEventCreate(&e1));
  // mode 0 only: bits 8.. of `mode` = workgroups per CU allowed (dynamic LDS used as an occupancy throttle)
  const int wg_per_cu = mode >> 8;
  mode &= 0xff;
  const size_t smem = wg_per_cu > 0 ? (size_t)(160 * 1024 / wg_per_cu) - 1024 : 0;
  if (smem > 64 * 1024) {
    DD_HIP(hipFuncSetAttribute((const void*)k_phase_read<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DD_HIP(hipFuncSetAttribute((const void*)k_phase_read<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  }
  auto sweep = [&]() -> int {
    if (mode == 0) {
      for (int l = 0; l < n_layers; ++l)
        for (int p = 0; p < n_phases; ++p) {
          const u32x4_t* src = a.base + (size_t)l * a.layer_stride16 + a.phase_off16[p];
          if (U == 8) k_phase_read<8><<<grid, 512, smem, st>>>(src, a.phase_n16[p], a.sink);
          else k_phase_read<16><<<grid, 512, smem, st>>>(src, a.phase_n16[p], a.sink);
        }
    } else {
      DD_HIP(hipMemsetAsync(scratch, 0, 64, st));
      if (mode == 1) {
        if (U == 8) k_persist_read<8, 0><<<grid, 512, 0, st>>>(a);
        else k_persist_read<16, 0><<<grid, 512, 0, st>>>(a);
      } else {
        if (U == 8) k_persist_read<8, 1><<<grid, 512, 0, st>>>(a);
        else k_persist_read<16, 1><<<grid, 512, 0, st>>>(a);
      }
    }
    DD_CHECK_LAUNCH();
    return DD_OK;
  };
  RC(sweep());
  DD_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) RC(sweep());
  DD_HIP(hipEventRecord(e1, st));
  DD_HIP(hipEventSynchronize(e1));
  float ms = 0;
  DD_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  int err = 0;
  DD_HIP(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(scratch);
  if (err) {
    dd_set_error("dd_persist_read_bench: grid barrier timed out (workgroups not co-resident?)");
    return DD_ESTATE;
  }
  return DD_OK;
}

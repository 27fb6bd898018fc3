// tools/dma_rate.hip -- how many LDS-DMA (global_load_lds) and plain vector loads a CU issues per microsecond, by active
// lanes and bytes per lane.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/dma_rate tools/dma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BYTES, int MODE>  // MODE 0: LDS-DMA, 1: global_load + ds_write, 2: global load only (kept in a register sum)
__global__ void __launch_bounds__(256) k(const float4* __restrict__ src, int nsrc_mask, int reps, int active, float* out) {
  __shared__ __attribute__((aligned(16))) float4 lds[1280];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned idx = (blockIdx.x * 977u + wave * 131u) & nsrc_mask;
  float acc = 0.f;
  for (int r = 0; r < reps; r++) {
    idx = (idx * 1664525u + 1013904223u) & nsrc_mask;  // a new window of `active` consecutive elements
    const unsigned base = idx & ~63u;
    if (lane < active) {
      if (MODE == 0) {
        if (BYTES == 16)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + lane),
                                           (__attribute__((address_space(3))) void*)(lds + wave * 320 + (r & 3) * 64), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const float*>(src) + base + lane),
                                           (__attribute__((address_space(3))) void*)(reinterpret_cast<float*>(lds) + wave * 320 + (r & 3) * 64), 4, 0, 0);
      } else if (MODE == 1) {
        if (BYTES == 16) lds[wave * 320 + (r & 3) * 64 + lane] = src[base + lane];
        else reinterpret_cast<float*>(lds)[wave * 320 + (r & 3) * 64 + lane] = reinterpret_cast<const float*>(src)[base + lane];
      } else {
        if (BYTES == 16) acc += src[base + lane].x;
        else acc += reinterpret_cast<const float*>(src)[base + lane];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (out) out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x].x + acc;
}

template <int BYTES, int MODE> void run(const float4* src, int mask, int reps, int active, int grid, float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL((k<BYTES, MODE>), dim3(grid), dim3(256), 0, 0, src, mask, reps, active, out);
  hipEventRecord(e0);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k<BYTES, MODE>), dim3(grid), dim3(256), 0, 0, src, mask, reps, active, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double instr = (double)grid * 4 * reps;
  printf("%-28s %2d B/lane %2d lanes: %8.1f us, %6.1f wave-instr/us/CU = one per %5.1f ns per CU (%.2f TB/s)\n", name, BYTES, active, ms * 1e3,
         instr / (ms * 1e3) / 256, ms * 1e6 * 256 / instr, instr * active * BYTES / (ms * 1e-3) / 1e12);
}

int main() {
  const int n = 1 << 20;  // 16 MB of float4: L2 / MALL resident
  float4* src;
  float* out;
  hipMalloc(&src, sizeof(float4) * n);
  hipMemset(src, 0, sizeof(float4) * n);
  hipMalloc(&out, 4 * 256 * 65536);
  const int grid = 32768, reps = 36;
  for (int active : {8, 16, 32, 64}) {
    run<16, 0>(src, n - 1, reps, active, grid, out, "LDS-DMA");
    run<4, 0>(src, n - 1, reps, active, grid, out, "LDS-DMA");
    run<16, 1>(src, n - 1, reps, active, grid, out, "global_load + ds_write");
    run<4, 1>(src, n - 1, reps, active, grid, out, "global_load + ds_write");
    run<16, 2>(src, n - 1, reps, active, grid, out, "global_load");
  }
  return 0;
}

// tools/launch_rate.hip -- how fast the chip launches workgroups of the COUNT sweep's shape (256 threads, 20 KB LDS, 8 per CU):
// empty workgroups, and workgroups whose waves idle for a given number of cycles (slot turnover).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/launch_rate tools/launch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int THREADS, int LDS_BYTES>
__global__ void __launch_bounds__(THREADS) k(int cycles, int skew, float* out) {
  __shared__ char lds[LDS_BYTES];
  const int wave = threadIdx.x >> 6;
  // each wave idles cycles * (1 + skew% * wave / 4): s_sleep 127 = 127 * 64 cycles
  long long t0 = __builtin_amdgcn_s_memtime();
  const long long want = (long long)cycles + (long long)cycles * skew * wave / 400;
  if (cycles > 0) {
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < want) __builtin_amdgcn_s_sleep(8);
  }
  if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = lds[threadIdx.x];
}

template <int THREADS, int LDS_BYTES> void run(int grid, int cycles, int skew, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL((k<THREADS, LDS_BYTES>), dim3(grid), dim3(THREADS), 0, 0, cycles, skew, nullptr);
  hipEventRecord(e0);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k<THREADS, LDS_BYTES>), dim3(grid), dim3(THREADS), 0, 0, cycles, skew, nullptr);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const int per_cu = LDS_BYTES ? (160 * 1024) / LDS_BYTES : 99;
  const int slots = 256 * (per_cu < 2048 / THREADS ? per_cu : 2048 / THREADS);
  const double ideal_us = cycles > 0 ? (double)grid / slots * cycles * (1.0 + skew * (THREADS / 64 - 1) / 400.0) / 2100.0 : 0.0;
  printf("%-12s grid %6d x %4d thr, %5d B LDS, idle %6d cyc (+%d%% skew): %8.1f us  (%.2f ns per workgroup; slots x life = %.1f us)\n", name, grid,
         THREADS, LDS_BYTES, cycles, skew, ms * 1e3, ms * 1e6 / grid, ideal_us);
}

int main() {
  for (int grid : {27000, 54872}) {
    run<256, 20480>(grid, 0, 0, "empty");
    run<128, 8192>(grid, 0, 0, "empty");
    run<64, 4096>(grid * 4, 0, 0, "empty");
    run<256, 20480>(grid, 15000, 0, "idle");
    run<256, 20480>(grid, 30000, 0, "idle");
    run<256, 20480>(grid, 30000, 40, "idle+skew");
    run<256, 10240>(grid, 30000, 0, "idle");
    run<128, 8192>(grid, 30000, 0, "idle");
  }
  return 0;
}

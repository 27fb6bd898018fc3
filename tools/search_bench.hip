// tools/search_bench.hip -- isolates the inner loop of the pair search: every wave runs the REAL search_group<>
// (nl_kernels.hpp) on a resident LDS tile set, no staging, no barriers, for many repetitions.  Reports shader
// cycles per wave-test (one i-particle against one 64-lane j-tile) per SIMD at several occupancies, for the
// COUNT and the FILL body.  Build with the library's flags:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o tools/search_bench tools/search_bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "../md_neighbor_list_amd/csrc/nl_kernels.hpp"

using namespace nl;

#define CHK(x)                                                             \
  do {                                                                     \
    hipError_t e = (x);                                                    \
    if (e != hipSuccess) {                                                 \
      printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

constexpr int NJ = 1088;  // 17 tiles, the mean stencil of BASELINE config 2

template <int MODE, int GC>
__global__ void __launch_bounds__(256) kb(SweepArgs<float> a, const Pos<float>* jsrc, const Pos<float>* isrc, int reps,
                                          unsigned long long* stamps, int32_t* sink) {
  extern __shared__ Pos<float> tile[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < NJ; k += 256) tile[k] = jsrc[k];
  __syncthreads();
  Pos<float> pi_l = isrc[(blockIdx.x * 4 + wave) % 64 * 8 + (lane & 7)];
  // FILL writes into a private region of the list so that stores are real; offsets per lane k
  const int32_t base_l = ((blockIdx.x * 4 + wave) * 8 + (lane & 7)) * 256;
  unsigned long long t0, t1;
  int32_t acc = 0;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; r++) acc += search_group<float, MODE, GC>(a, tile, NJ, NJ / 64, lane, pi_l, base_l);
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) stamps[blockIdx.x * 4 + wave] = t1 - t0;
  if (acc == 0x7fffffff) sink[0] = acc;
}

template <int MODE, int GC> int run(const char* name, SweepArgs<float> a, const Pos<float>* j, const Pos<float>* i,
                                    unsigned long long* stamps_d, int32_t* sink) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, reps = 40;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kb<MODE, GC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("%-22s", name);
  for (int bpc : {1, 2, 4, 6, 8}) {
    const size_t lds = (size_t)(160 * 1024 / bpc) & ~(size_t)1023;
    const int blocks = cus * bpc, nw = blocks * 4;
    hipLaunchKernelGGL((kb<MODE, GC>), dim3(blocks), dim3(256), lds, 0, a, j, i, 2, stamps_d, sink);
    hipLaunchKernelGGL((kb<MODE, GC>), dim3(blocks), dim3(256), lds, 0, a, j, i, reps, stamps_d, sink);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(nw);
    CHK(hipMemcpy(st.data(), stamps_d, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
    std::sort(st.begin(), st.end());
    const double tests = (double)reps * GC * (NJ / 64);
    printf(" | w%d med %6.2f max %6.2f", bpc, (double)st[nw / 2] / (tests * bpc), (double)st[nw - 1] / (tests * bpc));
  }
  printf("\n");
  return 0;
}

int main() {
  // one cell's worth of j-particles: uniform in a 3x3x3 block of cells of edge 3.386 (27 * 40 = 1080 + 8)
  std::mt19937 mt(5);
  std::uniform_real_distribution<float> u(0.f, 3.f * 3.386f);
  std::vector<Pos<float>> hj(NJ), hi(64 * 8);
  for (int k = 0; k < NJ; k++) hj[k] = {u(mt), u(mt), u(mt), (int32_t)(mt() % 1000000)};
  std::uniform_real_distribution<float> uc(3.386f, 2.f * 3.386f);
  for (auto& p : hi) p = {uc(mt), uc(mt), uc(mt), (int32_t)(mt() % 1000000)};
  Pos<float>*dj, *di;
  unsigned long long* stamps;
  int32_t *list, *sink;
  CHK(hipMalloc(&dj, sizeof(Pos<float>) * NJ));
  CHK(hipMalloc(&di, sizeof(Pos<float>) * hi.size()));
  CHK(hipMalloc(&stamps, 8 * 256 * 8 * 4));
  CHK(hipMalloc(&list, sizeof(int32_t) * 256 * 8 * 4 * 8 * 256));
  CHK(hipMalloc(&sink, 16));
  CHK(hipMemcpy(dj, hj.data(), sizeof(Pos<float>) * NJ, hipMemcpyHostToDevice));
  CHK(hipMemcpy(di, hi.data(), sizeof(Pos<float>) * hi.size(), hipMemcpyHostToDevice));
  SweepArgs<float> a{};
  a.rc2 = 3.3f * 3.3f;
  a.list = list;
  printf("search_group<> alone: shader cycles per wave-test per SIMD (w = waves per SIMD); med/max over waves: the SIMD\narbitrates oldest-first, so only the LAST wave\x27s finish time (max) measures throughput\n");
  run<MODE_COUNT, 4>("COUNT GC=4", a, dj, di, stamps, sink);
  run<MODE_COUNT, 5>("COUNT GC=5", a, dj, di, stamps, sink);
  run<MODE_COUNT, 6>("COUNT GC=6", a, dj, di, stamps, sink);
  run<MODE_FILL, 5>("FILL  GC=5", a, dj, di, stamps, sink);
  run<MODE_FILL, 6>("FILL  GC=6", a, dj, di, stamps, sink);
  return 0;
}

// tools/mfma_bench.hip -- isolates the tile loop of k_sweep_mfma_f32: every wave runs the REAL mf_rows<NB>
// (nl_sweep_mfma.hpp) on a resident LDS image of one synthetic cell (40 rows, 1056 staged particles = 66 tiles),
// no staging, no global traffic, for many repetitions.  Reports shader cycles per (wave, tile) step per SIMD at
// several occupancies; max over waves = throughput (the SIMD arbitrates oldest-first).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o tools/mfma_bench tools/mfma_bench.hip
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

constexpr int NJ = 1056, NI = 40, IOFF = 512;

template <int NB>
__global__ void __launch_bounds__(256, 4) kb(SweepArgs<float> a, int reps, unsigned long long* stamps, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  MfmaLds& L = *reinterpret_cast<MfmaLds*>(raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float cc = 1.5f * 3.386f;
  for (int k = tid; k < NJ; k += 256) {
    const Pos<float> v = a.sorted[k];
    const float ux = v.x - cc, uy = v.y - cc, uz = v.z - cc;
    L.comp[k] = ux, L.comp[MF_CSTR + k] = uy, L.comp[2 * MF_CSTR + k] = uz;
    L.comp[3 * MF_CSTR + k] = ux * ux + uy * uy + uz * uz;
    L.gid[k] = v.gid;
  }
  for (int k = tid; k < MF_ROWS * MF_WORDS; k += 256) L.words[k] = 0;
  if (tid < MF_ROWS) L.cnt[tid] = 0;
  __syncthreads();
  CellCtx c{};
  c.ibeg = IOFF, c.ni = NB * 16 - 8, c.total_j = NJ;
  c.seg_src = 0, c.seg_len = lane == 0 ? NJ : 0, c.seg_off = lane == 0 ? 0 : NJ;
  int32_t t_beg, nt;
  mf_tile_range(NJ / 16, wave, t_beg, nt);
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; r++) mf_rows<NB>(a, c, L, lane, IOFF, 0, t_beg, nt);
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) stamps[blockIdx.x * 4 + wave] = t1 - t0;
  if (L.words[tid] == 0x12345678u) sink[0] = 1;
}

template <int NB> int run(const char* name, SweepArgs<float> a, unsigned long long* stamps_d, uint32_t* sink) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, reps = 40;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kb<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("%-12s", name);
  for (int bpc : {1, 2, 3, 4}) {
    const size_t lds = (size_t)(160 * 1024 / bpc) & ~(size_t)1023;
    const int blocks = cus * bpc, nw = blocks * 4;
    hipLaunchKernelGGL((kb<NB>), dim3(blocks), dim3(256), lds, 0, a, 2, stamps_d, sink);
    hipLaunchKernelGGL((kb<NB>), dim3(blocks), dim3(256), lds, 0, a, reps, stamps_d, sink);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(nw);
    CHK(hipMemcpy(st.data(), stamps_d, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
    std::sort(st.begin(), st.end());
    const double steps = (double)reps * (NJ / 16) / 4.0;  // tile steps per wave (16.5)
    printf(" | w%d med %7.1f max %7.1f", bpc, (double)st[nw / 2] / (steps * bpc), (double)st[nw - 1] / (steps * bpc));
  }
  printf("\n");
  return 0;
}

int main() {
  std::mt19937 mt(5);
  std::uniform_real_distribution<float> u(0.f, 3.f * 3.386f);
  std::vector<Pos<float>> hj(NJ);
  for (int k = 0; k < NJ; k++) hj[k] = {u(mt), u(mt), u(mt), (int32_t)(mt() % 1000000)};
  std::uniform_real_distribution<float> uc(3.386f, 2.f * 3.386f);
  for (int k = IOFF; k < IOFF + 48; k++) hj[k] = {uc(mt), uc(mt), uc(mt), (int32_t)(mt() % 1000000)};
  Pos<float>* dj;
  unsigned long long* stamps;
  uint32_t* sink;
  CHK(hipMalloc(&dj, sizeof(Pos<float>) * NJ));
  CHK(hipMalloc(&stamps, 8 * 256 * 8 * 4));
  CHK(hipMalloc(&sink, 16));
  CHK(hipMemcpy(dj, hj.data(), sizeof(Pos<float>) * NJ, hipMemcpyHostToDevice));
  SweepArgs<float> a{};
  a.sorted = dj;
  a.rc2 = 3.3f * 3.3f;
  a.delta = 3.5e-4f;
  printf("mf_rows<NB> alone: shader cycles per tile step (NB MFMAs + their vector work) per SIMD; w = waves per SIMD\n");
  run<1>("NB=1", a, stamps, sink);
  run<2>("NB=2", a, stamps, sink);
  run<3>("NB=3", a, stamps, sink);
  return 0;
}

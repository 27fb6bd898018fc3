// tools/mfma_bench.hip -- isolates the tile loop of k_sweep_mfma_f32: every wave runs the REAL mf_unit
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

template <int NI, bool F16>
__global__ void __launch_bounds__(MF_WAVES * 64, 8) kb(SweepArgs<float> a, int reps, unsigned long long* stamps, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  MfmaLds& L = *reinterpret_cast<MfmaLds*>(raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float cc = 1.5f * 3.386f;
  for (int k = tid; k < NJ; k += MF_WAVES * 64) {
    const Pos<float> v = a.sorted[k];
    float ux = v.x - cc, uy = v.y - cc, uz = v.z - cc;
    if (F16) {
      ux *= a.mf_scale, uy *= a.mf_scale, uz *= a.mf_scale;
      L.comp[k] = __uint_as_float(mf_split16(ux)), L.comp[MF_CSTR + k] = __uint_as_float(mf_split16(uy));
      L.comp[2 * MF_CSTR + k] = __uint_as_float(mf_split16(uz));
      L.comp[3 * MF_CSTR + k] = __uint_as_float(mf_split16(ux * ux + uy * uy + uz * uz));
    } else {
      L.comp[k] = ux, L.comp[MF_CSTR + k] = uy, L.comp[2 * MF_CSTR + k] = uz;
      L.comp[3 * MF_CSTR + k] = ux * ux + uy * uy + uz * uz;
    }
    L.gid[k] = v.gid;
  }
  if (tid < MF_ROWS) L.cnt[tid] = 0;

  __syncthreads();
  CellCtx c{};
  c.ibeg = IOFF, c.ni = NI, c.total_j = NJ;
  c.seg_src = 0, c.seg_len = lane == 0 ? NJ : 0, c.seg_off = lane == 0 ? 0 : NJ;
  const int32_t ntiles = NJ / 16;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; r++)
    for (int32_t i0 = 0; i0 < NI; i0 += 16) mf_unit<F16>(a, c, L, lane, IOFF, i0, wave, ntiles);
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) stamps[blockIdx.x * MF_WAVES + wave] = t1 - t0;
  if (L.cnt[tid & 63] == 0x12345678) sink[0] = 1;
}

template <int NI, bool F16> int run(const char* name, SweepArgs<float> a, unsigned long long* stamps_d, uint32_t* sink) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, reps = 40;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kb<NI, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("%-12s", name);
  for (int bpc : {1, 2, 4, 6}) {
    const size_t lds = (size_t)(160 * 1024 / bpc) & ~(size_t)1023;
    const int blocks = cus * bpc, nw = blocks * MF_WAVES;
    hipLaunchKernelGGL((kb<NI, F16>), dim3(blocks), dim3(MF_WAVES * 64), lds, 0, a, 2, stamps_d, sink);
    hipLaunchKernelGGL((kb<NI, F16>), dim3(blocks), dim3(MF_WAVES * 64), lds, 0, a, reps, stamps_d, sink);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(nw);
    CHK(hipMemcpy(st.data(), stamps_d, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
    std::sort(st.begin(), st.end());
    // MFMA steps (one 16 x 16 block of tests) per SIMD: workgroups per CU x i-blocks x 66 tiles / 4 SIMDs
    const double steps = (double)reps * bpc * ((NI + 15) / 16) * (NJ / 16) / 4.0;
    printf(" | %d wg/CU (%d waves/SIMD) max %6.1f", bpc, bpc, (double)st[nw - 1] / steps);
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
  uint32_t* masks;
  CHK(hipMalloc(&masks, sizeof(uint32_t) * 64 * (IOFF + MF_ROWS)));
  a.masks = masks;  // (every workgroup writes the same rows: timing only)
  a.rc2 = 3.3f * 3.3f;
  a.delta = 3.5e-4f;
  printf("mf_unit alone: shader cycles per SIMD per MFMA step (one 16 x 16 block of tests: 1 MFMA + its vector work),\nlast wave to finish\n");
  a.mf_scale = 8.f, a.mf_scale2 = 64.f, a.delta16 = a.delta * 64.f;
  run<40, false>("fp32, 40 rows", a, stamps, sink);
  run<40, true>("f16,  40 rows", a, stamps, sink);
  run<64, true>("f16,  64 rows", a, stamps, sink);
  return 0;
}

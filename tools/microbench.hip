// tools/microbench.hip -- VALU issue-rate probe for gfx950: how many cycles does a wave64 v_add_f32 /
// v_pk_add_f32 / v_cmp cost at 1..8 waves per SIMD?  Decides whether packed fp32 halves the cost of the
// distance test.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 * 2, a2 = a0 * 3, a3 = a0 * 4, a4 = a0 * 5, a5 = a0 * 6, a6 = a0 * 7, a7 = a0 * 8;
  const float b = seed * 0.5f;
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {  // 8 independent v_add_f32
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (KIND == 1) {  // 4 independent v_pk_add_f32 (8 float adds)
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {b, b};
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
      a0 = p0.x, a1 = p0.y, a2 = p1.x, a3 = p1.y, a4 = p2.x, a5 = p2.y, a6 = p3.x, a7 = p3.y;
    } else if (KIND == 2) {  // 8 v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (KIND == 3) {  // 8 v_cmp_lt_f32 into SGPR pairs (the ballot path)
      unsigned long long m0, m1, m2, m3;
      asm volatile("v_cmp_lt_f32 %0, %4, %8\n v_cmp_lt_f32 %1, %5, %8\n v_cmp_lt_f32 %2, %6, %8\n v_cmp_lt_f32 %3, %7, %8\n"
                   "v_cmp_lt_f32 %0, %5, %8\n v_cmp_lt_f32 %1, %6, %8\n v_cmp_lt_f32 %2, %7, %8\n v_cmp_lt_f32 %3, %4, %8\n"
                   : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));
      a4 += (float)(m0 ^ m1 ^ m2 ^ m3);
    } else if (KIND == 4) {  // 8 v_add_f64
      double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = b;
      asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                   "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
      a0 = d0, a1 = d1, a2 = d2, a3 = d3;
    } else if (KIND == 5) {  // 8 v_sub with an SGPR operand (the form the sweep uses)
      asm volatile("v_subrev_f32 %0, %8, %0\n v_subrev_f32 %1, %8, %1\n v_subrev_f32 %2, %8, %2\n v_subrev_f32 %3, %8, %3\n"
                   "v_subrev_f32 %4, %8, %4\n v_subrev_f32 %5, %8, %5\n v_subrev_f32 %6, %8, %6\n v_subrev_f32 %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b));
    } else if (KIND == 6) {  // 8 v_mbcnt pairs -> 8 instructions (4 lo + 4 hi)
      unsigned int u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3);
      asm volatile("v_mbcnt_lo_u32_b32 %0, %0, 0\n v_mbcnt_hi_u32_b32 %0, %1, %0\n v_mbcnt_lo_u32_b32 %1, %1, 0\n v_mbcnt_hi_u32_b32 %1, %2, %1\n"
                   "v_mbcnt_lo_u32_b32 %2, %2, 0\n v_mbcnt_hi_u32_b32 %2, %3, %2\n v_mbcnt_lo_u32_b32 %3, %3, 0\n v_mbcnt_hi_u32_b32 %3, %0, %3\n"
                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
      a0 = __uint_as_float(u0), a1 = __uint_as_float(u1), a2 = __uint_as_float(u2), a3 = __uint_as_float(u3);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND> int run(const char* name, int ops_per_iter, float* out) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int iters = 20000;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  printf("%-28s", name);
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
    const int blocks = cus * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD = wps * iters * ops_per_iter ; cycles at 2.4 GHz
    const double instr = (double)wps * iters * ops_per_iter;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("  wps=%d: %.2f cyc/instr (%.3f ms)", wps, cyc / instr, ms);
  }
  printf("\n");
  return 0;
}

int main() {
  float* out;
  CHK(hipMalloc(&out, 256 * 8 * 256 * 4 * 2));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, clock %d kHz (cycles below assume 2.4 GHz)\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  run<0>("v_add_f32 x8", 8, out);
  run<1>("v_pk_add_f32 x4 (8 adds)", 4, out);
  run<2>("v_mul_f32 x8", 8, out);
  run<3>("v_cmp_lt_f32->sgpr x8", 8, out);
  run<4>("v_add_f64 x8", 8, out);
  run<5>("v_subrev_f32 sgpr x8", 8, out);
  run<6>("v_mbcnt lo/hi x8", 8, out);
  return 0;
}

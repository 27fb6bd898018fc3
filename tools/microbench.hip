// tools/microbench.hip -- VALU issue-rate probe for gfx950 (MI355X).
// For each instruction kind: real shader cycles per wave-instruction per SIMD at 1, 2, 4, 8 waves per SIMD, and
// the clock the chip actually holds (s_memtime vs the 100 MHz s_memrealtime).  One 160-KiB-LDS block per CU pins
// the placement (blocks of 256/512/1024 threads = 1/2/4 waves per SIMD; two 80-KiB blocks of 1024 for 8).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                  \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);      \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// 8 independent instructions, repeated 8 times = 64 per loop iteration
#define REP8(X) X X X X X X X X

template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* stamps, int iters, float seed) {
  extern __shared__ float lds[];
  float a0 = seed + threadIdx.x, a1 = a0 * 2, a2 = a0 * 3, a3 = a0 * 4, a4 = a0 * 5, a5 = a0 * 6, a6 = a0 * 7, a7 = a0 * 8;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  const float b = seed * 0.5f;
  const f2 bb = {b, b};
  const double db = b;
  const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b)));
  unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;
  __syncthreads();
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                        "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 1) {
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                        "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(bb));)
    } else if (KIND == 2) {
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                        "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(bb));)
    } else if (KIND == 3) {
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                        "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(bb));)
    } else if (KIND == 4) {
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                        "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 5) {  // compare into VCC (e32 encoding)
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                        "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 6) {  // compare into an arbitrary SGPR pair (e64 encoding)
      REP8(asm volatile("v_cmp_lt_f32 %0, %4, %8\n v_cmp_lt_f32 %1, %5, %8\n v_cmp_lt_f32 %2, %6, %8\n v_cmp_lt_f32 %3, %7, %8\n"
                        "v_cmp_lt_f32 %0, %5, %8\n v_cmp_lt_f32 %1, %6, %8\n v_cmp_lt_f32 %2, %7, %8\n v_cmp_lt_f32 %3, %4, %8\n"
                        : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));)
    } else if (KIND == 7) {  // integer compare with an SGPR source into an SGPR pair
      REP8(asm volatile("v_cmp_lt_i32 %0, %8, %4\n v_cmp_lt_i32 %1, %8, %5\n v_cmp_lt_i32 %2, %8, %6\n v_cmp_lt_i32 %3, %8, %7\n"
                        "v_cmp_lt_i32 %0, %8, %5\n v_cmp_lt_i32 %1, %8, %6\n v_cmp_lt_i32 %2, %8, %7\n v_cmp_lt_i32 %3, %8, %4\n"
                        : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(sb));)
    } else if (KIND == 8) {
      REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                        "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));)
    } else if (KIND == 9) {
      REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                        "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));)
    } else if (KIND == 10) {
      REP8(asm volatile("v_subrev_f32 %0, %8, %0\n v_subrev_f32 %1, %8, %1\n v_subrev_f32 %2, %8, %2\n v_subrev_f32 %3, %8, %3\n"
                        "v_subrev_f32 %4, %8, %4\n v_subrev_f32 %5, %8, %5\n v_subrev_f32 %6, %8, %6\n v_subrev_f32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sb));)
    } else if (KIND == 11) {  // the distance test as the sweep kernel issues it today: 8 instructions per test
      REP8(asm volatile(
               "v_subrev_f32 %0, %9, %4\n v_mul_f32 %0, %0, %0\n v_pk_add_f32 %1, %5, %6\n v_cmp_lt_i32 %2, %9, %4\n"
               "v_pk_mul_f32 %1, %1, %1\n s_nop 0\n v_add_f32 %0, %0, %7\n v_add_f32 %0, %8, %0\n v_cmp_nlt_f32 %3, %9, %0\n"
               : "=&v"(a4), "=&v"(p4), "=s"(m0), "=s"(m1)
               : "v"(a0), "v"(p0), "v"(bb), "v"(a1), "v"(a2), "s"(sb));)
    } else if (KIND == 12) {  // same test, all scalar-width ops (no packed): 10 instructions
      REP8(asm volatile(
               "v_subrev_f32 %0, %6, %3\n v_subrev_f32 %1, %6, %4\n v_subrev_f32 %2, %6, %5\n"
               "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n"
               "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_cmp_lt_i32 %7, %6, %3\n v_cmp_nlt_f32 %8, %6, %0\n"
               : "=&v"(a4), "=&v"(a5), "=&v"(a6)
               : "v"(a0), "v"(a1), "v"(a2), "s"(sb), "s"(m0), "s"(m1));)
    } else if (KIND == 14) {  // unpacked test + the scalar bookkeeping of the count pass (s_and, s_bcnt1, s_add)
      REP8(asm volatile(
               "v_subrev_f32 v40, s40, v46\n v_subrev_f32 v41, s41, v47\n v_subrev_f32 v42, s42, v48\n"
               "v_mul_f32 v40, v40, v40\n v_mul_f32 v41, v41, v41\n v_mul_f32 v42, v42, v42\n"
               "v_add_f32 v40, v40, v41\n v_add_f32 v40, v40, v42\n"
               "v_cmp_lt_i32 s[52:53], s47, v52\n v_cmp_nlt_f32 vcc, s46, v40\n"
               "s_and_b64 s[48:49], s[52:53], vcc\n s_bcnt1_i32_b64 s50, s[48:49]\n s_add_i32 s51, s51, s50\n"
               ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "s48", "s49", "s50", "s51", "s52", "s53", "vcc", "scc");)
    } else if (KIND == 15) {  // same, bookkeeping delayed by one test (consume the PREVIOUS test's masks)
      REP8(asm volatile(
               "v_subrev_f32 v40, s40, v46\n v_subrev_f32 v41, s41, v47\n v_subrev_f32 v42, s42, v48\n"
               "s_and_b64 s[48:49], s[52:53], s[54:55]\n"
               "v_mul_f32 v40, v40, v40\n v_mul_f32 v41, v41, v41\n v_mul_f32 v42, v42, v42\n"
               "s_bcnt1_i32_b64 s50, s[48:49]\n"
               "v_add_f32 v40, v40, v41\n v_add_f32 v40, v40, v42\n"
               "s_add_i32 s51, s51, s50\n"
               "v_cmp_lt_i32 s[52:53], s47, v52\n v_cmp_nlt_f32 s[54:55], s46, v40\n"
               ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "scc");)
    } else if (KIND == 16) {  // per-lane counting: no scalar ops at all (cndmask + cmp + addc), 11 VALU per test
      REP8(asm volatile(
               "v_subrev_f32 v40, s40, v46\n v_subrev_f32 v41, s41, v47\n v_subrev_f32 v42, s42, v48\n"
               "v_mul_f32 v40, v40, v40\n v_mul_f32 v41, v41, v41\n v_mul_f32 v42, v42, v42\n"
               "v_add_f32 v40, v40, v41\n v_add_f32 v40, v40, v42\n"
               "v_cmp_nlt_f32 vcc, s46, v40\n v_cndmask_b32 v43, v53, v52, vcc\n v_cmp_lt_i32 vcc, s47, v43\n"
               "v_addc_co_u32 v44, vcc, 0, v44, vcc\n"
               ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "vcc");)
    } else if (KIND == 40) {
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                        "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 41) {
      REP8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                        "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sb) : "vcc");)
    } else if (KIND == 42) {
      REP8(asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n"
                        "v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 43) {
      REP8(asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n"
                        "v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 44) {
      REP8(asm volatile("v_bfi_b32 %0, %8, %0, %0\n v_bfi_b32 %1, %8, %1, %1\n v_bfi_b32 %2, %8, %2, %2\n v_bfi_b32 %3, %8, %3, %3\n"
                        "v_bfi_b32 %4, %8, %4, %4\n v_bfi_b32 %5, %8, %5, %5\n v_bfi_b32 %6, %8, %6, %6\n v_bfi_b32 %7, %8, %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 45) {
      REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n"
                        "v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n v_addc_co_u32 %7, vcc, %7, %7, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 46) {
      REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 47) {
      REP8(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                        "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 48) {
      REP8(asm volatile("v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3\n"
                        "v_subrev_u32 %4, %8, %4\n v_subrev_u32 %5, %8, %5\n v_subrev_u32 %6, %8, %6\n v_subrev_u32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sb) : "vcc");)
    } else if (KIND == 49) {
      REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                        "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 50) {
      REP8(asm volatile("v_fmac_f32 %0, %8, %8\n v_fmac_f32 %1, %8, %8\n v_fmac_f32 %2, %8, %8\n v_fmac_f32 %3, %8, %8\n"
                        "v_fmac_f32 %4, %8, %8\n v_fmac_f32 %5, %8, %8\n v_fmac_f32 %6, %8, %8\n v_fmac_f32 %7, %8, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 51) {
      REP8(asm volatile("v_mul_legacy_f32 %0, %0, %8\n v_mul_legacy_f32 %1, %1, %8\n v_mul_legacy_f32 %2, %2, %8\n v_mul_legacy_f32 %3, %3, %8\n"
                        "v_mul_legacy_f32 %4, %4, %8\n v_mul_legacy_f32 %5, %5, %8\n v_mul_legacy_f32 %6, %6, %8\n v_mul_legacy_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 52) {
      REP8(asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                        "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sb) : "vcc");)
    } else if (KIND == 53) {
      REP8(asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n"
                        "v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 54) {
      REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                        "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 55) {
      REP8(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n"
                        "v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 56) {
      REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                        "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 58) {
      REP8(asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                        "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m0) : "vcc");)
    } else if (KIND == 59) {
      REP8(asm volatile("v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n"
                        "v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m0) : "vcc");)
    } else if (KIND == 20) {
      REP8(asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n"
                        "v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 21) {
      REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                        "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 22) {
      REP8(asm volatile("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n"
                        "v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 23) {
      REP8(asm volatile("v_min3_f32 %0, %0, |%8|, |%1|\n v_min3_f32 %1, %1, |%8|, |%2|\n v_min3_f32 %2, %2, |%8|, |%3|\n v_min3_f32 %3, %3, |%8|, |%4|\n"
                        "v_min3_f32 %4, %4, |%8|, |%5|\n v_min3_f32 %5, %5, |%8|, |%6|\n v_min3_f32 %6, %6, |%8|, |%7|\n v_min3_f32 %7, %7, |%8|, |%0|\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 24) {  // fp32 MFMA alone, 8 independent accumulators
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 v[60:63], v46, v47, v[56:59]\n v_mfma_f32_16x16x4_f32 v[64:67], v46, v47, v[56:59]\n"
                        "v_mfma_f32_16x16x4_f32 v[68:71], v46, v47, v[56:59]\n v_mfma_f32_16x16x4_f32 v[72:75], v46, v47, v[56:59]\n"
                        "v_mfma_f32_16x16x4_f32 v[76:79], v46, v47, v[56:59]\n v_mfma_f32_16x16x4_f32 v[80:83], v46, v47, v[56:59]\n"
                        "v_mfma_f32_16x16x4_f32 v[84:87], v46, v47, v[56:59]\n v_mfma_f32_16x16x4_f32 v[88:91], v46, v47, v[56:59]\n"
                        ::: "v46", "v47", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",
                            "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91");)
    } else if (KIND == 25) {  // f16 MFMA alone
      REP8(asm volatile("v_mfma_f32_16x16x32_f16 v[60:63], v[40:43], v[44:47], v[56:59]\n v_mfma_f32_16x16x32_f16 v[64:67], v[40:43], v[44:47], v[56:59]\n"
                        "v_mfma_f32_16x16x32_f16 v[68:71], v[40:43], v[44:47], v[56:59]\n v_mfma_f32_16x16x32_f16 v[72:75], v[40:43], v[44:47], v[56:59]\n"
                        "v_mfma_f32_16x16x32_f16 v[76:79], v[40:43], v[44:47], v[56:59]\n v_mfma_f32_16x16x32_f16 v[80:83], v[40:43], v[44:47], v[56:59]\n"
                        "v_mfma_f32_16x16x32_f16 v[84:87], v[40:43], v[44:47], v[56:59]\n v_mfma_f32_16x16x32_f16 v[88:91], v[40:43], v[44:47], v[56:59]\n"
                        ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",
                            "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91");)
#define TILE_VALU(A0, A1, A2, A3)                                                                          \
  "v_min3_f32 v92, |" A0 "|, |" A1 "|, v92\n v_min3_f32 v92, |" A2 "|, |" A3 "|, v92\n"               \
  "v_sub_u32 v93, v48, v52\n v_and_b32 v93, " A0 ", v93\n v_alignbit_b32 v94, v94, v93, 31\n"        \
  "v_sub_u32 v93, v49, v52\n v_and_b32 v93, " A1 ", v93\n v_alignbit_b32 v95, v95, v93, 31\n"        \
  "v_sub_u32 v93, v50, v52\n v_and_b32 v93, " A2 ", v93\n v_alignbit_b32 v96, v96, v93, 31\n"        \
  "v_sub_u32 v93, v51, v52\n v_and_b32 v93, " A3 ", v93\n v_alignbit_b32 v97, v97, v93, 31\n"        \
  "v_cmp_gt_f32 vcc, s46, v92\n"
#define TILE_CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v92", "v93", "v94", "v95", "v96", "v97", "vcc"
    } else if (KIND == 26) {  // one tile step of k_sweep_mfma_f32: fp32 MFMA into one accumulator, 15 VALU on the other
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 v[60:63], v46, v47, v[56:59]\n" TILE_VALU("v64", "v65", "v66", "v67")
                        "v_mfma_f32_16x16x4_f32 v[64:67], v46, v47, v[56:59]\n" TILE_VALU("v60", "v61", "v62", "v63") ::: TILE_CLOB);)
    } else if (KIND == 27) {  // same with the f16 MFMA (K = 32)
      REP8(asm volatile("v_mfma_f32_16x16x32_f16 v[60:63], v[40:43], v[44:47], v[56:59]\n" TILE_VALU("v64", "v65", "v66", "v67")
                        "v_mfma_f32_16x16x32_f16 v[64:67], v[40:43], v[44:47], v[56:59]\n" TILE_VALU("v60", "v61", "v62", "v63") ::: TILE_CLOB);)
    } else if (KIND == 28) {  // the 15 VALU of a tile step without any MFMA
      REP8(asm volatile(TILE_VALU("v64", "v65", "v66", "v67") TILE_VALU("v60", "v61", "v62", "v63") ::: TILE_CLOB);)
    } else if (KIND == 30) {  // test with the sign-bit tail: sub(r2 - rc2'), sub_u32(id order), and, alignbit; no scalar ops
      REP8(asm volatile(
               "v_subrev_f32 v40, s40, v46\n v_subrev_f32 v41, s41, v47\n v_subrev_f32 v42, s42, v48\n"
               "v_mul_f32 v40, v40, v40\n v_mul_f32 v41, v41, v41\n v_mul_f32 v42, v42, v42\n"
               "v_add_f32 v40, v40, v41\n v_add_f32 v40, v40, v42\n"
               "v_subrev_f32 v40, s46, v40\n v_sub_u32 v43, s47, v52\n v_and_b32 v40, v40, v43\n v_alignbit_b32 v44, v44, v40, 31\n"
               ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53");)
    } else if (KIND == 31) {  // the count-and-mask tail of k_sweep_count_masks_f32 today: 2 cmp, s_and, s_bcnt1, s_add, cndmask-or
      REP8(asm volatile(
               "v_subrev_f32 v40, s40, v46\n v_subrev_f32 v41, s41, v47\n v_subrev_f32 v42, s42, v48\n"
               "v_mul_f32 v40, v40, v40\n v_mul_f32 v41, v41, v41\n v_mul_f32 v42, v42, v42\n"
               "v_add_f32 v40, v40, v41\n v_add_f32 v40, v40, v42\n"
               "v_cmp_lt_i32 s[52:53], s47, v52\n v_cmp_nlt_f32 vcc, s46, v40\n"
               "s_and_b64 s[48:49], s[52:53], vcc\n v_or_b32 v43, 4, v44\n v_cndmask_b32 v44, v44, v43, s[48:49]\n"
               "s_bcnt1_i32_b64 s50, s[48:49]\n s_add_i32 s51, s51, s50\n"
               ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "s48", "s49", "s50", "s51", "s52", "s53", "vcc", "scc");)
    } else if (KIND == 13) {  // two tests per pass in packed form: 2 i-particles (SGPR pairs) against the same j
      // 8 packed + 4 compares for 2 tests = 6 instructions per test; fixed registers (timing only)
      REP8(asm volatile(
               "v_pk_add_f32 v[40:41], v[46:47], s[40:41] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
               "v_pk_add_f32 v[42:43], v[48:49], s[42:43] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
               "v_pk_add_f32 v[44:45], v[50:51], s[44:45] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
               "v_pk_mul_f32 v[40:41], v[40:41], v[40:41]\n"
               "v_pk_mul_f32 v[42:43], v[42:43], v[42:43]\n"
               "v_pk_mul_f32 v[44:45], v[44:45], v[44:45]\n"
               "v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n"
               "v_pk_add_f32 v[40:41], v[40:41], v[44:45]\n"
               "v_cmp_lt_i32 s[52:53], s47, v52\n"
               "v_cmp_lt_i32 s[54:55], s56, v52\n"
               "v_cmp_nlt_f32 s[48:49], s46, v40\n"
               "v_cmp_nlt_f32 s[50:51], s46, v41\n"
               ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");)
    }
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
  s += (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(m0 ^ m1 ^ m2 ^ m3);
  if (s == 12345.678f) out[0] = s + lds[threadIdx.x];
}

template <int KIND> int run(const char* name, int per_iter, float* out, unsigned long long* stamps_d) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int iters = 3000;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("%-36s", name);
  // (threads per block, blocks per CU): LDS per block = 160 KiB / blocks pins exactly that many blocks on a CU
  const int cfg[][2] = {{256, 1}, {256, 2}, {256, 4}, {256, 6}, {256, 8}, {1024, 1}, {1024, 2}, {512, 4}};
  for (auto& c : cfg) {
    const int threads = c[0], bpc = c[1];
    const int wps = threads / 256 * bpc;
    const int blocks = cus * bpc;
    const size_t lds = (size_t)(160 * 1024 / bpc) & ~(size_t)1023;
    const int nwaves = blocks * threads / 64;
    std::vector<unsigned long long> st(2 * nwaves);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), lds, 0, out, stamps_d, 100, 1.0f);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), lds, 0, out, stamps_d, iters, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(st.data(), stamps_d, sizeof(unsigned long long) * 2 * nwaves, hipMemcpyDeviceToHost));
    std::vector<double> cyc(nwaves);
    for (int w = 0; w < nwaves; w++) cyc[w] = (double)st[2 * w];
    std::sort(cyc.begin(), cyc.end());
    printf(" | %dx%d(w%d) %5.2f/%5.2f", bpc, threads, wps, cyc[nwaves / 2] / ((double)iters * per_iter * wps), cyc[nwaves - 1] / ((double)iters * per_iter * wps));
    if (getenv("MICRO_NS")) {  // the same in nanoseconds (s_memrealtime: 100 MHz) and the s_memtime rate that implies
      std::vector<double> rt(nwaves);
      for (int w = 0; w < nwaves; w++) rt[w] = (double)st[2 * w + 1] * 10.0;
      std::sort(rt.begin(), rt.end());
      printf(" [%.3f ns, memtime %.0f MHz]", rt[nwaves / 2] / ((double)iters * per_iter * wps), cyc[nwaves / 2] / rt[nwaves / 2] * 1e3);
    }
  }
  printf("\n");
  return 0;
}

int main() {
  float* out;
  unsigned long long* stamps;
  CHK(hipMalloc(&out, 4096));
  CHK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 256 * 8 * 16));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs. cycles = shader cycles (s_memtime) per wave-instruction per SIMD, median/max over waves\n",
         prop.gcnArchName, prop.multiProcessorCount);
  if (getenv("MICRO_OPS")) {  // issue cost of single VALU instructions, 8 independent registers
    run<0>("v_add_f32 v,v", 64, out, stamps);
    run<52>("v_add_f32 s,v", 64, out, stamps);
    run<10>("v_subrev_f32 s,v", 64, out, stamps);
    run<40>("v_mul_f32 v,v", 64, out, stamps);
    run<41>("v_mul_f32 s,v", 64, out, stamps);
    run<53>("v_mul_f32 v,v (square)", 64, out, stamps);
    run<51>("v_mul_legacy_f32", 64, out, stamps);
    run<50>("v_fmac_f32 (VOP2)", 64, out, stamps);
    run<4>("v_fma_f32", 64, out, stamps);
    run<47>("v_max_f32", 64, out, stamps);
    run<42>("v_or_b32", 64, out, stamps);
    run<56>("v_xor_b32", 64, out, stamps);
    run<21>("v_and_b32", 64, out, stamps);
    run<54>("v_add_u32", 64, out, stamps);
    run<22>("v_sub_u32", 64, out, stamps);
    run<48>("v_subrev_u32 s,v", 64, out, stamps);
    run<55>("v_lshlrev_b32", 64, out, stamps);
    run<49>("v_mov_b32", 64, out, stamps);
    run<20>("v_alignbit_b32", 64, out, stamps);
    run<43>("v_lshl_or_b32", 64, out, stamps);
    run<44>("v_bfi_b32", 64, out, stamps);
    run<45>("v_addc_co_u32 vcc", 64, out, stamps);
    run<46>("v_cndmask_b32 vcc", 64, out, stamps);
    run<58>("v_cndmask_b32 sgpr pair", 64, out, stamps);
    run<59>("v_cndmask_b32 vcc (src swapped)", 64, out, stamps);
    run<5>("v_cmp_lt_f32 -> vcc", 64, out, stamps);
    run<6>("v_cmp_lt_f32 -> sgpr pair", 64, out, stamps);
    run<12>("test body unpacked (10 instr) /test", 8, out, stamps);
    run<14>("test + s_and,s_bcnt1,s_add /test", 8, out, stamps);
    run<31>("test + count + mask bits /test", 8, out, stamps);
    run<30>("test + sign-bit tail (12 VALU, 0 SALU) /test", 8, out, stamps);
    return 0;
  }
  if (getenv("MICRO_TAIL")) {  // which hit-recording tail for the VALU sweep
    run<12>("test body unpacked (10 instr) /test", 8, out, stamps);
    run<14>("test + s_and,s_bcnt1,s_add /test", 8, out, stamps);
    run<31>("test + count + mask bits (today) /test", 8, out, stamps);
    run<30>("test + sign-bit tail (12 VALU, 0 SALU) /test", 8, out, stamps);
    return 0;
  }
  if (getenv("MICRO_MFMA")) {  // the instruction mix of k_sweep_mfma_f32 only
    run<22>("v_sub_u32", 64, out, stamps);
    run<21>("v_and_b32", 64, out, stamps);
    run<20>("v_alignbit_b32", 64, out, stamps);
    run<23>("v_min3_f32 |a|,|b|,c", 64, out, stamps);
    run<24>("v_mfma_f32_16x16x4_f32 (indep.)", 64, out, stamps);
    run<25>("v_mfma_f32_16x16x32_f16 (indep.)", 64, out, stamps);
    run<28>("tile step: 15 VALU only /step", 16, out, stamps);
    run<26>("tile step: f32 MFMA + 15 VALU /step", 16, out, stamps);
    run<27>("tile step: f16 MFMA + 15 VALU /step", 16, out, stamps);
    return 0;
  }
  run<0>("v_add_f32", 64, out, stamps);
  run<4>("v_fma_f32", 64, out, stamps);
  run<1>("v_pk_add_f32", 64, out, stamps);
  run<2>("v_pk_mul_f32", 64, out, stamps);
  run<3>("v_pk_fma_f32", 64, out, stamps);
  run<10>("v_subrev_f32 (sgpr src)", 64, out, stamps);
  run<5>("v_cmp_lt_f32 -> vcc", 64, out, stamps);
  run<6>("v_cmp_lt_f32 -> sgpr pair", 64, out, stamps);
  run<7>("v_cmp_lt_i32 sgpr src -> sgpr", 64, out, stamps);
  run<8>("v_add_f64", 64, out, stamps);
  run<9>("v_mul_f64", 64, out, stamps);
  run<11>("test body today (8 instr) /test", 8, out, stamps);
  run<12>("test body unpacked (10 instr) /test", 8, out, stamps);
  run<13>("test body 2i packed (12 instr) /2tests", 8, out, stamps);
  run<14>("test + s_and,s_bcnt1,s_add /test", 8, out, stamps);
  run<15>("test + delayed scalar ops /test", 8, out, stamps);
  run<16>("test + per-lane count (12 VALU) /test", 8, out, stamps);
  return 0;
}

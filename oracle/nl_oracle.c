/*
 * oracle/nl_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's scalar Verlet-list builder
 * (/root/reference/neighlist_cpu.hpp, class NeighList<Vec>) and of the brute-force
 * checker of its harness (/root/reference/make_list.cpp:79-128).  It exists to CHECK the
 * HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  Nothing under md_neighbor_list_amd/ links, imports or calls it, and the product
 * path has no CPU fallback.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 *   (1) the reference class itself, compiled unmodified from /root/reference into
 *       oracle/_ref/ (oracle/ref_driver.cpp; raw array equality with the
 *       -DWITHOUT_LOOP_FUSION variant, canonical equality with the other two), and
 *   (2) the golden vectors under tests/golden/ that were generated from that compiled
 *       reference (oracle/gen_golden.py), which is what runs where /root/reference is absent,
 *   (3) the reference harness's own known answers.
 *
 * Build: see oracle/Makefile.  -ffp-contract=off is mandatory (an FMA changes which pairs sit
 * on the r2 == rc2 boundary in fp32).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NLO_OK 0
#define NLO_ERR_ARG 1
#define NLO_ERR_NOMEM 2
#define NLO_ERR_OUT_OF_BOX 3

#define REAL float
#define SUF(x) x##_f32
#include "nl_oracle_impl.h"
#undef REAL
#undef SUF

#define REAL double
#define SUF(x) x##_f64
#include "nl_oracle_impl.h"
#undef REAL
#undef SUF

void nl_oracle_free(void* p) { free(p); }

static int cmp_i32(const void* a, const void* b) {
  const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}

/* sort_neighlist, make_list.cpp:120-128: ascending partners inside every CSR segment. */
void nl_oracle_canonicalize(int64_t N, const int64_t* key_pointer, int32_t* sorted_list) {
  for (int64_t i = 0; i < N; i++)
    if (key_pointer[i + 1] > key_pointer[i])  /* (an empty list may come as a null pointer: qsort must not see it) */
      qsort(sorted_list + key_pointer[i], (size_t)(key_pointer[i + 1] - key_pointer[i]), sizeof(int32_t), cmp_i32);
}

/* Order-independent pair-set hash used for the known answers in SURVEY.md (Appendix A):
 * h = sum over pairs of mix((u64(i) << 32) | u32(j)), mix(v): v *= 0x9E3779B97F4A7C15; v ^= v >> 29. */
uint64_t nl_oracle_hash(int64_t N, const int64_t* key_pointer, const int32_t* sorted_list) {
  uint64_t h = 0;
  for (int64_t i = 0; i < N; i++)
    for (int64_t p = key_pointer[i]; p < key_pointer[i + 1]; p++) {
      uint64_t v = ((uint64_t)(uint32_t)i << 32) | (uint32_t)sorted_list[p];
      v *= 0x9E3779B97F4A7C15ULL;
      v ^= v >> 29;
      h += v;
    }
  return h;
}

/* Same hash from a GPU-style full transposed list (make_list.cu:178-198 layout
 * list[k*row_stride + i], k < count[i]); only entries j > i are hashed so that it equals the
 * half-list hash. Returns the number of j > i entries through *nhalf. */
uint64_t nl_oracle_hash_transposed(int64_t N, int64_t row_stride, const int32_t* count,
                                   const int32_t* list, int64_t* nhalf) {
  uint64_t h = 0;
  int64_t n = 0;
  for (int64_t i = 0; i < N; i++)
    for (int32_t k = 0; k < count[i]; k++) {
      const int32_t j = list[(int64_t)k * row_stride + i];
      if (j <= i) continue;
      uint64_t v = ((uint64_t)(uint32_t)i << 32) | (uint32_t)j;
      v *= 0x9E3779B97F4A7C15ULL;
      v ^= v >> 29;
      h += v;
      n++;
    }
  if (nhalf) *nhalf = n;
  return h;
}

// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver around the REFERENCE's own classes, compiled from the sources where
// they lie (/root/reference/neighlist_cpu*.hpp, included with -I/root/reference, never copied)
// into oracle/_ref/*.so by oracle/Makefile.  It is used to (a) pin the C restatement in
// oracle/nl_oracle.c, (b) generate tests/golden/*, (c) time the reference CPU path for
// bench.py's cpu_baseline ("kind": "reference").  It mirrors what make_list.cpp:132-163 does:
// construct, Initialize(N) once, MakeNeighList(q, N) per build, read the accessors.
//
// One .so per reference variant, selected exactly as the reference Makefile does, by -D:
//   scalar: -DWITHOUT_LOOP_FUSION | -DLOOP_FUSION | -DLOOP_FUSION_SWP   (neighlist_cpu.hpp:426-432)
//   AVX2  : -DUSE_AVX2 -DUSE4x1 (etc.)                                   (neighlist_cpu_avx2.hpp:889-899)
//   AVX512: -DUSE_AVX512 -DUSE8x1 | -DUSE1x8                             (neighlist_cpu_avx512.hpp)
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <new>
#include <vector>

#if defined(USE_AVX512)
#include "neighlist_cpu_avx512.hpp"
#define NL_SIMD 1
template <class V> using RefList = NeighListAVX512<V>;
#elif defined(USE_AVX2)
#include "neighlist_cpu_avx2.hpp"
#define NL_SIMD 1
template <class V> using RefList = NeighListAVX2<V>;
#else
#include "neighlist_cpu.hpp"
#define NL_SIMD 0
template <class V> using RefList = NeighList<V>;
#endif

namespace {

#if NL_SIMD
// make_list.cpp:26-32: the SIMD builds use a 4-wide Vec.
template <class T> struct VecT { T x, y, z, w; };
#else
template <class T> struct VecT { T x, y, z; };
#endif

// The SIMD classes use aligned loads on q (32-B Vec) and on their own shfl_table_ member
// (neighlist_cpu_avx2.hpp:35,502; neighlist_cpu_avx512.hpp:36): icpc-built objects happened to be
// aligned, g++'s are not.  Instead of touching the reference we place the object so that the table is
// 64-B aligned: construct once, locate the table by its last row, then construct for real at the
// shifted address.
template <class L> struct Placed {
  void* raw = nullptr;
  L* obj = nullptr;
  template <class... A> void make(size_t table_bytes, const void* last_row, size_t row_bytes, A... a) {
    raw = nullptr;
    if (posix_memalign(&raw, 64, sizeof(L) + 128)) std::abort();
    size_t shift = 0;
    if (table_bytes) {
      L* probe = new (raw) L(a...);
      probe->Initialize(8);  // the table is filled by Initialize (neighlist_cpu_avx2.hpp:877)
      const unsigned char* b = reinterpret_cast<const unsigned char*>(probe);
      size_t off = sizeof(L);
      for (size_t o = 0; o + row_bytes <= sizeof(L); o += 4)
        if (!std::memcmp(b + o, last_row, row_bytes)) off = o;  // last match = last row
      probe->~L();
      if (off == sizeof(L)) std::abort();
      const size_t table_off = off + row_bytes - table_bytes;
      shift = (64 - table_off % 64) % 64;
    }
    obj = new (static_cast<unsigned char*>(raw) + shift) L(a...);
  }
  ~Placed() {
    if (obj) obj->~L();
    free(raw);
  }
};

template <class T>
int run(const T* q, int32_t stride, int32_t N, double rc, double Lx, double Ly, double Lz, int32_t loops,
        int32_t* nop, int32_t* kp, int32_t* list, int64_t list_cap, int32_t* npairs, double* seconds) {
  typedef VecT<T> V;
  V* v = nullptr;
  if (posix_memalign(reinterpret_cast<void**>(&v), 64, sizeof(V) * (size_t)(N > 0 ? N : 1))) return 2;
  for (int32_t i = 0; i < N; i++) {
    v[i].x = q[(size_t)i * stride + 0];
    v[i].y = q[(size_t)i * stride + 1];
    v[i].z = q[(size_t)i * stride + 2];
#if NL_SIMD
    v[i].w = 0;
#endif
  }
  Placed<RefList<V>> pl;
#if defined(USE_AVX512)
  const int64_t last[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  pl.make(sizeof(int64_t) * 256 * 8, last, sizeof(last), rc, Lx, Ly, Lz);
#elif defined(USE_AVX2)
  const int32_t last[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  pl.make(sizeof(int32_t) * 16 * 8, last, sizeof(last), rc, Lx, Ly, Lz);
#else
  pl.make(0, nullptr, 0, rc, Lx, Ly, Lz);
#endif
  RefList<V>& nl = *pl.obj;
  nl.Initialize(N);                                       // make_list.cpp:151
  const auto t0 = std::chrono::steady_clock::now();
  for (int32_t l = 0; l < loops; l++) nl.MakeNeighList(v, N);  // make_list.cpp:153-155
  const auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  const int32_t P = nl.number_of_pairs();                 // make_list.cpp:160-163
  *npairs = P;
  if (nop) std::memcpy(nop, nl.number_of_partners(), sizeof(int32_t) * (size_t)N);
  if (kp) std::memcpy(kp, nl.key_pointer(), sizeof(int32_t) * ((size_t)N + 1));
  int ret = 0;
  if (list) {
    if (P > list_cap) ret = 4;
    else std::memcpy(list, nl.sorted_list(), sizeof(int32_t) * (size_t)P);
  }
  free(v);
  return ret;
}

}  // namespace

extern "C" {

// Returns 0 on success.  q: N particles, `stride` scalars apart.  The reference sizes its pair buffers as
// 100*N (neighlist_cpu.hpp:37,76-78) and does not check: callers keep the mean half count well below 100.
// The SIMD classes are fp64-only, so those builds export only the f64 entry point.
#if !NL_SIMD
int nl_ref_build_f32(const float* q, int32_t stride, int32_t N, double rc, double Lx, double Ly, double Lz,
                     int32_t loops, int32_t* nop, int32_t* kp, int32_t* list, int64_t list_cap, int32_t* npairs,
                     double* seconds) {
  return run<float>(q, stride, N, rc, Lx, Ly, Lz, loops, nop, kp, list, list_cap, npairs, seconds);
}
#endif
int nl_ref_build_f64(const double* q, int32_t stride, int32_t N, double rc, double Lx, double Ly, double Lz,
                     int32_t loops, int32_t* nop, int32_t* kp, int32_t* list, int64_t list_cap, int32_t* npairs,
                     double* seconds) {
  return run<double>(q, stride, N, rc, Lx, Ly, Lz, loops, nop, kp, list, list_cap, npairs, seconds);
}
}

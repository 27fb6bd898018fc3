// neighlist_cpu.hpp -- the reference's scalar CPU class surface, NeighList<Vec> (reference neighlist_cpu.hpp:380-463),
// backed by the HIP library.  It has the same file name, class name and members as the reference header on purpose:
// the reference's CPU harness (make_list.cpp:148-163) compiles UNCHANGED against this file and then checks the GPU
// result with its own brute-force test (make_list.cpp:166-222).  oracle/Makefile builds exactly that
// (oracle/_ref/make_list_dropin) and tests/test_dropin.py runs it on the GPU.
//
//   NeighList<Vec>(double rc, double Lx, double Ly, double Lz)      neighlist_cpu.hpp:380-395
//   void     Initialize(int32_t N)                                   :408-415
//   void     MakeNeighList(Vec* q, int32_t N)      host pointer      :417-435
//   int32_t  number_of_pairs() const                                 :437-439
//   int32_t* sorted_list() / key_pointer() / number_of_partners()    :441-463   (host pointers)
//
// Vec is {Dtype x, y, z} (make_list.cpp:26-32) or a 4-wide {x, y, z, w}; Dtype is deduced from Vec::x.
// MakeNeighList copies the positions to the device and enqueues the build; the first accessor after it waits for
// the build and copies the list back.  Errors print the status and exit, as the reference's harness would.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "nl_hip.h"

#ifndef NL_SHIM_CHECK
#define NL_SHIM_CHECK(call)                                                                           \
  do {                                                                                                \
    const int nl_rc_ = (call);                                                                        \
    if (nl_rc_ != NL_OK) {                                                                            \
      std::fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, nl_rc_, nl_status_string(nl_rc_)); \
      std::exit(EXIT_FAILURE);                                                                        \
    }                                                                                                 \
  } while (0)
#endif

template <typename Vec> class NeighList {
  typedef decltype(Vec().x) Dtype;
  static_assert(std::is_same<Dtype, float>::value || std::is_same<Dtype, double>::value, "positions are float or double");
  static_assert(sizeof(Vec) == 3 * sizeof(Dtype) || sizeof(Vec) == 4 * sizeof(Dtype), "Vec is {x,y,z[,w]}");

  nl_handle_t h_ = nullptr;
  void* q_dev_ = nullptr;
  int32_t n_max_ = 0, n_ = 0;
  mutable bool fetched_ = false;
  mutable int32_t number_of_pairs_ = 0;
  mutable std::vector<int32_t> sorted_list_, key_pointer_, number_of_partners_;

  void Fetch() const {
    if (fetched_) return;
    const int32_t *kp = nullptr, *sl = nullptr, *nop = nullptr;
    int64_t p = 0;
    NL_SHIM_CHECK(nl_get_half_csr(h_, &kp, &sl, &nop, &p));
    number_of_pairs_ = static_cast<int32_t>(p);
    sorted_list_.resize(static_cast<size_t>(p) + 1);
    key_pointer_.resize(static_cast<size_t>(n_) + 1);
    number_of_partners_.resize(static_cast<size_t>(n_) + 1);
    NL_SHIM_CHECK(nl_buf_d2h(sorted_list_.data(), sl, sizeof(int32_t) * static_cast<size_t>(p)));
    NL_SHIM_CHECK(nl_buf_d2h(key_pointer_.data(), kp, sizeof(int32_t) * (static_cast<size_t>(n_) + 1)));
    NL_SHIM_CHECK(nl_buf_d2h(number_of_partners_.data(), nop, sizeof(int32_t) * static_cast<size_t>(n_)));
    fetched_ = true;
  }

 public:
  NeighList(const double search_length, const double Lx, const double Ly, const double Lz) {
    NL_SHIM_CHECK(nl_create(&h_, std::is_same<Dtype, float>::value ? NL_F32 : NL_F64, search_length, Lx, Ly, Lz, -1));
  }
  ~NeighList() {
    if (q_dev_) (void)nl_buf_free(q_dev_, nullptr);
    if (h_) (void)nl_destroy(h_);
  }
  const NeighList<Vec>& operator=(const NeighList<Vec>&) = delete;  // neighlist_cpu.hpp:400-406
  NeighList(const NeighList<Vec>&) = delete;
  NeighList<Vec>& operator=(NeighList<Vec>&&) = delete;
  NeighList(NeighList<Vec>&&) = delete;

  void Initialize(const int32_t particle_number) {
    NL_SHIM_CHECK(nl_initialize(h_, particle_number));
    if (q_dev_) (void)nl_buf_free(q_dev_, nullptr);
    NL_SHIM_CHECK(nl_buf_alloc(&q_dev_, nullptr, sizeof(Vec) * static_cast<size_t>(particle_number)));
    n_max_ = particle_number;
  }

  void MakeNeighList(Vec* q, const int32_t particle_number) {
    if (particle_number > n_max_) {
      std::fprintf(stderr, "NeighList::MakeNeighList: %d particles, initialized for %d\n", particle_number, n_max_);
      std::exit(EXIT_FAILURE);
    }
    n_ = particle_number;
    fetched_ = false;
    NL_SHIM_CHECK(nl_buf_h2d(q_dev_, q, sizeof(Vec) * static_cast<size_t>(particle_number)));
    NL_SHIM_CHECK(nl_make_list(h_, q_dev_, static_cast<int32_t>(sizeof(Vec) / sizeof(Dtype)), particle_number, nullptr, 0));
  }

  int32_t number_of_pairs() const {
    Fetch();
    return number_of_pairs_;
  }
  int32_t* sorted_list() {
    Fetch();
    return sorted_list_.data();
  }
  const int32_t* sorted_list() const {
    Fetch();
    return sorted_list_.data();
  }
  int32_t* key_pointer() {
    Fetch();
    return key_pointer_.data();
  }
  const int32_t* key_pointer() const {
    Fetch();
    return key_pointer_.data();
  }
  int32_t* number_of_partners() {
    Fetch();
    return number_of_partners_.data();
  }
  const int32_t* number_of_partners() const {
    Fetch();
    return number_of_partners_.data();
  }
};

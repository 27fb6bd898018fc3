// neighlist_gpu.hpp -- header-only C++ surface of the reference's GPU builder, on top of the C ABI (nl_hip.h).
//
// Source-compatible with what the reference's GPU harness touches (make_list.cu:113-142):
//   cuda_ptr<T>                 paired device + pinned-host buffer          (reference cuda_ptr.cuh:11-112)
//   NeighListGPU<Vec, Dtype>    ctor / Initialize / MakeNeighList / neigh_list / number_of_partners /
//                               number_of_pairs                            (reference neighlist_gpu.hpp:43-488)
// plus the scalar CPU class's accessors (key_pointer, sorted_list, half counts: reference neighlist_cpu.hpp:437-463),
// which are the native output of the HIP path.  Nothing here includes a CUDA or HIP header: the only dependency is
// libnl_hip.so.  The type is called cuda_ptr because that is the name the harness spells; it is a HIP buffer.
//
// Error behaviour follows the reference at this level: it aborts through checkCudaErrors / std::exit(1)
// (device_util.cuh:41-54), so a failing call prints the status and exits.  Callers that want status codes use the
// C ABI directly.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "nl_hip.h"

#define NL_SHIM_CHECK(call)                                                                           \
  do {                                                                                                \
    const int nl_rc_ = (call);                                                                        \
    if (nl_rc_ != NL_OK) {                                                                            \
      std::fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, nl_rc_, nl_status_string(nl_rc_)); \
      std::exit(EXIT_FAILURE);                                                                        \
    }                                                                                                 \
  } while (0)

template <typename T> struct cuda_ptr {
  T* dev_ptr = nullptr;   // cuda_ptr.cuh:13
  T* host_ptr = nullptr;  // cuda_ptr.cuh:14
  int size = -1;          // cuda_ptr.cuh:15

  cuda_ptr() {}
  ~cuda_ptr() { deallocate(); }
  cuda_ptr(const cuda_ptr&) = delete;                    // cuda_ptr.cuh:23-25
  const cuda_ptr& operator=(const cuda_ptr&) = delete;
  cuda_ptr& operator=(cuda_ptr&& o) noexcept {           // cuda_ptr.cuh:27-35
    if (this != &o) {
      deallocate();
      dev_ptr = o.dev_ptr, host_ptr = o.host_ptr, size = o.size, owns_dev_ = o.owns_dev_;
      o.dev_ptr = nullptr, o.host_ptr = nullptr, o.size = -1;
    }
    return *this;
  }
  cuda_ptr(cuda_ptr&& o) noexcept { *this = std::move(o); }

  void allocate(const int size_) {  // cuda_ptr.cuh:40-45
    deallocate();
    size = size_;
    void *d = nullptr, *h = nullptr;
    NL_SHIM_CHECK(nl_buf_alloc(&d, &h, sizeof(T) * static_cast<size_t>(size_ > 0 ? size_ : 0)));
    dev_ptr = static_cast<T*>(d), host_ptr = static_cast<T*>(h), owns_dev_ = true;
  }
  void host2dev(const int beg, const int count) {  // cuda_ptr.cuh:47-52
    NL_SHIM_CHECK(nl_buf_h2d(dev_ptr + beg, host_ptr + beg, sizeof(T) * static_cast<size_t>(count)));
  }
  void host2dev() { host2dev(0, size); }
  void dev2host(const int beg, const int count) {  // cuda_ptr.cuh:62-67
    NL_SHIM_CHECK(nl_buf_d2h(host_ptr + beg, dev_ptr + beg, sizeof(T) * static_cast<size_t>(count)));
  }
  void dev2host() { dev2host(0, size); }
  void set_val(const T val) { set_val(0, size, val); }  // cuda_ptr.cuh:79-82
  void set_val(const int beg, const int count, const T val) {  // cuda_ptr.cuh:84-89: fills host AND device
    std::fill(host_ptr + beg, host_ptr + beg + count, val);
    static_assert(sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 16 || sizeof(T) == 32, "4/8-byte scalars or 4-vectors");
    if (sizeof(T) == 4) {
      std::uint32_t p;
      std::copy_n(reinterpret_cast<const unsigned char*>(&val), 4, reinterpret_cast<unsigned char*>(&p));
      NL_SHIM_CHECK(nl_buf_fill32(dev_ptr + beg, p, static_cast<size_t>(count)));
    } else if (sizeof(T) == 8) {
      std::uint64_t p;
      std::copy_n(reinterpret_cast<const unsigned char*>(&val), 8, reinterpret_cast<unsigned char*>(&p));
      NL_SHIM_CHECK(nl_buf_fill64(dev_ptr + beg, p, static_cast<size_t>(count)));
    } else {
      host2dev(beg, count);  // vectors: push the host fill
    }
  }
  const T& operator[](const int i) const { return host_ptr[i]; }  // host view, cuda_ptr.cuh:91-97
  T& operator[](const int i) { return host_ptr[i]; }
  operator const T*() const { return dev_ptr; }  // device pointer, cuda_ptr.cuh:99-105
  operator T*() { return dev_ptr; }

  // (not in the reference) view of device memory owned by the builder, with an own pinned host mirror
  void borrow(T* dev, const int size_) {
    if (!owns_dev_ && host_ptr && size == size_) {
      dev_ptr = dev;
      return;
    }
    deallocate();
    void* h = nullptr;
    NL_SHIM_CHECK(nl_buf_alloc(nullptr, &h, sizeof(T) * static_cast<size_t>(size_ > 0 ? size_ : 0)));
    dev_ptr = dev, host_ptr = static_cast<T*>(h), size = size_, owns_dev_ = false;
  }

 private:
  bool owns_dev_ = true;
  void deallocate() {  // cuda_ptr.cuh:107-111
    if (dev_ptr || host_ptr) (void)nl_buf_free(owns_dev_ ? dev_ptr : nullptr, host_ptr);
    dev_ptr = nullptr, host_ptr = nullptr;
  }
};

template <typename Vec, typename Dtype> class NeighListGPU {
  static_assert(std::is_same<Dtype, float>::value || std::is_same<Dtype, double>::value, "Dtype is float or double");
  static_assert(sizeof(Vec) == 4 * sizeof(Dtype) || sizeof(Vec) == 3 * sizeof(Dtype), "Vec is {x,y,z[,w]} of Dtype");

  nl_handle_t h_ = nullptr;
  int32_t n_ = 0;
  cuda_ptr<int32_t> transposed_list_, number_of_partners_;        // neighlist_gpu.hpp:59
  cuda_ptr<int32_t> key_pointer_, sorted_list_, half_partners_;   // the CPU class's outputs

 public:
  NeighListGPU(const Dtype search_length, const Dtype Lx, const Dtype Ly, const Dtype Lz) {  // neighlist_gpu.hpp:236-255
    NL_SHIM_CHECK(nl_create(&h_, std::is_same<Dtype, float>::value ? NL_F32 : NL_F64, search_length, Lx, Ly, Lz, -1));
    // this class's contract is the FULL list (every j != i within the cut-off, kernel_impl.cuh:24-33)
    NL_SHIM_CHECK(nl_set_list_kind(h_, NL_LIST_FULL));
  }
  ~NeighListGPU() {
    if (h_) (void)nl_destroy(h_);
  }
  NeighListGPU(const NeighListGPU&) = delete;  // neighlist_gpu.hpp:260-266: neither copyable nor movable
  NeighListGPU& operator=(const NeighListGPU&) = delete;
  NeighListGPU(NeighListGPU&&) = delete;
  NeighListGPU& operator=(NeighListGPU&&) = delete;

  void Initialize(const int32_t particle_number) { NL_SHIM_CHECK(nl_initialize(h_, particle_number)); }  // :268-287

  // Extension: build the scalar CPU class's half list instead (key_pointer / sorted_list / half_number_of_partners
  // below; neigh_list() is then derived from it with scattered writes, about ten times slower).
  void UseHalfList(const bool half = true) { NL_SHIM_CHECK(nl_set_list_kind(h_, half ? NL_LIST_HALF : NL_LIST_FULL)); }

  // neighlist_gpu.hpp:289-293.  tblock_size / smem_hei chose among the reference's CUDA variants; ignored.
  void MakeNeighList(cuda_ptr<Vec>& q, const int32_t particle_number, const bool sync = true, int32_t tblock_size = 128,
                     const int32_t smem_hei = 7) {
    (void)tblock_size, (void)smem_hei;
    n_ = particle_number;
    NL_SHIM_CHECK(nl_make_list(h_, q.dev_ptr, static_cast<int32_t>(sizeof(Vec) / sizeof(Dtype)), particle_number, nullptr,
                               sync ? 1 : 0));
  }
  void Synchronize() { NL_SHIM_CHECK(nl_synchronize(h_)); }  // the harness's cudaDeviceSynchronize, make_list.cu:128

  cuda_ptr<int32_t>& neigh_list() {  // neighlist_gpu.hpp:468-474: full list, [k * N + i], -1 padded
    fetch_transposed();
    return transposed_list_;
  }
  cuda_ptr<int32_t>& number_of_partners() {  // neighlist_gpu.hpp:476-482: full counts
    fetch_transposed();
    return number_of_partners_;
  }
  int32_t number_of_pairs() const {  // neighlist_gpu.hpp:484-487: sum of the full counts = 2 x half pairs
    int64_t p = 0;
    NL_SHIM_CHECK(nl_number_of_pairs(h_, &p));
    return static_cast<int32_t>(2 * p);
  }

  // --- the scalar CPU class's outputs (neighlist_cpu.hpp:437-463), device-resident; after UseHalfList()
  int64_t half_number_of_pairs() const {
    int64_t p = 0;
    NL_SHIM_CHECK(nl_number_of_pairs(h_, &p));
    return p;
  }
  cuda_ptr<int32_t>& key_pointer() {
    fetch_half();
    return key_pointer_;
  }
  cuda_ptr<int32_t>& sorted_list() {
    fetch_half();
    return sorted_list_;
  }
  cuda_ptr<int32_t>& half_number_of_partners() {
    fetch_half();
    return half_partners_;
  }
  nl_handle_t handle() { return h_; }

 private:
  void fetch_transposed() {
    const int32_t *lst = nullptr, *cnt = nullptr;
    int64_t stride = 0;
    int32_t rows = 0;
    NL_SHIM_CHECK(nl_get_full_transposed(h_, &lst, &cnt, &stride, &rows));
    transposed_list_.borrow(const_cast<int32_t*>(lst), static_cast<int>(stride * std::max(rows, 1)));
    number_of_partners_.borrow(const_cast<int32_t*>(cnt), static_cast<int>(stride));
  }
  void fetch_half() {
    const int32_t *kp = nullptr, *sl = nullptr, *nop = nullptr;
    int64_t p = 0;
    NL_SHIM_CHECK(nl_get_half_csr(h_, &kp, &sl, &nop, &p));
    key_pointer_.borrow(const_cast<int32_t*>(kp), n_ + 1);
    sorted_list_.borrow(const_cast<int32_t*>(sl), static_cast<int>(p));
    half_partners_.borrow(const_cast<int32_t*>(nop), n_);
  }
};

// tests/sanitize/abi_stub.cpp -- TEST INFRASTRUCTURE: a host-memory stand-in for the part of the C ABI (include/nl_hip.h)
// that the two header-only shims (include/neighlist_cpu.hpp, include/neighlist_gpu.hpp) call, so that the shims' buffer
// handling can run under AddressSanitizer / UBSan on a machine without a GPU (GPU ASan is not
// available on this pool).  "Device" buffers are plain heap blocks; the list comes from the CPU oracle's restatement
// (oracle/nl_oracle.c, itself compiled with the sanitizers).  Never linked into the product library.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nl_hip.h"

extern "C" {
int nl_oracle_build_f32(const float*, int32_t, int64_t, double, double, double, double, int32_t*, int64_t*, int32_t**, int64_t*);
int nl_oracle_build_f64(const double*, int32_t, int64_t, double, double, double, double, int32_t*, int64_t*, int32_t**, int64_t*);
void nl_oracle_free(void*);
}

struct nl_handle_s {
  int dtype;
  double rc, L[3];
  int32_t n_max = -1, n = 0;
  int kind = NL_LIST_HALF;
  bool built = false;
  std::vector<int32_t> kp, list, cnt;        // the list of the last build in the requested kind (CSR)
  std::vector<int32_t> t_list, t_cnt;        // transposed full list
  int64_t npairs = 0;
  int32_t t_max = 0;
};

extern "C" {

const char* nl_status_string(int s) { return s == NL_OK ? "ok" : "error (stub)"; }

int nl_create(nl_handle_t* out, int dtype, double rc, double Lx, double Ly, double Lz, int) {
  if (!out || !(rc > 0)) return NL_ERR_ARG;
  for (double L : {Lx, Ly, Lz})
    if ((int)(L / rc) < 3) return NL_ERR_MESH;
  nl_handle_t h = new nl_handle_s();
  h->dtype = dtype, h->rc = rc, h->L[0] = Lx, h->L[1] = Ly, h->L[2] = Lz;
  *out = h;
  return NL_OK;
}
int nl_destroy(nl_handle_t h) {
  delete h;
  return NL_OK;
}
int nl_initialize(nl_handle_t h, int32_t n_max) {
  if (!h || n_max < 0) return NL_ERR_ARG;
  h->n_max = n_max;
  return NL_OK;
}
int nl_set_list_kind(nl_handle_t h, int kind) {
  h->kind = kind, h->built = false;
  return NL_OK;
}
int nl_synchronize(nl_handle_t h) { return h->built ? NL_OK : NL_ERR_STATE; }

int nl_make_list(nl_handle_t h, const void* q, int32_t stride, int32_t n, void*, int) {
  if (!h || h->n_max < 0) return NL_ERR_STATE;
  if (n > h->n_max || (stride != 3 && stride != 4)) return NL_ERR_ARG;
  std::vector<int32_t> nop((size_t)n + 1);
  std::vector<int64_t> kp((size_t)n + 1);
  int32_t* sl = nullptr;
  int64_t P = 0;
  const int rc = h->dtype == NL_F32
                     ? nl_oracle_build_f32(static_cast<const float*>(q), stride, n, h->rc, h->L[0], h->L[1], h->L[2], nop.data(), kp.data(), &sl, &P)
                     : nl_oracle_build_f64(static_cast<const double*>(q), stride, n, h->rc, h->L[0], h->L[1], h->L[2], nop.data(), kp.data(), &sl, &P);
  if (rc) return rc == 3 ? NL_ERR_OUT_OF_BOX : NL_ERR_ARG;
  h->n = n, h->npairs = P;
  // full counts / transposed list from the half list
  h->t_cnt.assign((size_t)n, 0);
  for (int32_t i = 0; i < n; i++)
    for (int64_t p = kp[i]; p < kp[i + 1]; p++) h->t_cnt[i]++, h->t_cnt[sl[p]]++;
  h->t_max = 0;
  for (int32_t i = 0; i < n; i++) h->t_max = h->t_cnt[i] > h->t_max ? h->t_cnt[i] : h->t_max;
  const int32_t rows = h->t_max > 0 ? h->t_max : 1;
  h->t_list.assign((size_t)rows * (size_t)(n > 0 ? n : 1), -1);
  std::vector<int32_t> cur((size_t)n, 0);
  for (int32_t i = 0; i < n; i++)
    for (int64_t p = kp[i]; p < kp[i + 1]; p++) {
      const int32_t j = sl[p];
      h->t_list[(size_t)cur[i]++ * n + i] = j;
      h->t_list[(size_t)cur[j]++ * n + j] = i;
    }
  if (h->kind == NL_LIST_HALF) {
    h->kp.resize((size_t)n + 1);
    for (int32_t i = 0; i <= n; i++) h->kp[i] = (int32_t)kp[i];
    h->list.assign(sl, sl + P);
    h->cnt.assign(nop.begin(), nop.begin() + n);
  } else {
    h->kp.assign((size_t)n + 1, 0);
    for (int32_t i = 0; i < n; i++) h->kp[i + 1] = h->kp[i] + h->t_cnt[i];
    h->list.resize((size_t)h->kp[n]);
    for (int32_t i = 0; i < n; i++)
      for (int32_t k = 0; k < h->t_cnt[i]; k++) h->list[(size_t)h->kp[i] + k] = h->t_list[(size_t)k * n + i];
    h->cnt = h->t_cnt;
  }
  nl_oracle_free(sl);
  h->built = true;
  return NL_OK;
}

int nl_get_half_csr(nl_handle_t h, const int32_t** kp, const int32_t** sl, const int32_t** nop, int64_t* np) {
  if (!h->built) return NL_ERR_STATE;
  if (h->kind != NL_LIST_HALF) return NL_ERR_STATE;
  if (kp) *kp = h->kp.data();
  if (sl) *sl = h->list.data();
  if (nop) *nop = h->cnt.data();
  if (np) *np = h->npairs;
  return NL_OK;
}
int nl_get_full_transposed(nl_handle_t h, const int32_t** lst, const int32_t** cnt, int64_t* stride, int32_t* mx) {
  if (!h->built) return NL_ERR_STATE;
  if (lst) *lst = h->t_list.data();
  if (cnt) *cnt = h->t_cnt.data();
  if (stride) *stride = h->n;
  if (mx) *mx = h->t_max;
  return NL_OK;
}
int nl_number_of_pairs(nl_handle_t h, int64_t* np) {
  if (!h->built) return NL_ERR_STATE;
  *np = h->npairs;
  return NL_OK;
}

int nl_buf_alloc(void** dev, void** host, size_t bytes) {
  if (dev) *dev = malloc(bytes ? bytes : 1);   // exact sizes: an overrun by the shim is an ASan report
  if (host) *host = malloc(bytes ? bytes : 1);
  return NL_OK;
}
int nl_buf_free(void* dev, void* host) {
  free(dev);
  free(host);
  return NL_OK;
}
int nl_buf_h2d(void* dev, const void* host, size_t bytes) {
  if (bytes) memcpy(dev, host, bytes);
  return NL_OK;
}
int nl_buf_d2h(void* host, const void* dev, size_t bytes) {
  if (bytes) memcpy(host, dev, bytes);
  return NL_OK;
}
int nl_buf_fill32(void* dev, uint32_t pattern, size_t count) {
  for (size_t i = 0; i < count; i++) memcpy(static_cast<char*>(dev) + 4 * i, &pattern, 4);
  return NL_OK;
}
int nl_buf_fill64(void* dev, uint64_t pattern, size_t count) {
  for (size_t i = 0; i < count; i++) memcpy(static_cast<char*>(dev) + 8 * i, &pattern, 8);
  return NL_OK;
}
}  // extern "C"

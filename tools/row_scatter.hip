// tools/row_scatter.hip -- floor of the expansion's output: one million rows of ~73 int32 entries (290 bytes), contiguous per
// row, written at their CSR places in the order the cell-sorted expansion meets them (a random permutation of the rows)
// or in memory order.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/row_scatter tools/row_scatter.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

// a wave writes `rows_per_wave` consecutive rows of `order`, 4 rows per group of stores
template <int MODE>  // 0: one store instruction per 64-entry chunk of a row; 1: the four rows back to back (full stores)
__global__ void __launch_bounds__(128) k(const int* __restrict__ order, const int* __restrict__ kp, int n, int rows_per_wave, int* __restrict__ list) {
  const int lane = threadIdx.x & 63, w = blockIdx.x * 2 + (threadIdx.x >> 6);
  const int r0 = w * rows_per_wave, r1 = min(r0 + rows_per_wave, n);
  for (int r = r0; r < r1; r += 4) {
    int base[4], len[4];
    for (int p = 0; p < 4; p++) {
      const int row = order[min(r + p, r1 - 1)];
      base[p] = kp[row], len[p] = r + p < r1 ? kp[row + 1] - kp[row] : 0;
    }
    if (MODE == 0) {
      for (int e = lane; e - lane < 128; e += 64)
        for (int p = 0; p < 4; p++)
          if (e < len[p]) list[base[p] + e] = e;
    } else if (MODE == 2) {  // 16 bytes per lane: lane t carries quad t of the four rows' quads, the 1-3 entries behind a row's last quad go as dwords
      const int q0 = len[0] >> 2, q1 = q0 + (len[1] >> 2), q2 = q1 + (len[2] >> 2), q3 = q2 + (len[3] >> 2);
      for (int t = lane; t - lane < q3; t += 64) {
        const int j = t - (t >= q0 ? q0 : 0) - (t >= q1 ? q1 - q0 : 0) - (t >= q2 ? q2 - q1 : 0);
        const int b = base[0] + (t >= q0 ? base[1] - base[0] : 0) + (t >= q1 ? base[2] - base[1] : 0) + (t >= q2 ? base[3] - base[2] : 0);
        if (t < q3) *reinterpret_cast<int4*>(list + b + 4 * j) = make_int4(j, j, j, j);
      }
      const int p = lane >> 2, k = lane & 3;  // lanes 0..15: the tails
      if (lane < 16) {
        const int l = p == 0 ? len[0] : p == 1 ? len[1] : p == 2 ? len[2] : len[3], bb = p == 0 ? base[0] : p == 1 ? base[1] : p == 2 ? base[2] : base[3];
        const int e = (l & ~3) + k;
        if (e < l) list[bb + e] = e;
      }
    } else {
      const int c0 = len[0], c1 = c0 + len[1], c2 = c1 + len[2], c3 = c2 + len[3];
      for (int t = lane; t - lane < c3; t += 64) {
        const int e = t - (t >= c0 ? c0 : 0) - (t >= c1 ? c1 - c0 : 0) - (t >= c2 ? c2 - c1 : 0);
        const int b = base[0] + (t >= c0 ? base[1] - base[0] : 0) + (t >= c1 ? base[2] - base[1] : 0) + (t >= c2 ? base[3] - base[2] : 0);
        if (t < c3) {
          if (MODE == 3) __builtin_nontemporal_store(e, list + b + e);
          else list[b + e] = e;
        }
      }
    }
  }
}

int main() {
  const int n = 1 << 20;
  std::mt19937 rng(5);
  std::vector<int> len(n), kp(n + 1), ident(n), perm(n);
  std::poisson_distribution<int> pd(72.6);
  for (int i = 0; i < n; i++) len[i] = pd(rng);
  kp[0] = 0;
  for (int i = 0; i < n; i++) kp[i + 1] = kp[i] + len[i];
  std::iota(ident.begin(), ident.end(), 0);
  perm = ident;
  std::shuffle(perm.begin(), perm.end(), rng);
  int *d_order, *d_kp, *d_list;
  hipMalloc(&d_order, 4 * n), hipMalloc(&d_kp, 4 * (n + 1)), hipMalloc(&d_list, 4 * (size_t)kp[n] + 1024);
  hipMemcpy(d_kp, kp.data(), 4 * (n + 1), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int which = 0; which < 2; which++) {
    hipMemcpy(d_order, which ? perm.data() : ident.data(), 4 * n, hipMemcpyHostToDevice);
    for (int mode = 1; mode < 4; mode++)
      for (int rpw : {20, 40}) {
        const int grid = (n + 2 * rpw - 1) / (2 * rpw);
        auto launch = [&]() {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(128), 0, 0, d_order, d_kp, n, rpw, d_list);
          else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(128), 0, 0, d_order, d_kp, n, rpw, d_list);
          else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(128), 0, 0, d_order, d_kp, n, rpw, d_list);
          else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(128), 0, 0, d_order, d_kp, n, rpw, d_list);
        };
        launch();
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        printf("%-8s rows, %s, %3d rows per wave: %7.1f us  (%.2f TB/s of list)\n", which ? "permuted" : "in order",
               mode == 3 ? "back to back, nt   " : mode == 2 ? "16 bytes per lane  " : mode ? "rows back to back " : "a store per chunk ", rpw, ms * 1e3, 4.0 * kp[n] / (ms * 1e-3) / 1e12);
      }
  }
  return 0;
}

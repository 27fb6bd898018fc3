// tools/mfma_check.hip -- checks the operand/result lane layout of v_mfma_f32_16x16x4_f32 assumed by nl_sweep_mfma.hpp:
// A[i][k]: lane = i + 16k;  B[k][j]: lane = j + 16k;  D[i][j]: lane = j + 16*(i/4), register i%4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  // A[i][k] row-major 16x4, B[k][j] 4x16
  float a = A[(l % 16) * 4 + l / 16];
  float b = B[(l / 16) * 16 + l % 16];
  f32x4 c = {0, 0, 0, 0};
  f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(4 * (l / 16) + r) * 16 + l % 16] = d[r];
}
int main() {
  float hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; i++) hA[i] = (float)((i * 7) % 13) - 3.f, hB[i] = (float)((i * 5) % 11) - 4.f;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float s = 0; for (int k = 0; k < 4; k++) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dD;
  (void)hipMalloc(&dA, 256), (void)hipMalloc(&dB, 256), (void)hipMalloc(&dD, 1024);
  (void)hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice), (void)hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; i++) if (hD[i] != ref[i]) bad++;
  printf("mfma_f32_16x16x4 layout check: %d mismatches of 256\n", bad);
  return bad != 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// verify operand layout of v_mfma_f32_32x32x8_f16: A[i][k]: lane = i + 32*(k/4), element k%4; B[k][j]: lane = j + 32*(k/4), element k%4
__global__ void k(const float* A, const float* B, float* D) {   // A [32][8], B [8][32], D [32][32]
  const unsigned lane = threadIdx.x;
  half4 a, b;
  for (int e = 0; e < 4; ++e) {
    a[e] = (_Float16)A[(lane & 31) * 8 + 4 * (lane >> 5) + e];
    b[e] = (_Float16)B[(4 * (lane >> 5) + e) * 32 + (lane & 31)];
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}
int main() {
  float hA[256], hB[256], hD[1024], ref[1024];
  for (int i = 0; i < 256; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 8; ++kk) s += hA[i * 8 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 1024; ++i) err += fabs(hD[i] - ref[i]);
  printf("layout check: sum |D - ref| = %g\n", err);
  return 0;
}

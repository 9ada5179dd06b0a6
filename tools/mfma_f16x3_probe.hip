// Hardware probe for the split-operand MLP kernels (csrc/mlp.hip): run on an MI355X, prints PASS/FAIL lines.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16x3_probe.hip -o tools/mfma_f16x3_probe && tools/mfma_f16x3_probe
// 1. operand lane map of v_mfma_f32_32x32x16_f16 (lane (r, h): A[r][8h+j], B[8h+j][r], j = 0..7)
// 2. an accumulator tile as the next product's B operand: element j of lane half h = row 16s + 8(j>>2) + 4h + (j&3)
// 3. f16 subnormal operands (are they honoured or flushed?)
// 4. error of the 3-product hi/lo split (a_h b_h + a_h b_l + a_l b_h, fp32 accumulate) against fp64
// 5. issue rate: 3 x f16 32x32x16 against the f32 32x32x2 form for the same 32x32x16 product
// 6. ds_read_b64_tr_b16 lane map
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned acc_row(unsigned r, unsigned h) { return (r & 3u) + 8u * (r >> 2) + 4u * h; }

// D[32][32] = A[32][16] * B[16][32]
__global__ void k_layout(const float* A, const float* B, float* D) {
  const unsigned lane = threadIdx.x, r = lane & 31u, h = lane >> 5;
  half8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)A[r * 16 + 8 * h + j];
    b[j] = (_Float16)B[(8 * h + j) * 32 + r];
  }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[acc_row(i, h) * 32 + r] = acc[i];
}

// X[32][32] = A1[32][16] * B1[16][32] in the accumulator; Y[32][32] = W[32][32] * X with X taken from the accumulator
__global__ void k_chain(const float* A1, const float* B1, const float* W, float* Y) {
  const unsigned lane = threadIdx.x, r = lane & 31u, h = lane >> 5;
  half8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)A1[r * 16 + 8 * h + j];
    b[j] = (_Float16)B1[(8 * h + j) * 32 + r];
  }
  f32x16 x;
  for (int i = 0; i < 16; ++i) x[i] = 0.f;
  x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, x, 0, 0, 0);
  f32x16 y;
  for (int i = 0; i < 16; ++i) y[i] = 0.f;
  for (int s = 0; s < 2; ++s) {
    half8 bx, aw;
    for (int j = 0; j < 8; ++j) {
      bx[j] = (_Float16)x[8 * s + j];
      const unsigned k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      aw[j] = (_Float16)W[r * 32 + k];
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, bx, y, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) Y[acc_row(i, h) * 32 + r] = y[i];
}

__global__ void k_subnormal(float* out) {
  const unsigned lane = threadIdx.x;
  half8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)0.f;
    b[j] = (_Float16)0.f;
  }
  if ((lane >> 5) == 0) {
    a[0] = (_Float16)9.5367431640625e-07f;   // 2^-20: an f16 subnormal
    b[0] = (_Float16)1024.f;
  }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];             // 2^-10 if honoured, 0 if flushed
}

// hi/lo split product C[M][N] = A[M][K] B[K][N], M = N = 32, K multiple of 16; one wave
__global__ void k_split(const float* A, const float* B, float* C3, float* C1, int K, float sa, float sb) {
  const unsigned lane = threadIdx.x, r = lane & 31u, h = lane >> 5;
  f32x16 c3, c1;
  for (int i = 0; i < 16; ++i) c3[i] = c1[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    half8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      const float av = A[r * K + k0 + 8 * h + j] * sa, bv = B[(k0 + 8 * h + j) * 32 + r] * sb;
      ah[j] = (_Float16)av;
      al[j] = (_Float16)(av - (float)ah[j]);
      bh[j] = (_Float16)bv;
      bl[j] = (_Float16)(bv - (float)bh[j]);
    }
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
  }
  const float inv = 1.0f / (sa * sb);
  for (int i = 0; i < 16; ++i) {
    C3[acc_row(i, h) * 32 + r] = c3[i] * inv;
    C1[acc_row(i, h) * 32 + r] = c1[i] * inv;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
  const unsigned lane = threadIdx.x & 63u;
  f32x16 acc[2];
  for (int t = 0; t < 2; ++t)
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  if (MODE == 0) {          // 3 x f16 32x32x16 per 32x32x16 product
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (_Float16)(0.001f * (lane + j));
      b[j] = (_Float16)(0.002f * (lane + 3 * j));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc[t], 0, 0, 0);
        }
    }
  } else {                  // 8 x f32 32x32x2 per 32x32x16 product
    float a = 0.001f * lane, b = 0.002f * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int s = 0; s < 8; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 2; ++t)
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  if (s == 12345.678f) out[0] = s;
}

// 6. ds_read_b64_tr_b16: LDS holds u16 value = its own index; lane 4q+p of each 16-lane group supplies the address of
// (row q, columns 4p..4p+3) of a [4][16] block of 16-bit elements with a row stride of `ld` elements
__global__ void k_tr(unsigned short* out, int ld) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3u, p = lane & 3u;
  // group g reads the block whose first row is 4g and first column 0
  const unsigned addr = (unsigned)(size_t)lds + 2u * ((4u * g + q) * ld + 4u * p);
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = (unsigned short)(v & 0xffff);
  out[lane * 4 + 1] = (unsigned short)((v >> 16) & 0xffff);
  out[lane * 4 + 2] = (unsigned short)((v >> 32) & 0xffff);
  out[lane * 4 + 3] = (unsigned short)((v >> 48) & 0xffff);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  float *dA, *dB, *dC, *dD;
  CK(hipMalloc(&dA, 1 << 20)); CK(hipMalloc(&dB, 1 << 20)); CK(hipMalloc(&dC, 1 << 20)); CK(hipMalloc(&dD, 1 << 20));
  // 1. layout (integer data, asymmetric)
  {
    float hA[512], hB[512], hD[1024], ref[1024];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
    CK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 1024; ++i) err += fabs(hD[i] - ref[i]);
    printf("1 layout 32x32x16_f16: sum|D-ref| = %g  %s\n", err, err == 0 ? "PASS" : "FAIL");
  }
  // 2. chain
  {
    float hA[512], hB[512], hW[1024], hY[1024], X[1024], ref[1024];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 5 - 2); hB[i] = (float)((i * 3) % 7 - 3); }
    for (int i = 0; i < 1024; ++i) hW[i] = (float)((i * 11) % 9 - 4);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; X[i * 32 + j] = s; }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += hW[i * 32 + k] * X[k * 32 + j]; ref[i * 32 + j] = s; }
    CK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, hW, sizeof hW, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    CK(hipMemcpy(hY, dD, sizeof hY, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 1024; ++i) err += fabs(hY[i] - ref[i]);
    printf("2 accumulator as B operand: sum|Y-ref| = %g  %s\n", err, err == 0 ? "PASS" : "FAIL");
  }
  // 3. subnormals
  {
    float h = -1.f;
    hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, dD);
    CK(hipMemcpy(&h, dD, 4, hipMemcpyDeviceToHost));
    printf("3 f16 subnormal operand 2^-20 * 2^10 = %g (%s)\n", h, h == 9.765625e-4f ? "honoured" : "FLUSHED");
  }
  // 4. split accuracy
  for (int K : {80, 112, 64, 1024}) {
    std::vector<float> hA(32 * K), hB(K * 32), h3(1024), h1(1024);
    srand(K);
    float ma = 0, mb = 0;
    for (auto& v : hA) { v = 0.2f * ((rand() / (float)RAND_MAX) * 2 - 1); ma = fmaxf(ma, fabsf(v)); }
    for (int i = 0; i < K * 32; ++i) { hB[i] = ((rand() / (float)RAND_MAX) * 2 - 1) * ((i % 32) < 8 ? 1e-4f : 1.f); mb = fmaxf(mb, fabsf(hB[i])); }
    const float sa = exp2f(floorf(14 - log2f(ma))), sb = exp2f(floorf(14 - log2f(mb)));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, K, sa, sb);
    CK(hipMemcpy(h3.data(), dC, 4096, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), dD, 4096, hipMemcpyDeviceToHost));
    double e3 = 0, e1 = 0, e32 = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0, den = 0; float c = 0;
      for (int k = 0; k < K; ++k) { s += (double)hA[i * K + k] * hB[k * 32 + j]; den += fabs((double)hA[i * K + k] * hB[k * 32 + j]); c = fmaf(hA[i * K + k], hB[k * 32 + j], c); }
      e3 = fmax(e3, fabs(h3[i * 32 + j] - s) / den); e1 = fmax(e1, fabs(h1[i * 32 + j] - s) / den); e32 = fmax(e32, fabs(c - s) / den);
    }
    printf("4 split product K=%d: max err / sum|ab|  f16x3 %.3e   f16x1 %.3e   fp32 fma chain %.3e\n", K, e3, e1, e32);
  }
  // 5. rate
  {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, blocks = 1024;
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, dD, iters);
        else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, dD, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double prods = (double)blocks * 4 * iters * 8;    // 32x32x16 products per wave: 4 u x 2 t
      printf("5 rate %s: %.3f ms, %.2f T 32x32x16-products/s (= %.1f TFLOP/s of fp32-equivalent math)\n", mode == 0 ? "3 x f16 32x32x16" : "8 x f32 32x32x2 ",
             ms, prods / (ms * 1e-3) / 1e12, prods * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
    }
  }
  // 6. transposed read
  {
    unsigned short *dO, hO[256];
    CK(hipMalloc(&dO, 512));
    for (int ld : {16, 72}) {
      hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, dO, ld);
      CK(hipMemcpy(hO, dO, 512, hipMemcpyDeviceToHost));
      // expectation (guide T10): lane i of group g receives column i of rows 4g..4g+3: element q = lds[(4g+q)*ld + i]
      int bad = 0;
      for (int lane = 0; lane < 64; ++lane) for (int q = 0; q < 4; ++q) bad += hO[lane * 4 + q] != (unsigned short)((4 * (lane >> 4) + q) * ld + (lane & 15));
      printf("6 ds_read_b64_tr_b16 ld=%d: %s; lane0 = %u %u %u %u, lane1 = %u %u %u %u, lane17 = %u %u %u %u\n", ld, bad ? "MISMATCH" : "PASS",
             hO[0], hO[1], hO[2], hO[3], hO[4], hO[5], hO[6], hO[7], hO[68], hO[69], hO[70], hO[71]);
    }
  }
  return 0;
}

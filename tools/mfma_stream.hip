// Ceiling of the weight-gradient loop shape: per step a wave reads 2 x 1 KB (float4 per lane) and issues NM
// v_mfma_f32_16x16x4_f32 on them, D steps of loads in flight.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int D, int NM>
__global__ __launch_bounds__(256, 2) void k(const float4* __restrict__ A, const float4* __restrict__ B, float* __restrict__ out,
                                            size_t steps_per_wave) {
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63u;
  const float4* pa = A + wave * steps_per_wave * 64 + lane;
  const float4* pb = B + wave * steps_per_wave * 64 + lane;
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra[D], rb[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { ra[d] = pa[d * 64]; rb[d] = pb[d * 64]; asm volatile("" ::: "memory"); }
  for (size_t q0 = 0; q0 < steps_per_wave; q0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float av[4] = {ra[d].x, ra[d].y, ra[d].z, ra[d].w};
      const float bv[4] = {rb[d].x, rb[d].y, rb[d].z, rb[d].w};
      const size_t qn = q0 + d + D < steps_per_wave ? q0 + d + D : 0;
      ra[d] = pa[qn * 64];
      rb[d] = pb[qn * 64];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int m = 0; m < NM; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m >> 2], bv[m & 3], acc[m], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}
int main() {
  const size_t P = 262144;                  // points; a step = 4 points x 64 channels x 4 B = 1 KB per operand
  const size_t steps = P / 4, bytes = steps * 1024;
  float4 *A, *B; float* out;
  hipMalloc(&A, bytes * 4); hipMalloc(&B, bytes * 4); hipMalloc(&out, 4);
  hipMemset(A, 0, bytes * 4); hipMemset(B, 0, bytes * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, int blocks, auto kern, int units) {
    // `units` independent (A,B) streams of P points each are swept by the grid (the roles kernel has 4: dW_in lo/hi, dW_h, dW_out)
    const size_t waves = (size_t)blocks * 4, spw = steps * units / waves;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, out, spw);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, B, out, spw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gb = 2.0 * bytes * units / 1e9;
    printf("%-28s blocks=%4d units=%d: %7.1f us  %5.2f TB/s\n", name, blocks, units, ms * 100.0, gb * 10 / ms);
  };
  for (int blocks : {512, 768, 1024}) {
    timeit("D=2 16 mfma/step", blocks, k<2, 16>, 4);
    timeit("D=3 16 mfma/step", blocks, k<3, 16>, 4);
    timeit("D=4 16 mfma/step", blocks, k<4, 16>, 4);
    timeit("D=6 16 mfma/step", blocks, k<6, 16>, 4);
  }
  return 0;
}

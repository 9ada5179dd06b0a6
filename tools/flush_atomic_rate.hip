// Micro-benchmark: what the end-of-run weight-gradient flush of the MLP backward kernels costs.  G workgroups of 256 threads each
// add T tiles of 32 x 32 floats into a [T x 1024] gradient: every workgroup hits the SAME T x 32 lines (256 adds per address at
// G = 256).  R replicas (workgroup b adds into replica b % R): the same number of atomics, 1 / R of the adds per address.
//   hipcc --offload-arch=gfx950 -O3 tools/flush_atomic_rate.hip -o tools/flush_atomic_rate && tools/flush_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ __launch_bounds__(256) void flush(float* g, uint32_t tiles, uint32_t replicas, uint32_t stride, int mode) {
  float* dst = g + (size_t)(blockIdx.x % replicas) * stride;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, j = lane & 31u, h = lane >> 5;
  for (uint32_t t = 0; t < tiles; ++t) {
    // mode 1: stagger the tile order by workgroup (the same addresses, different times)
    const uint32_t tt = mode == 1 ? (t + blockIdx.x) % tiles : t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t row = wave * 8u + 2u * i + h;
      atomicAdd(dst + (size_t)tt * 1024u + row * 32u + j, 1.0f + (float)t);
    }
  }
}
int main() {
  const uint32_t G = 256, stride = 64 * 1024;
  float* g;
  (void)hipMalloc(&g, (size_t)256 * stride * 4);
  (void)hipMemset(g, 0, (size_t)256 * stride * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (uint32_t groups : {256u, 512u})
    for (uint32_t tiles : {6u, 8u, 14u})
      for (int mode : {0, 1})
        for (uint32_t R : {1u, 2u, 8u, 32u, 256u}) {
          for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(flush, dim3(groups), dim3(256), 0, 0, g, tiles, R, stride, mode);
          (void)hipEventRecord(e0);
          for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(flush, dim3(groups), dim3(256), 0, 0, g, tiles, R, stride, mode);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1);
          printf("workgroups %3u tiles %2u %s replicas %3u: %6.1f us per launch (%.2f M atomics)\n", groups, tiles, mode ? "staggered" : "in order ", R,
                 ms / 20 * 1e3, groups * tiles * 1024.0 / 1e6);
        }
  (void)G;
  return 0;
}

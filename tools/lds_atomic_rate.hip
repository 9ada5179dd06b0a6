// Micro-benchmark: LDS atomic throughput on gfx950 (random addresses, 1024-thread workgroups, one per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t iters, uint32_t mask_lanes, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  float* bf = (float*)raw; uint32_t* bu = (uint32_t*)raw; unsigned long long* bl = (unsigned long long*)raw;
  for (uint32_t i = threadIdx.x; i < 32768; i += 1024) bu[i] = 0;
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const bool active = (threadIdx.x & 63u) < mask_lanes;
  for (uint32_t it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const uint32_t a = (s >> 8);
    if (active) {
      if (MODE == 0) atomicAdd(bf + (a & 32767u), 1.0f);
      if (MODE == 1) atomicAdd(bu + (a & 32767u), 1u);
      if (MODE == 2) atomicAdd(bl + (a & 16383u), 1ull);
      if (MODE == 3) { atomicAdd(bf + (a & 32766u), 1.0f); atomicAdd(bf + (a & 32766u) + 1, 1.0f); }
      if (MODE == 4) bf[a & 32767u] += 1.0f;     // plain RMW (racy) as a reference for the LDS pipe rate
      if (MODE == 5) atomicAdd((double*)raw + (a & 16383u), 1.0);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = bf[5] + (float)bu[7];
}
template <int MODE> void run(const char* name, uint32_t lanes) {
  float* out; hipMalloc(&out, 4096);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 131072, 0, 100, lanes, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 131072, 0, iters, lanes, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double lane_ops = 256.0 * 1024 * (lanes / 64.0) * iters * (MODE == 3 ? 2 : 1);
  printf("%-28s lanes/wave=%2u : %8.3f ms  %7.2f G lane-atomics/s chip  (%.2f cycles per wave-instr per CU @2.4GHz)\n", name, lanes, ms,
         lane_ops / ms / 1e6, ms * 1e-3 * 2.4e9 / (16.0 * iters * (MODE == 3 ? 2 : 1)));
  hipFree(out);
}
int main() {
  for (uint32_t lanes : {64u, 16u}) {
    run<0>("ds_add_f32 random", lanes);
    run<1>("ds_add_u32 random", lanes);
    run<2>("ds_add_u64 random", lanes);
    run<3>("ds_add_f32 pair (x, x+1)", lanes);
    run<4>("plain ds rmw", lanes);
    run<5>("ds_add_f64 random", lanes);
  }
  return 0;
}

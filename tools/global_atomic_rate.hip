// Micro-benchmark: global atomic throughput on gfx950 for random 8-byte rows of a table (hash-grid scatter shape).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(void* buf, uint32_t rows_mask, uint32_t iters) {
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  float* bf = (float*)buf; unsigned long long* bl = (unsigned long long*)buf; uint32_t* bu = (uint32_t*)buf;
  for (uint32_t it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const uint32_t r = (s >> 8) & rows_mask;
    if (MODE == 0) { atomicAdd(bf + 2 * r, 1.0f); atomicAdd(bf + 2 * r + 1, 1.0f); }      // float pair (8-byte row)
    if (MODE == 1) { atomicAdd(bl + 2 * r, 1ull); atomicAdd(bl + 2 * r + 1, 1ull); }      // u64 pair (16-byte row)
    if (MODE == 2) { atomicAdd(bu + 2 * r, 1u); atomicAdd(bu + 2 * r + 1, 1u); }          // u32 pair
    if (MODE == 3) { atomicAdd(bl + r, 1ull); }                                           // one u64 per row
  }
}
template <int MODE> void run(const char* name, uint32_t log2_rows) {
  void* buf; size_t bytes = ((size_t)1 << log2_rows) * 16; (void)hipMalloc(&buf, bytes); (void)hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const uint32_t iters = 64, blocks = 4096;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, (1u << log2_rows) - 1, 4);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, (1u << log2_rows) - 1, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double n = (double)blocks * 256 * iters * (MODE == 3 ? 1 : 2);
  printf("%-26s rows=2^%u : %8.3f ms  %8.2f G atomics/s\n", name, log2_rows, ms, n / ms / 1e6);
  (void)hipFree(buf);
}
int main() {
  for (uint32_t lr : {16u, 20u, 23u}) {
    run<0>("global f32 pair", lr); run<1>("global u64 pair", lr); run<2>("global u32 pair", lr); run<3>("global u64 single", lr);
  }
  return 0;
}

// HBM streaming ceilings on the box: float4 copy, float4 read-only sum, float4 fill.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read4(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}
// per-wave contiguous chunks with U loads in flight (the weight-gradient kernel's pattern)
template <int U>
__global__ void read4_chunked(const float4* __restrict__ a, float* __restrict__ out, size_t n, size_t per_wave) {
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63u;
  const float4* p = a + wave * per_wave + lane;
  float s = 0.f;
  for (size_t i = 0; i + U * 64 <= per_wave; i += U * 64) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * 64];
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (s == 123.456f) out[0] = s;
}
__global__ void fill4(float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 16;
  float4 *a, *b; float* out;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, double gb, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %7.1f us  %6.2f TB/s\n", name, ms * 100.0, gb * 10 / ms);
  };
  for (int blocks : {1024, 2048, 8192}) {
    printf("blocks=%d x 256\n", blocks);
    timeit("copy4 (1 GB rd + 1 GB wr)", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(copy4, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    timeit("read4 (1 GB)", bytes / 1e9, [&] { hipLaunchKernelGGL(read4, dim3(blocks), dim3(256), 0, 0, a, out, n); });
    timeit("fill4 (1 GB)", bytes / 1e9, [&] { hipLaunchKernelGGL(fill4, dim3(blocks), dim3(256), 0, 0, b, n); });
  }
  for (int blocks : {256, 512, 1024}) {
    const size_t waves = (size_t)blocks * 4, per_wave = n / waves;
    printf("chunked read, %d blocks x 256 (per-wave contiguous %zu KB)\n", blocks, per_wave * 16 / 1024);
    timeit("  8 loads in flight", bytes / 1e9, [&] { hipLaunchKernelGGL(read4_chunked<8>, dim3(blocks), dim3(256), 0, 0, a, out, n, per_wave); });
    timeit("  16 loads in flight", bytes / 1e9, [&] { hipLaunchKernelGGL(read4_chunked<16>, dim3(blocks), dim3(256), 0, 0, a, out, n, per_wave); });
  }
  return 0;
}

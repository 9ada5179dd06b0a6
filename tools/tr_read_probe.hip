// Probe: semantics of gfx950 ds_read_b64_tr_b16 (cdna_hip_programming.md T10) against the host model in
// tools/mfma_lds_model.py.  LDS holds element e at 16-bit slot e (value = e); every lane supplies a byte address from a table
// given by the host; the kernel returns the 4 x 16-bit elements each lane received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

__global__ void probe(const uint32_t* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t a = addr[threadIdx.x] + (uint32_t)(uintptr_t)lds;     // LDS byte address
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = (uint16_t)(v.x & 0xffffu);
  out[threadIdx.x * 4 + 1] = (uint16_t)(v.x >> 16);
  out[threadIdx.x * 4 + 2] = (uint16_t)(v.y & 0xffffu);
  out[threadIdx.x * 4 + 3] = (uint16_t)(v.y >> 16);
}

int main() {
  // model: per 16-lane group, lane 4q+p supplies row q, columns 4p..4p+3; lane i receives column i of rows 0..3
  int bad = 0;
  for (int trial = 0; trial < 8; ++trial) {
    std::vector<uint32_t> addr(64);
    srand(trial + 1);
    for (int l = 0; l < 64; ++l) addr[l] = 8u * (uint32_t)(rand() % 4000);
    uint32_t* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
    hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    std::vector<uint16_t> out(256);
    hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
    for (int g = 0; g < 4; ++g) {
      uint16_t block[4][16];
      for (int q = 0; q < 4; ++q) for (int p = 0; p < 4; ++p) for (int e = 0; e < 4; ++e)
        block[q][4 * p + e] = (uint16_t)(addr[16 * g + 4 * q + p] / 2 + e);
      for (int i = 0; i < 16; ++i) for (int q = 0; q < 4; ++q)
        if (out[(16 * g + i) * 4 + q] != block[q][i]) {
          if (bad < 10) printf("trial %d lane %d elem %d: got %u want %u\n", trial, 16 * g + i, q, out[(16 * g + i) * 4 + q], block[q][i]);
          ++bad;
        }
    }
    hipFree(d_addr); hipFree(d_out);
  }
  printf("tr_read_probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "model confirmed", bad);
  return bad ? 1 : 0;
}

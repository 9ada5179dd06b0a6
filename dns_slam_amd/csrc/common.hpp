// Shared host/device helpers for libdns_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dns_hip.h"

namespace dns {

void set_error(const char* fmt, ...);

// Sticky per-device error word in pinned host memory (host.cpp): kernels that consume device counters set a bit instead of
// storing out of bounds; poll_device_error() is a plain host read, no synchronisation.
constexpr uint32_t DNS_DEVERR_GROUP_CURSOR = 1u;
constexpr uint32_t DNS_DEVERR_MLP_RANGE = 2u;   // a row of an MLP backward operand lies beyond the 2 GiB a buffer descriptor addresses
uint32_t* device_error_word();               // device-visible pointer for kernel arguments (NULL before dns_init)
int poll_device_error(const char* what);     // DNS_OK, or DNS_E_LAUNCH + message while the word is non-zero

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DNS_E_LAUNCH;
  }
  return poll_device_error(what);
}

#define DNS_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      ::dns::set_error(__VA_ARGS__);           \
      return DNS_E_ARG;                        \
    }                                          \
  } while (0)

constexpr int WAVE = 64;

// One-time per-device kernel attributes (hipFuncAttributeMaxDynamicSharedMemorySize for every kernel that takes more than
// 64 KB of dynamic LDS).  hipFuncSetAttribute is a context-level call: it must not run while a stream is being captured
// into a hipGraph and it is not something to repeat on every launch, so each source file registers ONE function that
// sets the attributes of its kernels, dns_init() (host.cpp) runs them once per device, and the launch wrappers call
// ensure_ready(): a no-op after dns_init(), an implicit dns_init() on first use outside capture, DNS_E_STATE under capture.
typedef int (*AttrInitFn)();
struct AttrRegistrar {
  explicit AttrRegistrar(AttrInitFn fn);
};
int ensure_ready(hipStream_t st, const char* who);

// Per-kernel durations for bench.py's roofline (dns_kernel_timing, host.cpp): while timing is on, every launch made through
// DNS_LAUNCH is bracketed by an event pair on ITS stream; off (the default) the scope costs one relaxed atomic load.
struct KernelSpan {
  KernelSpan(const char* name, hipStream_t st);
  ~KernelSpan();
  const char* name;
  hipStream_t st;
  hipEvent_t e0;
  bool on;
};
#define DNS_LAUNCH(kern, grid, block, lds, st, ...)                    \
  do {                                                                  \
    dns::KernelSpan _span(#kern, st);                                   \
    hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);        \
  } while (0)
// Fill n_words 32-bit words with `value` by a KERNEL.  The library issues no hipMemsetAsync: captured into a hipGraph, memset
// nodes misbehave on ROCm 7.0.51831 -- after a host synchronisation that follows a replay, later replays no longer clear
// their destination (found in round 2 with tools/graph_sync_min.py: the sampling step is stable with the per-frame depth
// maximum supplied, i.e. without its memset, and goes wrong with it).
int fill_words(void* dst, uint32_t value, size_t n_words, hipStream_t st, const char* who);
int fill_words2(void* a, uint32_t va, size_t na, void* b, uint32_t vb, size_t nb, hipStream_t st, const char* who);
constexpr int MAX_DYN_LDS = 160 * 1024;      // gfx950: 160 KiB of LDS per CU, all of it available to one workgroup

// Device-side copy of the level table, passed by value as a kernel argument.
struct GridLevels {
  uint32_t n_levels;
  float scale[DNS_MAX_LEVELS];
  uint32_t resolution[DNS_MAX_LEVELS];
  uint32_t size[DNS_MAX_LEVELS];
  uint32_t offset[DNS_MAX_LEVELS];
  uint32_t hashed[DNS_MAX_LEVELS];
};

inline GridLevels to_levels(const DnsGridMeta* m) {
  GridLevels g;
  g.n_levels = m->n_levels;
  for (uint32_t l = 0; l < DNS_MAX_LEVELS; ++l) {
    g.scale[l] = m->scale[l];
    g.resolution[l] = m->resolution[l];
    g.size[l] = m->size[l];
    g.offset[l] = m->offset[l];
    g.hashed[l] = m->hashed[l];
  }
  return g;
}

namespace sp { struct XsIn; }                // split-row input of the MLP kernels (mlp_split.hpp); NULL = fp32 rows
// split-operand MLP kernels (mlp_split.hip); arguments validated by the C-ABI wrappers in mlp.hip
int launch_mlp_fwd_split(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, const float* params,
                         uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, float* y, uint32_t ldy,
                         uint32_t n_slots, const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride,
                         float* h_save, bool fp16_single, const unsigned char* prep, uint32_t prep_stride, const sp::XsIn* xs,
                         hipStream_t st, uint32_t n_in_w = 0);
int launch_mlp_bwd_split(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, const float* dy,
                         uint32_t lddy, const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                         uint32_t n_hidden_layers, float* d_x, uint32_t lddx, float* d_x2, uint32_t lddx2, float* d_params,
                         float* ws, uint32_t n_slots, const int32_t* row_index, const int32_t* tile_group,
                         uint32_t param_stride, int acc1, int acc2, bool fp16_single, const unsigned char* prep, uint32_t prep_stride,
                         bool with_dwin, const float* h_saved, const sp::XsIn* xs, hipStream_t st, uint32_t n_in_w = 0);
int launch_mlp_dwin(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, uint32_t n_in, uint32_t n_neurons,
                    uint32_t n_hidden_layers, float* d_params, const float* ws, uint32_t n_slots, const int32_t* row_index,
                    const int32_t* tile_group, uint32_t param_stride, bool fp16_single, hipStream_t st, uint32_t n_in_w = 0);
// prepared weight images (mlp_split.hip): bytes per weight set (forward part / both parts), and the launch of the two builders
uint32_t mlp_prepared_fwd_bytes(uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl);
uint32_t mlp_prepared_bytes(uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl);
int launch_mlp_prepare(const float* params, uint32_t param_stride, uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl,
                       uint32_t n_sets, unsigned char* blob, hipStream_t st, uint32_t n_in_w = 0);

}  // namespace dns

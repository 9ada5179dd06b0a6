// Shared host/device helpers for libdns_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dns_hip.h"

namespace dns {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DNS_E_LAUNCH;
  }
  return DNS_OK;
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

}  // namespace dns

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

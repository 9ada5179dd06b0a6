// Device helpers of the along-ray compositing kernels (composite.hip): wave scans / reductions and the forward state of one
// ray -- shared with the fused tracker iteration (track_fused.inc).
#pragma once
#include "common.hpp"

namespace dns {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// exclusive multiplicative scan across the 64 lanes
__device__ __forceinline__ float wave_excl_prod(float v, uint32_t lane) {
  float inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(inc, o);
    if (lane >= (uint32_t)o) inc *= t;
  }
  const float ex = __shfl_up(inc, 1);
  return lane == 0 ? 1.0f : ex;
}

// exclusive additive suffix scan: sum over lanes > lane
__device__ __forceinline__ float wave_suffix_excl_sum(float v, uint32_t lane) {
  float inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(inc, o);
    if (lane + o < 64u) inc += t;
  }
  const float ex = __shfl_down(inc, 1);
  return lane == 63 ? 0.0f : ex;
}

__device__ __forceinline__ float sigmoid10(float occ) { return 1.0f / (1.0f + expf(-10.0f * occ)); }
// the colour network's output activation (models/decoder.py:124) when the caller hands raw rgb LOGITS (DNS_COMPOSITE_RGB_LOGITS)
__device__ __forceinline__ float sigmoid1c(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int E>
struct RayState {
  float alpha[E], om[E], T[E], u[E], w[E], z[E];
  float sumu, depth;
};

template <int E>
__device__ __forceinline__ void ray_forward(const float* __restrict__ raw, const float* __restrict__ zv, uint32_t n,
                                            uint32_t S, uint32_t lane, RayState<E>& st) {
  float lp = 1.0f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t s = lane * E + e;
    const bool ok = s < S;
    const float occ = ok ? raw[((size_t)n * S + s) * 4 + 3] : 0.f;
    st.z[e] = ok ? zv[(size_t)n * S + s] : 0.f;
    st.alpha[e] = ok ? sigmoid10(occ) : 0.f;
    st.om[e] = ok ? (1.0f - st.alpha[e]) + 1e-10f : 1.0f;
    st.T[e] = lp;  // local exclusive product
    lp *= st.om[e];
  }
  const float pre = wave_excl_prod(lp, lane);
  float su = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    st.T[e] *= pre;
    st.u[e] = st.alpha[e] * st.T[e];
    su += st.u[e];
  }
  st.sumu = wave_sum(su);
  float sd = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    st.w[e] = st.u[e] / st.sumu;
    sd += st.w[e] * st.z[e];
  }
  st.depth = wave_sum(sd);
}

}  // namespace dns

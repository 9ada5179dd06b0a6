// Along-ray occupancy compositing: one wave64 per ray, wavefront shuffles for the scan and the reductions.
//
// Replaces raw2nerf_color (reference utils/common.py:506-537, occupancy mode) and the logit composite
// sum_s w_s * logits_s (slams/mapping.py:633, slams/tracking.py:212):
//   alpha = sigmoid(10 occ);  T_i = prod_{j<i}(1 - alpha_j + 1e-10);  u = alpha T;  w = u / sum(u)  (no eps, D9)
//   depth = sum w z;  var = sum w (z - depth)^2;  rgb = sum w c;  sem = sum w logits.
// A lane owns E = ceil(S/64) consecutive samples; the exclusive product is a local product followed by a
// 6-step wave scan (shuffle-up), reductions are xor-butterflies.  Memory-bound: (4+1+C)*4 B per sample.
#include "common.hpp"
#include "dev_composite.hpp"

namespace dns {

template <int E>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ zv,
                                                            const float* __restrict__ logits, uint32_t N, uint32_t S,
                                                            uint32_t C, float* __restrict__ depth, float* __restrict__ var,
                                                            float* __restrict__ rgb, float* __restrict__ weights,
                                                            float* __restrict__ sem, uint32_t rgbl) {
  extern __shared__ float lds[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  float* wl = lds + wave * (64 * E);
  RayState<E> st;
  ray_forward<E>(raw, zv, n, S, lane, st);
  float sv = 0.f, sc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t s = lane * E + e;
    const float t = st.z[e] - st.depth;
    sv += st.w[e] * t * t;
    if (s < S) {
      const float* r = raw + ((size_t)n * S + s) * 4;
      const float c0 = rgbl ? sigmoid1c(r[0]) : r[0], c1 = rgbl ? sigmoid1c(r[1]) : r[1], c2 = rgbl ? sigmoid1c(r[2]) : r[2];
      sc[0] += st.w[e] * c0;
      sc[1] += st.w[e] * c1;
      sc[2] += st.w[e] * c2;
      if (weights) weights[(size_t)n * S + s] = st.w[e];
    }
    wl[s] = st.w[e];
  }
  sv = wave_sum(sv);
  sc[0] = wave_sum(sc[0]);
  sc[1] = wave_sum(sc[1]);
  sc[2] = wave_sum(sc[2]);
  if (lane == 0) {
    depth[n] = st.depth;
    var[n] = sv;
    rgb[(size_t)n * 3 + 0] = sc[0];
    rgb[(size_t)n * 3 + 1] = sc[1];
    rgb[(size_t)n * 3 + 2] = sc[2];
  }
  if (C) {
    // lanes over classes: coalesced reads of logits[n, s, :], w_s broadcast from LDS (wave-private, no barrier)
    for (uint32_t c = lane; c < C; c += 64) {
      float acc = 0.f;
      const float* lp = logits + (size_t)n * S * C + c;
      for (uint32_t s = 0; s < S; ++s) acc += wl[s] * lp[(size_t)s * C];
      sem[(size_t)n * C + c] = acc;
    }
  }
}

template <int E>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ zv,
                                                            const float* __restrict__ logits, uint32_t N, uint32_t S,
                                                            uint32_t C, const float* __restrict__ d_depth,
                                                            const float* __restrict__ d_var, const float* __restrict__ d_rgb,
                                                            const float* __restrict__ d_weights,
                                                            const float* __restrict__ d_sem, float* __restrict__ d_raw,
                                                            float* __restrict__ d_logits, uint32_t rgbl) {
  extern __shared__ float lds[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  float* wl = lds + wave * (64 * E);
  RayState<E> st;
  ray_forward<E>(raw, zv, n, S, lane, st);
  const float gD = d_depth ? d_depth[n] : 0.f;
  const float gV = d_var ? d_var[n] : 0.f;
  float gc[3] = {0.f, 0.f, 0.f};
  if (d_rgb) {
    gc[0] = d_rgb[(size_t)n * 3];
    gc[1] = d_rgb[(size_t)n * 3 + 1];
    gc[2] = d_rgb[(size_t)n * 3 + 2];
  }
  // g_i = dL/dw_i (w treated as free), then through the normalisation and the transmittance product
  float g[E];
  float col[E][3];                               // the sample's colour (after the sigmoid when raw holds logits)
  float gbar = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t s = lane * E + e;
    g[e] = 0.f;
    col[e][0] = col[e][1] = col[e][2] = 0.f;
    if (s < S) {
      const float* r = raw + ((size_t)n * S + s) * 4;
      const float t = st.z[e] - st.depth;
      col[e][0] = rgbl ? sigmoid1c(r[0]) : r[0];
      col[e][1] = rgbl ? sigmoid1c(r[1]) : r[1];
      col[e][2] = rgbl ? sigmoid1c(r[2]) : r[2];
      float v = gc[0] * col[e][0] + gc[1] * col[e][1] + gc[2] * col[e][2] + gD * st.z[e] + gV * t * t;
      if (d_weights) v += d_weights[(size_t)n * S + s];
      if (C && d_sem) {
        const float* lp = logits + ((size_t)n * S + s) * C;
        const float* gs = d_sem + (size_t)n * C;
        float dot = 0.f;
        for (uint32_t c = 0; c < C; ++c) dot += gs[c] * lp[c];
        v += dot;
      }
      g[e] = v;
    }
    gbar += st.w[e] * g[e];
    wl[s] = st.w[e];
  }
  gbar = wave_sum(gbar);
  float du[E], lsum = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    du[e] = (g[e] - gbar) / st.sumu;
    lsum += du[e] * st.u[e];
  }
  const float after = wave_suffix_excl_sum(lsum, lane);  // sum over later lanes of du*u
  float tail = after;
#pragma unroll
  for (int e = E - 1; e >= 0; --e) {
    const uint32_t s = lane * E + e;
    const float dalpha = du[e] * st.T[e] - tail / st.om[e];
    tail += du[e] * st.u[e];
    if (s < S) {
      float* o = d_raw + ((size_t)n * S + s) * 4;
      float o0 = st.w[e] * gc[0], o1 = st.w[e] * gc[1], o2 = st.w[e] * gc[2];
      if (rgbl) {                                // through the sigmoid: d logit = d colour * (1 - s) * s
        o0 = o0 * (1.f - col[e][0]) * col[e][0];
        o1 = o1 * (1.f - col[e][1]) * col[e][1];
        o2 = o2 * (1.f - col[e][2]) * col[e][2];
      }
      o[0] = o0;
      o[1] = o1;
      o[2] = o2;
      o[3] = 10.0f * st.alpha[e] * (1.0f - st.alpha[e]) * dalpha;
    }
  }
  if (C && d_logits) {
    for (uint32_t c = lane; c < C; c += 64) {
      const float gs = d_sem ? d_sem[(size_t)n * C + c] : 0.f;
      float* lp = d_logits + (size_t)n * S * C + c;
      for (uint32_t s = 0; s < S; ++s) lp[(size_t)s * C] = wl[s] * gs;
    }
  }
}

}  // namespace dns

using namespace dns;

static int composite_fwd_impl(const char* who, const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S,
                              uint32_t C, float* depth, float* var, float* rgb, float* weights, float* sem, uint32_t rgbl,
                              void* stream) {
  if (N == 0) return DNS_OK;
  DNS_REQUIRE(raw && z && depth && var && rgb, "%s: NULL argument", who);
  DNS_REQUIRE(S >= 1 && S <= 256, "%s: S=%u out of range [1,256]", who, S);
  DNS_REQUIRE(C == 0 || (logits && sem), "%s: C>0 needs logits and sem", who);
  const uint32_t blocks = (N + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t E = S <= 64 ? 1 : (S <= 128 ? 2 : 4);
  const size_t ldsb = 4 * 64 * E * sizeof(float);
  if (E == 1) DNS_LAUNCH(composite_fwd_kernel<1>, dim3(blocks), dim3(256), ldsb, st, raw, z, logits, N, S, C, depth, var, rgb, weights, sem, rgbl);
  else if (E == 2) DNS_LAUNCH(composite_fwd_kernel<2>, dim3(blocks), dim3(256), ldsb, st, raw, z, logits, N, S, C, depth, var, rgb, weights, sem, rgbl);
  else DNS_LAUNCH(composite_fwd_kernel<4>, dim3(blocks), dim3(256), ldsb, st, raw, z, logits, N, S, C, depth, var, rgb, weights, sem, rgbl);
  return check_launch(who);
}

static int composite_bwd_impl(const char* who, const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S,
                              uint32_t C, const float* d_depth, const float* d_var, const float* d_rgb, const float* d_weights,
                              const float* d_sem, float* d_raw, float* d_logits, uint32_t rgbl, void* stream) {
  if (N == 0) return DNS_OK;
  DNS_REQUIRE(raw && z && d_raw, "%s: NULL argument", who);
  DNS_REQUIRE(S >= 1 && S <= 256, "%s: S=%u out of range [1,256]", who, S);
  DNS_REQUIRE(C == 0 || logits, "%s: C>0 needs logits", who);
  const uint32_t blocks = (N + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t E = S <= 64 ? 1 : (S <= 128 ? 2 : 4);
  const size_t ldsb = 4 * 64 * E * sizeof(float);
  if (E == 1) DNS_LAUNCH(composite_bwd_kernel<1>, dim3(blocks), dim3(256), ldsb, st, raw, z, logits, N, S, C, d_depth, d_var, d_rgb, d_weights, d_sem, d_raw, d_logits, rgbl);
  else if (E == 2) DNS_LAUNCH(composite_bwd_kernel<2>, dim3(blocks), dim3(256), ldsb, st, raw, z, logits, N, S, C, d_depth, d_var, d_rgb, d_weights, d_sem, d_raw, d_logits, rgbl);
  else DNS_LAUNCH(composite_bwd_kernel<4>, dim3(blocks), dim3(256), ldsb, st, raw, z, logits, N, S, C, d_depth, d_var, d_rgb, d_weights, d_sem, d_raw, d_logits, rgbl);
  return check_launch(who);
}

extern "C" int dns_composite_fwd(const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S,
                                 uint32_t C, float* depth, float* var, float* rgb, float* weights, float* sem,
                                 void* stream) {
  return composite_fwd_impl("dns_composite_fwd", raw, z, logits, N, S, C, depth, var, rgb, weights, sem, 0u, stream);
}

extern "C" int dns_composite_bwd(const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S,
                                 uint32_t C, const float* d_depth, const float* d_var, const float* d_rgb,
                                 const float* d_weights, const float* d_sem, float* d_raw, float* d_logits,
                                 void* stream) {
  return composite_bwd_impl("dns_composite_bwd", raw, z, logits, N, S, C, d_depth, d_var, d_rgb, d_weights, d_sem, d_raw, d_logits, 0u,
                            stream);
}

extern "C" int dns_composite_fwd_ex(const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S, uint32_t C,
                                    float* depth, float* var, float* rgb, float* weights, float* sem, uint32_t flags, void* stream) {
  DNS_REQUIRE((flags & ~DNS_COMPOSITE_RGB_LOGITS) == 0, "dns_composite_fwd_ex: unknown flags 0x%x", flags);
  return composite_fwd_impl("dns_composite_fwd_ex", raw, z, logits, N, S, C, depth, var, rgb, weights, sem, flags & DNS_COMPOSITE_RGB_LOGITS,
                            stream);
}

extern "C" int dns_composite_bwd_ex(const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S, uint32_t C,
                                    const float* d_depth, const float* d_var, const float* d_rgb, const float* d_weights,
                                    const float* d_sem, float* d_raw, float* d_logits, uint32_t flags, void* stream) {
  DNS_REQUIRE((flags & ~DNS_COMPOSITE_RGB_LOGITS) == 0, "dns_composite_bwd_ex: unknown flags 0x%x", flags);
  return composite_bwd_impl("dns_composite_bwd_ex", raw, z, logits, N, S, C, d_depth, d_var, d_rgb, d_weights, d_sem, d_raw, d_logits,
                            flags & DNS_COMPOSITE_RGB_LOGITS, stream);
}

// Fused Adam over every parameter tensor of the optimise step in ONE launch.
//
// Replaces torch.optim.Adam.step (reference slams/mapping.py:464,910, slams/tracking.py:120-124,339; default betas /
// eps, no weight decay, no amsgrad).  torch's foreach implementation issues ~15 launches per step
// over (table | MLPs | per-class pool | 2K pose tensors); here a tiny "tick" kernel advances the step count kept on
// the device (so the update is hipGraph-replayable) and publishes the two bias corrections, and one kernel walks all
// tensors.  Streaming, HBM-bound: 16 B read + 12 B written per parameter.
#include "common.hpp"

namespace dns {

constexpr int ADAM_MAX_TENSORS = 32;

struct AdamBatch {
  float* p[ADAM_MAX_TENSORS];
  const float* g[ADAM_MAX_TENSORS];
  float* m[ADAM_MAX_TENSORS];
  float* v[ADAM_MAX_TENSORS];
  uint32_t block_end[ADAM_MAX_TENSORS];   // exclusive prefix of workgroups per tensor
  uint32_t n[ADAM_MAX_TENSORS];
  float lr[ADAM_MAX_TENSORS];
  uint32_t n_tensors;
};

// state[0] = step count (float), state[1] = 1 - beta1^t, state[2] = 1 - beta2^t
__global__ void adam_tick_kernel(float* __restrict__ state, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float t = state[0] + 1.0f;
    state[0] = t;
    state[1] = 1.0f - powf(beta1, t);
    state[2] = 1.0f - powf(beta2, t);
  }
}

__global__ __launch_bounds__(256) void adam_step_kernel(AdamBatch b, const float* __restrict__ state, float beta1, float beta2,
                                                        float eps) {
  uint32_t t = 0;
  while (t + 1 < b.n_tensors && blockIdx.x >= b.block_end[t]) ++t;
  const uint32_t first = t ? b.block_end[t - 1] : 0u;
  const uint32_t i0 = (blockIdx.x - first) * 1024u + threadIdx.x * 4u;
  const uint32_t n = b.n[t];
  if (i0 >= n) return;
  const float bc1 = state[1], bc2 = state[2];
  const float step_size = b.lr[t] / bc1;
  const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  float* __restrict__ p = b.p[t];
  const float* __restrict__ g = b.g[t];
  float* __restrict__ m = b.m[t];
  float* __restrict__ v = b.v[t];
  const float ob1 = 1.0f - beta1, ob2 = 1.0f - beta2;
  if (i0 + 3u < n && (((uintptr_t)(p + i0) | (uintptr_t)(g + i0) | (uintptr_t)(m + i0) | (uintptr_t)(v + i0)) & 15u) == 0) {
    // the common case: four 16-byte accesses instead of sixteen guarded dwords
    const float4 g4 = *reinterpret_cast<const float4*>(g + i0);
    float4 m4 = *reinterpret_cast<const float4*>(m + i0), v4 = *reinterpret_cast<const float4*>(v + i0);
    float4 p4 = *reinterpret_cast<const float4*>(p + i0);
    const float ge[4] = {g4.x, g4.y, g4.z, g4.w};
    float me[4] = {m4.x, m4.y, m4.z, m4.w}, ve[4] = {v4.x, v4.y, v4.z, v4.w}, pe[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      me[e] = beta1 * me[e] + ob1 * ge[e];
      ve[e] = beta2 * ve[e] + ob2 * ge[e] * ge[e];
      const float denom = sqrtf(ve[e]) * inv_sqrt_bc2 + eps;
      pe[e] -= step_size * (me[e] / denom);
    }
    *reinterpret_cast<float4*>(m + i0) = make_float4(me[0], me[1], me[2], me[3]);
    *reinterpret_cast<float4*>(v + i0) = make_float4(ve[0], ve[1], ve[2], ve[3]);
    *reinterpret_cast<float4*>(p + i0) = make_float4(pe[0], pe[1], pe[2], pe[3]);
    return;
  }
#pragma unroll
  for (uint32_t e = 0; e < 4; ++e) {
    const uint32_t i = i0 + e;
    if (i < n) {
      const float gi = g[i];
      const float mi = beta1 * m[i] + ob1 * gi;                     // exp_avg.lerp_(grad, 1 - beta1)
      const float vi = beta2 * v[i] + ob2 * gi * gi;                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      m[i] = mi;
      v[i] = vi;
      const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;           // (exp_avg_sq.sqrt() / sqrt(bc2)).add_(eps)
      p[i] -= step_size * (mi / denom);                             // param.addcdiv_(exp_avg, denom, value = -step_size)
    }
  }
}

}  // namespace dns

using namespace dns;

extern "C" int dns_adam_step(const DnsAdamTensor* tensors, uint32_t n_tensors, float beta1, float beta2, float eps,
                             float* state, void* stream) {
  DNS_REQUIRE(tensors && state, "dns_adam_step: NULL argument");
  DNS_REQUIRE(n_tensors <= ADAM_MAX_TENSORS, "dns_adam_step: at most %d tensors per call (got %u)", ADAM_MAX_TENSORS, n_tensors);
  AdamBatch b;
  uint32_t nb = 0, k = 0;
  for (uint32_t i = 0; i < n_tensors; ++i) {
    if (!tensors[i].g || tensors[i].n == 0) continue;               // no gradient this step (e.g. a frozen pose)
    DNS_REQUIRE(tensors[i].p && tensors[i].m && tensors[i].v, "dns_adam_step: tensor %u has a NULL buffer", i);
    DNS_REQUIRE(tensors[i].n < (1ull << 32), "dns_adam_step: tensor %u too large", i);
    b.p[k] = tensors[i].p; b.g[k] = tensors[i].g; b.m[k] = tensors[i].m; b.v[k] = tensors[i].v;
    b.n[k] = (uint32_t)tensors[i].n;
    b.lr[k] = tensors[i].lr;
    nb += (uint32_t)((tensors[i].n + 1023) / 1024);
    b.block_end[k] = nb;
    ++k;
  }
  b.n_tensors = k;
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(adam_tick_kernel, dim3(1), dim3(64), 0, st, state, beta1, beta2);
  if (k) DNS_LAUNCH(adam_step_kernel, dim3(nb), dim3(256), 0, st, b, state, beta1, beta2, eps);
  return check_launch("dns_adam_step");
}

// Fused loss reductions of the optimise steps.
//
// Mapper (reference slams/mapping.py:110-126, 887-907 + utils/common.py:769-802):
//   p  = mean((gt_color - pred_color)^2)                 l  = cross_entropy(pred_logits, gt_label)
//   d  = mean |gt_depth - pred_depth| over gt_depth > 0  lt = mean((coarse - fine)^2) over [P, 33]
//   fs, opacity = get_opacity_loss(z, gt_depth, fine[..., -1], truncation = opacity_sigma, sigma = 0.05)  (D5)
// Tracker (slams/tracking.py:85-96): masked p, masked |d - d^| / sqrt(var + 1e-10), masked CE.
//
// The reference spends ~60 tiny launches (and two host syncs) on these; here: one pass over the rays and one over
// the points accumulate numerators and counts into a 16-float buffer, a 1-thread kernel turns them into the loss
// terms and the per-element gradient coefficients (masked-mean denominators, the count_nonzero branch flag of
// common.py:794 as a 0/1 factor), and the backward is one pass over rays and one over points.
// Between "sums" and "finalize" the caller may all-reduce the 16 floats (multi-GPU: global numerators and counts).
// Memory-bound streaming: 264 B read per sample forward, 264 B read + 264 B written backward.
#include <stdlib.h>
#include "common.hpp"

namespace dns {

// sums[] layout
enum { S_P = 0, S_D, S_L, S_LT, S_FS, S_OP, S_NV, S_ND, S_NFRONT, S_NOMASK, S_COUNT = 16,
       S_PARTIALS = 32, S_MAX_BLOCKS = 1024 };   // workspace behind the 16 results: DNS_LOSS_SUMS_FLOATS
constexpr uint32_t LOSS_POINT_BLOCKS = 1024;  // x 256 threads x 4 quads per trip; <= S_MAX_BLOCKS (5 partial sums each)
// out[] layout: terms p,d,l,lt,fs,op, total ; coefficients
enum { O_P = 0, O_D, O_L, O_LT, O_FS, O_OP, O_TOTAL, O_CP = 8, O_CD, O_CL, O_CLT, O_CFS, O_COP };

struct LossCfg {
  float lambda_p, lambda_d, lambda_l, lambda_lt, lambda_fs, lambda_op;
  float truncation, sigma;
  uint32_t N, S, C, L;     // rays, samples per ray, classes, latent width (33)
  int tracker;             // 1: depth term is |d-d^|/sqrt(var+1e-10), no latent / opacity terms
};

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) t += sh[w];
  return t;  // valid on thread 0
}

__device__ __forceinline__ bool ray_valid(const uint8_t* valid, uint32_t n) { return valid ? valid[n] != 0 : true; }

struct RayTerms {
  float sp, sd, sl, nv, nd;
};
// one ray's numerators and counts (slams/mapping.py:110-126, slams/tracking.py:85-96)
__device__ __forceinline__ RayTerms ray_terms(const LossCfg& c, uint32_t n, const float* __restrict__ pred_color,
                                              const float* __restrict__ pred_depth, const float* __restrict__ pred_var,
                                              const float* __restrict__ logits, const float* __restrict__ gt_color,
                                              const float* __restrict__ gt_depth, const int64_t* __restrict__ gt_label,
                                              const uint8_t* __restrict__ valid) {
  RayTerms r = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < c.N && ray_valid(valid, n)) {
    r.nv = 1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float e = gt_color[(size_t)n * 3 + k] - pred_color[(size_t)n * 3 + k];
      r.sp += e * e;
    }
    const float gd = gt_depth[n];
    if (c.tracker) {
      r.sd = fabsf(gd - pred_depth[n]) / sqrtf(pred_var[n] + 1e-10f);
      r.nd = 1.f;
    } else if (gd > 0.f) {
      r.sd = fabsf(gd - pred_depth[n]);
      r.nd = 1.f;
    }
    if (c.C) {
      const float* lg = logits + (size_t)n * c.C;
      float mx = -INFINITY;
      for (uint32_t k = 0; k < c.C; ++k) mx = fmaxf(mx, lg[k]);
      float se = 0.f;
      for (uint32_t k = 0; k < c.C; ++k) se += expf(lg[k] - mx);
      const int64_t lab = gt_label[n];
      const float ll = (lab >= 0 && lab < (int64_t)c.C) ? lg[lab] : 0.f;
      r.sl = (mx + logf(se)) - ll;
    }
  }
  return r;
}

__global__ __launch_bounds__(256) void loss_ray_sums_kernel(LossCfg c, const float* __restrict__ pred_color,
                                                            const float* __restrict__ pred_depth,
                                                            const float* __restrict__ pred_var,
                                                            const float* __restrict__ logits,
                                                            const float* __restrict__ gt_color,
                                                            const float* __restrict__ gt_depth,
                                                            const int64_t* __restrict__ gt_label,
                                                            const uint8_t* __restrict__ valid, float* __restrict__ sums,
                                                            uint32_t n_partials) {
  __shared__ float sh[16];
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  const RayTerms rt = ray_terms(c, n, pred_color, pred_depth, pred_var, logits, gt_color, gt_depth, gt_label, valid);
  const float sp = rt.sp, sd = rt.sd, sl = rt.sl, nv = rt.nv, nd = rt.nd;
  float t;
  t = block_sum(sp, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_P, t);
  t = block_sum(sd, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_D, t);
  t = block_sum(sl, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_L, t);
  t = block_sum(nv, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_NV, t);
  t = block_sum(nd, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_ND, t);
  // second stage of the point pass (loss_point_sums_kernel, launched BEFORE this kernel): workgroup 0 adds up the partial
  // sums the point workgroups parked behind the results.  The kernel boundary orders the two -- no fence, no ticket counter
  // (a device-scope release per workgroup, i.e. an L2 write-back on this multi-die part, was the point kernel's tail).
  if (blockIdx.x == 0 && n_partials) {
    const float* partial = sums + S_PARTIALS;
    const int dst[5] = {S_LT, S_FS, S_OP, S_NFRONT, S_NOMASK};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float v = 0.f;
      for (uint32_t b = threadIdx.x; b < n_partials; b += blockDim.x) v += partial[b * 5u + i];
      const float tt = block_sum(v, sh);
      if (threadIdx.x == 0) sums[dst[i]] = tt;               // only this workgroup writes these five words
    }
  }
}

__device__ __forceinline__ float sigmoid10f(float x) { return 1.0f / (1.0f + expf(-10.0f * x)); }

// Point pass over the FLAT [P * L] latent arrays (L = 33 is not a vector width: a thread-per-point walk reads 132-byte
// rows at a 132-byte lane stride; the flat walk is fully coalesced).  Each thread takes aligned float4s, two at a time,
// so that 64 bytes per lane are in flight (with one dword per trip the kernel ran at the memory LATENCY: 97 us for 69 MB);
// (point, channel, ray) follow the element index incrementally -- one pair of divisions per float4, not per element.
// The element that is a point's LAST channel also evaluates the free-space / opacity terms (D5: last channel = "occ").
struct PointAcc {
  float slt, sfs, sop, nfr, nom;
};

__device__ __forceinline__ void loss_point_elem(const LossCfg& c, float f, float co, uint32_t p, uint32_t k, uint32_t n,
                                                const float* __restrict__ z, const float* __restrict__ gt_depth,
                                                const uint8_t* __restrict__ valid, PointAcc& a) {
  if (!ray_valid(valid, n)) return;
  const float d0 = co - f;
  a.slt += d0 * d0;
  if (k + 1 == c.L) {
    const float d = gt_depth[n], zz = z[p];
    const float occ = sigmoid10f(f);
    const float front = zz < (d - c.truncation) ? 1.f : 0.f;
    const float back = zz > (d + c.truncation) ? 1.f : 0.f;
    const float dm = d > 0.f ? 1.f : 0.f;
    const float om = (1.f - front) * (1.f - back) * dm;
    const float a1 = occ * front * dm;
    a.sfs += a1 * a1;
    const float r = (zz - d) / c.sigma;
    const float pseudo = 0.5f * expf(-0.5f * r * r);
    const float a2 = occ * om - pseudo * om;
    a.sop += a2 * a2;
    a.nfr += front;
    a.nom += om;
  }
}

// One aligned float4 of the flat arrays.  Every side load (validity of the <= 2 rays it touches, z / depth of the at
// most one point whose last channel it holds: L > 4) is issued up front on clamped indices, so nothing in the
// arithmetic below waits on a dependent load.
__device__ __forceinline__ void loss_point_quad(const LossCfg& c, uint32_t e0, const float4& f4, const float4& c4,
                                                const float* __restrict__ z, const float* __restrict__ gt_depth,
                                                const uint8_t* __restrict__ valid, PointAcc& a) {
  const uint32_t p0 = e0 / c.L, k0 = e0 - p0 * c.L;
  const uint32_t n0 = p0 / c.S;
  const uint32_t p1 = p0 + (k0 + 3u >= c.L ? 1u : 0u);                  // point of the quad's last element
  const uint32_t n1 = (p1 != p0 && p1 - n0 * c.S == c.S) ? n0 + 1u : n0;
  const bool v0 = ray_valid(valid, n0), v1 = ray_valid(valid, n1);
  const uint32_t jl = c.L - 1u - k0;                                    // offset of point p0's last channel
  const bool has_last = jl < 4u;
  const float zz = z[p0], d = gt_depth[n0];                             // used only when has_last (p0 < N*S always)
  const float fv[4] = {f4.x, f4.y, f4.z, f4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool second = k0 + (uint32_t)j >= c.L;                        // element belongs to p1
    const bool ok = second ? v1 : v0;
    const float d0 = cv[j] - fv[j];
    a.slt += ok ? d0 * d0 : 0.f;
  }
  if (has_last && v0) {
    const float f = jl == 0 ? fv[0] : jl == 1 ? fv[1] : jl == 2 ? fv[2] : fv[3];
    const float occ = sigmoid10f(f);
    const float front = zz < (d - c.truncation) ? 1.f : 0.f;
    const float back = zz > (d + c.truncation) ? 1.f : 0.f;
    const float dm = d > 0.f ? 1.f : 0.f;
    const float om = (1.f - front) * (1.f - back) * dm;
    const float a1 = occ * front * dm;
    a.sfs += a1 * a1;
    const float r = (zz - d) / c.sigma;
    const float pseudo = 0.5f * expf(-0.5f * r * r);
    const float a2 = occ * om - pseudo * om;
    a.sop += a2 * a2;
    a.nfr += front;
    a.nom += om;
  }
}

__global__ __launch_bounds__(256) void loss_point_sums_kernel(LossCfg c, const float* __restrict__ fine,
                                                              const float* __restrict__ coarse,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ gt_depth,
                                                              const uint8_t* __restrict__ valid,
                                                              float* __restrict__ sums) {
  __shared__ float sh[4];
  // the 16 result words are cleared HERE (mapper mode): the ray kernel that adds into them is the next launch
  if (blockIdx.x == 0 && threadIdx.x < S_COUNT) sums[threadIdx.x] = 0.f;
  const uint32_t E = c.N * c.S * c.L;               // < 2^32 (checked on the host): 32-bit divides, not 64-bit
  PointAcc a = {0.f, 0.f, 0.f, 0.f, 0.f};
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gstride = gridDim.x * blockDim.x;
  const bool vec = ((((uintptr_t)fine) | ((uintptr_t)coarse)) & 15u) == 0;
  const uint32_t Q = vec ? E / 4u : 0u;
  const float4* __restrict__ f4p = reinterpret_cast<const float4*>(fine);
  const float4* __restrict__ c4p = reinterpret_cast<const float4*>(coarse);
  for (uint32_t q = gtid; q < Q; q += 4u * gstride) {
    // four quads per trip, clamped indices: eight unconditional 16-byte loads requested together (128 B per lane in flight)
    uint32_t qq[4];
    float4 fv[4], cv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      qq[u] = q + (uint32_t)u * gstride;
      const uint32_t qc = qq[u] < Q ? qq[u] : q;
      fv[u] = f4p[qc];
      cv[u] = c4p[qc];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (qq[u] < Q) loss_point_quad(c, 4u * qq[u], fv[u], cv[u], z, gt_depth, valid, a);
  }
  for (uint32_t e = 4u * Q + gtid; e < E; e += gstride) {      // unaligned arrays / the last E % 4 elements
    const uint32_t p = e / c.L, k = e - p * c.L;
    loss_point_elem(c, fine[e], coarse[e], p, k, p / c.S, z, gt_depth, valid, a);
  }
  // Two-stage reduction: every workgroup parks its 5 partial sums; loss_ray_sums_kernel (the next launch) adds them up.
  // (5 float atomics per workgroup on the same 5 words serialise at ~9 ns each: 33 us of this kernel's 67 at 768 workgroups.)
  float* partial = sums + S_PARTIALS;
  const float vals[5] = {a.slt, a.sfs, a.sop, a.nfr, a.nom};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float t = block_sum(vals[i], sh);
    if (threadIdx.x == 0) partial[blockIdx.x * 5u + i] = t;
  }
}

// sums -> the loss terms and the per-element gradient coefficients (one thread)
__device__ __forceinline__ void finalize_terms(const LossCfg& c, const float* sums, float* out) {
  const float nv = sums[S_NV], nd = sums[S_ND];
  const float np = nv * (float)c.S;                      // points of valid rays
  const float p = sums[S_P] / (3.f * nv);
  const float d = sums[S_D] / nd;                        // 0/0 -> NaN exactly like torch's mean of an empty selection
  const float l = c.C ? sums[S_L] / nv : 0.f;
  float lt = 0.f, fs = 0.f, op = 0.f, flag = 0.f;
  if (!c.tracker) {
    lt = sums[S_LT] / (np * (float)c.L);
    flag = (sums[S_NFRONT] > 0.f && sums[S_NOMASK] > 0.f) ? 1.f : 0.f;   // utils/common.py:794
    fs = flag * sums[S_FS] / np;
    op = flag * sums[S_OP] / np;
  }
  out[O_P] = p; out[O_D] = d; out[O_L] = l; out[O_LT] = lt; out[O_FS] = fs; out[O_OP] = op;
  out[O_TOTAL] = c.lambda_p * p + c.lambda_d * d + c.lambda_l * l + c.lambda_lt * lt + c.lambda_fs * fs + c.lambda_op * op;
  out[O_CP] = c.lambda_p * 2.f / (3.f * nv);
  out[O_CD] = c.lambda_d / nd;
  out[O_CL] = c.lambda_l / nv;
  out[O_CLT] = c.tracker ? 0.f : c.lambda_lt * 2.f / (np * (float)c.L);
  out[O_CFS] = c.tracker ? 0.f : flag * c.lambda_fs * 2.f / np;
  out[O_COP] = c.tracker ? 0.f : flag * c.lambda_op * 2.f / np;
}

__global__ void loss_finalize_kernel(LossCfg c, const float* __restrict__ sums, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  finalize_terms(c, sums, out);
}

// one ray's gradients of the three ray terms
__device__ __forceinline__ void ray_bwd_one(const LossCfg& c, const float* out, float g, uint32_t n,
                                            const float* __restrict__ pred_color, const float* __restrict__ pred_depth,
                                            const float* __restrict__ pred_var, const float* __restrict__ logits,
                                            const float* __restrict__ gt_color, const float* __restrict__ gt_depth,
                                            const int64_t* __restrict__ gt_label, const uint8_t* __restrict__ valid,
                                            float* __restrict__ d_color, float* __restrict__ d_depth, float* __restrict__ d_var,
                                            float* __restrict__ d_logits) {
  const bool ok = ray_valid(valid, n);
#pragma unroll
  for (int k = 0; k < 3; ++k)
    d_color[(size_t)n * 3 + k] = ok ? -g * out[O_CP] * (gt_color[(size_t)n * 3 + k] - pred_color[(size_t)n * 3 + k]) : 0.f;
  const float gd = gt_depth[n];
  float dd = 0.f, dv = 0.f;
  if (ok) {
    const float e = gd - pred_depth[n];
    const float sg = e > 0.f ? -1.f : (e < 0.f ? 1.f : 0.f);       // d|gt - pred| / dpred
    if (c.tracker) {
      const float sv = sqrtf(pred_var[n] + 1e-10f);
      dd = g * out[O_CD] * sg / sv;
      dv = g * out[O_CD] * fabsf(e) * (-0.5f) / (sv * sv * sv);
    } else if (gd > 0.f) {
      dd = g * out[O_CD] * sg;
    }
  }
  d_depth[n] = dd;
  if (d_var) d_var[n] = dv;
  if (c.C) {
    const float* lg = logits + (size_t)n * c.C;
    float* dl = d_logits + (size_t)n * c.C;
    if (!ok) {
      for (uint32_t k = 0; k < c.C; ++k) dl[k] = 0.f;
    } else {
      float mx = -INFINITY;
      for (uint32_t k = 0; k < c.C; ++k) mx = fmaxf(mx, lg[k]);
      float se = 0.f;
      for (uint32_t k = 0; k < c.C; ++k) se += expf(lg[k] - mx);
      const int64_t lab = gt_label[n];
      const float cl = g * out[O_CL];
      for (uint32_t k = 0; k < c.C; ++k) dl[k] = cl * (expf(lg[k] - mx) / se - ((int64_t)k == lab ? 1.f : 0.f));
    }
  }
}

__global__ __launch_bounds__(256) void loss_ray_bwd_kernel(LossCfg c, const float* __restrict__ out,
                                                           const float* __restrict__ g_total,
                                                           const float* __restrict__ pred_color,
                                                           const float* __restrict__ pred_depth,
                                                           const float* __restrict__ pred_var,
                                                           const float* __restrict__ logits,
                                                           const float* __restrict__ gt_color,
                                                           const float* __restrict__ gt_depth,
                                                           const int64_t* __restrict__ gt_label,
                                                           const uint8_t* __restrict__ valid, float* __restrict__ d_color,
                                                           float* __restrict__ d_depth, float* __restrict__ d_var,
                                                           float* __restrict__ d_logits, const float* __restrict__ sums,
                                                           float* __restrict__ out_w) {
  // sums != NULL (dns_loss_finalize_bwd): the finalize runs HERE -- every workgroup turns the 16 sums into the terms and
  // coefficients for itself (thirty flops), workgroup 0 also writes them to out_w for the kernels that follow; the one-thread
  // finalize launch in front of this kernel is gone from the iteration's critical chain
  __shared__ float s_out[16];
  if (sums) {
    if (threadIdx.x == 0) {
      finalize_terms(c, sums, s_out);
      s_out[7] = 0.f; s_out[14] = 0.f; s_out[15] = 0.f;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 16) out_w[threadIdx.x] = s_out[threadIdx.x];
    out = s_out;
  }
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= c.N) return;
  ray_bwd_one(c, out, g_total[0], n, pred_color, pred_depth, pred_var, logits, gt_color, gt_depth, gt_label, valid, d_color, d_depth,
              d_var, d_logits);
}

// Rays' sums + finalize + rays' backward as ONE single-workgroup launch (dns_loss_rays): the three are a few microseconds of
// work each behind a launch each -- on the mapper's critical chain, and four of the tracker's ~33 launches per iteration (its
// 16-word clear included).  Valid when nothing has to happen between the sums and the coefficients (no all-reduce).
// 1024 threads walk the rays twice; the sums are exact per-thread partials reduced in a fixed order (no atomics).
__global__ __launch_bounds__(1024) void loss_rays_fused_kernel(LossCfg c, const float* __restrict__ pred_color,
                                                               const float* __restrict__ pred_depth,
                                                               const float* __restrict__ pred_var,
                                                               const float* __restrict__ logits,
                                                               const float* __restrict__ gt_color,
                                                               const float* __restrict__ gt_depth,
                                                               const int64_t* __restrict__ gt_label,
                                                               const uint8_t* __restrict__ valid, float* __restrict__ sums,
                                                               uint32_t n_partials, float* __restrict__ out,
                                                               const float* __restrict__ g_total, float* __restrict__ d_color,
                                                               float* __restrict__ d_depth, float* __restrict__ d_var,
                                                               float* __restrict__ d_logits) {
  __shared__ float sh[16];
  __shared__ float s_sums[S_COUNT], s_out[16];
  RayTerms a = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (uint32_t n = threadIdx.x; n < c.N; n += blockDim.x) {
    const RayTerms r = ray_terms(c, n, pred_color, pred_depth, pred_var, logits, gt_color, gt_depth, gt_label, valid);
    a.sp += r.sp; a.sd += r.sd; a.sl += r.sl; a.nv += r.nv; a.nd += r.nd;
  }
  if (threadIdx.x < S_COUNT) s_sums[threadIdx.x] = 0.f;
  const float rv[5] = {a.sp, a.sd, a.sl, a.nv, a.nd};
  const int rd[5] = {S_P, S_D, S_L, S_NV, S_ND};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float t = block_sum(rv[i], sh);
    if (threadIdx.x == 0) s_sums[rd[i]] = t;
  }
  if (n_partials) {                                  // the point pass's parked partial sums (loss_point_sums_kernel, launched before)
    const float* partial = sums + S_PARTIALS;
    const int dst[5] = {S_LT, S_FS, S_OP, S_NFRONT, S_NOMASK};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float v = 0.f;
      for (uint32_t b = threadIdx.x; b < n_partials; b += blockDim.x) v += partial[b * 5u + i];
      const float t = block_sum(v, sh);
      if (threadIdx.x == 0) s_sums[dst[i]] = t;
    }
  }
  if (threadIdx.x == 0) {
    finalize_terms(c, s_sums, s_out);
    s_out[7] = 0.f; s_out[14] = 0.f; s_out[15] = 0.f;
  }
  __syncthreads();
  if (threadIdx.x < S_COUNT) {
    sums[threadIdx.x] = s_sums[threadIdx.x];
    out[threadIdx.x] = s_out[threadIdx.x];
  }
  const float g = g_total[0];
  for (uint32_t n = threadIdx.x; n < c.N; n += blockDim.x)
    ray_bwd_one(c, s_out, g, n, pred_color, pred_depth, pred_var, logits, gt_color, gt_depth, gt_label, valid, d_color, d_depth, d_var,
                d_logits);
}

__global__ __launch_bounds__(256) void loss_point_bwd_kernel(LossCfg c, const float* __restrict__ out,
                                                             const float* __restrict__ g_total,
                                                             const float* __restrict__ fine,
                                                             const float* __restrict__ coarse,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ gt_depth,
                                                             const uint8_t* __restrict__ valid, float* __restrict__ d_fine,
                                                             float* __restrict__ d_coarse, uint32_t ldd_fine,
                                                             const float* __restrict__ d_occ, uint32_t ld_occ) {
  const uint32_t E = c.N * c.S * c.L;
  const float g = g_total[0];
  const float clt = g * out[O_CLT], cfs = g * out[O_CFS], cop = g * out[O_COP];
  // One aligned float4 of the flat [P * L] arrays per step, two steps per trip (four 16-byte loads requested together); the
  // (point, channel, ray) of the quad's first element come from ONE pair of integer divisions and advance incrementally
  // (a pair of divisions per ELEMENT was a third of this kernel's time).
  auto elem = [&](uint32_t p, uint32_t k, uint32_t n, bool ok, float f, float co, float& df, float& dc) {
    // (d_occ: the compositing's gradient of the occupancy logit = column 0 of the fine latents, slams/mapping.py:626-627,
    //  added here instead of by a read-modify-write pass over the strided column)
    df = (k == 0 && d_occ) ? d_occ[(size_t)p * ld_occ] : 0.f;
    dc = 0.f;
    if (!ok) return;
    const float d0 = co - f;
    dc = clt * d0;
    df -= clt * d0;
    if (k + 1 == c.L) {
      const float d = gt_depth[n], zz = z[p];
      const float occ = sigmoid10f(f);
      const float front = zz < (d - c.truncation) ? 1.f : 0.f;
      const float back = zz > (d + c.truncation) ? 1.f : 0.f;
      const float dm = d > 0.f ? 1.f : 0.f;
      const float om = (1.f - front) * (1.f - back) * dm;
      const float r = (zz - d) / c.sigma;
      const float pseudo = 0.5f * expf(-0.5f * r * r);
      // d/docc of cfs/2*(occ*front*dm)^2 + cop/2*(occ*om - pseudo*om)^2, then occ' = 10 occ (1 - occ)
      const float docc = cfs * (occ * front * dm) * (front * dm) + cop * (occ * om - pseudo * om) * om;
      df += docc * 10.f * occ * (1.f - occ);
    }
  };
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  const bool vec = ((((uintptr_t)fine) | ((uintptr_t)coarse) | ((uintptr_t)d_coarse)) & 15u) == 0;
  const uint32_t Q = vec ? E / 4u : 0u;
  const float4* __restrict__ f4p = reinterpret_cast<const float4*>(fine);
  const float4* __restrict__ c4p = reinterpret_cast<const float4*>(coarse);
  for (uint32_t q0 = gtid; q0 < Q; q0 += 2u * stride) {
    const uint32_t q1 = q0 + stride;
    const bool two = q1 < Q;
    const uint32_t qq[2] = {q0, two ? q1 : q0};
    const float4 fa[2] = {f4p[qq[0]], f4p[qq[1]]}, ca[2] = {c4p[qq[0]], c4p[qq[1]]};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      const uint32_t e0 = 4u * qq[u];
      uint32_t p = e0 / c.L, k = e0 - p * c.L;
      uint32_t n = p / c.S, sidx = p - n * c.S;
      bool ok = ray_valid(valid, n);
      const float fv[4] = {fa[u].x, fa[u].y, fa[u].z, fa[u].w}, cv[4] = {ca[u].x, ca[u].y, ca[u].z, ca[u].w};
      float dcv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float df;
        elem(p, k, n, ok, fv[j], cv[j], df, dcv[j]);
        d_fine[(size_t)p * ldd_fine + k] = df;
        if (++k == c.L) {                                  // next point (at most once per quad: L > 4 is not required, just likely)
          k = 0;
          ++p;
          if (++sidx == c.S) {
            sidx = 0;
            ++n;
            ok = n < c.N ? ray_valid(valid, n) : false;
          }
        }
      }
      *reinterpret_cast<float4*>(d_coarse + e0) = make_float4(dcv[0], dcv[1], dcv[2], dcv[3]);
    }
  }
  for (uint32_t e = 4u * Q + gtid; e < E; e += stride) {       // unaligned arrays / the last E % 4 elements
    const uint32_t p = e / c.L, k = e - p * c.L;
    const uint32_t n = p / c.S;
    float df, dc;
    elem(p, k, n, ray_valid(valid, n), fine[e], coarse[e], df, dc);
    d_fine[(size_t)p * ldd_fine + k] = df;
    d_coarse[e] = dc;
  }
}

}  // namespace dns

using namespace dns;

static LossCfg make_cfg(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker) {
  LossCfg c;
  c.lambda_p = lambdas[0]; c.lambda_d = lambdas[1]; c.lambda_l = lambdas[2];
  c.lambda_lt = lambdas[3]; c.lambda_fs = lambdas[4]; c.lambda_op = lambdas[5];
  c.truncation = lambdas[6]; c.sigma = lambdas[7];
  c.N = N; c.S = S; c.C = C; c.L = L; c.tracker = tracker;
  return c;
}

extern "C" int dns_loss_sums(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                             const float* pred_color, const float* pred_depth, const float* pred_var,
                             const float* pred_logits, const float* gt_color, const float* gt_depth,
                             const int64_t* gt_label, const uint8_t* valid, const float* fine, const float* coarse,
                             const float* z, float* sums, void* stream) {
  DNS_REQUIRE(lambdas && sums, "dns_loss_sums: NULL argument");
  hipStream_t st = (hipStream_t)stream;
  if (tracker || N == 0) {                           // mapper mode: cleared by the point kernel
    const int rc = fill_words(sums, 0u, S_COUNT, st, "dns_loss_sums");
    if (rc != DNS_OK) return rc;
  }
  if (N == 0) return DNS_OK;
  DNS_REQUIRE(pred_color && pred_depth && gt_color && gt_depth, "dns_loss_sums: NULL ray tensor");
  DNS_REQUIRE(C == 0 || (pred_logits && gt_label), "dns_loss_sums: C > 0 needs logits and labels");
  DNS_REQUIRE(!tracker || pred_var, "dns_loss_sums: tracker mode needs pred_var");
  DNS_REQUIRE(tracker || (fine && coarse && z && L >= 1 && S >= 1), "dns_loss_sums: mapper mode needs fine, coarse, z");
  DNS_REQUIRE(tracker || (uint64_t)N * S * L < (1ull << 32) - (1ull << 22), "dns_loss_sums: N*S*L must stay below 2^32");
  const LossCfg c = make_cfg(lambdas, N, S, C, L, tracker);
  uint32_t blocks = 0;
  if (!tracker) {
    const uint64_t E = (uint64_t)N * S * L;
    // few, fat workgroups: each parks 5 partial sums that workgroup 0 of the ray kernel (launched next) adds up
    static const uint32_t cap_env = [] { const char* e = getenv("DNS_LOSS_BLOCKS"); const long n = e ? atol(e) : 0; return (uint32_t)(n >= 1 && n <= (long)S_MAX_BLOCKS ? n : 0); }();
    const uint32_t cap = cap_env ? cap_env : LOSS_POINT_BLOCKS;
    blocks = (uint32_t)((E + 255) / 256 < cap ? (E + 255) / 256 : cap);
    DNS_LAUNCH(loss_point_sums_kernel, dim3(blocks), dim3(256), 0, st, c, fine, coarse, z, gt_depth, valid, sums);
  }
  DNS_LAUNCH(loss_ray_sums_kernel, dim3((N + 255) / 256), dim3(256), 0, st, c, pred_color, pred_depth, pred_var,
                     pred_logits, gt_color, gt_depth, gt_label, valid, sums, blocks);
  return check_launch("dns_loss_sums");
}

extern "C" int dns_loss_rays(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                             const float* pred_color, const float* pred_depth, const float* pred_var, const float* pred_logits,
                             const float* gt_color, const float* gt_depth, const int64_t* gt_label, const uint8_t* valid,
                             const float* fine, const float* coarse, const float* z, float* sums, float* out, const float* g_total,
                             float* d_color, float* d_depth, float* d_var, float* d_logits, void* stream) {
  DNS_REQUIRE(lambdas && sums && out && g_total, "dns_loss_rays: NULL argument");
  DNS_REQUIRE(N >= 1 && N <= (1u << 20), "dns_loss_rays: N %u outside [1, 2^20] (use dns_loss_sums / _finalize / _bwd)", N);
  DNS_REQUIRE(pred_color && pred_depth && gt_color && gt_depth && d_color && d_depth, "dns_loss_rays: NULL ray tensor");
  DNS_REQUIRE(C == 0 || (pred_logits && gt_label && d_logits), "dns_loss_rays: C > 0 needs logits, labels and d_logits");
  DNS_REQUIRE(!tracker || pred_var, "dns_loss_rays: tracker mode needs pred_var");
  DNS_REQUIRE(tracker || (fine && coarse && z && L >= 1 && S >= 1), "dns_loss_rays: mapper mode needs fine, coarse, z");
  DNS_REQUIRE(tracker || (uint64_t)N * S * L < (1ull << 32) - (1ull << 22), "dns_loss_rays: N*S*L must stay below 2^32");
  const LossCfg c = make_cfg(lambdas, N, S, C, L, tracker);
  hipStream_t st = (hipStream_t)stream;
  uint32_t blocks = 0;
  if (!tracker) {
    const uint64_t E = (uint64_t)N * S * L;
    const uint32_t cap = LOSS_POINT_BLOCKS;
    blocks = (uint32_t)((E + 255) / 256 < cap ? (E + 255) / 256 : cap);
    DNS_LAUNCH(loss_point_sums_kernel, dim3(blocks), dim3(256), 0, st, c, fine, coarse, z, gt_depth, valid, sums);
  }
  DNS_LAUNCH(loss_rays_fused_kernel, dim3(1), dim3(1024), 0, st, c, pred_color, pred_depth, pred_var, pred_logits, gt_color, gt_depth,
             gt_label, valid, sums, blocks, out, g_total, d_color, d_depth, d_var, d_logits);
  return check_launch("dns_loss_rays");
}

extern "C" int dns_loss_finalize(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                                 const float* sums, float* out, void* stream) {
  DNS_REQUIRE(lambdas && sums && out, "dns_loss_finalize: NULL argument");
  const LossCfg c = make_cfg(lambdas, N, S, C, L, tracker);
  DNS_LAUNCH(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c, sums, out);
  return check_launch("dns_loss_finalize");
}

extern "C" int dns_loss_bwd(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                            const float* out, const float* g_total, const float* pred_color, const float* pred_depth,
                            const float* pred_var, const float* pred_logits, const float* gt_color,
                            const float* gt_depth, const int64_t* gt_label, const uint8_t* valid, const float* fine,
                            const float* coarse, const float* z, float* d_color, float* d_depth, float* d_var,
                            float* d_logits, float* d_fine, float* d_coarse, uint32_t ldd_fine, void* stream) {
  if (N == 0) return DNS_OK;
  DNS_REQUIRE(ldd_fine == 0 || ldd_fine >= L, "dns_loss_bwd: ldd_fine %u < L %u", ldd_fine, L);
  DNS_REQUIRE(lambdas && out && g_total && d_color && d_depth, "dns_loss_bwd: NULL argument");
  DNS_REQUIRE(C == 0 || d_logits, "dns_loss_bwd: C > 0 needs d_logits");
  DNS_REQUIRE(tracker || (d_fine != nullptr) == (d_coarse != nullptr), "dns_loss_bwd: d_fine and d_coarse come together (both NULL: "
              "the ray part only, the point part follows with dns_loss_bwd_points)");
  const LossCfg c = make_cfg(lambdas, N, S, C, L, tracker);
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(loss_ray_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, st, c, out, g_total, pred_color, pred_depth,
                     pred_var, pred_logits, gt_color, gt_depth, gt_label, valid, d_color, d_depth, d_var, d_logits, (const float*)nullptr,
                     (float*)nullptr);
  if (!tracker && d_fine) {
    const uint64_t E = (uint64_t)N * S * L;
    const uint32_t blocks = (uint32_t)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
    DNS_LAUNCH(loss_point_bwd_kernel, dim3(blocks), dim3(256), 0, st, c, out, g_total, fine, coarse, z, gt_depth,
                       valid, d_fine, d_coarse, ldd_fine ? ldd_fine : L, nullptr, 0u);
  }
  return check_launch("dns_loss_bwd");
}

extern "C" int dns_loss_finalize_bwd(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                                     const float* sums, float* out, const float* g_total, const float* pred_color,
                                     const float* pred_depth, const float* pred_var, const float* pred_logits, const float* gt_color,
                                     const float* gt_depth, const int64_t* gt_label, const uint8_t* valid, float* d_color,
                                     float* d_depth, float* d_var, float* d_logits, void* stream) {
  DNS_REQUIRE(N >= 1, "dns_loss_finalize_bwd: N must be >= 1 (dns_loss_finalize handles the empty batch)");
  DNS_REQUIRE(lambdas && sums && out && g_total && d_color && d_depth, "dns_loss_finalize_bwd: NULL argument");
  DNS_REQUIRE(C == 0 || d_logits, "dns_loss_finalize_bwd: C > 0 needs d_logits");
  const LossCfg c = make_cfg(lambdas, N, S, C, L, tracker);
  DNS_LAUNCH(loss_ray_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, c, (const float*)nullptr, g_total,
             pred_color, pred_depth, pred_var, pred_logits, gt_color, gt_depth, gt_label, valid, d_color, d_depth, d_var, d_logits, sums,
             out);
  return check_launch("dns_loss_finalize_bwd");
}

extern "C" int dns_loss_bwd_points(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, const float* out,
                                   const float* g_total, const float* gt_depth, const uint8_t* valid, const float* fine,
                                   const float* coarse, const float* z, float* d_fine, float* d_coarse, uint32_t ldd_fine,
                                   const float* d_occ, uint32_t ld_occ, void* stream) {
  if (N == 0) return DNS_OK;
  DNS_REQUIRE(lambdas && out && g_total && gt_depth && fine && coarse && z && d_fine && d_coarse, "dns_loss_bwd_points: NULL argument");
  DNS_REQUIRE(ldd_fine == 0 || ldd_fine >= L, "dns_loss_bwd_points: ldd_fine %u < L %u", ldd_fine, L);
  DNS_REQUIRE(!d_occ || ld_occ >= 1, "dns_loss_bwd_points: ld_occ");
  DNS_REQUIRE((uint64_t)N * S * L < (1ull << 32) - (1ull << 22), "dns_loss_bwd_points: N*S*L must stay below 2^32");
  const LossCfg c = make_cfg(lambdas, N, S, C, L, 0);
  const uint64_t E = (uint64_t)N * S * L;
  const uint32_t blocks = (uint32_t)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
  DNS_LAUNCH(loss_point_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c, out, g_total, fine, coarse, z, gt_depth, valid,
             d_fine, d_coarse, ldd_fine ? ldd_fine : L, d_occ, ld_occ);
  return check_launch("dns_loss_bwd_points");
}

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
#include "common.hpp"

namespace dns {

// sums[] layout
enum { S_P = 0, S_D, S_L, S_LT, S_FS, S_OP, S_NV, S_ND, S_NFRONT, S_NOMASK, S_COUNT = 16 };
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

__global__ __launch_bounds__(256) void loss_ray_sums_kernel(LossCfg c, const float* __restrict__ pred_color,
                                                            const float* __restrict__ pred_depth,
                                                            const float* __restrict__ pred_var,
                                                            const float* __restrict__ logits,
                                                            const float* __restrict__ gt_color,
                                                            const float* __restrict__ gt_depth,
                                                            const int64_t* __restrict__ gt_label,
                                                            const uint8_t* __restrict__ valid, float* __restrict__ sums) {
  __shared__ float sh[4];
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  float sp = 0.f, sd = 0.f, sl = 0.f, nv = 0.f, nd = 0.f;
  if (n < c.N && ray_valid(valid, n)) {
    nv = 1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float e = gt_color[(size_t)n * 3 + k] - pred_color[(size_t)n * 3 + k];
      sp += e * e;
    }
    const float gd = gt_depth[n];
    if (c.tracker) {
      sd = fabsf(gd - pred_depth[n]) / sqrtf(pred_var[n] + 1e-10f);
      nd = 1.f;
    } else if (gd > 0.f) {
      sd = fabsf(gd - pred_depth[n]);
      nd = 1.f;
    }
    if (c.C) {
      const float* lg = logits + (size_t)n * c.C;
      float mx = -INFINITY;
      for (uint32_t k = 0; k < c.C; ++k) mx = fmaxf(mx, lg[k]);
      float se = 0.f;
      for (uint32_t k = 0; k < c.C; ++k) se += expf(lg[k] - mx);
      const int64_t lab = gt_label[n];
      const float ll = (lab >= 0 && lab < (int64_t)c.C) ? lg[lab] : 0.f;
      sl = (mx + logf(se)) - ll;
    }
  }
  float t;
  t = block_sum(sp, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_P, t);
  t = block_sum(sd, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_D, t);
  t = block_sum(sl, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_L, t);
  t = block_sum(nv, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_NV, t);
  t = block_sum(nd, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_ND, t);
}

__device__ __forceinline__ float sigmoid10f(float x) { return 1.0f / (1.0f + expf(-10.0f * x)); }

// Point pass, one thread per ELEMENT of the [P, L] latent arrays (L = 33 is not a vector width: a thread-per-point
// walk reads 132-byte rows at a 132-byte lane stride; the flat walk is fully coalesced).  The thread that owns a
// point's LAST channel also evaluates the free-space / opacity terms of that point (D5: last channel = "occ").
__global__ __launch_bounds__(256) void loss_point_sums_kernel(LossCfg c, const float* __restrict__ fine,
                                                              const float* __restrict__ coarse,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ gt_depth,
                                                              const uint8_t* __restrict__ valid,
                                                              float* __restrict__ sums) {
  __shared__ float sh[4];
  const uint32_t E = c.N * c.S * c.L;               // < 2^32 (checked on the host): 32-bit divides, not 64-bit
  float slt = 0.f, sfs = 0.f, sop = 0.f, nfr = 0.f, nom = 0.f;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    const uint32_t p = e / c.L, k = e - p * c.L;
    const uint32_t n = p / c.S;
    if (!ray_valid(valid, n)) continue;
    const float f = fine[e];
    const float d0 = coarse[e] - f;
    slt += d0 * d0;
    if (k + 1 == c.L) {
      const float d = gt_depth[n], zz = z[p];
      const float occ = sigmoid10f(f);
      const float front = zz < (d - c.truncation) ? 1.f : 0.f;
      const float back = zz > (d + c.truncation) ? 1.f : 0.f;
      const float dm = d > 0.f ? 1.f : 0.f;
      const float om = (1.f - front) * (1.f - back) * dm;
      const float a1 = occ * front * dm;
      sfs += a1 * a1;
      const float r = (zz - d) / c.sigma;
      const float pseudo = 0.5f * expf(-0.5f * r * r);
      const float a2 = occ * om - pseudo * om;
      sop += a2 * a2;
      nfr += front;
      nom += om;
    }
  }
  float t;
  t = block_sum(slt, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_LT, t);
  t = block_sum(sfs, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_FS, t);
  t = block_sum(sop, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_OP, t);
  t = block_sum(nfr, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_NFRONT, t);
  t = block_sum(nom, sh); if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + S_NOMASK, t);
}

__global__ void loss_finalize_kernel(LossCfg c, const float* __restrict__ sums, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
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
                                                           float* __restrict__ d_logits) {
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= c.N) return;
  const float g = g_total[0];
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

__global__ __launch_bounds__(256) void loss_point_bwd_kernel(LossCfg c, const float* __restrict__ out,
                                                             const float* __restrict__ g_total,
                                                             const float* __restrict__ fine,
                                                             const float* __restrict__ coarse,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ gt_depth,
                                                             const uint8_t* __restrict__ valid, float* __restrict__ d_fine,
                                                             float* __restrict__ d_coarse) {
  const uint32_t E = c.N * c.S * c.L;
  const float g = g_total[0];
  const float clt = g * out[O_CLT], cfs = g * out[O_CFS], cop = g * out[O_COP];
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    const uint32_t p = e / c.L, k = e - p * c.L;
    const uint32_t n = p / c.S;
    float df = 0.f, dc = 0.f;
    if (ray_valid(valid, n)) {
      const float f = fine[e];
      const float d0 = coarse[e] - f;
      dc = clt * d0;
      df = -clt * d0;
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
    }
    d_fine[e] = df;
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
  if (hipMemsetAsync(sums, 0, sizeof(float) * S_COUNT, st) != hipSuccess) {
    set_error("dns_loss_sums: memset failed");
    return DNS_E_LAUNCH;
  }
  if (N == 0) return DNS_OK;
  DNS_REQUIRE(pred_color && pred_depth && gt_color && gt_depth, "dns_loss_sums: NULL ray tensor");
  DNS_REQUIRE(C == 0 || (pred_logits && gt_label), "dns_loss_sums: C > 0 needs logits and labels");
  DNS_REQUIRE(!tracker || pred_var, "dns_loss_sums: tracker mode needs pred_var");
  DNS_REQUIRE(tracker || (fine && coarse && z && L >= 1 && S >= 1), "dns_loss_sums: mapper mode needs fine, coarse, z");
  DNS_REQUIRE(tracker || (uint64_t)N * S * L < (1ull << 32) - (1ull << 22), "dns_loss_sums: N*S*L must stay below 2^32");
  const LossCfg c = make_cfg(lambdas, N, S, C, L, tracker);
  hipLaunchKernelGGL(loss_ray_sums_kernel, dim3((N + 255) / 256), dim3(256), 0, st, c, pred_color, pred_depth, pred_var,
                     pred_logits, gt_color, gt_depth, gt_label, valid, sums);
  if (!tracker) {
    const uint64_t E = (uint64_t)N * S * L;
    // few, fat workgroups: every workgroup ends in 5 atomics on the same 5 words (same-address atomics serialise)
    const uint32_t blocks = (uint32_t)((E + 255) / 256 < 768 ? (E + 255) / 256 : 768);
    hipLaunchKernelGGL(loss_point_sums_kernel, dim3(blocks), dim3(256), 0, st, c, fine, coarse, z, gt_depth, valid, sums);
  }
  return check_launch("dns_loss_sums");
}

extern "C" int dns_loss_finalize(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                                 const float* sums, float* out, void* stream) {
  DNS_REQUIRE(lambdas && sums && out, "dns_loss_finalize: NULL argument");
  const LossCfg c = make_cfg(lambdas, N, S, C, L, tracker);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c, sums, out);
  return check_launch("dns_loss_finalize");
}

extern "C" int dns_loss_bwd(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                            const float* out, const float* g_total, const float* pred_color, const float* pred_depth,
                            const float* pred_var, const float* pred_logits, const float* gt_color,
                            const float* gt_depth, const int64_t* gt_label, const uint8_t* valid, const float* fine,
                            const float* coarse, const float* z, float* d_color, float* d_depth, float* d_var,
                            float* d_logits, float* d_fine, float* d_coarse, void* stream) {
  if (N == 0) return DNS_OK;
  DNS_REQUIRE(lambdas && out && g_total && d_color && d_depth, "dns_loss_bwd: NULL argument");
  DNS_REQUIRE(C == 0 || d_logits, "dns_loss_bwd: C > 0 needs d_logits");
  DNS_REQUIRE(tracker || (d_fine && d_coarse), "dns_loss_bwd: mapper mode needs d_fine and d_coarse");
  const LossCfg c = make_cfg(lambdas, N, S, C, L, tracker);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_ray_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, st, c, out, g_total, pred_color, pred_depth,
                     pred_var, pred_logits, gt_color, gt_depth, gt_label, valid, d_color, d_depth, d_var, d_logits);
  if (!tracker) {
    const uint64_t E = (uint64_t)N * S * L;
    const uint32_t blocks = (uint32_t)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
    hipLaunchKernelGGL(loss_point_bwd_kernel, dim3(blocks), dim3(256), 0, st, c, out, g_total, fine, coarse, z, gt_depth,
                       valid, d_fine, d_coarse);
  }
  return check_launch("dns_loss_bwd");
}

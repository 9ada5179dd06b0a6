// Glue of one mapping iteration (slams/mapping.py:590-627, 553-556, 129-143) as single launches.  Each of these is a
// handful of elementwise torch ops in the reference (and in this package's autograd path); written out they cost one launch
// each and no intermediate tensors.  HBM-bound streaming kernels, no LDS, no matrix work.
#include "common.hpp"
#include "split_rows.hpp"

namespace dns {

// class id of point p (tiled: label[p mod N], the reference's layout; else label[p / S]) -> row of the pooled
// per-class parameters through the class -> row table (-1: no network)
__global__ __launch_bounds__(256) void class_slots_kernel(const int64_t* __restrict__ labels, uint32_t N, uint32_t S, uint32_t tiled,
                                                          const int64_t* __restrict__ lut, uint32_t n_lut,
                                                          int64_t* __restrict__ slot) {
  const uint32_t P = N * S;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const int64_t c = labels[tiled ? p % N : p / S];
    slot[p] = (c >= 0 && c < (int64_t)n_lut) ? lut[c] : (int64_t)-1;
  }
}

// feat[p] = (fine[p, 1 : 1 + hidden] | code[p] * trunc(p)),  raw[p, 3] = fine[p, 0]
// trunc = (1 - [z < 0.95 d]) (1 - [z > 1.05 d]) [d > 0]   (slams/mapping.py:553-556; d = the ray's measured depth)
// One thread per (point, quad q): it moves latent quad q AND code quad q of the row (whichever exist), so that every lane of
// a wave runs the same straight-line code and all its loads are requested before the first store.
__global__ __launch_bounds__(256) void feature_block_kernel(const float* __restrict__ fine, uint32_t ld_fine, uint32_t hidden,
                                                            const float* __restrict__ code, uint32_t C,
                                                            const float* __restrict__ z, const float* __restrict__ gt_depth,
                                                            uint32_t P, uint32_t S, float* __restrict__ feat, uint32_t ld_feat,
                                                            float* __restrict__ raw) {
  const uint32_t qh = hidden / 4u, qc = C / 4u, qn = qh > qc ? qh : qc;       // quads per point handled by consecutive lanes
  const uint32_t total = P * qn;                                             // < 2^32 (host check)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t p = i / qn, q = i - p * qn;
    const bool lat = q < qh, cod = q < qc;
    const float* f = fine + (size_t)p * ld_fine + 1u + 4u * (lat ? q : 0u);      // 4-byte aligned only: scalar loads
    const float f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3], occ = f[-1];
    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
    float t = 0.f;
    if (code) {
      cv = *reinterpret_cast<const float4*>(code + (size_t)p * C + 4u * (cod ? q : 0u));
      const float d = gt_depth[p / S], zz = z[p];
      const float front = zz < d * 0.95f ? 1.f : 0.f, back = zz > d * 1.05f ? 1.f : 0.f, pos = d > 0.f ? 1.f : 0.f;
      t = (1.f - front) * (1.f - back) * pos;
    }
    float* o = feat + (size_t)p * ld_feat;
    if (lat) *reinterpret_cast<float4*>(o + 4u * q) = make_float4(f0, f1, f2, f3);
    if (cod) *reinterpret_cast<float4*>(o + hidden + 4u * q) = make_float4(cv.x * t, cv.y * t, cv.z * t, cv.w * t);
    if (q == 0 && raw) raw[(size_t)p * 4u + 3u] = occ;
  }
}

// The same block written in the split-row format (split_rows.hpp) for the colour / logit networks' second input segment, plus
// -- optionally -- as fp32 rows (the streaming dW_in kernel reads fp32).  The point's qn lanes hold its hidden + C values: the
// row maximum is three or four lane-exchange steps.  The code may come as n_ref slabs that are AVERAGED (the mean over the
// reference frames of Decoder.merge's latents, models/decoder.py:76): code [n_frames][n_ref][pts_per_frame][C].
__global__ __launch_bounds__(256) void feature_block_split_kernel(const float* __restrict__ fine, uint32_t ld_fine, uint32_t hidden,
                                                                  const float* __restrict__ code, uint32_t C, uint32_t n_ref,
                                                                  uint32_t pts_per_frame, const float* __restrict__ z,
                                                                  const float* __restrict__ gt_depth, uint32_t P, uint32_t S,
                                                                  float* __restrict__ feat, uint32_t ld_feat,
                                                                  _Float16* __restrict__ xs, uint32_t ldxs, int32_t* __restrict__ xexp,
                                                                  uint32_t hi_only, float* __restrict__ raw, uint32_t qn) {
  const uint32_t qh = hidden / 4u, qc = C / 4u;
  const uint32_t total = P * qn, F = hidden + C;
  const float inv_ref = n_ref > 1u ? 1.0f / (float)n_ref : 1.0f;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t p = i / qn, q = i - p * qn;
    const bool lat = q < qh, cod = q < qc;
    const float* f = fine + (size_t)p * ld_fine + 1u + 4u * (lat ? q : 0u);
    float f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
    const float occ = f[-1];
    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (code) {
      float t;
      {
        const float d = gt_depth[p / S], zz = z[p];
        const float front = zz < d * 0.95f ? 1.f : 0.f, back = zz > d * 1.05f ? 1.f : 0.f, pos = d > 0.f ? 1.f : 0.f;
        t = (1.f - front) * (1.f - back) * pos;
      }
      if (n_ref <= 1u) {
        cv = *reinterpret_cast<const float4*>(code + (size_t)p * C + 4u * (cod ? q : 0u));
      } else {
        const uint32_t fr = p / pts_per_frame, pl = p - fr * pts_per_frame;
        const float* c0 = code + ((size_t)fr * n_ref * pts_per_frame + pl) * C + 4u * (cod ? q : 0u);
        for (uint32_t r = 0; r < n_ref; ++r) {
          const float4 v = *reinterpret_cast<const float4*>(c0 + (size_t)r * pts_per_frame * C);
          cv.x += v.x; cv.y += v.y; cv.z += v.z; cv.w += v.w;
        }
        cv.x *= inv_ref; cv.y *= inv_ref; cv.z *= inv_ref; cv.w *= inv_ref;
      }
      cv.x *= t; cv.y *= t; cv.z *= t; cv.w *= t;
    }
    if (!lat) f0 = f1 = f2 = f3 = 0.f;
    if (!cod) cv = make_float4(0.f, 0.f, 0.f, 0.f);
    float m = fmaxf(fmaxf(fmaxf(fabsf(f0), fabsf(f1)), fmaxf(fabsf(f2), fabsf(f3))),
                    fmaxf(fmaxf(fabsf(cv.x), fabsf(cv.y)), fmaxf(fabsf(cv.z), fabsf(cv.w))));
    const float sum = ((f0 + f1) + (f2 + f3)) + ((cv.x + cv.y) + (cv.z + cv.w));
    if (!(fabsf(sum) < INFINITY)) m = INFINITY;            // fmaxf drops a NaN: a non-finite row keeps exponent 0 and propagates
    for (uint32_t o = 1; o < qn; o <<= 1) m = fmaxf(m, __shfl_xor(m, (int)o));
    const int e = (hi_only & 2u) ? 0 : sr::scale_exp(m);      // (DNS_SPLIT_PLAIN: half rows -- unscaled f16, no exponent)
    const float sc = ldexpf(1.0f, e);
    if (feat) {
      float* o = feat + (size_t)p * ld_feat;
      if (lat) *reinterpret_cast<float4*>(o + 4u * q) = make_float4(f0, f1, f2, f3);
      if (cod) *reinterpret_cast<float4*>(o + hidden + 4u * q) = cv;
    }
    if (xs == nullptr) {
      if (q == 0 && raw) raw[(size_t)p * 4u + 3u] = occ;
      continue;
    }
    _Float16* row = xs + (size_t)p * ldxs;
    uint2 h, l;
    if (lat) {
      sr::split_pair(f0, f1, sc, h.x, l.x);
      sr::split_pair(f2, f3, sc, h.y, l.y);
      *reinterpret_cast<uint2*>(row + 4u * q) = h;
      if (!hi_only) *reinterpret_cast<uint2*>(row + F + 4u * q) = l;
    }
    if (cod) {
      sr::split_pair(cv.x, cv.y, sc, h.x, l.x);
      sr::split_pair(cv.z, cv.w, sc, h.y, l.y);
      *reinterpret_cast<uint2*>(row + hidden + 4u * q) = h;
      if (!hi_only) *reinterpret_cast<uint2*>(row + F + hidden + 4u * q) = l;
    }
    if (q == 0) {
      if (!(hi_only & 2u)) xexp[p] = e;
      if (raw) raw[(size_t)p * 4u + 3u] = occ;
    }
  }
}

__device__ __forceinline__ float sigmoid1(float x) { return 1.f / (1.f + expf(-x)); }

// raw[p, 0:3] = sigmoid(raw[p, 0:3]) in place; column 3 (the occupancy logit) is left alone
__global__ __launch_bounds__(256) void rgb_sigmoid_kernel(float4* __restrict__ raw, uint32_t P) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    float4 v = raw[p];
    v.x = sigmoid1(v.x); v.y = sigmoid1(v.y); v.z = sigmoid1(v.z);
    raw[p] = v;
  }
}

// d_col[p, 0:3] = d_raw[p, 0:3] * s (1 - s)  (s = raw[p, 0:3], the sigmoid's output), d_col[p, 3] = 0;
// d_occ[p * ld_occ] (+)= d_raw[p, 3]
__global__ __launch_bounds__(256) void raw_bwd_kernel(const float4* __restrict__ d_raw, const float4* __restrict__ raw, uint32_t P,
                                                      float4* __restrict__ d_col, float* __restrict__ d_occ, uint32_t ld_occ,
                                                      uint32_t accumulate) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const float4 g = d_raw[p], s = raw[p];
    d_col[p] = make_float4(g.x * (1.f - s.x) * s.x, g.y * (1.f - s.y) * s.y, g.z * (1.f - s.z) * s.z, 0.f);
    float* o = d_occ + (size_t)p * ld_occ;
    *o = accumulate ? *o + g.w : g.w;
  }
}

// pts[i, j, k] = float( (c_mar + r_off * c_off + r_jit * c_vox) + (i, j, k) * c_vox )  in float64 -- the normalised
// coordinates of the smoothness lattice (slams/mapping.py:133-143 folded to one affine map, Mapper.smoothness)
struct LatticeConsts { double vox[3], off[3], mar[3]; };
__global__ __launch_bounds__(256) void lattice_points_kernel(const float* __restrict__ r6, LatticeConsts c, uint32_t n,
                                                             const int32_t* __restrict__ order, uint32_t total,
                                                             float* __restrict__ pts) {
  double b[3];
  for (int a = 0; a < 3; ++a) {
    const double t = c.mar[a] + (double)r6[a] * c.off[a];          // addcmul, then addcmul: two rounded steps each
    b[a] = t + (double)r6[3 + a] * c.vox[a];
  }
  for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < total; m += gridDim.x * blockDim.x) {
    const uint32_t e = order ? (uint32_t)order[m] : m;             // output row m holds lattice element e (x-major index)
    const uint32_t k = e % n, j = (e / n) % n, i = e / (n * n);
    pts[3u * m + 0u] = (float)(b[0] + (double)i * c.vox[0]);
    pts[3u * m + 1u] = (float)(b[1] + (double)j * c.vox[1]);
    pts[3u * m + 2u] = (float)(b[2] + (double)k * c.vox[2]);
  }
}

// The index arithmetic behind the pixel draws of one iteration (select_uv + select_by_class, utils/common.py:274,313-328, all K
// frames at once) and what follows from the drawn pixels alone: per frame f, pix[f, 0:n1] = i1[f, :] (uniform picks),
// pix[f, n1 + s] = sorted[start[f, s] + min(int64(u[f, s] * count[f, s]), count[f, s] - 1)] (class-balanced picks from the
// frame's pixels sorted by class), labels[f, r] = int64(label[f, pix]) and dmax[f] = max(0, max_r depth[f, pix]) as float bits
// (what sample_along_rays takes the maximum of, :581,591).  One workgroup per frame: the maximum needs no atomics.
__global__ __launch_bounds__(256) void draw_finish_kernel(const int64_t* __restrict__ i1, const double* __restrict__ u,
                                                          const double* __restrict__ count_f64, const int64_t* __restrict__ count_m1,
                                                          const int64_t* __restrict__ start_flat, const int64_t* __restrict__ sorted_flat,
                                                          const float* __restrict__ depth, const float* __restrict__ label,
                                                          uint32_t n1, uint32_t n2, uint32_t HW, int64_t* __restrict__ pix,
                                                          int64_t* __restrict__ labels, uint32_t* __restrict__ dmax) {
  __shared__ float red[4];
  const uint32_t f = blockIdx.x, npf = n1 + n2;
  float m = 0.f;
  for (uint32_t r = threadIdx.x; r < npf; r += blockDim.x) {
    int64_t p;
    if (r < n1) {
      p = i1[(size_t)f * n1 + r];
    } else {
      const size_t s = (size_t)f * n2 + (r - n1);
      int64_t j = (int64_t)(u[s] * count_f64[s]);
      j = j < count_m1[s] ? j : count_m1[s];
      p = sorted_flat[start_flat[s] + j];
    }
    pix[(size_t)f * npf + r] = p;
    labels[(size_t)f * npf + r] = (int64_t)label[(size_t)f * HW + p];
    m = fmaxf(m, depth[(size_t)f * HW + p]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) dmax[f] = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// ---- tracker glue (slams/tracking.py:171-172, 326-335; utils/common.py:571-574) --------------------------------------------
// valid[n] = gt_depth[n] > min_depth && inside[n]
__global__ __launch_bounds__(256) void track_mask_kernel(const float* __restrict__ gt_depth, const uint8_t* __restrict__ inside,
                                                         uint32_t N, float min_depth, uint8_t* __restrict__ valid) {
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) valid[n] = (gt_depth[n] > min_depth && inside[n] != 0) ? 1 : 0;
}

// keep-best of the tracking loop: if (loss < best_loss) { best_loss = loss; best_cam = (quat | trans) } -- the camera that
// PRODUCED the loss, i.e. before this iteration's Adam step (a NaN loss compares false and keeps the old best)
__global__ void keep_best_kernel(const float* __restrict__ loss, const float* __restrict__ quat, const float* __restrict__ trans,
                                 float* __restrict__ best_loss, float* __restrict__ best_cam) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float l = loss[0];
  if (l < best_loss[0]) {
    best_loss[0] = l;
    for (int i = 0; i < 4; ++i) best_cam[i] = quat[i];
    for (int i = 0; i < 3; ++i) best_cam[4 + i] = trans[i];
  }
}

// sample_along_rays' forced mid-sample: t[idx] = 0.5 unless some t already equals 0.5 (utils/common.py:572-574); one wave
__global__ __launch_bounds__(64) void force_half_kernel(float* __restrict__ t, uint32_t n, uint32_t idx) {
  bool any = false;
  for (uint32_t i = threadIdx.x; i < n; i += 64u) any = any || t[i] == 0.5f;
  if (!__any(any) && threadIdx.x == 0 && idx < n) t[idx] = 0.5f;
}

}  // namespace dns

using namespace dns;

static inline uint32_t grid_for(uint64_t n, uint32_t cap = 8192) {
  const uint64_t b = (n + 255) / 256;
  return (uint32_t)(b < cap ? (b ? b : 1) : cap);
}

extern "C" int dns_class_slots(const int64_t* labels, uint32_t N, uint32_t S, int tiled, const int64_t* lut, uint32_t n_lut,
                               int64_t* slot_of_point, void* stream) {
  if ((uint64_t)N * S == 0) return DNS_OK;
  DNS_REQUIRE(labels && lut && slot_of_point, "dns_class_slots: NULL argument");
  DNS_REQUIRE((uint64_t)N * S < (1ull << 31), "dns_class_slots: too many points");
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(class_slots_kernel, dim3(grid_for((uint64_t)N * S)), dim3(256), 0, st, labels, N, S, tiled ? 1u : 0u, lut, n_lut,
             slot_of_point);
  return check_launch("dns_class_slots");
}

extern "C" int dns_feature_block(const float* fine, uint32_t ld_fine, uint32_t hidden, const float* code, uint32_t C,
                                 const float* z, const float* gt_depth, uint32_t N, uint32_t S, float* feat, uint32_t ld_feat,
                                 float* raw, void* stream) {
  const uint64_t P = (uint64_t)N * S;
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(fine && feat, "dns_feature_block: NULL argument");
  DNS_REQUIRE(P < (1ull << 31), "dns_feature_block: too many points");
  DNS_REQUIRE(hidden % 4 == 0 && C % 4 == 0 && hidden >= 4 && ld_feat % 4 == 0 && ld_feat >= hidden + C &&
              ld_fine >= hidden + 1, "dns_feature_block: hidden %u / C %u / ld_feat %u / ld_fine %u", hidden, C, ld_feat, ld_fine);
  DNS_REQUIRE(((uintptr_t)feat & 15) == 0 && (!code || ((uintptr_t)code & 15) == 0), "dns_feature_block: feat / code must be 16-byte aligned");
  DNS_REQUIRE(!code || (z && gt_depth), "dns_feature_block: a code needs z and gt_depth");
  hipStream_t st = (hipStream_t)stream;
  const uint32_t qn = (hidden > C ? hidden : C) / 4u;
  DNS_REQUIRE(P * qn < (1ull << 32), "dns_feature_block: too many points");
  DNS_LAUNCH(feature_block_kernel, dim3(grid_for(P * qn, 16384)), dim3(256), 0, st, fine, ld_fine, hidden, code, C, z,
             gt_depth, (uint32_t)P, S, feat, ld_feat, raw);
  return check_launch("dns_feature_block");
}

extern "C" int dns_feature_block_split(const float* fine, uint32_t ld_fine, uint32_t hidden, const float* code, uint32_t C,
                                       uint32_t n_ref, uint32_t pts_per_frame, const float* z, const float* gt_depth, uint32_t N,
                                       uint32_t S, float* feat, uint32_t ld_feat, void* xs_out, uint32_t ldxs, int32_t* xexp,
                                       uint32_t flags, float* raw, void* stream) {
  const uint64_t P = (uint64_t)N * S;
  if (P == 0) return DNS_OK;
  const bool plain = (flags & DNS_SPLIT_PLAIN) != 0;       // half rows: unscaled f16, no exponents, hi plane only
  DNS_REQUIRE(fine && (xs_out ? (xexp != nullptr || plain) : feat != nullptr), "dns_feature_block_split: NULL argument");
  DNS_REQUIRE((flags & ~(DNS_SPLIT_HI_ONLY | DNS_SPLIT_PLAIN)) == 0, "dns_feature_block_split: unknown flags 0x%x", flags);
  DNS_REQUIRE(P < (1ull << 31), "dns_feature_block_split: too many points");
  const uint32_t F = hidden + C;
  const bool hi_only = (flags & DNS_SPLIT_HI_ONLY) != 0 || plain;
  DNS_REQUIRE(hidden % 4 == 0 && C % 4 == 0 && hidden >= 4 && ld_fine >= hidden + 1 && F % 8 == 0,
              "dns_feature_block_split: hidden %u / C %u / ld_fine %u", hidden, C, ld_fine);
  DNS_REQUIRE(!feat || (ld_feat % 4 == 0 && ld_feat >= F && ((uintptr_t)feat & 15) == 0), "dns_feature_block_split: feat alignment / ld_feat");
  DNS_REQUIRE(!xs_out || (ldxs % 8 == 0 && ldxs >= (hi_only ? F : 2 * F) && ((uintptr_t)xs_out & 15) == 0),
              "dns_feature_block_split: xs_out must be 16-byte aligned with ldxs %% 8 == 0 and ldxs >= %u", hi_only ? F : 2 * F);
  DNS_REQUIRE(!code || ((uintptr_t)code & 15) == 0, "dns_feature_block_split: code must be 16-byte aligned");
  DNS_REQUIRE(!code || (z && gt_depth), "dns_feature_block_split: a code needs z and gt_depth");
  DNS_REQUIRE(n_ref <= 1 || (pts_per_frame > 0 && P % pts_per_frame == 0), "dns_feature_block_split: n_ref slabs need pts_per_frame dividing N S");
  const uint32_t qn = (hidden > C ? hidden : C) / 4u;
  DNS_REQUIRE(qn >= 1 && qn <= 16 && (qn & (qn - 1)) == 0, "dns_feature_block_split: max(hidden, C) / 4 must be a power of two <= 16");
  DNS_REQUIRE(P * qn < (1ull << 32), "dns_feature_block_split: too many points");
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(feature_block_split_kernel, dim3(grid_for(P * qn, 16384)), dim3(256), 0, st, fine, ld_fine, hidden, code, C, n_ref,
             pts_per_frame, z, gt_depth, (uint32_t)P, S, feat, ld_feat, reinterpret_cast<_Float16*>(xs_out), ldxs, xexp,
             (hi_only ? 1u : 0u) | (plain ? 2u : 0u), raw, qn);
  return check_launch("dns_feature_block_split");
}

extern "C" int dns_rgb_sigmoid(float* raw, uint32_t P, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(raw && ((uintptr_t)raw & 15) == 0, "dns_rgb_sigmoid: raw must be a 16-byte aligned [P,4] array");
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(rgb_sigmoid_kernel, dim3(grid_for(P)), dim3(256), 0, st, reinterpret_cast<float4*>(raw), P);
  return check_launch("dns_rgb_sigmoid");
}

extern "C" int dns_raw_bwd(const float* d_raw, const float* raw, uint32_t P, float* d_col, float* d_occ, uint32_t ld_occ,
                           int accumulate, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(d_raw && raw && d_col && d_occ && ld_occ >= 1, "dns_raw_bwd: NULL argument");
  DNS_REQUIRE((((uintptr_t)d_raw | (uintptr_t)raw | (uintptr_t)d_col) & 15) == 0, "dns_raw_bwd: [P,4] arrays must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(raw_bwd_kernel, dim3(grid_for(P)), dim3(256), 0, st, reinterpret_cast<const float4*>(d_raw),
             reinterpret_cast<const float4*>(raw), P, reinterpret_cast<float4*>(d_col), d_occ, ld_occ, accumulate ? 1u : 0u);
  return check_launch("dns_raw_bwd");
}

extern "C" int dns_lattice_points(const float* r6, const double* consts9, uint32_t n, const int32_t* order, uint32_t count,
                                  float* pts, void* stream) {
  if (n == 0) return DNS_OK;
  DNS_REQUIRE(r6 && consts9 && pts, "dns_lattice_points: NULL argument");
  DNS_REQUIRE((uint64_t)n * n * n < (1ull << 30), "dns_lattice_points: lattice too large");
  const uint32_t total = count ? count : n * n * n;
  DNS_REQUIRE(total <= n * n * n && (count == 0 || order), "dns_lattice_points: count needs an element list of at most n^3 entries");
  LatticeConsts c;
  for (int a = 0; a < 3; ++a) { c.vox[a] = consts9[a]; c.off[a] = consts9[3 + a]; c.mar[a] = consts9[6 + a]; }
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(lattice_points_kernel, dim3(grid_for((uint64_t)total)), dim3(256), 0, st, r6, c, n, order, total, pts);
  return check_launch("dns_lattice_points");
}

extern "C" int dns_track_mask(const float* gt_depth, const uint8_t* inside, uint32_t N, float min_depth, uint8_t* valid, void* stream) {
  if (N == 0) return DNS_OK;
  DNS_REQUIRE(gt_depth && inside && valid, "dns_track_mask: NULL argument");
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(track_mask_kernel, dim3((N + 255) / 256), dim3(256), 0, st, gt_depth, inside, N, min_depth, valid);
  return check_launch("dns_track_mask");
}

extern "C" int dns_keep_best(const float* loss, const float* quat, const float* trans, float* best_loss, float* best_cam, void* stream) {
  DNS_REQUIRE(loss && quat && trans && best_loss && best_cam, "dns_keep_best: NULL argument");
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(keep_best_kernel, dim3(1), dim3(64), 0, st, loss, quat, trans, best_loss, best_cam);
  return check_launch("dns_keep_best");
}

extern "C" int dns_force_half(float* t, uint32_t n, uint32_t idx, void* stream) {
  if (n == 0) return DNS_OK;
  DNS_REQUIRE(t, "dns_force_half: NULL argument");
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(force_half_kernel, dim3(1), dim3(64), 0, st, t, n, idx);
  return check_launch("dns_force_half");
}

extern "C" int dns_draw_finish(const int64_t* i1, const double* u, const double* count_f64, const int64_t* count_m1,
                               const int64_t* start_flat, const int64_t* sorted_flat, const float* depth, const float* label,
                               uint32_t n_frames, uint32_t n1, uint32_t n2, uint32_t HW, int64_t* pix, int64_t* labels,
                               uint32_t* dmax, void* stream) {
  if (n_frames == 0 || n1 + n2 == 0) return DNS_OK;
  DNS_REQUIRE(depth && label && pix && labels && dmax, "dns_draw_finish: NULL argument");
  DNS_REQUIRE(n1 == 0 || i1, "dns_draw_finish: n1 > 0 needs i1");
  DNS_REQUIRE(n2 == 0 || (u && count_f64 && count_m1 && start_flat && sorted_flat), "dns_draw_finish: n2 > 0 needs the class tables");
  hipStream_t st = (hipStream_t)stream;
  DNS_LAUNCH(draw_finish_kernel, dim3(n_frames), dim3(256), 0, st, i1, u, count_f64, count_m1, start_flat, sorted_flat, depth, label,
             n1, n2, HW, pix, labels, dmax);
  return check_launch("dns_draw_finish");
}

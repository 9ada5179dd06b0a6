// 2-D feature lookup of the feature branch: project every point into the reference frames and fetch the image code.
//
// Replaces the tensor-level body of feature_matching / feature_searching (reference utils/common.py:632-673): the
// reference first up-samples each [64, H/2, W/2] stem feature map to full resolution (F.interpolate, bilinear,
// align_corners=True: 627 MB written per call at Replica resolution), then takes the feature at the ROUNDED projected
// pixel.  Here the bilinear value at that integer full-resolution pixel is evaluated directly from the half-resolution
// map (channels-last, so the four taps of a pixel are four contiguous 256-byte rows): one wave64 per (reference frame,
// point), one lane per channel.  Forward only: the code carries no gradient (frozen stem; rounding blocks the pose path).
// Memory-bound gather: 4 * C * 4 bytes read per (frame, point), served from L2 / Infinity Cache.
#include "common.hpp"

namespace dns {

struct Mat3 {
  float k[9];
};

__global__ __launch_bounds__(256) void feature_gather_kernel(const float* __restrict__ pts, const float* __restrict__ w2c,
                                                             Mat3 K, const float* __restrict__ feat, uint32_t R, uint32_t P,
                                                             uint32_t C, int h, int w, int H, int W,
                                                             float* __restrict__ code, uint8_t* __restrict__ mask_out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t pair = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (uint64_t)R * P) return;
  const uint32_t r = (uint32_t)(pair / P), p = (uint32_t)(pair % P);
  const float x = pts[(size_t)p * 3], y = pts[(size_t)p * 3 + 1], z = pts[(size_t)p * 3 + 2];
  const float* m = w2c + 16 * r;
  // camera-frame point, y and z flipped (utils/common.py:650-652)
  const float cx_ = fmaf(m[3], 1.0f, fmaf(m[2], z, fmaf(m[1], y, m[0] * x)));
  const float cy_ = -fmaf(m[7], 1.0f, fmaf(m[6], z, fmaf(m[5], y, m[4] * x)));
  const float cz_ = -fmaf(m[11], 1.0f, fmaf(m[10], z, fmaf(m[9], y, m[8] * x)));
  const float q0 = fmaf(K.k[2], cz_, fmaf(K.k[1], cy_, K.k[0] * cx_));
  const float q1 = fmaf(K.k[5], cz_, fmaf(K.k[4], cy_, K.k[3] * cx_));
  const float q2 = fmaf(K.k[8], cz_, fmaf(K.k[7], cy_, K.k[6] * cx_));
  const float u = rintf(q0 / (q2 + 1e-5f));            // torch.round: half to even
  const float v = rintf(q1 / (q2 + 1e-5f));
  const bool ok = (u > 0.f) && (u < (float)(W - 1)) && (v > 0.f) && (v < (float)(H - 1)) && (cz_ > 0.f);
  if (lane == 0 && mask_out) mask_out[pair] = ok ? 1 : 0;
  float* out = code + pair * C;
  if (!ok) {
    for (uint32_t c = lane; c < C; c += 64) out[c] = 0.f;
    return;
  }
  // F.interpolate(..., align_corners=True) evaluated at the integer pixel (u, v)
  const float sx_scale = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const float sy_scale = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sx = __fmul_rn(sx_scale, u), sy = __fmul_rn(sy_scale, v);
  const int x0 = (int)sx, y0 = (int)sy;
  const float lx = sx - (float)x0, ly = sy - (float)y0;
  const int x1 = x0 + (x0 < w - 1 ? 1 : 0), y1 = y0 + (y0 < h - 1 ? 1 : 0);
  const float* f = feat + (size_t)r * h * w * C;
  const float* f00 = f + ((size_t)y0 * w + x0) * C;
  const float* f01 = f + ((size_t)y0 * w + x1) * C;
  const float* f10 = f + ((size_t)y1 * w + x0) * C;
  const float* f11 = f + ((size_t)y1 * w + x1) * C;
  for (uint32_t c = lane; c < C; c += 64)
    out[c] = (1.f - ly) * ((1.f - lx) * f00[c] + lx * f01[c]) + ly * ((1.f - lx) * f10[c] + lx * f11[c]);
}


// ---- the in-loop form of the branch (slams/mapping.py:532-557, slams/tracking.py:162-165 inside every optimise iteration) ----
// One launch for the K target frames of an iteration: reference rr = f R + r looks at the points of frame f = rr / R
// (pts [K, P_f, 3]).  Besides the code -- written at a row stride (straight into the Merge network's input row, behind the
// OneBlob columns) -- it writes the RELATIVE point pts - o_rr that Decoder.merge encodes (utils/common.py:675).
__global__ __launch_bounds__(256) void feature_gather_frames_kernel(const float* __restrict__ pts, const float* __restrict__ w2c,
                                                                    const float* __restrict__ origin, Mat3 K,
                                                                    const float* __restrict__ feat, uint32_t n_ref_total,
                                                                    uint32_t R, uint32_t Pf, uint32_t C, int h, int w, int H, int W,
                                                                    float* __restrict__ code, uint32_t ld_code,
                                                                    float* __restrict__ rel_out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t pair = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (uint64_t)n_ref_total * Pf) return;
  const uint32_t rr = (uint32_t)(pair / Pf), pl = (uint32_t)(pair % Pf);
  const size_t p = (size_t)(rr / R) * Pf + pl;
  const float x = pts[p * 3], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
  const float* m = w2c + 16 * rr;
  const float cx_ = fmaf(m[3], 1.0f, fmaf(m[2], z, fmaf(m[1], y, m[0] * x)));
  const float cy_ = -fmaf(m[7], 1.0f, fmaf(m[6], z, fmaf(m[5], y, m[4] * x)));
  const float cz_ = -fmaf(m[11], 1.0f, fmaf(m[10], z, fmaf(m[9], y, m[8] * x)));
  const float q0 = fmaf(K.k[2], cz_, fmaf(K.k[1], cy_, K.k[0] * cx_));
  const float q1 = fmaf(K.k[5], cz_, fmaf(K.k[4], cy_, K.k[3] * cx_));
  const float q2 = fmaf(K.k[8], cz_, fmaf(K.k[7], cy_, K.k[6] * cx_));
  const float u = rintf(q0 / (q2 + 1e-5f));
  const float v = rintf(q1 / (q2 + 1e-5f));
  const bool ok = (u > 0.f) && (u < (float)(W - 1)) && (v > 0.f) && (v < (float)(H - 1)) && (cz_ > 0.f);
  if (lane < 3u && rel_out) {
    const float c = lane == 0 ? x : (lane == 1 ? y : z);
    rel_out[pair * 3 + lane] = c - origin[rr * 3 + lane];
  }
  float* out = code + pair * ld_code;
  if (!ok) {
    for (uint32_t c = lane; c < C; c += 64) out[c] = 0.f;
    return;
  }
  const float sx_scale = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const float sy_scale = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sx = __fmul_rn(sx_scale, u), sy = __fmul_rn(sy_scale, v);
  const int x0 = (int)sx, y0 = (int)sy;
  const float lx = sx - (float)x0, ly = sy - (float)y0;
  const int x1 = x0 + (x0 < w - 1 ? 1 : 0), y1 = y0 + (y0 < h - 1 ? 1 : 0);
  const float* f = feat + (size_t)rr * h * w * C;
  const float* f00 = f + ((size_t)y0 * w + x0) * C;
  const float* f01 = f + ((size_t)y0 * w + x1) * C;
  const float* f10 = f + ((size_t)y1 * w + x0) * C;
  const float* f11 = f + ((size_t)y1 * w + x1) * C;
  for (uint32_t c = lane; c < C; c += 64)
    out[c] = (1.f - ly) * ((1.f - lx) * f00[c] + lx * f01[c]) + ly * ((1.f - lx) * f10[c] + lx * f11[c]);
}

// The four-channel form of both kernels above (C % 4 == 0 and 16-byte aligned rows: the stem maps have 64 channels): 16 lanes
// per (reference, point) pair, one float4 of channels per lane -- a wave works on FOUR pairs, so the projection (the same ~40
// vector instructions in every lane of a pair) is issued a quarter as often and every tap is one 16-byte load per lane.  The
// one-lane-per-channel kernels were issue-bound at 64 channels (193 us per cfg3 iteration for 786 432 pairs; the 805 MB of taps
// come out of L2 / MALL, 201 MB are written).  Same expressions per channel: same values.
template <bool FRAMES>
__global__ __launch_bounds__(256) void feature_gather4_kernel(const float* __restrict__ pts, const float* __restrict__ w2c,
                                                              const float* __restrict__ origin, Mat3 K,
                                                              const float* __restrict__ feat, uint32_t n_ref_total, uint32_t R,
                                                              uint32_t Pf, uint32_t C, int h, int w, int H, int W,
                                                              float* __restrict__ code, uint32_t ld_code,
                                                              float* __restrict__ rel_out, uint8_t* __restrict__ mask_out) {
  const uint32_t sub = threadIdx.x & 15u;
  const uint64_t pair = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (pair >= (uint64_t)n_ref_total * Pf) return;
  const uint32_t rr = (uint32_t)(pair / Pf), pl = (uint32_t)(pair % Pf);
  const size_t p = FRAMES ? (size_t)(rr / R) * Pf + pl : (size_t)pl;
  const float x = pts[p * 3], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
  const float* m = w2c + 16 * rr;
  const float cx_ = fmaf(m[3], 1.0f, fmaf(m[2], z, fmaf(m[1], y, m[0] * x)));
  const float cy_ = -fmaf(m[7], 1.0f, fmaf(m[6], z, fmaf(m[5], y, m[4] * x)));
  const float cz_ = -fmaf(m[11], 1.0f, fmaf(m[10], z, fmaf(m[9], y, m[8] * x)));
  const float q0 = fmaf(K.k[2], cz_, fmaf(K.k[1], cy_, K.k[0] * cx_));
  const float q1 = fmaf(K.k[5], cz_, fmaf(K.k[4], cy_, K.k[3] * cx_));
  const float q2 = fmaf(K.k[8], cz_, fmaf(K.k[7], cy_, K.k[6] * cx_));
  const float u = rintf(q0 / (q2 + 1e-5f));
  const float v = rintf(q1 / (q2 + 1e-5f));
  const bool ok = (u > 0.f) && (u < (float)(W - 1)) && (v > 0.f) && (v < (float)(H - 1)) && (cz_ > 0.f);
  if (FRAMES) {
    if (sub < 3u && rel_out) {
      const float c = sub == 0 ? x : (sub == 1 ? y : z);
      rel_out[pair * 3 + sub] = c - origin[rr * 3 + sub];
    }
  } else if (sub == 0 && mask_out) {
    mask_out[pair] = ok ? 1 : 0;
  }
  float4* out = reinterpret_cast<float4*>(code + pair * ld_code);
  const uint32_t C4 = C / 4u;
  if (!ok) {
    for (uint32_t c = sub; c < C4; c += 16) out[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float sx_scale = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const float sy_scale = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sx = __fmul_rn(sx_scale, u), sy = __fmul_rn(sy_scale, v);
  const int x0 = (int)sx, y0 = (int)sy;
  const float lx = sx - (float)x0, ly = sy - (float)y0;
  const int x1 = x0 + (x0 < w - 1 ? 1 : 0), y1 = y0 + (y0 < h - 1 ? 1 : 0);
  const float* f = feat + (size_t)rr * h * w * C;
  const float4* f00 = reinterpret_cast<const float4*>(f + ((size_t)y0 * w + x0) * C);
  const float4* f01 = reinterpret_cast<const float4*>(f + ((size_t)y0 * w + x1) * C);
  const float4* f10 = reinterpret_cast<const float4*>(f + ((size_t)y1 * w + x0) * C);
  const float4* f11 = reinterpret_cast<const float4*>(f + ((size_t)y1 * w + x1) * C);
  for (uint32_t c = sub; c < C4; c += 16) {
    const float4 a = f00[c], b = f01[c], d = f10[c], e = f11[c];
    float4 o;
    o.x = (1.f - ly) * ((1.f - lx) * a.x + lx * b.x) + ly * ((1.f - lx) * d.x + lx * e.x);
    o.y = (1.f - ly) * ((1.f - lx) * a.y + lx * b.y) + ly * ((1.f - lx) * d.y + lx * e.y);
    o.z = (1.f - ly) * ((1.f - lx) * a.z + lx * b.z) + ly * ((1.f - lx) * d.z + lx * e.z);
    o.w = (1.f - ly) * ((1.f - lx) * a.w + lx * b.w) + ly * ((1.f - lx) * d.w + lx * e.w);
    out[c] = o;
  }
}
static bool gather4_ok(const float* feat, const float* code, uint32_t C, uint32_t ld_code) {
  return C % 4u == 0 && ld_code % 4u == 0 && (((uintptr_t)feat | (uintptr_t)code) & 15) == 0 && getenv("DNS_FEATURE_GATHER_LANES") == nullptr;
}

// World -> camera matrices of the K x R reference views of an iteration (slams/mapping.py:534-547): reference rr takes the pose
// of target frame src[rr] >= 0 AS IT STANDS IN THE OPTIMISER (quaternion -> rotation as get_rotation_from_quad,
// utils/common.py:406-429: two_s = 2 / |q|^2, no normalisation; detached: the code carries no pose gradient through the
// projection) or, src[rr] < 0, the stored keyframe pose fixed_c2w[rr].  w2c = inverse (torch.inverse in the reference: here the
// affine inverse with the 3 x 3 block inverted by cofactors in float64), origin = the pose's translation (refer_o, :674).
__global__ void refer_poses_kernel(const float* __restrict__ quat, const float* __restrict__ trans, const int32_t* __restrict__ src,
                                   const float* __restrict__ fixed_c2w, uint32_t n, float* __restrict__ w2c, float* __restrict__ origin) {
  const uint32_t rr = blockIdx.x * blockDim.x + threadIdx.x;
  if (rr >= n) return;
  double A[9], t[3];
  const int k = src[rr];
  if (k >= 0) {
    const float qw = quat[4 * k], qx = quat[4 * k + 1], qy = quat[4 * k + 2], qz = quat[4 * k + 3];
    const float two_s = 2.0f / (qw * qw + qx * qx + qy * qy + qz * qz);
    const float Rm[9] = {1.0f - two_s * (qy * qy + qz * qz), two_s * (qx * qy - qz * qw), two_s * (qx * qz + qy * qw),
                         two_s * (qx * qy + qz * qw), 1.0f - two_s * (qx * qx + qz * qz), two_s * (qy * qz - qx * qw),
                         two_s * (qx * qz - qy * qw), two_s * (qy * qz + qx * qw), 1.0f - two_s * (qx * qx + qy * qy)};
    for (int i = 0; i < 9; ++i) A[i] = (double)Rm[i];
    for (int i = 0; i < 3; ++i) t[i] = (double)trans[3 * k + i];
  } else {
    const float* c = fixed_c2w + 16 * rr;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) A[3 * i + j] = (double)c[4 * i + j];
      t[i] = (double)c[4 * i + 3];
    }
  }
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  const double id = 1.0 / det;
  const double I[9] = {c00 * id, (A[2] * A[7] - A[1] * A[8]) * id, (A[1] * A[5] - A[2] * A[4]) * id,
                       c01 * id, (A[0] * A[8] - A[2] * A[6]) * id, (A[2] * A[3] - A[0] * A[5]) * id,
                       c02 * id, (A[1] * A[6] - A[0] * A[7]) * id, (A[0] * A[4] - A[1] * A[3]) * id};
  float* o = w2c + 16 * rr;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) o[4 * i + j] = (float)I[3 * i + j];
    o[4 * i + 3] = (float)(-(I[3 * i] * t[0] + I[3 * i + 1] * t[1] + I[3 * i + 2] * t[2]));
  }
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
  for (int i = 0; i < 3; ++i) origin[3 * rr + i] = (float)t[i];
}

// Backward of (mean over the R references, truncation mask): d_lat[f, r, pl, c] = d_code[p, c] * trunc(p) / R for every r
// (models/decoder.py:76, slams/mapping.py:553-557); the source columns are CLEARED afterwards (the colour / logit networks add
// their input gradients into them every iteration).  One thread per (point, quad of channels).
__global__ __launch_bounds__(256) void merge_dy_kernel(float* __restrict__ d_code, uint32_t ld_dcode, uint32_t C, uint32_t R,
                                                       uint32_t Pf, const float* __restrict__ z, const float* __restrict__ gt_depth,
                                                       uint32_t P, uint32_t S, float* __restrict__ d_lat) {
  const uint32_t qc = C / 4u, total = P * qc;
  const float inv = 1.0f / (float)R;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t p = i / qc, q = i - p * qc;
    float4* src = reinterpret_cast<float4*>(d_code + (size_t)p * ld_dcode + 4u * q);
    float4 g = *src;
    const float d = gt_depth[p / S], zz = z[p];
    const float front = zz < d * 0.95f ? 1.f : 0.f, back = zz > d * 1.05f ? 1.f : 0.f, pos = d > 0.f ? 1.f : 0.f;
    const float t = (1.f - front) * (1.f - back) * pos * inv;
    g.x *= t; g.y *= t; g.z *= t; g.w *= t;
    *src = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t fr = p / Pf, pl = p - fr * Pf;
    float* dst = d_lat + ((size_t)fr * R * Pf + pl) * C + 4u * q;
    for (uint32_t r = 0; r < R; ++r) *reinterpret_cast<float4*>(dst + (size_t)r * Pf * C) = g;
  }
}

// d_pts[p, :] += sum_r d_rel[f, r, pl, :]   (refer_p = pts - refer_o, utils/common.py:675: the points' share of Merge's gradient)
__global__ __launch_bounds__(256) void add_ref_sum_kernel(const float* __restrict__ d_rel, uint32_t R, uint32_t Pf, uint32_t P,
                                                          float* __restrict__ d_pts) {
  const uint32_t total = P * 3u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t p = i / 3u, a = i - 3u * p;
    const uint32_t fr = p / Pf, pl = p - fr * Pf;
    const float* src = d_rel + ((size_t)fr * R * Pf + pl) * 3u + a;
    float s = 0.f;
    for (uint32_t r = 0; r < R; ++r) s += src[(size_t)r * Pf * 3u];
    d_pts[i] += s;
  }
}

}  // namespace dns

using namespace dns;

extern "C" int dns_feature_gather_frames(const float* pts, const float* w2c, const float* origin, const float* K, const float* feat,
                                         uint32_t n_frames, uint32_t R, uint32_t pts_per_frame, uint32_t C, int h, int w, int H, int W,
                                         float* code, uint32_t ld_code, float* rel_out, void* stream) {
  const uint64_t pairs = (uint64_t)n_frames * R * pts_per_frame;
  if (pairs == 0) return DNS_OK;
  DNS_REQUIRE(pts && w2c && K && feat && code, "dns_feature_gather_frames: NULL argument");
  DNS_REQUIRE(!rel_out || origin, "dns_feature_gather_frames: rel_out needs the reference origins");
  DNS_REQUIRE(C >= 1 && ld_code >= C && h >= 1 && w >= 1 && H >= 1 && W >= 1, "dns_feature_gather_frames: bad dimensions");
  DNS_REQUIRE((pairs + 3) / 4 < (1ull << 31), "dns_feature_gather_frames: too many points");
  Mat3 Km;
  for (int i = 0; i < 9; ++i) Km.k[i] = K[i];
  if (gather4_ok(feat, code, C, ld_code))
    DNS_LAUNCH(feature_gather4_kernel<true>, dim3((uint32_t)((pairs + 15) / 16)), dim3(256), 0, (hipStream_t)stream, pts, w2c, origin,
               Km, feat, n_frames * R, R, pts_per_frame, C, h, w, H, W, code, ld_code, rel_out, (uint8_t*)nullptr);
  else
    DNS_LAUNCH(feature_gather_frames_kernel, dim3((uint32_t)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pts, w2c, origin, Km,
               feat, n_frames * R, R, pts_per_frame, C, h, w, H, W, code, ld_code, rel_out);
  return check_launch("dns_feature_gather_frames");
}

extern "C" int dns_refer_poses(const float* quat, const float* trans, const int32_t* src, const float* fixed_c2w, uint32_t n,
                               float* w2c, float* origin, void* stream) {
  if (n == 0) return DNS_OK;
  DNS_REQUIRE(src && fixed_c2w && w2c && origin && quat && trans, "dns_refer_poses: NULL argument");
  DNS_LAUNCH(refer_poses_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, quat, trans, src, fixed_c2w, n, w2c, origin);
  return check_launch("dns_refer_poses");
}

extern "C" int dns_merge_dy(float* d_code, uint32_t ld_dcode, uint32_t C, uint32_t R, uint32_t pts_per_frame, const float* z,
                            const float* gt_depth, uint32_t N, uint32_t S, float* d_lat, void* stream) {
  const uint64_t P = (uint64_t)N * S;
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(d_code && z && gt_depth && d_lat, "dns_merge_dy: NULL argument");
  DNS_REQUIRE(C % 4 == 0 && C >= 4 && ld_dcode % 4 == 0 && ld_dcode >= C && R >= 1 && pts_per_frame > 0 && P % pts_per_frame == 0,
              "dns_merge_dy: C %u / ld %u / R %u / pts_per_frame %u", C, ld_dcode, R, pts_per_frame);
  DNS_REQUIRE((((uintptr_t)d_code | (uintptr_t)d_lat) & 15) == 0, "dns_merge_dy: 16-byte alignment");
  DNS_REQUIRE(P * (C / 4) < (1ull << 32), "dns_merge_dy: too many points");
  const uint64_t n = P * (C / 4);
  const uint32_t blocks = (uint32_t)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  DNS_LAUNCH(merge_dy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_code, ld_dcode, C, R, pts_per_frame, z, gt_depth,
             (uint32_t)P, S, d_lat);
  return check_launch("dns_merge_dy");
}

extern "C" int dns_add_ref_sum(const float* d_rel, uint32_t R, uint32_t pts_per_frame, uint32_t P, float* d_pts, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(d_rel && d_pts && R >= 1 && pts_per_frame > 0 && P % pts_per_frame == 0, "dns_add_ref_sum: bad argument");
  DNS_REQUIRE((uint64_t)P * 3 < (1ull << 32), "dns_add_ref_sum: too many points");
  const uint64_t n = (uint64_t)P * 3;
  const uint32_t blocks = (uint32_t)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  DNS_LAUNCH(add_ref_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_rel, R, pts_per_frame, P, d_pts);
  return check_launch("dns_add_ref_sum");
}

extern "C" int dns_feature_gather(const float* pts, const float* w2c, const float* K, const float* feat, uint32_t R,
                                  uint32_t P, uint32_t C, int h, int w, int H, int W, float* code, uint8_t* mask,
                                  void* stream) {
  if (R == 0 || P == 0) return DNS_OK;
  DNS_REQUIRE(pts && w2c && K && feat && code, "dns_feature_gather: NULL argument");
  DNS_REQUIRE(C >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1, "dns_feature_gather: bad dimensions");
  Mat3 Km;
  for (int i = 0; i < 9; ++i) Km.k[i] = K[i];
  const uint64_t pairs = (uint64_t)R * P;
  DNS_REQUIRE((pairs + 3) / 4 < (1ull << 31), "dns_feature_gather: too many points");
  if (gather4_ok(feat, code, C, C))
    DNS_LAUNCH(feature_gather4_kernel<false>, dim3((uint32_t)((pairs + 15) / 16)), dim3(256), 0, (hipStream_t)stream, pts, w2c,
               (const float*)nullptr, Km, feat, R, R, P, C, h, w, H, W, code, C, (float*)nullptr, mask);
  else
    DNS_LAUNCH(feature_gather_kernel, dim3((uint32_t)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pts, w2c, Km,
               feat, R, P, C, h, w, H, W, code, mask);
  return check_launch("dns_feature_gather");
}

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

}  // namespace dns

using namespace dns;

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
  DNS_LAUNCH(feature_gather_kernel, dim3((uint32_t)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pts, w2c, Km,
                     feat, R, P, C, h, w, H, W, code, mask);
  return check_launch("dns_feature_gather");
}

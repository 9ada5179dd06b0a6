// Device helpers of the ray generation / depth-guided sampling kernels (sample.hip) -- shared with the fused tracker iteration
// (track_fused.inc).  Every product or sum that decides a value bit for bit is spelled __fmul_rn / __fadd_rn / __dmul_rn, so
// the result does not depend on the translation unit's -ffp-contract setting.
#pragma once
#include "common.hpp"

namespace dns {

struct Cam {
  float fx, fy, cx, cy;
};
struct BoundD {
  double b[6];  // b0x,b1x,b0y,b1y,b0z,b1z
};

__device__ __forceinline__ uint32_t f2key(float f) {
  if (f != f) return 0xFFC00000u;  // every NaN sorts after +inf (torch.sort puts NaN last), below the pad key
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(b);
}

__device__ __forceinline__ double max_nan(double a, double b) { return (a != a || b != b) ? (a + b) : (a > b ? a : b); }
__device__ __forceinline__ double min_nan(double a, double b) { return (a != a || b != b) ? (a + b) : (a < b ? a : b); }

__device__ __forceinline__ void quat_to_R(const float* __restrict__ q, float R[9]) {
  const float qr = q[0], qi = q[1], qj = q[2], qk = q[3];
  const float nrm = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(qr, qr), __fmul_rn(qi, qi)), __fmul_rn(qj, qj)), __fmul_rn(qk, qk));
  const float two_s = 2.0f / nrm;
  const float jj = __fmul_rn(qj, qj), kk = __fmul_rn(qk, qk), ii = __fmul_rn(qi, qi);
  const float ij = __fmul_rn(qi, qj), kr = __fmul_rn(qk, qr), ik = __fmul_rn(qi, qk), jr = __fmul_rn(qj, qr);
  const float jk = __fmul_rn(qj, qk), ir = __fmul_rn(qi, qr);
  R[0] = __fsub_rn(1.0f, __fmul_rn(two_s, __fadd_rn(jj, kk)));
  R[1] = __fmul_rn(two_s, __fsub_rn(ij, kr));
  R[2] = __fmul_rn(two_s, __fadd_rn(ik, jr));
  R[3] = __fmul_rn(two_s, __fadd_rn(ij, kr));
  R[4] = __fsub_rn(1.0f, __fmul_rn(two_s, __fadd_rn(ii, kk)));
  R[5] = __fmul_rn(two_s, __fsub_rn(jk, ir));
  R[6] = __fmul_rn(two_s, __fsub_rn(ik, jr));
  R[7] = __fmul_rn(two_s, __fadd_rn(jk, ir));
  R[8] = __fsub_rn(1.0f, __fmul_rn(two_s, __fadd_rn(ii, jj)));
}

__device__ __forceinline__ void pixel_dir(int64_t q, int H0, int W0, int wwin, Cam cam, int& row, int& col, float dir[3]) {
  row = H0 + (int)(q / wwin);
  col = W0 + (int)(q % wwin);
  dir[0] = __fdiv_rn(__fsub_rn((float)col, cam.cx), cam.fx);
  dir[1] = -__fdiv_rn(__fsub_rn((float)row, cam.cy), cam.fy);
  dir[2] = -1.0f;
}

// z values of one ray (utils/common.py:561-599) as sorted order-preserving keys; lane owns elements e*64+lane.
// `far` is far_bb AFTER the +0.01 (fp64).  Surface part fp32, uniform part fp64 then rounded, as the reference.
template <int E>
__device__ __forceinline__ void sample_and_sort(float gd, double far, float dmax, const float* __restrict__ t_uniform,
                                                const float* __restrict__ t_surf, const float* __restrict__ t_zero,
                                                int nu, int ns, uint32_t lane, uint32_t (&key)[E]) {
  const int S = nu + ns;
  const double hi = (double)__fmul_rn(dmax, 1.2f);
  double farc = far;                      // torch.clamp(far, 0, hi): NaN propagates
  if (farc == farc) {
    farc = farc < 0.0 ? 0.0 : farc;
    farc = farc > hi ? hi : farc;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int s = e * 64 + (int)lane;
    float zv;
    if (s < nu) {
      const float tv = t_uniform[s];
      const float near = __fmul_rn(gd, 0.001f);
      const float a = __fmul_rn(near, __fsub_rn(1.0f, tv));
      zv = (float)__dadd_rn((double)a, __dmul_rn(farc, (double)tv));
    } else if (s < S) {
      const int k = s - nu;
      if (gd > 0.f) {
        const float t = t_surf[k];
        zv = __fadd_rn(__fmul_rn(__fmul_rn(0.95f, gd), __fsub_rn(1.0f, t)), __fmul_rn(__fmul_rn(1.05f, gd), t));
      } else {
        const float t = t_zero[k];
        zv = __fadd_rn(__fmul_rn(0.001f, __fsub_rn(1.0f, t)), __fmul_rn(dmax, t));
      }
    } else {
      zv = 0.f;
    }
    key[e] = (s < S) ? f2key(zv) : 0xFFFFFFFFu;
  }
  // bitonic sort of 64*E keys; element id g = e*64 + lane
  constexpr int NTOT = 64 * E;
#pragma unroll
  for (int k = 2; k <= NTOT; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      if (j >= 64) {
        const int je = j >> 6;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int pe = e ^ je;
          if (pe > e) {
            const int g = e * 64 + (int)lane;
            const bool up = (g & k) == 0;
            const uint32_t lo = min(key[e], key[pe]), hi2 = max(key[e], key[pe]);
            key[e] = up ? lo : hi2;
            key[pe] = up ? hi2 : lo;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int g = e * 64 + (int)lane;
          const uint32_t other = (uint32_t)__shfl_xor((int)key[e], j);
          const bool up = (g & k) == 0;
          const bool lower = (g & j) == 0;
          key[e] = (lower == up) ? min(key[e], other) : max(key[e], other);
        }
      }
    }
  }
}

}  // namespace dns

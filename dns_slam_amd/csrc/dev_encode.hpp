// Device helpers of the point-encoding kernels (encode.hip): hash / dense row index, the OneBlob kernel integrals and the fp64
// normalisation of a point -- shared with the fused tracker iteration (track_fused.inc).  encode.hip is compiled with
// -ffp-contract=off (value-exact arithmetic shared with oracle/tcnn_ref.py); the functions whose results a contraction would change
// carry the same setting as a function-level pragma, so they give the same bits in every translation unit.
#pragma once
#include "common.hpp"

namespace dns {

struct Bound6 {
  double b0[3];
  double inv_unused[3];
  double b1[3];
};

__device__ __forceinline__ uint32_t grid_row(uint32_t gx, uint32_t gy, uint32_t gz, uint32_t res,
                                             uint32_t size, uint32_t hashed) {
  uint32_t idx;
  if (hashed) {
    idx = gx ^ (gy * 2654435761u) ^ (gz * 805459861u);
    // hashed levels are exactly 2^T rows
    return idx & (size - 1u);
  }
  idx = gx + gy * res + gz * res * res;
  if (idx >= size) idx %= size;
  return idx;
}

__device__ __forceinline__ float quartic_cdf(float v, float n) {
#pragma clang fp contract(off)
  const float u = v * n;
  const float u2 = u * u;
  const float u4 = u2 * u2;
  const float r = (15.0f / 16.0f) * u * (1.0f - (2.0f / 3.0f) * u2 + (1.0f / 5.0f) * u4) + 0.5f;
  return fminf(fmaxf(r, 0.0f), 1.0f);
}

__device__ __forceinline__ float quartic_pdf(float v, float n) {
#pragma clang fp contract(off)
  const float u = v * n;
  const float u2 = u * u;
  if (u2 > 1.0f) return 0.0f;
  const float t = 1.0f - u2;
  return (15.0f / 16.0f) * n * t * t;
}

// G(b) of tcnn's kernel_one_blob at bin edge b (PDF: its derivative with respect to the distance): the kernel centred at x and
// its two periodic images; edge n_bins is edge 0 one period on (+1 for the cdf).  Expression order as in the full loops.
template <bool PDF>
__device__ __forceinline__ float oneblob_edge(uint32_t b, uint32_t n_bins, float n, float xa) {
#pragma clang fp contract(off)
  const uint32_t bb = b < n_bins ? b : 0u;
  const float d = (float)bb / n - xa;
  float g = PDF ? quartic_pdf(d, n) + quartic_pdf(d - 1.0f, n) + quartic_pdf(d + 1.0f, n)
                : quartic_cdf(d, n) + quartic_cdf(d - 1.0f, n) + quartic_cdf(d + 1.0f, n);
  if (!PDF && b >= n_bins) g += 1.0f;
  return g;
}

// The quartic kernel has radius 1/n: away from x (and its images x -+ 1) the cdf terms are EXACTLY 0 or 1 (clamped) and the
// pdf terms exactly 0, so a bin's value G(b+1) - G(b) is exactly zero unless one of its edges lies within one bin of
// c = x n + s n, s in {-1, 0, 1}.  Only the five bins around floor(c) are evaluated (two bins of margin against rounding),
// in ascending order, with the same expressions as the loop over all n + 1 edges: identical results for finite inputs
// from 18 + 6 (the last bin, see below) instead of 51 kernel evaluations per coordinate.  fn(bin, G(bin + 1) - G(bin)) is called for every bin that may be
// non-zero.  (An Inf / NaN in a far bin's upstream gradient no longer turns 0 * Inf into NaN in the backward.)
template <bool PDF, typename F>
__device__ __forceinline__ void oneblob_windows(uint32_t n_bins, float n, float xa, F fn) {
  const float xn = xa * n;
#pragma unroll
  for (int w = -1; w <= 1; ++w) {
    const float c = xn + (float)w * n;
    if (!(c > -3.0f && c < n + 3.0f)) continue;          // also skips NaN
    const int k = (int)floorf(c);
    const int j0 = max(k - 2, 0), j1 = min(k + 2, (int)n_bins - 2);     // the last bin: below
    if (j0 > j1) continue;
    float left = oneblob_edge<PDF>((uint32_t)j0, n_bins, n, xa);
    for (int j = j0; j <= j1; ++j) {
      const float right = oneblob_edge<PDF>((uint32_t)j + 1u, n_bins, n, xa);
      fn((uint32_t)j, right - left);
      left = right;
    }
  }
  // The LAST bin is always evaluated: its right edge is DEFINED as edge 0 (+1 for the cdf) -- the wrap of kernel_one_blob --,
  // so it is non-zero whenever edge 0 lies in ANY image's support (e.g. x = -0.96: edge 0 is inside the image at x + 1 and
  // bin n-1 sees it although no image is near edge n), and for the cdf even with both edges saturated (x = 3: G = 0
  // everywhere, bin n-1 = 1).
  fn(n_bins - 1u, oneblob_edge<PDF>(n_bins, n_bins, n, xa) - oneblob_edge<PDF>(n_bins - 1u, n_bins, n, xa));
}

__device__ __forceinline__ void load_point(const float* __restrict__ in, const Bound6& bd, bool normalise,
                                           uint32_t p, float x[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float v = in[(size_t)p * 3 + a];
    if (normalise) v = (float)(((double)v - bd.b0[a]) / (bd.b1[a] - bd.b0[a]));
    x[a] = v;
  }
}

}  // namespace dns

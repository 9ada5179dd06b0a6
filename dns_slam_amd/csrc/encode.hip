// Point encoding kernels: fp64 normalisation + OneBlob + multi-resolution hash grid.
//
// Replaces tcnn OneBlob / HashGrid behind Pos_Encoding.forward (reference models/decoder.py:45-48,
// models/pos_encoding.py:31-46,61-71) and the normalisation of slams/mapping.py:608.
//
// Layout: one lane = one point; all 16 levels x 8 corners of a point are issued by the same lane, so
// 128 independent 8-byte gathers are in flight per lane (the table is L2 / Infinity-Cache resident:
// 6.8 MB at T=2^16, 58 MB at T=2^20) and the lane's 32 grid features leave as 16-byte stores.
// Bound: memory system (random 8-B gathers, 1024 B per point), not MFMA.
//
// Arithmetic contract shared with oracle/tcnn_ref.py: pos = x*scale + 0.5 is two IEEE roundings
// (__fmul_rn/__fadd_rn, never contracted), the cell is (uint32)(int)floorf(pos); hash primes
// {1, 2654435761, 805459861}; index % level size.  Those make the table rows bit-exact.
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
  const float u = v * n;
  const float u2 = u * u;
  const float u4 = u2 * u2;
  const float r = (15.0f / 16.0f) * u * (1.0f - (2.0f / 3.0f) * u2 + (1.0f / 5.0f) * u4) + 0.5f;
  return fminf(fmaxf(r, 0.0f), 1.0f);
}

__device__ __forceinline__ float quartic_pdf(float v, float n) {
  const float u = v * n;
  const float u2 = u * u;
  if (u2 > 1.0f) return 0.0f;
  const float t = 1.0f - u2;
  return (15.0f / 16.0f) * n * t * t;
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

__global__ __launch_bounds__(256) void encode_fwd_kernel(const float* __restrict__ in, Bound6 bd, int normalise,
                                                         uint32_t P, uint32_t n_bins,
                                                         const float2* __restrict__ table, GridLevels lv,
                                                         float* __restrict__ x_out, float* __restrict__ pe_out,
                                                         uint32_t ld_pe, float* __restrict__ grid_out,
                                                         uint32_t ld_grid) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float x[3];
  load_point(in, bd, normalise != 0, p, x);
  if (x_out) {
    x_out[(size_t)p * 3 + 0] = x[0];
    x_out[(size_t)p * 3 + 1] = x[1];
    x_out[(size_t)p * 3 + 2] = x[2];
  }
  if (pe_out) {
    const float n = (float)n_bins;
    float* row = pe_out + (size_t)p * ld_pe;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float xa = x[a];
      float first = 0.f, left = 0.f;
      for (uint32_t b = 0; b <= n_bins; ++b) {
        float g;
        if (b < n_bins) {
          const float d = (float)b / n - xa;
          g = quartic_cdf(d, n) + quartic_cdf(d - 1.0f, n) + quartic_cdf(d + 1.0f, n);
          if (b == 0) first = g;
        } else {
          g = first + 1.0f;  // right edge of the last bin wraps (tcnn kernel_one_blob)
        }
        if (b > 0) row[a * n_bins + b - 1] = g - left;
        left = g;
      }
    }
  }
  if (grid_out) {
    float* row = grid_out + (size_t)p * ld_grid;
#pragma unroll 4
    for (uint32_t l = 0; l < lv.n_levels; ++l) {
      const float s = lv.scale[l];
      const uint32_t res = lv.resolution[l], size = lv.size[l], hashed = lv.hashed[l];
      const float2* __restrict__ t = table + lv.offset[l];
      float f[3];
      uint32_t g[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float pos = __fadd_rn(__fmul_rn(x[a], s), 0.5f);
        const float fl = floorf(pos);
        g[a] = (uint32_t)(int)fl;
        f[a] = pos - fl;
      }
      float2 v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c)
        v[c] = t[grid_row(g[0] + (c & 1), g[1] + ((c >> 1) & 1), g[2] + ((c >> 2) & 1), res, size, hashed)];
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float w = ((c & 1) ? f[0] : 1.0f - f[0]) * ((c & 2) ? f[1] : 1.0f - f[1]) * ((c & 4) ? f[2] : 1.0f - f[2]);
        a0 += w * v[c].x;
        a1 += w * v[c].y;
      }
      row[2 * l] = a0;
      row[2 * l + 1] = a1;
    }
  }
}

__global__ __launch_bounds__(256) void encode_bwd_kernel(const float* __restrict__ xin, Bound6 bd, int scale_by_bound,
                                                         uint32_t P, uint32_t n_bins,
                                                         const float2* __restrict__ table, GridLevels lv,
                                                         const float* __restrict__ d_pe, uint32_t ld_dpe,
                                                         const float* __restrict__ d_grid, uint32_t ld_dgrid,
                                                         float* __restrict__ d_table, float* __restrict__ d_x) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float x[3];
  x[0] = xin[(size_t)p * 3 + 0];
  x[1] = xin[(size_t)p * 3 + 1];
  x[2] = xin[(size_t)p * 3 + 2];
  float dx[3] = {0.f, 0.f, 0.f};
  if (d_pe && d_x) {
    const float n = (float)n_bins;
    const float* row = d_pe + (size_t)p * ld_dpe;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float xa = x[a];
      float first = 0.f, left = 0.f, acc = 0.f;
      for (uint32_t b = 0; b <= n_bins; ++b) {
        float g;
        if (b < n_bins) {
          const float d = (float)b / n - xa;
          g = quartic_pdf(d, n) + quartic_pdf(d - 1.0f, n) + quartic_pdf(d + 1.0f, n);
          if (b == 0) first = g;
        } else {
          g = first;
        }
        if (b > 0) acc -= row[a * n_bins + b - 1] * (g - left);  // d out_b / dx = -(g(b+1) - g(b))
        left = g;
      }
      dx[a] += acc;
    }
  }
  if (d_grid) {
    const float* row = d_grid + (size_t)p * ld_dgrid;
#pragma unroll 2
    for (uint32_t l = 0; l < lv.n_levels; ++l) {
      const float g0 = row[2 * l], g1 = row[2 * l + 1];
      const float s = lv.scale[l];
      const uint32_t res = lv.resolution[l], size = lv.size[l], hashed = lv.hashed[l];
      const uint32_t off = lv.offset[l];
      float f[3];
      uint32_t g[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float pos = __fadd_rn(__fmul_rn(x[a], s), 0.5f);
        const float fl = floorf(pos);
        g[a] = (uint32_t)(int)fl;
        f[a] = pos - fl;
      }
      uint32_t r[8];
#pragma unroll
      for (int c = 0; c < 8; ++c)
        r[c] = off + grid_row(g[0] + (c & 1), g[1] + ((c >> 1) & 1), g[2] + ((c >> 2) & 1), res, size, hashed);
      if (d_table && (g0 != 0.f || g1 != 0.f)) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float w = ((c & 1) ? f[0] : 1.0f - f[0]) * ((c & 2) ? f[1] : 1.0f - f[1]) * ((c & 4) ? f[2] : 1.0f - f[2]);
          atomicAdd(d_table + 2 * (size_t)r[c], w * g0);
          atomicAdd(d_table + 2 * (size_t)r[c] + 1, w * g1);
        }
      }
      if (d_x) {
        float dot[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float2 v = table[r[c]];
          dot[c] = v.x * g0 + v.y * g1;
        }
        const float wx0 = 1.0f - f[0], wx1 = f[0], wy0 = 1.0f - f[1], wy1 = f[1], wz0 = 1.0f - f[2], wz1 = f[2];
        // d/dx: pairs (c, c|1);  d/dy: (c, c|2);  d/dz: (c, c|4)
        const float ddx = wy0 * wz0 * (dot[1] - dot[0]) + wy1 * wz0 * (dot[3] - dot[2]) +
                          wy0 * wz1 * (dot[5] - dot[4]) + wy1 * wz1 * (dot[7] - dot[6]);
        const float ddy = wx0 * wz0 * (dot[2] - dot[0]) + wx1 * wz0 * (dot[3] - dot[1]) +
                          wx0 * wz1 * (dot[6] - dot[4]) + wx1 * wz1 * (dot[7] - dot[5]);
        const float ddz = wx0 * wy0 * (dot[4] - dot[0]) + wx1 * wy0 * (dot[5] - dot[1]) +
                          wx0 * wy1 * (dot[6] - dot[2]) + wx1 * wy1 * (dot[7] - dot[3]);
        dx[0] += s * ddx;
        dx[1] += s * ddy;
        dx[2] += s * ddz;
      }
    }
  }
  if (d_x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float v = dx[a];
      if (scale_by_bound) v = (float)((double)v / (bd.b1[a] - bd.b0[a]));
      d_x[(size_t)p * 3 + a] = v;
    }
  }
}

__global__ __launch_bounds__(256) void hashgrid_indices_kernel(const float* __restrict__ xin, uint32_t P, GridLevels lv,
                                                               uint32_t* __restrict__ rows) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float x[3] = {xin[(size_t)p * 3], xin[(size_t)p * 3 + 1], xin[(size_t)p * 3 + 2]};
  for (uint32_t l = 0; l < lv.n_levels; ++l) {
    uint32_t g[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) g[a] = (uint32_t)(int)floorf(__fadd_rn(__fmul_rn(x[a], lv.scale[l]), 0.5f));
#pragma unroll
    for (int c = 0; c < 8; ++c)
      rows[((size_t)p * lv.n_levels + l) * 8 + c] =
          lv.offset[l] + grid_row(g[0] + (c & 1), g[1] + ((c >> 1) & 1), g[2] + ((c >> 2) & 1), lv.resolution[l],
                                  lv.size[l], lv.hashed[l]);
  }
}

static Bound6 make_bound(const double* bound) {
  Bound6 b;
  for (int a = 0; a < 3; ++a) {
    b.b0[a] = bound ? bound[2 * a] : 0.0;
    b.b1[a] = bound ? bound[2 * a + 1] : 1.0;
    b.inv_unused[a] = 0.0;
  }
  return b;
}

}  // namespace dns

using namespace dns;

extern "C" int dns_encode_fwd(const float* in, const double* bound, uint32_t P, uint32_t n_bins, const float* table,
                              const DnsGridMeta* meta, float* x_out, float* pe_out, uint32_t ld_pe, float* grid_out,
                              uint32_t ld_grid, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(in != nullptr, "dns_encode_fwd: in is NULL");
  if (pe_out) DNS_REQUIRE(n_bins >= 1 && n_bins <= 64 && ld_pe >= 3 * n_bins, "dns_encode_fwd: n_bins %u / ld_pe %u", n_bins, ld_pe);
  GridLevels lv = {};
  if (grid_out) {
    DNS_REQUIRE(meta && table, "dns_encode_fwd: grid requested without table/meta");
    DNS_REQUIRE(meta->n_features == 2, "dns_encode_fwd: n_features must be 2");
    DNS_REQUIRE(ld_grid >= 2 * meta->n_levels, "dns_encode_fwd: ld_grid %u too small", ld_grid);
    lv = to_levels(meta);
  }
  const uint32_t blocks = (P + 255) / 256;
  hipLaunchKernelGGL(encode_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, make_bound(bound),
                     bound ? 1 : 0, P, n_bins, (const float2*)table, lv, x_out, pe_out, ld_pe, grid_out, ld_grid);
  return check_launch("dns_encode_fwd");
}

extern "C" int dns_encode_bwd(const float* x, const double* bound, uint32_t P, uint32_t n_bins, const float* table,
                              const DnsGridMeta* meta, const float* d_pe, uint32_t ld_dpe, const float* d_grid,
                              uint32_t ld_dgrid, float* d_table, float* d_x, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(x != nullptr, "dns_encode_bwd: x is NULL");
  GridLevels lv = {};
  if (d_grid) {
    DNS_REQUIRE(meta && table, "dns_encode_bwd: d_grid given without table/meta");
    DNS_REQUIRE(meta->n_features == 2, "dns_encode_bwd: n_features must be 2");
    lv = to_levels(meta);
  }
  if (d_pe) DNS_REQUIRE(n_bins >= 1 && n_bins <= 64 && ld_dpe >= 3 * n_bins, "dns_encode_bwd: n_bins %u / ld %u", n_bins, ld_dpe);
  const uint32_t blocks = (P + 255) / 256;
  hipLaunchKernelGGL(encode_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, make_bound(bound),
                     bound ? 1 : 0, P, n_bins, (const float2*)table, lv, d_pe, ld_dpe, d_grid, ld_dgrid, d_table, d_x);
  return check_launch("dns_encode_bwd");
}

extern "C" int dns_hashgrid_indices(const float* x, uint32_t P, const DnsGridMeta* meta, uint32_t* rows, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(x && meta && rows, "dns_hashgrid_indices: NULL argument");
  hipLaunchKernelGGL(hashgrid_indices_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, P,
                     to_levels(meta), rows);
  return check_launch("dns_hashgrid_indices");
}

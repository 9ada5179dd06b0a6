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
#include <type_traits>
#include "common.hpp"
#include "split_rows.hpp"
#include <cstdlib>
#include <cstring>
#include "dev_encode.hpp"

namespace dns {

// One lane = one point.  TILED = the [P, 3*n_bins + 2*L] output rows are contiguous (OneBlob | grid in one buffer):
// the lane's channels go to an LDS tile (row stride +1 float: conflict-free) and the workgroup's tile leaves as one
// contiguous, fully coalesced run.  Direct 16-byte row stores at a 320-byte lane stride were measured to cost 3.4x
// the bytes in HBM writes (rocprofv3 WRITE_SIZE 286 MB for an 84 MB output): every store instruction touches 64 lines.
// TILED: the workgroup's output rows leave as contiguous runs through an LDS tile (direct 16-byte row stores at a 320-byte
// lane stride cost 3.4x the bytes in HBM writes).  The tile is used TWICE -- OneBlob columns, flush, then grid columns,
// flush -- so it holds max(pe_dim, g_dim) + 1 floats per point (25 KB instead of 41 KB): LDS, not registers (68 VGPRs),
// is what limits this gather kernel's occupancy, and 12 waves per CU hide the table-gather latency better than 6.
// HALF (with TILED; ABI v12, DNS_SPLIT_PLAIN): the row leaves as plain f16 -- the flush converts eight tile values to one
// 16-byte store; pe_out / grid_out then point at halfs and ld_pe / ld_grid count halfs.  (The scaled split-row writer below
// evaluates OneBlob twice and flushes in four steps: 98.6 us against this kernel's 84 at 262 144 points.)
template <bool TILED, bool HALF = false>
__global__ __launch_bounds__(128) void encode_fwd_kernel(const float* __restrict__ in, Bound6 bd, int normalise,
                                                         uint32_t P, uint32_t n_bins,
                                                         const float2* __restrict__ table, GridLevels lv,
                                                         float* __restrict__ x_out, float* __restrict__ pe_out,
                                                         uint32_t ld_pe, float* __restrict__ grid_out,
                                                         uint32_t ld_grid, float2* __restrict__ dydx, uint32_t pe_ph,
                                                         uint32_t g_ph) {
  extern __shared__ float tile[];
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pe_dim = 3 * n_bins;
  // (TILED) the tile is used pe_ph + g_ph times: OneBlob in 1 or 3 phases (all axes / one axis each), the grid in g_ph groups of
  // levels -- every phase ends in a flush, so the tile holds the widest PHASE + 1 floats per point
  const uint32_t lpp = lv.n_levels / g_ph;                // levels per grid phase (host: g_ph divides n_levels)
  const uint32_t ldt = max(pe_dim / pe_ph, 2u * lpp) + 1;
  const bool live = p < P;
  float x[3] = {0.f, 0.f, 0.f};
  if (live) load_point(in, bd, normalise != 0, p, x);
  if (live && x_out) {
    x_out[(size_t)p * 3 + 0] = x[0];
    x_out[(size_t)p * 3 + 1] = x[1];
    x_out[(size_t)p * 3 + 2] = x[2];
  }
  float* trow = tile + threadIdx.x * ldt;
  auto col_ptr = [](float* base, uint32_t col) -> float* {           // column `col` of an output row (HALF: the row holds halfs)
    return HALF ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(base) + col) : base + col;
  };
  // rows [p0, p0 + rows) x columns [c0, c0 + nc) of the row-major output with leading dimension ld, from the tile
  auto flush = [&](float* out_base, uint32_t ld, uint32_t nc) {
    __syncthreads();
    const uint32_t p0 = blockIdx.x * blockDim.x;
    const uint32_t rows = min(blockDim.x, P - p0);
    float* out = out_base + (size_t)p0 * ld;
    // (row, column) advance incrementally (an integer division per element was a visible cost); 16-byte global accesses
    // where the rows allow it (the LDS tile's odd row stride keeps its side at dwords)
    if constexpr (HALF) {                              // nc, ld multiples of 8, out_base 16-byte aligned (host-checked)
      _Float16* outh = reinterpret_cast<_Float16*>(out_base) + (size_t)p0 * ld;
      const uint32_t nq = nc >> 3, dr = blockDim.x / nq, dc = blockDim.x - dr * nq;
      uint32_t r = threadIdx.x / nq, c = threadIdx.x - r * nq;
      for (uint32_t i = threadIdx.x; i < rows * nq; i += blockDim.x) {
        const float* t = tile + r * ldt + 8 * c;
        typedef _Float16 half8v __attribute__((ext_vector_type(8)));
        half8v h;
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = (_Float16)t[k];
        *reinterpret_cast<half8v*>(outh + (size_t)r * ld + 8 * c) = h;
        r += dr;
        c += dc;
        if (c >= nq) { c -= nq; ++r; }
      }
    } else if (((nc | ld) & 3u) == 0 && ((((uintptr_t)out_base) & 15u) == 0)) {
      const uint32_t nq = nc >> 2, dr = blockDim.x / nq, dc = blockDim.x - dr * nq;
      uint32_t r = threadIdx.x / nq, c = threadIdx.x - r * nq;
      for (uint32_t i = threadIdx.x; i < rows * nq; i += blockDim.x) {
        const float* t = tile + r * ldt + 4 * c;
        *reinterpret_cast<float4*>(out + (size_t)r * ld + 4 * c) = make_float4(t[0], t[1], t[2], t[3]);
        r += dr;
        c += dc;
        if (c >= nq) { c -= nq; ++r; }
      }
    } else {
      const uint32_t dr = blockDim.x / nc, dc = blockDim.x - dr * nc;
      uint32_t r = threadIdx.x / nc, c = threadIdx.x - r * nc;
      for (uint32_t i = threadIdx.x; i < rows * nc; i += blockDim.x) {
        out[(size_t)r * ld + c] = tile[r * ldt + c];
        r += dr;
        c += dc;
        if (c >= nc) { c -= nc; ++r; }
      }
    }
    __syncthreads();
  };
  if (pe_out) {
    const float n = (float)n_bins;
    float* row = TILED ? trow : pe_out + (size_t)p * ld_pe;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const uint32_t c0 = (TILED && pe_ph == 3u) ? 0u : a * n_bins;      // the axis' first tile / row column
      if (live) {
        const float xa = x[a];
        if (n_bins >= 8u && fabsf(xa) < 4.0f) {            // windows of different images cannot overlap
          for (uint32_t b = 0; b < n_bins; ++b) row[c0 + b] = 0.f;
          oneblob_windows<false>(n_bins, n, xa, [&](uint32_t j, float v) { row[c0 + j] = v; });
        } else {
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
            if (b > 0) row[c0 + b - 1] = g - left;
            left = g;
          }
        }
      }
      if (TILED && pe_ph == 3u) flush(col_ptr(pe_out, a * n_bins), ld_pe, n_bins);
    }
    if (TILED && pe_ph != 3u) flush(pe_out, ld_pe, pe_dim);
  }
  if (grid_out) {
    for (uint32_t ph = 0; ph < g_ph; ++ph) {
    if (live) {
      float* row = TILED ? trow - 2u * lpp * ph : grid_out + (size_t)p * ld_grid;     // (TILED: level l's pair at tile column 2 (l - lpp ph))
#pragma unroll 4
      for (uint32_t l = lpp * ph; l < lpp * (ph + 1u); ++l) {
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
        if (dydx) {
          // d(feature) / d(normalised coordinate), both features, per axis -- what tcnn's kernel_grid keeps as dy_dx when the
          // input needs a gradient (the poses do, through pts): the backward then needs no second gather of the 8 corners.
          // Layout [level][axis][point] float2: a wave's 64 points are 512 contiguous bytes per store.
          const float wx0 = 1.0f - f[0], wx1 = f[0], wy0 = 1.0f - f[1], wy1 = f[1], wz0 = 1.0f - f[2], wz1 = f[2];
          float2 jx, jy, jz;
          jx.x = s * (wy0 * wz0 * (v[1].x - v[0].x) + wy1 * wz0 * (v[3].x - v[2].x) + wy0 * wz1 * (v[5].x - v[4].x) + wy1 * wz1 * (v[7].x - v[6].x));
          jx.y = s * (wy0 * wz0 * (v[1].y - v[0].y) + wy1 * wz0 * (v[3].y - v[2].y) + wy0 * wz1 * (v[5].y - v[4].y) + wy1 * wz1 * (v[7].y - v[6].y));
          jy.x = s * (wx0 * wz0 * (v[2].x - v[0].x) + wx1 * wz0 * (v[3].x - v[1].x) + wx0 * wz1 * (v[6].x - v[4].x) + wx1 * wz1 * (v[7].x - v[5].x));
          jy.y = s * (wx0 * wz0 * (v[2].y - v[0].y) + wx1 * wz0 * (v[3].y - v[1].y) + wx0 * wz1 * (v[6].y - v[4].y) + wx1 * wz1 * (v[7].y - v[5].y));
          jz.x = s * (wx0 * wy0 * (v[4].x - v[0].x) + wx1 * wy0 * (v[5].x - v[1].x) + wx0 * wy1 * (v[6].x - v[2].x) + wx1 * wy1 * (v[7].x - v[3].x));
          jz.y = s * (wx0 * wy0 * (v[4].y - v[0].y) + wx1 * wy0 * (v[5].y - v[1].y) + wx0 * wy1 * (v[6].y - v[2].y) + wx1 * wy1 * (v[7].y - v[3].y));
          dydx[((size_t)l * 3 + 0) * P + p] = jx;
          dydx[((size_t)l * 3 + 1) * P + p] = jy;
          dydx[((size_t)l * 3 + 2) * P + p] = jz;
        }
      }
    }
    if (TILED) flush(col_ptr(grid_out, 2u * lpp * ph), ld_grid, 2u * lpp);
    }
  }
}
// The same encoding written in the SPLIT-ROW format of split_rows.hpp (and, optionally, as fp32 rows as well): what the MLP
// kernels of fused_step.MapStep / TrackStep read.  One lane = one point, outputs through the same two-phase LDS tile as
// encode_fwd_kernel<true> (max(pe_dim, g_dim) + 1 floats per point: LDS, not registers, limits the occupancy of this gather
// kernel).  The row's exponent needs max |row| over BOTH halves before the first value is converted:
//   A  OneBlob -> tile (fp32), its maximum; flushed as fp32 columns [0, pe_dim) when f32_out is wanted
//   B  grid gather -> tile (fp32), its maximum -> e = scale_exp(max) (bit for bit what mlp_split.hpp's x_row_max / scale_exp
//      derived from the fp32 row); fp32 flush; then the 2 L values as hi | lo pairs, flushed into the row's two planes
//   C  OneBlob evaluated a second time (the OneBlob half of this kernel is free: measured 99.7 vs 96.0 us with / without it),
//      converted with e and flushed.
// xs_out [P][ldxs] halfs: columns [0, K) hi, [K, 2K) lo (K = pe_dim + g_dim); hi_only: the lo plane is not written.
__global__ __launch_bounds__(128) void encode_fwd_split_kernel(const float* __restrict__ in, Bound6 bd, int normalise, uint32_t P,
                                                               uint32_t n_bins, const float2* __restrict__ table, GridLevels lv,
                                                               float* __restrict__ x_out, float* __restrict__ f32_out, uint32_t ld32,
                                                               uint32_t* __restrict__ xs_out, uint32_t ldxs_w, int32_t* __restrict__ xexp,
                                                               int hi_only, float2* __restrict__ dydx) {
  extern __shared__ float tile[];
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pe_dim = 3 * n_bins, g_dim = 2 * lv.n_levels, K = pe_dim + g_dim;
  const uint32_t ldt = max(pe_dim, g_dim) + 1;
  const bool live = p < P;
  float x[3] = {0.f, 0.f, 0.f};
  if (live) load_point(in, bd, normalise != 0, p, x);
  if (live && x_out) {
    x_out[(size_t)p * 3 + 0] = x[0];
    x_out[(size_t)p * 3 + 1] = x[1];
    x_out[(size_t)p * 3 + 2] = x[2];
  }
  float* trow = tile + threadIdx.x * ldt;
  // tile columns [tc0, tc0 + nc) of the workgroup's rows -> columns [c0, c0 + nc) of the row-major `out` (32-bit words, leading
  // dimension ld); nc, c0, ld multiples of 4 and out 16-byte aligned (checked on the host): 16-byte global stores
  auto flush = [&](uint32_t* out_base, uint32_t ld, uint32_t c0, uint32_t nc, uint32_t tc0) {
    __syncthreads();
    const uint32_t p0 = blockIdx.x * blockDim.x;
    const uint32_t rows = min(blockDim.x, P - p0);
    uint32_t* out = out_base + (size_t)p0 * ld + c0;
    const uint32_t nq = nc >> 2, dr = blockDim.x / nq, dc = blockDim.x - dr * nq;
    uint32_t r = threadIdx.x / nq, c = threadIdx.x - r * nq;
    for (uint32_t i = threadIdx.x; i < rows * nq; i += blockDim.x) {
      const uint32_t* t = reinterpret_cast<const uint32_t*>(tile) + r * ldt + tc0 + 4 * c;
      *reinterpret_cast<uint4*>(out + (size_t)r * ld + 4 * c) = make_uint4(t[0], t[1], t[2], t[3]);
      r += dr;
      c += dc;
      if (c >= nq) { c -= nq; ++r; }
    }
    __syncthreads();
  };
  auto oneblob_row = [&]() -> float {                     // the lane's pe_dim OneBlob values -> its tile row; their maximum
    const float n = (float)n_bins;
    float m = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float xa = x[a];
      if (n_bins >= 8u && fabsf(xa) < 4.0f) {
        for (uint32_t b = 0; b < n_bins; ++b) trow[a * n_bins + b] = 0.f;
        oneblob_windows<false>(n_bins, n, xa, [&](uint32_t j, float v) { trow[a * n_bins + j] = v; m = fmaxf(m, fabsf(v)); });
        continue;
      }
      float first = 0.f, left = 0.f;
      for (uint32_t b = 0; b <= n_bins; ++b) {
        float g;
        if (b < n_bins) {
          const float d = (float)b / n - xa;
          g = quartic_cdf(d, n) + quartic_cdf(d - 1.0f, n) + quartic_cdf(d + 1.0f, n);
          if (b == 0) first = g;
        } else {
          g = first + 1.0f;
        }
        if (b > 0) { trow[a * n_bins + b - 1] = g - left; m = fmaxf(m, fabsf(g - left)); }
        left = g;
      }
    }
    return m;
  };
  // ---- A
  float rmax = 0.f;
  bool bad = false;                                       // fmaxf drops a NaN: a non-finite value must reach the exponent rule
  if (live) {
    rmax = oneblob_row();
    for (uint32_t c = 0; c < pe_dim; ++c) bad = bad || !(fabsf(trow[c]) < INFINITY);
  }
  if (f32_out) flush(reinterpret_cast<uint32_t*>(f32_out), ld32, 0, pe_dim, 0);
  else __syncthreads();
  const bool plain = (hi_only & 2) != 0;                  // DNS_SPLIT_PLAIN: unscaled f16 values, no exponent
  // ---- B
  if (live) {
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
      trow[2 * l] = a0;
      trow[2 * l + 1] = a1;
      rmax = fmaxf(rmax, fmaxf(fabsf(a0), fabsf(a1)));
      bad = bad || !(fabsf(a0) < INFINITY) || !(fabsf(a1) < INFINITY);
      if (dydx) {
        const float wx0 = 1.0f - f[0], wx1 = f[0], wy0 = 1.0f - f[1], wy1 = f[1], wz0 = 1.0f - f[2], wz1 = f[2];
        float2 jx, jy, jz;
        jx.x = s * (wy0 * wz0 * (v[1].x - v[0].x) + wy1 * wz0 * (v[3].x - v[2].x) + wy0 * wz1 * (v[5].x - v[4].x) + wy1 * wz1 * (v[7].x - v[6].x));
        jx.y = s * (wy0 * wz0 * (v[1].y - v[0].y) + wy1 * wz0 * (v[3].y - v[2].y) + wy0 * wz1 * (v[5].y - v[4].y) + wy1 * wz1 * (v[7].y - v[6].y));
        jy.x = s * (wx0 * wz0 * (v[2].x - v[0].x) + wx1 * wz0 * (v[3].x - v[1].x) + wx0 * wz1 * (v[6].x - v[4].x) + wx1 * wz1 * (v[7].x - v[5].x));
        jy.y = s * (wx0 * wz0 * (v[2].y - v[0].y) + wx1 * wz0 * (v[3].y - v[1].y) + wx0 * wz1 * (v[6].y - v[4].y) + wx1 * wz1 * (v[7].y - v[5].y));
        jz.x = s * (wx0 * wy0 * (v[4].x - v[0].x) + wx1 * wy0 * (v[5].x - v[1].x) + wx0 * wy1 * (v[6].x - v[2].x) + wx1 * wy1 * (v[7].x - v[3].x));
        jz.y = s * (wx0 * wy0 * (v[4].y - v[0].y) + wx1 * wy0 * (v[5].y - v[1].y) + wx0 * wy1 * (v[6].y - v[2].y) + wx1 * wy1 * (v[7].y - v[3].y));
        dydx[((size_t)l * 3 + 0) * P + p] = jx;
        dydx[((size_t)l * 3 + 1) * P + p] = jy;
        dydx[((size_t)l * 3 + 2) * P + p] = jz;
      }
    }
  }
  const int e = plain ? 0 : sr::scale_exp(bad ? INFINITY : rmax);
  const float sc = ldexpf(1.0f, e);
  if (live && !plain) xexp[p] = e;
  if (f32_out) flush(reinterpret_cast<uint32_t*>(f32_out), ld32, pe_dim, g_dim, 0);
  // the lane's own row back out of the tile, as packed pairs: hi pairs at tile columns [0, n/2), lo pairs at [n/2, n)
  auto repack = [&](uint32_t n) {
    uint32_t* urow = reinterpret_cast<uint32_t*>(trow);
    uint32_t hi[32], lo[32];                               // n <= 64 values (host-checked)
#pragma unroll
    for (uint32_t i = 0; i < 32; ++i) {
      if (2 * i < n) sr::split_pair(trow[2 * i], trow[2 * i + 1], sc, hi[i], lo[i]);
    }
#pragma unroll
    for (uint32_t i = 0; i < 32; ++i) {
      if (2 * i < n) {
        urow[i] = hi[i];
        urow[n / 2 + i] = lo[i];
      }
    }
  };
  if (live) repack(g_dim);
  flush(xs_out, ldxs_w, pe_dim / 2, g_dim / 2, 0);
  if (!hi_only) flush(xs_out, ldxs_w, K / 2 + pe_dim / 2, g_dim / 2, g_dim / 2);
  // (plain rows: the OneBlob values need no row maximum, but the tile holds one half of the row at a time: evaluated again like
  //  the scaled form -- measured free beside the gather, 99.7 vs 96.0 us)
  // ---- C
  if (live) {
    (void)oneblob_row();
    repack(pe_dim);
  }
  flush(xs_out, ldxs_w, 0, pe_dim / 2, 0);
  if (!hi_only) flush(xs_out, ldxs_w, K / 2, pe_dim / 2, pe_dim / 2);
}

template <bool TILED>
__global__ __launch_bounds__(128) void encode_bwd_kernel(const float* __restrict__ xin, Bound6 bd, int scale_by_bound,
                                                         uint32_t P, uint32_t n_bins,
                                                         const float2* __restrict__ table, GridLevels lv,
                                                         const float* __restrict__ d_pe, uint32_t ld_dpe,
                                                         const float* __restrict__ d_grid, uint32_t ld_dgrid,
                                                         float* __restrict__ d_table, float* __restrict__ d_x,
                                                         const float2* __restrict__ dydx, uint32_t pe_ph, uint32_t g_ph) {
  extern __shared__ float tile[];
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pe_dim = 3 * n_bins;
  const uint32_t lpp = g_ph ? lv.n_levels / g_ph : 0u;    // levels per grid phase (TILED; host: g_ph divides n_levels; 0: OneBlob columns only)
  // TILED: the workgroup's gradient rows come in through an LDS tile (coalesced row reads instead of one 320-byte-strided
  // row per lane), in TWO phases like the forward -- OneBlob columns, then grid columns -- so that the tile holds
  // max(pe_dim, g_dim) + 1 floats per point (25 KB, 12 waves per CU) instead of the whole row (41 KB, 6 waves): this
  // kernel is latency-bound (dependent LDS reads, 8-byte gathers), occupancy is what it lacked.
  // (round 5: pe_ph = 3 / g_ph = 2 phases -- one OneBlob axis / half the levels at a time, 17 floats per point: see encode_fwd_kernel)
  const uint32_t ldt = max(pe_dim / pe_ph, 2u * lpp) + 1;
  const bool live = p < P;
  auto stage = [&](uint32_t col0, uint32_t nc) {         // columns [col0, col0 + nc) of the workgroup's rows -> tile
    __syncthreads();
    const uint32_t p0 = blockIdx.x * blockDim.x;
    const uint32_t rows = min(blockDim.x, P - p0);
    const float* src = d_pe + (size_t)p0 * ld_dpe + col0;
    if (((nc | ld_dpe | col0) & 3u) == 0 && ((((uintptr_t)d_pe) & 15u) == 0)) {
      const uint32_t nq = nc >> 2, dr = blockDim.x / nq, dc = blockDim.x - dr * nq;
      uint32_t r = threadIdx.x / nq, c = threadIdx.x - r * nq;
      for (uint32_t i = threadIdx.x; i < rows * nq; i += blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * ld_dpe + 4 * c);
        float* t = tile + r * ldt + 4 * c;
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        r += dr;
        c += dc;
        if (c >= nq) { c -= nq; ++r; }
      }
    } else {
      const uint32_t dr = blockDim.x / nc, dc = blockDim.x - dr * nc;
      uint32_t r = threadIdx.x / nc, c = threadIdx.x - r * nc;
      for (uint32_t i = threadIdx.x; i < rows * nc; i += blockDim.x) {
        tile[r * ldt + c] = src[(size_t)r * ld_dpe + c];
        r += dr;
        c += dc;
        if (c >= nc) { c -= nc; ++r; }
      }
    }
    __syncthreads();
  };
  float x[3] = {0.f, 0.f, 0.f};
  if (live) {
    x[0] = xin[(size_t)p * 3 + 0];
    x[1] = xin[(size_t)p * 3 + 1];
    x[2] = xin[(size_t)p * 3 + 2];
  }
  const bool pe_split = TILED && pe_ph == 3u;
  if (TILED && !pe_split) stage(0, pe_dim);
  float dx[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (pe_split) stage(a * n_bins, n_bins);               // (uniform: every thread of the workgroup)
    if (live && d_pe && d_x) {
      const float n = (float)n_bins;
      const float* row = TILED ? tile + threadIdx.x * ldt - (pe_split ? a * n_bins : 0u) : d_pe + (size_t)p * ld_dpe;
      const float xa = x[a];
      if (n_bins >= 8u && fabsf(xa) < 4.0f) {
        float accw = 0.f;
        oneblob_windows<true>(n_bins, n, xa, [&](uint32_t j, float v) { accw -= row[a * n_bins + j] * v; });
        dx[a] += accw;
        continue;
      }
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
  for (uint32_t ph = 0; ph < (TILED ? g_ph : 1u); ++ph) {
  const uint32_t l_lo = TILED ? lpp * ph : 0u, l_hi = TILED ? lpp * (ph + 1u) : lv.n_levels;
  if (TILED) stage(pe_dim + 2u * l_lo, 2u * (l_hi - l_lo));
  if (live && d_grid && dydx && d_x && !d_table) {
    // the forward kept d(features)/dx: a streaming dot product, no gather (coalesced 8-byte reads, lane = point).
    // (Requesting all 48 values before the first staging barrier was measured: 96 more registers, 0.102 -> 0.113 ms.)
    const float* row = TILED ? tile + threadIdx.x * ldt - 2u * l_lo : d_grid + (size_t)p * ld_dgrid;
#pragma unroll 4
    for (uint32_t l = l_lo; l < l_hi; ++l) {
      const float g0 = row[2 * l], g1 = row[2 * l + 1];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float2 j = dydx[((size_t)l * 3 + a) * P + p];
        dx[a] += j.x * g0 + j.y * g1;
      }
    }
  } else if (live && d_grid) {
    const float* row = TILED ? tile + threadIdx.x * ldt - 2u * l_lo : d_grid + (size_t)p * ld_dgrid;
#pragma unroll 2
    for (uint32_t l = l_lo; l < l_hi; ++l) {
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
  }   // grid phases
  if (live && d_x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float v = dx[a];
      if (scale_by_bound) v = (float)((double)v / (bd.b1[a] - bd.b0[a]));
      d_x[(size_t)p * 3 + a] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Table gradient by LDS binning (float64 bins in the binned kernel, 64-bit fixed point in the queue kernel).
//
// Per-corner global atomics (one lane = one random row) run ~17x below the coalesced atomic rate on gfx950
// and serialise on the 4096-cell coarse levels.  Instead a workgroup owns one CHUNK of one level (8192 rows held
// in LDS), sweeps a slice of the points, recomputes that level's 8 corner rows per point (cheap integer math)
// and accumulates the corners that fall in its chunk; the chunk then leaves as fully coalesced global float
// atomics (256 B per wave instruction, the full-rate shape).
//
// The LDS accumulators are 64 bits wide, never fp32: measured on MI355X (tools/lds_atomic_rate.hip) ds_add_f32 retires
// ~1 lane per 3 cycles per CU (195 cycles per wave instruction, 0.2 T lane-atomics/s chip-wide) while ds_add_u64 takes 11.9
// cycles per wave instruction (3.3 T/s) and ds_add_f64 21.6.  (fp32 bins over 16 384-row chunks -- half the re-hashing --
// were measured in round 2: 476 us against 312 us per 262 144 points.)  Round 1 added every contribution w*g as an integer
// (scaled by a power of two chosen from max|dL/dy| of the launch so the fp32 product converts to int64 exactly): exact,
// order-independent sums, at 11 vector instructions per converted value in a VALU-bound kernel.  The binned kernel now adds
// the fp32 product into a float64 bin (ds_add_f64): 276 us, sums good to ~1e-16 whatever the order; the queue kernel (large
// tables) keeps the fixed-point form.
struct BinPlan {
  uint32_t n_levels;
  uint32_t chunk_rows;                    // rows per chunk
  uint32_t strided_dense;                 // dense levels: one contiguous run of points per thread
  uint32_t dense_runs;                    // ... whose per-cell sums are kept in registers until the cell changes
  uint32_t job_prefix[DNS_MAX_LEVELS + 1];  // prefix sum over levels of chunks[l] * slices[l]
  uint32_t group_prefix[DNS_MAX_LEVELS + 1];  // prefix sum over levels of slices[l]: a group = one (level, slice)
  uint32_t xcd_major;                     // 1: blockIdx -> (xcd = b % 8, q = b / 8), a group's chunks adjacent in q
  uint32_t chunks[DNS_MAX_LEVELS];
  uint32_t slices[DNS_MAX_LEVELS];
};

// d_grid rows [P, ld] (32 contiguous floats per point) -> level-major [L][P] float2, so that a binned job reads
// only its level's 8 bytes per point; also max |d_grid| of the launch (bit pattern, atomicMax) for the fixed-point scale.
// Row replay (DNS_SCATTER_REPLAY): per hashed level of at most 2^16 rows the 8 corner rows of every point are computed HERE,
// once, and stored as eight 16-bit values (16 bytes per point and level, coalesced): the binned kernel's 8 visits per level
// then extract and compare them instead of hashing the 8 corners again in every visit.
struct ReplayPlan {
  int32_t slot[DNS_MAX_LEVELS];                  // level -> index of its [P] uint4 plane in rows16, -1 = not replayed
};

constexpr uint32_t DG_TILES = 2;                 // 256-point tiles per workgroup of the transpose
__global__ __launch_bounds__(256) void dgrid_transpose_kernel(const float* __restrict__ d_grid, uint32_t ld, uint32_t P,
                                                              uint32_t n_levels, float2* __restrict__ dg_t,
                                                              uint32_t* __restrict__ gmax, const float* __restrict__ xin,
                                                              GridLevels lv, ReplayPlan rp, uint4* __restrict__ rows16,
                                                              uint32_t n_tiles) {
  // 256 points per workgroup through an LDS tile: rows are read as whole 128-byte lines (8 lanes x 16 B per point),
  // level planes are written as 2-KB contiguous runs (lane = point).  Row stride 34 floats keeps both sides <= 2-way.
  // A workgroup takes DG_TILES consecutive tiles: the NEXT tile's rows are requested before this tile goes through LDS and out, so
  // the read stream of one tile overlaps the write stream of the one before (one tile per workgroup ran the two phases of the
  // whole grid in lock-step: 26 us for 67 MB).
  constexpr uint32_t LDT = 34;
  __shared__ float tile[256 * LDT];
  const uint32_t nf = n_levels * 2;                  // floats per row (<= 32 with the supported 16 levels)
  const bool vec = ((ld & 3u) == 0) && ((((uintptr_t)d_grid) & 15u) == 0) && ((nf & 3u) == 0);
  float m = 0.f;
  bool bad = false;                                  // a NaN / Inf in the upstream gradient
  // all eight 16-byte loads of a thread are requested before the first is used (clamped addresses, no branch between
  // them: a guarded load per trip made every trip wait its own memory round trip)
  auto issue = [&](uint32_t p0, float4 (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t e = threadIdx.x + 256u * i;
      const uint32_t r = e >> 3, q = e & 7u;
      const uint32_t pc = min(p0 + r, P - 1u), qc = min(4u * q, nf - 4u);
      v[i] = *reinterpret_cast<const float4*>(d_grid + (size_t)pc * ld + qc);
    }
  };
  float4 v[8], vn[8];
  const uint32_t tile0 = blockIdx.x * n_tiles;
  if (vec && tile0 * 256u < P) issue(tile0 * 256u, v);
  for (uint32_t t = 0; t < n_tiles; ++t) {
    const uint32_t p0 = (tile0 + t) * 256u;
    if (p0 >= P) break;                              // uniform
    if (vec) {
      if (t + 1 < n_tiles && p0 + 256u < P) issue(p0 + 256u, vn);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t e = threadIdx.x + 256u * i;
        const uint32_t r = e >> 3, q = e & 7u;
        if (!(p0 + r < P && 4u * q < nf)) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = tile + r * LDT + 4 * q;
        d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
        bad = bad || !(fabsf(v[i].x) < INFINITY) || !(fabsf(v[i].y) < INFINITY) || !(fabsf(v[i].z) < INFINITY) || !(fabsf(v[i].w) < INFINITY);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = vn[i];
    } else {
      for (uint32_t e = threadIdx.x; e < 256u * 8u; e += 256u) {
        const uint32_t r = e >> 3, q = e & 7u;
        const uint32_t p = p0 + r;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < P && 4 * q < nf) {
          const float* src = d_grid + (size_t)p * ld + 4 * q;
          w.x = src[0];
          if (4 * q + 1 < nf) w.y = src[1];
          if (4 * q + 2 < nf) w.z = src[2];
          if (4 * q + 3 < nf) w.w = src[3];
        }
        float* d = tile + r * LDT + 4 * q;
        d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(w.x), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w))));
        bad = bad || !(fabsf(w.x) < INFINITY) || !(fabsf(w.y) < INFINITY) || !(fabsf(w.z) < INFINITY) || !(fabsf(w.w) < INFINITY);
      }
    }
    __syncthreads();
    const uint32_t p = p0 + threadIdx.x;
    if (p < P) {
      for (uint32_t l = 0; l < n_levels; ++l)
        dg_t[(size_t)l * P + p] = make_float2(tile[threadIdx.x * LDT + 2 * l], tile[threadIdx.x * LDT + 2 * l + 1]);
      if (rows16) {
        const float x0 = xin[(size_t)p * 3], x1 = xin[(size_t)p * 3 + 1], x2 = xin[(size_t)p * 3 + 2];
        for (uint32_t l = 0; l < n_levels; ++l) {
          const int slot = rp.slot[l];
          if (slot < 0) continue;                      // uniform
          const float sc = lv.scale[l];
          const uint32_t g0 = (uint32_t)(int)floorf(__fadd_rn(__fmul_rn(x0, sc), 0.5f));
          const uint32_t g1 = (uint32_t)(int)floorf(__fadd_rn(__fmul_rn(x1, sc), 0.5f));
          const uint32_t g2 = (uint32_t)(int)floorf(__fadd_rn(__fmul_rn(x2, sc), 0.5f));
          const uint32_t mask = lv.size[l] - 1u;       // hashed levels are exactly 2^T rows (<= 2^16 here)
          const uint32_t ay0 = g1 * 2654435761u, ay1 = ay0 + 2654435761u, az0 = g2 * 805459861u, az1 = az0 + 805459861u;
          uint32_t r[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) r[c] = ((g0 + (uint32_t)(c & 1)) ^ ((c & 2) ? ay1 : ay0) ^ ((c & 4) ? az1 : az0)) & mask;
          rows16[(size_t)slot * P + p] = make_uint4(r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16));
        }
      }
    }
    __syncthreads();                                   // the tile is overwritten by the next trip
  }
  // max |d_grid|: wave reduce, then ONE conditional atomic per workgroup -- 4096 unconditional same-address atomics
  // serialised into ~40 us of this kernel's 56; a (possibly stale) read of the running max lets all but the first few
  // workgroups skip theirs (staleness only costs a redundant atomic, never a wrong maximum)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wmax[4];
  if ((threadIdx.x & 63u) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float bm = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (bm > 0.f && bm < INFINITY && bm > __uint_as_float(__atomic_load_n(gmax, __ATOMIC_RELAXED))) atomicMax(gmax, __float_as_uint(bm));
  }
  // The fixed-point bins cannot carry NaN / Inf (fmaxf drops a NaN, the integer conversion of a non-finite product is
  // undefined): a non-finite upstream gradient raises gmax[1] instead and the scatter kernels then write NaN into every
  // row they own, so a diverged step shows up in d_table exactly as it does with per-corner float atomics.
  if (__any(bad) && (threadIdx.x & 63u) == 0 && __atomic_load_n(gmax + 1, __ATOMIC_RELAXED) == 0u) atomicOr(gmax + 1, 1u);
}

// round-to-nearest-even of an fp32 value with |v| < 2^40 to a 64-bit integer: v = hi * 2^16 + lo exactly (hi = trunc(v /
// 2^16), |hi| < 2^24; lo = v - hi * 2^16 is exact in fp32), and rint(v) = hi * 2^16 + rint(lo) because hi * 2^16 is even
__device__ __forceinline__ long long fixed_rn(float v) {
  const float hi_f = truncf(v * (1.0f / 65536.0f));
  const float lo_f = fmaf(-hi_f, 65536.0f, v);
  const int hi = (int)hi_f, lo = (int)rintf(lo_f);
  return ((long long)hi << 16) + (long long)lo;
}

// the same for |v| < 2^51 through the float64 adder: v + 1.5 * 2^52 has v's nearest integer (ties to even) in its low mantissa
// bits; subtracting the constant's bit pattern leaves it as a two's-complement 64-bit integer (4 instructions instead of 11)
__device__ __forceinline__ unsigned long long fixed_rn_f64(float v) {
  const double d = (double)v + 6755399441055744.0;
  return (unsigned long long)__double_as_longlong(d) - 0x4338000000000000ull;
}

__global__ __launch_bounds__(1024) void hashgrid_bwd_binned_kernel(const float* __restrict__ xin, uint32_t P,
                                                                    GridLevels lv, BinPlan plan,
                                                                    const float2* __restrict__ dg_t,
                                                                    const uint32_t* __restrict__ gmax,
                                                                    float* __restrict__ d_table, ReplayPlan rp,
                                                                    const uint4* __restrict__ rows16) {
  // Bins: one float64 per table float, added with ds_add_f64.  Every w*g product is formed in fp32 (as tcnn forms it) and
  // summed in float64, so a cell's sum carries ~1e-16 of relative error whatever the order of the adds -- after the final
  // rounding to fp32 the result is the correctly rounded sum except on near-ties.  Measured LDS atomic rates on MI355X
  // (tools/lds_atomic_rate.hip, cycles per wave-instruction per CU): ds_add_u64 11.9, ds_add_f64 21.6, ds_add_f32 195.
  // Round 1 used 64-bit fixed-point bins (ds_add_u64 of an 11-instruction exact float -> fixed conversion per value): this
  // kernel is VALU-bound, the conversions cost more than the slower atomic (312 -> 276 us per 262 144 points).
  extern __shared__ __attribute__((aligned(16))) unsigned long long bins[];
  double* const dbins = reinterpret_cast<double*>(bins);
  const float mx = __uint_as_float(*gmax);
  const bool poisoned = gmax[1] != 0u;           // NaN / Inf upstream (dgrid_transpose_kernel): NaN into the owned rows
  if (!(mx > 0.f) && !poisoned) return;          // all-zero upstream gradient: nothing to add (uniform exit)
  // job -> (level, chunk, slice).  All chunks of one (level, slice) group sweep the SAME points, so they should share
  // an L2: workgroups are dealt round-robin over the 8 XCDs (xcd = blockIdx % 8), hence group g goes to XCD g % 8 and
  // its chunks sit at consecutive q = blockIdx / 8 -- they start together on that XCD's 32 CUs and 7 of 8 point reads
  // hit L2 instead of each XCD pulling the slice from HBM on its own.
  uint32_t l = 0, chunk, slice;
  if (plan.xcd_major) {
    const uint32_t xcd = blockIdx.x & 7u;
    uint32_t q = blockIdx.x >> 3;
    bool found = false;
    for (l = 0; l < plan.n_levels; ++l) {
      const uint32_t g_lo = plan.group_prefix[l], g_hi = plan.group_prefix[l + 1];
      const uint32_t first = g_lo + ((xcd + 8u - (g_lo & 7u)) & 7u);          // first group of this level on this XCD
      const uint32_t cnt = first < g_hi ? (g_hi - first + 7u) / 8u : 0u;
      const uint32_t nj = cnt * plan.chunks[l];
      if (q < nj) {
        slice = first + 8u * (q / plan.chunks[l]) - g_lo;
        chunk = q % plan.chunks[l];
        found = true;
        break;
      }
      q -= nj;
    }
    if (!found) return;                          // padding workgroup of the XCD-major grid (uniform exit)
  } else {
    while (l + 1 < plan.n_levels && blockIdx.x >= plan.job_prefix[l + 1]) ++l;
    const uint32_t rem = blockIdx.x - plan.job_prefix[l];
    chunk = rem / plan.slices[l];
    slice = rem % plan.slices[l];
  }
  const uint32_t ns = plan.slices[l];
  const uint32_t base = chunk * plan.chunk_rows;
  const uint32_t size = lv.size[l];
  const uint32_t rows = min(plan.chunk_rows, size - base);
  if (poisoned) {                                // uniform
    float* out = d_table + 2 * ((size_t)lv.offset[l] + base);
    for (uint32_t i = threadIdx.x; i < rows * 2; i += blockDim.x) out[i] = __uint_as_float(0x7fc00000u);
    return;
  }
  for (uint32_t i = threadIdx.x; i < rows * 2; i += blockDim.x) dbins[i] = 0.0;
  __syncthreads();
  const float s = lv.scale[l];
  const uint32_t res = lv.resolution[l], hashed = lv.hashed[l];
  const uint32_t p_lo = (uint32_t)(((uint64_t)P * slice) / ns), p_hi = (uint32_t)(((uint64_t)P * (slice + 1)) / ns);
  const float2* __restrict__ dgl = dg_t + (size_t)l * P;
  // Hashed levels: consecutive lanes = consecutive points (coalesced reads; the hash spreads neighbouring cells).
  // Dense (coarse) levels: neighbouring samples of a ray sit in the same cell, so a coalesced walk puts ~64 lanes on a
  // handful of LDS addresses per atomic; there each thread walks its own contiguous run instead (DNS_SCATTER_WALK=c
  // forces the coalesced walk everywhere, for measurement).
  const uint32_t n_it = (p_hi - p_lo + blockDim.x - 1) / blockDim.x;
  const bool strided = !hashed && plan.strided_dense;
  auto point_of = [&](uint32_t it) -> uint32_t {
    return strided ? p_lo + threadIdx.x * n_it + it : p_lo + it * blockDim.x + threadIdx.x;
  };
  // software pipeline: the next point's 20 bytes are requested before this point's corners are processed, so the
  // two L2 round trips per step hide under the integer / LDS work of the 4 waves per SIMD this kernel can hold
  float2 gg_n = make_float2(0.f, 0.f);
  float xn[3] = {0.f, 0.f, 0.f};
  {
    const uint32_t p = point_of(0);
    if (n_it > 0 && p < p_hi) {
      gg_n = dgl[p];
      xn[0] = xin[(size_t)p * 3];
      xn[1] = xin[(size_t)p * 3 + 1];
      xn[2] = xin[(size_t)p * 3 + 2];
    }
  }
  // The kernel is VALU-bound (SQ counters: 77 % VALU-busy, 288 vector instructions per 64-point visit before this form):
  // the per-axis hash / stride terms are computed once per point (two quarter-rate multiplies instead of two per corner
  // and again per hit), the dense / hashed split is hoisted to a uniform branch, and the bins take the fp32 product as a
  // float64 (one conversion) instead of a 64-bit fixed-point value (11 instructions).
  auto sweep = [&](auto hashed_tag) {
    constexpr bool HASHED = decltype(hashed_tag)::value;
    const uint32_t mask = size - 1u, res2 = res * res;
    for (uint32_t it = 0; it < n_it; ++it) {
      const float2 gg = gg_n;
      const float xc[3] = {xn[0], xn[1], xn[2]};
      const bool live = point_of(it) < p_hi;
      gg_n = make_float2(0.f, 0.f);
      if (it + 1 < n_it) {
        const uint32_t pn = point_of(it + 1);
        if (pn < p_hi) {
          gg_n = dgl[pn];
          xn[0] = xin[(size_t)pn * 3];
          xn[1] = xin[(size_t)pn * 3 + 1];
          xn[2] = xin[(size_t)pn * 3 + 2];
        }
      }
      const bool work = live && !(gg.x == 0.f && gg.y == 0.f);
      float f[3];
      uint32_t g[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float pos = __fadd_rn(__fmul_rn(xc[a], s), 0.5f);
        const float fl = floorf(pos);
        g[a] = (uint32_t)(int)fl;
        f[a] = pos - fl;
      }
      // per-axis terms of the row index: hashed  x ^ y*P1 ^ z*P2,  dense  x + y*res + z*res^2
      const uint32_t ax0 = g[0], ax1 = g[0] + 1u;
      const uint32_t ay0 = HASHED ? g[1] * 2654435761u : g[1] * res, ay1 = ay0 + (HASHED ? 2654435761u : res);
      const uint32_t az0 = HASHED ? g[2] * 805459861u : g[2] * res2, az1 = az0 + (HASHED ? 805459861u : res2);
      auto row_of = [&](uint32_t x, uint32_t y, uint32_t z) -> uint32_t {
        if (HASHED) return (x ^ y ^ z) & mask;                            // hashed levels are exactly 2^T rows
        uint32_t idx = x + y + z;
        if (idx >= size) idx %= size;
        return idx;
      };
      const uint32_t rows_eff = work ? rows : 0u;                          // idle lanes never hit
      // Corners are handled as the four x-PAIRS (x, x + 1 at the same y, z): on a hashed level the two rows of a pair differ
      // only in low bits (x ^ (x + 1) is a run of ones, the chunk is the row's high bits), so a pair lands in ONE chunk --
      // about half of the points have a pair here (~4 random chunks of 8 per point) -- and on a dense level the two rows are
      // neighbours.  A branch-free pass builds the lane's 4-bit pair mask, then the wave pops one pair per lane per round: half
      // the rounds of a per-corner loop (SQ counters, round 4: this kernel is VALU-bound, 75 % busy, and the pop loop -- not the
      // hashing, which a stored-rows variant removed without any effect -- is most of its vector instructions).
      uint32_t hit_mask = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t y = (c & 1) ? ay1 : ay0, z = (c & 2) ? az1 : az0;
        const uint32_t l0 = row_of(ax0, y, z) - base, l1 = row_of(ax1, y, z) - base;
        hit_mask |= ((l0 < rows_eff || l1 < rows_eff) ? 1u : 0u) << c;
      }
      while (__any(hit_mask != 0)) {
        if (hit_mask) {
          const uint32_t c = (uint32_t)__ffs((int)hit_mask) - 1u;
          hit_mask &= hit_mask - 1u;
          const uint32_t y = (c & 1u) ? ay1 : ay0, z = (c & 2u) ? az1 : az0;
          const uint32_t l0 = row_of(ax0, y, z) - base, l1 = row_of(ax1, y, z) - base;
          const float wyz = ((c & 1u) ? f[1] : 1.0f - f[1]) * ((c & 2u) ? f[2] : 1.0f - f[2]);
          const float w0 = (1.0f - f[0]) * wyz, w1 = f[0] * wyz;
          if (l0 < rows_eff) {
            atomicAdd(dbins + 2 * l0, (double)(w0 * gg.x));
            atomicAdd(dbins + 2 * l0 + 1, (double)(w0 * gg.y));
          }
          if (l1 < rows_eff) {
            atomicAdd(dbins + 2 * l1, (double)(w1 * gg.x));
            atomicAdd(dbins + 2 * l1 + 1, (double)(w1 * gg.y));
          }
        }
      }
    }
  };
  // Dense levels, strided walk: a thread's consecutive points are consecutive samples of a ray (or Morton neighbours of the
  // lattice) and stay in one cell for a while on these coarse levels, and near a camera a thousand rays share a handful of
  // cells -- per-point atomics pile many deep on single LDS addresses (measured: the four dense levels of the 16-level grid cost
  // 76 us of the launch).  The thread keeps the 8 corners x 2 features of its CURRENT cell in float64 registers and adds them to
  // the bins when the cell changes: one set of atomics per run instead of per point, with few lanes active at a time.
  auto sweep_dense_runs = [&]() {
    const uint32_t res2 = res * res;
    uint32_t cur = 0xffffffffu;                  // row of the current cell's corner (0, 0, 0); a live cell never has this value
    bool in_range = false;
    double acc[8][2];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c][0] = acc[c][1] = 0.0;
    auto flush = [&]() {
#pragma unroll
      for (uint32_t c = 0; c < 8; ++c) {
        uint32_t idx = cur + (c & 1u) + ((c >> 1) & 1u) * res + (c >> 2) * res2;
        if (idx >= size) idx %= size;
        const uint32_t local = idx - base;
        if (local < rows) {
          atomicAdd(dbins + 2 * local, acc[c][0]);
          atomicAdd(dbins + 2 * local + 1, acc[c][1]);
        }
        acc[c][0] = acc[c][1] = 0.0;
      }
    };
    for (uint32_t it = 0; it < n_it; ++it) {
      const float2 gg = gg_n;
      const float xc[3] = {xn[0], xn[1], xn[2]};
      const bool live = point_of(it) < p_hi;
      gg_n = make_float2(0.f, 0.f);
      if (it + 1 < n_it) {
        const uint32_t pn = point_of(it + 1);
        if (pn < p_hi) {
          gg_n = dgl[pn];
          xn[0] = xin[(size_t)pn * 3];
          xn[1] = xin[(size_t)pn * 3 + 1];
          xn[2] = xin[(size_t)pn * 3 + 2];
        }
      }
      if (!(live && !(gg.x == 0.f && gg.y == 0.f))) continue;
      float f[3];
      uint32_t g[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float pos = __fadd_rn(__fmul_rn(xc[a], s), 0.5f);
        const float fl = floorf(pos);
        g[a] = (uint32_t)(int)fl;
        f[a] = pos - fl;
      }
      const uint32_t r000 = g[0] + g[1] * res + g[2] * res2;
      if (r000 != cur) {
        if (in_range) flush();
        cur = r000;
        // a cell none of whose rows can lie in this chunk (no wrap-around: the last row is below the level's size) is skipped whole
        const uint32_t last = r000 + 1u + res + res2;
        in_range = last >= size || (last >= base && r000 < base + rows);
      }
      if (!in_range) continue;
#pragma unroll
      for (uint32_t c = 0; c < 8; ++c) {
        const float w = ((c & 1u) ? f[0] : 1.0f - f[0]) * ((c & 2u) ? f[1] : 1.0f - f[1]) * ((c & 4u) ? f[2] : 1.0f - f[2]);
        acc[c][0] += (double)(w * gg.x);
        acc[c][1] += (double)(w * gg.y);
      }
    }
    if (in_range) flush();
  };
  // Row replay: the level's 8 corner rows of every point were stored by dgrid_transpose_kernel (eight 16-bit values): a visit
  // is a 16-byte load, eight extract-and-compare steps and the three fractions for the weights -- no floor / convert / hash.
  auto sweep_replay = [&](const uint4* __restrict__ rl) {
    uint4 rn = make_uint4(0u, 0u, 0u, 0u);
    {
      const uint32_t p = point_of(0);
      if (n_it > 0 && p < p_hi) rn = rl[p];
    }
    for (uint32_t it = 0; it < n_it; ++it) {
      const float2 gg = gg_n;
      const float xc[3] = {xn[0], xn[1], xn[2]};
      const uint4 rr = rn;
      const bool live = point_of(it) < p_hi;
      gg_n = make_float2(0.f, 0.f);
      if (it + 1 < n_it) {
        const uint32_t pn = point_of(it + 1);
        if (pn < p_hi) {
          gg_n = dgl[pn];
          rn = rl[pn];
          xn[0] = xin[(size_t)pn * 3];
          xn[1] = xin[(size_t)pn * 3 + 1];
          xn[2] = xin[(size_t)pn * 3 + 2];
        }
      }
      const bool work = live && !(gg.x == 0.f && gg.y == 0.f);
      const uint32_t rows_eff = work ? rows : 0u;
      const uint32_t w4[4] = {rr.x, rr.y, rr.z, rr.w};
      uint32_t hit_mask = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t local = ((c & 1) ? (w4[c >> 1] >> 16) : (w4[c >> 1] & 0xffffu)) - base;
        hit_mask |= (local < rows_eff ? 1u : 0u) << c;
      }
      if (!__any(hit_mask != 0)) continue;       // (uniform) most visits of a wave whose points miss this chunk end here
      float f[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float pos = __fadd_rn(__fmul_rn(xc[a], s), 0.5f);
        f[a] = pos - floorf(pos);
      }
      while (__any(hit_mask != 0)) {
        if (hit_mask) {
          const uint32_t c = (uint32_t)__ffs((int)hit_mask) - 1u;
          hit_mask &= hit_mask - 1u;
          const uint32_t wsel = (c >> 1) == 0u ? rr.x : ((c >> 1) == 1u ? rr.y : ((c >> 1) == 2u ? rr.z : rr.w));
          const uint32_t local = ((c & 1u) ? (wsel >> 16) : (wsel & 0xffffu)) - base;
          const float w = ((c & 1u) ? f[0] : 1.0f - f[0]) * ((c & 2u) ? f[1] : 1.0f - f[1]) * ((c & 4u) ? f[2] : 1.0f - f[2]);
          atomicAdd(dbins + 2 * local, (double)(w * gg.x));
          atomicAdd(dbins + 2 * local + 1, (double)(w * gg.y));
        }
      }
    }
  };
  if (hashed && rows16 && rp.slot[l] >= 0) sweep_replay(rows16 + (size_t)rp.slot[l] * P);
  else if (hashed) sweep(std::true_type{});
  else if (strided && plan.dense_runs) sweep_dense_runs();
  else sweep(std::false_type{});
  __syncthreads();
  float* out = d_table + 2 * ((size_t)lv.offset[l] + base);
  for (uint32_t i = threadIdx.x; i < rows * 2; i += blockDim.x) {
    const double v = dbins[i];
    if (v != 0.0) atomicAdd(out + i, (float)v);
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


// ------------------------------------------------------------------------------------------------------------------
// Hashed levels of LARGE tables, partition form.  The binned kernel above visits every (point, level) once per 8192-row
// chunk of the level: 8 visits at T = 2^16, 128 at T = 2^20 (11 ms of an 18 ms iteration at 1 M points).  Here every
// (point, level) is hashed ONCE: pass 1 partitions its 8 corner contributions {local row, w*g0, w*g1} by chunk through
// LDS and appends each chunk's run to that chunk's global queue (structure of arrays: coalesced both ways); pass 2
// streams a queue slice into the same 64-bit fixed-point LDS bins with every lane busy.  The queues cost 24 bytes of
// HBM traffic per corner, which at 8 chunks per level is MORE than the redundant hashing it removes (measured at
// T = 2^16, 262 k points: 141 + 122 us against ~190 us for the same levels binned) -- so the host picks this form from
// 16 chunks per level up.  Queues are sized for the uniform-hash expectation plus slack; what does not fit goes straight
// to d_table with float atomics (correct, merely slow: only reachable with adversarially clustered points).
constexpr uint32_t PART_MAX_CHUNKS = 256;      // chunks per level (8192-row chunks: T <= 2^21; the lists' 4096-row chunks: T <= 2^20)
constexpr uint32_t PART_THREADS = 256;         // points per pass-1 workgroup
constexpr uint32_t PART_ENTRIES = PART_THREADS * 8;

struct PartPlan {
  uint32_t n;                                  // levels handled by the partition form
  uint32_t level[DNS_MAX_LEVELS];              // their level indices
  uint32_t chunks[DNS_MAX_LEVELS];             // 8192-row chunks of the level
  uint32_t qoff[DNS_MAX_LEVELS + 1];           // first queue of the level (prefix sum of chunks)
  uint32_t cap[DNS_MAX_LEVELS];                // entries per queue of the level
  uint64_t qbase[DNS_MAX_LEVELS];              // float offset of the level's first queue
  uint32_t chunk_shift;                        // log2(rows per chunk) = 13
  uint32_t slices;                             // pass-2 workgroups per queue
};

__global__ __launch_bounds__(PART_THREADS) void hashgrid_bwd_partition_kernel(const float* __restrict__ xin, uint32_t P,
                                                                               GridLevels lv, PartPlan pp,
                                                                               const float2* __restrict__ dg_t,
                                                                               uint32_t* __restrict__ qcount,
                                                                               float* __restrict__ queues,
                                                                               float* __restrict__ d_table) {
  __shared__ uint32_t cnt[PART_MAX_CHUNKS], base[PART_MAX_CHUNKS + 1], gofs[PART_MAX_CHUNKS];
  __shared__ uint32_t st_row[PART_ENTRIES];
  __shared__ float st_v0[PART_ENTRIES], st_v1[PART_ENTRIES];
  __shared__ uint8_t st_chunk[PART_ENTRIES];
  const uint32_t p = blockIdx.x * PART_THREADS + threadIdx.x;
  const bool live = p < P;
  float xc[3] = {0.f, 0.f, 0.f};
  if (live) {
    xc[0] = xin[(size_t)p * 3];
    xc[1] = xin[(size_t)p * 3 + 1];
    xc[2] = xin[(size_t)p * 3 + 2];
  }
  const uint32_t row_mask = (1u << pp.chunk_shift) - 1u;
  for (uint32_t li = 0; li < pp.n; ++li) {
    const uint32_t l = pp.level[li], C = pp.chunks[li], cap = pp.cap[li];
    const float2 gg = live ? dg_t[(size_t)l * P + p] : make_float2(0.f, 0.f);
    const bool work = live && !(gg.x == 0.f && gg.y == 0.f);
    const float s = lv.scale[l];
    const uint32_t size = lv.size[l], res = lv.resolution[l];
    const bool hashed = lv.hashed[l] != 0;       // uniform
    float f[3];
    uint32_t g[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float pos = __fadd_rn(__fmul_rn(xc[a], s), 0.5f);
      const float fl = floorf(pos);
      g[a] = (uint32_t)(int)fl;
      f[a] = pos - fl;
    }
    // per-axis terms of the row index: hashed  x ^ y*P1 ^ z*P2 (size = 2^T),  dense  (x + y*res + z*res^2) mod size
    const uint32_t ax0 = g[0], ax1 = g[0] + 1u;
    const uint32_t ay0 = hashed ? g[1] * 2654435761u : g[1] * res, ay1 = ay0 + (hashed ? 2654435761u : res);
    const uint32_t az0 = hashed ? g[2] * 805459861u : g[2] * res * res, az1 = az0 + (hashed ? 805459861u : res * res);
    for (uint32_t i = threadIdx.x; i < C; i += PART_THREADS) cnt[i] = 0;
    __syncthreads();
    uint32_t rows8[8], slot8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint32_t x = (c & 1) ? ax1 : ax0, y = (c & 2) ? ay1 : ay0, z = (c & 4) ? az1 : az0;
      uint32_t r;
      if (hashed) {
        r = (x ^ y ^ z) & (size - 1u);
      } else {
        r = x + y + z;
        if (r >= size) r %= size;
      }
      rows8[c] = r;
      slot8[c] = work ? atomicAdd(&cnt[r >> pp.chunk_shift], 1u) : 0u;
    }
    __syncthreads();
    // exclusive scan of the <= 128 chunk counts by the first wave, and one global reservation per chunk
    if (threadIdx.x < 64u) {
      uint32_t run = 0;
      for (uint32_t c0 = 0; c0 < C; c0 += 64u) {
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t v = c < C ? cnt[c] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t t = __shfl_up(incl, o);
          if ((int)threadIdx.x >= o) incl += t;
        }
        if (c < C) {
          base[c] = run + incl - v;
          gofs[c] = v ? atomicAdd(&qcount[pp.qoff[li] + c], v) : 0u;
        }
        run += __shfl(incl, 63);
      }
      if (threadIdx.x == 0) base[C] = run;
    }
    __syncthreads();
    if (work) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float w = ((c & 1) ? f[0] : 1.0f - f[0]) * ((c & 2) ? f[1] : 1.0f - f[1]) * ((c & 4) ? f[2] : 1.0f - f[2]);
        const uint32_t ch = rows8[c] >> pp.chunk_shift;
        const uint32_t at = base[ch] + slot8[c];
        st_row[at] = rows8[c] & row_mask;
        st_v0[at] = w * gg.x;
        st_v1[at] = w * gg.y;
        st_chunk[at] = (uint8_t)ch;
      }
    }
    __syncthreads();
    const uint32_t total = base[C];
    for (uint32_t i = threadIdx.x; i < total; i += PART_THREADS) {
      const uint32_t ch = st_chunk[i];
      const uint32_t at = gofs[ch] + (i - base[ch]);
      if (at < cap) {
        float* q = queues + pp.qbase[li] + (size_t)ch * cap * 3u;
        reinterpret_cast<uint32_t*>(q)[at] = st_row[i];
        q[cap + at] = st_v0[i];
        q[2u * (size_t)cap + at] = st_v1[i];
      } else {                                     // queue full (clustered points on a dense level): straight to the table
        float* t = d_table + 2 * ((size_t)lv.offset[l] + ((size_t)ch << pp.chunk_shift) + st_row[i]);
        atomicAdd(t, st_v0[i]);
        atomicAdd(t + 1, st_v1[i]);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void hashgrid_bwd_queue_kernel(GridLevels lv, PartPlan pp, const uint32_t* __restrict__ qcount,
                                                                   const float* __restrict__ queues,
                                                                   const uint32_t* __restrict__ gmax,
                                                                   float* __restrict__ d_table) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long bins[];
  const float mx = __uint_as_float(*gmax);
  const bool poisoned = gmax[1] != 0u;           // NaN / Inf upstream: NaN into the queue's rows (see the binned kernel)
  if (!(mx > 0.f) && !poisoned) return;          // all-zero upstream gradient (uniform exit)
  int ex;
  (void)frexpf(mx, &ex);
  const float scale = ldexpf(1.0f, 40 - ex);     // as in the binned kernel: |w*g| * scale < 2^40
  const double inv_scale = (double)ldexpf(1.0f, ex - 40);
  const uint32_t qi = blockIdx.x / pp.slices, slice = blockIdx.x % pp.slices;   // queue = (level, chunk)
  uint32_t li = 0;
  while (li + 1 < pp.n && qi >= pp.qoff[li + 1]) ++li;
  const uint32_t ch = qi - pp.qoff[li], l = pp.level[li], cap = pp.cap[li];
  const uint32_t row0 = ch << pp.chunk_shift;
  const uint32_t rows = min(1u << pp.chunk_shift, lv.size[l] - row0);
  if (poisoned) {                                // uniform
    float* out = d_table + 2 * ((size_t)lv.offset[l] + row0);
    for (uint32_t i = threadIdx.x; i < rows * 2; i += blockDim.x) out[i] = __uint_as_float(0x7fc00000u);
    return;
  }
  const uint32_t n = min(qcount[qi], cap);
  const uint32_t lo = (uint32_t)(((uint64_t)n * slice) / pp.slices), hi = (uint32_t)(((uint64_t)n * (slice + 1)) / pp.slices);
  if (lo >= hi) return;                          // uniform per workgroup
  for (uint32_t i = threadIdx.x; i < rows * 2; i += blockDim.x) bins[i] = 0ull;
  __syncthreads();
  const float* q = queues + pp.qbase[li] + (size_t)ch * cap * 3u;
  const uint32_t* qr = reinterpret_cast<const uint32_t*>(q);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const uint32_t r = qr[i];
    const float v0 = q[cap + i], v1 = q[2u * (size_t)cap + i];
    atomicAdd(bins + 2 * r, (unsigned long long)fixed_rn(v0 * scale));
    atomicAdd(bins + 2 * r + 1, (unsigned long long)fixed_rn(v1 * scale));
  }
  __syncthreads();
  float* out = d_table + 2 * ((size_t)lv.offset[l] + row0);
  for (uint32_t i = threadIdx.x; i < rows * 2; i += blockDim.x) {
    const long long v = (long long)bins[i];
    if (v != 0) atomicAdd(out + i, (float)((double)v * inv_scale));
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Multi-chunk levels, PAIR-LIST form (round 4, DNS_SCATTER_LISTS).  The binned kernel's vector work is its chunk visits (every
// point-level is hashed and tested once per chunk of its level, ~175 vector instructions per 64-point visit of which 32 lanes'
// worth is useful), the queue form's cost is 24 bytes of traffic per corner.  Here pass 1 hashes every point-level once and
// appends, per x-PAIR of corners (x, x + 1 at the same y, z: one chunk on a hashed level), ONE 32-bit word {point, pair} to the
// list of the pair's chunk -- 16 bytes per point-level instead of 192 --, and pass 2 gives every lane one list entry: it fetches
// the entry's point (12 + 8 bytes, consecutive entries are nearly consecutive points), forms the pair's two rows and weights
// and adds into the chunk's float64 LDS bins -- no tests, no idle lanes.  A pair whose two rows straddle a chunk boundary (dense
// levels only) is entered in both chunks; the consumer range-checks each corner.  A full list (clustered points on a dense
// level) sends the pair straight to d_table with float atomics.
constexpr uint32_t LIST_THREADS = 256;           // points per pass-1 workgroup
constexpr uint32_t LIST_TILES = 8;               // 256-point tiles per pass-1 workgroup (<= 32: one bit each; DNS_LIST_TILES)

struct ListPlan {
  uint32_t n;                                  // levels handled by the pair-list form
  uint32_t level[DNS_MAX_LEVELS];
  uint32_t chunks[DNS_MAX_LEVELS];
  uint32_t qoff[DNS_MAX_LEVELS + 1];           // first list of the level (prefix sum of chunks)
  uint32_t cap[DNS_MAX_LEVELS];                // entries per list of the level
  uint64_t qbase[DNS_MAX_LEVELS];              // word offset of the level's first list
  uint32_t chunk_shift;                        // log2(rows per chunk) = 13
  uint32_t slices;                             // pass-2 workgroups per list (static slicing: hashed levels only)
  uint32_t tiles;                              // 256-point tiles per pass-1 workgroup
  // Dense levels (round 4, second half): their lists fill by where the rays are, so (a) they are sized EXACTLY -- a counting
  // sweep, a scan kernel that places each chunk's list inside the level's region (at most 8 entries per point), a writing sweep --
  // and (b) pass 2 cuts every list into jobs of `target` entries from the actual counts (`balanced`: the scan kernel's job
  // prefix, searched in LDS) instead of a fixed number of slices per list.
  uint32_t dense[DNS_MAX_LEVELS];              // 1: exact-size lists; qbase = the level's region, cap = its size (8 P)
  uint32_t n_dense;
  uint32_t dense_idx[DNS_MAX_LEVELS];          // list-level indices of the dense levels (grid.y of the writing sweep)
  uint32_t balanced;                           // 1: jobs from the scan kernel
  uint32_t target;                             // entries per balanced job
  uint32_t max_jobs;                           // upper bound of the balanced job count (grid of pass 2)
};

// the two rows (level-relative) and weights of x-pair c (bit 0: y + 1, bit 1: z + 1) of a point
struct PairRows {
  uint32_t l0, l1;
  float w0, w1;
};
__device__ __forceinline__ void pair_cell(const float xc[3], float s, uint32_t g[3], float f[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float pos = __fadd_rn(__fmul_rn(xc[a], s), 0.5f);
    const float fl = floorf(pos);
    g[a] = (uint32_t)(int)fl;
    f[a] = pos - fl;
  }
}
__device__ __forceinline__ void pair_rows_only(const uint32_t g[3], uint32_t c, bool hashed, uint32_t res, uint32_t size,
                                               uint32_t& l0, uint32_t& l1) {
  const uint32_t gy = g[1] + (c & 1u), gz = g[2] + ((c >> 1) & 1u);
  if (hashed) {                                  // uniform per level
    const uint32_t h = (gy * 2654435761u) ^ (gz * 805459861u);
    l0 = (g[0] ^ h) & (size - 1u);
    l1 = ((g[0] + 1u) ^ h) & (size - 1u);
  } else {
    uint32_t r = g[0] + gy * res + gz * res * res;
    if (r >= size) r %= size;
    l0 = r;
    r = g[0] + 1u + gy * res + gz * res * res;
    if (r >= size) r %= size;
    l1 = r;
  }
}
__device__ __forceinline__ PairRows pair_rows(const float xc[3], float s, uint32_t c, bool hashed, uint32_t res, uint32_t size) {
  uint32_t g[3];
  float f[3];
  pair_cell(xc, s, g, f);
  PairRows r;
  pair_rows_only(g, c, hashed, res, size, r.l0, r.l1);
  const float wyz = ((c & 1u) ? f[1] : 1.0f - f[1]) * ((c & 2u) ? f[2] : 1.0f - f[2]);
  r.w0 = (1.0f - f[0]) * wyz;
  r.w1 = f[0] * wyz;
  return r;
}

__global__ __launch_bounds__(LIST_THREADS) void hashgrid_bwd_pairlist_kernel(const float* __restrict__ xin, uint32_t P,
                                                                              GridLevels lv, ListPlan lp,
                                                                              const float2* __restrict__ dg_t,
                                                                              const uint32_t* __restrict__ gmax,
                                                                              uint32_t* __restrict__ qcount,
                                                                              uint32_t* __restrict__ lists,
                                                                              float* __restrict__ d_table, uint32_t mode,
                                                                              const uint32_t* __restrict__ lbase) {
  // mode 0: every list level; a hashed level is counted, reserved and written here, a DENSE level is only counted (qcount = the
  // lists' sizes).  mode 1: the dense levels' writing sweep (blockIdx.y indexes lp.dense_idx; qcount = the lists' cursors, lbase =
  // where the scan kernel placed each list inside its level's region).
  // A workgroup takes LIST_TILES x 256 consecutive points of one level.  Sweep A counts its entries per chunk in LDS, the first
  // wave then reserves one run per chunk in the global lists -- ONE atomic per chunk and 2048 points: a reservation per 256 points
  // put 4096 same-address atomics per list behind each other and took longer than everything else --, sweep B forms the chunks
  // again (two multiplies and a few xors per pair) and writes every entry at its run's next free slot.
  __shared__ uint32_t cnt[PART_MAX_CHUNKS], gofs[PART_MAX_CHUNKS];
  if (!(__uint_as_float(*gmax) > 0.f) || gmax[1] != 0u) return;   // all-zero or poisoned upstream gradient: pass 2 handles both
  const uint32_t li = mode ? lp.dense_idx[blockIdx.y] : blockIdx.y;
  const uint32_t l = lp.level[li], C = lp.chunks[li], cap = lp.cap[li];
  const bool exact = lp.dense[li] != 0;          // uniform
  const float s = lv.scale[l];
  const uint32_t size = lv.size[l], res = lv.resolution[l];
  const bool hashed = lv.hashed[l] != 0;         // uniform
  const float2* __restrict__ dgl = dg_t + (size_t)l * P;
  const uint32_t p_base = blockIdx.x * (LIST_THREADS * lp.tiles) + threadIdx.x;
  for (uint32_t i = threadIdx.x; i < C; i += LIST_THREADS) cnt[i] = 0;
  __syncthreads();
  uint32_t work_bits = 0;                        // bit t: the point of tile t carries a gradient
#pragma unroll 2
  for (uint32_t t = 0; t < lp.tiles; ++t) {
    const uint32_t p = p_base + t * LIST_THREADS;
    if (p >= P) break;
    const float2 gg = dgl[p];
    if (gg.x == 0.f && gg.y == 0.f) continue;
    work_bits |= 1u << t;
    const float xc[3] = {xin[(size_t)p * 3], xin[(size_t)p * 3 + 1], xin[(size_t)p * 3 + 2]};
    uint32_t g[3];
    float f[3];
    pair_cell(xc, s, g, f);
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
      uint32_t l0, l1;
      pair_rows_only(g, c, hashed, res, size, l0, l1);
      const uint32_t c0 = l0 >> lp.chunk_shift, c1 = l1 >> lp.chunk_shift;
      atomicAdd(&cnt[c0], 1u);
      if (c1 != c0) atomicAdd(&cnt[c1], 1u);
    }
  }
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < C; c += LIST_THREADS) {
    const uint32_t v = cnt[c];
    gofs[c] = v ? atomicAdd(&qcount[lp.qoff[li] + c], v) : 0u;
    if (mode) gofs[c] += lbase[lp.qoff[li] + c];  // exact lists: slot inside the level's region
    cnt[c] = 0;                                  // now the run's next free slot
  }
  if (exact && !mode) return;                    // (uniform) dense level, counting launch: nothing is written yet
  __syncthreads();
  uint32_t* __restrict__ ql = lists + lp.qbase[li];
  float* __restrict__ tl = d_table + 2 * (size_t)lv.offset[l];
#pragma unroll 2
  for (uint32_t t = 0; t < lp.tiles; ++t) {
    if (!((work_bits >> t) & 1u)) continue;
    const uint32_t p = p_base + t * LIST_THREADS;
    const float xc[3] = {xin[(size_t)p * 3], xin[(size_t)p * 3 + 1], xin[(size_t)p * 3 + 2]};
    uint32_t g[3];
    float f[3];
    pair_cell(xc, s, g, f);
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
      uint32_t l0, l1;
      pair_rows_only(g, c, hashed, res, size, l0, l1);
      const uint32_t c0 = l0 >> lp.chunk_shift, c1 = l1 >> lp.chunk_shift;
      const uint32_t at0 = gofs[c0] + atomicAdd(&cnt[c0], 1u);
      const uint32_t at1 = c1 != c0 ? gofs[c1] + atomicAdd(&cnt[c1], 1u) : 0u;
      if (exact) {                               // every entry has its place
        ql[at0] = (p << 2) | c;
        if (c1 != c0) ql[at1] = (p << 2) | c;
        continue;
      }
      const bool over0 = at0 >= cap, over1 = c1 != c0 && at1 >= cap;
      if (!over0) ql[(size_t)c0 * cap + at0] = (p << 2) | c;
      if (c1 != c0 && !over1) ql[(size_t)c1 * cap + at1] = (p << 2) | c;
      if (over0 || over1) {                      // list full: the pair's corners in that chunk go straight to the table
        const float2 gg = dgl[p];
        const float wyz = ((c & 1u) ? f[1] : 1.0f - f[1]) * ((c & 2u) ? f[2] : 1.0f - f[2]);
        const float w0 = (1.0f - f[0]) * wyz, w1 = f[0] * wyz;
        if (over0) {
          atomicAdd(tl + 2 * (size_t)l0, w0 * gg.x);
          atomicAdd(tl + 2 * (size_t)l0 + 1, w0 * gg.y);
        }
        if (c1 == c0 ? over0 : over1) {
          atomicAdd(tl + 2 * (size_t)l1, w1 * gg.x);
          atomicAdd(tl + 2 * (size_t)l1 + 1, w1 * gg.y);
        }
      }
    }
  }
}

// Exclusive prefix sums over the lists (one workgroup, <= 8192 lists): where each dense list starts inside its level's region,
// and how many jobs of `target` entries precede each list (jobstart[n_lists] = all of them).
__global__ __launch_bounds__(1024) void hashgrid_bwd_pairscan_kernel(ListPlan lp, const uint32_t* __restrict__ qcount,
                                                                      uint32_t* __restrict__ lbase, uint32_t* __restrict__ jobstart) {
  __shared__ uint32_t s_n[8192], s_wave[16];
  // the plan's per-level arrays in LDS (a per-thread index into the kernel-argument struct would go through scratch memory), the
  // level of every list in one byte, the counts read once
  __shared__ uint32_t s_qoff[DNS_MAX_LEVELS + 1], s_cap[DNS_MAX_LEVELS], s_dense[DNS_MAX_LEVELS];
  __shared__ uint8_t s_lvl[8192];
  const uint32_t n_lists = lp.qoff[lp.n];
  if (threadIdx.x <= DNS_MAX_LEVELS) s_qoff[threadIdx.x] = lp.qoff[threadIdx.x < DNS_MAX_LEVELS ? threadIdx.x : DNS_MAX_LEVELS];
  if (threadIdx.x < DNS_MAX_LEVELS) {
    s_cap[threadIdx.x] = lp.cap[threadIdx.x];
    s_dense[threadIdx.x] = lp.dense[threadIdx.x];
  }
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < n_lists; q += blockDim.x) {
    uint32_t li = 0;
    while (li + 1 < lp.n && q >= s_qoff[li + 1]) ++li;
    s_lvl[q] = (uint8_t)li;
    s_n[q] = s_dense[li] ? qcount[q] : min(qcount[q], s_cap[li]);
  }
  __syncthreads();
  constexpr uint32_t PER = 8;                    // lists per thread
  const uint32_t t0 = threadIdx.x * PER;
  auto level_of = [&](uint32_t q) { return (uint32_t)s_lvl[q]; };
  uint32_t cnt[PER];
#pragma unroll
  for (uint32_t k = 0; k < PER; ++k) cnt[k] = t0 + k < n_lists ? s_n[t0 + k] : 0u;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {         // pass 0: entries of dense lists (placement), pass 1: jobs of every list
    uint32_t v[PER], run = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      const uint32_t q = t0 + k;
      uint32_t x = 0;
      if (q < n_lists) {
        const uint32_t n = cnt[k];
        x = pass == 0 ? (s_dense[level_of(q)] ? n : 0u) : (n + lp.target - 1u) / lp.target;
      }
      v[k] = run;
      run += x;
    }
    uint32_t incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if ((int)(threadIdx.x & 63u) >= o) incl += t;
    }
    __syncthreads();
    if ((threadIdx.x & 63u) == 63u) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += s_wave[w];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k)
      if (t0 + k < n_lists) s_n[t0 + k] = base + v[k];
    if (threadIdx.x == 1023u) s_n[8191] = base + run;     // grand total (n_lists < 8192 is checked on the host)
    __syncthreads();
    if (pass == 0) {
#pragma unroll
      for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t q = t0 + k;
        if (q < n_lists) lbase[q] = s_n[q] - s_n[s_qoff[level_of(q)]];   // restart at every level
      }
    } else {
#pragma unroll
      for (uint32_t k = 0; k < PER; ++k)
        if (t0 + k < n_lists) jobstart[t0 + k] = s_n[t0 + k];
      if (threadIdx.x == 0) jobstart[n_lists] = s_n[8191];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void hashgrid_bwd_pairbins_kernel(const float* __restrict__ xin, uint32_t P, GridLevels lv,
                                                                      ListPlan lp, const float2* __restrict__ dg_t,
                                                                      const uint32_t* __restrict__ qcount,
                                                                      const uint32_t* __restrict__ lists,
                                                                      const uint32_t* __restrict__ gmax,
                                                                      float* __restrict__ d_table,
                                                                      const uint32_t* __restrict__ lbase,
                                                                      const uint32_t* __restrict__ jobstart) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long bins[];
  const float mx = __uint_as_float(*gmax);
  const bool poisoned = gmax[1] != 0u;           // NaN / Inf upstream: NaN into the list's rows (see the binned kernel)
  if (!(mx > 0.f) && !poisoned) return;          // all-zero upstream gradient (uniform exit)
  // 64-bit FIXED-POINT bins (|w g| * 2^(40 - ex) < 2^40, rounded to nearest: exact, order-independent sums with 2^24 entries of
  // headroom): this kernel is bound by the LDS atomic unit, not by its vector instructions, and ds_add_u64 issues at 11.9 cycles
  // per wave instruction against 21.6 for ds_add_f64 -- the conversion that lost in the (vector-bound) sweep form pays here
  int ex;
  (void)frexpf(mx, &ex);
  //  gradients below 2^-87 would push the scale 2^(40 - ex) past fp32's range (inf, then NaN into the bins): the
  // exponent is held at -80 -- such values then lose low bits below 2^-120 of themselves, i.e. nothing fp32 could represent
  // in the sum anyway
  ex = max(ex, -80);
  const float scale = ldexpf(1.0f, 40 - ex);
  const double inv_scale = (double)ldexpf(1.0f, ex - 40);
  uint32_t qi, slice;                            // list = (level, chunk), and which part of it
  if (lp.balanced && !poisoned) {
    // job -> list: the job prefix (n_lists + 1 words, behind the bins in LDS) is searched for the last list that starts at or
    // before this job
    uint32_t* js = reinterpret_cast<uint32_t*>(bins + ((size_t)2 << lp.chunk_shift));
    const uint32_t n_lists = lp.qoff[lp.n];
    for (uint32_t i = threadIdx.x; i <= n_lists; i += blockDim.x) js[i] = jobstart[i];
    __syncthreads();
    if (blockIdx.x >= js[n_lists]) return;       // (uniform) past the last job of this launch
    uint32_t lo_q = 0, hi_q = n_lists;           // invariant: js[lo_q] <= job < js[hi_q]
    while (hi_q - lo_q > 1u) {
      const uint32_t mid = (lo_q + hi_q) >> 1;
      if (js[mid] <= blockIdx.x) lo_q = mid; else hi_q = mid;
    }
    qi = lo_q;
    slice = blockIdx.x - js[qi];
    __syncthreads();                             // js lives in LDS the bins do not use; nothing else to order
  } else if (lp.balanced) {                      // poisoned: one job per list writes the NaNs
    qi = blockIdx.x;
    slice = 0;
    if (qi >= lp.qoff[lp.n]) return;
  } else {
    qi = blockIdx.x / lp.slices;
    slice = blockIdx.x % lp.slices;
  }
  uint32_t li = 0;
  while (li + 1 < lp.n && qi >= lp.qoff[li + 1]) ++li;
  const uint32_t ch = qi - lp.qoff[li], l = lp.level[li], cap = lp.cap[li];
  const uint32_t row0 = ch << lp.chunk_shift;
  const uint32_t size = lv.size[l], res = lv.resolution[l];
  const uint32_t rows = min(1u << lp.chunk_shift, size - row0);
  if (poisoned) {                                // uniform
    float* out = d_table + 2 * ((size_t)lv.offset[l] + row0);
    if (slice == 0)
      for (uint32_t i = threadIdx.x; i < rows * 2; i += blockDim.x) out[i] = __uint_as_float(0x7fc00000u);
    return;
  }
  const uint32_t n = lp.dense[li] ? qcount[qi] : min(qcount[qi], cap);
  uint32_t lo, hi;
  if (lp.balanced) {
    lo = slice * lp.target;
    hi = min(n, lo + lp.target);
  } else {
    lo = (uint32_t)(((uint64_t)n * slice) / lp.slices);
    hi = (uint32_t)(((uint64_t)n * (slice + 1)) / lp.slices);
  }
  if (lo >= hi) return;                          // uniform per workgroup
  const uint32_t* __restrict__ q = lists + lp.qbase[li] + (lp.dense[li] ? (size_t)lbase[qi] : (size_t)ch * cap);
  const float2* __restrict__ dgl = dg_t + (size_t)l * P;
  const float s = lv.scale[l];
  const bool hashed = lv.hashed[l] != 0;         // uniform
  // The loop is a chain of dependent memory round trips (entry -> point -> bins) on few waves per SIMD (one 128-KB workgroup per
  // CU): every thread keeps U entries of a trip in flight together, two stages deep -- the entries of trip t + 2 and the points of
  // trip t + 1 are requested while trip t computes.
  constexpr uint32_t U = 4;
  // Lane <-> entry.  A list is in point order; on a DENSE level consecutive samples of a ray share a cell, hence a pair's two rows,
  // and with consecutive entries on consecutive lanes a wave's atomics pile several deep on one address: there lane j of wave w
  // takes entry j * n_waves + w of every round of blockDim.x entries (neighbouring lanes n_waves entries apart; the workgroup's
  // waves share the lines they read).  Hashed levels keep the coalesced order (measured 20 % faster there).
  const uint32_t nt = blockDim.x;
  const uint32_t i0 = lo + (hashed ? threadIdx.x : (threadIdx.x & 63u) * (nt >> 6) + (threadIdx.x >> 6));
  uint32_t e_n[U], e_nn[U];
  float xn[U][3];
  float2 gn[U];
#pragma unroll
  for (uint32_t u = 0; u < U; ++u) e_n[u] = (i0 + u * nt < hi) ? q[i0 + u * nt] : 0u;
#pragma unroll
  for (uint32_t u = 0; u < U; ++u) e_nn[u] = (i0 + (U + u) * nt < hi) ? q[i0 + (U + u) * nt] : 0u;
#pragma unroll
  for (uint32_t u = 0; u < U; ++u) {
    const uint32_t pn = e_n[u] >> 2;             // 0 past the end: a valid point, never used
    xn[u][0] = xin[(size_t)pn * 3];
    xn[u][1] = xin[(size_t)pn * 3 + 1];
    xn[u][2] = xin[(size_t)pn * 3 + 2];
    gn[u] = dgl[pn];
  }
  // the bins are cleared while the first entries and points are on their way
  for (uint32_t i = threadIdx.x; i < rows * 2; i += blockDim.x) bins[i] = 0ull;
  __syncthreads();
  for (uint32_t i = i0; i < hi; i += U * nt) {
    uint32_t e[U];
    float xc[U][3];
    float2 gg[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      e[u] = e_n[u];
      xc[u][0] = xn[u][0];
      xc[u][1] = xn[u][1];
      xc[u][2] = xn[u][2];
      gg[u] = gn[u];
      e_n[u] = e_nn[u];
    }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      const uint32_t pn = e_n[u] >> 2;
      xn[u][0] = xin[(size_t)pn * 3];
      xn[u][1] = xin[(size_t)pn * 3 + 1];
      xn[u][2] = xin[(size_t)pn * 3 + 2];
      gn[u] = dgl[pn];
    }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) e_nn[u] = (i + (2u * U + u) * nt < hi) ? q[i + (2u * U + u) * nt] : 0u;
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      if (i + u * nt >= hi) break;
      const PairRows r = hashed ? pair_rows(xc[u], s, e[u] & 3u, true, res, size) : pair_rows(xc[u], s, e[u] & 3u, false, res, size);
      const uint32_t a0 = r.l0 - row0, a1 = r.l1 - row0;
      if (a0 < rows) {
        atomicAdd(bins + 2 * a0, fixed_rn_f64(r.w0 * gg[u].x * scale));
        atomicAdd(bins + 2 * a0 + 1, fixed_rn_f64(r.w0 * gg[u].y * scale));
      }
      if (a1 < rows) {
        atomicAdd(bins + 2 * a1, fixed_rn_f64(r.w1 * gg[u].x * scale));
        atomicAdd(bins + 2 * a1 + 1, fixed_rn_f64(r.w1 * gg[u].y * scale));
      }
    }
  }
  __syncthreads();
  float* out = d_table + 2 * ((size_t)lv.offset[l] + row0);
  for (uint32_t i = threadIdx.x; i < rows * 2; i += blockDim.x) {
    const long long v = (long long)bins[i];
    if (v != 0) atomicAdd(out + i, (float)((double)v * inv_scale));
  }
}

// dense levels of at least this many 8192-row chunks go through (exact-size) lists, smaller ones through the run-combining sweep
// (DNS_LIST_DENSE_MIN overrides, for measurement; 0x7fffffff = none)
static uint32_t list_dense_min_chunks() {
  static const uint32_t v = [] { const char* e = getenv("DNS_LIST_DENSE_MIN"); const long n = e ? atol(e) : 0; return (uint32_t)(n >= 1 ? n : 6); }();
  return v;
}

// Lists: 4 pairs per point-level spread over the level's chunks.  Hashed levels get the uniform-hash expectation + 1/8 slack,
// dense levels (spatially clustered points) four times the expectation; what does not fit takes the atomics fallback.
static uint32_t list_chunk_shift() {
  static const uint32_t v = [] { const char* e = getenv("DNS_LIST_SHIFT"); const long n = e ? atol(e) : 0; return (uint32_t)(n >= 10 && n <= 13 ? n : 12); }();
  return v;
}

static uint32_t list_threads(uint32_t shift) {
  static const uint32_t v = [] { const char* e = getenv("DNS_LIST_THREADS"); const long n = e ? atol(e) : 0; return (uint32_t)(n >= 64 && n <= 1024 && n % 64 == 0 ? n : 0); }();
  return v ? v : 1024u >> (13u - shift);
}

static bool list_plan(const GridLevels& lv, uint32_t P, uint32_t queue_cap, uint32_t target_jobs, ListPlan& lp) {
  lp.n = 0;
  lp.n_dense = 0;
  for (uint32_t i = 0; i < DNS_MAX_LEVELS; ++i) lp.dense[i] = lp.dense_idx[i] = 0;
  lp.chunk_shift = list_chunk_shift();
  uint32_t queues = 0;
  uint64_t words = 0;
  if (P >= (1u << 30)) return false;             // {point, pair} in 32 bits
  for (uint32_t l = 0; l < lv.n_levels; ++l) {
    const uint32_t chunks = (lv.size[l] + (1u << lp.chunk_shift) - 1u) >> lp.chunk_shift;
    if (lv.size[l] <= 8192u || chunks > PART_MAX_CHUNKS) continue;   // one-chunk levels stay with the sweep
    if (lv.hashed[l] && (lv.size[l] & (lv.size[l] - 1u))) continue;
    const bool dense = !lv.hashed[l];
    if (dense && ((lv.size[l] + 8191u) >> 13) < list_dense_min_chunks()) continue;   // small dense levels: the run-combining sweep
    const uint64_t expect = ((uint64_t)P * 4u + chunks - 1) / chunks;
    uint64_t cap = expect + expect / 8u + 4096u;
    if (cap > (uint64_t)P * 8u) cap = (uint64_t)P * 8u;          // a level emits at most 8 entries per point
    if (queue_cap) cap = queue_cap;                               // caller-chosen capacity (tests: the overflow fallback)
    if (dense) cap = (uint64_t)P * 8u;                            // exact lists: the level's whole region
    cap = (cap + 3u) & ~3ull;
    if (cap > 0x7FFFFFFFull) return false;
    const uint32_t i = lp.n++;
    lp.level[i] = l;
    lp.chunks[i] = chunks;
    lp.qoff[i] = queues;
    lp.cap[i] = (uint32_t)cap;
    lp.qbase[i] = words;
    lp.dense[i] = dense ? 1u : 0u;
    if (dense) lp.dense_idx[lp.n_dense++] = i;
    queues += chunks;
    words += dense ? cap : (uint64_t)chunks * cap;
  }
  lp.qoff[lp.n] = queues;
  for (uint32_t i = lp.n; i < DNS_MAX_LEVELS; ++i) {
    lp.level[i] = 0;
    lp.chunks[i] = 0;
    lp.cap[i] = 0;
    lp.qbase[i] = words;
    lp.qoff[i + 1] = queues;
  }
  if (!lp.n || queues >= 8192u) return false;
  lp.slices = (target_jobs + queues - 1) / queues;
  if (lp.slices < 1) lp.slices = 1;
  lp.balanced = lp.n_dense ? 1u : 0u;
  lp.target = 16u * list_threads(lp.chunk_shift);                 // 16 entries per thread
  uint64_t mj = queues;                                           // every list's last, partial job
  for (uint32_t i = 0; i < lp.n; ++i) mj += (lp.dense[i] ? (uint64_t)lp.cap[i] : (uint64_t)lp.chunks[i] * lp.cap[i]) / lp.target;
  if (mj > 0x7FFFFFFFull) return false;
  lp.max_jobs = (uint32_t)mj;
  static const uint32_t tiles_env = [] { const char* e = getenv("DNS_LIST_TILES"); const long n = e ? atol(e) : 0; return (uint32_t)(n >= 1 && n <= 32 ? n : 0); }();
  lp.tiles = tiles_env ? tiles_env : LIST_TILES;
  return true;
}

static uint64_t list_words(const ListPlan& lp) {
  uint64_t w = 0;
  for (uint32_t i = 0; i < lp.n; ++i) w += lp.dense[i] ? (uint64_t)lp.cap[i] : (uint64_t)lp.chunks[i] * lp.cap[i];
  return w;
}

// Levels with at least PART_MIN_CHUNKS chunks go through the partition form; hashed levels get the uniform-hash
// expectation + 1/8 slack per queue, dense levels (spatially clustered points) twice the expectation.
constexpr uint32_t PART_MIN_CHUNKS = 16;

static bool part_plan(const GridLevels& lv, uint32_t P, uint32_t min_chunks, uint32_t queue_cap, PartPlan& pp,
                      const bool* skip = nullptr) {
  pp.n = 0;
  pp.chunk_shift = 13;
  uint32_t queues = 0;
  uint64_t floats = 0;
  for (uint32_t l = 0; l < lv.n_levels; ++l) {
    const uint32_t chunks = (lv.size[l] + 8191u) >> 13;
    if (chunks < min_chunks || chunks > PART_MAX_CHUNKS || (skip && skip[l])) continue;
    if (lv.hashed[l] && (lv.size[l] & (lv.size[l] - 1u))) continue;
    const uint64_t expect = ((uint64_t)P * 8u + chunks - 1) / chunks;
    uint64_t cap = (lv.hashed[l] ? expect + expect / 8u : 2u * expect) + 4096u;
    if (queue_cap) cap = queue_cap;                               // caller-chosen capacity (tests: the overflow fallback)
    cap = (cap + 3u) & ~3ull;
    if (cap > 0x7FFFFFFFull) return false;
    const uint32_t i = pp.n++;
    pp.level[i] = l;
    pp.chunks[i] = chunks;
    pp.qoff[i] = queues;
    pp.cap[i] = (uint32_t)cap;
    pp.qbase[i] = floats;
    queues += chunks;
    floats += (uint64_t)chunks * cap * 3u;
  }
  pp.qoff[pp.n] = queues;
  for (uint32_t i = pp.n; i < DNS_MAX_LEVELS; ++i) {
    pp.level[i] = 0;
    pp.chunks[i] = 0;
    pp.cap[i] = 0;
    pp.qbase[i] = floats;
    pp.qoff[i + 1] = queues;
  }
  if (!pp.n) return false;
  pp.slices = (1024u + queues - 1) / queues;
  if (pp.slices < 1) pp.slices = 1;
  return true;
}

static uint64_t part_floats(const PartPlan& pp) {
  uint64_t f = 0;
  for (uint32_t i = 0; i < pp.n; ++i) f += (uint64_t)pp.chunks[i] * pp.cap[i] * 3u;
  return f;
}

static uint32_t part_min_chunks(uint32_t flags) {
  const uint32_t form = flags & DNS_SCATTER_MASK;
  if (form == DNS_SCATTER_QUEUES) return 2u;                   // every multi-chunk level through the queues
  if (form == DNS_SCATTER_BINNED) return PART_MAX_CHUNKS + 1u;  // none
  return PART_MIN_CHUNKS;
}

// workgroups of the sweep form (DNS_BIN_JOBS overrides, for measurement): ~1280 when it carries the hashed levels too (above),
// one round of the chip when only the dense levels are left to it (every job zeroes and flushes a whole chunk)
static uint32_t bin_target_jobs(bool lists) {
  static const uint32_t v = [] { const char* e = getenv("DNS_BIN_JOBS"); const long n = e ? atol(e) : 0; return (uint32_t)(n > 0 && n < 65536 ? n : 0); }();
  return v ? v : (lists ? 512u : 1280u);
}

static uint32_t bin_threads() {
  static const uint32_t v = [] { const char* e = getenv("DNS_BIN_THREADS"); const long n = e ? atol(e) : 0; return (uint32_t)(n >= 64 && n <= 1024 && n % 64 == 0 ? n : 1024); }();
  return v;
}

// pass-2 workgroups of the pair-list form (DNS_LIST_JOBS overrides, for measurement)
static uint32_t list_target_jobs() {
  static const uint32_t v = [] {
    const char* e = getenv("DNS_LIST_JOBS");
    const long n = e ? atol(e) : 0;
    return (uint32_t)(n > 0 && n < 65536 ? n : 1536);
  }();
  return v;
}

// Workspace of the table scatter, in floats: [level-major gradient copy | max word, non-finite flag, pad | list counters | queue
// counters + queues of the partition form | lists of the pair-list form | replayed rows (16-byte aligned)]
struct ScatterWs {
  bool part, lists;
  PartPlan pp;
  ListPlan lp;
  bool in_part[DNS_MAX_LEVELS], in_list[DNS_MAX_LEVELS];
  uint64_t gmax, qcount, queues, lcount, lwords, replay, total;
};
static ScatterWs scatter_ws(uint32_t P, const GridLevels& lv, uint32_t flags, uint32_t queue_cap) {
  ScatterWs w = {};
  w.lists = ((flags & DNS_SCATTER_LISTS) || (flags & DNS_SCATTER_MASK) == DNS_SCATTER_AUTO) && list_plan(lv, P, queue_cap, list_target_jobs(), w.lp);
  if (w.lists)
    for (uint32_t i = 0; i < w.lp.n; ++i) w.in_list[w.lp.level[i]] = true;
  w.part = part_plan(lv, P, part_min_chunks(flags), queue_cap, w.pp, w.in_list);
  if (w.part)
    for (uint32_t i = 0; i < w.pp.n; ++i) w.in_part[w.pp.level[i]] = true;
  uint64_t n = (uint64_t)P * lv.n_levels * 2;
  w.gmax = n;
  n += 4;
  w.lcount = n;                                  // directly behind the max words: one fill clears max words, counts and cursors
  if (w.lists) n += (uint64_t)4u * w.lp.qoff[w.lp.n] + 4u;      // [counts | cursors | list starts | job prefix (+1)]
  w.qcount = n;
  if (w.part) n += (uint64_t)DNS_MAX_LEVELS * PART_MAX_CHUNKS;
  w.queues = n;
  if (w.part) n += part_floats(w.pp);
  w.lwords = n;
  if (w.lists) n += list_words(w.lp);
  n = (n + 3u) & ~(uint64_t)3u;
  w.replay = n;
  if (flags & DNS_SCATTER_REPLAY) n += (uint64_t)P * lv.n_levels * 4;
  w.total = n;
  return w;
}

// How many times the tiled encoder uses its LDS tile (encode_fwd_kernel): OneBlob in 1 or 3 phases, the grid in g_ph groups of
// levels.  The tile -- not the 68 registers -- sets this gather kernel's occupancy.  DNS_ENC_PHASES = "pe_ph,g_ph" (measurement knob).
static void encode_tile_phases(uint32_t n_bins, uint32_t n_levels, bool with_pe, uint32_t& pe_ph, uint32_t& g_ph) {
  static const int env_pe = [] { const char* e = getenv("DNS_ENC_PHASES"); return e ? atoi(e) : 0; }();
  static const int env_g = [] { const char* e = getenv("DNS_ENC_PHASES"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 0; }();
  // default 3 + 2 phases (a 17-float tile row, 8.7 KB per workgroup, where one OneBlob phase + one grid phase needs 49 floats, 25 KB:
  // the registers then allow 14 workgroups per CU instead of the tile's 6): 262 144 points along rays 97.1 -> 91.3 us, uniformly
  // random points 155 -> 132 us, the cfg2 step 1.532 -> 1.526 ms (round 5; "3,8": 90.5 / 135.8, "1,2": 95.7 / 153)
  pe_ph = (env_pe == 3 || env_pe == 1) ? (uint32_t)env_pe : 3u;
  g_ph = env_g >= 1 ? (uint32_t)env_g : 2u;
  if (!with_pe) pe_ph = 1u;
  if (n_levels == 0u || n_levels % g_ph != 0u || ((2u * n_levels / g_ph) % 8u) != 0u) g_ph = 1u;   // (flush granularity: 8 columns)
  if (pe_ph == 3u && (n_bins % 8u) != 0u) pe_ph = 1u;
}

static int encode_init_attrs() {
  const int bytes = 8192 * 2 * (int)sizeof(unsigned long long);   // one 8192-row chunk of 64-bit bins: 128 KB
  if (hipFuncSetAttribute((const void*)hashgrid_bwd_binned_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
      hipFuncSetAttribute((const void*)hashgrid_bwd_queue_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
      // (+ the job prefix of balanced dense lists: 4 (n_lists + 1) bytes behind the bins; with DNS_LIST_SHIFT=13 the bins alone
      //  are the whole 128 KB -- ADVICE r4)
      hipFuncSetAttribute((const void*)hashgrid_bwd_pairbins_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) != hipSuccess ||
      hipFuncSetAttribute((const void*)encode_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (3 * 64 + 1) * (int)sizeof(float)) != hipSuccess ||
      hipFuncSetAttribute((const void*)encode_fwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (3 * 64 + 1) * (int)sizeof(float)) != hipSuccess) {
    set_error("dns_init: hipFuncSetAttribute failed for the hash-grid scatter kernels");
    return DNS_E_LAUNCH;
  }
  return DNS_OK;
}
static AttrRegistrar encode_attr_registrar(encode_init_attrs);

}  // namespace dns

using namespace dns;

extern "C" int dns_encode_fwd(const float* in, const double* bound, uint32_t P, uint32_t n_bins, const float* table,
                              const DnsGridMeta* meta, float* x_out, float* pe_out, uint32_t ld_pe, float* grid_out,
                              uint32_t ld_grid, float* dy_dx, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(in != nullptr, "dns_encode_fwd: in is NULL");
  DNS_REQUIRE(!dy_dx || grid_out, "dns_encode_fwd: dy_dx needs the grid encoding");
  DNS_REQUIRE(!dy_dx || (((uintptr_t)dy_dx) & 7u) == 0, "dns_encode_fwd: dy_dx must be 8-byte aligned");
  if (pe_out) DNS_REQUIRE(n_bins >= 1 && n_bins <= 64 && ld_pe >= 3 * n_bins, "dns_encode_fwd: n_bins %u / ld_pe %u", n_bins, ld_pe);
  GridLevels lv = {};
  if (grid_out) {
    DNS_REQUIRE(meta && table, "dns_encode_fwd: grid requested without table/meta");
    DNS_REQUIRE(meta->n_features == 2, "dns_encode_fwd: n_features must be 2");
    DNS_REQUIRE(ld_grid >= 2 * meta->n_levels, "dns_encode_fwd: ld_grid %u too small", ld_grid);
    lv = to_levels(meta);
  }
  const uint32_t blocks = (P + 127) / 128;
  const uint32_t pe_dim = 3 * n_bins, g_dim = grid_out ? 2 * meta->n_levels : 0;
  // (OneBlob alone into rows of any width -- Decoder.merge's input rows -- also leaves through the LDS tile: direct row stores at
  //  a lane stride of a whole row cost 3.4x the bytes in HBM writes)
  const bool tiled = (pe_out && grid_out && ld_pe == pe_dim + g_dim && ld_grid == ld_pe && grid_out == pe_out + pe_dim) ||
                     (pe_out && !grid_out && pe_dim > 0);
  if (tiled) {
    uint32_t pe_ph, g_ph;
    encode_tile_phases(n_bins, grid_out ? meta->n_levels : 0, pe_out != nullptr, pe_ph, g_ph);
    const uint32_t w_pe = pe_out ? pe_dim / pe_ph : 0, w_g = grid_out ? g_dim / g_ph : 0;
    const size_t lds_bytes = (size_t)128 * ((w_pe > w_g ? w_pe : w_g) + 1) * sizeof(float);
    DNS_LAUNCH(encode_fwd_kernel<true>, dim3(blocks), dim3(128), lds_bytes, (hipStream_t)stream, in, make_bound(bound),
                       bound ? 1 : 0, P, n_bins, (const float2*)table, lv, x_out, pe_out, ld_pe, grid_out, ld_grid,
                       (float2*)dy_dx, pe_ph, g_ph);
  } else {
    DNS_LAUNCH(encode_fwd_kernel<false>, dim3(blocks), dim3(128), 0, (hipStream_t)stream, in, make_bound(bound),
                       bound ? 1 : 0, P, n_bins, (const float2*)table, lv, x_out, pe_out, ld_pe, grid_out, ld_grid,
                       (float2*)dy_dx, 1u, 1u);
  }
  return check_launch("dns_encode_fwd");
}

extern "C" int dns_encode_fwd_split(const float* in, const double* bound, uint32_t P, uint32_t n_bins, const float* table,
                                    const DnsGridMeta* meta, float* x_out, float* f32_out, uint32_t ld32, void* xs_out,
                                    uint32_t ldxs, int32_t* xexp, uint32_t flags, float* dy_dx, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE((flags & ~(DNS_SPLIT_HI_ONLY | DNS_SPLIT_PLAIN)) == 0, "dns_encode_fwd_split: unknown flags 0x%x", flags);
  const bool plain = (flags & DNS_SPLIT_PLAIN) != 0;       // half rows: unscaled f16, no exponents, hi plane only
  DNS_REQUIRE(in && table && meta && xs_out && (xexp || plain), "dns_encode_fwd_split: NULL argument");
  DNS_REQUIRE(meta->n_features == 2, "dns_encode_fwd_split: n_features must be 2");
  DNS_REQUIRE(!dy_dx || (((uintptr_t)dy_dx) & 7u) == 0, "dns_encode_fwd_split: dy_dx must be 8-byte aligned");
  const uint32_t pe_dim = 3 * n_bins, g_dim = 2 * meta->n_levels, K = pe_dim + g_dim;
  const bool hi_only = (flags & DNS_SPLIT_HI_ONLY) != 0 || plain;
  DNS_REQUIRE(n_bins >= 1 && (pe_dim % 8) == 0 && (g_dim % 8) == 0 && pe_dim <= 64 && g_dim <= 64,
              "dns_encode_fwd_split: OneBlob / grid widths %u / %u must be multiples of 8 and <= 64", pe_dim, g_dim);
  DNS_REQUIRE((ldxs % 8) == 0 && ldxs >= (hi_only ? K : 2 * K) && (((uintptr_t)xs_out) & 15u) == 0,
              "dns_encode_fwd_split: xs_out must be 16-byte aligned with ldxs %% 8 == 0 and ldxs >= %u", hi_only ? K : 2 * K);
  DNS_REQUIRE(!f32_out || ((ld32 % 4) == 0 && ld32 >= K && (((uintptr_t)f32_out) & 15u) == 0),
              "dns_encode_fwd_split: f32_out must be 16-byte aligned with ld32 %% 4 == 0 and ld32 >= %u", K);
  GridLevels lv = to_levels(meta);
  const uint32_t blocks = (P + 127) / 128;
  const size_t lds_bytes = (size_t)128 * ((pe_dim > g_dim ? pe_dim : g_dim) + 1) * sizeof(float);
  if (plain && !f32_out) {
    // half rows alone: the one-pass tiled encoder with an f16 flush (pe_out / grid_out / their strides in halfs)
    _Float16* xh = reinterpret_cast<_Float16*>(xs_out);
    uint32_t pe_ph, g_ph;
    encode_tile_phases(n_bins, meta->n_levels, true, pe_ph, g_ph);
    // the f16 flush writes HALF as many bytes per row and phase: the grid columns leave in ONE phase (64-byte row pieces; two phases
    // = 32-byte pieces cost 0.46 against 0.41 ms per cfg5_fp16 iteration -- the whole step 3.54 against 3.45 --, nothing at cfg2_fp16)
    static const bool env_phases = getenv("DNS_ENC_PHASES") != nullptr;
    if (!env_phases) g_ph = 1u;
    const uint32_t w_pe = pe_dim / pe_ph, w_g = g_dim / g_ph;
    const size_t lds_plain = (size_t)128 * ((w_pe > w_g ? w_pe : w_g) + 1) * sizeof(float);
    DNS_LAUNCH((encode_fwd_kernel<true, true>), dim3(blocks), dim3(128), lds_plain, (hipStream_t)stream, in, make_bound(bound), bound ? 1 : 0, P,
               n_bins, (const float2*)table, lv, x_out, reinterpret_cast<float*>(xh), ldxs, reinterpret_cast<float*>(xh + pe_dim), ldxs,
               (float2*)dy_dx, pe_ph, g_ph);
    return check_launch("dns_encode_fwd_split");
  }
  DNS_LAUNCH(encode_fwd_split_kernel, dim3(blocks), dim3(128), lds_bytes, (hipStream_t)stream, in, make_bound(bound), bound ? 1 : 0, P,
             n_bins, (const float2*)table, lv, x_out, f32_out, ld32, (uint32_t*)xs_out, ldxs / 2, xexp, (hi_only ? 1 : 0) | (plain ? 2 : 0),
             (float2*)dy_dx);
  return check_launch("dns_encode_fwd_split");
}

extern "C" int dns_encode_bwd(const float* x, const double* bound, uint32_t P, uint32_t n_bins, const float* table,
                              const DnsGridMeta* meta, const float* d_pe, uint32_t ld_dpe, const float* d_grid,
                              uint32_t ld_dgrid, float* d_table, float* d_x, const float* dy_dx, float* ws, uint32_t flags,
                              uint32_t queue_cap, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(x != nullptr, "dns_encode_bwd: x is NULL");
  DNS_REQUIRE(!dy_dx || (((uintptr_t)dy_dx) & 7u) == 0, "dns_encode_bwd: dy_dx must be 8-byte aligned");
  DNS_REQUIRE((flags & ~(DNS_SCATTER_MASK | DNS_SCATTER_REPLAY | DNS_SCATTER_LISTS)) == 0, "dns_encode_bwd: unknown flags 0x%x", flags);
  GridLevels lv = {};
  if (d_grid) {
    DNS_REQUIRE(meta && table, "dns_encode_bwd: d_grid given without table/meta");
    DNS_REQUIRE(meta->n_features == 2, "dns_encode_bwd: n_features must be 2");
    lv = to_levels(meta);
  }
  if (d_pe) DNS_REQUIRE(n_bins >= 1 && n_bins <= 64 && ld_dpe >= 3 * n_bins, "dns_encode_bwd: n_bins %u / ld %u", n_bins, ld_dpe);
  hipStream_t st = (hipStream_t)stream;
  const uint32_t blocks = (P + 255) / 256;
  // table gradient: LDS-binned scatter unless the caller asks for the per-corner global atomics (or passes no workspace)
  const bool binned = d_table && d_grid && ws && meta->n_levels <= 16 &&   // measured faster at T=2^16 (17x) and T=2^20 (2.5x)
                      (flags & DNS_SCATTER_MASK) != DNS_SCATTER_ATOMIC;
  if (binned) {
    const int rc = ensure_ready(st, "dns_encode_bwd");
    if (rc != DNS_OK) return rc;
  }
  float* d_table_direct = binned ? nullptr : d_table;
  if (d_x || d_table_direct) {
    const uint32_t blocks128 = (P + 127) / 128;
    const bool tiled = d_pe && d_grid && d_x && ld_dpe == ld_dgrid && d_grid == d_pe + 3 * n_bins &&
                       ld_dpe == 3 * n_bins + 2 * lv.n_levels;
    // OneBlob columns only (Decoder.merge's relative points: 3 x P rows of a [., 48 + 64] matrix per cfg3 iteration): the same
    // tile staging without grid phases -- the direct form's lanes each walk their own 448-byte-strided row (190 us for 786 432
    // rows; through the tile: the rows' 192 bytes come in as coalesced runs)
    const bool tiled_pe = d_pe && !d_grid && d_x && !d_table_direct && 128u * (3u * n_bins + 1u) * sizeof(float) <= 64u * 1024u;
    if (tiled || tiled_pe) {
      uint32_t pe_ph, g_ph;
      encode_tile_phases(n_bins, tiled_pe ? 1u : lv.n_levels, true, pe_ph, g_ph);
      if (tiled_pe) g_ph = 0u;
      if ((n_bins % 4u) != 0u) pe_ph = 1u;              // (the staging reads 16 bytes at a time)
      const uint32_t w_pe = 3 * n_bins / pe_ph, w_g = g_ph ? 2 * lv.n_levels / g_ph : 0u;
      DNS_LAUNCH(encode_bwd_kernel<true>, dim3(blocks128), dim3(128), (size_t)128 * ((w_pe > w_g ? w_pe : w_g) + 1) * sizeof(float), st, x,
                         make_bound(bound), bound ? 1 : 0, P, n_bins, (const float2*)table, lv, d_pe, ld_dpe, d_grid, ld_dgrid,
                         d_table_direct, d_x, (const float2*)dy_dx, pe_ph, g_ph);
    } else {
      DNS_LAUNCH(encode_bwd_kernel<false>, dim3(blocks128), dim3(128), 0, st, x, make_bound(bound), bound ? 1 : 0, P,
                         n_bins, (const float2*)table, lv, d_pe, ld_dpe, d_grid, ld_dgrid, d_table_direct, d_x,
                         (const float2*)dy_dx, 1u, 1u);
    }
  }
  if (binned) {
    BinPlan plan;
    plan.n_levels = lv.n_levels;
    static const uint32_t rows_env = [] { const char* e = getenv("DNS_BIN_ROWS"); const long n = e ? atol(e) : 0; return (uint32_t)(n >= 1024 && n <= 8192 ? n : 0); }();
    plan.chunk_rows = rows_env ? rows_env : 8192u;
    plan.xcd_major = 1u;
    plan.strided_dense = 1u;
    static const bool no_runs = [] { const char* e = getenv("DNS_DENSE_RUNS"); return e && e[0] == '0'; }();
    plan.dense_runs = no_runs ? 0u : 1u;
    // multi-chunk levels: pair lists (DNS_SCATTER_LISTS), else the partition form for levels of large tables
    // (DNS_SCATTER_QUEUES sends every multi-chunk level there, _BINNED none)
    const ScatterWs W = scatter_ws(P, lv, flags, queue_cap);
    const PartPlan& pp = W.pp;
    const bool part = W.part;
    bool in_part[DNS_MAX_LEVELS];
    for (uint32_t l = 0; l < DNS_MAX_LEVELS; ++l) in_part[l] = W.in_part[l] || W.in_list[l];
    uint32_t total_chunks = 0, chunk_of[DNS_MAX_LEVELS];
    for (uint32_t l = 0; l < lv.n_levels; ++l) {
      chunk_of[l] = in_part[l] ? 0u : (lv.size[l] + plan.chunk_rows - 1) / plan.chunk_rows;
      total_chunks += chunk_of[l];
    }
    if (total_chunks == 0) total_chunks = 1;
    // ~1280 workgroups in all (five per CU; one fits a CU at a time): a dense level's jobs are sliced 4x (one chunk) / 2x finer,
    // see below.  Round 2 aimed at 512 (two rounds): stand-alone the kernel does not care (221-224 us at 5, 10 slices per hashed
    // level), but inside the two-stream step finer jobs leave fewer CUs idle behind the last round and interleave better with
    // the other stream's kernels: 2.08-2.10 -> 2.05-2.07 ms per step at 10-12 slices, worse again at 16-20
    uint32_t weight = 0;
    for (uint32_t l = 0; l < lv.n_levels; ++l) weight += chunk_of[l] * (lv.hashed[l] ? 1u : (chunk_of[l] == 1 ? 4u : 2u));
    if (weight == 0) weight = 1;
    uint32_t ns = (bin_target_jobs(W.lists) + weight - 1) / weight;
    if (ns < 1) ns = 1;
    const uint32_t max_ns = (P + bin_threads() - 1) / bin_threads();   // at least ~one point per thread
    if (ns > max_ns) ns = max_ns ? max_ns : 1;
    uint32_t jobs = 0, groups = 0;
    uint32_t per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t l = 0; l < lv.n_levels; ++l) {
      plan.chunks[l] = chunk_of[l];
      // dense (coarse) levels: every corner of every point lands in the chunk -> ~4x the work per point
      uint32_t nsl = lv.hashed[l] ? ns : ns * (chunk_of[l] == 1 ? 4u : 2u);
      if (nsl > max_ns) nsl = max_ns ? max_ns : 1;
      plan.slices[l] = nsl;
      plan.job_prefix[l] = jobs;
      plan.group_prefix[l] = groups;
      for (uint32_t g = groups; g < groups + nsl; ++g) per_xcd[g & 7u] += chunk_of[l];
      jobs += chunk_of[l] * nsl;
      groups += nsl;
    }
    for (uint32_t l = lv.n_levels; l <= DNS_MAX_LEVELS; ++l) {
      plan.job_prefix[l] = jobs;
      plan.group_prefix[l] = groups;
    }
    if (plan.xcd_major) {
      uint32_t mx = 0;
      for (int i = 0; i < 8; ++i) mx = per_xcd[i] > mx ? per_xcd[i] : mx;
      jobs = 8u * mx;                                          // padded: workgroups past an XCD's last job exit at once
    }
    const size_t lds_bytes = (size_t)8192u * 2 * sizeof(unsigned long long);          // queue / list kernels: 8192-row chunks
    const size_t bin_lds = (size_t)plan.chunk_rows * 2 * sizeof(unsigned long long);
    // row replay: hashed levels of <= 2^16 rows that this (binned) form handles in more than one chunk
    ReplayPlan rp;
    uint32_t n_replay = 0;
    for (uint32_t l = 0; l < DNS_MAX_LEVELS; ++l) {
      const bool yes = (flags & DNS_SCATTER_REPLAY) && l < lv.n_levels && lv.hashed[l] && !in_part[l] && lv.size[l] <= 65536u && chunk_of[l] > 1u;
      rp.slot[l] = yes ? (int32_t)n_replay++ : -1;
    }
    DNS_REQUIRE(!n_replay || (((uintptr_t)ws) & 15u) == 0, "dns_encode_bwd: DNS_SCATTER_REPLAY needs a 16-byte aligned workspace");
    uint4* rows16 = n_replay ? reinterpret_cast<uint4*>(ws + W.replay) : nullptr;
    uint32_t* gmax = (uint32_t*)(ws + W.gmax);
    {                                                                          // max word, non-finite flag, pad
      const int rc = fill_words(gmax, 0u, 4 + (W.lists ? 2u * W.lp.qoff[W.lp.n] : 0u), st, "dns_encode_bwd");
      if (rc != DNS_OK) return rc;
    }
    static const uint32_t dg_tiles = [] { const char* e = getenv("DNS_DG_TILES"); const long n = e ? atol(e) : 0; return (uint32_t)(n >= 1 && n <= 64 ? n : DG_TILES); }();
    DNS_LAUNCH(dgrid_transpose_kernel, dim3((blocks + dg_tiles - 1) / dg_tiles), dim3(256), 0, st, d_grid, ld_dgrid, P, lv.n_levels, (float2*)ws, gmax, x, lv, rp,
               rows16, dg_tiles);
    if (jobs)
      DNS_LAUNCH(hashgrid_bwd_binned_kernel, dim3(jobs), dim3(bin_threads()), bin_lds, st, x, P, lv, plan, (const float2*)ws, gmax, d_table,
                 rp, (const uint4*)rows16);
    if (W.lists) {
      const ListPlan& lp = W.lp;
      uint32_t* lcount = reinterpret_cast<uint32_t*>(ws + W.lcount);
      uint32_t* lists = reinterpret_cast<uint32_t*>(ws + W.lwords);
      const uint32_t n_lists = lp.qoff[lp.n];
      uint32_t* lcursor = lcount + n_lists;
      uint32_t* lbase = lcursor + n_lists;
      uint32_t* jobstart = lbase + n_lists;
      const uint32_t gx = (P + LIST_THREADS * lp.tiles - 1) / (LIST_THREADS * lp.tiles);
      DNS_LAUNCH(hashgrid_bwd_pairlist_kernel, dim3(gx, lp.n), dim3(LIST_THREADS), 0, st, x, P, lv, lp, (const float2*)ws, gmax, lcount,
                 lists, d_table, 0u, (const uint32_t*)nullptr);
      size_t bins_lds = lds_bytes >> (13u - lp.chunk_shift);
      uint32_t jobs2 = n_lists * lp.slices;
      if (lp.balanced) {
        DNS_LAUNCH(hashgrid_bwd_pairscan_kernel, dim3(1), dim3(1024), 0, st, lp, (const uint32_t*)lcount, lbase, jobstart);
        DNS_LAUNCH(hashgrid_bwd_pairlist_kernel, dim3(gx, lp.n_dense), dim3(LIST_THREADS), 0, st, x, P, lv, lp, (const float2*)ws, gmax,
                   lcursor, lists, d_table, 1u, (const uint32_t*)lbase);
        bins_lds += (size_t)4u * (n_lists + 1u);
        jobs2 = lp.max_jobs;
      }
      DNS_REQUIRE(bins_lds <= (size_t)MAX_DYN_LDS, "dns_encode_bwd: the pair-list bins (%zu B: DNS_LIST_SHIFT=%u, %u lists) exceed the "
                  "%d B of LDS a workgroup can have", bins_lds, lp.chunk_shift, n_lists, MAX_DYN_LDS);
      DNS_LAUNCH(hashgrid_bwd_pairbins_kernel, dim3(jobs2), dim3(list_threads(lp.chunk_shift)), bins_lds, st, x, P, lv, lp,
                 (const float2*)ws, lcount, lists, gmax, d_table, (const uint32_t*)lbase, (const uint32_t*)jobstart);
    }
    if (part) {
      uint32_t* qcount = reinterpret_cast<uint32_t*>(ws + W.qcount);
      float* queues = ws + W.queues;
      {
        const int rc = fill_words(qcount, 0u, pp.qoff[pp.n], st, "dns_encode_bwd");
        if (rc != DNS_OK) return rc;
      }
      DNS_LAUNCH(hashgrid_bwd_partition_kernel, dim3((P + PART_THREADS - 1) / PART_THREADS), dim3(PART_THREADS), 0, st, x, P, lv,
                         pp, (const float2*)ws, qcount, queues, d_table);
      DNS_LAUNCH(hashgrid_bwd_queue_kernel, dim3(pp.qoff[pp.n] * pp.slices), dim3(1024), lds_bytes, st, lv, pp,
                         qcount, queues, gmax, d_table);
    }
  }
  return check_launch("dns_encode_bwd");
}

extern "C" int dns_hashgrid_indices(const float* x, uint32_t P, const DnsGridMeta* meta, uint32_t* rows, void* stream) {
  if (P == 0) return DNS_OK;
  DNS_REQUIRE(x && meta && rows, "dns_hashgrid_indices: NULL argument");
  DNS_LAUNCH(hashgrid_indices_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, P,
                     to_levels(meta), rows);
  return check_launch("dns_hashgrid_indices");
}

extern "C" uint64_t dns_encode_bwd_ws_floats(uint32_t P, const DnsGridMeta* meta, uint32_t flags, uint32_t queue_cap) {
  if (!meta) return 0;
  const GridLevels lv = to_levels(meta);
  return scatter_ws(P, lv, flags, queue_cap).total;
}

// Ray generation + depth-guided sampling, one wave64 per ray.
//
// Replaces the per-frame body of Mapper/Tracker.get_target_samples (reference slams/mapping.py:487-531,
// slams/tracking.py:133-160): get_rotation_from_quad (utils/common.py:447-458 -> quad2rotation :406-429),
// get_sample_uv / select_uv (:266-293; the drawn indices are an INPUT), get_rays_from_uv (:248-264), the fp64
// box far clip (mapping.py:519-527), sample_along_rays (common.py:561-599) and pts = o + d*z (mapping.py:531).
//
// The reference builds two HxW meshgrids per call just to index them, launches ~40 tiny kernels and a
// device-wide sort over [n,47]; here a wave owns a ray, every lane builds one (or E) z value and the per-ray
// sort is a bitonic network over wave shuffles on order-preserving uint keys.  All arithmetic that decides
// a value bit-for-bit (the fp64 promotions through `bound`, the unfused fp32 products) is spelled with
// __fmul_rn/__fadd_rn so the compiler cannot contract it; this file is compiled with -ffp-contract=off too.
// Latency-bound (a few hundred bytes per ray); launch count is what it removes.
#include "common.hpp"
#include "dev_sample.hpp"

namespace dns {

__global__ void depth_max_kernel(const int64_t* __restrict__ pix_idx, const float* __restrict__ depth, int H, int W,
                                 int H0, int W0, int wwin, int n_frames, int npf, uint32_t* __restrict__ ws) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = n < n_frames * npf;
  const int f = live ? n / npf : -1;
  float d = 0.f;
  if (live) {
    const int64_t q = pix_idx[n];
    const int row = H0 + (int)(q / wwin), col = W0 + (int)(q % wwin);
    d = depth[((size_t)f * H + row) * W + col];
  }
  // one atomic per wave when the wave's rays share a frame (the usual case): a full-image render puts 307 200 rays on
  // ONE address otherwise (measured 300 us of same-address atomics)
  const int f0 = __shfl(f, 0);
  if (__all(f == f0 || f < 0)) {
    float m = d;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && f0 >= 0 && m > 0.f) atomicMax(ws + f0, __float_as_uint(m));
    return;
  }
  if (live && d > 0.f) atomicMax(ws + f, __float_as_uint(d));  // non-negative floats order like their bit patterns
}

template <int E>
__global__ __launch_bounds__(256) void raygen_sample_kernel(
    const int64_t* __restrict__ pix_idx, const float* __restrict__ color, const float* __restrict__ depth,
    const float* __restrict__ label, const float* __restrict__ quat, const float* __restrict__ trans, Cam cam,
    BoundD bd, int H, int W, int H0, int W0, int wwin, int n_frames, int npf, const float* __restrict__ t_uniform,
    const float* __restrict__ t_surf, const float* __restrict__ t_zero, int nu, int ns, int jstride,
    const uint32_t* __restrict__ dmax_ws, float* __restrict__ rays_o, float* __restrict__ rays_d,
    float* __restrict__ gt_color, float* __restrict__ gt_depth, int64_t* __restrict__ gt_label,
    uint8_t* __restrict__ inside, float* __restrict__ z_out, float* __restrict__ pts_out) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= n_frames * npf) return;
  const int f = n / npf;
  const int S = nu + ns;
  int row, col;
  float dir[3];
  pixel_dir(pix_idx[n], H0, W0, wwin, cam, row, col, dir);
  const size_t pix = ((size_t)f * H + row) * W + col;
  float R[9];
  quat_to_R(quat + 4 * f, R);
  float o[3], d[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    o[a] = trans[3 * f + a];
    d[a] = __fadd_rn(__fadd_rn(__fmul_rn(dir[0], R[3 * a]), __fmul_rn(dir[1], R[3 * a + 1])), __fmul_rn(dir[2], R[3 * a + 2]));
  }
  const float gd = depth[pix];
  // box far clip in fp64 (bound is float64 in the reference, so everything promotes)
  double far = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double t0 = (bd.b[2 * a] - (double)o[a]) / (double)d[a];
    const double t1 = (bd.b[2 * a + 1] - (double)o[a]) / (double)d[a];
    const double m = max_nan(t0, t1);
    far = (a == 0) ? m : min_nan(far, m);
  }
  const bool in = far >= (double)gd;
  far += 0.01;
  const float dmax = __uint_as_float(dmax_ws[f]);
  uint32_t key[E];
  // jstride = n_surface: frame f has its own jitter rows (the reference draws them per frame, utils/common.py:571,582)
  sample_and_sort<E>(gd, far, dmax, t_uniform, t_surf + (size_t)f * jstride, t_zero + (size_t)f * jstride, nu, ns, lane, key);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int s = e * 64 + (int)lane;
    if (s < S) {
      const float zv = key2f(key[e]);
      z_out[(size_t)n * S + s] = zv;
      if (pts_out) {
        float* p = pts_out + ((size_t)n * S + s) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a] = __fadd_rn(o[a], __fmul_rn(d[a], zv));
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      rays_o[(size_t)n * 3 + a] = o[a];
      rays_d[(size_t)n * 3 + a] = d[a];
      gt_color[(size_t)n * 3 + a] = color[pix * 3 + a];
    }
    gt_depth[n] = gd;
    gt_label[n] = (int64_t)label[pix];
    inside[n] = in ? 1 : 0;
  }
}


// get_sample_uv / select_uv gather + get_rays_from_uv (utils/common.py:248-304), and get_all_rays (:540-559), for a
// rotation given as a MATRIX (the free functions' signature): one thread per ray.  image [H, W, C] rows are gathered for the
// drawn pixels (C <= 8: rgb | depth | label).  Same un-fused arithmetic as raygen_sample_kernel (sum(dirs * R, -1)).
__global__ __launch_bounds__(256) void rays_from_pixels_kernel(const int64_t* __restrict__ pix_idx, const float* __restrict__ image,
                                                               int C, const float* __restrict__ R, const float* __restrict__ T,
                                                               Cam cam, int W, int H0, int W0, int wwin, int n,
                                                               float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                               float* __restrict__ sample, float* __restrict__ ij) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int row, col;
  float dir[3];
  pixel_dir(pix_idx ? pix_idx[r] : (int64_t)r, H0, W0, wwin, cam, row, col, dir);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    rays_o[(size_t)r * 3 + a] = T[a];
    rays_d[(size_t)r * 3 + a] =
        __fadd_rn(__fadd_rn(__fmul_rn(dir[0], R[3 * a]), __fmul_rn(dir[1], R[3 * a + 1])), __fmul_rn(dir[2], R[3 * a + 2]));
  }
  if (sample) {
    const float* src = image + ((size_t)row * W + col) * C;
    for (int c = 0; c < C; ++c) sample[(size_t)r * C + c] = src[c];
  }
  if (ij) {
    ij[2 * (size_t)r] = (float)col;
    ij[2 * (size_t)r + 1] = (float)row;
  }
}

// Stand-alone sample_along_rays(gt_depth, n_samples, n_surface, far_bb) (utils/common.py:561): far_bb is an input.
__global__ void depth_max_flat_kernel(const float* __restrict__ depth, int n, uint32_t* __restrict__ ws) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float m = (i < n && depth[i] > 0.f) ? depth[i] : 0.f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));     // one atomic per wave, not per ray, on this one word
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(ws, __float_as_uint(m));
}

template <int E>
__global__ __launch_bounds__(256) void sample_along_rays_kernel(const float* __restrict__ depth, const double* __restrict__ far_bb,
                                                                int n_rays, const float* __restrict__ t_uniform,
                                                                const float* __restrict__ t_surf, const float* __restrict__ t_zero,
                                                                int nu, int ns, const uint32_t* __restrict__ dmax_ws,
                                                                float* __restrict__ z_out) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= n_rays) return;
  const int S = nu + ns;
  uint32_t key[E];
  sample_and_sort<E>(depth[n], far_bb[n], __uint_as_float(dmax_ws[0]), t_uniform, t_surf, t_zero, nu, ns, lane, key);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int s = e * 64 + (int)lane;
    if (s < S) z_out[(size_t)n * S + s] = key2f(key[e]);
  }
}

// d_pts [n,S,3] (+ optional direct d_rays_o / d_rays_d) -> per-frame sums of dL/dR (9) and dL/dT (3).
// grid = (ceil(npf / (4 RB_RAYS)), n_frames): a workgroup owns 4 * RB_RAYS rays of ONE frame (RB_RAYS per wave, a wave reads a
// ray's d_pts row coalesced), reduces in registers + LDS and issues 12 atomics.  RB_RAYS = 2: 4096 rays give 512 workgroups
// (16 rays per wave left three quarters of the CUs without work and each wave with 16 dependent reductions in a row: 26 us).
constexpr int RB_RAYS = 2;
__global__ __launch_bounds__(256) void raygen_bwd_reduce_kernel(const int64_t* __restrict__ pix_idx, Cam cam, int H0, int W0,
                                                                int wwin, int n_frames, int npf, int S,
                                                                const float* __restrict__ z, const float* __restrict__ d_pts,
                                                                const float* __restrict__ d_ro, const float* __restrict__ d_rd,
                                                                float* __restrict__ ws) {
  __shared__ float part[4][12];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const int f = blockIdx.y;
  float acc[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) acc[e] = 0.f;
  for (int i = 0; i < RB_RAYS; ++i) {
    const int r = blockIdx.x * (4 * RB_RAYS) + wave * RB_RAYS + i;
    if (r >= npf) break;
    const int n = f * npf + r;
    float go[3] = {0.f, 0.f, 0.f}, gdv[3] = {0.f, 0.f, 0.f};
    if (d_pts) {
      for (int s = lane; s < S; s += 64) {
        const float zv = z[(size_t)n * S + s];
        const float* g = d_pts + ((size_t)n * S + s) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          go[a] += g[a];
          gdv[a] += g[a] * zv;
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          go[a] += __shfl_xor(go[a], off);
          gdv[a] += __shfl_xor(gdv[a], off);
        }
    }
    if (d_ro)
      for (int a = 0; a < 3; ++a) go[a] += d_ro[(size_t)n * 3 + a];
    if (d_rd)
      for (int a = 0; a < 3; ++a) gdv[a] += d_rd[(size_t)n * 3 + a];
    int row, col;
    float dir[3];
    pixel_dir(pix_idx[n], H0, W0, wwin, cam, row, col, dir);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[3 * a + b] += gdv[a] * dir[b];  // dL/dR[a][b] = sum dL/dd[a] * dir[b]
      acc[9 + a] += go[a];
    }
  }
  if (lane == 0)
#pragma unroll
    for (int e = 0; e < 12; ++e) part[wave][e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 12) {
    // the workgroup's 12 partial sums, parked: raygen_bwd_pose_kernel (the next launch) adds a frame's workgroups up -- no
    // atomics on 12 shared words, and no zero fill of the workspace ahead of this kernel
    const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    ws[((size_t)f * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = v;
  }
}

// one wave per frame: sums the frame's n_part partial [12] vectors, then the quaternion chain rule on lane 0
__global__ __launch_bounds__(64) void raygen_bwd_pose_kernel(const float* __restrict__ quat, const float* __restrict__ ws,
                                                             int n_frames, int n_part, float* __restrict__ d_quat,
                                                             float* __restrict__ d_trans) {
  const int f = blockIdx.x;
  float G[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    float v = 0.f;
    for (int g = threadIdx.x; g < n_part; g += 64) v += ws[((size_t)f * n_part + g) * 12 + e];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    G[e] = v;
  }
  if (threadIdx.x != 0) return;
  const float r = quat[4 * f], i = quat[4 * f + 1], j = quat[4 * f + 2], k = quat[4 * f + 3];
  const float nrm = r * r + i * i + j * j + k * k;
  const float two_s = 2.0f / nrm;
  // R = I + two_s * M(q)
  const float M[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k),
                      j * k - i * r,    i * k - j * r, j * k + i * r, -(i * i + j * j)};
  float gm = 0.f;
  for (int e = 0; e < 9; ++e) gm += G[e] * M[e];
  const float dr = -k * G[1] + j * G[2] + k * G[3] - i * G[5] - j * G[6] + i * G[7];
  const float di = j * (G[1] + G[3]) + k * (G[2] + G[6]) - 2.f * i * (G[4] + G[8]) + r * (G[7] - G[5]);
  const float dj = -2.f * j * (G[0] + G[8]) + i * (G[1] + G[3]) + r * (G[2] - G[6]) + k * (G[5] + G[7]);
  const float dk = -2.f * k * (G[0] + G[4]) + r * (G[3] - G[1]) + i * (G[2] + G[6]) + j * (G[5] + G[7]);
  const float c = two_s * two_s * gm;
  if (d_quat) {
    d_quat[4 * f + 0] += two_s * dr - c * r;
    d_quat[4 * f + 1] += two_s * di - c * i;
    d_quat[4 * f + 2] += two_s * dj - c * j;
    d_quat[4 * f + 3] += two_s * dk - c * k;
  }
  if (d_trans)
    for (int a = 0; a < 3; ++a) d_trans[3 * f + a] += G[9 + a];
}

}  // namespace dns

using namespace dns;

static Cam make_cam(const double* cam) {
  Cam c;
  c.fx = (float)cam[0];
  c.fy = (float)cam[1];
  c.cx = (float)cam[2];
  c.cy = (float)cam[3];
  return c;
}

extern "C" int dns_raygen_sample(const int64_t* pix_idx, const float* color, const float* depth, const float* label,
                                 const float* quat, const float* trans, const double* cam, const double* bound, int H,
                                 int W, int H0, int H1, int W0, int W1, int n_frames, int n_per_frame,
                                 const float* t_uniform, const float* t_surf, const float* t_zero, int n_uniform,
                                 int n_surface, int jitter_stride, uint32_t* depth_max_ws, int depth_max_given, float* rays_o,
                                 float* rays_d, float* gt_color,
                                 float* gt_depth, int64_t* gt_label, uint8_t* inside, float* z, float* pts,
                                 void* stream) {
  DNS_REQUIRE(pix_idx && color && depth && label && quat && trans && cam && bound, "dns_raygen_sample: NULL input");
  DNS_REQUIRE(rays_o && rays_d && gt_color && gt_depth && gt_label && inside && z && depth_max_ws, "dns_raygen_sample: NULL output");
  DNS_REQUIRE(0 <= H0 && H0 < H1 && H1 <= H && 0 <= W0 && W0 < W1 && W1 <= W, "dns_raygen_sample: bad window");
  DNS_REQUIRE(n_uniform >= 0 && n_surface >= 1 && n_uniform + n_surface <= 256, "dns_raygen_sample: samples per ray %d+%d out of range", n_uniform, n_surface);
  DNS_REQUIRE(n_uniform == 0 || t_uniform, "dns_raygen_sample: t_uniform is NULL");
  DNS_REQUIRE(t_surf && t_zero, "dns_raygen_sample: jitter vectors are NULL");
  DNS_REQUIRE(jitter_stride == 0 || jitter_stride >= n_surface, "dns_raygen_sample: jitter_stride %d (0 = one pair shared by all frames, else >= n_surface)", jitter_stride);
  const int n = n_frames * n_per_frame;
  if (n <= 0) return DNS_OK;
  hipStream_t st = (hipStream_t)stream;
  const int wwin = W1 - W0;
  if (!depth_max_given) {
    {
      const int rc = fill_words(depth_max_ws, 0u, (size_t)n_frames, st, "dns_raygen_sample");
      if (rc != DNS_OK) return rc;
    }
    DNS_LAUNCH(depth_max_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pix_idx, depth, H, W, H0, W0, wwin, n_frames, n_per_frame, depth_max_ws);
  }
  BoundD bd;
  for (int i = 0; i < 6; ++i) bd.b[i] = bound[i];
  const Cam c = make_cam(cam);
  const int S = n_uniform + n_surface;
  const dim3 grid((n + 3) / 4), block(256);
#define LAUNCH_RS(E)                                                                                                   \
  DNS_LAUNCH(raygen_sample_kernel<E>, grid, block, 0, st, pix_idx, color, depth, label, quat, trans, c, bd, H, \
                     W, H0, W0, wwin, n_frames, n_per_frame, t_uniform, t_surf, t_zero, n_uniform, n_surface,          \
                     jitter_stride, depth_max_ws, rays_o, rays_d, gt_color, gt_depth, gt_label, inside, z, pts)
  if (S <= 64) LAUNCH_RS(1);
  else if (S <= 128) LAUNCH_RS(2);
  else LAUNCH_RS(4);
#undef LAUNCH_RS
  return check_launch("dns_raygen_sample");
}

extern "C" int dns_sample_along_rays(const float* gt_depth, const double* far_bb, int n_rays, const float* t_uniform,
                                     const float* t_surf, const float* t_zero, int n_uniform, int n_surface,
                                     uint32_t* depth_max_ws, float* z, void* stream) {
  DNS_REQUIRE(gt_depth && far_bb && t_surf && t_zero && depth_max_ws && z, "dns_sample_along_rays: NULL argument");
  DNS_REQUIRE(n_uniform >= 0 && n_surface >= 1 && n_uniform + n_surface <= 256, "dns_sample_along_rays: samples per ray %d+%d out of range", n_uniform, n_surface);
  DNS_REQUIRE(n_uniform == 0 || t_uniform, "dns_sample_along_rays: t_uniform is NULL");
  if (n_rays <= 0) return DNS_OK;
  hipStream_t st = (hipStream_t)stream;
  {
    const int rc = fill_words(depth_max_ws, 0u, 1, st, "dns_sample_along_rays");
    if (rc != DNS_OK) return rc;
  }
  DNS_LAUNCH(depth_max_flat_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, st, gt_depth, n_rays, depth_max_ws);
  const int S = n_uniform + n_surface;
  const dim3 grid((n_rays + 3) / 4), block(256);
  if (S <= 64) DNS_LAUNCH(sample_along_rays_kernel<1>, grid, block, 0, st, gt_depth, far_bb, n_rays, t_uniform, t_surf, t_zero, n_uniform, n_surface, depth_max_ws, z);
  else if (S <= 128) DNS_LAUNCH(sample_along_rays_kernel<2>, grid, block, 0, st, gt_depth, far_bb, n_rays, t_uniform, t_surf, t_zero, n_uniform, n_surface, depth_max_ws, z);
  else DNS_LAUNCH(sample_along_rays_kernel<4>, grid, block, 0, st, gt_depth, far_bb, n_rays, t_uniform, t_surf, t_zero, n_uniform, n_surface, depth_max_ws, z);
  return check_launch("dns_sample_along_rays");
}

extern "C" int dns_raygen_bwd(const int64_t* pix_idx, const float* quat, const double* cam, int H0, int H1, int W0,
                              int W1, int n_frames, int n_per_frame, int S, const float* z, const float* d_pts,
                              const float* d_rays_o, const float* d_rays_d, float* ws, float* d_quat, float* d_trans,
                              void* stream) {
  DNS_REQUIRE(pix_idx && quat && cam && ws, "dns_raygen_bwd: NULL argument");
  DNS_REQUIRE(!d_pts || z, "dns_raygen_bwd: d_pts needs z");
  (void)H1;
  const int n = n_frames * n_per_frame;
  if (n <= 0) return DNS_OK;
  hipStream_t st = (hipStream_t)stream;
  const int n_part = (n_per_frame + 4 * RB_RAYS - 1) / (4 * RB_RAYS);          // == dns_raygen_bwd_ws_floats / (12 n_frames)
  DNS_LAUNCH(raygen_bwd_reduce_kernel, dim3(n_part, n_frames), dim3(256), 0, st, pix_idx, make_cam(cam), H0, W0, W1 - W0,
                     n_frames, n_per_frame, S, z, d_pts, d_rays_o, d_rays_d, ws);
  DNS_LAUNCH(raygen_bwd_pose_kernel, dim3(n_frames), dim3(64), 0, st, quat, ws, n_frames, n_part, d_quat, d_trans);
  return check_launch("dns_raygen_bwd");
}

extern "C" uint64_t dns_raygen_bwd_ws_floats(int n_frames, int n_per_frame) {
  if (n_frames <= 0 || n_per_frame <= 0) return 0;
  return (uint64_t)12 * (uint64_t)n_frames * (uint64_t)((n_per_frame + 4 * RB_RAYS - 1) / (4 * RB_RAYS));
}

extern "C" int dns_rays_from_pixels(const int64_t* pix_idx, const float* image, int C, const float* R, const float* T,
                                    const double* cam, int H, int W, int H0, int H1, int W0, int W1, int n, float* rays_o,
                                    float* rays_d, float* sample, float* ij, void* stream) {
  if (n <= 0) return DNS_OK;
  DNS_REQUIRE(R && T && cam && rays_o && rays_d, "dns_rays_from_pixels: NULL argument");
  DNS_REQUIRE(H0 >= 0 && W0 >= 0 && H1 <= H && W1 <= W && H1 > H0 && W1 > W0, "dns_rays_from_pixels: window [%d,%d)x[%d,%d) outside %dx%d", H0, H1, W0, W1, H, W);
  DNS_REQUIRE(pix_idx || n <= (H1 - H0) * (W1 - W0), "dns_rays_from_pixels: n exceeds the window (pix_idx == NULL means pixel r = ray r)");
  DNS_REQUIRE(!sample || (image && C >= 1 && C <= 8), "dns_rays_from_pixels: sample rows need image and 1 <= C <= 8");
  Cam c;
  c.fx = (float)cam[0]; c.fy = (float)cam[1]; c.cx = (float)cam[2]; c.cy = (float)cam[3];
  DNS_LAUNCH(rays_from_pixels_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pix_idx, image, C, R, T,
                     c, W, H0, W0, W1 - W0, n, rays_o, rays_d, sample, ij);
  return check_launch("dns_rays_from_pixels");
}

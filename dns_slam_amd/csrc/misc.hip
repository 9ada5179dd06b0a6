// Small fused helpers of the mapping iteration.
//
// (1) Total-variation smoothness of the coarse occupancy on an n^3 lattice (reference slams/mapping.py:151-157):
//     loss = (sum (occ[i+1]-occ[i])^2 over x, y, z) / sample_points^3, occ = coarse[:, 0].  The reference spends a
//     dozen launches (three slices, three pows, three sums) on it; here one reduction kernel and one gradient kernel.
// (2) Grouping of points by weight-set id for the grouped MLP (the per-class dispatch of Mapper.fine_fn,
//     slams/mapping.py:590-601, which loops over classes with two host syncs each): a counting sort into 128-slot
//     tiles -- histogram, one-workgroup scan (padded tile offsets + tile -> group map), scatter -- instead of
//     ~25 torch launches (argsort, cumsum, searchsorted, ...).
#include "common.hpp"

namespace dns {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// The lattice is a slab of nx x-planes of n x n points (nx = n: the whole cube).  With `halo` the LAST plane belongs to the
// next slab (multi-GPU: the lattice is cut along x, one slab per rank): it only closes the x-differences of plane nx-2, its
// own y / z differences are the next rank's -- the slabs' sums add up to the cube's sum, term for term.
__global__ __launch_bounds__(256) void tv_fwd_kernel(const float* __restrict__ lat, uint32_t ld, uint32_t nx, uint32_t n,
                                                     uint32_t halo, float inv_norm, float* __restrict__ out) {
  __shared__ float sh[4];
  const uint32_t total = nx * n * n;
  float acc = 0.f;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const uint32_t k = e % n, j = (e / n) % n, i = e / (n * n);
    const bool own = !(halo && i + 1 == nx);
    const float v = lat[(size_t)e * ld];
    if (i + 1 < nx) { const float d = lat[(size_t)(e + n * n) * ld] - v; acc += d * d; }
    if (own && j + 1 < n) { const float d = lat[(size_t)(e + n) * ld] - v; acc += d * d; }
    if (own && k + 1 < n) { const float d = lat[(size_t)(e + 1) * ld] - v; acc += d * d; }
  }
  acc = wave_sum_f(acc);
  if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (sh[0] + sh[1] + sh[2] + sh[3]) * inv_norm);
}

// d_lat [n^3, ld]: column 0 = d loss / d occ, the other columns zero (the gradient of coarse[:, 0:1] padded back)
// Flat walk over the [n^3 * ld] output (coalesced stores; a thread-per-point walk writes 132-byte rows at a 132-byte lane
// stride, 33 scattered store instructions per wave): the element that is a row's column 0 evaluates the stencil.
__global__ __launch_bounds__(256) void tv_bwd_kernel(const float* __restrict__ lat, uint32_t ld, uint32_t nx, uint32_t n,
                                                     uint32_t halo, float inv_norm, const float* __restrict__ g,
                                                     float* __restrict__ d_lat) {
  const uint32_t total = nx * n * n * ld;              // < 2^32 (checked on the host)
  const float c = 2.0f * inv_norm * g[0];
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const uint32_t e = t / ld;
    float out = 0.f;
    if (t - e * ld == 0) {
      const uint32_t k = e % n, j = (e / n) % n, i = e / (n * n);
      const bool own = !(halo && i + 1 == nx);
      const float v = lat[(size_t)e * ld];
      float a = 0.f;
      if (i + 1 < nx) a -= lat[(size_t)(e + n * n) * ld] - v;
      if (i > 0) a += v - lat[(size_t)(e - n * n) * ld];
      if (own && j + 1 < n) a -= lat[(size_t)(e + n) * ld] - v;
      if (own && j > 0) a += v - lat[(size_t)(e - n) * ld];
      if (own && k + 1 < n) a -= lat[(size_t)(e + 1) * ld] - v;
      if (own && k > 0) a += v - lat[(size_t)(e - 1) * ld];
      out = c * a;
    }
    d_lat[t] = out;
  }
}

// ---- grouping ---------------------------------------------------------------------------------------------------
constexpr uint32_t GROUP_MAX = 256;

__global__ __launch_bounds__(256) void group_hist_kernel(const int64_t* __restrict__ slot, uint32_t P, uint32_t G,
                                                         uint32_t* __restrict__ counts) {
  __shared__ uint32_t h[GROUP_MAX];
  for (uint32_t i = threadIdx.x; i < G; i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const int64_t s = slot[p];
    if (s >= 0 && s < (int64_t)G) atomicAdd(&h[(uint32_t)s], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < G; i += blockDim.x)
    if (h[i]) atomicAdd(counts + i, h[i]);
}

// one workgroup: padded start of every group (multiples of 128), tile -> group map, cursors = padded starts
__global__ __launch_bounds__(256) void group_scan_kernel(const uint32_t* __restrict__ counts, uint32_t G, uint32_t min_count,
                                                         uint32_t n_tiles, uint32_t* __restrict__ cursor,
                                                         int32_t* __restrict__ tile_group) {
  __shared__ uint32_t start[GROUP_MAX + 1];
  __shared__ uint32_t cn[GROUP_MAX];                 // (the counts once, in LDS: the tile loop below looked them up in memory per tile)
  for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) cn[g] = counts[g];
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t o = 0;
    for (uint32_t g = 0; g < G; ++g) {
      start[g] = o;
      o += (cn[g] + 127u) / 128u * 128u;
    }
    start[G] = o;
  }
  __syncthreads();
  for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) cursor[g] = start[g];
  for (uint32_t t = threadIdx.x; t < n_tiles; t += blockDim.x) {
    const uint32_t pos = t * 128u;
    int grp = -1;
    if (pos < start[G]) {
      uint32_t lo = 0, hi = G;                     // last g with start[g] <= pos
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (start[mid] <= pos) lo = mid; else hi = mid;
      }
      if (cn[lo] >= min_count) grp = (int)lo;      // Mapper.fine_fn: a class needs more than one point (mapping.py:597)
    }
    tile_group[t] = grp;
  }
}

// K points per thread (point p = (workgroup K + k) 256 + thread: every k is one coalesced run of the slot array).  The kernel's
// cost is its GLOBAL atomics -- one per (workgroup, group present) on the group's cursor, same-address atomics that retire at
// ~10 ns each: a frame render's 4.2 M points per chunk took 179 us with one point per thread (16 384 workgroups) however the
// points were ranked inside the workgroup; K = 8 makes an eighth of them.  Ranks inside the workgroup: one LDS atomic per
// (wave, group present in it) -- the wave's first unranked lane names a group, its members rank themselves by a ballot.
template <int K>
__global__ __launch_bounds__(256) void group_scatter_kernel(const int64_t* __restrict__ slot, uint32_t P, uint32_t G,
                                                            uint32_t* __restrict__ cursor, int32_t* __restrict__ row_index,
                                                            uint32_t n_slots, uint32_t* __restrict__ err) {
  __shared__ uint32_t cnt[GROUP_MAX];
  __shared__ uint32_t base[GROUP_MAX];
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t i = threadIdx.x; i < G; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  int s[K];
  uint32_t rank[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t p = (blockIdx.x * K + k) * blockDim.x + threadIdx.x;
    int64_t v = -1;
    if (p < P) v = slot[p];
    s[k] = (v >= 0 && v < (int64_t)G) ? (int)v : -1;
    rank[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    bool todo = s[k] >= 0;
    while (__any(todo)) {
      const uint64_t act = __ballot(todo);
      const int leader = __ffsll((long long)act) - 1;
      const int cls = __shfl(s[k], leader);
      const bool mine = todo && s[k] == cls;
      const uint64_t m = __ballot(mine);
      uint32_t first = 0;
      if ((int)lane == leader) first = atomicAdd(&cnt[(uint32_t)cls], (uint32_t)__popcll(m));
      first = (uint32_t)__shfl((int)first, leader);
      if (mine) {
        rank[k] = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo = false;
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < G; i += blockDim.x) base[i] = cnt[i] ? atomicAdd(cursor + i, cnt[i]) : 0u;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (s[k] >= 0) {
      // The cursors are DEVICE state (written by group_scan_kernel, advanced by atomics): a stale or corrupted one must end as
      // an error code, not as a store past the table (round 2's memory-access fault was exactly such a consumer)
      const uint32_t p = (blockIdx.x * K + k) * blockDim.x + threadIdx.x;
      const uint32_t at = base[(uint32_t)s[k]] + rank[k];
      if (at < n_slots) row_index[at] = (int32_t)p;
      else if (err) atomicOr(err, DNS_DEVERR_GROUP_CURSOR);
    }
  }
}

static void launch_group_scatter(const int64_t* slot_of_point, uint32_t P, uint32_t n_groups, uint32_t* cursor, int32_t* row_index,
                                 uint32_t n_slots, hipStream_t st) {
  if (P >= (1u << 20))
    DNS_LAUNCH(group_scatter_kernel<8>, dim3((P + 2047) / 2048), dim3(256), 0, st, slot_of_point, P, n_groups, cursor, row_index, n_slots,
               device_error_word());
  else
    DNS_LAUNCH(group_scatter_kernel<1>, dim3((P + 255) / 256), dim3(256), 0, st, slot_of_point, P, n_groups, cursor, row_index, n_slots,
               device_error_word());
}

}  // namespace dns

using namespace dns;

extern "C" int dns_tv_fwd(const float* lat, uint32_t ld, uint32_t nx, uint32_t n, int halo, uint32_t sample_points, float* out,
                          void* stream) {
  DNS_REQUIRE(lat && out && n >= 1 && nx >= 1 && ld >= 1, "dns_tv_fwd: bad argument");
  DNS_REQUIRE(!halo || nx >= 2, "dns_tv_fwd: a slab with a halo plane needs at least one plane of its own");
  DNS_REQUIRE((uint64_t)nx * n * n < (1ull << 31), "dns_tv_fwd: lattice too large");
  hipStream_t st = (hipStream_t)stream;
  {
    const int rc = fill_words(out, 0u, 1, st, "dns_tv_fwd");
    if (rc != DNS_OK) return rc;
  }
  const uint32_t total = nx * n * n;
  const uint32_t blocks = (total + 255) / 256 < 256 ? (total + 255) / 256 : 256;   // each ends in one atomic on ONE word
  const float inv = 1.0f / ((float)sample_points * (float)sample_points * (float)sample_points);
  DNS_LAUNCH(tv_fwd_kernel, dim3(blocks), dim3(256), 0, st, lat, ld, nx, n, halo ? 1u : 0u, inv, out);
  return check_launch("dns_tv_fwd");
}

extern "C" int dns_tv_bwd(const float* lat, uint32_t ld, uint32_t nx, uint32_t n, int halo, uint32_t sample_points, const float* g,
                          float* d_lat, void* stream) {
  DNS_REQUIRE(lat && g && d_lat && n >= 1 && nx >= 1 && ld >= 1, "dns_tv_bwd: bad argument");
  DNS_REQUIRE(!halo || nx >= 2, "dns_tv_bwd: a slab with a halo plane needs at least one plane of its own");
  DNS_REQUIRE((uint64_t)nx * n * n * ld < 0xFFFFFFFFull, "dns_tv_bwd: lattice too large for 32-bit indexing");
  const uint32_t total = nx * n * n * ld;
  const uint32_t blocks = (total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192;
  const float inv = 1.0f / ((float)sample_points * (float)sample_points * (float)sample_points);
  DNS_LAUNCH(tv_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, lat, ld, nx, n, halo ? 1u : 0u, inv, g, d_lat);
  return check_launch("dns_tv_bwd");
}

extern "C" int dns_group_slots(const int64_t* slot_of_point, uint32_t P, uint32_t n_groups, uint32_t min_count,
                               uint32_t n_slots, uint32_t* ws, int32_t* row_index, int32_t* tile_group, void* stream) {
  DNS_REQUIRE(slot_of_point && ws && row_index && tile_group, "dns_group_slots: NULL argument");
  DNS_REQUIRE(n_groups >= 1 && n_groups <= GROUP_MAX, "dns_group_slots: n_groups %u out of range [1,%u]", n_groups, GROUP_MAX);
  DNS_REQUIRE(n_slots % 128u == 0 && (uint64_t)n_slots >= (uint64_t)P + 127ull * n_groups,
              "dns_group_slots: n_slots %u too small for %u points in %u groups", n_slots, P, n_groups);
  hipStream_t st = (hipStream_t)stream;
  uint32_t* counts = ws;
  uint32_t* cursor = ws + GROUP_MAX;
  {
    int rc = ensure_ready(st, "dns_group_slots");
    if (rc == DNS_OK) rc = fill_words2(counts, 0u, GROUP_MAX, row_index, 0xFFFFFFFFu, n_slots, st, "dns_group_slots");   // -1 = padding slot
    if (rc != DNS_OK) return rc;
  }
  if (P) {
    const uint32_t hb = (P + 255) / 256 < 512 ? (P + 255) / 256 : 512;
    DNS_LAUNCH(group_hist_kernel, dim3(hb), dim3(256), 0, st, slot_of_point, P, n_groups, counts);
  }
  DNS_LAUNCH(group_scan_kernel, dim3(1), dim3(256), 0, st, counts, n_groups, min_count, n_slots / 128u, cursor, tile_group);
  if (P) launch_group_scatter(slot_of_point, P, n_groups, cursor, row_index, n_slots, st);
  return check_launch("dns_group_slots");
}

extern "C" int dns_group_scatter(const int64_t* slot_of_point, uint32_t P, uint32_t n_groups, uint32_t* cursor, uint32_t n_slots,
                                 int32_t* row_index, void* stream) {
  DNS_REQUIRE(slot_of_point && cursor && row_index, "dns_group_scatter: NULL argument");
  DNS_REQUIRE(n_groups >= 1 && n_groups <= GROUP_MAX, "dns_group_scatter: n_groups %u out of range [1,%u]", n_groups, GROUP_MAX);
  hipStream_t st = (hipStream_t)stream;
  const int rc = ensure_ready(st, "dns_group_scatter");
  if (rc != DNS_OK) return rc;
  if (P) launch_group_scatter(slot_of_point, P, n_groups, cursor, row_index, n_slots, st);
  return check_launch("dns_group_scatter");
}

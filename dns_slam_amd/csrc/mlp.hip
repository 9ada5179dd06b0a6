// Bias-free ReLU MLP on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Replaces tcnn.Network{CutlassMLP} (reference models/decoder.py:58-64,84-90,101-116,
// slams/mapping.py:737-743): y = W_out relu(W_h relu(W_in x)), no bias.
//
// Orientation (CDNA4-first, not a CUTLASS tiling): POINTS live on the MFMA column/lane axis, weights are
// the A operand.  D[32 feat x 32 pts] += A[32 feat x 2] * B[2 x 32 pts]:
//   lane l: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; D: col = l&31, row = (r&3)+8*(r>>2)+4*(l>>5).
// With points on lanes, a layer's accumulator IS the next layer's B operand (lane half h already holds the
// rows whose bit 2 equals h), so activations never leave registers between layers: no LDS round trip, no
// transposes.  Weights are staged once per workgroup into LDS in "A-operand images" -- for each k-step the
// 64 dwords the 64 lanes need, contiguous -- so every A fetch is one conflict-free ds_read_b32.
// Input rows are read as two contiguous halves (lane half h reads x[row][h*K/2 .. ) as float4).
//
// A workgroup = 4 waves = 128 point slots; a wave = one 32-point tile.  Optional row_index maps slot ->
// row of x / y (-1 = padding) and tile_group selects per-128-slot weight sets (per-class fine decoders,
// slams/mapping.py:590-601) without copying activations.
//
// Roofline: exact-fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s); these kernels are MFMA-bound for
// n_in >= 80 (x traffic 320-448 B/point vs 14-45 kFLOP/point).
#include <stdlib.h>
#include "common.hpp"

namespace dns {

#ifdef DNS_TRACE
// developer instrumentation (tools/mlp_trace.py builds a separate library with -DDNS_TRACE): issue-time stamps of one
// wave of two workgroups, 100 MHz wall clock
__device__ unsigned long long dns_trace_buf[2][64];
#define DNS_STAMP(i)                                                                                         \
  do {                                                                                                       \
    const uint32_t i_ = (i);                                                                                 \
    if ((threadIdx.x == 0) && (blockIdx.x == 0 || blockIdx.x == 300) && i_ < 64)                             \
      dns_trace_buf[blockIdx.x ? 1 : 0][i_] = wall_clock64();                                                \
  } while (0)
#else
#define DNS_STAMP(i) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t acc_row(uint32_t r, uint32_t h) { return (r & 3u) + 8u * (r >> 2) + 4u * h; }

// Images are built by walking M in MEMORY order (aligned float4s, the quad e of the matrix starts at float 4e): every
// workgroup re-reads the weights, so this walk is a large part of a launch's fixed cost.  It is split in two so that a
// kernel can put the loads of ALL its matrices in flight first (image_load), zero the image area and pass the barrier
// while they fly, and only then scatter (image_scatter) through the inverse of the k map: one memory latency per
// weight set instead of one per matrix (a gather in image order cost ~50 dependent L2 round trips per thread).
// Requires C % 4 == 0 (true for every layer: n_in % 8 == 0, hidden width 32 / 64) and R * C / 4 <= MAXQ * blockDim.
enum KMap { K_SPLIT = 0, K_CHAIN = 1, K_PAIR = 2 };
// A-operand image: img[(rt*nsteps + s)*64 + lane] = Meff[rt*32 + (lane&31)][kmap(s, lane>>5)]
// Meff[i][k] = transpose ? M[k][i] : M[i][k];  M is row-major [R x C] (logical rows/cols of M).
// K_SPLIT: k = h*khalf + s (s < khalf);  K_CHAIN: k = (s/16)*32 + acc_row(s%16, h);  K_PAIR: k = 2s + h.
template <int MAXQ>
struct ImgQuads {
  float4 v[MAXQ];
};

template <int MAXQ>
__device__ __forceinline__ void image_load(ImgQuads<MAXQ>& q, const float* __restrict__ M, uint32_t R, uint32_t C) {
  const uint32_t nq = R * (C >> 2);
  const bool vec = (((uintptr_t)M) & 15u) == 0;
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    const uint32_t e = threadIdx.x + j * blockDim.x;
    q.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < nq) {
      const float* src = M + (size_t)e * 4u;
      if (vec) {
        q.v[j] = *reinterpret_cast<const float4*>(src);
      } else {
        q.v[j] = make_float4(src[0], src[1], src[2], src[3]);
      }
    }
  }
}

template <int MAXQ>
__device__ __forceinline__ void image_scatter(const ImgQuads<MAXQ>& q, float* __restrict__ img, uint32_t R, uint32_t C,
                                              bool transpose, uint32_t row_tiles, uint32_t nsteps, int kmap, uint32_t khalf,
                                              uint32_t klimit) {
  auto step_of = [&](uint32_t k, uint32_t& st, uint32_t& h) -> bool {   // inverse of kmap; false: k has no slot
    if (k >= klimit) return false;
    if (kmap == K_SPLIT) {
      h = k >= khalf ? 1u : 0u;
      st = k - h * khalf;
    } else if (kmap == K_PAIR) {
      h = k & 1u;
      st = k >> 1;
    } else {
      const uint32_t r32 = k & 31u;
      h = (r32 >> 2) & 1u;
      st = (k >> 5) * 16u + ((r32 & 3u) | ((r32 >> 3) << 2));
    }
    return st < nsteps;
  };
  const uint32_t qpr = C >> 2;                       // quads per row of M
  const uint32_t nq = R * qpr;
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    const uint32_t e = threadIdx.x + j * blockDim.x;
    if (e >= nq) continue;
    const uint32_t rr = e / qpr, cc = (e - rr * qpr) * 4u;
    const float4 v = q.v[j];
    const float vv[4] = {v.x, v.y, v.z, v.w};
    if (!transpose) {                                // Meff[i = rr][k = cc + c]
      const uint32_t rt = rr >> 5;
      if (rt >= row_tiles) continue;
#pragma unroll
      for (uint32_t c = 0; c < 4u; ++c) {
        uint32_t st, h;
        if (step_of(cc + c, st, h)) img[(rt * nsteps + st) * 64u + (rr & 31u) + 32u * h] = vv[c];
      }
    } else {                                         // Meff[i = cc + c][k = rr]: four neighbouring lanes of one step
      const uint32_t rt = cc >> 5;
      uint32_t st, h;
      if (rt < row_tiles && step_of(rr, st, h))
        *reinterpret_cast<float4*>(img + (rt * nsteps + st) * 64u + (cc & 31u) + 32u * h) = v;
    }
  }
}

// ---- fp16 compute mode (tcnn's CutlassMLP precision: fp16 operands, fp32 accumulate; BASELINE configs[4]) ----
// v_mfma_f32_32x32x8_f16: A[i][k] lane = i + 32*(k/4), element k%4; B[k][j] lane = j + 32*(k/4), element k%4 (checked on
// the hardware: tools/mfma16_layout.hip); D as the fp32 32x32 tile.  Registers 4g..4g+3 of an accumulator are rows
// 8g + 4h + {0,1,2,3}: converted to four halfs they ARE the B operand of K-step g of the next layer, so the chaining of
// the fp32 kernels carries over with 4x fewer, 4x faster matrix instructions.  Images hold 4 halfs per lane and step
// (natural k order: k = 8s + 4h + e); accumulators, hidden activations kept for the backward, dH and the weight
// gradients stay fp32.
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ half4_t to_half4(float a, float b, float c, float d) {
  half4_t h;
  h[0] = (_Float16)a; h[1] = (_Float16)b; h[2] = (_Float16)c; h[3] = (_Float16)d;
  return h;
}

template <int MAXQ>
__device__ __forceinline__ void image_scatter16(const ImgQuads<MAXQ>& q, float* __restrict__ img_f, uint32_t R, uint32_t C,
                                                bool transpose, uint32_t row_tiles, uint32_t nsteps, uint32_t klimit) {
  _Float16* img = reinterpret_cast<_Float16*>(img_f);
  const uint32_t qpr = C >> 2;
  const uint32_t nq = R * qpr;
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    const uint32_t e = threadIdx.x + j * blockDim.x;
    if (e >= nq) continue;
    const uint32_t rr = e / qpr, cc = (e - rr * qpr) * 4u;
    const float4 v = q.v[j];
    if (!transpose) {                                // Meff[i = rr][k = cc .. cc+3]: one lane slot of one step
      const uint32_t rt = rr >> 5, st = cc >> 3;
      if (rt < row_tiles && st < nsteps && cc < klimit)
        *reinterpret_cast<half4_t*>(img + ((size_t)(rt * nsteps + st) * 64u + (rr & 31u) + 32u * ((cc >> 2) & 1u)) * 4u) =
            to_half4(v.x, v.y, v.z, v.w);
    } else {                                         // Meff[i = cc + c][k = rr]: one element of four neighbouring lanes
      const uint32_t st = rr >> 3, h = (rr >> 2) & 1u, el = rr & 3u;
      if (rr < klimit && st < nsteps) {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t c = 0; c < 4u; ++c) {
          const uint32_t i = cc + c, rt = i >> 5;
          if (rt < row_tiles) img[((size_t)(rt * nsteps + st) * 64u + (i & 31u) + 32u * h) * 4u + el] = (_Float16)vv[c];
        }
      }
    }
  }
}

__device__ __forceinline__ void lds_zero(float* __restrict__ p, uint32_t n_floats) {   // n_floats % 4 == 0
  for (uint32_t e = threadIdx.x; e < n_floats / 4u; e += blockDim.x) reinterpret_cast<float4*>(p)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// VALU-row image for the trailing (n_out % 32) output rows: imgv[(v*nsteps + s)*2 + h] = W[row0+v][chain k(s,h)]
__device__ void build_valu_image(float* __restrict__ img, const float* __restrict__ W, uint32_t row0, uint32_t nrows,
                                 uint32_t C, uint32_t nsteps) {
  const uint32_t total = nrows * nsteps * 2u;
  for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
    const uint32_t h = e & 1u;
    const uint32_t s = (e >> 1) % nsteps;
    const uint32_t v = (e >> 1) / nsteps;
    const uint32_t k = (s >> 4) * 32u + acc_row(s & 15u, h);
    img[e] = W[(size_t)(row0 + v) * C + k];
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

struct MlpShape {
  uint32_t n_in, n_out, out_pad;
  uint32_t mt;   // MFMA output row tiles
  uint32_t vr;   // trailing output rows done on the VALU
};

__host__ __device__ inline MlpShape make_shape(uint32_t n_in, uint32_t n_out) {
  MlpShape s;
  s.n_in = n_in;
  s.n_out = n_out;
  s.out_pad = (n_out + 15u) / 16u * 16u;
  s.mt = n_out / 32u;
  s.vr = n_out - 32u * s.mt;
  if (s.vr > 8u) {
    s.mt += 1;
    s.vr = 0;
  }
  return s;
}

// ---- layer 0: acc[t] = W_in[t-th 32 rows] * x  (x streamed from global as float4, split halves) ----
// The lane's half row (<= 64 floats) is fetched as two groups of eight 16-byte loads, the second group issued
// before the first group's MFMAs, so the row's memory latency is paid once, under matrix work.
struct XRegs {
  float4 a[8], b[8];                      // the lane's half row: <= 64 floats
};

__device__ __forceinline__ void x_load(XRegs& xr, const float* __restrict__ xrow, bool valid, uint32_t khalf) {
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(xrow);
  const uint32_t n4 = khalf >> 2;   // <= 16
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xr.a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid && (uint32_t)j < n4) xr.a[j] = x4[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xr.b[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid && (uint32_t)(8 + j) < n4) xr.b[j] = x4[8 + j];
  }
}

template <int NT>
__device__ __forceinline__ void layer_in_regs(const XRegs& xr, uint32_t khalf, const float* __restrict__ img, uint32_t lane,
                                              f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = zero16();
  const uint32_t n4 = khalf >> 2;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t q = half * 8 + j;
      if (q < n4) {
        const float4 xv = half ? xr.b[j] : xr.a[j];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t s = q * 4 + e;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float a = img[(t * khalf + s) * 64u + lane];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[e], acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void layer_in(const float* __restrict__ xrow, bool valid, uint32_t khalf,
                                         const float* __restrict__ img, uint32_t lane, f32x16 (&acc)[NT]) {
  XRegs xr;
  x_load(xr, xrow, valid, khalf);
  layer_in_regs<NT>(xr, khalf, img, lane, acc);
}

// ---- chained layer: out[t] = W[t-th 32 rows] * act   (act in accumulator layout = B operand) ----
// No conditionals inside: a per-MFMA `if (t < out_tiles)` (even a uniform one) makes the compiler emit
// ds_read -> s_waitcnt lgkmcnt(0) -> MFMA -> branch for every step instead of batching the LDS reads ahead of the MFMAs;
// callers pick NT_OUT by the tile count they need.
template <int NT_OUT, int NT_IN>
__device__ __forceinline__ void layer_chain(const f32x16 (&act)[NT_IN], const float* __restrict__ img, uint32_t lane,
                                            f32x16 (&out)[NT_OUT]) {
  constexpr uint32_t nsteps = NT_IN * 16;
#pragma unroll
  for (int t = 0; t < NT_OUT; ++t) out[t] = zero16();
#pragma unroll
  for (int ti = 0; ti < NT_IN; ++ti) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t s = ti * 16 + r;
      const float b = act[ti][r];
#pragma unroll
      for (int t = 0; t < NT_OUT; ++t) {
        const float a = img[(t * nsteps + s) * 64u + lane];
        out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, out[t], 0, 0, 0);
      }
    }
  }
}

// one or two output tiles (uniform choice), results in o[0..nt)
template <int NT_IN>
__device__ __forceinline__ void layer_chain_1or2(const f32x16 (&act)[NT_IN], const float* __restrict__ img, uint32_t lane,
                                                 uint32_t nt, f32x16 (&o)[2]) {
  if (nt >= 2u) {
    layer_chain<2, NT_IN>(act, img, lane, o);
  } else {
    f32x16 o1[1];
    layer_chain<1, NT_IN>(act, img, lane, o1);
    o[0] = o1[0];
    o[1] = o1[0];
  }
}

template <int NT_OUT, int NT_IN>
__device__ __forceinline__ void layer_chain16(const f32x16 (&act)[NT_IN], const float* __restrict__ img_f, uint32_t lane,
                                              f32x16 (&out)[NT_OUT]) {
  constexpr uint32_t nsteps = NT_IN * 4;
  const half4_t* __restrict__ img = reinterpret_cast<const half4_t*>(img_f);
#pragma unroll
  for (int t = 0; t < NT_OUT; ++t) out[t] = zero16();
#pragma unroll
  for (int ti = 0; ti < NT_IN; ++ti) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint32_t st = ti * 4 + g;
      const half4_t b = to_half4(act[ti][4 * g], act[ti][4 * g + 1], act[ti][4 * g + 2], act[ti][4 * g + 3]);
#pragma unroll
      for (int t = 0; t < NT_OUT; ++t) out[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(img[(t * nsteps + st) * 64u + lane], b, out[t], 0, 0, 0);
    }
  }
}

template <int NT_IN>
__device__ __forceinline__ void layer_chain16_1or2(const f32x16 (&act)[NT_IN], const float* __restrict__ img, uint32_t lane,
                                                   uint32_t nt, f32x16 (&o)[2]) {
  if (nt >= 2u) {
    layer_chain16<2, NT_IN>(act, img, lane, o);
  } else {
    f32x16 o1[1];
    layer_chain16<1, NT_IN>(act, img, lane, o1);
    o[0] = o1[0];
    o[1] = o1[0];
  }
}

// Backward values (dY, dH) can sit far below fp16's normal range (6e-5): before each fp16 conversion the wave scales its
// tile by a power of two chosen from the tile's max magnitude (exact, undone in fp32 after the matrix products) -- a
// per-tile dynamic version of tcnn's loss scaling.
__device__ __forceinline__ float wave_pow2_scale(float local_max) {
  float m = local_max;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (!(m > 0.f) || !(m < INFINITY)) return 1.0f;
  int ex;
  (void)frexpf(m, &ex);                              // m < 2^ex
  return ldexpf(1.0f, min(max(8 - ex, -100), 100));   // scaled max in [2^7, 2^8); clamped: 2^(8-ex) must stay finite
                                                      // for tiles whose largest magnitude is denormal-small
}

template <int NT>
__device__ __forceinline__ float tile_max_abs(const f32x16 (&a)[NT]) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(a[t][r]));
  return m;
}

template <int NT>
__device__ __forceinline__ void tile_scale(f32x16 (&a)[NT], float s) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[t][r] *= s;
}

template <int NT>
__device__ __forceinline__ void relu(f32x16 (&a)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[t][r] = fmaxf(a[t][r], 0.f);
}

// ---- row-coalesced global I/O of accumulator-layout tiles through a wave-private LDS staging tile ----
// In the accumulator layout a lane owns ONE point, so a direct float4 store (or load) instruction touches 64 different
// cache lines for 1 KB of payload; the texture-addresser retires ~1 line per 2 cycles, and at 16 waves per CU those
// line operations -- not MFMA, not HBM -- were what bounded the MLP kernels (measured with in-kernel time stamps: the
// next tile's first load could not issue for 30-40 us behind one tile's stores).  Staged through LDS the same 1 KB
// leaves as 8 rows x 128 contiguous bytes: 8 lines per instruction.
constexpr uint32_t STG_LD = 36;              // floats per staged row: 32 + 4 (16-byte aligned rows, spread over banks)
constexpr uint32_t STG_FLOATS = 32 * STG_LD;

__device__ __forceinline__ void wave_lds_fence() {
  // LDS instructions of one wave execute in order; this only stops the compiler from moving them across
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// a: 32 features x 32 points (accumulator layout) -> rows dst[row * ld + 0..31], row < nrows
__device__ __forceinline__ void store_tile_staged(float* __restrict__ dst, uint32_t ld, uint32_t nrows, const f32x16& a,
                                                  float* __restrict__ stg, uint32_t lane) {
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(stg + pt * STG_LD + 8 * g + 4 * h) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
  wave_lds_fence();
  const uint32_t rr = lane >> 3, c4 = 4u * (lane & 7u);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = rr + 8 * i;
    const float4 v = *reinterpret_cast<const float4*>(stg + row * STG_LD + c4);
    if (row < nrows) *reinterpret_cast<float4*>(dst + (size_t)row * ld + c4) = v;
  }
  wave_lds_fence();
}

// Row ids of the wave's 32-slot tile, kept in LDS (wave-private, STG_ROWS ints after the staging tile) so that every
// coalesced access pattern can look up "the row of tile-row i" without re-reading the index.
constexpr uint32_t STG_ROWS = 64;              // two tables: the tile being computed and the one being prefetched
constexpr uint32_t STG_WAVE_FLOATS = STG_FLOATS + STG_ROWS;

__device__ __forceinline__ void tile_rows_publish(int* __restrict__ rows_lds, const int32_t* __restrict__ row_index,
                                                  uint32_t slot0, uint32_t n_slots, bool live, uint32_t lane) {
  if (lane < 32u) {
    const uint32_t slot = slot0 + lane;
    int row = -1;
    if (live && slot < n_slots) row = row_index ? row_index[slot] : (int)slot;
    rows_lds[lane] = row;
  }
  wave_lds_fence();
}

// ---- layer 0 input: 32 columns (k = 32c .. 32c+31) of the tile's 32 rows per chunk ----
// Global side: lane (r8 = l>>3, j = l&7) reads float4 j of rows r8, r8+8, r8+16, r8+24: 8 rows x 128 contiguous bytes
// per instruction.  LDS side: written de-interleaved (even k -> [pt][0..15], odd k -> [pt][16..31]) so that the MFMA
// lane (pt, h) finds its 16 B operands of the chunk (k = 32c + 2s + h, the K_PAIR map) as 4 aligned float4s.
struct XChunk {
  float4 v[4];
};

// Optional second input segment: columns [n_in1, n_in) of the network input come from x2 (its own row stride), so that
// cat(a, b) inputs (reference models/decoder.py:123-124) are never materialised.  n_in1 % 4 == 0.
struct XSeg {
  const float* x2;
  uint32_t ldx2, n_in1;
};
struct DxSeg {
  float* dx2;
  uint32_t lddx2, acc1, acc2;              // acc: read-add-write instead of overwrite
};

__device__ __forceinline__ void x_chunk_issue(XChunk& xc, const float* __restrict__ x, uint32_t ldx, const XSeg& seg,
                                              uint32_t n_in, const int* __restrict__ rows_lds, uint32_t c, uint32_t lane) {
  const uint32_t col = 32u * c + 4u * (lane & 7u);
  const bool second = seg.x2 != nullptr && col >= seg.n_in1;
  const float* base = second ? seg.x2 + (col - seg.n_in1) : x + col;
  const uint32_t ld = second ? seg.ldx2 : ldx;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rows_lds[(lane >> 3) + 8 * i];
    xc.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row >= 0 && col < n_in) xc.v[i] = *reinterpret_cast<const float4*>(base + (size_t)row * ld);
  }
}

__device__ __forceinline__ void x_chunk_commit(const XChunk& xc, float* __restrict__ stg, uint32_t lane) {
  const uint32_t j = lane & 7u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* r = stg + ((lane >> 3) + 8 * i) * STG_LD;
    *reinterpret_cast<float2*>(r + 2 * j) = make_float2(xc.v[i].x, xc.v[i].z);
    *reinterpret_cast<float2*>(r + 16 + 2 * j) = make_float2(xc.v[i].y, xc.v[i].w);
  }
  wave_lds_fence();
}

// 16 steps of the chunk (fewer in the last one: `steps`, a multiple of 4), image in the K_PAIR map with khalf steps
template <int NT>
__device__ __forceinline__ void layer_in_chunk(const float* __restrict__ stg, const float* __restrict__ img, uint32_t khalf,
                                               uint32_t c, uint32_t steps, uint32_t lane, f32x16 (&acc)[NT]) {
  const float* src = stg + (lane & 31u) * STG_LD + 16u * (lane >> 5);
  float xs[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(src + 4 * g);
    xs[4 * g] = v.x; xs[4 * g + 1] = v.y; xs[4 * g + 2] = v.z; xs[4 * g + 3] = v.w;
  }
  wave_lds_fence();                                // the staging tile may be overwritten once these reads are issued
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if ((uint32_t)(4 * g) < steps) {               // one uniform branch per 4-step group
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t st = 16u * c + 4 * g + e;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float a = img[(t * khalf + st) * 64u + lane];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[4 * g + e], acc[t], 0, 0, 0);
        }
      }
    }
  }
}

// fp16 mode: the chunk is staged in natural column order and lane (pt, h) reads the 4 B operands (4 halfs each) of the
// chunk's <= 4 K-steps (k = 32c + 8s + 4h + e) as 4 float4s
__device__ __forceinline__ void x_chunk_commit_nat(const XChunk& xc, float* __restrict__ stg, uint32_t lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(stg + ((lane >> 3) + 8 * i) * STG_LD + 4u * (lane & 7u)) = xc.v[i];
  wave_lds_fence();
}

template <int NT>
__device__ __forceinline__ void layer_in_chunk16(const float* __restrict__ stg, const float* __restrict__ img_f, uint32_t nsteps,
                                                 uint32_t c, uint32_t steps, uint32_t lane, f32x16 (&acc)[NT]) {
  const half4_t* __restrict__ img = reinterpret_cast<const half4_t*>(img_f);
  const float* src = stg + (lane & 31u) * STG_LD + 4u * (lane >> 5);
  half4_t xs[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(src + 8 * g);
    xs[g] = to_half4(v.x, v.y, v.z, v.w);
  }
  wave_lds_fence();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if ((uint32_t)g < steps) {                      // uniform
      const uint32_t st = 4u * c + g;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(img[(t * nsteps + st) * 64u + lane], xs[g], acc[t], 0, 0, 0);
    }
  }
}

// a: 32 output features x 32 points -> y[row][col0 + f], f < ncols, rows through the tile's row table; dword stores with
// lane = feature: 2 rows x 128 bytes per instruction (any ldy / alignment)
__device__ __forceinline__ void store_tile_rows_scalar(float* __restrict__ y, uint32_t ldy, uint32_t col0, uint32_t ncols,
                                                       const int* __restrict__ rows_lds, const f32x16& a,
                                                       float* __restrict__ stg, uint32_t lane) {
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(stg + pt * STG_LD + 8 * g + 4 * h) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
  wave_lds_fence();
  const uint32_t f = lane & 31u;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t r = (lane >> 5) + 2 * i;
    const int row = rows_lds[r];
    const float v = stg[r * STG_LD + f];
    if (row >= 0 && f < ncols) y[(size_t)row * ldy + col0 + f] = v;
  }
  wave_lds_fence();
}

// 32 consecutive slot rows x 32 floats (src + row * ld) -> accumulator layout, in two halves so that several tiles'
// loads can be in flight before the first is committed
struct TileQuads {
  float4 v[4];
};

__device__ __forceinline__ void load_tile_issue(TileQuads& q, const float* __restrict__ src, uint32_t ld, uint32_t nrows,
                                                uint32_t lane) {
  const uint32_t rr = lane >> 3, c4 = 4u * (lane & 7u);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = rr + 8 * i;
    q.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) q.v[i] = *reinterpret_cast<const float4*>(src + (size_t)row * ld + c4);
  }
}

__device__ __forceinline__ void load_tile_commit(const TileQuads& q, f32x16& a, float* __restrict__ stg, uint32_t lane) {
  const uint32_t rr = lane >> 3, c4 = 4u * (lane & 7u);
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(stg + (rr + 8 * i) * STG_LD + c4) = q.v[i];
  wave_lds_fence();
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(stg + pt * STG_LD + 8 * g + 4 * h);
    a[4 * g] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
  }
  wave_lds_fence();
}

// dy columns 32c .. 32c+31 of the tile's rows -> staging, de-interleaved for the K_PAIR map (see x_chunk_commit); dword
// loads with lane = column (any lddy): 2 rows x 128 bytes per instruction
__device__ __forceinline__ void dy_chunk_stage(const float* __restrict__ dy, uint32_t lddy, uint32_t n_out,
                                               const int* __restrict__ rows_lds, uint32_t c, float* __restrict__ stg,
                                               uint32_t lane) {
  const uint32_t f = lane & 31u, col = 32u * c + f;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = rows_lds[(lane >> 5) + 2 * i];
    v[i] = (row >= 0 && col < n_out) ? dy[(size_t)row * lddy + col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) stg[((lane >> 5) + 2 * i) * STG_LD + (f & 1u) * 16u + (f >> 1)] = v[i];
  wave_lds_fence();
}

// fp16 mode: both (<= 2) 32-column chunks of the tile's dY rows are loaded first (their max picks the wave's power-of-two
// scale), then staged one at a time in natural column order
struct DyTile {
  float v[2][16];
};

__device__ __forceinline__ float dy_tile_load(DyTile& d, const float* __restrict__ dy, uint32_t lddy, uint32_t n_out,
                                              const int* __restrict__ rows_lds, uint32_t lane) {
  float m = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const uint32_t col = 32u * c + (lane & 31u);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = rows_lds[(lane >> 5) + 2 * i];
      d.v[c][i] = (row >= 0 && col < n_out) ? dy[(size_t)row * lddy + col] : 0.f;
      m = fmaxf(m, fabsf(d.v[c][i]));
    }
  }
  return m;
}

__device__ __forceinline__ void dy_tile_commit_nat(const DyTile& d, int c, float scale, float* __restrict__ stg, uint32_t lane) {
  const uint32_t f = lane & 31u;
#pragma unroll
  for (int i = 0; i < 16; ++i) stg[((lane >> 5) + 2 * i) * STG_LD + f] = (c ? d.v[1][i] : d.v[0][i]) * scale;
  wave_lds_fence();
}

// a: 32 features x 32 points -> dst[row][col0 + f] (f < ncols, ncols % 4 == 0, 16-byte aligned rows), rows through the
// tile's row table; optional read-add-write
__device__ __forceinline__ void store_tile_rows_vec(float* __restrict__ dst1, uint32_t ld1, const DxSeg& seg, uint32_t n_in1,
                                                    uint32_t col0, uint32_t ncols, const int* __restrict__ rows_lds,
                                                    const f32x16& a, float* __restrict__ stg, uint32_t lane) {
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(stg + pt * STG_LD + 8 * g + 4 * h) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
  wave_lds_fence();
  const uint32_t rr = lane >> 3, c4 = 4u * (lane & 7u);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t r = rr + 8 * i;
    const int row = rows_lds[r];
    float4 v = *reinterpret_cast<const float4*>(stg + r * STG_LD + c4);
    if (row >= 0 && c4 < ncols) {
      const uint32_t col = col0 + c4;
      const bool second = seg.dx2 != nullptr && col >= n_in1;
      float4* p = second ? reinterpret_cast<float4*>(seg.dx2 + (size_t)row * seg.lddx2 + (col - n_in1))
                         : reinterpret_cast<float4*>(dst1 + (size_t)row * ld1 + col);
      if (second ? seg.acc2 : seg.acc1) {
        const float4 u = *p;
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      *p = v;
    }
  }
  wave_lds_fence();
}

// same, dword accesses with lane = feature (any ld / ncols)
__device__ __forceinline__ void store_tile_rows_scalar_acc(float* __restrict__ dst1, uint32_t ld1, const DxSeg& seg,
                                                           uint32_t n_in1, uint32_t col0, uint32_t ncols,
                                                           const int* __restrict__ rows_lds, const f32x16& a,
                                                           float* __restrict__ stg, uint32_t lane) {
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(stg + pt * STG_LD + 8 * g + 4 * h) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
  wave_lds_fence();
  const uint32_t f = lane & 31u;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t r = (lane >> 5) + 2 * i;
    const int row = rows_lds[r];
    float v = stg[r * STG_LD + f];
    if (row >= 0 && f < ncols) {
      const uint32_t col = col0 + f;
      const bool second = seg.dx2 != nullptr && col >= n_in1;
      float* p = second ? seg.dx2 + (size_t)row * seg.lddx2 + (col - n_in1) : dst1 + (size_t)row * ld1 + col;
      if (second ? seg.acc2 : seg.acc1) v += *p;
      *p = v;
    }
  }
  wave_lds_fence();
}

// LDS layout helpers (floats)
template <int NN, int NL>
struct FwdLds {
  static __host__ __device__ uint32_t img_in(uint32_t) { return 0; }
  static __host__ __device__ uint32_t img_h(uint32_t n_in) { return NN * n_in; }
  static __host__ __device__ uint32_t img_out(uint32_t n_in) { return NN * n_in + (NL - 1) * NN * NN; }
  static __host__ __device__ uint32_t img_valu(uint32_t n_in, uint32_t mt) { return img_out(n_in) + mt * 32 * NN; }
  static __host__ __device__ uint32_t total(uint32_t n_in, uint32_t mt, uint32_t vr) { return img_valu(n_in, mt) + vr * NN; }
};

template <int NN, int NL, bool F16>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(const float* __restrict__ x, uint32_t ldx, XSeg seg,
                                                      const float* __restrict__ params, MlpShape sh,
                                                      float* __restrict__ y, uint32_t ldy, uint32_t n_slots,
                                                      const int32_t* __restrict__ row_index,
                                                      const int32_t* __restrict__ tile_group, uint32_t param_stride,
                                                      uint32_t tiles_per_block, float* __restrict__ h_save) {
  constexpr int NT = NN / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using L = FwdLds<NN, NL>;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t khalf = sh.n_in / 2;
  const uint32_t n_chunks = (sh.n_in + 31u) / 32u;
  const uint32_t n_btiles = (n_slots + 127u) / 128u;
  const uint32_t bt0 = blockIdx.x * tiles_per_block;
  const uint32_t bt1 = min(bt0 + tiles_per_block, n_btiles);
  float* stg = lds + L::total(sh.n_in, sh.mt, sh.vr) + wave * STG_WAVE_FLOATS;
  int* rows_all = reinterpret_cast<int*>(stg + STG_FLOATS);
  XChunk xc[4];                                  // n_in <= 128: at most 4 chunks of 32 columns
  bool have_x = false;
  uint32_t cur_buf = 0;
  int cur_group = -2;
  uint32_t stamp = 0;
  (void)stamp;
  DNS_STAMP(stamp++);
  for (uint32_t bt = bt0; bt < bt1; ++bt) {
    const int grp = tile_group ? tile_group[bt] : 0;
    if (grp != cur_group) {
      __syncthreads();
      if (grp >= 0) {
        const float* pw = params + (size_t)grp * param_stride;
        const float* wout = pw + NN * sh.n_in + (NL - 1) * NN * NN;
        const uint32_t out_rows = min(sh.n_out, sh.mt * 32u);          // rows of W_out done on the matrix cores
        ImgQuads<8> q_in;                                              // 64 x 128 / 4 / 256 threads
        ImgQuads<4> q_h, q_out;
        image_load(q_in, pw, NN, sh.n_in);
        if (NL == 2) image_load(q_h, pw + NN * sh.n_in, NN, NN);
        if (sh.mt) image_load(q_out, wout, out_rows, NN);
        lds_zero(lds, L::total(sh.n_in, sh.mt, sh.vr));
        __syncthreads();
        if (F16) {
          image_scatter16(q_in, lds + L::img_in(sh.n_in), NN, sh.n_in, false, NT, sh.n_in / 8u, sh.n_in);
          if (NL == 2) image_scatter16(q_h, lds + L::img_h(sh.n_in), NN, NN, false, NT, NT * 4, NN);
          if (sh.mt) image_scatter16(q_out, lds + L::img_out(sh.n_in), out_rows, NN, false, sh.mt, NT * 4, NN);
        } else {
          image_scatter(q_in, lds + L::img_in(sh.n_in), NN, sh.n_in, false, NT, khalf, K_PAIR, 0, sh.n_in);
          if (NL == 2) image_scatter(q_h, lds + L::img_h(sh.n_in), NN, NN, false, NT, NN / 2, K_CHAIN, 0, NN);
          if (sh.mt) image_scatter(q_out, lds + L::img_out(sh.n_in), out_rows, NN, false, sh.mt, NN / 2, K_CHAIN, 0, NN);
        }
        if (sh.vr) build_valu_image(lds + L::img_valu(sh.n_in, sh.mt), wout, sh.mt * 32, sh.vr, NN, NN / 2);
      }
      cur_group = grp;
      __syncthreads();
      DNS_STAMP(stamp++);
    }
    if (grp < 0) continue;
    const uint32_t slot0 = bt * 128u + wave * 32u;
    const uint32_t nrows = slot0 < n_slots ? min(32u, n_slots - slot0) : 0u;
    // The tile's x rows (all <= 4 chunks of 32 columns) were requested while the PREVIOUS tile ran its later layers;
    // only a workgroup's first live tile pays the latency here.
    if (have_x) {
      cur_buf ^= 1u;
    } else {
      tile_rows_publish(rows_all + 32u * cur_buf, row_index, slot0, n_slots, true, lane);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((uint32_t)c < n_chunks) x_chunk_issue(xc[c], x, ldx, seg, sh.n_in, rows_all + 32u * cur_buf, c, lane);
    }
    const int* rows_lds = rows_all + 32u * cur_buf;

    // layer 0: x goes through the staging tile in 32-column chunks
    f32x16 a0[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) a0[t] = zero16();
    DNS_STAMP(stamp++);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if ((uint32_t)c < n_chunks) {
        if (F16) {
          x_chunk_commit_nat(xc[c], stg, lane);
          layer_in_chunk16<NT>(stg, lds + L::img_in(sh.n_in), sh.n_in / 8u, c, min(4u, sh.n_in / 8u - 4u * c), lane, a0);
        } else {
          x_chunk_commit(xc[c], stg, lane);
          DNS_STAMP(stamp++);
          layer_in_chunk<NT>(stg, lds + L::img_in(sh.n_in), khalf, c, min(16u, khalf - 16u * c), lane, a0);
          DNS_STAMP(stamp++);
        }
      }
    }
    have_x = false;
    for (uint32_t nbt = bt + 1; nbt < bt1; ++nbt) {
      if ((tile_group ? tile_group[nbt] : 0) < 0) continue;
      int* nrows_lds = rows_all + 32u * (cur_buf ^ 1u);
      tile_rows_publish(nrows_lds, row_index, nbt * 128u + wave * 32u, n_slots, true, lane);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((uint32_t)c < n_chunks) x_chunk_issue(xc[c], x, ldx, seg, sh.n_in, nrows_lds, c, lane);
      have_x = true;
      break;
    }
    DNS_STAMP(stamp++);
    relu<NT>(a0);
    if (h_save) {                                  // kept for the backward
#pragma unroll
      for (int t = 0; t < NT; ++t) store_tile_staged(h_save + (size_t)slot0 * NN + t * 32, NN, nrows, a0[t], stg, lane);
    }
    DNS_STAMP(stamp++);
    f32x16 a1[NT];
    if (NL == 2) {
      if (F16) layer_chain16<NT, NT>(a0, lds + L::img_h(sh.n_in), lane, a1);
      else layer_chain<NT, NT>(a0, lds + L::img_h(sh.n_in), lane, a1);
      relu<NT>(a1);
      if (h_save) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          store_tile_staged(h_save + (size_t)(n_slots + slot0) * NN + t * 32, NN, nrows, a1[t], stg, lane);
      }
    }
    DNS_STAMP(stamp++);
    const f32x16(&hl)[NT] = (NL == 2) ? a1 : a0;
    if (sh.mt) {
      f32x16 o[2];
      if (F16) layer_chain16_1or2<NT>(hl, lds + L::img_out(sh.n_in), lane, sh.mt, o);
      else layer_chain_1or2<NT>(hl, lds + L::img_out(sh.n_in), lane, sh.mt, o);
      store_tile_rows_scalar(y, ldy, 0, min(32u, sh.n_out), rows_lds, o[0], stg, lane);
      if (sh.mt > 1) store_tile_rows_scalar(y, ldy, 32, sh.n_out - 32u, rows_lds, o[1], stg, lane);
    }
    DNS_STAMP(stamp++);
    if (sh.vr) {
      const uint32_t h = lane >> 5;
      const int row = rows_lds[lane & 31u];
      const float* iv = lds + L::img_valu(sh.n_in, sh.mt);
      for (uint32_t v = 0; v < sh.vr; ++v) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += iv[(v * (NN / 2) + t * 16 + r) * 2 + h] * hl[t][r];
        sum += __shfl_xor(sum, 32);
        if (row >= 0 && h == 0) y[(size_t)row * ldy + sh.mt * 32 + v] = sum;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, data path: recompute hidden activations, dH_l = W^T dH_{l+1} (.) relu', dX = W_in^T dH_1.
// Writes H_l and dH_l (slot-major [n_slots, NN]) to the workspace for the weight-gradient GEMMs.
// ws layout: [H_1 | dH_1 | H_2 | dH_2] each n_slots*NN floats.
// SAVED = the forward kept its hidden activations (h_saved): no recompute, so the forward images W_in / W_h are
// not staged (LDS 100 KB -> ~57 KB) and 144 of the 338 MFMAs per tile (2x64, 80 in) disappear.
template <int NN, int NL, bool SAVED>
struct BwdLds {
  static __host__ __device__ uint32_t in_pad(uint32_t n_in) { return (n_in + 31u) / 32u * 32u; }
  static __host__ __device__ uint32_t ko2(uint32_t n_out) { return (n_out + 1u) / 2u; }
  static __host__ __device__ uint32_t ko2p(uint32_t n_out) { return (ko2(n_out) + 3u) & ~3u; }   // image steps: 4-step chunks
  static __host__ __device__ uint32_t img_in(uint32_t) { return 0; }
  static __host__ __device__ uint32_t img_h(uint32_t n_in) { return SAVED ? 0 : NN * n_in; }
  static __host__ __device__ uint32_t img_outT(uint32_t n_in) { return SAVED ? 0 : img_h(n_in) + (NL - 1) * NN * NN; }
  static __host__ __device__ uint32_t img_hT(uint32_t n_in, uint32_t n_out) { return img_outT(n_in) + NN * 2 * ko2p(n_out); }
  static __host__ __device__ uint32_t img_inT(uint32_t n_in, uint32_t n_out) { return img_hT(n_in, n_out) + (NL - 1) * NN * NN; }
  static __host__ __device__ uint32_t total(uint32_t n_in, uint32_t n_out, bool need_dx) {
    return img_inT(n_in, n_out) + (need_dx ? in_pad(n_in) * NN : 0);
  }
};

template <int NN, int NL, bool SAVED, bool F16>
__global__ __launch_bounds__(512) void mlp_bwd_data_kernel(const float* __restrict__ x, uint32_t ldx,
                                                           const float* __restrict__ dy, uint32_t lddy,
                                                           const float* __restrict__ params, MlpShape sh,
                                                           float* __restrict__ dx, uint32_t lddx, DxSeg dseg, uint32_t n_in1,
                                                           float* __restrict__ ws, uint32_t n_slots,
                                                           const int32_t* __restrict__ row_index,
                                                           const int32_t* __restrict__ tile_group,
                                                           uint32_t param_stride, uint32_t tiles_per_block,
                                                           const float* __restrict__ h_saved, uint32_t h_stride) {
  constexpr int NT = NN / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using L = BwdLds<NN, NL, SAVED>;
  // 8 waves share one set of LDS weight images (they cap the CU at one workgroup): waves 0-3 take one 128-slot
  // tile, waves 4-7 the next one when it belongs to the same weight set -- two waves per SIMD hide each other's
  // LDS / MFMA latencies.
  const uint32_t lane = threadIdx.x & 63u, wave8 = threadIdx.x >> 6;
  const uint32_t wave = wave8 & 3u, wsel = wave8 >> 2;
  const uint32_t khalf = sh.n_in / 2;
  const uint32_t ko2 = L::ko2(sh.n_out), ko2p = L::ko2p(sh.n_out);
  const uint32_t ns16 = (sh.n_out + 7u) / 8u;      // fp16 mode: K-steps of 8 outputs
  const uint32_t in_tiles = L::in_pad(sh.n_in) / 32u;
  const uint32_t n_btiles = (n_slots + 127u) / 128u;
  const uint32_t bt0 = blockIdx.x * tiles_per_block;
  const uint32_t bt1 = min(bt0 + tiles_per_block, n_btiles);
  // workspace: recompute path [H1 | dH1 | H2 | dH2]; saved path [dH1 | dH2] (H stays in h_saved)
  float* wsH1 = SAVED ? nullptr : ws;
  float* wsD1 = SAVED ? ws : ws + (size_t)n_slots * NN;
  float* wsH2 = SAVED ? nullptr : ws + (size_t)2 * n_slots * NN;
  float* wsD2 = SAVED ? ws + (size_t)n_slots * NN : ws + (size_t)3 * n_slots * NN;
  float* stg = lds + L::total(sh.n_in, sh.n_out, dx != nullptr) + wave8 * STG_WAVE_FLOATS;
  int* rows_lds = reinterpret_cast<int*>(stg + STG_FLOATS);
  (void)ko2;
  int cur_group = -2;
  uint32_t stamp = 0;
  (void)stamp;
  DNS_STAMP(stamp++);
  for (uint32_t btb = bt0; btb < bt1;) {
    const int grp = tile_group ? tile_group[btb] : 0;
    const uint32_t nb = (btb + 1 < bt1 && (tile_group ? tile_group[btb + 1] : 0) == grp) ? 2u : 1u;
    const uint32_t bt = btb + wsel;
    btb += nb;
    if (grp != cur_group) {
      __syncthreads();
      if (grp >= 0) {
        const float* pw = params + (size_t)grp * param_stride;
        const float* wh = pw + NN * sh.n_in;
        const float* wout = wh + (NL - 1) * NN * NN;
        ImgQuads<4> q_in;                                              // 64 x 128 / 4 / 512 threads
        ImgQuads<2> q_h, q_out;
        image_load(q_in, pw, NN, sh.n_in);                             // serves both W_in and W_in^T
        if (NL == 2) image_load(q_h, wh, NN, NN);
        image_load(q_out, wout, sh.n_out, NN);
        lds_zero(lds, L::total(sh.n_in, sh.n_out, dx != nullptr));
        __syncthreads();
        if (!SAVED) image_scatter(q_in, lds + L::img_in(sh.n_in), NN, sh.n_in, false, NT, khalf, K_SPLIT, khalf, sh.n_in);
        if (F16) {                                 // SAVED only (host-checked): transposed images, natural k order
          if (NL == 2) image_scatter16(q_h, lds + L::img_hT(sh.n_in, sh.n_out), NN, NN, true, NT, NT * 4, NN);
          image_scatter16(q_out, lds + L::img_outT(sh.n_in), sh.n_out, NN, true, NT, ns16, sh.n_out);
          if (dx) image_scatter16(q_in, lds + L::img_inT(sh.n_in, sh.n_out), NN, sh.n_in, true, in_tiles, NT * 4, NN);
        } else {
          if (NL == 2) {
            if (!SAVED) image_scatter(q_h, lds + L::img_h(sh.n_in), NN, NN, false, NT, NN / 2, K_CHAIN, 0, NN);
            image_scatter(q_h, lds + L::img_hT(sh.n_in, sh.n_out), NN, NN, true, NT, NN / 2, K_CHAIN, 0, NN);
          }
          // A = W_out^T: rows = hidden (NN), k over outputs in pairs (k = 2s + h)
          image_scatter(q_out, lds + L::img_outT(sh.n_in), sh.n_out, NN, true, NT, ko2p, K_PAIR, 0, sh.n_out);
          if (dx) image_scatter(q_in, lds + L::img_inT(sh.n_in, sh.n_out), NN, sh.n_in, true, in_tiles, NN / 2, K_CHAIN, 0, NN);
        }
      }
      cur_group = grp;
      __syncthreads();
      DNS_STAMP(stamp++);
    }
    if (wsel >= nb) continue;                      // no second tile of this weight set: waves 4-7 sit this one out
    const uint32_t slot0 = bt * 128u + wave * 32u;
    const uint32_t nrows = slot0 < n_slots ? min(32u, n_slots - slot0) : 0u;
    if (grp < 0) {
      // padding tile of an empty group: zero its workspace rows so the weight GEMMs see zeros
      const f32x16 z = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!SAVED) store_tile_staged(wsH1 + (size_t)slot0 * NN + t * 32, NN, nrows, z, stg, lane);
        store_tile_staged(wsD1 + (size_t)slot0 * NN + t * 32, NN, nrows, z, stg, lane);
        if (NL == 2) {
          if (!SAVED) store_tile_staged(wsH2 + (size_t)slot0 * NN + t * 32, NN, nrows, z, stg, lane);
          store_tile_staged(wsD2 + (size_t)slot0 * NN + t * 32, NN, nrows, z, stg, lane);
        }
      }
      continue;
    }
    tile_rows_publish(rows_lds, row_index, slot0, n_slots, true, lane);
    const uint32_t h = lane >> 5;
    f32x16 h1[NT], h2[NT];
    if (SAVED) {
      // all <= 4 hidden tiles requested at once (row-coalesced), committed to the accumulator layout one by one
      TileQuads qh[NL * NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        load_tile_issue(qh[t], h_saved + (size_t)slot0 * NN + t * 32, NN, nrows, lane);
        if (NL == 2) load_tile_issue(qh[NT + t], h_saved + (size_t)(h_stride + slot0) * NN + t * 32, NN, nrows, lane);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        load_tile_commit(qh[t], h1[t], stg, lane);
        if (NL == 2) load_tile_commit(qh[NT + t], h2[t], stg, lane);
      }
    } else {
      const int row = rows_lds[lane & 31u];
      const bool valid = row >= 0;
      const float* xrow = x + (size_t)(valid ? row : 0) * ldx + h * khalf;
      layer_in<NT>(xrow, valid, khalf, lds + L::img_in(sh.n_in), lane, h1);
      relu<NT>(h1);
      if (NL == 2) {
        layer_chain<NT, NT>(h1, lds + L::img_h(sh.n_in), lane, h2);
        relu<NT>(h2);
      }
    }
    DNS_STAMP(stamp++);   // h loads issued / recompute done
    // dH_last = W_out^T dY: dy staged in 32-column chunks (k = 32c + 2s + h)
    f32x16 dl[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) dl[t] = zero16();
    {
      const float* img = lds + L::img_outT(sh.n_in);
      const uint32_t n_dy_chunks = (sh.n_out + 31u) / 32u;
      if (F16) {
        DyTile dt;
        const float sc = wave_pow2_scale(dy_tile_load(dt, dy, lddy, sh.n_out, rows_lds, lane));
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if ((uint32_t)c < n_dy_chunks) {
            dy_tile_commit_nat(dt, c, sc, stg, lane);
            layer_in_chunk16<NT>(stg, img, ns16, c, min(4u, ns16 - 4u * c), lane, dl);
          }
        }
        tile_scale<NT>(dl, 1.0f / sc);
      } else {
        for (uint32_t c = 0; c < n_dy_chunks; ++c) {
          dy_chunk_stage(dy, lddy, sh.n_out, rows_lds, c, stg, lane);
          layer_in_chunk<NT>(stg, img, ko2p, c, min(16u, ko2p - 16u * c), lane, dl);
        }
      }
    }
    DNS_STAMP(stamp++);   // dl mfma issued
    const f32x16(&hl)[NT] = (NL == 2) ? h2 : h1;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dl[t][r] = hl[t][r] > 0.f ? dl[t][r] : 0.f;
    f32x16 d1[NT];
    if (NL == 2) {
      if (F16) {
        const float sc = wave_pow2_scale(tile_max_abs<NT>(dl));
        f32x16 dls[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) dls[t] = dl[t] * sc;
        layer_chain16<NT, NT>(dls, lds + L::img_hT(sh.n_in, sh.n_out), lane, d1);
        tile_scale<NT>(d1, 1.0f / sc);
      } else {
        layer_chain<NT, NT>(dl, lds + L::img_hT(sh.n_in, sh.n_out), lane, d1);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[t][r] = h1[t][r] > 0.f ? d1[t][r] : 0.f;
    }
    DNS_STAMP(stamp++);   // relu' + chain d1
    const f32x16(&dfirst)[NT] = (NL == 2) ? d1 : dl;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!SAVED) store_tile_staged(wsH1 + (size_t)slot0 * NN + t * 32, NN, nrows, h1[t], stg, lane);
      store_tile_staged(wsD1 + (size_t)slot0 * NN + t * 32, NN, nrows, dfirst[t], stg, lane);
      if (NL == 2) {
        if (!SAVED) store_tile_staged(wsH2 + (size_t)slot0 * NN + t * 32, NN, nrows, h2[t], stg, lane);
        store_tile_staged(wsD2 + (size_t)slot0 * NN + t * 32, NN, nrows, dl[t], stg, lane);
      }
    }
    DNS_STAMP(stamp++);   // ws stores
    if (dx) {
      // dX = W_in^T dH_1, two input tiles at a time to bound accumulator registers
      const float* img = lds + L::img_inT(sh.n_in, sh.n_out);
      f32x16 dfs[NT];                                // fp16 mode: dH_1 scaled into fp16's range once for all input tiles
      float inv_dfs = 1.0f;
      if (F16) {
        const float sc = wave_pow2_scale(tile_max_abs<NT>(dfirst));
#pragma unroll
        for (int t = 0; t < NT; ++t) dfs[t] = dfirst[t] * sc;
        inv_dfs = 1.0f / sc;
      }
      const bool vec = ((lddx & 3u) == 0) && ((((uintptr_t)dx) & 15u) == 0) &&
                       (!dseg.dx2 || (((dseg.lddx2 & 3u) == 0) && ((((uintptr_t)dseg.dx2) & 15u) == 0)));
      for (uint32_t it0 = 0; it0 < in_tiles; it0 += 2) {
        f32x16 o[2];
        const uint32_t nt = min(2u, in_tiles - it0);
        if (F16) {
          // image tile stride: NT*4 steps x 64 lanes x 4 halfs = NT*512 floats
          layer_chain16_1or2<NT>(dfs, img + (size_t)it0 * (NT * 4) * 64u * 2u, lane, nt, o);
#pragma unroll
          for (int t = 0; t < 2; ++t) o[t] *= inv_dfs;
        } else {
          layer_chain_1or2<NT>(dfirst, img + (size_t)it0 * (NN / 2) * 64u, lane, nt, o);
        }
        DNS_STAMP(stamp++);   // dX chain issued
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if ((uint32_t)t < nt) {
            const uint32_t col0 = (it0 + t) * 32u;
            const uint32_t ncols = min(32u, sh.n_in - col0);           // n_in % 8 == 0
            if (vec) store_tile_rows_vec(dx, lddx, dseg, n_in1, col0, ncols, rows_lds, o[t], stg, lane);
            else store_tile_rows_scalar_acc(dx, lddx, dseg, n_in1, col0, ncols, rows_lds, o[t], stg, lane);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradients: C[M x N] += sum_p A[p][m] * B[p][n]   (K dimension = points), per parameter group.
// A operand lane (i,h) reads A[p0+2s+h][mt*32+i]; B operand lane (j,h) reads B[p0+2s+h][nt*32+j]: both are
// 128-byte row segments straight from global/L2 -- no LDS.  Each wave owns up to 2 of the <=8 output tiles.
struct GemmTnArgs {
  const float* A;
  uint32_t lda;
  const int32_t* a_index;  // optional slot -> row (-1 = zero row)
  uint32_t M;
  const float* B;
  uint32_t ldb;
  const int32_t* b_index;
  uint32_t N;
  float* C;                // [M x ldc] (+ group*c_stride)
  uint32_t ldc;
  uint32_t c_stride;
  uint32_t n_slots;
  const int32_t* tile_group;  // per 128 slots
  uint32_t tiles_per_block;
};

// Staging of 64 rows x Wp columns (Wp = 32..128, multiple of 32) of a row-major matrix into LDS, split in two
// halves so the global loads of sub-tile k+1 fly while the MFMAs of sub-tile k run: stage_load issues every
// 16-byte load of the thread (4 threads per row, Wp/4 consecutive columns each) into registers, stage_store
// writes them to LDS after the barrier that retires the previous sub-tile.
struct StageRegs {
  float4 v[8];
};

__device__ __forceinline__ void stage_load(StageRegs& rg, uint32_t Wp, const float* __restrict__ src, uint32_t ld,
                                           const int32_t* __restrict__ index, uint32_t W, uint32_t p0, uint32_t n_slots) {
  const uint32_t r = threadIdx.x >> 2, q = threadIdx.x & 3u;
  const uint32_t cw = Wp >> 2;               // columns per thread: 8, 16, 24 or 32
  const uint32_t c0 = q * cw;
  const uint32_t slot = p0 + r;
  int row = -1;
  if (slot < n_slots) row = index ? index[slot] : (int)slot;
  const float* sp = src + (size_t)(row >= 0 ? row : 0) * ld;
  const bool vec = ((ld & 3u) == 0) && ((((uintptr_t)src) & 15u) == 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    rg.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t c = c0 + 4 * j;
    if ((uint32_t)(4 * j) < cw && row >= 0) {
      if (vec && c + 3 < W) {
        rg.v[j] = *reinterpret_cast<const float4*>(sp + c);
      } else {
        if (c < W) rg.v[j].x = sp[c];
        if (c + 1 < W) rg.v[j].y = sp[c + 1];
        if (c + 2 < W) rg.v[j].z = sp[c + 2];
        if (c + 3 < W) rg.v[j].w = sp[c + 3];
      }
    }
  }
}

__device__ __forceinline__ void stage_store(const StageRegs& rg, float* __restrict__ dst, uint32_t Wp) {
  const uint32_t r = threadIdx.x >> 2, q = threadIdx.x & 3u;
  const uint32_t cw = Wp >> 2;
  const uint32_t c0 = q * cw;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if ((uint32_t)(4 * j) < cw) *reinterpret_cast<float4*>(dst + r * Wp + c0 + 4 * j) = rg.v[j];
}

struct GemmTnBatch {
  GemmTnArgs g[4];
};

// Weight gradients C[M x N] += sum_slots A[slot][m] * B[slot][n] (K dimension = points), per parameter group.
// The workgroup stages a 64-slot sub-tile of A [64 x M] and B [64 x N] through LDS (coalesced, through the optional
// slot->row index) and all waves read their MFMA operands from it -- lane (i,h) of step s reads row 2s+h, column
// tile*32+i: two 128-byte row segments per ds_read_b32, conflict-free for any row stride.  Each wave owns up to 2
// of the <= 8 output tiles; blockIdx.y selects one of up to three GEMMs (dW_in, dW_hidden, dW_out) of one network.
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GemmTnArgs& g = batch.g[blockIdx.y];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t M = g.M, N = g.N;
  const uint32_t MT = (M + 31u) / 32u, NT = (N + 31u) / 32u;
  const uint32_t Mp = MT * 32u, Np = NT * 32u;      // LDS row strides (padded columns are zero-filled)
  float* la = lds;
  float* lb = lds + 64u * Mp;
  const uint32_t ntiles = MT * NT;
  const uint32_t n_btiles = (g.n_slots + 127u) / 128u;
  const uint32_t bt0 = blockIdx.x * g.tiles_per_block;
  const uint32_t bt1 = min(bt0 + g.tiles_per_block, n_btiles);
  const uint32_t i = lane & 31u, h = lane >> 5;
  f32x16 acc[2];
  acc[0] = zero16();
  acc[1] = zero16();
  const uint32_t tile_id[2] = {wave, wave + 4u};
  int cur_group = -2;
  auto flush = [&](int grp) {
    if (grp < 0) return;
    float* C = g.C + (size_t)grp * g.c_stride;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (tile_id[t] < ntiles) {
        const uint32_t mt = tile_id[t] / NT, nt = tile_id[t] % NT;
        const uint32_t col = nt * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t row = mt * 32 + acc_row(r, h);
          if (row < M && col < N) atomicAdd(C + (size_t)row * g.ldc + col, acc[t][r]);
        }
      }
      acc[t] = zero16();
    }
  };
  // sub-tile sequence of this workgroup: (bt, sub) pairs, skipping tiles of empty groups
  auto group_of = [&](uint32_t bt) -> int { return g.tile_group ? g.tile_group[bt] : 0; };
  auto next_live = [&](uint32_t& bt, uint32_t& sub) -> bool {   // advance to the next sub-tile that exists
    while (bt < bt1) {
      if (group_of(bt) >= 0 && bt * 128u + sub * 64u < g.n_slots) return true;
      ++bt;
      sub = 0;
    }
    return false;
  };
  uint32_t bt = bt0, sub = 0;
  StageRegs ra, rb;
  bool have = next_live(bt, sub);
  if (have) {
    stage_load(ra, Mp, g.A, g.lda, g.a_index, M, bt * 128u + sub * 64u, g.n_slots);
    stage_load(rb, Np, g.B, g.ldb, g.b_index, N, bt * 128u + sub * 64u, g.n_slots);
  }
  while (have) {
    const int grp = group_of(bt);
    if (grp != cur_group) {
      flush(cur_group);
      cur_group = grp;
    }
    __syncthreads();                         // previous sub-tile's MFMA reads are done
    stage_store(ra, la, Mp);
    stage_store(rb, lb, Np);
    __syncthreads();
    // request the next sub-tile before computing on this one
    uint32_t nbt = bt, nsub = sub + 1;
    if (nsub == 2) {
      nsub = 0;
      ++nbt;
    }
    const bool more = next_live(nbt, nsub);
    if (more) {
      stage_load(ra, Mp, g.A, g.lda, g.a_index, M, nbt * 128u + nsub * 64u, g.n_slots);
      stage_load(rb, Np, g.B, g.ldb, g.b_index, N, nbt * 128u + nsub * 64u, g.n_slots);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (tile_id[t] < ntiles) {
        const uint32_t mt = tile_id[t] / NT, nt = tile_id[t] % NT;
        const float* pa = la + h * Mp + mt * 32 + i;
        const float* pb = lb + h * Np + nt * 32 + i;
#pragma unroll 8
        for (uint32_t st = 0; st < 32u; ++st)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * st * Mp], pb[2 * st * Np], acc[t], 0, 0, 0);
      }
    }
    bt = nbt;
    sub = nsub;
    have = more;
  }
  flush(cur_group);
}

// ------------------------------------------------------------------------------------------------
// Weight gradients, direct-operand form (default).  v_mfma_f32_16x16x4_f32 takes A[16 x 4] with lane l holding
// A[i = l&15][k = l>>4] and B[4 x 16] with lane l holding B[k = l>>4][j = l&15]; the K axis is the point axis.  A wave
// that reads one float4 per lane from a row-major [points x 64] matrix -- lane l: point p0 + (l>>4), channels
// 4*(l&15) .. +3, i.e. 1 KB contiguous per load -- holds in component c of that float4 exactly the A (or B) operand
// of the 16-channel subset S_c = {4i + c}.  So operands go from global memory straight into the MFMA with no LDS
// staging, no transposes and no barriers: acc[ca][cb] += mfma(a.comp[ca], b.comp[cb]) accumulates dW[4i+ca][4j+cb]
// over the wave's points, 16 MFMAs per (64 x 64) block of dW per 4 points.
//
// Work split: a "unit" is one <= 64 x <= 64 block of one weight matrix (dW_in splits in two column halves when
// n_in > 64, so a network has up to 4 units: dW_in lo/hi, dW_hidden, dW_out -- 16 MFMAs per step each, i.e. balanced).
// ONE launch per network; wave w of every workgroup owns unit w and sweeps all slots of the workgroup's tiles, so the
// four SIMDs of a CU run the four units side by side, the per-launch fixed cost (ramp, prologue, flush) is paid once
// instead of once per matrix, and no cross-wave reduction exists.  Loads are issued D steps ahead; the flush goes
// through a wave-private LDS transpose so that each atomic instruction covers 256 contiguous bytes of one dW row.
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t GEMM_MAX_TPB = 64;        // 128-slot tiles per workgroup (their weight-set ids are staged in LDS)

struct GemmUnit {
  const float* A;          // [slots or rows x lda], M <= 64 channels -> rows of dW
  const float* B;          // [slots or rows x ldb], N <= 64 channels (pre-offset to the unit's first column)
  const int32_t* a_index;  // optional slot -> row (-1 = padding)
  const int32_t* b_index;
  float* C;                // dW block origin (+ group * c_stride)
  uint32_t lda, ldb, ldc, M, N;
  uint32_t kind;           // bit 0: a_index, bit 1: b_index, bit 2: A rows are float4-addressable
};

struct GemmRoles {
  GemmUnit u[4];
  const int32_t* tile_group;  // per 128 slots (NULL: one weight set)
  uint32_t n_units, n_slots, tiles_per_block, c_stride, strided;
};

// The loop body is branch-free around its loads on purpose: every load is unconditional on a clamped address (and
// the A operand zeroed at use when the point is not real), so the compiler's vmcnt bookkeeping keeps D steps of data
// loads plus the NEXT round's index loads in flight instead of draining the queue at a conditional.
template <bool AIDX, bool BIDX, bool AVEC, bool BLOCKRED = false>
__device__ __forceinline__ void gemm_unit_run(const GemmUnit& u, uint32_t n_slots, uint32_t bt0, uint32_t tstride, uint32_t ntl,
                                              const int* __restrict__ grp_lds, float* __restrict__ wlds, uint32_t c_stride,
                                              uint32_t sshift = 5u, uint32_t soff = 0u) {
  // sshift / soff: this wave's share of a tile's 32 steps -- all of them (5, 0), or the 8 steps [soff, soff + 8) of EVERY
  // tile (3, 8 * wave): the waves of a workgroup then split each tile and carry exactly the same load (per-class launch)
  constexpr int D = 8;                       // load steps in flight (measured: 8 beats 4 and 2 here, 263 vs 283 us)
  const uint32_t lane = threadIdx.x & 63u;
  const float* __restrict__ A = u.A;
  const float* __restrict__ B = u.B;
  const int32_t* __restrict__ ai = u.a_index;
  const int32_t* __restrict__ bi = u.b_index;
  const uint32_t lda = u.lda, ldb = u.ldb, M = u.M, N = u.N;
  const uint32_t nsteps = ntl << sshift;     // 32 (or 8) steps of 4 points per 128-slot tile
  const uint32_t smask = (1u << sshift) - 1u;
  const uint32_t k = lane >> 4, c4 = 4u * (lane & 15u);
  // Columns past M / N are CLAMPED, not zeroed: D[i][j] depends on A row i and B column j only, so whatever such a lane
  // reads lands in output rows >= M / columns >= N, which the flush never writes.  Invalid POINTS (K axis: padding
  // slots, dead tiles, the ragged tail) read row 0 and have their A operand zeroed at use.
  uint32_t a_col[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) a_col[e] = AVEC ? (c4 < M ? c4 : 0u) : min(c4 + e, M - 1u);
  const uint32_t b_col = c4 < N ? c4 : 0u;   // N % 4 == 0 (host-checked)
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ra[D], rb[D];
  int ia[D], ib[D];                          // slot -> row of the step whose data is requested next round
  bool rv[D];                                // this lane's point of the step is real
  // liveness of a step's tile is looked up ONCE per group of D steps (D divides 32: a group never straddles tiles): a
  // per-step LDS read + s_waitcnt lgkmcnt(0) would put an LDS round trip into every 16-MFMA step of the in-order stream
  auto tile_live = [&](uint32_t q) -> bool { return q < nsteps && grp_lds[min(q >> sshift, ntl - 1u)] >= 0; };   // uniform
  auto slot_ok = [&](uint32_t q, bool live, uint32_t& slot) -> bool {
    slot = (bt0 + (q >> sshift) * tstride) * 128u + ((q & smask) + soff) * 4u + k;
    return live && slot < n_slots;
  };
  auto issue_idx = [&](int& xa, int& xb, uint32_t q, bool live) {
    if (!AIDX && !BIDX) return;
    uint32_t slot;
    const bool ok = slot_ok(q, live, slot);
    const uint32_t sl = ok ? slot : 0u;      // slot 0 exists
    if (AIDX) xa = ai[sl];
    if (BIDX) xb = bi[sl];
  };
  auto issue_data = [&](float4& a, float4& b, bool& valid, int xa, int xb, uint32_t q, bool live) {
    uint32_t slot;
    bool ok = slot_ok(q, live, slot);
    if (AIDX) ok = ok && xa >= 0;
    if (BIDX) ok = ok && xb >= 0;
    valid = ok;
    const uint32_t arow = ok ? (AIDX ? (uint32_t)xa : slot) : 0u;
    const uint32_t brow = ok ? (BIDX ? (uint32_t)xb : slot) : 0u;
    const float* pa = A + (size_t)arow * lda;
    const float* pb = B + (size_t)brow * ldb;
    typedef float nt4 __attribute__((ext_vector_type(4)));
    if (AVEC) {
      const nt4 t = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(pa + a_col[0]));   // read once: stream
      a = make_float4(t[0], t[1], t[2], t[3]);
    } else {
      a.x = pa[a_col[0]]; a.y = pa[a_col[1]]; a.z = pa[a_col[2]]; a.w = pa[a_col[3]];
    }
    const nt4 t = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(pb + b_col));
    b = make_float4(t[0], t[1], t[2], t[3]);
  };
  auto flush = [&](int grp) {
    if (grp < 0) return;
    // wave-private LDS transpose: D tile (ca, cb), lane (i>>2)*16 + j, register i&3 holds dW[4i+ca][4j+cb]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        reinterpret_cast<float4*>(wlds)[(a * 4 + b) * 64 + lane] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    __builtin_amdgcn_wave_barrier();
    float* C = u.C + (size_t)grp * c_stride;
    const uint32_t col = lane, cb = col & 3u, j = col >> 2;
    // every wave of a unit adds into the same M rows when the launch ends: start each wave at its own row so that
    // the 2 x 256 waves spread over all rows at any moment instead of queueing on the same addresses
    const uint32_t rot = (blockIdx.x * 29u) % M;
    for (uint32_t r0 = 0; r0 < M; ++r0) {
      const uint32_t row = r0 + rot < M ? r0 + rot : r0 + rot - M;
      const uint32_t ca = row & 3u, i = row >> 2;
      const float v = wlds[(((ca * 4u + cb) * 64u + (i >> 2) * 16u + j) * 4u) + (i & 3u)];
      if (col < N && v != 0.f) atomicAdd(C + (size_t)row * u.ldc + col, v);
    }
    __builtin_amdgcn_wave_barrier();
  };

  // prologue: indices of steps 0..D-1, then their data + the indices of steps D..2D-1
  {
    const bool live0 = tile_live(0), live1 = tile_live(D);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      ia[d] = 0;
      ib[d] = 0;
      issue_idx(ia[d], ib[d], (uint32_t)d, live0);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      issue_data(ra[d], rb[d], rv[d], ia[d], ib[d], (uint32_t)d, live0);
      issue_idx(ia[d], ib[d], (uint32_t)(d + D), live1);
      asm volatile("" ::: "memory");
    }
  }
  int cur = -2;
  for (uint32_t q0 = 0; q0 < nsteps; q0 += D) {
    const bool live_d = tile_live(q0 + D), live_i = tile_live(q0 + 2 * D);
    if ((q0 & smask) == 0) {
      const int grp = grp_lds[q0 >> sshift];
      if (grp >= 0 && grp != cur) {
        flush(cur);
        cur = grp;
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const uint32_t q = q0 + d;
      const bool va = rv[d];
      const float av[4] = {va ? ra[d].x : 0.f, va ? ra[d].y : 0.f, va ? ra[d].z : 0.f, va ? ra[d].w : 0.f};
      const float bv[4] = {rb[d].x, rb[d].y, rb[d].z, rb[d].w};
      issue_data(ra[d], rb[d], rv[d], ia[d], ib[d], q + D, live_d);
      issue_idx(ia[d], ib[d], q + 2 * D, live_i);
      asm volatile("" ::: "memory");         // keep the loads in program order: vmcnt waits are positional
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  if (BLOCKRED) {                            // one weight set: the workgroup adds its four waves' blocks up before the atomics
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        reinterpret_cast<float4*>(wlds)[(a * 4 + b) * 64 + lane] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
  } else {
    flush(cur);
  }
}

// One weight set (every launch but the per-class fine decoders): a workgroup's four waves work on the SAME unit, each on
// its own tiles (wave gw of the unit takes tiles gw, gw + n_gw, ... -- the dealing of gemm_roles_kernel with one wave per
// unit and workgroup), and their four 64x64 blocks are added in LDS before the global atomics: 4x fewer of them.  The
// atomics were 25-34 us of a 133 us launch (measured by compiling them out): 2048 waves x 4096 lane-atomics on 13 k
// addresses shared by every wave of a unit, all issued when the launch ends.
constexpr uint32_t GEMM_UNIT_WAVES = 4;     // waves of a workgroup that share a unit (8: one 128 KB workgroup per CU, no faster)
__global__ __launch_bounds__(64 * GEMM_UNIT_WAVES, 2) void gemm_units_kernel(GemmRoles r) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int* live_lds = reinterpret_cast<int*>(lds);                     // all zero: every tile handed to a wave is live
  int* tg = live_lds + GEMM_MAX_TPB;                               // weight-set ids of this workgroup's tile range
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t n_btiles = (r.n_slots + 127u) / 128u;
  const uint32_t unit = blockIdx.x % r.n_units, ub = blockIdx.x / r.n_units;
  const uint32_t nbu = gridDim.x / r.n_units;
  const GemmUnit& u = r.u[unit];
  float* wl = lds + 2u * GEMM_MAX_TPB;
  float* wlds = wl + wave * 4096u;
  // per-class launch: contiguous tile range (few weight sets per workgroup); one weight set: the whole launch is one run
  const bool grouped = r.tile_group != nullptr;
  const uint32_t t_lo = grouped ? ub * r.tiles_per_block : 0u;
  const uint32_t t_hi = grouped ? min(t_lo + r.tiles_per_block, n_btiles) : n_btiles;
  for (uint32_t t = threadIdx.x; t < GEMM_MAX_TPB; t += blockDim.x) {
    live_lds[t] = 0;
    tg[t] = (grouped && t_lo + t < t_hi) ? r.tile_group[t_lo + t] : 0;
  }
  __syncthreads();
  const uint32_t M = u.M, N = u.N;
  const uint32_t col = threadIdx.x & 63u, cb = col & 3u, j = col >> 2;
  const uint32_t rot = (ub * 29u) % M;
  uint32_t s0 = t_lo;
  while (s0 < t_hi) {                          // runs of one weight set (uniform over the workgroup)
    int g = 0;
    uint32_t s1 = t_hi;
    if (grouped) {
      g = tg[s0 - t_lo];
      s1 = s0 + 1u;
      while (s1 < t_hi && tg[s1 - t_lo] == g) ++s1;
    }
    if (g >= 0) {
      // wave w of the run: tiles first, first + stride, ...  (one weight set: wave gw of the unit's nbu * W waves)
      // one weight set: wave gw of the unit's nbu * W waves takes tiles gw, gw + nbu * W, ...; per-class launch: every
      // wave takes 8 of the 32 steps of each tile of the run, so the waves of a workgroup finish together
      const uint32_t stride = grouped ? 1u : nbu * GEMM_UNIT_WAVES;
      const uint32_t first = grouped ? s0 : ub * GEMM_UNIT_WAVES + wave;
      const uint32_t ntl = first < s1 ? (s1 - first + stride - 1u) / stride : 0u;
      const uint32_t sshift = grouped ? 3u : 5u, soff = grouped ? 8u * wave : 0u;
      static_assert(GEMM_UNIT_WAVES == 4, "the per-class split gives each of 4 waves 8 of a tile's 32 steps");
      switch (u.kind) {                        // uniform per workgroup
        case 4: gemm_unit_run<false, false, true, true>(u, r.n_slots, first, stride, ntl, live_lds, wlds, r.c_stride, sshift, soff); break;
        case 6: gemm_unit_run<false, true, true, true>(u, r.n_slots, first, stride, ntl, live_lds, wlds, r.c_stride, sshift, soff); break;
        case 5: gemm_unit_run<true, false, true, true>(u, r.n_slots, first, stride, ntl, live_lds, wlds, r.c_stride, sshift, soff); break;
        case 0: gemm_unit_run<false, false, false, true>(u, r.n_slots, first, stride, ntl, live_lds, wlds, r.c_stride, sshift, soff); break;
        case 1: gemm_unit_run<true, false, false, true>(u, r.n_slots, first, stride, ntl, live_lds, wlds, r.c_stride, sshift, soff); break;
        default: break;
      }
      __syncthreads();
      // D tile (ca, cb), lane (i>>2)*16 + j, register i&3 holds dW[4i+ca][4j+cb]; rows start at a per-workgroup offset
      // so that the workgroups of a unit do not queue on the same addresses
      float* C = u.C + (size_t)g * r.c_stride;
      for (uint32_t r0 = threadIdx.x >> 6; r0 < M; r0 += GEMM_UNIT_WAVES) {
        const uint32_t row = r0 + rot < M ? r0 + rot : r0 + rot - M;
        const uint32_t ca = row & 3u, i = row >> 2;
        const uint32_t idx = (((ca * 4u + cb) * 64u + (i >> 2) * 16u + j) * 4u) + (i & 3u);
        float v = 0.f;
#pragma unroll
        for (uint32_t w = 0; w < GEMM_UNIT_WAVES; ++w) v += wl[w * 4096u + idx];
        if (col < N && v != 0.f) atomicAdd(C + (size_t)row * u.ldc + col, v);
      }
      __syncthreads();
    }
    s0 = s1;
  }
}

// Builds the unit list of one network's <= 3 weight-gradient GEMMs; returns false when some operand does not fit the
// direct form (the caller then uses the LDS-staged kernel).
static bool make_roles(const GemmTnArgs* g, int ng, GemmRoles& r) {
  r.n_units = 0;
  for (int k = 0; k < ng; ++k) {
    const GemmTnArgs& a = g[k];
    if (a.M > 64u || a.N > 128u || (a.N & 3u) || (a.ldb & 3u) || (((uintptr_t)a.B) & 15u)) return false;
    const bool avec = (a.M & 3u) == 0 && (a.lda & 3u) == 0 && ((((uintptr_t)a.A) & 15u) == 0);
    const uint32_t kind = (a.a_index ? 1u : 0u) | (a.b_index ? 2u : 0u) | (avec ? 4u : 0u);
    if (!(kind == 4u || kind == 6u || kind == 5u || kind == 0u || kind == 1u)) return false;
    for (uint32_t c0 = 0; c0 < a.N; c0 += 64u) {
      if (r.n_units == 4u) return false;
      GemmUnit& u = r.u[r.n_units++];
      u.A = a.A; u.B = a.B + c0; u.a_index = a.a_index; u.b_index = a.b_index; u.C = a.C + c0;
      u.lda = a.lda; u.ldb = a.ldb; u.ldc = a.ldc; u.M = a.M; u.N = (a.N - c0 < 64u) ? a.N - c0 : 64u;
      u.kind = kind;
    }
  }
  for (uint32_t k = r.n_units; k < 4u; ++k) r.u[k] = r.u[0];
  r.tile_group = g[0].tile_group;
  r.n_slots = g[0].n_slots;
  r.c_stride = g[0].c_stride;
  return r.n_units > 0;
}

static bool shape_ok(uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl) {
  return (nn == 32 || nn == 64) && (nl == 1 || nl == 2) && n_in >= 8 && n_in <= 128 && (n_in % 8) == 0 && n_out >= 1 &&
         n_out <= 64;
}

static uint32_t pick_tiles_per_block(uint32_t n_btiles, uint32_t target_blocks) {
  // contiguous tile ranges, one wave of workgroups: the forward holds 2 workgroups of 4 waves per CU (LDS: weight
  // images + staging ~ 72 KB), the backward 1 workgroup of 8 waves (~ 98 KB) that takes tiles in pairs
  uint32_t tpb = (n_btiles + target_blocks - 1) / target_blocks;
  if (tpb < 1) tpb = 1;
  return tpb;
}

template <int NN, int NL>
static bool set_mlp_attrs() {
  bool ok = true;
  auto set = [&](const void* f) { ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) == hipSuccess; };
  set((const void*)mlp_fwd_kernel<NN, NL, false>);
  set((const void*)mlp_fwd_kernel<NN, NL, true>);
  set((const void*)mlp_bwd_data_kernel<NN, NL, true, true>);
  set((const void*)mlp_bwd_data_kernel<NN, NL, true, false>);
  set((const void*)mlp_bwd_data_kernel<NN, NL, false, false>);
  return ok;
}

static int mlp_init_attrs() {
  bool ok = set_mlp_attrs<32, 1>() && set_mlp_attrs<32, 2>() && set_mlp_attrs<64, 1>() && set_mlp_attrs<64, 2>();
  ok = ok && hipFuncSetAttribute((const void*)gemm_units_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) == hipSuccess;
  ok = ok && hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) == hipSuccess;
  if (!ok) {
    set_error("dns_init: hipFuncSetAttribute failed for the MLP kernels");
    return DNS_E_LAUNCH;
  }
  return DNS_OK;
}
static AttrRegistrar mlp_attr_registrar(mlp_init_attrs);

}  // namespace dns

using namespace dns;

#ifdef DNS_TRACE
extern "C" int dns_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dns::dns_trace_buf), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
#endif

extern "C" uint64_t dns_mlp_bwd_ws_floats(uint32_t n_slots, uint32_t n_neurons, uint32_t n_hidden_layers) {
  return (uint64_t)2 * n_hidden_layers * n_slots * n_neurons;
}

static int check_segments(const char* who, const float* x2, uint32_t ldx2, uint32_t n_in1, uint32_t n_in) {
  if (!x2) return DNS_OK;
  DNS_REQUIRE(n_in1 >= 4 && n_in1 < n_in && (n_in1 % 4) == 0, "%s: first input segment must hold a multiple of 4 columns in (0, n_in)", who);
  DNS_REQUIRE((ldx2 % 4) == 0 && (((uintptr_t)x2) % 16) == 0 && ldx2 >= n_in - n_in1,
              "%s: x2 must be 16-byte aligned with ldx2 %% 4 == 0 and ldx2 >= n_in - n_in1", who);
  return DNS_OK;
}

extern "C" int dns_mlp_fwd(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1,
                           const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                           uint32_t n_hidden_layers, float* y, uint32_t ldy, uint32_t n_slots, const int32_t* row_index,
                           const int32_t* tile_group, uint32_t param_stride, float* h_save, uint32_t flags, void* stream) {
  if (n_slots == 0) return DNS_OK;
  DNS_REQUIRE(x && params && y, "dns_mlp_fwd: NULL argument");
  DNS_REQUIRE(!h_save || (((uintptr_t)h_save) % 16) == 0, "dns_mlp_fwd: h_save must be 16-byte aligned");
  DNS_REQUIRE(shape_ok(n_in, n_out, n_neurons, n_hidden_layers), "dns_mlp_fwd: unsupported shape in=%u out=%u neurons=%u layers=%u",
              n_in, n_out, n_neurons, n_hidden_layers);
  DNS_REQUIRE((ldx % 4) == 0 && (((uintptr_t)x) % 16) == 0, "dns_mlp_fwd: x must be 16-byte aligned with ldx %% 4 == 0");
  DNS_REQUIRE(ldy >= n_out, "dns_mlp_fwd: ldy < n_out");
  {
    const int rc = check_segments("dns_mlp_fwd", x2, ldx2, n_in1, n_in);
    if (rc != DNS_OK) return rc;
  }
  if (!(flags & DNS_MLP_EXACT_F32)) {
    const int rc = ensure_ready((hipStream_t)stream, "dns_mlp_fwd");
    if (rc != DNS_OK) return rc;
    return launch_mlp_fwd_split(x, ldx, x2, ldx2, n_in1, params, n_in, n_out, n_neurons, n_hidden_layers, y, ldy, n_slots,
                                row_index, tile_group, param_stride, h_save, (flags & DNS_MLP_FP16) != 0, (hipStream_t)stream);
  }
  const XSeg seg = {x2, ldx2, x2 ? n_in1 : n_in};
  const MlpShape sh = make_shape(n_in, n_out);
  const uint32_t n_btiles = (n_slots + 127u) / 128u;
  const uint32_t tpb = pick_tiles_per_block(n_btiles, 512);
  const uint32_t blocks = (n_btiles + tpb - 1) / tpb;
  hipStream_t st = (hipStream_t)stream;
  {
    const int rc = ensure_ready(st, "dns_mlp_fwd");
    if (rc != DNS_OK) return rc;
  }
#define LAUNCH_FWD2(NN, NL, H)                                                                                     \
  {                                                                                                                \
    const size_t lds_bytes = ((size_t)FwdLds<NN, NL>::total(n_in, sh.mt, sh.vr) + 4 * STG_WAVE_FLOATS) * sizeof(float); \
    hipLaunchKernelGGL((mlp_fwd_kernel<NN, NL, H>), dim3(blocks), dim3(256), lds_bytes, st, x, ldx, seg, params, sh, y, \
                       ldy, n_slots, row_index, tile_group, param_stride, tpb, h_save);                            \
  }
#define LAUNCH_FWD(NN, NL)                \
  {                                       \
    if (flags & DNS_MLP_FP16) LAUNCH_FWD2(NN, NL, true) \
    else LAUNCH_FWD2(NN, NL, false)       \
  }
  if (n_neurons == 32 && n_hidden_layers == 1) LAUNCH_FWD(32, 1)
  else if (n_neurons == 32 && n_hidden_layers == 2) LAUNCH_FWD(32, 2)
  else if (n_neurons == 64 && n_hidden_layers == 1) LAUNCH_FWD(64, 1)
  else LAUNCH_FWD(64, 2)
#undef LAUNCH_FWD
#undef LAUNCH_FWD2
  return check_launch("dns_mlp_fwd");
}

extern "C" int dns_mlp_bwd(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, const float* dy,
                           uint32_t lddy, const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                           uint32_t n_hidden_layers, float* d_x, uint32_t lddx, float* d_x2, uint32_t lddx2,
                           float* d_params, float* ws, uint32_t n_slots, const int32_t* row_index,
                           const int32_t* tile_group, uint32_t param_stride, const float* h_saved, int accumulate_dx,
                           void* stream) {
  if (n_slots == 0) return DNS_OK;
  DNS_REQUIRE(x && dy && params && ws, "dns_mlp_bwd: NULL argument");
  DNS_REQUIRE(shape_ok(n_in, n_out, n_neurons, n_hidden_layers), "dns_mlp_bwd: unsupported shape in=%u out=%u neurons=%u layers=%u",
              n_in, n_out, n_neurons, n_hidden_layers);
  DNS_REQUIRE((ldx % 4) == 0 && (((uintptr_t)x) % 16) == 0, "dns_mlp_bwd: x must be 16-byte aligned with ldx %% 4 == 0");
  DNS_REQUIRE((((uintptr_t)ws) % 16) == 0, "dns_mlp_bwd: ws must be 16-byte aligned");
  {
    const int rc = check_segments("dns_mlp_bwd", x2, ldx2, n_in1, n_in);
    if (rc != DNS_OK) return rc;
  }
  if (!x2) n_in1 = n_in;
  const bool fp16 = (accumulate_dx & (int)DNS_MLP_FP16) != 0;
  const bool exact = (accumulate_dx & (int)DNS_MLP_EXACT_F32) != 0;
  DNS_REQUIRE(!exact || !fp16 || h_saved, "dns_mlp_bwd: the fp16 mode needs the hidden activations kept by dns_mlp_fwd (h_saved)");
  DNS_REQUIRE(!exact || !x2 || h_saved, "dns_mlp_bwd: a two-segment input needs the hidden activations kept by dns_mlp_fwd (h_saved)");
  DNS_REQUIRE(!x2 || !d_x || d_x2, "dns_mlp_bwd: d_x2 is required with a two-segment input when d_x is asked for");
  if (d_x) DNS_REQUIRE(lddx >= n_in1 && ((lddx % 4) != 0 || (((uintptr_t)d_x) % 16) == 0), "dns_mlp_bwd: d_x alignment / lddx");
  if (d_x && x2) DNS_REQUIRE(lddx2 >= n_in - n_in1, "dns_mlp_bwd: lddx2 < n_in - n_in1");
  const MlpShape sh = make_shape(n_in, n_out);
  hipStream_t st = (hipStream_t)stream;
  {
    const int rc = ensure_ready(st, "dns_mlp_bwd");
    if (rc != DNS_OK) return rc;
  }
  if (!(accumulate_dx & (int)DNS_MLP_EXACT_F32)) {
    return launch_mlp_bwd_split(x, ldx, x2, ldx2, n_in1, dy, lddy, params, n_in, n_out, n_neurons, n_hidden_layers, d_x, lddx,
                                d_x2, lddx2, d_params, ws, n_slots, row_index, tile_group, param_stride, accumulate_dx & 1,
                                (accumulate_dx >> 1) & 1, fp16, st);
  }
  const uint32_t NNr = n_neurons;
  const char* gb_env = nullptr;
  const DxSeg dseg = {(d_x && x2) ? d_x2 : nullptr, lddx2, (uint32_t)(accumulate_dx & 1), (uint32_t)((accumulate_dx >> 1) & 1)};
  const uint32_t n_btiles = (n_slots + 127u) / 128u;
  const uint32_t tpb = pick_tiles_per_block(n_btiles, 256);
  const uint32_t blocks = (n_btiles + tpb - 1) / tpb;
#define LAUNCH_BWD2(NN, NL, SV, H)                                                                                  \
  {                                                                                                                 \
    const size_t lds_bytes = ((size_t)BwdLds<NN, NL, SV>::total(n_in, n_out, d_x != nullptr) + 8 * STG_WAVE_FLOATS) * sizeof(float); \
    hipLaunchKernelGGL((mlp_bwd_data_kernel<NN, NL, SV, H>), dim3(blocks), dim3(512), lds_bytes, st, x, ldx, dy, lddy, \
                       params, sh, d_x, lddx, dseg, n_in1, ws, n_slots, row_index, tile_group, param_stride, tpb,   \
                       h_saved, n_slots);                                                                           \
  }
#define LAUNCH_BWD(NN, NL)            \
  {                                   \
    if (fp16) LAUNCH_BWD2(NN, NL, true, true) \
    else if (h_saved) LAUNCH_BWD2(NN, NL, true, false) \
    else LAUNCH_BWD2(NN, NL, false, false)   \
  }
  if (n_neurons == 32 && n_hidden_layers == 1) LAUNCH_BWD(32, 1)
  else if (n_neurons == 32 && n_hidden_layers == 2) LAUNCH_BWD(32, 2)
  else if (n_neurons == 64 && n_hidden_layers == 1) LAUNCH_BWD(64, 1)
  else LAUNCH_BWD(64, 2)
#undef LAUNCH_BWD
#undef LAUNCH_BWD2
  int rc = check_launch("dns_mlp_bwd(data)");
  if (rc != DNS_OK) return rc;
  if (!d_params) return DNS_OK;
  // weight gradients: dW_in = dH1^T X ; dW_h = dH2^T H1 ; dW_out = dY^T H_last
  const size_t SN = (size_t)n_slots * NNr;
  const float* wsH1 = h_saved ? h_saved : ws;
  const float* wsD1 = h_saved ? ws : ws + SN;
  const float* wsH2 = h_saved ? h_saved + SN : ws + 2 * SN;
  const float* wsD2 = h_saved ? ws + SN : ws + 3 * SN;
  GemmTnBatch batch;
  int ng = 0;
  auto add = [&](const float* A, uint32_t lda, const int32_t* ai, uint32_t M, const float* B, uint32_t ldb,
                 const int32_t* bi, uint32_t N, float* Cp, uint32_t ldc) {
    GemmTnArgs& g = batch.g[ng++];
    g.A = A; g.lda = lda; g.a_index = ai; g.M = M;
    g.B = B; g.ldb = ldb; g.b_index = bi; g.N = N;
    g.C = Cp; g.ldc = ldc; g.c_stride = param_stride;
    g.n_slots = n_slots; g.tile_group = tile_group; g.tiles_per_block = tpb;
  };
  add(wsD1, NNr, nullptr, NNr, x, ldx, row_index, n_in1, d_params, n_in);
  if (x2) add(wsD1, NNr, nullptr, NNr, x2, ldx2, row_index, n_in - n_in1, d_params + n_in1, n_in);
  float* dwo = d_params + (size_t)NNr * n_in;
  if (n_hidden_layers == 2) {
    add(wsD2, NNr, nullptr, NNr, wsH1, NNr, nullptr, NNr, dwo, NNr);
    dwo += (size_t)NNr * NNr;
  }
  add(dy, lddy, row_index, n_out, (n_hidden_layers == 2) ? wsH2 : wsH1, NNr, nullptr, NNr, dwo, NNr);
  for (int k = ng; k < 4; ++k) batch.g[k] = batch.g[0];
  GemmRoles roles;
  const bool staged = !make_roles(batch.g, ng, roles);
  DNS_REQUIRE(!(staged && x2), "dns_mlp_bwd: this two-segment input shape is not supported by the weight-gradient kernel");
  if (staged) {                               // LDS-staged 32x32x2 form: operand shapes the direct form does not take, or A/B
    const uint32_t max_cols = ((n_in + 31u) / 32u) * 32u + NNr > 2 * NNr ? ((n_in + 31u) / 32u) * 32u + NNr : 2 * NNr;
    const uint32_t out_cols = ((n_out + 31u) / 32u) * 32u + NNr;
    const size_t gemm_lds = (size_t)64 * (max_cols > out_cols ? max_cols : out_cols) * sizeof(float);
    uint32_t gblocks = gb_env ? (uint32_t)atoi(gb_env) : 512u;
    if (gblocks < 1) gblocks = 1;
    uint32_t gtpb = (n_btiles + gblocks - 1) / gblocks;
    if (gtpb < 1) gtpb = 1;
    gblocks = (n_btiles + gtpb - 1) / gtpb;
    for (int k = 0; k < 4; ++k) batch.g[k].tiles_per_block = gtpb;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(gblocks, ng), dim3(256), gemm_lds, st, batch);
  } else {
    // two workgroups per CU by default (each wave = one unit on its own SIMD); tiles per workgroup bounded by the
    // LDS weight-set table
    uint32_t nb = gb_env ? (uint32_t)atoi(gb_env) : 512u;   // 2 workgroups per CU: measured 275 vs 285 us at 256
    if (nb > n_btiles) nb = n_btiles;
    const uint32_t nb_min = (n_btiles + GEMM_MAX_TPB - 1) / GEMM_MAX_TPB;
    if (nb < nb_min) nb = nb_min;
    if (nb < 1) nb = 1;
    roles.tiles_per_block = (n_btiles + nb - 1) / nb;
    nb = (n_btiles + roles.tiles_per_block - 1) / roles.tiles_per_block;
    roles.strided = roles.tile_group == nullptr;
    {
      // GEMM_UNIT_WAVES waves of a workgroup share a unit and reduce in LDS (nb = waves per unit, as above)
      const uint32_t nb_req = gb_env ? (uint32_t)atoi(gb_env) : 512u;
      uint32_t nbu = ((roles.tile_group ? nb_req : nb) + GEMM_UNIT_WAVES - 1u) / GEMM_UNIT_WAVES;
      if (nbu > n_btiles) nbu = n_btiles;
      const uint32_t nbu_min = roles.tile_group ? (n_btiles + GEMM_MAX_TPB - 1) / GEMM_MAX_TPB
                                                : (n_btiles + GEMM_UNIT_WAVES * GEMM_MAX_TPB - 1) / (GEMM_UNIT_WAVES * GEMM_MAX_TPB);
      if (nbu < nbu_min) nbu = nbu_min;
      if (nbu < 1) nbu = 1;
      roles.tiles_per_block = (n_btiles + nbu - 1) / nbu;          // per-class launch: contiguous tile range per workgroup
      const size_t units_lds = (2 * GEMM_MAX_TPB + (size_t)GEMM_UNIT_WAVES * 4096) * sizeof(float);
      hipLaunchKernelGGL(gemm_units_kernel, dim3(nbu * roles.n_units), dim3(64 * GEMM_UNIT_WAVES), units_lds, st, roles);
    }
  }
  return check_launch("dns_mlp_bwd(weights)");
}

// Bias-free ReLU MLP on the CDNA4 matrix cores with SPLIT OPERANDS: every fp32 operand v is carried as two halfs
// hi = f16(v s), lo = f16(v s - hi) (s a power of two), and a product is three v_mfma_f32_32x32x16_f16 with fp32
// accumulation,  a_lo b_hi + a_hi b_lo + a_hi b_hi  (the a_lo b_lo term is below 2^-22 of the product).
//
// Replaces tcnn.Network{CutlassMLP} (reference models/decoder.py:58-64,84-90,101-116, slams/mapping.py:737-743):
// y = W_out relu(W_h relu(W_in x)), no bias.
//
// Why: gfx950 has no reduced-precision fp32 path (no xf32) and its exact-fp32 MFMA runs at the fp32 VECTOR rate, 1/16
// of the f16 rate.  Three f16 products per fp32 product were measured (tools/mfma_f16x3_probe.hip, MI355X) at 3.3x the
// throughput of v_mfma_f32_32x32x2_f32 with an error of 6-8e-8 of sum|a b| -- BELOW the 1.1-1.5e-7 of an fp32 fma chain
// (22 operand bits + an accumulator wider than fp32 inside a K = 16 step), f16 subnormals honoured.
//
// Orientation (unchanged from the fp32 kernels this file replaces): POINTS live on the MFMA column / lane axis, weights
// are the A operand.  v_mfma_f32_32x32x16_f16: lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.  With points on lanes a layer's
// accumulator IS the next layer's B operand: registers 8s .. 8s+7 of an accumulator tile are K-step s, element j of lane
// half h being row 16 s + 8 (j >> 2) + 4 h + (j & 3) -- the weight images are stored in that k order ("CHAIN" map), so
// activations never leave registers between layers.
//
// Scaling: f16 has 30 binades of normal range, so every operand is scaled by a power of two that puts its largest
// magnitude in [2^13, 2^14): weights per matrix (workgroup-wide max when the LDS images are built), activations and
// gradients PER POINT (a lane pair's max; the scale factors out of the point's output column), all tracked as integer
// exponents and undone exactly with ldexp.  NaN / Inf propagate (a non-finite maximum leaves the scale at 1).
#pragma once
#include "common.hpp"
#include "split_rows.hpp"

namespace dns {
namespace sp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

using sr::TGT_EXP;                           // scaled maxima lie in [2^13, 2^14)
using sr::scale_exp;                         // exponent k such that m * 2^k lies there; 0 for m = 0 or a non-finite m

struct Frag {
  half8 hi, lo;
};

__device__ __forceinline__ uint32_t acc_row(uint32_t r, uint32_t h) { return (r & 3u) + 8u * (r >> 2) + 4u * h; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

__device__ __forceinline__ float pow2f(int k) { return ldexpf(1.0f, k); }

// Cross-lane maxima without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 (address arithmetic, an LDS-queue
// instruction, s_waitcnt lgkmcnt(0)): a wave that is alone on its SIMD pays ~100 cycles for each, and the per-tile row maximum
// ran twelve of them back to back (round 5 stamps: 2.6 k cycles of a 26 k-cycle tile).  DPP modifiers and gfx950's
// v_permlane32_swap move the values inside the vector ALU instead.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// max over the 8 lanes that share lane >> 3 (what xor-shuffles by 1, 2, 4 give), in every one of them
__device__ __forceinline__ float max_lanes8(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));            // quad_perm [1,0,3,2]: lane ^ 1
  v = fmaxf(v, dpp_mov<0x4E>(v));            // quad_perm [2,3,0,1]: lane ^ 2
  return fmaxf(v, dpp_mov<0x141>(v));        // row_half_mirror: the other quad of the 8 lanes (every quad already holds its maximum)
}
// max(v[lane], v[lane ^ 32]) in every lane (what an xor-shuffle by 32 gives)
__device__ __forceinline__ float max_halves(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// 8 fp32 values (already in the lane's k order) -> hi / lo halfs of v * s
__device__ __forceinline__ Frag split8(const float (&v)[8], float s) {
  Frag f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * s;
    f.hi[j] = (_Float16)x;
    f.lo[j] = (_Float16)(x - (float)f.hi[j]);
  }
  return f;
}

template <int PREC>
__device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 acc) {
  if (PREC == 3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, acc, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, acc, 0, 0, 0);
}

// ---- products whose K axis is the POINT axis (the weight gradients) ------------------------------------------------
// There the scale of an operand cannot follow the point (the point is the summation index) and one accumulator collects
// tiles of very different magnitude (gradients of rays that already fit beside rays that do not), so these products use the
// format that needs no scaling at all: every fp32 value as THREE bf16 parts h + m + l (8 + 8 + 8 significant bits, fp32's
// exponent range) and six v_mfma_f32_32x32x16_bf16 per product, h h + (h m + m h) + (h l + l h + m m); the dropped terms
// are below 2^-23 of the product.  (Scaled f16 pairs with per-wave "ratchet" exponents were tried first: every place that
// rescales or restarts an accumulator outside a matrix instruction costs the compiler a second copy of all accumulators.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Frag3 {
  bf16x8 h, m, l;
};

__device__ __forceinline__ Frag3 split8_bf3(const float (&v)[8]) {
  Frag3 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f.h[j] = (__bf16)v[j];
    const float r = v[j] - (float)f.h[j];
    f.m[j] = (__bf16)r;
    f.l[j] = (__bf16)(r - (float)f.m[j]);
  }
  return f;
}

// PREC == 1 (tcnn's half-precision mode): two parts (16 bits, at least fp16's 11) and three products
#ifndef DNS_WGRAD_PRODUCTS
#define DNS_WGRAD_PRODUCTS 6
#endif
template <int PREC>
__device__ __forceinline__ f32x16 mma_pt(const Frag3& a, const Frag3& b, f32x16 acc) {
  if (PREC == 3 && DNS_WGRAD_PRODUCTS == 6) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
}

// ---- weight images ----------------------------------------------------------------------------------------------
// A-operand image of Meff [rows x K]: fragment (row tile rt, K-step s, part p in {hi, lo}) = 64 lanes x 8 halfs (1 KB);
// lane (r, h) element j holds part_p(scale * Meff[32 rt + r][kmap(s, h, j)]).  NAT map k = 16 s + 8 h + j (operands
// staged from memory in column order), CHAIN map k = 16 s + 8 (j >> 2) + 4 h + (j & 3) (operands that are accumulators).
enum KMap { K_NAT = 0, K_CHAIN = 1 };

__device__ __forceinline__ const uint4* frag_ptr(const _Float16* img, uint32_t nsteps, uint32_t rt, uint32_t s, uint32_t part,
                                                 uint32_t lane) {
  return reinterpret_cast<const uint4*>(img) + ((rt * nsteps + s) * 2u + part) * 64u + lane;
}

template <int PREC>
__device__ __forceinline__ Frag load_frag(const _Float16* img, uint32_t nsteps, uint32_t rt, uint32_t s, uint32_t lane) {
  Frag f;
  const uint4 hi = *frag_ptr(img, nsteps, rt, s, 0, lane);
  f.hi = *reinterpret_cast<const half8*>(&hi);
  if (PREC == 3) {
    const uint4 lo = *frag_ptr(img, nsteps, rt, s, 1, lane);
    f.lo = *reinterpret_cast<const half8*>(&lo);
  } else {
    f.lo = f.hi;
  }
  return f;
}

// Images are built by walking M in MEMORY order (aligned float4s): every workgroup re-reads the weights, so all loads
// of all matrices are put in flight first (image_load), the maxima are reduced and the image area zeroed while they fly,
// and only then are the values split and scattered (image_scatter).  M is row-major [R x C], C % 4 == 0.
template <int MAXQ>
struct ImgQuads {
  float4 v[MAXQ];
};

// ld: the matrix's row stride in memory (>= C; ld > C: only the first C columns of every row are the matrix -- DNS_MLP_LIVE_IN,
// a first layer whose trailing input columns are identically zero)
template <int MAXQ>
__device__ __forceinline__ float image_load(ImgQuads<MAXQ>& q, const float* __restrict__ M, uint32_t R, uint32_t C, uint32_t ld = 0) {
  if (ld == 0) ld = C;
  const uint32_t qpr = C >> 2, nq = R * qpr;
  const bool vec = ((((uintptr_t)M) & 15u) == 0) && ((ld & 3u) == 0);
  float m = 0.f;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    const uint32_t e = threadIdx.x + j * blockDim.x;
    q.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < nq) {
      const uint32_t rr = e / qpr;
      const float* src = M + (size_t)rr * ld + (size_t)(e - rr * qpr) * 4u;
      if (vec) q.v[j] = *reinterpret_cast<const float4*>(src);
      else q.v[j] = make_float4(src[0], src[1], src[2], src[3]);
    }
    const float a = fmaxf(fmaxf(fabsf(q.v[j].x), fabsf(q.v[j].y)), fmaxf(fabsf(q.v[j].z), fabsf(q.v[j].w)));
    m = fmaxf(m, a);
    bad = bad || !(a < INFINITY);            // fmaxf drops a NaN: keep non-finite weights visible
  }
  return bad ? INFINITY : m;
}

// Meff[i][k] = transpose ? M[k][i] : M[i][k]; entries with i >= 32 row_tiles or k outside the image stay zero
template <int MAXQ>
__device__ __forceinline__ void image_scatter(const ImgQuads<MAXQ>& q, _Float16* __restrict__ img, uint32_t R, uint32_t C,
                                              bool transpose, uint32_t row_tiles, uint32_t nsteps, int kmap, float scale) {
  auto slot = [&](uint32_t i, uint32_t k, uint32_t& idx) -> bool {   // half index of element (i, k) of part hi
    const uint32_t s = k >> 4, qq = k & 15u;
    uint32_t h, j;
    if (kmap == K_NAT) {
      h = qq >> 3;
      j = qq & 7u;
    } else {
      h = (qq >> 2) & 1u;
      j = ((qq >> 3) << 2) | (qq & 3u);
    }
    const uint32_t rt = i >> 5;
    if (rt >= row_tiles || s >= nsteps) return false;
    idx = ((((rt * nsteps + s) * 2u) * 64u) + (i & 31u) + 32u * h) * 8u + j;
    return true;
  };
  const uint32_t qpr = C >> 2, nq = R * qpr;
#pragma unroll
  for (int jq = 0; jq < MAXQ; ++jq) {
    const uint32_t e = threadIdx.x + jq * blockDim.x;
    if (e >= nq) continue;
    const uint32_t rr = e / qpr, cc = (e - rr * qpr) * 4u;
    const float vv[4] = {q.v[jq].x * scale, q.v[jq].y * scale, q.v[jq].z * scale, q.v[jq].w * scale};
    _Float16 hi[4], lo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      hi[c] = (_Float16)vv[c];
      lo[c] = (_Float16)(vv[c] - (float)hi[c]);
    }
    uint32_t idx;
    if (!transpose) {                        // i = rr, k = cc .. cc+3: four consecutive halfs of one lane slot
      if (slot(rr, cc, idx)) {
        half4v h4, l4;
#pragma unroll
        for (int c = 0; c < 4; ++c) { h4[c] = hi[c]; l4[c] = lo[c]; }
        *reinterpret_cast<half4v*>(img + idx) = h4;
        *reinterpret_cast<half4v*>(img + idx + 64u * 8u) = l4;     // part lo = the next 64-lane block
      }
    } else {                                 // i = cc + c, k = rr: one element of four neighbouring lanes
#pragma unroll
      for (uint32_t c = 0; c < 4u; ++c) {
        if (slot(cc + c, rr, idx)) {
          img[idx] = hi[c];
          img[idx + 64u * 8u] = lo[c];
        }
      }
    }
  }
}

__device__ __forceinline__ void lds_zero16(void* p, uint32_t bytes) {   // bytes % 16 == 0
  uint4* q = reinterpret_cast<uint4*>(p);
  for (uint32_t e = threadIdx.x; e < bytes / 16u; e += blockDim.x) q[e] = make_uint4(0u, 0u, 0u, 0u);
}

// Prepared weight images (dns_mlp_prepare): the image area of a kernel's LDS exactly as its prologue builds it, followed by the
// three scale exponents -- a workgroup then COPIES the area in (13-25 16-byte loads per thread, L2 hits) instead of loading the
// fp32 matrices, reducing their maxima across the workgroup (two barriers), splitting and scattering them: the weights change
// once per optimiser step, the prologue ran once per workgroup per launch.
__device__ __forceinline__ void images_copy_in(unsigned char* lds, const unsigned char* __restrict__ src, uint32_t bytes,
                                               const unsigned char* __restrict__ exps, int* wexp) {
  const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
  uint4* d4 = reinterpret_cast<uint4*>(lds);
  // eight 16-byte requests per thread in flight before the first LDS store (round 5: rolled, every trip of this loop was a
  // dependent load -> store pair -- 26 L2 round trips for a 104 KB backward image set, a third of a small launch's time)
  const uint32_t n = bytes / 16u, stride = blockDim.x;
  uint32_t e = threadIdx.x;
  for (; e + 7u * stride < n; e += 8u * stride) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = s4[e + k * stride];
#pragma unroll
    for (int k = 0; k < 8; ++k) d4[e + k * stride] = v[k];
  }
  for (; e < n; e += stride) d4[e] = s4[e];
  if (threadIdx.x < 3) wexp[threadIdx.x] = reinterpret_cast<const int*>(exps)[threadIdx.x];
}
__device__ __forceinline__ void images_copy_out(const unsigned char* lds, unsigned char* __restrict__ dst, uint32_t bytes, const int* wexp) {
  const uint4* s4 = reinterpret_cast<const uint4*>(lds);
  uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
  for (uint32_t e = threadIdx.x; e < bytes / 16u; e += blockDim.x) d4[e] = s4[e];
  if (threadIdx.x < 4) reinterpret_cast<int*>(dst + bytes)[threadIdx.x] = threadIdx.x < 3 ? wexp[threadIdx.x] : 0;
}

// workgroup-wide maxima of up to 3 values -> power-of-two exponents in wexp[0..2] (LDS); red: 3 * nwaves floats of LDS
__device__ __forceinline__ void block_scale_exps(float m0, float m1, float m2, float* red, int* wexp) {
  float m[3] = {m0, m1, m2};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m[i] = fmaxf(m[i], __shfl_xor(m[i], o));
  }
  const uint32_t wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63u) == 0) {
    red[wave * 3 + 0] = m[0];
    red[wave * 3 + 1] = m[1];
    red[wave * 3 + 2] = m[2];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float mm = 0.f;
    for (uint32_t w = 0; w < nw; ++w) mm = fmaxf(mm, red[w * 3 + threadIdx.x]);
    wexp[threadIdx.x] = scale_exp(mm);
  }
  __syncthreads();
}

// ---- row-coalesced global I/O through a wave-private LDS staging tile (32 rows x 36 floats) -----------------------
// In the accumulator layout a lane owns ONE point, so a direct float4 access touches 64 cache lines per instruction; staged
// through LDS the same kilobyte moves as 8 rows x 128 contiguous bytes.
constexpr uint32_t STG_LD = 36;
constexpr uint32_t STG_FLOATS = 32 * STG_LD;
constexpr uint32_t STG_ROWS = 64;            // two row tables: the tile being computed and the one being prefetched
constexpr uint32_t STG_RMAX = 32;            // per-row maxima of the tile's input rows
constexpr uint32_t STG_WAVE_FLOATS = STG_FLOATS + STG_ROWS + STG_RMAX;
constexpr uint32_t FWD_WAVE_FLOATS = STG_WAVE_FLOATS + 32;   // the forward kernel's wave area: + the tile's 32 row offsets in y

// (Measured in round 2: volatile staging accesses instead of this fence -- so that unrelated LDS reads, e.g. the next weight
// fragments, may move across a staging round trip -- let hipcc hoist so much that every backward instance spilled 60-198
// registers and the float4 accesses were split into dwords.  The fence doubles as the scheduling barrier that keeps the
// register pressure where it is.)
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("" ::: "memory");             // LDS instructions of one wave execute in order: compiler fence only
  __builtin_amdgcn_wave_barrier();
}

// ---- buffer-addressed global I/O (round 5) -------------------------------------------------------------------------
// A wave that is alone on its SIMD pays every memory round trip it waits for, and gfx9's vmcnt counts loads AND stores in
// issue order: a guarded access (`if (row >= 0) store`) is a divergent branch around a vector-memory instruction, behind which
// hipcc can no longer count what is in flight and waits for vmcnt(0) -- round 4's backward kernel drained EVERY outstanding
// store before each of its twelve dX stores per tile (ISA: s_waitcnt vmcnt(0) in front of every v_pk_add / store pair).
// Raw buffer instructions make the guard part of the ADDRESS instead: an offset at or beyond num_records reads zeros and
// drops the store, so the tile's accesses are straight-line code (exact vmcnt(N) waits), carry a 32-bit offset (no 64-bit
// address arithmetic per access) and need no clamp / select around them.  Every matrix handled this way must lie below
// BUF_OOB bytes from its base (2 GiB: checked on the host for slot-ordered rows, per row on the device behind a row table).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t BUF_OOB = 0x80000000u;    // = num_records of every descriptor: this offset and everything above is out of range
constexpr uint32_t BUF_LIMIT = BUF_OOB - 65536u;   // first byte offset of a row that is refused (room for the row itself)

__device__ __forceinline__ rsrc_t buf_make(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)BUF_OOB, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(rsrc_t r, uint32_t off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float buf_load1(rsrc_t r, uint32_t off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ void buf_store4(rsrc_t r, uint32_t off, const float4& v) {
  u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)off, 0, 0);
}
// byte offset of row `row` (< 0: a padding slot) in a matrix of row stride ld floats; rows at or beyond row_limit (the first row
// whose offset in the widest matrix would pass BUF_LIMIT: computed once on the host side of the kernel) are dropped like a
// padding slot and counted in `bad`, which the kernel reports once, at its end, through the sticky device error word
__device__ __forceinline__ uint32_t buf_row_off(int row, uint32_t ld, bool ok) { return ok ? (uint32_t)row * (ld * 4u) : BUF_OOB; }

__device__ __forceinline__ void tile_rows_publish(int* __restrict__ rows_lds, const int32_t* __restrict__ row_index,
                                                  uint32_t slot0, uint32_t n_slots, uint32_t lane) {
  if (lane < 32u) {
    const uint32_t slot = slot0 + lane;
    int row = -1;
    if (slot < n_slots) row = row_index ? row_index[slot] : (int)slot;
    rows_lds[lane] = row;
  }
  wave_lds_fence();
}

struct XSeg {                                // optional second input segment: columns [n_in1, n_in) come from x2
  const float* x2;
  uint32_t ldx2, n_in1;
};

struct XChunk {                              // 32 columns of the tile's 32 rows: lane (r8 = l >> 3, j = l & 7) holds
  float4 v[4];                               // float4 j of rows r8, r8+8, r8+16, r8+24
};

// Unconditional loads (no exec-masked branches: the four requests leave back to back): a padding slot reads row 0 and a column
// past n_in reads the row's last four -- finite stand-ins that cannot reach any result: a padding slot's outputs are not
// stored and its dY is zero (hence every gradient it feeds), the weight images' columns past n_in are zero, and a duplicate of
// a valid column cannot raise the row maximum the point's scale comes from.
__device__ __forceinline__ void x_chunk_issue(XChunk& xc, const float* __restrict__ x, uint32_t ldx, const XSeg& seg,
                                              uint32_t n_in, const int* __restrict__ rows_lds, uint32_t c, uint32_t lane) {
  const uint32_t col = min(32u * c + 4u * (lane & 7u), n_in - 4u);
  const bool second = seg.x2 != nullptr && col >= seg.n_in1;
  const float* base = second ? seg.x2 + (col - seg.n_in1) : x + col;
  const uint32_t ld = second ? seg.ldx2 : ldx;
  int rows[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rows[i] = max(rows_lds[(lane >> 3) + 8 * i], 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) xc.v[i] = *reinterpret_cast<const float4*>(base + (size_t)(uint32_t)rows[i] * ld);
}

// ---- SPLIT-ROW input (split_rows.hpp): the tile's operand fragments straight from memory ------------------------------
// The producer of the rows (dns_encode_fwd_split, dns_feature_block_split) has already scaled and split them: lane (r = l & 31,
// h = l >> 5) loads, for K-step s, the 8 halfs at columns 16 s + 8 h of row r of the hi plane (and of the lo plane, lo_off halfs
// further) -- its B-operand fragment in the NAT map, 16 bytes per part and lane; the row's exponent is one more dword.  No LDS
// staging tile, no row maximum, no conversion: what x_chunk_issue / x_row_max / x_chunk_commit / x_chunk_read / split8 did per
// tile in every launch that reads the rows.  Two segments (columns [0, n_in1) of s1's rows, the rest from s2's; n_in1 % 16 == 0):
// the segments carry their own exponents; xs_align brings both to the smaller one with an exact power-of-two multiply of the
// packed halfs (a down-scaled lo part may lose bits below 2^-24 of the scaled row: < 2^-37 of the row's maximum).
struct XsSeg {
  const _Float16* rows;
  const int32_t* exps;
  uint32_t ld, lo_off;                       // halfs; lo_off = 0: no lo plane (half-width rows, PREC 1 only)
};
struct XsIn {
  XsSeg s1, s2;                              // s2.rows == nullptr: one segment
  uint32_t n_in1;
};
struct XsTile {
  Frag f[8];                                 // n_in <= 128: at most 8 K-steps
  int e1, e2;
};

template <int PREC>
__device__ __forceinline__ void xs_issue(XsTile& t, const XsIn& in, uint32_t ns0, const int* __restrict__ rows_lds, uint32_t lane) {
  const uint32_t row = (uint32_t)max(rows_lds[lane & 31u], 0);      // a padding slot reads row 0: nothing of it is stored
  const uint32_t h8 = 8u * (lane >> 5);
  const bool two = in.s2.rows != nullptr;
  const uint32_t ns1 = two ? (in.n_in1 >> 4) : ns0;
  const _Float16* r1 = in.s1.rows + (size_t)row * in.s1.ld + h8;
  const _Float16* r2 = two ? in.s2.rows + (size_t)row * in.s2.ld + h8 : r1;
  const uint32_t lo1 = in.s1.lo_off, lo2 = in.s2.lo_off;
  t.e1 = in.s1.exps[row];
  t.e2 = two ? in.s2.exps[row] : t.e1;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if ((uint32_t)s < ns0) {                 // uniform
      const bool second = (uint32_t)s >= ns1;
      const _Float16* ph = second ? r2 + 16u * ((uint32_t)s - ns1) : r1 + 16u * (uint32_t)s;
      t.f[s].hi = *reinterpret_cast<const half8*>(ph);
      if (PREC == 3) t.f[s].lo = *reinterpret_cast<const half8*>(ph + (second ? lo2 : lo1));
      else t.f[s].lo = t.f[s].hi;
    }
  }
}

// the point's exponent (operands hold 2^e * value); with two segments both are brought to the smaller exponent first
template <int PREC>
__device__ __forceinline__ int xs_align(XsTile& t, const XsIn& in, uint32_t ns0) {
  if (in.s2.rows == nullptr) return t.e1;
  const int e = min(t.e1, t.e2);
  const _Float16 m1 = (_Float16)ldexpf(1.0f, max(e - t.e1, -24)), m2 = (_Float16)ldexpf(1.0f, max(e - t.e2, -24));
  const uint32_t ns1 = in.n_in1 >> 4;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if ((uint32_t)s < ns0) {
      const _Float16 m = (uint32_t)s >= ns1 ? m2 : m1;
      t.f[s].hi = t.f[s].hi * m;
      if (PREC == 3) t.f[s].lo = t.f[s].lo * m;
      else t.f[s].lo = t.f[s].hi;
    }
  }
  return e;
}

// first layer on split-row fragments: out[t] += W_in[t-th 32 rows] * x.  The A fragments of K-step s+1 are requested before step
// s's matrix instructions (as in layer_chain); the scheduling barrier per step keeps hipcc from hoisting ALL steps' fragment reads
// in front of the first product (16 fragments of 8 registers: measured as 130 spilled registers in the forward kernel).
template <int PREC, int NT_OUT>
__device__ __forceinline__ void layer_first_xs(const XsTile& x, uint32_t ns_live, const _Float16* img, uint32_t nsteps, uint32_t lane,
                                               f32x16 (&out)[NT_OUT]) {
  Frag an[NT_OUT];
#pragma unroll
  for (int t = 0; t < NT_OUT; ++t) an[t] = load_frag<PREC>(img, nsteps, t, 0, lane);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    if ((uint32_t)ks < ns_live) {              // uniform
      Frag ac[NT_OUT];
#pragma unroll
      for (int t = 0; t < NT_OUT; ++t) ac[t] = an[t];
      if ((uint32_t)ks + 1u < ns_live) {
#pragma unroll
        for (int t = 0; t < NT_OUT; ++t) an[t] = load_frag<PREC>(img, nsteps, t, ks + 1, lane);
      }
#pragma unroll
      for (int t = 0; t < NT_OUT; ++t) out[t] = mma<PREC>(ac[t], x.f[ks], out[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Saved hidden activations (dns_mlp_fwd's h_save, slot-major rows of n_neurons floats): 32 columns of the tile's 32 slots in
// the XChunk load layout, and the read that turns the staged chunk into an ACCUMULATOR-layout tile (lane = point column,
// register r = hidden row (r & 3) + 8 (r >> 2) + 4 h) -- the inverse of store_tile_rows_scalar's staging write.
__device__ __forceinline__ void h_chunk_issue(XChunk& hc, const float* __restrict__ hs, uint32_t nn, uint32_t slot0, uint32_t n_slots,
                                              uint32_t t, uint32_t lane) {
  const float* base = hs + 32u * t + 4u * (lane & 7u);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t slot = min(slot0 + (lane >> 3) + 8u * i, n_slots - 1u);
    // component-wise: a plain struct copy here and in x_chunk_commit is two memcpys (global -> private -> LDS) that the
    // compiler leaves in scratch memory -- the chunk is never read as a value on this path
    const float4 t = *reinterpret_cast<const float4*>(base + (size_t)slot * nn);
    hc.v[i] = make_float4(t.x, t.y, t.z, t.w);
  }
}
__device__ __forceinline__ f32x16 acc_tile_read(const float* __restrict__ stg, uint32_t lane) {
  const float* src = stg + (lane & 31u) * STG_LD + 4u * (lane >> 5);
  f32x16 a;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(src + 8 * g);
    a[4 * g] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
  }
  wave_lds_fence();
  return a;
}

// max |x| of every row of the tile over all its chunks (still in the load layout: 8 lanes share a row) -> rmax[32] in LDS.
// The point's scale must be known before its first operand is converted, and this costs 12 cross-lane steps per tile
// where keeping the whole half row in registers until its maximum is known cost 48 registers.
template <int NCH>
__device__ __forceinline__ void x_row_max(const XChunk (&xc)[NCH], uint32_t n_chunks, float* __restrict__ rmax, uint32_t lane) {
  float rm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if ((uint32_t)c < n_chunks) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        rm[i] = fmaxf(rm[i], fmaxf(fmaxf(fabsf(xc[c].v[i].x), fabsf(xc[c].v[i].y)), fmaxf(fabsf(xc[c].v[i].z), fabsf(xc[c].v[i].w))));
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rm[i] = max_lanes8(rm[i]);
    if ((lane & 7u) == 0) rmax[(lane >> 3) + 8 * i] = rm[i];
  }
  wave_lds_fence();
}

__device__ __forceinline__ void x_chunk_commit(const XChunk& xc, float* __restrict__ stg, uint32_t lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(stg + ((lane >> 3) + 8 * i) * STG_LD + 4u * (lane & 7u)) = xc.v[i];
  wave_lds_fence();
}

// the lane's operands of the staged 32-column chunk: columns 16 s + 8 h + j (NAT map), s = 0, 1
__device__ __forceinline__ void x_chunk_read(float (&xs)[16], const float* __restrict__ stg, uint32_t lane) {
  const float* src = stg + (lane & 31u) * STG_LD + 8u * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(src + 16 * s), b = *reinterpret_cast<const float4*>(src + 16 * s + 4);
    xs[8 * s + 0] = a.x; xs[8 * s + 1] = a.y; xs[8 * s + 2] = a.z; xs[8 * s + 3] = a.w;
    xs[8 * s + 4] = b.x; xs[8 * s + 5] = b.y; xs[8 * s + 6] = b.z; xs[8 * s + 7] = b.w;
  }
  wave_lds_fence();                          // the staging tile may be overwritten once these reads are issued
}

// a (32 features x 32 points, accumulator layout, already unscaled) -> y[row][col0 + f], f < ncols, rows through the
// tile's row table; dword stores with lane = feature: 2 rows x 128 bytes per instruction (any ldy / alignment)
__device__ __forceinline__ void store_tile_rows_scalar(float* __restrict__ y, uint32_t ldy, uint32_t col0, uint32_t ncols,
                                                       const int* __restrict__ rows_lds, const f32x16& a,
                                                       float* __restrict__ stg, uint32_t lane) {
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(stg + pt * STG_LD + 8 * g + 4 * h) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
  wave_lds_fence();
  // every row and value is read from LDS FIRST, the sixteen stores follow under their masks: with the guard around each
  // (read, read, store) triple hipcc emits ds_read, s_waitcnt lgkmcnt(0), branch, store sixteen times over (round 3, ISA)
  const uint32_t f = lane & 31u;
  int rows[16];
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t r = (lane >> 5) + 2 * i;
    rows[i] = rows_lds[r];
    v[i] = stg[r * STG_LD + f];
  }
  const bool colok = f < ncols;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (rows[i] >= 0 && colok) y[(size_t)(uint32_t)rows[i] * ldy + col0 + f] = v[i];
  wave_lds_fence();
}

// a: 32 features x 32 points -> rows dst[row * ld + 0..31] of 32 consecutive slot rows, row < nrows (16-byte aligned rows)
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

template <int NT>
__device__ __forceinline__ float tile_max(const f32x16 (&a)[NT]) {      // after ReLU: values are >= 0 (or NaN)
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(a[t][r]));
  return max_halves(m);                      // the point's other lane half
}

// out[t] = W[t-th 32 rows] * act, act = accumulator tiles scaled by 2^kf on conversion; image in the CHAIN map.
// The A fragments of K-step s+1 are requested before step s's operand is converted, so their LDS latency hides under
// the conversion and the matrix instructions instead of stalling every step (s_waitcnt lgkmcnt(0) in front of each group).
template <int PREC, int NT_OUT, int NT_IN>
__device__ __forceinline__ void layer_chain(const f32x16 (&act)[NT_IN], float f, const _Float16* img, uint32_t lane,
                                            f32x16 (&out)[NT_OUT]) {
  constexpr uint32_t nsteps = NT_IN * 2;
#pragma unroll
  for (int t = 0; t < NT_OUT; ++t) out[t] = zero16();
  Frag an[NT_OUT];
#pragma unroll
  for (int t = 0; t < NT_OUT; ++t) an[t] = load_frag<PREC>(img, nsteps, t, 0, lane);
#pragma unroll
  for (int st = 0; st < (int)nsteps; ++st) {
    Frag ac[NT_OUT];
#pragma unroll
    for (int t = 0; t < NT_OUT; ++t) ac[t] = an[t];
    if (st + 1 < (int)nsteps) {
#pragma unroll
      for (int t = 0; t < NT_OUT; ++t) an[t] = load_frag<PREC>(img, nsteps, t, st + 1, lane);
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = act[st >> 1][8 * (st & 1) + j];
    const Frag b = split8(v, f);
#pragma unroll
    for (int t = 0; t < NT_OUT; ++t) out[t] = mma<PREC>(ac[t], b, out[t]);
  }
}

struct DxSeg {
  float* dx2;
  uint32_t lddx2, acc1, acc2;                // acc: read-add-write instead of overwrite
  uint32_t col_lo;                           // input-gradient columns below this one have no consumer: not stored (DNS_MLP_DX_FROM)
};

// The values a read-add-write store of store_tile_rows_vec will add to: requested EARLY (all input-gradient tiles of a
// point tile at once, before their matrix products), because a wave that is alone on its SIMD pays the full memory latency
// of a load that is issued right in front of its use (measured: the accumulating networks took 220-260 us, the overwriting
// one 144).  Zero where nothing is to be added.
struct DxOld {
  float4 v[4];
};

__device__ __forceinline__ void dx_old_issue(DxOld& o, const float* __restrict__ dst1, uint32_t ld1, const DxSeg& seg,
                                             uint32_t n_in1, uint32_t col0, uint32_t ncols, const int* __restrict__ rows_lds,
                                             uint32_t lane) {
  const uint32_t rr = lane >> 3, c4 = 4u * (lane & 7u);
  const uint32_t col = col0 + c4;
  const bool second = seg.dx2 != nullptr && col >= n_in1;
  // (selecting the two FIELDS by a lane condition makes hipcc index the kernel-argument segment with a vector load, whose
  //  s_waitcnt vmcnt(0) then drains every request the wave has in flight -- found in the ISA in round 3)
  const bool a1 = seg.acc1 != 0, a2 = seg.acc2 != 0;
  const bool acc = (second && a2) || (!second && a1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rows_lds[rr + 8 * i];
    o.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (acc && row >= 0 && c4 < ncols && col >= seg.col_lo) {
      const float4* p = second ? reinterpret_cast<const float4*>(seg.dx2 + (size_t)row * seg.lddx2 + (col - n_in1))
                               : reinterpret_cast<const float4*>(dst1 + (size_t)row * ld1 + col);
      o.v[i] = *p;
    }
  }
}

// a: 32 features x 32 points (true scale) -> dst[row][col0 + f] (f < ncols, ncols % 4 == 0, 16-byte aligned rows), rows
// through the tile's row table; `old` (dx_old_issue) is added; optional second segment for columns >= n_in1
__device__ __forceinline__ void store_tile_rows_vec(float* __restrict__ dst1, uint32_t ld1, const DxSeg& seg, uint32_t n_in1,
                                                    uint32_t col0, uint32_t ncols, const int* __restrict__ rows_lds,
                                                    const f32x16& a, const DxOld& old, float* __restrict__ stg, uint32_t lane) {
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
    if (row >= 0 && c4 < ncols && col0 + c4 >= seg.col_lo) {
      const uint32_t col = col0 + c4;
      const bool second = seg.dx2 != nullptr && col >= n_in1;
      float4* p = second ? reinterpret_cast<float4*>(seg.dx2 + (size_t)row * seg.lddx2 + (col - n_in1))
                         : reinterpret_cast<float4*>(dst1 + (size_t)row * ld1 + col);
      const float4 u = old.v[i];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      *p = v;
    }
  }
  wave_lds_fence();
}

// same, dword accesses with lane = feature (any ld / alignment)
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
    if (row >= 0 && f < ncols && col0 + f >= seg.col_lo) {
      const uint32_t col = col0 + f;
      const bool second = seg.dx2 != nullptr && col >= n_in1;
      float* p = second ? seg.dx2 + (size_t)row * seg.lddx2 + (col - n_in1) : dst1 + (size_t)row * ld1 + col;
      const bool a1 = seg.acc1 != 0, a2 = seg.acc2 != 0;
      if ((second && a2) || (!second && a1)) v += *p;
      *p = v;
    }
  }
  wave_lds_fence();
}

// dy columns 32c .. 32c+31 of the tile's rows as dword loads with lane = column (any lddy): 2 rows x 128 B per instruction
struct DyChunk {
  float v[16];
};

__device__ __forceinline__ void dy_chunk_issue(DyChunk& d, const float* __restrict__ dy, uint32_t lddy, uint32_t n_out,
                                               const int* __restrict__ rows_lds, uint32_t c, uint32_t lane) {
  // unconditional loads from clamped addresses (a padding slot reads row 0, a column past n_out the last one), zeroed by a
  // select afterwards: sixteen requests back to back instead of sixteen (LDS read, wait, branch, load) rounds
  const uint32_t col = 32u * c + (lane & 31u);
  const uint32_t colc = min(col, n_out - 1u);
  int rows[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) rows[i] = rows_lds[(lane >> 5) + 2 * i];
#pragma unroll
  for (int i = 0; i < 16; ++i) d.v[i] = dy[(size_t)(uint32_t)max(rows[i], 0) * lddy + colc];
#pragma unroll
  for (int i = 0; i < 16; ++i) d.v[i] = (rows[i] >= 0 && col < n_out) ? d.v[i] : 0.f;
}

__device__ __forceinline__ void dy_chunk_commit(const DyChunk& d, float* __restrict__ stg, uint32_t lane) {
#pragma unroll
  for (int i = 0; i < 16; ++i) stg[((lane >> 5) + 2 * i) * STG_LD + (lane & 31u)] = d.v[i];
  wave_lds_fence();
}

// the staged 32 x 32 tile T[row][col] read COLUMN-wise: lane (c = l & 31, h) gets rows 16 s + 8 h + j of column c
__device__ __forceinline__ void stage_read_cols(float (&xs)[16], const float* __restrict__ stg, uint32_t lane) {
  const float* src = stg + 8u * (lane >> 5) * STG_LD + (lane & 31u);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) xs[8 * s + j] = src[(16 * s + j) * STG_LD];
}

__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}

// transpose_frags in two halves, for callers that put other work between the staging tile's write and its read (the LDS round
// trip then hides behind that work) and matrix instructions beside the read's conversions: the caller places a
// wave_lds_fence() behind the LAST use of the staged tile (i.e. before the next transpose_write)
__device__ __forceinline__ void transpose_write(const f32x16& a, int k_lane, float* __restrict__ stg, uint32_t lane) {
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) stg[acc_row(r, h) * STG_LD + pt] = ldexpf(a[r], -k_lane);
  wave_lds_fence();
}
struct TrRaw {
  float4 x[4];                               // the lane's two K-steps of the transposed tile, still fp32
};
__device__ __forceinline__ void transpose_read(TrRaw& t, const float* __restrict__ stg, uint32_t lane) {
  const float* src = stg + (lane & 31u) * STG_LD + 8u * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    t.x[2 * s] = *reinterpret_cast<const float4*>(src + 16 * s);
    t.x[2 * s + 1] = *reinterpret_cast<const float4*>(src + 16 * s + 4);
  }
}
__device__ __forceinline__ void transpose_split(const TrRaw& t, Frag3 (&out)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float4 x0 = t.x[2 * s], x1 = t.x[2 * s + 1];
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    out[s] = split8_bf3(v);
  }
}

// accumulator tile (holding 2^k_lane * value, k_lane per lane) -> the two K-step fragments of the point-transposed form of
// the TRUE values (lane = feature, elements = 8 consecutive points)
__device__ __forceinline__ void transpose_frags(const f32x16& a, int k_lane, Frag3 (&out)[2], float* __restrict__ stg, uint32_t lane) {
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) stg[acc_row(r, h) * STG_LD + pt] = ldexpf(a[r], -k_lane);
  wave_lds_fence();
  const float* src = stg + pt * STG_LD + 8u * h;     // here "pt" is the feature row of the transposed read
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float4 x0 = *reinterpret_cast<const float4*>(src + 16 * s), x1 = *reinterpret_cast<const float4*>(src + 16 * s + 4);
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    out[s] = split8_bf3(v);
  }
  wave_lds_fence();
}

// ---- the backward kernel's buffer-addressed tile I/O -----------------------------------------------------------------
// Per wave and tile three tables of 32 byte offsets (one per slot of the tile; BUF_OOB for a padding slot) in LDS: the rows
// of dy, of d_x and of d_x2.  An access adds its lane's column bytes -- or takes BUF_OOB where the lane has nothing to do.
constexpr uint32_t BWD_OFFS = 96;            // ints: dy | d_x | d_x2
constexpr uint32_t BWD_WAVE_FLOATS = STG_WAVE_FLOATS + BWD_OFFS;

__device__ __forceinline__ void tile_offs_publish(uint32_t* __restrict__ offs, const int* __restrict__ rows_lds, uint32_t lddy,
                                                  uint32_t lddx, uint32_t lddx2, uint32_t row_limit, uint32_t& bad, uint32_t lane) {
  if (lane < 32u) {
    const int row = rows_lds[lane];
    const bool ok = (uint32_t)row < row_limit;          // a padding slot (-1) fails the unsigned compare as well
    bad |= (row >= 0 && !ok) ? 1u : 0u;
    offs[lane] = buf_row_off(row, lddy, ok);
    offs[32u + lane] = buf_row_off(row, lddx, ok);
    offs[64u + lane] = buf_row_off(row, lddx2, ok);
  }
  wave_lds_fence();
}

// dy columns 32c .. 32c+31 of the tile's rows, lane = column: 2 rows x 128 B per instruction.  A padding slot reads zeros
// (BUF_OOB); a column past n_out reads the last valid one -- a finite stand-in that cannot reach a result: the rows of the
// W_out^T image past n_out are zero, the weight-gradient rows past n_out are not flushed, and a duplicate of a valid column
// cannot raise the maximum the point's scale comes from.
__device__ __forceinline__ void dy_chunk_issue_buf(DyChunk& d, rsrc_t rdy, const uint32_t* __restrict__ dyoff, uint32_t n_out,
                                                   uint32_t c, uint32_t lane) {
  const uint32_t colb = 4u * min(32u * c + (lane & 31u), n_out - 1u);
  uint32_t off[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) off[i] = dyoff[(lane >> 5) + 2 * i];
#pragma unroll
  for (int i = 0; i < 16; ++i) d.v[i] = buf_load1(rdy, off[i] + colb);
}

// the input gradient's two segments as buffer descriptors + what is constant over the kernel
struct DxIo {
  rsrc_t r1, r2;
  uint32_t two, acc1, acc2, n_in1, col_lo;    // two: a second segment exists (columns >= n_in1 go to r2)
};

// The lane's four byte offsets (rows (lane >> 3) + 8 i, columns col0 + 4 (lane & 7) ..+3) of column tile [col0, col0 + ncols)
// in segment `which`; BUF_OOB where the lane's columns are not part of it (past ncols, below col_lo, the other segment)
__device__ __forceinline__ void dx_tile_offs(uint32_t (&f)[4], const DxIo& io, const uint32_t* __restrict__ offs, uint32_t which,
                                             uint32_t col0, uint32_t ncols, uint32_t lane) {
  const uint32_t c4 = 4u * (lane & 7u), col = col0 + c4, rr = lane >> 3;
  const bool second = io.two != 0u && col >= io.n_in1;
  const bool mine = c4 < ncols && col >= io.col_lo && (second == (which != 0u));
  const uint32_t cb = 4u * (which ? col - io.n_in1 : col);
  const uint32_t* t = offs + 32u + 32u * which;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t o = t[rr + 8 * i] + cb;    // (BUF_OOB + a few hundred bytes stays out of range)
    f[i] = mine ? o : BUF_OOB;
  }
}

// which segments column tile [col0, col0 + 32) touches (uniform)
__device__ __forceinline__ void dx_tile_segs(const DxIo& io, uint32_t col0, uint32_t ncols, bool& use1, bool& use2) {
  use1 = io.two == 0u || col0 < io.n_in1;
  use2 = io.two != 0u && col0 + ncols > io.n_in1;
}

// what a read-add-write store of the tile will add to (zeros where nothing is added): requested a column tile AHEAD of its
// store and BEFORE the stores of the tile in front of it, so that the wait for these loads never includes a store
__device__ __forceinline__ void dx_old_issue_buf(DxOld& o, const DxIo& io, const uint32_t* __restrict__ offs, uint32_t col0,
                                                 uint32_t ncols, uint32_t lane) {
  bool use1, use2;
  dx_tile_segs(io, col0, ncols, use1, use2);
  const bool l1 = use1 && io.acc1 != 0u, l2 = use2 && io.acc2 != 0u;
  uint32_t f1[4], f2[4];
  dx_tile_offs(f1, io, offs, 0u, col0, ncols, lane);
  dx_tile_offs(f2, io, offs, 1u, col0, ncols, lane);
  if (l1 && l2) {
    // a tile that straddles the segment boundary: a lane belongs to ONE segment -- the two batches write disjoint lanes of the
    // same registers (exec-masked), nothing to merge afterwards
    const bool second = col0 + 4u * (lane & 7u) >= io.n_in1;
    if (!second) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o.v[i] = buf_load4(io.r1, f1[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o.v[i] = buf_load4(io.r2, f2[i]);
    }
  } else if (l1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = buf_load4(io.r1, f1[i]);
  } else if (l2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = buf_load4(io.r2, f2[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// a: 32 features x 32 points (true scale) -> the tile's rows of d_x / d_x2; straight-line: no guarded access.  `old` (what a
// read-add-write store adds to; only read when `add`) is consumed BEFORE the next column tile's values are requested into the
// same registers (has_next), and those requests leave before this tile's stores.
__device__ __forceinline__ void store_tile_rows_buf(const DxIo& io, const uint32_t* __restrict__ offs, uint32_t col0, uint32_t ncols,
                                                    const f32x16& a, DxOld& old, bool add, bool has_next, uint32_t next_ncols,
                                                    float* __restrict__ stg, uint32_t lane) {
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(stg + pt * STG_LD + 8 * g + 4 * h) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
  wave_lds_fence();
  bool use1, use2;
  dx_tile_segs(io, col0, ncols, use1, use2);
  const uint32_t rr = lane >> 3, c4 = 4u * (lane & 7u);
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(stg + (rr + 8 * i) * STG_LD + c4);
  if (add) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 u = old.v[i];
      v[i].x += u.x; v[i].y += u.y; v[i].z += u.z; v[i].w += u.w;
    }
    // The adds first, THEN the requests into the same registers: left to itself hipcc sinks the adds below the requests (they
    // are only needed by the stores), gives the requests registers of their own and copies them behind an s_waitcnt vmcnt(0)
    // at the end of the iteration.  The empty asm pins the sums in front of the requests.
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
    if (has_next) dx_old_issue_buf(old, io, offs, col0 + 32u, next_ncols, lane);
  }
  {
    // the first segment's four stores are UNCONDITIONAL (a tile that lies in the second segment sends four out-of-range
    // offsets): a store count the compiler can rely on keeps the wait for `old` at vmcnt(N >= 4) instead of vmcnt(<= 3)
    uint32_t f[4];
    dx_tile_offs(f, io, offs, 0u, col0, ncols, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) buf_store4(io.r1, f[i], v[i]);
  }
  if (use2) {
    uint32_t f[4];
    dx_tile_offs(f, io, offs, 1u, col0, ncols, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) buf_store4(io.r2, f[i], v[i]);
  }
  wave_lds_fence();
}

// a (32 features x 32 points, accumulator layout, already unscaled) -> y[row][col0 + f], f < ncols, rows through the tile's
// offset table (BUF_OOB = a padding slot); dword stores with lane = feature: 2 rows x 128 bytes per instruction, any row
// stride / alignment; straight-line (store_tile_rows_scalar's guarded stores kept hipcc from counting what is in flight)
__device__ __forceinline__ void store_tile_rows_scalar_buf(rsrc_t ry, const uint32_t* __restrict__ yoff, uint32_t col0, uint32_t ncols,
                                                           const f32x16& a, float* __restrict__ stg, uint32_t lane) {
  const uint32_t pt = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(stg + pt * STG_LD + 8 * g + 4 * h) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
  wave_lds_fence();
  const uint32_t f = lane & 31u;
  const uint32_t cb = f < ncols ? 4u * (col0 + f) : BUF_OOB;
  uint32_t off[16];
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t r = (lane >> 5) + 2 * i;
    off[i] = yoff[r];
    v[i] = stg[r * STG_LD + f];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    // (an out-of-range row offset plus an out-of-range column marker would wrap: the column marker wins by select)
    const uint32_t o = f < ncols ? off[i] + cb : BUF_OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), ry, (int)o, 0, 0);
  }
  wave_lds_fence();
}

// a: 32 features x 32 points -> rows (row0 + r) of a slot-major [n][ld] matrix behind descriptor r (r < nrows), columns
// col0 .. col0 + 31; 16-byte aligned rows
__device__ __forceinline__ void store_tile_staged_buf(rsrc_t rd, uint32_t row0_bytes, uint32_t ld, uint32_t col0, uint32_t nrows,
                                                      const f32x16& a, float* __restrict__ stg, uint32_t lane) {
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
    const uint32_t off = row < nrows ? row0_bytes + (row * ld + col0 + c4) * 4u : BUF_OOB;
    buf_store4(rd, off, v);
  }
  wave_lds_fence();
}

// ---- LDS layout of the forward kernel (bytes) ----
template <int NN, int NL>
struct FwdLds {
  static __host__ __device__ uint32_t ns0(uint32_t n_in) { return (n_in + 15u) / 16u; }
  static __host__ __device__ uint32_t mt(uint32_t n_out) { return (n_out + 31u) / 32u; }
  static __host__ __device__ uint32_t img_in(uint32_t) { return 0; }
  static __host__ __device__ uint32_t img_h(uint32_t n_in) { return (NN / 32) * ns0(n_in) * 2048u; }
  static __host__ __device__ uint32_t img_out(uint32_t n_in) { return img_h(n_in) + (NL - 1) * (NN / 32) * (NN / 16) * 2048u; }
  static __host__ __device__ uint32_t misc(uint32_t n_in, uint32_t n_out) { return img_out(n_in) + mt(n_out) * (NN / 16) * 2048u; }
  static __host__ __device__ uint32_t stage(uint32_t n_in, uint32_t n_out) { return misc(n_in, n_out) + 256u; }   // exps + reduction
  static __host__ __device__ uint32_t total(uint32_t n_in, uint32_t n_out, uint32_t nwaves) {
    return stage(n_in, n_out) + nwaves * FWD_WAVE_FLOATS * 4u;
  }
};


// the forward kernel's prologue: the three weight matrices of one weight set -> scaled f16 hi/lo operand images in LDS
template <int NN, int NL>
__device__ __forceinline__ void build_fwd_images(unsigned char* lds, const float* __restrict__ pw, uint32_t n_in, uint32_t n_out,
                                                 uint32_t n_in_w = 0) {
  if (n_in_w == 0) n_in_w = n_in;                                 // storage width of W_in's rows (> n_in: DNS_MLP_LIVE_IN)
  using L = FwdLds<NN, NL>;
  constexpr int NT = NN / 32;
  const uint32_t ns0 = L::ns0(n_in), mt = L::mt(n_out);
  _Float16* img_in = reinterpret_cast<_Float16*>(lds + L::img_in(n_in));
  _Float16* img_h = reinterpret_cast<_Float16*>(lds + L::img_h(n_in));
  _Float16* img_out = reinterpret_cast<_Float16*>(lds + L::img_out(n_in));
  int* wexp = reinterpret_cast<int*>(lds + L::misc(n_in, n_out));
  float* red = reinterpret_cast<float*>(lds + L::misc(n_in, n_out) + 64);
  const float* wh = pw + NN * n_in_w;
  const float* wout = wh + (NL - 1) * NN * NN;
  ImgQuads<8> q_in;                                              // 64 x 128 / 4 / 256 threads
  ImgQuads<4> q_h, q_out;
  const float m_in = image_load(q_in, pw, NN, n_in, n_in_w);
  const float m_h = (NL == 2) ? image_load(q_h, wh, NN, NN) : 0.f;
  const float m_out = image_load(q_out, wout, n_out, NN);
  lds_zero16(lds, L::misc(n_in, n_out));
  block_scale_exps(m_in, m_h, m_out, red, wexp);                 // two barriers: the zero fill is complete
  image_scatter(q_in, img_in, NN, n_in, false, NT, ns0, K_NAT, pow2f(wexp[0]));
  if (NL == 2) image_scatter(q_h, img_h, NN, NN, false, NT, NN / 16, K_CHAIN, pow2f(wexp[1]));
  image_scatter(q_out, img_out, n_out, NN, false, mt, NN / 16, K_CHAIN, pow2f(wexp[2]));
}


// ---- the forward kernel's arguments and body (the __global__ wrapper: mlp_split.hip) ----------------------------------
struct FwdArgs {
  const float* x;
  uint32_t ldx;
  XSeg seg;
  const float* params;
  uint32_t n_in, n_out;
  float* y;
  uint32_t ldy, n_slots;
  const int32_t* row_index;
  const int32_t* tile_group;
  uint32_t param_stride, tiles_per_block;
  float* h_save;
  const unsigned char* prep;                 // prepared images (NULL: build them from params)
  uint32_t prep_stride;                      // bytes per weight set
  XsIn xs;                                   // split-row input (XS kernels; x / seg unused then)
  uint32_t n_in_w;                           // storage width of W_in's rows (= n_in, or larger with DNS_MLP_LIVE_IN)
  uint32_t* err;                             // the library's sticky device error word (DNS_DEVERR_MLP_RANGE)
};

// XS: the input arrives in the split-row format (split_rows.hpp) and goes from memory straight into the B operand
// (a __device__ body so that the fused tracker iteration, track_fused.inc, can run it as one of its phases: vb = the workgroup's
//  index in this launch's grid, lds = the workgroup's dynamic LDS)
template <int NN, int NL, int PREC, bool XS = false>
__device__ __forceinline__ void mlp_fwd_body(const FwdArgs& a, uint32_t vb, unsigned char* lds) {
  constexpr int NT = NN / 32;
  using L = FwdLds<NN, NL>;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n_in = a.n_in, n_out = a.n_out;
  const uint32_t ns0 = L::ns0(n_in), mt = L::mt(n_out);
  const uint32_t n_chunks = (n_in + 31u) / 32u;
  const uint32_t n_btiles = (a.n_slots + 127u) / 128u;
  const uint32_t bt0 = vb * a.tiles_per_block;
  const uint32_t bt1 = min(bt0 + a.tiles_per_block, n_btiles);
  const _Float16* img_in = reinterpret_cast<const _Float16*>(lds + L::img_in(n_in));
  const _Float16* img_h = reinterpret_cast<const _Float16*>(lds + L::img_h(n_in));
  const _Float16* img_out = reinterpret_cast<const _Float16*>(lds + L::img_out(n_in));
  int* wexp = reinterpret_cast<int*>(lds + L::misc(n_in, n_out));
  float* stg = reinterpret_cast<float*>(lds + L::stage(n_in, n_out)) + wave * FWD_WAVE_FLOATS;
  int* rows_all = reinterpret_cast<int*>(stg + STG_FLOATS);
  XChunk xc[XS ? 1 : 4];                         // n_in <= 128: at most 4 chunks of 32 columns
  XsTile xt;                                     // (XS) the tile's operand fragments
  uint32_t cur_buf = 1;
  int cur_group = -2;
  const rsrc_t r_y = buf_make(a.y), r_hs = buf_make(a.h_save);
  uint32_t* yoff = reinterpret_cast<uint32_t*>(stg + STG_WAVE_FLOATS);      // byte offsets of the tile's rows in y (BUF_OOB: padding)
  const uint32_t y_row_limit = BUF_LIMIT / (4u * max(a.ldy, 1u));
  uint32_t bad_rows = 0;
  // the rows and inputs of live tile `nbt` -> row table `buf`, registers xc / xt
  auto request = [&](uint32_t nbt, uint32_t buf) {
    int* nrows_lds = rows_all + 32u * buf;
    tile_rows_publish(nrows_lds, a.row_index, nbt * 128u + wave * 32u, a.n_slots, lane);
    if constexpr (XS) {
      xs_issue<PREC>(xt, a.xs, ns0, nrows_lds, lane);
    } else {
      // all FOUR chunks, unconditionally: a chunk past n_in re-reads the row's last four columns (x_chunk_issue clamps: one
      // 16-byte piece per row, L1 hits) and is never consumed -- with the requests under `c < n_chunks` hipcc cannot count on
      // any request behind a chunk's and waits for vmcnt(3..0) at every chunk's first use: for every chunk behind it AND the
      // previous tile's output stores
#pragma unroll
      for (int c = 0; c < 4; ++c) x_chunk_issue(xc[c], a.x, a.ldx, a.seg, n_in, nrows_lds, c, lane);
    }
  };
  // The workgroup's FIRST live tile is requested here -- in front of the weight images' copy-in / build, which it does not
  // depend on --, every later one while the tile in front of it runs its later layers.  One request site inside the loop:
  // with a second one at its top ("not requested yet") the two definitions of the input registers met in a copy behind
  // s_waitcnt vmcnt(0) at every tile start, i.e. behind every output store of the previous tile (round 5, ISA).
  for (uint32_t nbt = bt0; nbt < bt1; ++nbt) {
    if ((a.tile_group ? a.tile_group[nbt] : 0) < 0) continue;
    request(nbt, 0u);
    break;
  }
  for (uint32_t bt = bt0; bt < bt1; ++bt) {
    const int grp = a.tile_group ? a.tile_group[bt] : 0;
    if (grp != cur_group) {
      __syncthreads();
      if (grp >= 0) {
        if (a.prep) {
          const unsigned char* src = a.prep + (size_t)grp * a.prep_stride;
          images_copy_in(lds, src, L::misc(n_in, n_out), src + L::misc(n_in, n_out), wexp);
        }
        if (!a.prep) build_fwd_images<NN, NL>(lds, a.params + (size_t)grp * a.param_stride, n_in, n_out, a.n_in_w);
      }
      cur_group = grp;
      __syncthreads();
    }
    if (grp < 0) continue;
    const uint32_t slot0 = bt * 128u + wave * 32u;
    const uint32_t nrows = slot0 < a.n_slots ? min(32u, a.n_slots - slot0) : 0u;
    // the tile's x rows (all <= 4 chunks of 32 columns) were requested while the PREVIOUS live tile ran its later layers
    cur_buf ^= 1u;
    const int* rows_lds = rows_all + 32u * cur_buf;
    if (lane < 32u) {                                // the rows' byte offsets in y (a padding slot: out of range)
      const int row = rows_lds[lane];
      const bool ok = (uint32_t)row < y_row_limit;
      bad_rows |= (row >= 0 && !ok) ? 1u : 0u;
      yoff[lane] = buf_row_off(row, a.ldy, ok);
    }
    wave_lds_fence();
    int kc;                                        // cumulative exponent: accumulators hold 2^kc * (true value)
    f32x16 a0[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) a0[t] = zero16();
    if constexpr (XS) {
      kc = xs_align<PREC>(xt, a.xs, ns0) + wexp[0];
      layer_first_xs<PREC, NT>(xt, ns0, img_in, ns0, lane, a0);
    } else {
    float* rmax = stg + STG_FLOATS + STG_ROWS;
    x_row_max<4>(xc, n_chunks, rmax, lane);
    kc = scale_exp(rmax[lane & 31u]);
    const float sx = pow2f(kc);
    kc += wexp[0];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if ((uint32_t)c < n_chunks) {                // uniform
        float xs[16];
        x_chunk_commit(xc[c], stg, lane);
        x_chunk_read(xs, stg, lane);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const uint32_t ks = 2u * c + s;
          if (ks < ns0) {                          // uniform
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = xs[8 * s + j];
            const Frag b = split8(v, sx);
#pragma unroll
            for (int t = 0; t < NT; ++t) a0[t] = mma<PREC>(load_frag<PREC>(img_in, ns0, t, ks, lane), b, a0[t]);
          }
        }
      }
    }
    }
    // request the next live tile's rows now: they arrive while this tile runs its later layers
    for (uint32_t nbt = bt + 1; nbt < bt1; ++nbt) {
      if ((a.tile_group ? a.tile_group[nbt] : 0) < 0) continue;
      request(nbt, cur_buf ^ 1u);
      break;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) a0[t][r] = fmaxf(a0[t][r], 0.f);
    if (a.h_save) {                                // kept for callers that want the hidden activations
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x16 hv;
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = ldexpf(a0[t][r], -kc);
        store_tile_staged_buf(r_hs, slot0 * (uint32_t)(NN * 4), NN, t * 32, nrows, hv, stg, lane);
      }
    }
    f32x16 a1[NT];
    if (NL == 2) {
      const int kf = scale_exp(tile_max<NT>(a0)) ;
      layer_chain<PREC, NT, NT>(a0, pow2f(kf), img_h, lane, a1);
      kc += kf + wexp[1];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a1[t][r] = fmaxf(a1[t][r], 0.f);
      if (a.h_save) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x16 hv;
#pragma unroll
          for (int r = 0; r < 16; ++r) hv[r] = ldexpf(a1[t][r], -kc);
          store_tile_staged_buf(r_hs, (a.n_slots + slot0) * (uint32_t)(NN * 4), NN, t * 32, nrows, hv, stg, lane);
        }
      }
    }
    const f32x16(&hl)[NT] = (NL == 2) ? a1 : a0;
    const int kf = scale_exp(tile_max<NT>(hl));
    kc += kf + wexp[2];
    const float fo = pow2f(kf);
    if (mt >= 2u) {
      f32x16 o[2];
      layer_chain<PREC, 2, NT>(hl, fo, img_out, lane, o);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = ldexpf(o[t][r], -kc);
      store_tile_rows_scalar_buf(r_y, yoff, 0, 32u, o[0], stg, lane);
      store_tile_rows_scalar_buf(r_y, yoff, 32, n_out - 32u, o[1], stg, lane);
    } else {
      f32x16 o[1];
      layer_chain<PREC, 1, NT>(hl, fo, img_out, lane, o);
#pragma unroll
      for (int r = 0; r < 16; ++r) o[0][r] = ldexpf(o[0][r], -kc);
      store_tile_rows_scalar_buf(r_y, yoff, 0, n_out, o[0], stg, lane);
    }
  }
  if (bad_rows != 0u && a.err) atomicOr(a.err, DNS_DEVERR_MLP_RANGE);
}



struct BwdArgs {
  const float* x;
  uint32_t ldx;
  XSeg seg;
  const float* dy;
  uint32_t lddy;
  const float* params;
  uint32_t n_in, n_out;
  float* dx;
  uint32_t lddx;
  DxSeg dseg;
  float* d_params;
  float* dh1;                                // [n_slots, n_neurons] workspace: dH_1 for the dW_in kernel (NULL without d_params)
  uint32_t n_slots;
  const int32_t* row_index;
  const int32_t* tile_group;
  uint32_t param_stride, tiles_per_block;
  const float* h_saved;                      // [n_hidden_layers][n_slots][n_neurons] hidden activations the forward kept (NULL: recompute)
  const unsigned char* prep;                 // prepared images of the backward kernel (NULL: build them from params)
  uint32_t prep_stride;                      // bytes per weight set
  XsIn xs;                                   // split-row input (xs.s1.rows != NULL: x / seg are unused)
  uint32_t n_waves;                          // waves per workgroup the launcher planned the grid for (4, or 8: frozen-scene form)
  uint32_t n_in_w;                           // storage width of W_in's rows (= n_in, or larger with DNS_MLP_LIVE_IN)
  uint32_t* err;                             // the library's sticky device error word (DNS_DEVERR_MLP_RANGE)
#ifdef DNS_BWD_TRACE
  unsigned long long* trace;                 // tools build only (make trace): s_memtime stamps at the phase boundaries
#endif
};
#ifdef DNS_BWD_TRACE
extern unsigned long long* g_bwd_trace;
#define DNS_TR(slot)                                                                                        \
  do {                                                                                                      \
    if (a.trace && lane == 0 && tr_tile < 6u)                                                               \
      a.trace[((size_t)(vb * 4u + wave) * 8u + tr_tile) * 8u + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
// run-level stamps: record 7 of the wave, slots 0 run start, 1 images built, 2 tiles done, 3 weight gradients flushed
#define DNS_TR_RUN(slot)                                                                                    \
  do {                                                                                                      \
    if (a.trace && lane == 0 && run0 == bt0)                                                                \
      a.trace[((size_t)(vb * 4u + wave) * 8u + 7u) * 8u + (slot)] = __builtin_readcyclecounter();   \
  } while (0)
#else
#define DNS_TR(slot) do { } while (0)
#define DNS_TR_RUN(slot) do { } while (0)
#endif

// per-(n_neurons, n_hidden_layers) launchers: one translation unit each (mlp_split_bwd_*.hip)
int launch_bwd_32_1(const BwdArgs& a, uint32_t blocks, bool fp16_single, hipStream_t st);
int launch_bwd_32_2(const BwdArgs& a, uint32_t blocks, bool fp16_single, hipStream_t st);
int launch_bwd_64_1(const BwdArgs& a, uint32_t blocks, bool fp16_single, hipStream_t st);
int launch_bwd_64_2(const BwdArgs& a, uint32_t blocks, bool fp16_single, hipStream_t st);
#define DNS_DECL_PREP(nn, nl)                                                                                                  \
  uint32_t prepared_bwd_bytes_##nn##_##nl(uint32_t n_in, uint32_t n_out);                                                     \
  int prepare_##nn##_##nl(const float* params, uint32_t param_stride, uint32_t n_in, uint32_t n_out, uint32_t n_sets,         \
                          unsigned char* blob, uint32_t fwd_bytes, uint32_t blob_stride, hipStream_t st, uint32_t n_in_w);
DNS_DECL_PREP(32, 1)
DNS_DECL_PREP(32, 2)
DNS_DECL_PREP(64, 1)
DNS_DECL_PREP(64, 2)
#undef DNS_DECL_PREP

// adds the workgroup's four copies of one 32 x 32 accumulator tile (rows = dW rows, lanes = dW columns) and issues the
// float atomics: one 128-byte row segment per lane half per instruction
template <uint32_t WAVE_FLOATS = STG_WAVE_FLOATS>
__device__ __forceinline__ void flush_tile(const f32x16& a, float* __restrict__ dst, uint32_t ld, uint32_t rows_valid,
                                           uint32_t cols_valid, float* __restrict__ stg_base, uint32_t wave, uint32_t lane) {
  float* stg = stg_base + wave * WAVE_FLOATS;
  const uint32_t j = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) stg[acc_row(r, h) * STG_LD + j] = a[r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = wave * 8u + 2u * i + h;
    float v = 0.f;
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) v += stg_base[w * WAVE_FLOATS + row * STG_LD + j];
    if (row < rows_valid && j < cols_valid && v != 0.f) atomicAdd(dst + (size_t)row * ld + j, v);
  }
  __syncthreads();
}


}  // namespace sp
}  // namespace dns

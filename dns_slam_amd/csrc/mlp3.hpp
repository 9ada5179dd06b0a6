// Bias-free ReLU MLP on the CDNA4 matrix cores, THREE-PART bf16 operands ("mlp3"): every fp32 operand v is carried as
// h = bf16(v), m = bf16(v - h), l = bf16(v - h - m) (8 + 8 + 8 significant bits, fp32's exponent range: no scaling, no
// per-point exponents, no maxima) and a product is six v_mfma_f32_32x32x16_bf16 with fp32 accumulation,
//     h h + (h m + m h) + (h l + l h + m m);      dropped terms < 2^-23 of the product.
//
// Replaces tcnn.Network{CutlassMLP} (reference models/decoder.py:58-64,84-90,101-116, slams/mapping.py:737-743):
// y = W_out relu(W_h relu(W_in x)), no bias.
//
// Round 3 redesign of the backward.  Round 2's kernels split activations into scaled f16 pairs for
// the products over a feature axis (three MFMAs) and into bf16 triples for the products over the point axis (six), and spent
// 13.5 vector instructions per matrix instruction doing it (two splits per tensor, maxima, ldexp, LDS transposes of fp32
// tiles, both orientations of every weight matrix in LDS).  Here every tensor is split ONCE into bf16 triples, in whatever
// layout it is born in; the 16-bit parts then go wherever they are needed:
//   * as they stand, as the B operand of the next layer (an accumulator's registers 8s .. 8s+7 are K-step s: CHAIN order);
//   * through a 16-bit LDS tile and back with ds_read_b64_tr_b16 (gfx950's transposing LDS read, T10 of the CDNA guide) as
//     point-major operands of the weight-gradient products;
//   * the weights live in LDS ONCE: one image per matrix serves the row-wise read (A = W, ds_read_b128 / ds_read_b64) and the
//     transposed read (A = W^T, ds_read_b64_tr_b16), where round 2 kept two images.  That halves the weight LDS and is what
//     lets dW_in join the kernel (no dH_1 workspace, no second streaming kernel).
// With twice the matrix instructions and a third of the vector instructions the kernel is bound by the matrix pipe at ONE
// wave per SIMD, which is all the weight-gradient accumulators leave room for.
//
// Lane maps (v_mfma_f32_32x32x16_bf16): lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] and B[k = 8h + j][col r],
// j = 0..7; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.  tools/mfma_lds_model.py restates every address function
// below on the host and checks the products and the LDS bank conflicts; tools/tr_read_probe.hip checks the transposing read's
// semantics on the device.
#pragma once
#include "common.hpp"

namespace dns {
namespace m3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Frag3 {
  bf16x8 h, m, l;
};

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

__device__ __forceinline__ uint32_t acc_row(uint32_t r, uint32_t h) { return (r & 3u) + 8u * (r >> 2) + 4u * h; }

// 8 fp32 values -> three bf16 parts (hipcc: v_cvt_pk_bf16_f32 for each pair; a NaN stays a NaN, an infinity becomes one)
__device__ __forceinline__ Frag3 split8(const float (&v)[8]) {
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

__device__ __forceinline__ f32x16 mma6(const Frag3& a, const Frag3& b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
}

// K-step `half` (0 / 1) of an accumulator tile as an operand (CHAIN order: element j = register 8 half + j)
template <int HALF>
__device__ __forceinline__ Frag3 split_half(const f32x16& a) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = a[8 * HALF + j];
  return split8(v);
}

// an accumulator tile -> the fragments of its two K-steps (CHAIN order: element j of K-step s = register 8s + j)
__device__ __forceinline__ void split_tile(const f32x16& a, Frag3& f0, Frag3& f1) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = a[j];
  f0 = split8(v);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = a[8 + j];
  f1 = split8(v);
}

// Lanes of ONE wave exchange data through LDS: the hardware executes a wave's LDS instructions in order, so all that is
// needed is that the compiler keeps the reads behind the writes -- a wavefront-scope fence, which emits no instruction and,
// unlike a wave barrier or an asm statement, leaves matrix and vector instructions free to be scheduled across it.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- LDS access ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned char* lds_ptr;

// every offset handed to these is a multiple of the access size by construction; said explicitly, because hipcc otherwise
// splits the 8- and 16-byte accesses into dwords (ds_write_b32 / ds_read2_b32: half the LDS rate)
__device__ __forceinline__ uint4 lds_read16(const unsigned char* base, uint32_t off) {
  return *reinterpret_cast<const uint4*>(__builtin_assume_aligned(base + off, 16));
}
__device__ __forceinline__ uint2 lds_read8(const unsigned char* base, uint32_t off) {
  return *reinterpret_cast<const uint2*>(__builtin_assume_aligned(base + off, 8));
}
__device__ __forceinline__ void lds_write8(unsigned char* base, uint32_t off, uint2 v) {
  *reinterpret_cast<uint2*>(__builtin_assume_aligned(base + off, 8)) = v;
}
__device__ __forceinline__ void lds_write16(unsigned char* base, uint32_t off, uint4 v) {
  *reinterpret_cast<uint4*>(__builtin_assume_aligned(base + off, 16)) = v;
}
// ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns of 16-bit elements; lane 4q + p of the group
// supplies the address of row q, columns 4p .. 4p+3; lane i receives column i, row q in element q.  EXEC must be all ones.
__device__ __forceinline__ uint2 lds_read_tr(const unsigned char* base, uint32_t off) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
  return *reinterpret_cast<const uint2*>(&v);
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return *reinterpret_cast<const bf16x8*>(&v); }
__device__ __forceinline__ bf16x8 as_bf16x8(uint2 a, uint2 b) {
  const uint4 v = make_uint4(a.x, a.y, b.x, b.y);
  return *reinterpret_cast<const bf16x8*>(&v);
}
__device__ __forceinline__ uint4 as_uint4(const bf16x8& v) { return *reinterpret_cast<const uint4*>(&v); }

// ---- weight images ----------------------------------------------------------------------------------------------
// One image per matrix M [R x C] (row-major fp32 in memory): three 16-bit planes (h, m, l), each [R_pad][C_pad] in 8-row x
// 32-column subtiles of 512 B, the 16-byte chunk ch of a row XOR-ed with (row >> 2) & 3 inside its subtile:
//   off(row, ch) = ((row >> 3) (C_pad / 32) + (ch >> 2)) 512 + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3))
// Conflict-free for the row read of the NAT order (ds_read_b128), the transposed reads of both orders (ds_read_b64_tr_b16);
// 2-way for the two 8-byte row reads of the CHAIN order (tools/mfma_lds_model.py).
struct WImg {
  uint32_t base;        // byte offset of plane h in the workgroup's LDS
  uint32_t plane;       // bytes per plane
  uint32_t cs;          // C_pad / 32: subtiles per 8-row group
};

__device__ __forceinline__ uint32_t w_off(uint32_t row, uint32_t ch, uint32_t cs) {
  return ((row >> 3) * cs + (ch >> 2)) * 512u + 64u * (row & 7u) + 16u * ((ch & 3u) ^ ((row >> 2) & 3u));
}

enum KOrder { K_NAT = 0, K_CHAIN = 1 };

// A = M: lane (r, h) element j = M[32 t + r][k(s, h, j)].  K_CHAIN_PERM: the image was built with w_image_build<true>, which
// swaps the two middle 4-column blocks of every 16 columns (column 16 s + 8 a + 4 b + e is stored at 16 s + 8 b + 4 a + e), so
// that the CHAIN order's eight elements of a lane half are ONE 16-byte chunk: a ds_read_b128 like the NAT read, conflict-free.
// Reading the CHAIN order from an unpermuted image takes two 8-byte reads per plane, which hipcc merges into
// ds_read2st64_b64 (banked mod 32, 16-lane groups): 4-way conflicts -- measured as an LDS-bound forward kernel (round 3).
enum { K_CHAIN_PERM = 2 };
template <int ORDER>
__device__ __forceinline__ Frag3 w_row_frag(const unsigned char* lds, const WImg& w, uint32_t t, uint32_t s, uint32_t lane) {
  Frag3 f;
  const uint32_t row = 32u * t + (lane & 31u), h = lane >> 5;
  if (ORDER == K_NAT || ORDER == K_CHAIN_PERM) {
    const uint32_t o = w.base + w_off(row, 2u * s + h, w.cs);
    f.h = as_bf16x8(lds_read16(lds, o));
    f.m = as_bf16x8(lds_read16(lds, o + w.plane));
    f.l = as_bf16x8(lds_read16(lds, o + 2u * w.plane));
  } else {
    const uint32_t o0 = w.base + w_off(row, 2u * s, w.cs) + 8u * h, o1 = w.base + w_off(row, 2u * s + 1u, w.cs) + 8u * h;
    f.h = as_bf16x8(lds_read8(lds, o0), lds_read8(lds, o1));
    f.m = as_bf16x8(lds_read8(lds, o0 + w.plane), lds_read8(lds, o1 + w.plane));
    f.l = as_bf16x8(lds_read8(lds, o0 + 2u * w.plane), lds_read8(lds, o1 + 2u * w.plane));
  }
  return f;
}

// A = M^T: lane (r, h) element j = M[k(s, h, j)][32 t + r]; two transposing reads per plane
template <int ORDER>
__device__ __forceinline__ Frag3 w_tr_frag(const unsigned char* lds, const WImg& w, uint32_t t, uint32_t s, uint32_t lane) {
  const uint32_t g = lane >> 4, wl = lane & 15u, h = g >> 1, gh = g & 1u, q = wl >> 2, p = wl & 3u;
  const uint32_t ch = 4u * t + 2u * gh + (p >> 1);
  uint32_t o[2];
#pragma unroll
  for (uint32_t u = 0; u < 2u; ++u) {
    const uint32_t krow = ORDER == K_NAT ? (16u * s + 8u * h + 4u * u + q) : (16u * s + 8u * u + 4u * h + q);
    o[u] = w.base + w_off(krow, ch, w.cs) + 8u * (p & 1u);
  }
  Frag3 f;
  f.h = as_bf16x8(lds_read_tr(lds, o[0]), lds_read_tr(lds, o[1]));
  f.m = as_bf16x8(lds_read_tr(lds, o[0] + w.plane), lds_read_tr(lds, o[1] + w.plane));
  f.l = as_bf16x8(lds_read_tr(lds, o[0] + 2u * w.plane), lds_read_tr(lds, o[1] + 2u * w.plane));
  return f;
}

__device__ __forceinline__ void lds_zero16(unsigned char* p, uint32_t bytes) {   // bytes % 16 == 0, whole workgroup
  uint4* q = reinterpret_cast<uint4*>(p);
  for (uint32_t e = threadIdx.x; e < bytes / 16u; e += blockDim.x) q[e] = make_uint4(0u, 0u, 0u, 0u);
}

// three packed dword pairs (4 consecutive elements of each part) of four fp32 values
struct Quad3 {
  uint2 h, m, l;
};
__device__ __forceinline__ Quad3 split4(float4 v) {
  const float a[4] = {v.x, v.y, v.z, v.w};
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 h, m, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = (__bf16)a[j];
    const float r = a[j] - (float)h[j];
    m[j] = (__bf16)r;
    l[j] = (__bf16)(r - (float)m[j]);
  }
  Quad3 q;
  q.h = *reinterpret_cast<const uint2*>(&h);
  q.m = *reinterpret_cast<const uint2*>(&m);
  q.l = *reinterpret_cast<const uint2*>(&l);
  return q;
}

// the whole workgroup writes M [R x C] (C % 4 == 0, rows beyond R / columns beyond C stay zero) into its image;
// PERM: columns stored in the chain-permuted order (see w_row_frag).  MAXQ >= ceil(R C / 4 / blockDim.x): every thread's
// float4 loads are issued FIRST and split afterwards -- as a plain loop (load, wait, split, store per iteration) the prologue
// paid one L2 round trip per quad, ~13 us of a 60 us launch whose waves only run four tiles each.
template <int MAXQ, bool PERM = false>
__device__ __forceinline__ void w_image_build(unsigned char* lds, const WImg& w, const float* __restrict__ M, uint32_t R, uint32_t C) {
  const uint32_t qpr = C >> 2, nq = R * qpr;
  const bool vec = (((uintptr_t)M) & 15u) == 0;
  float4 v[MAXQ];
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const uint32_t e = min(threadIdx.x + k * blockDim.x, nq - 1u);
    const float* src = M + (size_t)e * 4u;
    v[k] = vec ? *reinterpret_cast<const float4*>(src) : make_float4(src[0], src[1], src[2], src[3]);
  }
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const uint32_t e = threadIdx.x + k * blockDim.x;
    if (e < nq) {
      const uint32_t row = e / qpr, c4 = (e - row * qpr) * 4u;
      const Quad3 q = split4(v[k]);
      const uint32_t cp = PERM ? ((c4 & ~15u) | ((c4 & 4u) << 1) | ((c4 & 8u) >> 1)) : c4;
      const uint32_t o = w.base + w_off(row, cp >> 3, w.cs) + 2u * (cp & 7u);
      lds_write8(lds, o, q.h);
      lds_write8(lds, o + w.plane, q.m);
      lds_write8(lds, o + 2u * w.plane, q.l);
    }
  }
}

// ---- 32 x 32 tiles of 16-bit parts (wave-private slots of 3 x 2 KB) -------------------------------------------------
// T[row][col], 64-byte rows, chunk ch (8 elements) of a row at 64 row + 16 (ch ^ ((row >> 2) & 3)); planes 2 KB apart.
constexpr uint32_t TILE_PLANE = 2048u;
constexpr uint32_t SLOT_BYTES = 3u * TILE_PLANE;

__device__ __forceinline__ uint32_t t_off(uint32_t row, uint32_t ch) { return 64u * row + 16u * (ch ^ ((row >> 2) & 3u)); }

// the two K-step fragments (CHAIN order) of an accumulator tile -> T[point][feature]: lane (p, h) stores registers 4g .. 4g+3
// (features 8g + 4h ..) as 8 bytes at T[p][8g + 4h]; 4 ds_write_b64 per plane
__device__ __forceinline__ void t_store_acc(unsigned char* lds, uint32_t slot, const Frag3& f0, const Frag3& f1, uint32_t lane) {
  const uint32_t p = lane & 31u, h = lane >> 5;
  const uint4 parts[2][3] = {{as_uint4(f0.h), as_uint4(f0.m), as_uint4(f0.l)}, {as_uint4(f1.h), as_uint4(f1.m), as_uint4(f1.l)}};
#pragma unroll
  for (uint32_t s = 0; s < 2u; ++s)
#pragma unroll
    for (uint32_t pl = 0; pl < 3u; ++pl) {
      const uint4 v = parts[s][pl];
      lds_write8(lds, slot + pl * TILE_PLANE + t_off(p, 2u * s) + 8u * h, make_uint2(v.x, v.y));
      lds_write8(lds, slot + pl * TILE_PLANE + t_off(p, 2u * s + 1u) + 8u * h, make_uint2(v.z, v.w));
    }
}

// row-transposed operand of a stored tile: lane (c, h) element j = T[row 16 s + 8 h + j][column c]
__device__ __forceinline__ Frag3 t_tr_frag(const unsigned char* lds, uint32_t slot, uint32_t s, uint32_t lane) {
  const uint32_t g = lane >> 4, wl = lane & 15u, h = g >> 1, gh = g & 1u, q = wl >> 2, p = wl & 3u;
  uint32_t o[2];
#pragma unroll
  for (uint32_t u = 0; u < 2u; ++u) o[u] = slot + t_off(16u * s + 8u * h + 4u * u + q, 2u * gh + (p >> 1)) + 8u * (p & 1u);
  Frag3 f;
  f.h = as_bf16x8(lds_read_tr(lds, o[0]), lds_read_tr(lds, o[1]));
  f.m = as_bf16x8(lds_read_tr(lds, o[0] + TILE_PLANE), lds_read_tr(lds, o[1] + TILE_PLANE));
  f.l = as_bf16x8(lds_read_tr(lds, o[0] + 2u * TILE_PLANE), lds_read_tr(lds, o[1] + 2u * TILE_PLANE));
  return f;
}

// row-wise operand: lane (row, h) element j = T[row][16 s + 8 h + j]
__device__ __forceinline__ Frag3 t_row_frag(const unsigned char* lds, uint32_t slot, uint32_t s, uint32_t lane) {
  const uint32_t o = slot + t_off(lane & 31u, 2u * s + (lane >> 5));
  Frag3 f;
  f.h = as_bf16x8(lds_read16(lds, o));
  f.m = as_bf16x8(lds_read16(lds, o + TILE_PLANE));
  f.l = as_bf16x8(lds_read16(lds, o + 2u * TILE_PLANE));
  return f;
}

// ---- global I/O (row-coalesced through the tile's row table) ---------------------------------------------------------
struct XSeg {                                // optional second input segment: columns [n_in1, n_in) come from x2
  const float* x2;
  uint32_t ldx2, n_in1;
};

struct XChunk {                              // 32 columns of the tile's 32 rows in the LOAD layout: lane (r8 = l >> 3,
  float4 v[4];                               // c4 = l & 7) holds columns 4 c4 .. 4 c4 + 3 of rows r8, r8 + 8, r8 + 16, r8 + 24
};

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

// Unconditional loads (no exec-masked branches, so the four requests leave back to back): a padding slot reads row 0 and a
// column past n_in reads the row's last four -- finite stand-ins that cannot reach any result: a padding slot's dY is zero
// (hence every gradient it feeds), the W_in image's columns past n_in are zero, and dW_in's columns past n_in are not flushed.
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

// The next tile's x and dY rows are pulled into L2 a tile ahead by LDS-DMA loads of one dword per 64 bytes into a dump area
// (no destination registers; never read): the real requests, one chunk ahead of their use, then find their lines in L2
// (~300 cycles, which the 24 matrix instructions of a chunk cover) instead of in HBM (2000+ under load, which they do not).
__device__ __forceinline__ void prefetch_rows(const float* __restrict__ x, uint32_t ldx, const XSeg& seg, uint32_t n_in,
                                              const float* __restrict__ dy, uint32_t lddy, uint32_t n_out,
                                              const int* __restrict__ rows_lds, unsigned char* lds, uint32_t dump, uint32_t lane) {
  const int row = max(rows_lds[lane & 31u], 0);
  const uint32_t nt = (n_in + 15u) / 16u;
  for (uint32_t k = 0; 2u * k < nt; ++k) {      // uniform
    const uint32_t col = min(16u * (2u * k + (lane >> 5)), n_in - 4u);
    const bool second = seg.x2 != nullptr && col >= seg.n_in1;
    const float* src = second ? seg.x2 + (size_t)(uint32_t)row * seg.ldx2 + (col - seg.n_in1) : x + (size_t)(uint32_t)row * ldx + col;
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(lds + dump), 4, 0, 0);
  }
  const uint32_t no = (n_out + 15u) / 16u;
  for (uint32_t k = 0; 2u * k < no; ++k) {
    const uint32_t col = min(16u * (2u * k + (lane >> 5)), n_out - 1u);
    __builtin_amdgcn_global_load_lds(dy + (size_t)(uint32_t)row * lddy + col, (__attribute__((address_space(3))) void*)(lds + dump), 4, 0, 0);
  }
}

// split the chunk where it stands and write the tile image T[point][column]
__device__ __forceinline__ void x_chunk_commit(const XChunk& xc, unsigned char* lds, uint32_t slot, uint32_t lane) {
#pragma unroll
  for (uint32_t i = 0; i < 4u; ++i) {
    const Quad3 q = split4(xc.v[i]);
    const uint32_t o = slot + t_off((lane >> 3) + 8u * i, (lane & 7u) >> 1) + 8u * (lane & 1u);
    lds_write8(lds, o, q.h);
    lds_write8(lds, o + TILE_PLANE, q.m);
    lds_write8(lds, o + 2u * TILE_PLANE, q.l);
  }
}

// dY columns 32 c .. 32 c + 31 in the COLUMN layout: lane (col, hh = l >> 5), register 8 s + j = dy[row 16 s + 8 hh + j][col]
// (dword loads, two rows x 128 contiguous bytes per instruction, any lddy): already the point-major operand of dW_out
struct DyChunk {
  float v[16];
};
__device__ __forceinline__ void dy_chunk_issue(DyChunk& d, const float* __restrict__ dy, uint32_t lddy, uint32_t n_out,
                                               const int* __restrict__ rows_lds, uint32_t c, uint32_t lane) {
  const uint32_t col = 32u * c + (lane & 31u), hh = lane >> 5;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = rows_lds[16 * s + 8 * hh + j];
      d.v[8 * s + j] = (row >= 0 && col < n_out) ? dy[(size_t)row * lddy + col] : 0.f;
    }
}

// fp32 staging tile (32 x 36 floats = 4608 B, inside a slot) for row-coalesced stores of an accumulator tile
constexpr uint32_t STG_LD = 36;

struct DxSeg {
  float* dx2;
  uint32_t lddx2, acc1, acc2;                // acc: read-add-write instead of overwrite
};
struct DxOld {
  float4 v[4];
};

__device__ __forceinline__ void dx_old_issue(DxOld& o, const float* __restrict__ dst1, uint32_t ld1, const DxSeg& seg,
                                             uint32_t n_in1, uint32_t col0, uint32_t ncols, const int* __restrict__ rows_lds,
                                             uint32_t lane) {
  const uint32_t rr = lane >> 3, c4 = 4u * (lane & 7u);
  const uint32_t col = col0 + c4;
  const bool second = seg.dx2 != nullptr && col >= n_in1;
  const bool a1 = seg.acc1 != 0, a2 = seg.acc2 != 0;     // (selecting the FIELDS by a lane condition makes hipcc index the kernel
  const bool acc = (second && a2) || (!second && a1);   //  argument segment with a vector load, which drains every request in flight)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rows_lds[rr + 8 * i];
    o.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (acc && row >= 0 && c4 < ncols) {
      const float4* p = second ? reinterpret_cast<const float4*>(seg.dx2 + (size_t)row * seg.lddx2 + (col - n_in1))
                               : reinterpret_cast<const float4*>(dst1 + (size_t)row * ld1 + col);
      o.v[i] = *p;
    }
  }
}


// phase boundary: nothing is scheduled across it.  With 512 registers to fill, hipcc otherwise hoists the next phases'
// loads and fragment reads over the current one and spills the weight-gradient accumulators it was given the room for.
__device__ __forceinline__ void phase_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// a: 32 features x 32 points -> dst[row][col0 + f] (f < ncols, ncols % 4 == 0, 16-byte aligned rows), rows through the tile's
// row table; `old` is added; optional second segment for columns >= n_in1
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
    if (row >= 0 && c4 < ncols) {
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
__device__ __forceinline__ void store_tile_rows_scalar(float* __restrict__ dst1, uint32_t ld1, const DxSeg& seg, uint32_t n_in1,
                                                       uint32_t col0, uint32_t ncols, const int* __restrict__ rows_lds,
                                                       const f32x16& a, float* __restrict__ stg, uint32_t lane) {
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
      const bool a1 = seg.acc1 != 0, a2 = seg.acc2 != 0;
      if ((second && a2) || (!second && a1)) v += *p;
      *p = v;
    }
  }
  wave_lds_fence();
}

// adds the workgroup's four copies of one 32 x 32 accumulator tile (rows = dW rows, lanes = dW columns) in LDS and issues the
// float atomics: one 128-byte row segment per lane half per instruction.  red: 4 x 32 x 36 floats of LDS.
__device__ __forceinline__ void flush_tile(const f32x16& a, float* __restrict__ dst, uint32_t ld, uint32_t rows_valid,
                                           uint32_t cols_valid, float* __restrict__ red, uint32_t wave, uint32_t lane) {
  float* stg = red + wave * (32u * STG_LD);
  const uint32_t j = lane & 31u, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) stg[acc_row(r, h) * STG_LD + j] = a[r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = wave * 8u + 2u * i + h;
    float v = 0.f;
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) v += red[w * (32u * STG_LD) + row * STG_LD + j];
    if (row < rows_valid && j < cols_valid && v != 0.f) atomicAdd(dst + (size_t)row * ld + j, v);
  }
  __syncthreads();
}

// a: 32 output features x 32 points -> y[row][col0 + f], f < ncols (any ldy / alignment: dword stores with lane = feature,
// two rows x 128 contiguous bytes per instruction), rows through the tile's row table
__device__ __forceinline__ void store_tile_rows_plain(float* __restrict__ y, uint32_t ldy, uint32_t col0, uint32_t ncols,
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

}  // namespace m3
}  // namespace dns

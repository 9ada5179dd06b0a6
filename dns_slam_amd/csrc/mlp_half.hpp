// Bias-free ReLU MLP in tcnn's OWN arithmetic ("half rows", ABI v12): f16 activations, f16 weight operands, fp32 accumulation
// inside v_mfma_f32_32x32x16_f16, a STATIC loss scale on the gradients -- what tcnn.Network{CutlassMLP} computes for the
// reference (models/decoder.py:58-64,84-90,94,101-116: half-precision networks, `.float()` on their outputs; tcnn's
// default loss_scale = 128).  BASELINE configs[4] ("fp16 MFMA MLP path") runs on these kernels; the fp32-grade split-operand
// kernels of mlp_split.hpp stay the parity path.
//
// Nothing of the split-operand bookkeeping exists here: no row maxima, no per-point exponents, no ldexp, no hi / lo parts.
//   * activations travel between kernels as plain f16 rows [points][K] (the encoder and the feature block write them once);
//   * a layer's accumulator, ReLU-ed and packed to f16, IS the next layer's B operand (CHAIN order, as in mlp_split.hpp);
//   * the weights live in LDS ONCE per matrix: one f16 image serves the row-wise read (A = W: ds_read_b128 / 2 x ds_read_b64)
//     and gfx950's transposing read (A = W^T: ds_read_b64_tr_b16) -- half the LDS of two orientations, which is what lets
//     dW_in join the backward kernel (no dH_1 workspace, no second streaming kernel);
//   * the point-major operands of the weight-gradient products are 16-bit tiles written to LDS and read back transposed.
// The image layouts, the transposing-read addresses and their bank conflicts were restated and checked on the host in
// round 3 (tools/mfma_lds_model.py: every product exact on integer data, every read conflict-free but the CHAIN row read,
// 2-way) and the instruction's semantics on the device (tools/tr_read_probe.hip).
//
// Lane maps (v_mfma_f32_32x32x16_f16): lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] and B[k = 8h + j][col r],
// j = 0..7; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.  Points live on the column / lane axis.
#pragma once
#include "mlp_split.hpp"                     // buffer-addressed tile I/O, the fp32 staging-tile stores (namespace dns::sp)

namespace dns {
namespace hf {

using sp::f32x16;
using sp::half8;
using sp::zero16;
using sp::acc_row;
using sp::rsrc_t;
using sp::buf_make;
using sp::BUF_OOB;
using sp::BUF_LIMIT;
using sp::STG_LD;
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// phase boundary: nothing is scheduled across it (with 512 registers to fill hipcc otherwise hoists the next phases' loads and
// fragment reads over the current one and spills the weight-gradient accumulators it was given the room for)
// Between the batch of LDS reads that feeds a chain of matrix instructions and the chain: left to itself hipcc (at its register
// limit) sinks every fragment read to just in front of the instruction that uses it -- read, s_waitcnt lgkmcnt(0), matrix
// instruction, 60-odd times per tile, each wait the full LDS round trip of a wave that is alone on its SIMD (counters: 40 % of
// the wave's cycles in s_waitcnt).  With the reads issued first the waits in front of the matrix instructions become counted
// lgkmcnt(N): the fragments stream in underneath the chain.
__device__ __forceinline__ void reads_issued() { __builtin_amdgcn_sched_barrier(0); }

__device__ __forceinline__ void phase_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ---- LDS access: every offset is a multiple of the access size by construction ---------------------------------------
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
// ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns of 16-bit elements; lane 4q + p of the group supplies
// the address of row q, columns 4p .. 4p+3; lane i receives column i, row q in element q.  EXEC must be all ones.
__device__ __forceinline__ uint2 lds_read_tr(const unsigned char* base, uint32_t off) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
  return *reinterpret_cast<const uint2*>(&v);
}
__device__ __forceinline__ half8 as_half8(uint4 v) { return *reinterpret_cast<const half8*>(&v); }
__device__ __forceinline__ half8 as_half8(uint2 a, uint2 b) {
  const uint4 v = make_uint4(a.x, a.y, b.x, b.y);
  return *reinterpret_cast<const half8*>(&v);
}
__device__ __forceinline__ uint4 as_uint4(const half8& v) { return *reinterpret_cast<const uint4*>(&v); }

__device__ __forceinline__ f32x16 mma(const half8& a, const half8& b, f32x16 acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}

// registers 8 HALF .. 8 HALF + 7 of an accumulator tile as a K-step operand (CHAIN order: element j = register 8 HALF + j)
template <int HALF>
__device__ __forceinline__ half8 pack_half(const f32x16& a) {
  half8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (_Float16)a[8 * HALF + j];
  return r;
}

// ---- packed 16-bit helpers (VOP3P: one instruction per two values) -------------------------------------------------------
// ReLU on packed f16 as SIGNED 16-bit integers: a negative float (sign bit set, -0 included) is a negative integer, a
// non-negative one keeps its bits -- max(x, 0) is exactly relu, and f16(relu(v)) == relu(f16(v)) (rounding keeps the sign).
__device__ __forceinline__ uint32_t pk_relu(uint32_t x) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(x));
  return r;
}
// ReLU' mask of two ACTIVATIONS (relu-ed f16: bits >= 0 as signed integers): 0xffff per half whose activation is > 0
__device__ __forceinline__ uint32_t pk_pos_mask(uint32_t act) {
  uint32_t n, m;
  asm("v_pk_sub_i16 %0, 0, %1" : "=v"(n) : "v"(act));          // -bits: negative exactly where the activation is non-zero
  // sign fill (op_sel_hi [0, 1]: BOTH halves take their shift count from the low half of the inline constant -- by default the
  // high half would read the constant's upper 16 bits, i.e. shift by 0)
  asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(m) : "v"(n));
  return m;
}
__device__ __forceinline__ half8 relu8(const half8& v) {
  const uint4 u = as_uint4(v);
  return as_half8(make_uint4(pk_relu(u.x), pk_relu(u.y), pk_relu(u.z), pk_relu(u.w)));
}
// gradient operand (.) relu'(activation operand): 3 packed instructions per two values on operands that exist anyway -- the fp32
// compare-and-select needs the fp32 activations alive until the gradient arrives (64 registers at 64 x 2)
__device__ __forceinline__ half8 relu_mask(const half8& grad, const half8& act) {
  const uint4 g = as_uint4(grad), h = as_uint4(act);
  return as_half8(make_uint4(g.x & pk_pos_mask(h.x), g.y & pk_pos_mask(h.y), g.z & pk_pos_mask(h.z), g.w & pk_pos_mask(h.w)));
}

// ---- LDS images and the lane's part of their addresses -------------------------------------------------------------------
// Weight image of M [R x C], f16: [R_pad][C_pad] in 8-row x 32-column subtiles of 512 B, the 16-byte chunk ch of a row XOR-ed
// with (row >> 2) & 3 inside its subtile:
//   w_off(row, ch) = ((row >> 3) (C_pad / 32) + (ch >> 2)) 512 + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3))
// Tile T[32 rows][32 cols] f16 (wave-private, 2 KB): 64-byte rows, chunk ch of a row at t_off = 64 row + 16 (ch ^ ((row >> 2) & 3)).
// K orders of an operand fragment: NAT k = 16 s + 8 h + j;  CHAIN k = 16 s + 8 (j >> 2) + 4 h + (j & 3) (an accumulator's registers).
//
// Every address below is  (a function of the LANE)  +  (a compile-time constant of the fragment: tile, K-step, image).  The lane
// parts -- 19 values -- are formed ONCE per kernel (LaneAddr) and the constants ride in the LDS instructions' 16-bit offset
// fields: the first version of this file formed each address where it was used and spent 650 of a tile's 1500 vector
// instructions on it (or, when hipcc hoisted them out of the tile loop, 327 registers).
__device__ __forceinline__ uint32_t w_off(uint32_t row, uint32_t ch, uint32_t cs) {
  return ((row >> 3) * cs + (ch >> 2)) * 512u + 64u * (row & 7u) + 16u * ((ch & 3u) ^ ((row >> 2) & 3u));
}
constexpr uint32_t TILE_BYTES = 2048u;
__device__ __forceinline__ uint32_t t_off(uint32_t row, uint32_t ch) { return 64u * row + 16u * (ch ^ ((row >> 2) & 3u)); }

struct LaneAddr {
  uint32_t wrow_nat[2];     // A = W_in, NAT: K-step parity (cs = cs_in)
  uint32_t wrow_chain[4];   // A = W_h / W_out, CHAIN: chunk 2 s + e of the row, (2 (s & 1) + e) = 0..3 (cs = NN / 32)
  uint32_t wtr_nat[2];      // A = W_out^T, NAT: u = 0, 1 (cs = NN / 32)
  uint32_t wtr_chain[2];    // A = W^T, CHAIN: u = 0, 1 (any cs)
  uint32_t t_row[2];        // tile, row-wise operand / 16-byte row-chunk store: K-step s = 0, 1     (+ base of the wave's area)
  uint32_t t_tr[2];         // tile, transposed operand: u = 0, 1                                         (+ base)
  uint32_t t_st[4];         // tile, accumulator store: chunk c = 0..3                                    (+ base)
  uint32_t x_st;            // tile, 16-byte store of the input rows' load layout                         (+ base)
};
// cs_in: subtiles per 8-row group of the W_in image; cs_h: of the W_h / W_out images; wave_base: byte offset of the wave's area
__device__ __forceinline__ LaneAddr lane_addr(uint32_t lane, uint32_t cs_in, uint32_t cs_h, uint32_t wave_base) {
  LaneAddr A;
  const uint32_t r = lane & 31u, h = lane >> 5;
  const uint32_t g = lane >> 4, wl = lane & 15u, gh = g & 1u, q = wl >> 2, p = wl & 3u;   // (h = g >> 1)
  const uint32_t rx = (r >> 2) & 3u, cp = 2u * gh + (p >> 1);
#pragma unroll
  for (uint32_t par = 0; par < 2u; ++par) A.wrow_nat[par] = (r >> 3) * cs_in * 512u + 64u * (r & 7u) + 16u * ((2u * par + h) ^ rx);
#pragma unroll
  for (uint32_t c = 0; c < 4u; ++c) A.wrow_chain[c] = (r >> 3) * cs_h * 512u + 64u * (r & 7u) + 16u * (c ^ rx) + 8u * h;
#pragma unroll
  for (uint32_t u = 0; u < 2u; ++u) {
    A.wtr_nat[u] = h * cs_h * 512u + 64u * q + 16u * (cp ^ (2u * h + u)) + 8u * (p & 1u);
    A.wtr_chain[u] = 64u * (4u * h + q) + 16u * (cp ^ (2u * u + h)) + 8u * (p & 1u);
    A.t_row[u] = wave_base + 64u * r + 16u * ((2u * u + h) ^ rx);
    A.t_tr[u] = wave_base + 512u * h + 64u * q + 16u * (cp ^ (2u * h + u)) + 8u * (p & 1u);
  }
#pragma unroll
  for (uint32_t c = 0; c < 4u; ++c) A.t_st[c] = wave_base + 64u * r + 16u * (c ^ rx) + 8u * h;
  A.x_st = wave_base + 64u * (lane >> 2) + 16u * ((lane & 3u) ^ ((lane >> 4) & 3u));
  return A;
}
// Through an empty asm at the top of every tile: the values stay where they are, but hipcc may not fold them with the constants
// into ~150 loop-invariant sums held in registers across the loop.
__device__ __forceinline__ void lane_addr_pin(LaneAddr& A) {
#pragma unroll
  for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(A.wrow_nat[i]), "+v"(A.wtr_nat[i]), "+v"(A.wtr_chain[i]), "+v"(A.t_row[i]), "+v"(A.t_tr[i]));
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(A.wrow_chain[i]), "+v"(A.t_st[i]));
  asm volatile("" : "+v"(A.x_st));
}

struct WImg {
  uint32_t base;        // byte offset in the workgroup's LDS
  uint32_t cs;          // C_pad / 32: subtiles per 8-row group
};
enum KOrder { K_NAT = 0, K_CHAIN = 1 };

// A = M: lane (r, h) element j = M[32 t + r][k(s, h, j)].  NAT: the W_in image (cs = cs_in); CHAIN: W_h / W_out (cs = cs_h)
template <int ORDER>
__device__ __forceinline__ half8 w_row_frag(const unsigned char* lds, const LaneAddr& A, const WImg& w, uint32_t t, uint32_t s) {
  const uint32_t c0 = w.base + (4u * t * w.cs + (s >> 1)) * 512u;
  if (ORDER == K_NAT) return as_half8(lds_read16(lds, A.wrow_nat[s & 1u] + c0));
  return as_half8(lds_read8(lds, A.wrow_chain[2u * (s & 1u)] + c0), lds_read8(lds, A.wrow_chain[2u * (s & 1u) + 1u] + c0));
}
// A = M^T: lane (r, h) element j = M[k(s, h, j)][32 t + r]; two transposing reads.  NAT: the W_out image (cs = cs_h); CHAIN: any
template <int ORDER>
__device__ __forceinline__ half8 w_tr_frag(const unsigned char* lds, const LaneAddr& A, const WImg& w, uint32_t t, uint32_t s) {
  if (ORDER == K_NAT) {
    const uint32_t c0 = w.base + (2u * s * w.cs + t) * 512u;
    return as_half8(lds_read_tr(lds, A.wtr_nat[0] + c0), lds_read_tr(lds, A.wtr_nat[1] + c0 + 256u));
  }
  const uint32_t c0 = w.base + (2u * s * w.cs + t) * 512u;
  return as_half8(lds_read_tr(lds, A.wtr_chain[0] + c0), lds_read_tr(lds, A.wtr_chain[1] + c0 + w.cs * 512u));
}

__device__ __forceinline__ void lds_zero16(unsigned char* p, uint32_t bytes) {   // bytes % 16 == 0, whole workgroup
  uint4* q = reinterpret_cast<uint4*>(p);
  for (uint32_t e = threadIdx.x; e < bytes / 16u; e += blockDim.x) q[e] = make_uint4(0u, 0u, 0u, 0u);
}

// the whole workgroup converts M [R x C] (row stride ld floats, C % 4 == 0; rows beyond R / columns beyond C stay zero) into its image
__device__ __forceinline__ void w_image_build(unsigned char* lds, const WImg& w, const float* __restrict__ M, uint32_t R, uint32_t C,
                                              uint32_t ld) {
  const uint32_t qpr = C >> 2, nq = R * qpr;
  const bool vec = ((((uintptr_t)M) & 15u) == 0) && ((ld & 3u) == 0);
  for (uint32_t e = threadIdx.x; e < nq; e += blockDim.x) {
    const uint32_t row = e / qpr, c4 = (e - row * qpr) * 4u;
    const float* src = M + (size_t)row * ld + c4;
    const float4 v = vec ? *reinterpret_cast<const float4*>(src) : make_float4(src[0], src[1], src[2], src[3]);
    typedef _Float16 half4v __attribute__((ext_vector_type(4)));
    half4v hv;
    hv[0] = (_Float16)v.x; hv[1] = (_Float16)v.y; hv[2] = (_Float16)v.z; hv[3] = (_Float16)v.w;
    lds_write8(lds, w.base + w_off(row, c4 >> 3, w.cs) + 2u * (c4 & 7u), *reinterpret_cast<const uint2*>(&hv));
  }
}

// The same in two steps, for a prologue that is a large share of a small launch (the forward kernel: 10.7 us at one tile per
// wave): ALL of a weight set's loads are put in flight first (w_image_load; the rolled loop above is a dependent load -> convert
// -> store chain per trip), the image area is cleared while they fly, then the values are converted and stored (w_image_store).
// MAXQ >= R C / 4 / blockDim.x.
template <int MAXQ>
struct WQuads {
  float4 v[MAXQ];
};
template <int MAXQ>
__device__ __forceinline__ void w_image_load(WQuads<MAXQ>& q, const float* __restrict__ M, uint32_t R, uint32_t C, uint32_t ld) {
  const uint32_t qpr = C >> 2, nq = R * qpr;
  const bool vec = ((((uintptr_t)M) & 15u) == 0) && ((ld & 3u) == 0);
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    const uint32_t e = min(threadIdx.x + (uint32_t)j * blockDim.x, nq - 1u);      // (past the end: the last quad again, not stored)
    const uint32_t row = e / qpr, c4 = (e - row * qpr) * 4u;
    const float* src = M + (size_t)row * ld + c4;
    if (vec) q.v[j] = *reinterpret_cast<const float4*>(src);
    else q.v[j] = make_float4(src[0], src[1], src[2], src[3]);
  }
}
template <int MAXQ>
__device__ __forceinline__ void w_image_store(const WQuads<MAXQ>& q, unsigned char* lds, const WImg& w, uint32_t R, uint32_t C) {
  const uint32_t qpr = C >> 2, nq = R * qpr;
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    const uint32_t e = threadIdx.x + (uint32_t)j * blockDim.x;
    if (e < nq) {
      const uint32_t row = e / qpr, c4 = (e - row * qpr) * 4u;
      typedef _Float16 half4v __attribute__((ext_vector_type(4)));
      half4v hv;
      hv[0] = (_Float16)q.v[j].x; hv[1] = (_Float16)q.v[j].y; hv[2] = (_Float16)q.v[j].z; hv[3] = (_Float16)q.v[j].w;
      lds_write8(lds, w.base + w_off(row, c4 >> 3, w.cs) + 2u * (c4 & 7u), *reinterpret_cast<const uint2*>(&hv));
    }
  }
}

// ---- 32 x 32 tiles of f16 (wave-private, 2 KB); `slot` = byte offset from the wave's area ---------------------------------------
// the two K-step fragments (CHAIN order) of an accumulator tile -> T[point][feature]: lane (p, h) stores registers 4g .. 4g+3
// (features 8g + 4h ..) as 8 bytes at T[p][8g + 4h]; 4 ds_write_b64
__device__ __forceinline__ void t_store_acc(unsigned char* lds, const LaneAddr& A, uint32_t slot, const half8& f0, const half8& f1) {
  const uint4 v0 = as_uint4(f0), v1 = as_uint4(f1);
  lds_write8(lds, A.t_st[0] + slot, make_uint2(v0.x, v0.y));
  lds_write8(lds, A.t_st[1] + slot, make_uint2(v0.z, v0.w));
  lds_write8(lds, A.t_st[2] + slot, make_uint2(v1.x, v1.y));
  lds_write8(lds, A.t_st[3] + slot, make_uint2(v1.z, v1.w));
}
// row-transposed operand of a stored tile: lane (c, h) element j = T[row 16 s + 8 h + j][column c]
__device__ __forceinline__ half8 t_tr_frag(const unsigned char* lds, const LaneAddr& A, uint32_t slot, uint32_t s) {
  return as_half8(lds_read_tr(lds, A.t_tr[0] + slot + 1024u * s), lds_read_tr(lds, A.t_tr[1] + slot + 1024u * s + 256u));
}
// row-wise operand: lane (row, h) element j = T[row][16 s + 8 h + j]
__device__ __forceinline__ half8 t_row_frag(const unsigned char* lds, const LaneAddr& A, uint32_t slot, uint32_t s) {
  return as_half8(lds_read16(lds, A.t_row[s] + slot));
}

// ---- the tile's input rows: f16 [rows][ld] in memory (two segments) -> IT images T_c[point][column 32 c ..] --------------------
struct HSeg {                                // optional second input segment: columns [n_in1, n_in) come from x2
  const _Float16* x2;
  uint32_t ldx2, n_in1;                      // halfs
};
template <int IT>
struct XRegs {
  uint4 v[IT][2];                            // chunk c: lane (row = (l >> 2) + 16 i, piece = l & 3) holds columns 32 c + 8 piece .. + 7
};
// Unconditional 16-byte loads (no exec-masked branches: the requests leave back to back).  A padding slot reads row 0 and a
// piece past n_in re-reads the row's last one -- finite stand-ins that cannot reach a result: a padding slot's dY is zero
// (hence every gradient it feeds) and its outputs are not stored; K-steps past n_in are never run and the columns of dW_in
// past n_in are not flushed.
template <int IT>
__device__ __forceinline__ void x_tile_issue(XRegs<IT>& xr, const _Float16* __restrict__ x, uint32_t ldx, const HSeg& seg, uint32_t n_in,
                                             const int* __restrict__ rows_lds, uint32_t lane) {
  int rows[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) rows[i] = max(rows_lds[(lane >> 2) + 16 * i], 0);
#pragma unroll
  for (int c = 0; c < IT; ++c) {
    const uint32_t col = min(32u * c + 8u * (lane & 3u), n_in - 8u);
    const bool second = seg.x2 != nullptr && col >= seg.n_in1;
    const _Float16* base = second ? seg.x2 + (col - seg.n_in1) : x + col;
    const uint32_t ld = second ? seg.ldx2 : ldx;
#pragma unroll
    for (int i = 0; i < 2; ++i) xr.v[c][i] = *reinterpret_cast<const uint4*>(base + (size_t)(uint32_t)rows[i] * ld);
  }
}
template <int IT>
__device__ __forceinline__ void x_tile_commit(const XRegs<IT>& xr, unsigned char* lds, const LaneAddr& A, uint32_t ximg) {
#pragma unroll
  for (int c = 0; c < IT; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_write16(lds, A.x_st + ximg + TILE_BYTES * c + 1024u * i, xr.v[c][i]);
}

// ---- dY: fp32 rows [rows][lddy] -> the COLUMN layout: lane (col, hh = l >> 5), register 8 s + j = dy[row 16 s + 8 hh + j][col]
// (dword buffer loads, two rows x 128 contiguous bytes per instruction, any lddy): already the point-major operand of dW_out.
// A padding slot (row offset BUF_OOB) and a column past n_out read zeros.
struct DyRegs {
  float v[16];
};
__device__ __forceinline__ void dy_issue(DyRegs& d, rsrc_t rdy, const uint32_t* __restrict__ dyoff, uint32_t n_out, uint32_t c, uint32_t lane) {
  const uint32_t col = 32u * c + (lane & 31u), hh = lane >> 5;
  uint32_t off[16];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) off[8 * s + j] = dyoff[16 * s + 8 * hh + j];
#pragma unroll
  for (int i = 0; i < 16; ++i) d.v[i] = sp::buf_load1(rdy, col < n_out ? off[i] + 4u * col : BUF_OOB);
}
// times the loss scale, rounded to f16, into the image U[output column][point] (one 16-byte store per K-step)
__device__ __forceinline__ void dy_commit(const DyRegs& d, float scale, unsigned char* lds, const LaneAddr& A, uint32_t slot) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    half8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (_Float16)(d.v[8 * s + j] * scale);
    lds_write16(lds, A.t_row[s] + slot, as_uint4(f));
  }
}

// n_out = 33 (hidden + 1: the coarse and the fine networks -- every MT = 2 launch of the step): the second 32-column chunk of dY
// holds ONE live column.  Loaded as that column alone -- lane = point, one dword -- instead of 16 dword loads whose 31 other lanes
// read nothing, and written as row 0 of a zeroed U_1 image: 17 registers and loads per tile instead of 32.
__device__ __forceinline__ void dy_issue_col32(float& v, rsrc_t rdy, const uint32_t* __restrict__ dyoff, uint32_t lane) {
  v = sp::buf_load1(rdy, dyoff[lane & 31u] + 4u * 32u);            // (a padding slot's offset is BUF_OOB: + 128 stays out of range)
}
__device__ __forceinline__ void dy_commit_col32(float v, float scale, unsigned char* lds, uint32_t wave_base, uint32_t slot, uint32_t lane) {
  // rows 1 .. 31 of U_1 are zero, row 0 = the column: the tile is cleared (two 16-byte stores per lane) and row 0 written behind it
  // (a wave's LDS instructions execute in order); the clear is needed every tile where the staging tile of the dX stores lies over it
  const uint32_t base = wave_base + slot;
  lds_write16(lds, base + 32u * lane, make_uint4(0u, 0u, 0u, 0u));
  lds_write16(lds, base + 32u * lane + 16u, make_uint4(0u, 0u, 0u, 0u));
  const _Float16 hv = (_Float16)(v * scale);
  if (lane < 32u) *reinterpret_cast<_Float16*>(lds + base + t_off(0u, lane >> 3) + 2u * (lane & 7u)) = hv;
}

// ---- kernel arguments ------------------------------------------------------------------------------------------------------
struct HFwdArgs {
  const _Float16* x;
  uint32_t ldx;
  HSeg seg;
  const float* params;
  uint32_t n_in, n_in_w, n_out;              // live input width; row stride of W_in in params (>= n_in); outputs
  float* y;
  uint32_t ldy, n_slots;
  const int32_t* row_index;
  const int32_t* tile_group;
  uint32_t param_stride, tiles_per_block;
  uint32_t* err;
};
struct HBwdArgs {
  const _Float16* x;
  uint32_t ldx;
  HSeg seg;
  const float* dy;
  uint32_t lddy;
  const float* params;
  uint32_t n_in, n_in_w, n_out;
  float* dx;
  uint32_t lddx;
  sp::DxSeg dseg;
  float* d_params;
  uint32_t n_slots;
  const int32_t* row_index;
  const int32_t* tile_group;
  uint32_t param_stride, tiles_per_block;
  float scale, inv_scale;                    // the static loss scale S on dY (tcnn: 128) and 1 / S on everything that leaves
  uint32_t* err;
};

// per-(n_neurons, n_hidden_layers) launchers of the backward kernel: one translation unit each (mlp_half_bwd_*.hip)
int launch_hbwd_32_1(const HBwdArgs& a, uint32_t blocks, hipStream_t st);
int launch_hbwd_32_2(const HBwdArgs& a, uint32_t blocks, hipStream_t st);
int launch_hbwd_64_1(const HBwdArgs& a, uint32_t blocks, hipStream_t st);
int launch_hbwd_64_2(const HBwdArgs& a, uint32_t blocks, hipStream_t st);

// adds the workgroup's four copies of one 32 x 32 accumulator tile (rows = dW rows, lanes = dW columns) in LDS, takes the loss
// scale out and issues the float atomics: one 128-byte row segment per lane half per instruction.  red: 4 x 32 x 36 floats.
__device__ __forceinline__ void flush_tile(const f32x16& a, float* __restrict__ dst, uint32_t ld, uint32_t rows_valid, uint32_t cols_valid,
                                           float inv_scale, float* __restrict__ red, uint32_t wave, uint32_t lane) {
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
    v *= inv_scale;
    if (row < rows_valid && j < cols_valid && v != 0.f) atomicAdd(dst + (size_t)row * ld + j, v);
  }
  __syncthreads();
}

}  // namespace hf
}  // namespace dns

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
// `lane` passes through an empty asm: every LDS address is a function of the lane index and constant over the tile loop, so hipcc
// otherwise computes ALL of a tile's ~600 addresses once, in front of the loop, and keeps them in registers across it (measured:
// 327 vector registers of working set beside 224 accumulators, 71 spilled); behind the asm the addresses of a phase are formed
// in that phase and die with it.
__device__ __forceinline__ void phase_fence(uint32_t& lane) {
  asm volatile("" : "+v"(lane) : : "memory");
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

// registers 8 HALF .. 8 HALF + 7 of an accumulator tile, times f, as a K-step operand (CHAIN order: element j = register 8 HALF + j)
template <int HALF>
__device__ __forceinline__ half8 pack_half(const f32x16& a, float f = 1.0f) {
  half8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (_Float16)(a[8 * HALF + j] * f);
  return r;
}

// ReLU' of a K-step operand of ACTIVATIONS (f16, >= +0 after the ReLU: an accumulator that starts at +0 never sums to -0) applied
// to the matching K-step operand of gradients: a half passes where the activation's bits are non-zero.  Three packed
// instructions per two values (v_pk_min_u16, v_pk_mul_lo_u16 / v_pk_sub_u16, v_and_b32) on operands that exist anyway -- the
// fp32 compare-and-select needs the fp32 activations alive until the gradient arrives (64 registers at 64 x 2).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half8 relu_mask(const half8& grad, const half8& act) {
  const uint4 g = *reinterpret_cast<const uint4*>(&grad), h = *reinterpret_cast<const uint4*>(&act);
  const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, hw[4] = {h.x, h.y, h.z, h.w};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u16x2 one = {1, 1}, zero = {0, 0};
    const u16x2 hv = *reinterpret_cast<const u16x2*>(&hw[i]);
    const u16x2 m = zero - __builtin_elementwise_min(hv, one);       // 0 or 0xffff per half
    o[i] = gw[i] & *reinterpret_cast<const uint32_t*>(&m);
  }
  const uint4 r = make_uint4(o[0], o[1], o[2], o[3]);
  return *reinterpret_cast<const half8*>(&r);
}

// ---- weight images ------------------------------------------------------------------------------------------------------
// One f16 image per matrix M [R x C]: [R_pad][C_pad] in 8-row x 32-column subtiles of 512 B, the 16-byte chunk ch of a row
// XOR-ed with (row >> 2) & 3 inside its subtile:
//   off(row, ch) = ((row >> 3) (C_pad / 32) + (ch >> 2)) 512 + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3))
struct WImg {
  uint32_t base;        // byte offset in the workgroup's LDS
  uint32_t cs;          // C_pad / 32: subtiles per 8-row group
};
__device__ __forceinline__ uint32_t w_off(uint32_t row, uint32_t ch, uint32_t cs) {
  return ((row >> 3) * cs + (ch >> 2)) * 512u + 64u * (row & 7u) + 16u * ((ch & 3u) ^ ((row >> 2) & 3u));
}
enum KOrder { K_NAT = 0, K_CHAIN = 1 };      // NAT k = 16 s + 8 h + j;  CHAIN k = 16 s + 8 (j >> 2) + 4 h + (j & 3)

// A = M: lane (r, h) element j = M[32 t + r][k(s, h, j)]
template <int ORDER>
__device__ __forceinline__ half8 w_row_frag(const unsigned char* lds, const WImg& w, uint32_t t, uint32_t s, uint32_t lane) {
  const uint32_t row = 32u * t + (lane & 31u), h = lane >> 5;
  if (ORDER == K_NAT) return as_half8(lds_read16(lds, w.base + w_off(row, 2u * s + h, w.cs)));
  return as_half8(lds_read8(lds, w.base + w_off(row, 2u * s, w.cs) + 8u * h), lds_read8(lds, w.base + w_off(row, 2u * s + 1u, w.cs) + 8u * h));
}
// A = M^T: lane (r, h) element j = M[k(s, h, j)][32 t + r]; two transposing reads
template <int ORDER>
__device__ __forceinline__ half8 w_tr_frag(const unsigned char* lds, const WImg& w, uint32_t t, uint32_t s, uint32_t lane) {
  const uint32_t g = lane >> 4, wl = lane & 15u, h = g >> 1, gh = g & 1u, q = wl >> 2, p = wl & 3u;
  const uint32_t ch = 4u * t + 2u * gh + (p >> 1);
  uint32_t o[2];
#pragma unroll
  for (uint32_t u = 0; u < 2u; ++u) {
    const uint32_t krow = ORDER == K_NAT ? (16u * s + 8u * h + 4u * u + q) : (16u * s + 8u * u + 4u * h + q);
    o[u] = w.base + w_off(krow, ch, w.cs) + 8u * (p & 1u);
  }
  return as_half8(lds_read_tr(lds, o[0]), lds_read_tr(lds, o[1]));
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

// ---- 32 x 32 tiles of f16 (wave-private, 2 KB) ---------------------------------------------------------------------------
// T[row][col], 64-byte rows, chunk ch (8 elements) of a row at 64 row + 16 (ch ^ ((row >> 2) & 3)).
constexpr uint32_t TILE_BYTES = 2048u;
__device__ __forceinline__ uint32_t t_off(uint32_t row, uint32_t ch) { return 64u * row + 16u * (ch ^ ((row >> 2) & 3u)); }

// the two K-step fragments (CHAIN order) of an accumulator tile -> T[point][feature]: lane (p, h) stores registers 4g .. 4g+3
// (features 8g + 4h ..) as 8 bytes at T[p][8g + 4h]; 4 ds_write_b64
__device__ __forceinline__ void t_store_acc(unsigned char* lds, uint32_t slot, const half8& f0, const half8& f1, uint32_t lane) {
  const uint32_t p = lane & 31u, h = lane >> 5;
  const uint4 v0 = as_uint4(f0), v1 = as_uint4(f1);
  lds_write8(lds, slot + t_off(p, 0u) + 8u * h, make_uint2(v0.x, v0.y));
  lds_write8(lds, slot + t_off(p, 1u) + 8u * h, make_uint2(v0.z, v0.w));
  lds_write8(lds, slot + t_off(p, 2u) + 8u * h, make_uint2(v1.x, v1.y));
  lds_write8(lds, slot + t_off(p, 3u) + 8u * h, make_uint2(v1.z, v1.w));
}
// row-transposed operand of a stored tile: lane (c, h) element j = T[row 16 s + 8 h + j][column c]
__device__ __forceinline__ half8 t_tr_frag(const unsigned char* lds, uint32_t slot, uint32_t s, uint32_t lane) {
  const uint32_t g = lane >> 4, wl = lane & 15u, h = g >> 1, gh = g & 1u, q = wl >> 2, p = wl & 3u;
  uint32_t o[2];
#pragma unroll
  for (uint32_t u = 0; u < 2u; ++u) o[u] = slot + t_off(16u * s + 8u * h + 4u * u + q, 2u * gh + (p >> 1)) + 8u * (p & 1u);
  return as_half8(lds_read_tr(lds, o[0]), lds_read_tr(lds, o[1]));
}
// row-wise operand: lane (row, h) element j = T[row][16 s + 8 h + j]
__device__ __forceinline__ half8 t_row_frag(const unsigned char* lds, uint32_t slot, uint32_t s, uint32_t lane) {
  return as_half8(lds_read16(lds, slot + t_off(lane & 31u, 2u * s + (lane >> 5))));
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
__device__ __forceinline__ void x_tile_commit(const XRegs<IT>& xr, unsigned char* lds, uint32_t ximg, uint32_t lane) {
#pragma unroll
  for (int c = 0; c < IT; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_write16(lds, ximg + TILE_BYTES * c + t_off((lane >> 2) + 16u * i, lane & 3u), xr.v[c][i]);
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
__device__ __forceinline__ void dy_commit(const DyRegs& d, float scale, unsigned char* lds, uint32_t slot, uint32_t lane) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    half8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (_Float16)(d.v[8 * s + j] * scale);
    lds_write16(lds, slot + t_off(lane & 31u, 2u * s + (lane >> 5)), as_uint4(f));
  }
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

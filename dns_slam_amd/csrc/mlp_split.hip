// Forward kernel, dW_in kernel and the host-side launchers of the split-operand MLP (device helpers: mlp_split.hpp; the
// backward kernel: mlp_split_bwd.inc, one translation unit per (n_neurons, n_hidden_layers)).
#include <cstdlib>
#include "mlp_split.hpp"

namespace dns {
namespace sp {
#ifdef DNS_BWD_TRACE
unsigned long long* g_bwd_trace = nullptr;
#endif

template <int NN, int NL, int PREC, bool XS = false>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  mlp_fwd_body<NN, NL, PREC, XS>(a, blockIdx.x, lds);
}

// =================================================================================================================
// dW_in[n][c] = sum_points dH_1[point][n] * X[point][c]: a streaming product whose K axis is the point axis.  Both
// operands are read straight from memory in the form the matrix instruction wants -- lane = channel, elements = 8
// consecutive points: dword loads, 2 rows x 128 contiguous bytes per instruction (dH_1 is slot-major, X goes through the
// slot -> row table) -- split into three bf16 parts (scale-free, mlp_split.hpp) and multiplied; no LDS in the loop.  A wave
// owns ALL NT x IT tiles of dW_in for its 32-point tiles (<= 128 accumulator registers), the workgroup's waves are added
// in LDS at the end of a run.
struct DwinArgs {
  const float* dh1;                          // [n_slots, NN]
  const float* x;
  uint32_t ldx;
  XSeg seg;
  uint32_t n_in, n_in_w;                     // live input columns; row stride of dW_in (>= n_in)
  float* d_params;                           // dW_in block of the group = d_params + group * param_stride
  uint32_t n_slots;
  const int32_t* row_index;
  const int32_t* tile_group;
  uint32_t param_stride, tiles_per_block;
};

template <int NN, int IT, int PREC>
__global__ __launch_bounds__(256, 2) void mlp_dwin_kernel(DwinArgs a) {
  constexpr int NT = NN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* stg_base = reinterpret_cast<float*>(lds);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  int* rows_lds = reinterpret_cast<int*>(stg_base + wave * STG_WAVE_FLOATS + STG_FLOATS);
  const uint32_t n_in = a.n_in;
  const uint32_t n_btiles = (a.n_slots + 127u) / 128u;
  const uint32_t bt0 = blockIdx.x * a.tiles_per_block;
  const uint32_t bt1 = min(bt0 + a.tiles_per_block, n_btiles);
  auto group_of = [&](uint32_t t) -> int { return a.tile_group ? a.tile_group[t] : 0; };
  const uint32_t ch = lane & 31u, pt8 = 8u * (lane >> 5);
  f32x16 acc[NT][IT];
  for (uint32_t run0 = bt0; run0 < bt1;) {       // runs of tiles of one weight set (see mlp_split_bwd.inc)
    const int grp = group_of(run0);
    uint32_t run1 = run0 + 1;
    while (run1 < bt1 && group_of(run1) == grp) ++run1;
    if (grp < 0) {
      run0 = run1;
      continue;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < IT; ++j) acc[i][j] = zero16();
    for (uint32_t bt = run0; bt < run1; ++bt) {
      const uint32_t slot0 = bt * 128u + wave * 32u;
      tile_rows_publish(rows_lds, a.row_index, slot0, a.n_slots, lane);
      Frag3 fa[NT][2];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          // unconditional loads on clamped slots, zeroed afterwards: a guarded load is a branch, and eight branches in a row
          // keep the compiler from issuing the eight loads back to back (each then waits a memory round trip on its own)
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t slot = slot0 + 16u * s + pt8 + j;
            v[j] = a.dh1[(size_t)min(slot, a.n_slots - 1u) * NN + 32 * t + ch];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (slot0 + 16u * s + pt8 + j) < a.n_slots ? v[j] : 0.f;
          fa[t][s] = split8_bf3(v);
        }
#pragma unroll
      for (int ct = 0; ct < IT; ++ct) {
        const uint32_t col_raw = 32u * ct + ch;
        const bool col_ok = col_raw < n_in;
        const uint32_t col = col_ok ? col_raw : n_in - 1u;                  // clamped: the load is unconditional (see above)
        const bool second = a.seg.x2 != nullptr && col >= a.seg.n_in1;
        const float* base = second ? a.seg.x2 + (col - a.seg.n_in1) : a.x + col;
        const uint32_t ld = second ? a.seg.ldx2 : a.ldx;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float v[8];
          int rowv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) rowv[j] = rows_lds[16 * s + pt8 + j];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = base[(size_t)max(rowv[j], 0) * ld];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (rowv[j] >= 0 && col_ok) ? v[j] : 0.f;
          const Frag3 fb = split8_bf3(v);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t][ct] = mma_pt<PREC>(fa[t][s], fb, acc[t][ct]);
        }
      }
    }
    __syncthreads();
    {
      float* dW = a.d_params + (size_t)grp * a.param_stride;
      // Column tiles in an order ROTATED by the workgroup index: every workgroup adds into the same dW_in, and all of them reach
      // this point at about the same time -- in one common order each line takes the adds of all 512 workgroups back to back
      // (tools/flush_atomic_rate.hip: 18.7 us in one order, 13.9 rotated, 10.8 = the chip's atomic throughput).  The accumulators
      // are registers: a rolled loop over the rotation with one case per column tile, not run-time indexing.
#pragma unroll 1
      for (uint32_t k = 0; k < (uint32_t)IT; ++k) {
        const uint32_t jj = (k + blockIdx.x) % (uint32_t)IT;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
          if (jj == (uint32_t)j) {                  // uniform
#pragma unroll
            for (int i = 0; i < NT; ++i)
              flush_tile(acc[i][j], dW + (size_t)(32 * i) * a.n_in_w + 32 * j, a.n_in_w, 32u, n_in > 32u * j ? n_in - 32u * j : 0u, stg_base, wave, lane);
          }
        }
      }
    }
    run0 = run1;
  }
}

template <int NN, int NL>
static bool set_fwd_attrs() {
  return hipFuncSetAttribute((const void*)mlp_fwd_kernel<NN, NL, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) == hipSuccess &&
         hipFuncSetAttribute((const void*)mlp_fwd_kernel<NN, NL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) == hipSuccess &&
         hipFuncSetAttribute((const void*)mlp_fwd_kernel<NN, NL, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) == hipSuccess &&
         hipFuncSetAttribute((const void*)mlp_fwd_kernel<NN, NL, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) == hipSuccess;
}

static int split_init_attrs() {
  const bool ok = set_fwd_attrs<32, 1>() && set_fwd_attrs<32, 2>() && set_fwd_attrs<64, 1>() && set_fwd_attrs<64, 2>();
  if (!ok) {
    set_error("dns_init: hipFuncSetAttribute failed for the split-operand MLP kernels");
    return DNS_E_LAUNCH;
  }
  return DNS_OK;
}
static AttrRegistrar split_attr_registrar(split_init_attrs);

}  // namespace sp

// launch helpers used by dns_mlp_fwd / dns_mlp_bwd (mlp.hip): arguments already validated there
int launch_mlp_fwd_split(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, const float* params,
                         uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, float* y, uint32_t ldy,
                         uint32_t n_slots, const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride,
                         float* h_save, bool fp16_single, const unsigned char* prep, uint32_t prep_stride, const sp::XsIn* xs,
                         hipStream_t st, uint32_t n_in_w) {
  using namespace sp;
  FwdArgs a;
  a.n_in_w = n_in_w ? n_in_w : n_in;
  a.xs = xs ? *xs : XsIn{};
  a.prep = prep; a.prep_stride = prep_stride;
  a.x = x; a.ldx = ldx; a.seg = {x2, ldx2, x2 ? n_in1 : n_in}; a.params = params; a.n_in = n_in; a.n_out = n_out;
  a.y = y; a.ldy = ldy; a.n_slots = n_slots; a.row_index = row_index; a.tile_group = tile_group; a.param_stride = param_stride;
  a.h_save = h_save;
  // y and h_save are addressed through buffer descriptors (mlp_split.hpp): 2 GiB per matrix.  Slot-ordered rows are checked
  // here; rows behind a row table on the device (DNS_DEVERR_MLP_RANGE)
  a.err = device_error_word();
  DNS_REQUIRE((row_index ? 0ull : (uint64_t)n_slots) * ldy * 4ull < (uint64_t)BUF_LIMIT &&
                  (!h_save || 2ull * n_slots * n_neurons * 4ull < (uint64_t)BUF_LIMIT),
              "dns_mlp_fwd: %u slots x %u output floats per row exceed the 2 GiB per matrix this kernel addresses", n_slots, ldy);
  const uint32_t n_btiles = (n_slots + 127u) / 128u;
  uint32_t tpb = (n_btiles + 511u) / 512u;       // ~2 workgroups of 4 waves per CU, contiguous tile ranges
  if (tpb < 1) tpb = 1;
  a.tiles_per_block = tpb;
  const uint32_t blocks = (n_btiles + tpb - 1) / tpb;
#define LAUNCH_FWD(NN, NL)                                                                                           \
  {                                                                                                                  \
    const size_t lds_bytes = FwdLds<NN, NL>::total(n_in, n_out, 4);                                                  \
    if (xs) {                                                                                                        \
      if (fp16_single) DNS_LAUNCH((mlp_fwd_kernel<NN, NL, 1, true>), dim3(blocks), dim3(256), lds_bytes, st, a);     \
      else DNS_LAUNCH((mlp_fwd_kernel<NN, NL, 3, true>), dim3(blocks), dim3(256), lds_bytes, st, a);                 \
    } else if (fp16_single) DNS_LAUNCH((mlp_fwd_kernel<NN, NL, 1>), dim3(blocks), dim3(256), lds_bytes, st, a);      \
    else DNS_LAUNCH((mlp_fwd_kernel<NN, NL, 3>), dim3(blocks), dim3(256), lds_bytes, st, a);                         \
  }
  if (n_neurons == 32 && n_hidden_layers == 1) LAUNCH_FWD(32, 1)
  else if (n_neurons == 32 && n_hidden_layers == 2) LAUNCH_FWD(32, 2)
  else if (n_neurons == 64 && n_hidden_layers == 1) LAUNCH_FWD(64, 1)
  else LAUNCH_FWD(64, 2)
#undef LAUNCH_FWD
  return check_launch("dns_mlp_fwd");
}

int launch_mlp_bwd_split(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, const float* dy,
                         uint32_t lddy, const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                         uint32_t n_hidden_layers, float* d_x, uint32_t lddx, float* d_x2, uint32_t lddx2, float* d_params,
                         float* ws, uint32_t n_slots, const int32_t* row_index, const int32_t* tile_group,
                         uint32_t param_stride, int acc1, int acc2, bool fp16_single, const unsigned char* prep, uint32_t prep_stride,
                         bool with_dwin, const float* h_saved, const sp::XsIn* xs, hipStream_t st, uint32_t n_in_w) {
  using namespace sp;
  BwdArgs a;
  a.n_in_w = n_in_w ? n_in_w : n_in;
  a.xs = xs ? *xs : XsIn{};
  a.h_saved = h_saved;
  a.prep = prep; a.prep_stride = prep_stride;
#ifdef DNS_BWD_TRACE
  a.trace = g_bwd_trace;
#endif
  a.x = x; a.ldx = ldx; a.seg = {x2, ldx2, x2 ? n_in1 : n_in}; a.dy = dy; a.lddy = lddy; a.params = params;
  a.n_in = n_in; a.n_out = n_out; a.dx = d_x; a.lddx = lddx;
  a.dseg = {(d_x && x2) ? d_x2 : nullptr, lddx2, (uint32_t)acc1 & 1u, (uint32_t)acc2, (uint32_t)acc1 >> 8};   // (col_lo rides in acc1's upper bits)
  a.d_params = d_params; a.dh1 = d_params ? ws : nullptr;
  a.n_slots = n_slots; a.row_index = row_index; a.tile_group = tile_group; a.param_stride = param_stride;
  // The kernel addresses dy, d_x, d_x2, dH_1 and the row table through buffer descriptors (mlp_split.hpp): 2 GiB per matrix.
  // Slot-ordered rows are checked here; rows behind a row table are checked on the device (DNS_DEVERR_MLP_RANGE).
  a.err = device_error_word();
  {
    uint64_t ld_max = lddy;
    if (d_x) ld_max = lddx > ld_max ? lddx : ld_max;
    if (d_x && x2 && d_x2) ld_max = lddx2 > ld_max ? lddx2 : ld_max;
    const uint64_t rows = row_index ? 0ull : (uint64_t)n_slots;
    DNS_REQUIRE(rows * ld_max * 4ull < (uint64_t)BUF_LIMIT && (uint64_t)n_slots * n_neurons * 4ull < (uint64_t)BUF_LIMIT &&
                    (uint64_t)n_slots * 4ull < (uint64_t)BUF_LIMIT,
                "dns_mlp_bwd: %u slots x %llu floats per row exceed the 2 GiB per matrix this kernel addresses", n_slots,
                (unsigned long long)ld_max);
  }
  // Without weight gradients (the tracker's frozen scene; a network without a weight-set table) the kernel has no persistent
  // accumulators and runs with EIGHT waves per workgroup -- two per SIMD on one set of LDS weight images: 1.28-1.37x
  const bool wide = !d_params && !tile_group && !h_saved;
  a.n_waves = wide ? 8u : 4u;
  const uint32_t bt_slots = 32u * a.n_waves;
  const uint32_t n_btiles = (n_slots + bt_slots - 1u) / bt_slots;
  uint32_t tpb = (n_btiles + 255u) / 256u;       // one workgroup of 4 waves per CU, contiguous tile ranges
  if (tpb < 1) tpb = 1;
  a.tiles_per_block = tpb;
  const uint32_t blocks = (n_btiles + tpb - 1) / tpb;
  int rc;
  if (n_neurons == 32 && n_hidden_layers == 1) rc = launch_bwd_32_1(a, blocks, fp16_single, st);
  else if (n_neurons == 32 && n_hidden_layers == 2) rc = launch_bwd_32_2(a, blocks, fp16_single, st);
  else if (n_neurons == 64 && n_hidden_layers == 1) rc = launch_bwd_64_1(a, blocks, fp16_single, st);
  else rc = launch_bwd_64_2(a, blocks, fp16_single, st);
  if (rc != DNS_OK || !d_params || !with_dwin) return rc;
  // dW_in = dH_1^T X from the workspace the first kernel wrote
  return launch_mlp_dwin(x, ldx, x2, ldx2, n_in1, n_in, n_neurons, n_hidden_layers, d_params, ws, n_slots, row_index, tile_group,
                         param_stride, fp16_single, st, n_in_w);
}

// the streaming weight-gradient kernel of the first layer on its own (dns_mlp_dwin; the second half of dns_mlp_bwd)
int launch_mlp_dwin(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, uint32_t n_in, uint32_t n_neurons,
                    uint32_t n_hidden_layers, float* d_params, const float* ws, uint32_t n_slots, const int32_t* row_index,
                    const int32_t* tile_group, uint32_t param_stride, bool fp16_single, hipStream_t st, uint32_t n_in_w) {
  using namespace sp;
  (void)n_hidden_layers;
  const uint32_t n_btiles = (n_slots + 127u) / 128u;
  DwinArgs d;
  d.dh1 = ws; d.x = x; d.ldx = ldx; d.seg = {x2, ldx2, x2 ? n_in1 : n_in}; d.n_in = n_in; d.n_in_w = n_in_w ? n_in_w : n_in; d.d_params = d_params; d.n_slots = n_slots;
  d.row_index = row_index; d.tile_group = tile_group; d.param_stride = param_stride;
  // two workgroups per CU (DNS_DWIN_BLOCKS: measurement knob -- every workgroup ends in 6-8 tiles of float atomics onto the SAME
  // dW_in addresses, so fewer workgroups = fewer atomics, more = more loads in flight)
  static const uint32_t dwin_blocks = [] { const char* e = getenv("DNS_DWIN_BLOCKS"); const long n = e ? atol(e) : 0; return (uint32_t)(n >= 64 && n <= 4096 ? n : 512); }();
  uint32_t tpb2 = (n_btiles + dwin_blocks - 1u) / dwin_blocks;
  if (tpb2 < 1) tpb2 = 1;
  d.tiles_per_block = tpb2;
  const uint32_t blocks2 = (n_btiles + tpb2 - 1) / tpb2;
  const size_t lds2 = 4 * STG_WAVE_FLOATS * sizeof(float);
  const uint32_t it = (n_in + 31u) / 32u;
#define LAUNCH_DWIN2(NN, IT)                                                                                          \
  {                                                                                                                   \
    if (fp16_single) DNS_LAUNCH((mlp_dwin_kernel<NN, IT, 1>), dim3(blocks2), dim3(256), lds2, st, d);        \
    else DNS_LAUNCH((mlp_dwin_kernel<NN, IT, 3>), dim3(blocks2), dim3(256), lds2, st, d);                     \
  }
#define LAUNCH_DWIN(NN)                  \
  {                                      \
    if (it <= 1) LAUNCH_DWIN2(NN, 1)     \
    else if (it == 2) LAUNCH_DWIN2(NN, 2) \
    else if (it == 3) LAUNCH_DWIN2(NN, 3) \
    else LAUNCH_DWIN2(NN, 4)             \
  }
  if (n_neurons == 32) LAUNCH_DWIN(32)
  else LAUNCH_DWIN(64)
#undef LAUNCH_DWIN
#undef LAUNCH_DWIN2
  return check_launch("dns_mlp_bwd(dW_in)");
}

// ---- prepared weight images (dns_mlp_prepare): per weight set [forward image area | 16 B exponents | backward image area
// (with the dX image) | 16 B exponents]
static uint32_t prepared_fwd_bytes(uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl) {
  using namespace sp;
  uint32_t b;
  if (nn == 32 && nl == 1) b = FwdLds<32, 1>::misc(n_in, n_out);
  else if (nn == 32) b = FwdLds<32, 2>::misc(n_in, n_out);
  else if (nl == 1) b = FwdLds<64, 1>::misc(n_in, n_out);
  else b = FwdLds<64, 2>::misc(n_in, n_out);
  return b + 16u;
}
static uint32_t prepared_bwd_bytes(uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl) {
  using namespace sp;
  if (nn == 32 && nl == 1) return prepared_bwd_bytes_32_1(n_in, n_out);
  if (nn == 32) return prepared_bwd_bytes_32_2(n_in, n_out);
  if (nl == 1) return prepared_bwd_bytes_64_1(n_in, n_out);
  return prepared_bwd_bytes_64_2(n_in, n_out);
}
uint32_t mlp_prepared_fwd_bytes(uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl) { return prepared_fwd_bytes(n_in, n_out, nn, nl); }
uint32_t mlp_prepared_bytes(uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl) {
  return prepared_fwd_bytes(n_in, n_out, nn, nl) + prepared_bwd_bytes(n_in, n_out, nn, nl);
}
// n_in: the width the images are built for (the LIVE width with DNS_MLP_LIVE_IN); n_in_w: the row stride of W_in in `params`
int launch_mlp_prepare(const float* params, uint32_t param_stride, uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl,
                       uint32_t n_sets, unsigned char* blob, hipStream_t st, uint32_t n_in_w) {
  using namespace sp;
  if (n_in_w == 0) n_in_w = n_in;
  const uint32_t stride = mlp_prepared_bytes(n_in, n_out, nn, nl), fb = prepared_fwd_bytes(n_in, n_out, nn, nl);
  if (nn == 32 && nl == 1) return prepare_32_1(params, param_stride, n_in, n_out, n_sets, blob, fb, stride, st, n_in_w);
  if (nn == 32) return prepare_32_2(params, param_stride, n_in, n_out, n_sets, blob, fb, stride, st, n_in_w);
  if (nl == 1) return prepare_64_1(params, param_stride, n_in, n_out, n_sets, blob, fb, stride, st, n_in_w);
  return prepare_64_2(params, param_stride, n_in, n_out, n_sets, blob, fb, stride, st, n_in_w);
}

}  // namespace dns

#ifdef DNS_BWD_TRACE
extern "C" void dns_debug_bwd_trace(unsigned long long* buf) { dns::sp::g_bwd_trace = buf; }
#endif

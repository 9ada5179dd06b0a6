// Forward kernel, launchers and C-ABI entry points of the half-rows MLP (mlp_half.hpp; the backward kernel: mlp_half_bwd.inc,
// one translation unit per (n_neurons, n_hidden_layers)).  Replaces tcnn.Network{CutlassMLP} in ITS OWN precision (reference
// models/decoder.py:58-64,84-90,94,101-116): f16 activations and weight operands, fp32 accumulation, `.float()` outputs.
#include "mlp_half.hpp"

namespace dns {
namespace hf {

// LDS of the forward kernel: the three weight images (single f16 plane each, sized for the launch's widths) and per wave an
// fp32 staging tile for the output stores, two row tables and the rows' byte offsets in y
constexpr uint32_t HFWD_WAVE_BYTES = (sp::STG_FLOATS + 64u + 32u) * 4u;

template <int NN, int NL>
struct HFwdLds {
  static __host__ __device__ uint32_t it(uint32_t n_in) { return (n_in + 31u) / 32u; }
  static __host__ __device__ uint32_t mt(uint32_t n_out) { return (n_out + 31u) / 32u; }
  static __host__ __device__ uint32_t img_h(uint32_t n_in) { return NN * 32u * it(n_in) * 2u; }
  static __host__ __device__ uint32_t img_out(uint32_t n_in) { return img_h(n_in) + (NL - 1) * NN * NN * 2u; }
  static __host__ __device__ uint32_t waves(uint32_t n_in, uint32_t n_out) { return img_out(n_in) + 32u * mt(n_out) * NN * 2u; }
  static __host__ __device__ uint32_t total(uint32_t n_in, uint32_t n_out) { return waves(n_in, n_out) + 4u * HFWD_WAVE_BYTES; }
};

// One wave = one 32-point tile at a time; its B-operand fragments of the first layer come straight from the f16 rows in
// memory (lane (r, h) loads the 16 bytes at columns 16 s + 8 h of row r for K-step s: no staging, no conversion), the next
// tile's while this one runs its later layers.  y: fp32 rows through the tile's row table (dword stores, any ldy).
template <int NN, int NL>
__global__ __launch_bounds__(256, 2) void mlp_half_fwd_kernel(HFwdArgs a) {
  constexpr int NT = NN / 32;
  using L = HFwdLds<NN, NL>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n_in = a.n_in, n_out = a.n_out;
  const uint32_t ns0 = n_in >> 4, mt = L::mt(n_out);
  const uint32_t n_btiles = (a.n_slots + 127u) / 128u;
  const uint32_t bt0 = blockIdx.x * a.tiles_per_block;
  const uint32_t bt1 = min(bt0 + a.tiles_per_block, n_btiles);
  const WImg w_in{0u, L::it(n_in)}, w_h{L::img_h(n_in), NN / 32u}, w_out{L::img_out(n_in), NN / 32u};
  const uint32_t img_bytes = L::waves(n_in, n_out);
  float* stg = reinterpret_cast<float*>(lds + img_bytes + wave * HFWD_WAVE_BYTES);
  int* rows_all = reinterpret_cast<int*>(stg + sp::STG_FLOATS);
  uint32_t* yoff = reinterpret_cast<uint32_t*>(stg + sp::STG_FLOATS + 64u);
  const rsrc_t r_y = buf_make(a.y);
  const uint32_t y_row_limit = BUF_LIMIT / (4u * max(a.ldy, 1u));
  uint32_t bad_rows = 0;
  LaneAddr A = lane_addr(lane, L::it(n_in), NN / 32u, 0u);          // (only the weight-image parts are used here)
  uint4 xf[8];                                   // the tile's B fragments of the first layer (n_in <= 128: at most 8 K-steps)
  uint32_t cur_buf = 1;
  int cur_group = -2;
  auto request = [&](uint32_t nbt, uint32_t buf) {
    int* nrows_lds = rows_all + 32u * buf;
    sp::tile_rows_publish(nrows_lds, a.row_index, nbt * 128u + wave * 32u, a.n_slots, lane);
    const uint32_t row = (uint32_t)max(nrows_lds[lane & 31u], 0);       // a padding slot reads row 0: nothing of it is stored
    const uint32_t h8 = 8u * (lane >> 5);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const uint32_t col = min(16u * (uint32_t)s + h8, n_in - 8u);       // (K-steps past n_in: loaded from the last piece, never used)
      const bool second = a.seg.x2 != nullptr && col >= a.seg.n_in1;
      const _Float16* p = second ? a.seg.x2 + (size_t)row * a.seg.ldx2 + (col - a.seg.n_in1) : a.x + (size_t)row * a.ldx + col;
      xf[s] = *reinterpret_cast<const uint4*>(p);
    }
  };
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
        const float* pw = a.params + (size_t)grp * a.param_stride;
        const float* wh = pw + NN * a.n_in_w;
        const float* wout = wh + (NL - 1) * NN * NN;
        WQuads<NN * 128 / 1024> q_in;                  // n_in <= 128
        WQuads<NN * NN / 1024> q_h;
        WQuads<64 * NN / 1024> q_out;                  // n_out <= 64
        w_image_load(q_in, pw, NN, n_in, a.n_in_w);
        if constexpr (NL == 2) w_image_load(q_h, wh, NN, NN, NN);
        w_image_load(q_out, wout, n_out, NN, NN);
        lds_zero16(lds, img_bytes);
        __syncthreads();
        w_image_store(q_in, lds, w_in, NN, n_in);
        if constexpr (NL == 2) w_image_store(q_h, lds, w_h, NN, NN);
        w_image_store(q_out, lds, w_out, n_out, NN);
      }
      cur_group = grp;
      __syncthreads();
    }
    if (grp < 0) continue;
    cur_buf ^= 1u;
    lane_addr_pin(A);
    const int* rows_lds = rows_all + 32u * cur_buf;
    if (lane < 32u) {                                // the rows' byte offsets in y (a padding slot: out of range)
      const int row = rows_lds[lane];
      const bool ok = (uint32_t)row < y_row_limit;
      bad_rows |= (row >= 0 && !ok) ? 1u : 0u;
      yoff[lane] = sp::buf_row_off(row, a.ldy, ok);
    }
    sp::wave_lds_fence();
    f32x16 h1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) h1[t] = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if ((uint32_t)s < ns0) {                       // uniform
#pragma unroll
        for (int t = 0; t < NT; ++t) h1[t] = mma(w_row_frag<K_NAT>(lds, A, w_in, t, s), as_half8(xf[s]), h1[t]);
      }
    }
    // the next live tile's rows: requested now, they arrive while this tile runs its later layers
    for (uint32_t nbt = bt + 1; nbt < bt1; ++nbt) {
      if ((a.tile_group ? a.tile_group[nbt] : 0) < 0) continue;
      request(nbt, cur_buf ^ 1u);
      break;
    }
    half8 h1p[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      h1p[t][0] = relu8(pack_half<0>(h1[t]));
      h1p[t][1] = relu8(pack_half<1>(h1[t]));
    }
    half8 h2p[NT][2];
    if constexpr (NL == 2) {
      f32x16 h2[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) h2[t] = zero16();
#pragma unroll
      for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int hf_ = 0; hf_ < 2; ++hf_)
#pragma unroll
          for (int t = 0; t < NT; ++t) h2[t] = mma(w_row_frag<K_CHAIN>(lds, A, w_h, t, 2 * tk + hf_), h1p[tk][hf_], h2[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        h2p[t][0] = relu8(pack_half<0>(h2[t]));
        h2p[t][1] = relu8(pack_half<1>(h2[t]));
      }
    }
    half8(&hlp)[NT][2] = (NL == 2) ? h2p : h1p;
    auto out_tile = [&](uint32_t m) -> f32x16 {
      f32x16 o = zero16();
#pragma unroll
      for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int hf_ = 0; hf_ < 2; ++hf_) o = mma(w_row_frag<K_CHAIN>(lds, A, w_out, m, 2 * tk + hf_), hlp[tk][hf_], o);
      return o;
    };
    sp::store_tile_rows_scalar_buf(r_y, yoff, 0u, min(32u, n_out), out_tile(0u), stg, lane);
    if (mt >= 2u) sp::store_tile_rows_scalar_buf(r_y, yoff, 32u, n_out - 32u, out_tile(1u), stg, lane);
  }
  if (bad_rows != 0u && a.err) atomicOr(a.err, DNS_DEVERR_MLP_RANGE);
}

template <int NN, int NL>
static bool set_fwd_attr() {
  return hipFuncSetAttribute((const void*)mlp_half_fwd_kernel<NN, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) == hipSuccess;
}
static int half_init_attrs() {
  if (!(set_fwd_attr<32, 1>() && set_fwd_attr<32, 2>() && set_fwd_attr<64, 1>() && set_fwd_attr<64, 2>())) {
    set_error("dns_init: hipFuncSetAttribute failed for the half-rows MLP forward kernels");
    return DNS_E_LAUNCH;
  }
  return DNS_OK;
}
static AttrRegistrar half_attr_registrar(half_init_attrs);

}  // namespace hf
}  // namespace dns

using namespace dns;

static bool half_shape_ok(uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl) {
  return (nn == 32 || nn == 64) && (nl == 1 || nl == 2) && n_in >= 16 && n_in <= 128 && (n_in % 16) == 0 && n_out >= 1 && n_out <= 64;
}
// DNS_MLP_LIVE_IN(n): as in dns_mlp_fwd -- the kernels run as an n-input network on the same parameter tensor
static uint32_t half_live_in(uint32_t flags, uint32_t n_in, uint32_t n_in1, bool two) {
  const uint32_t n = (flags >> 16) & 0xffu;
  if (n == 0) return n_in;
  if (n > n_in || (n % 16) != 0 || (two && n <= n_in1)) return 0;
  return n;
}
static int half_check_rows(const char* who, const void* x, uint32_t ldx, const void* x2, uint32_t ldx2, uint32_t n_in1, uint32_t n_live) {
  DNS_REQUIRE((ldx % 8) == 0 && (((uintptr_t)x) % 16) == 0, "%s: x must be 16-byte aligned f16 rows with ldx %% 8 == 0", who);
  if (!x2) {
    DNS_REQUIRE(ldx >= n_live, "%s: ldx < n_in", who);
    return DNS_OK;
  }
  DNS_REQUIRE(n_in1 >= 8 && n_in1 < n_live && (n_in1 % 8) == 0 && ldx >= n_in1, "%s: first input segment must hold a multiple of 8 columns in (0, n_in)", who);
  DNS_REQUIRE((ldx2 % 8) == 0 && (((uintptr_t)x2) % 16) == 0 && ldx2 >= n_live - n_in1,
              "%s: x2 must be 16-byte aligned f16 rows with ldx2 %% 8 == 0 and ldx2 >= n_in - n_in1", who);
  return DNS_OK;
}

extern "C" int dns_mlp_fwd_half(const void* x, uint32_t ldx, const void* x2, uint32_t ldx2, uint32_t n_in1, const float* params,
                                uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, float* y, uint32_t ldy,
                                uint32_t n_slots, const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride,
                                uint32_t flags, void* stream) {
  using namespace hf;
  if (n_slots == 0) return DNS_OK;
  DNS_REQUIRE(x && params && y, "dns_mlp_fwd_half: NULL argument");
  DNS_REQUIRE((flags & ~0xff0000u) == 0, "dns_mlp_fwd_half: unknown flags 0x%x", flags);
  const uint32_t n_live = half_live_in(flags, n_in, n_in1, x2 != nullptr);
  DNS_REQUIRE(n_live != 0, "dns_mlp_fwd_half: bad DNS_MLP_LIVE_IN value (a multiple of 16 in (n_in1, n_in])");
  DNS_REQUIRE(half_shape_ok(n_live, n_out, n_neurons, n_hidden_layers) && n_in <= 128 && (n_in % 4) == 0,
              "dns_mlp_fwd_half: unsupported shape in=%u (live %u) out=%u neurons=%u layers=%u", n_in, n_live, n_out, n_neurons, n_hidden_layers);
  DNS_REQUIRE(ldy >= n_out, "dns_mlp_fwd_half: ldy < n_out");
  {
    const int rc = half_check_rows("dns_mlp_fwd_half", x, ldx, x2, ldx2, n_in1, n_live);
    if (rc != DNS_OK) return rc;
  }
  const int rc = ensure_ready((hipStream_t)stream, "dns_mlp_fwd_half");
  if (rc != DNS_OK) return rc;
  HFwdArgs a;
  a.x = (const _Float16*)x; a.ldx = ldx; a.seg = {(const _Float16*)x2, ldx2, x2 ? n_in1 : n_live};
  a.params = params; a.n_in = n_live; a.n_in_w = n_in; a.n_out = n_out; a.y = y; a.ldy = ldy; a.n_slots = n_slots;
  a.row_index = row_index; a.tile_group = tile_group; a.param_stride = param_stride;
  a.err = device_error_word();
  DNS_REQUIRE((row_index ? 0ull : (uint64_t)n_slots) * ldy * 4ull < (uint64_t)BUF_LIMIT,
              "dns_mlp_fwd_half: %u slots x %u output floats per row exceed the 2 GiB per matrix this kernel addresses", n_slots, ldy);
  const uint32_t n_btiles = (n_slots + 127u) / 128u;
  uint32_t tpb = (n_btiles + 511u) / 512u;       // ~2 workgroups of 4 waves per CU, contiguous tile ranges
  if (tpb < 1) tpb = 1;
  a.tiles_per_block = tpb;
  const uint32_t blocks = (n_btiles + tpb - 1) / tpb;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_HFWD(NN, NL) DNS_LAUNCH((mlp_half_fwd_kernel<NN, NL>), dim3(blocks), dim3(256), (HFwdLds<NN, NL>::total(n_live, n_out)), st, a)
  if (n_neurons == 32 && n_hidden_layers == 1) LAUNCH_HFWD(32, 1);
  else if (n_neurons == 32) LAUNCH_HFWD(32, 2);
  else if (n_hidden_layers == 1) LAUNCH_HFWD(64, 1);
  else LAUNCH_HFWD(64, 2);
#undef LAUNCH_HFWD
  return check_launch("dns_mlp_fwd_half");
}

extern "C" int dns_mlp_bwd_half(const void* x, uint32_t ldx, const void* x2, uint32_t ldx2, uint32_t n_in1, const float* dy,
                                uint32_t lddy, const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                                uint32_t n_hidden_layers, float* d_x, uint32_t lddx, float* d_x2, uint32_t lddx2, float* d_params,
                                uint32_t n_slots, const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride,
                                int accumulate_dx, float loss_scale, void* stream) {
  using namespace hf;
  if (n_slots == 0) return DNS_OK;
  DNS_REQUIRE(x && dy && params, "dns_mlp_bwd_half: NULL argument");
  DNS_REQUIRE((accumulate_dx & ~(3 | (int)DNS_MLP_DX_FIRST | 0x7fff0000)) == 0, "dns_mlp_bwd_half: unknown accumulate_dx bits 0x%x", accumulate_dx);
  DNS_REQUIRE(loss_scale > 0.f && loss_scale < 65536.f, "dns_mlp_bwd_half: loss_scale %g outside (0, 65536)", (double)loss_scale);
  const uint32_t dx_from = ((uint32_t)accumulate_dx >> 24) & 0x7fu;
  const bool dx_first = (accumulate_dx & (int)DNS_MLP_DX_FIRST) != 0;
  const uint32_t n_live = half_live_in((uint32_t)accumulate_dx, n_in, n_in1, x2 != nullptr);
  DNS_REQUIRE(n_live != 0, "dns_mlp_bwd_half: bad DNS_MLP_LIVE_IN value (a multiple of 16 in (n_in1, n_in])");
  DNS_REQUIRE(half_shape_ok(n_live, n_out, n_neurons, n_hidden_layers) && n_in <= 128 && (n_in % 4) == 0,
              "dns_mlp_bwd_half: unsupported shape in=%u (live %u) out=%u neurons=%u layers=%u", n_in, n_live, n_out, n_neurons, n_hidden_layers);
  DNS_REQUIRE(lddy >= n_out, "dns_mlp_bwd_half: lddy < n_out");
  {
    const int rc = half_check_rows("dns_mlp_bwd_half", x, ldx, x2, ldx2, n_in1, n_live);
    if (rc != DNS_OK) return rc;
  }
  DNS_REQUIRE(!dx_first || (x2 && d_x), "dns_mlp_bwd_half: DNS_MLP_DX_FIRST needs a two-segment input and d_x");
  DNS_REQUIRE((dx_from % 4) == 0 && dx_from < n_live && (dx_from == 0 || d_x), "dns_mlp_bwd_half: bad DNS_MLP_DX_FROM column %u", dx_from);
  if (d_x) {
    DNS_REQUIRE((lddx % 4) == 0 && (((uintptr_t)d_x) % 16) == 0 && lddx >= (x2 ? n_in1 : n_live),
                "dns_mlp_bwd_half: d_x must be 16-byte aligned with lddx %% 4 == 0 and cover its segment");
    if (x2 && !dx_first) {
      DNS_REQUIRE(d_x2 && (lddx2 % 4) == 0 && (((uintptr_t)d_x2) % 16) == 0 && lddx2 >= n_live - n_in1 && (n_in1 % 4) == 0,
                  "dns_mlp_bwd_half: d_x2 must be given, 16-byte aligned with lddx2 %% 4 == 0 and cover columns [n_in1, n_in)");
    }
  }
  const int rc = ensure_ready((hipStream_t)stream, "dns_mlp_bwd_half");
  if (rc != DNS_OK) return rc;
  HBwdArgs a;
  a.x = (const _Float16*)x; a.ldx = ldx; a.seg = {(const _Float16*)x2, ldx2, x2 ? n_in1 : n_live};
  a.dy = dy; a.lddy = lddy; a.params = params; a.n_in = n_live; a.n_in_w = n_in; a.n_out = n_out;
  a.dx = d_x; a.lddx = lddx;
  a.dseg = {(d_x && x2 && !dx_first) ? d_x2 : nullptr, lddx2, (uint32_t)accumulate_dx & 1u, ((uint32_t)accumulate_dx >> 1) & 1u, dx_from};
  a.d_params = d_params; a.n_slots = n_slots; a.row_index = row_index; a.tile_group = tile_group; a.param_stride = param_stride;
  a.scale = loss_scale; a.inv_scale = 1.0f / loss_scale;
  a.err = device_error_word();
  {
    uint64_t ld_max = lddy;
    if (d_x) ld_max = lddx > ld_max ? lddx : ld_max;
    if (a.dseg.dx2) ld_max = lddx2 > ld_max ? lddx2 : ld_max;
    const uint64_t rows = row_index ? 0ull : (uint64_t)n_slots;
    DNS_REQUIRE(rows * ld_max * 4ull < (uint64_t)BUF_LIMIT && (uint64_t)n_slots * 4ull < (uint64_t)BUF_LIMIT,
                "dns_mlp_bwd_half: %u slots x %llu floats per row exceed the 2 GiB per matrix this kernel addresses", n_slots,
                (unsigned long long)ld_max);
  }
  const uint32_t n_btiles = (n_slots + 127u) / 128u;
  uint32_t tpb = (n_btiles + 255u) / 256u;       // one workgroup of 4 waves per CU, contiguous tile ranges
  if (tpb < 1) tpb = 1;
  a.tiles_per_block = tpb;
  const uint32_t blocks = (n_btiles + tpb - 1) / tpb;
  hipStream_t st = (hipStream_t)stream;
  if (n_neurons == 32 && n_hidden_layers == 1) return launch_hbwd_32_1(a, blocks, st);
  if (n_neurons == 32) return launch_hbwd_32_2(a, blocks, st);
  if (n_hidden_layers == 1) return launch_hbwd_64_1(a, blocks, st);
  return launch_hbwd_64_2(a, blocks, st);
}

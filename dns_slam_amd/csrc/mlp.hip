// C-ABI entry points of the bias-free ReLU MLP (tcnn.Network{CutlassMLP}: reference models/decoder.py:58-64,84-90,101-116,
// slams/mapping.py:737-743): argument checks and dispatch.  The kernels are the split-operand ones of mlp_split.hip /
// mlp_split_bwd.inc (f16 hi/lo operands on v_mfma_f32_32x32x16_f16, fp32 accumulation; see mlp_split.hpp).  The exact-fp32
// kernels on v_mfma_f32_32x32x2_f32 that this file used to hold were retired with them: three f16 products are 3.3x faster
// than the fp32 instruction and their error is below an fp32 fma chain's (tools/mfma_f16x3_probe.hip).
#include "mlp_split.hpp"

namespace dns {

static bool shape_ok(uint32_t n_in, uint32_t n_out, uint32_t nn, uint32_t nl) {
  return (nn == 32 || nn == 64) && (nl == 1 || nl == 2) && n_in >= 8 && n_in <= 128 && (n_in % 8) == 0 && n_out >= 1 &&
         n_out <= 64;
}

}  // namespace dns

using namespace dns;

extern "C" uint64_t dns_mlp_bwd_ws_floats(uint32_t n_slots, uint32_t n_neurons, uint32_t n_hidden_layers) {
  (void)n_hidden_layers;
  return (uint64_t)n_slots * n_neurons;      // dH_1, slot-major: the operand of the dW_in kernel
}

// DNS_MLP_LIVE_IN(n): input columns [n, n_in) are identically zero -- the kernels run as an n-input network on the SAME parameter
// tensor (W_in keeps its row stride n_in).  Returns the live width (n_in when the field is 0) or 0 on a bad value.
static uint32_t live_in(uint32_t flags, uint32_t n_in, uint32_t n_in1, bool two) {
  const uint32_t n = (flags >> 16) & 0xffu;
  if (n == 0) return n_in;
  if (n > n_in || (n % 8) != 0 || (two && n <= n_in1)) return 0;
  return n;
}

// (any_row_alignment: the FORWARD reads its second segment with 16-byte loads from rows of any 4-byte alignment -- the HSA ABI runs
//  the memory pipeline in unaligned-access mode, the fused tracker kernel reads the latent out of the coarse network's [P, 33] rows
//  this way -- so a caller may pass a column slice of another matrix, e.g. fine[:, 1:] with ldx2 = 33, instead of a packed copy)
static int check_segments(const char* who, const float* x2, uint32_t ldx2, uint32_t n_in1, uint32_t n_in, bool any_row_alignment = false) {
  if (!x2) return DNS_OK;
  DNS_REQUIRE(n_in1 >= 4 && n_in1 < n_in && (n_in1 % 4) == 0, "%s: first input segment must hold a multiple of 4 columns in (0, n_in)", who);
  if (any_row_alignment) {
    DNS_REQUIRE((((uintptr_t)x2) % 4) == 0 && ldx2 >= n_in - n_in1, "%s: x2 must be 4-byte aligned with ldx2 >= n_in - n_in1", who);
    return DNS_OK;
  }
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
  DNS_REQUIRE((flags & ~(DNS_MLP_FP16 | DNS_MLP_PREPARED | 0xff0000u)) == 0, "dns_mlp_fwd: unknown flags 0x%x", flags);
  const bool prepared = (flags & DNS_MLP_PREPARED) != 0;
  DNS_REQUIRE(!prepared || (((uintptr_t)params) % 16) == 0, "dns_mlp_fwd: prepared images must be 16-byte aligned");
  DNS_REQUIRE(!h_save || (((uintptr_t)h_save) % 16) == 0, "dns_mlp_fwd: h_save must be 16-byte aligned");
  const uint32_t n_live = live_in(flags, n_in, n_in1, x2 != nullptr);
  DNS_REQUIRE(n_live != 0, "dns_mlp_fwd: bad DNS_MLP_LIVE_IN value (a multiple of 8 in (n_in1, n_in])");
  DNS_REQUIRE(shape_ok(n_in, n_out, n_neurons, n_hidden_layers), "dns_mlp_fwd: unsupported shape in=%u out=%u neurons=%u layers=%u",
              n_in, n_out, n_neurons, n_hidden_layers);
  DNS_REQUIRE((ldx % 4) == 0 && (((uintptr_t)x) % 16) == 0, "dns_mlp_fwd: x must be 16-byte aligned with ldx %% 4 == 0");
  DNS_REQUIRE(ldy >= n_out, "dns_mlp_fwd: ldy < n_out");
  {
    const int rc = check_segments("dns_mlp_fwd", x2, ldx2, n_in1, n_live, true);
    if (rc != DNS_OK) return rc;
  }
  const int rc = ensure_ready((hipStream_t)stream, "dns_mlp_fwd");
  if (rc != DNS_OK) return rc;
  // DNS_MLP_PREPARED: `params` is what dns_mlp_prepare wrote (the forward images come first in every weight set's block)
  const unsigned char* prep = prepared ? reinterpret_cast<const unsigned char*>(params) : nullptr;
  return launch_mlp_fwd_split(x, ldx, x2, ldx2, n_in1, params, n_live, n_out, n_neurons, n_hidden_layers, y, ldy, n_slots,
                              row_index, tile_group, param_stride, h_save, (flags & DNS_MLP_FP16) != 0, prep,
                              mlp_prepared_bytes(n_live, n_out, n_neurons, n_hidden_layers), nullptr, (hipStream_t)stream, n_in);   // (prepared images: laid out for the LIVE width, dns_mlp_prepare)
}

extern "C" int dns_mlp_bwd(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, const float* dy,
                           uint32_t lddy, const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                           uint32_t n_hidden_layers, float* d_x, uint32_t lddx, float* d_x2, uint32_t lddx2,
                           float* d_params, float* ws, uint32_t n_slots, const int32_t* row_index,
                           const int32_t* tile_group, uint32_t param_stride, const float* h_saved, int accumulate_dx,
                           void* stream) {
  if (n_slots == 0) return DNS_OK;
  DNS_REQUIRE(x && dy && params, "dns_mlp_bwd: NULL argument");
  DNS_REQUIRE((accumulate_dx & ~(3 | (int)DNS_MLP_FP16 | (int)DNS_MLP_PREPARED | (int)DNS_MLP_NO_DWIN | (int)DNS_MLP_DX_FIRST | 0x7fff0000)) == 0,
              "dns_mlp_bwd: unknown accumulate_dx bits 0x%x", accumulate_dx);
  const uint32_t dx_from = ((uint32_t)accumulate_dx >> 24) & 0x7fu;
  DNS_REQUIRE(dx_from % 4 == 0 && dx_from < n_in, "dns_mlp_bwd: DNS_MLP_DX_FROM(%u) must be a multiple of 4 below n_in", dx_from);
  const uint32_t n_live = live_in((uint32_t)accumulate_dx, n_in, n_in1, x2 != nullptr);
  DNS_REQUIRE(n_live != 0 && (n_live == n_in || !h_saved),
              "dns_mlp_bwd: bad DNS_MLP_LIVE_IN value (a multiple of 8 in (n_in1, n_in], not with h_saved)");
  const bool dx_first = (accumulate_dx & (int)DNS_MLP_DX_FIRST) != 0;
  DNS_REQUIRE(!dx_first || x2, "dns_mlp_bwd: DNS_MLP_DX_FIRST needs a two-segment input");
  const bool prepared = (accumulate_dx & (int)DNS_MLP_PREPARED) != 0;
  DNS_REQUIRE(!prepared || (((uintptr_t)params) % 16) == 0, "dns_mlp_bwd: prepared images must be 16-byte aligned");
  DNS_REQUIRE(shape_ok(n_in, n_out, n_neurons, n_hidden_layers), "dns_mlp_bwd: unsupported shape in=%u out=%u neurons=%u layers=%u",
              n_in, n_out, n_neurons, n_hidden_layers);
  DNS_REQUIRE((ldx % 4) == 0 && (((uintptr_t)x) % 16) == 0, "dns_mlp_bwd: x must be 16-byte aligned with ldx %% 4 == 0");
  DNS_REQUIRE(!d_params || (ws && (((uintptr_t)ws) % 16) == 0), "dns_mlp_bwd: d_params needs a 16-byte aligned workspace ws");
  DNS_REQUIRE(!h_saved || (((uintptr_t)h_saved) % 16) == 0, "dns_mlp_bwd: h_saved must be 16-byte aligned");
  {
    const int rc = check_segments("dns_mlp_bwd", x2, ldx2, n_in1, n_live);
    if (rc != DNS_OK) return rc;
  }
  if (!x2) n_in1 = n_live;
  DNS_REQUIRE(!x2 || !d_x || d_x2 || dx_first, "dns_mlp_bwd: d_x2 is required with a two-segment input when d_x is asked for");
  if (d_x) DNS_REQUIRE(lddx >= n_in1 && ((lddx % 4) != 0 || (((uintptr_t)d_x) % 16) == 0), "dns_mlp_bwd: d_x alignment / lddx");
  if (d_x && x2 && !dx_first) DNS_REQUIRE(lddx2 >= n_live - n_in1, "dns_mlp_bwd: lddx2 < live columns of the second segment");
  if (dx_first) d_x2 = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int rc = ensure_ready(st, "dns_mlp_bwd");
  if (rc != DNS_OK) return rc;
  return launch_mlp_bwd_split(x, ldx, x2, ldx2, n_in1, dy, lddy, params, n_live, n_out, n_neurons, n_hidden_layers, d_x, lddx,
                              d_x2, lddx2, d_params, ws, n_slots, row_index, tile_group, param_stride,
                              (accumulate_dx & 1) | (int)(dx_from << 8),
                              (accumulate_dx >> 1) & 1, (accumulate_dx & (int)DNS_MLP_FP16) != 0,
                              prepared ? reinterpret_cast<const unsigned char*>(params) + mlp_prepared_fwd_bytes(n_live, n_out, n_neurons, n_hidden_layers)
                                       : nullptr,
                              mlp_prepared_bytes(n_live, n_out, n_neurons, n_hidden_layers), (accumulate_dx & (int)DNS_MLP_NO_DWIN) == 0, h_saved, nullptr, st,
                              n_in);
}

extern "C" int dns_mlp_dwin(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, uint32_t n_in,
                            uint32_t n_neurons, uint32_t n_hidden_layers, float* d_params, const float* ws, uint32_t n_slots,
                            const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride, uint32_t flags, void* stream) {
  if (n_slots == 0) return DNS_OK;
  DNS_REQUIRE(x && d_params && ws, "dns_mlp_dwin: NULL argument");
  DNS_REQUIRE((flags & ~(DNS_MLP_FP16 | 0xff0000u)) == 0, "dns_mlp_dwin: unknown flags 0x%x", flags);
  const uint32_t n_live = live_in(flags, n_in, n_in1, x2 != nullptr);
  DNS_REQUIRE(n_live != 0, "dns_mlp_dwin: bad DNS_MLP_LIVE_IN value");
  DNS_REQUIRE(shape_ok(n_in, 1, n_neurons, n_hidden_layers), "dns_mlp_dwin: unsupported shape in=%u neurons=%u layers=%u", n_in,
              n_neurons, n_hidden_layers);
  DNS_REQUIRE((ldx % 4) == 0 && (((uintptr_t)x) % 16) == 0 && (((uintptr_t)ws) % 16) == 0, "dns_mlp_dwin: x / ws must be 16-byte aligned with ldx %% 4 == 0");
  {
    const int rc = check_segments("dns_mlp_dwin", x2, ldx2, n_in1, n_live);
    if (rc != DNS_OK) return rc;
  }
  if (!x2) n_in1 = n_live;
  hipStream_t st = (hipStream_t)stream;
  const int rc = ensure_ready(st, "dns_mlp_dwin");
  if (rc != DNS_OK) return rc;
  return launch_mlp_dwin(x, ldx, x2, ldx2, n_in1, n_live, n_neurons, n_hidden_layers, d_params, ws, n_slots, row_index, tile_group,
                         param_stride, (flags & DNS_MLP_FP16) != 0, st, n_in);
}

extern "C" uint64_t dns_mlp_prepared_floats(uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers) {
  if (!shape_ok(n_in, n_out, n_neurons, n_hidden_layers)) return 0;
  return mlp_prepared_bytes(n_in, n_out, n_neurons, n_hidden_layers) / 4u;
}

extern "C" int dns_mlp_prepare(const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers,
                               uint32_t n_sets, uint32_t param_stride, float* prepared, uint32_t flags, void* stream) {
  if (n_sets == 0) return DNS_OK;
  DNS_REQUIRE(params && prepared, "dns_mlp_prepare: NULL argument");
  DNS_REQUIRE((flags & ~0xff0000u) == 0, "dns_mlp_prepare: unknown flags 0x%x (DNS_MLP_LIVE_IN(n) only)", flags);
  const uint32_t n_live = live_in(flags, n_in, 0, false);
  DNS_REQUIRE(n_live != 0, "dns_mlp_prepare: bad DNS_MLP_LIVE_IN value (a multiple of 8 in (0, n_in])");
  DNS_REQUIRE(shape_ok(n_in, n_out, n_neurons, n_hidden_layers), "dns_mlp_prepare: unsupported shape in=%u out=%u neurons=%u layers=%u",
              n_in, n_out, n_neurons, n_hidden_layers);
  DNS_REQUIRE((((uintptr_t)prepared) % 16) == 0, "dns_mlp_prepare: prepared must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int rc = ensure_ready(st, "dns_mlp_prepare");
  if (rc != DNS_OK) return rc;
  // the images are built (and the blob is laid out) for the LIVE width; W_in keeps its row stride n_in in `params`
  return launch_mlp_prepare(params, param_stride, n_live, n_out, n_neurons, n_hidden_layers, n_sets,
                            reinterpret_cast<unsigned char*>(prepared), st, n_in);
}

// ---- split-row input (include/dns_hip.h, DnsSplitRows) ---------------------------------------------------------------
static int make_xs(const char* who, const DnsSplitRows* x, const DnsSplitRows* x2, uint32_t n_in1, uint32_t n_in, bool hi_only_ok,
                   sp::XsIn& out) {
  DNS_REQUIRE(x && x->rows && x->exps, "%s: x / its rows / its exponents are NULL", who);
  DNS_REQUIRE((n_in % 16) == 0, "%s: split-row input needs n_in %% 16 == 0 (got %u)", who, n_in);
  auto seg_ok = [&](const DnsSplitRows* s, uint32_t k) {
    return (s->ld % 8) == 0 && (((uintptr_t)s->rows) % 16) == 0 && (s->lo_off % 8) == 0 && s->ld >= k &&
           (s->lo_off == 0 ? hi_only_ok : (s->lo_off >= k && s->ld >= s->lo_off + k));
  };
  if (x2) {
    DNS_REQUIRE(x2->rows && x2->exps, "%s: x2 rows / exponents are NULL", who);
    DNS_REQUIRE(n_in1 >= 16 && n_in1 < n_in && (n_in1 % 16) == 0, "%s: n_in1 must be a multiple of 16 in (0, n_in)", who);
    DNS_REQUIRE(seg_ok(x, n_in1) && seg_ok(x2, n_in - n_in1), "%s: split rows must be 16-byte aligned, ld %% 8 == 0, planes inside the row "
                "(a missing lo plane needs DNS_MLP_FP16)", who);
  } else {
    DNS_REQUIRE(seg_ok(x, n_in), "%s: split rows must be 16-byte aligned, ld %% 8 == 0, planes inside the row (a missing lo plane "
                "needs DNS_MLP_FP16)", who);
  }
  out.s1 = {reinterpret_cast<const _Float16*>(x->rows), x->exps, x->ld, x->lo_off};
  if (x2) out.s2 = {reinterpret_cast<const _Float16*>(x2->rows), x2->exps, x2->ld, x2->lo_off};
  else out.s2 = {nullptr, nullptr, 0u, 0u};
  out.n_in1 = x2 ? n_in1 : n_in;
  return DNS_OK;
}

extern "C" int dns_mlp_fwd_split(const DnsSplitRows* x, const DnsSplitRows* x2, uint32_t n_in1, const float* params, uint32_t n_in,
                                 uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, float* y, uint32_t ldy,
                                 uint32_t n_slots, const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride,
                                 uint32_t flags, void* stream) {
  if (n_slots == 0) return DNS_OK;
  DNS_REQUIRE(params && y, "dns_mlp_fwd_split: NULL argument");
  DNS_REQUIRE((flags & ~(DNS_MLP_FP16 | DNS_MLP_PREPARED)) == 0, "dns_mlp_fwd_split: unknown flags 0x%x", flags);
  const bool prepared = (flags & DNS_MLP_PREPARED) != 0;
  DNS_REQUIRE(!prepared || (((uintptr_t)params) % 16) == 0, "dns_mlp_fwd_split: prepared images must be 16-byte aligned");
  DNS_REQUIRE(shape_ok(n_in, n_out, n_neurons, n_hidden_layers), "dns_mlp_fwd_split: unsupported shape in=%u out=%u neurons=%u layers=%u",
              n_in, n_out, n_neurons, n_hidden_layers);
  DNS_REQUIRE(ldy >= n_out, "dns_mlp_fwd_split: ldy < n_out");
  sp::XsIn xs;
  {
    const int rc = make_xs("dns_mlp_fwd_split", x, x2, n_in1, n_in, (flags & DNS_MLP_FP16) != 0, xs);
    if (rc != DNS_OK) return rc;
  }
  const int rc = ensure_ready((hipStream_t)stream, "dns_mlp_fwd_split");
  if (rc != DNS_OK) return rc;
  const unsigned char* prep = prepared ? reinterpret_cast<const unsigned char*>(params) : nullptr;
  return launch_mlp_fwd_split(nullptr, 0, nullptr, 0, xs.n_in1, params, n_in, n_out, n_neurons, n_hidden_layers, y, ldy, n_slots,
                              row_index, tile_group, param_stride, nullptr, (flags & DNS_MLP_FP16) != 0, prep,
                              mlp_prepared_bytes(n_in, n_out, n_neurons, n_hidden_layers), &xs, (hipStream_t)stream);
}

extern "C" int dns_mlp_bwd_split(const DnsSplitRows* x, const DnsSplitRows* x2, uint32_t n_in1, const float* dy, uint32_t lddy,
                                 const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers,
                                 float* d_x, uint32_t lddx, float* d_x2, uint32_t lddx2, float* d_params, float* ws, uint32_t n_slots,
                                 const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride, int accumulate_dx,
                                 void* stream) {
  if (n_slots == 0) return DNS_OK;
  DNS_REQUIRE(dy && params, "dns_mlp_bwd_split: NULL argument");
  DNS_REQUIRE((accumulate_dx & ~(3 | (int)DNS_MLP_FP16 | (int)DNS_MLP_PREPARED | (int)DNS_MLP_NO_DWIN)) == 0,
              "dns_mlp_bwd_split: unknown accumulate_dx bits 0x%x", accumulate_dx);
  const bool prepared = (accumulate_dx & (int)DNS_MLP_PREPARED) != 0;
  DNS_REQUIRE(!prepared || (((uintptr_t)params) % 16) == 0, "dns_mlp_bwd_split: prepared images must be 16-byte aligned");
  DNS_REQUIRE(shape_ok(n_in, n_out, n_neurons, n_hidden_layers), "dns_mlp_bwd_split: unsupported shape in=%u out=%u neurons=%u layers=%u",
              n_in, n_out, n_neurons, n_hidden_layers);
  DNS_REQUIRE(!d_params || (ws && (((uintptr_t)ws) % 16) == 0), "dns_mlp_bwd_split: d_params needs a 16-byte aligned workspace ws");
  sp::XsIn xs;
  {
    const int rc = make_xs("dns_mlp_bwd_split", x, x2, n_in1, n_in, (accumulate_dx & (int)DNS_MLP_FP16) != 0, xs);
    if (rc != DNS_OK) return rc;
  }
  n_in1 = xs.n_in1;
  DNS_REQUIRE(!x2 || !d_x || d_x2, "dns_mlp_bwd_split: d_x2 is required with a two-segment input when d_x is asked for");
  if (d_x) DNS_REQUIRE(lddx >= n_in1 && ((lddx % 4) != 0 || (((uintptr_t)d_x) % 16) == 0), "dns_mlp_bwd_split: d_x alignment / lddx");
  if (d_x && x2) DNS_REQUIRE(lddx2 >= n_in - n_in1, "dns_mlp_bwd_split: lddx2 < n_in - n_in1");
  hipStream_t st = (hipStream_t)stream;
  const int rc = ensure_ready(st, "dns_mlp_bwd_split");
  if (rc != DNS_OK) return rc;
  // (x2 of the fp32 form is only consulted for "two segments": hand the launcher a non-NULL token when there are two)
  const float* two = x2 ? reinterpret_cast<const float*>(x2->rows) : nullptr;
  return launch_mlp_bwd_split(nullptr, 0, two, 0, n_in1, dy, lddy, params, n_in, n_out, n_neurons, n_hidden_layers, d_x, lddx,
                              d_x2, lddx2, d_params, ws, n_slots, row_index, tile_group, param_stride, accumulate_dx & 1,
                              (accumulate_dx >> 1) & 1, (accumulate_dx & (int)DNS_MLP_FP16) != 0,
                              prepared ? reinterpret_cast<const unsigned char*>(params) + mlp_prepared_fwd_bytes(n_in, n_out, n_neurons, n_hidden_layers)
                                       : nullptr,
                              mlp_prepared_bytes(n_in, n_out, n_neurons, n_hidden_layers), false, nullptr, &xs, st);
}

// C-ABI entry points of the fused tracker iteration (track_fused.inc): argument checks, workspace carving, dispatch on the
// networks' (n_neurons, n_hidden_layers).  Replaces the per-iteration body of Tracker.run (reference slams/tracking.py:313-340).
#include <stdlib.h>
#include "mlp_split.hpp"
#include "track_fused.hpp"

namespace dns {

struct TrackWs {
  size_t buf, x3, dydx, lat, feat, raw, logit, d_col, d_logit, d_featx, d_buf, z, part, total;
};
static TrackWs track_ws(uint32_t N, uint32_t S, uint32_t ld, uint32_t L, uint32_t hid, uint32_t nf, uint32_t C, uint32_t rpw) {
  const size_t P = (size_t)N * S;
  auto al = [](size_t v) { return (v + 63u) & ~(size_t)63u; };           // 256-byte aligned blocks
  TrackWs w;
  size_t o = 0;
  w.buf = o; o = al(o + P * ld);
  w.x3 = o; o = al(o + P * 3);
  w.dydx = o; o = al(o + (size_t)L * 3 * P * 2);
  w.lat = o; o = al(o + P * (hid + 1));
  w.feat = o; o = al(o + P * nf);
  w.raw = o; o = al(o + P * 4);
  w.logit = o; o = al(o + P * (C ? C : 1));
  w.d_col = o; o = al(o + P * 4);
  w.d_logit = o; o = al(o + P * (C ? C : 1));
  w.d_featx = o; o = al(o + P * (4 + nf));
  w.d_buf = o; o = al(o + P * ld);
  w.z = o; o = al(o + P);
  w.part = o; o = al(o + (size_t)((N + rpw - 1) / rpw) * 16);
  w.total = o;
  return w;
}

static uint32_t rays_per_wg(uint32_t S) { return S <= 32u ? 4u : 2u; }

}  // namespace dns

using namespace dns;

static int track_shape(const char* who, const DnsTrackFused* a, uint32_t& ld, uint32_t& mtl) {
  DNS_REQUIRE(a && a->meta, "%s: NULL argument", who);
  const uint32_t S = a->n_uniform + a->n_surface;
  DNS_REQUIRE(a->n_rays >= 1 && S >= 1 && S <= 64, "%s: the fused iteration handles 1..64 samples per ray (got %u)", who, S);
  const uint32_t pe = 3u * a->n_bins, L = a->meta->n_levels;
  ld = pe + 2u * L;
  DNS_REQUIRE(a->meta->n_features == 2 && a->n_bins >= 1 && a->n_bins <= 64 && (pe % 4u) == 0 && (L % 2u) == 0 && L <= DNS_MAX_LEVELS,
              "%s: encoding outside the fused form (n_bins %u, levels %u)", who, a->n_bins, L);
  DNS_REQUIRE(ld > 64u && ld <= 96u && (ld % 8u) == 0, "%s: the coarse network's input width %u is outside (64, 96]", who, ld);
  DNS_REQUIRE(a->hidden >= 32u && a->hidden <= 63u && (a->hidden % 4u) == 0, "%s: latent width %u outside [32, 63]", who, a->hidden);
  DNS_REQUIRE((a->n_feat % 4u) == 0 && a->n_feat >= a->hidden && pe + a->n_feat > 64u && pe + a->n_feat <= 128u && ((pe + a->n_feat) % 8u) == 0,
              "%s: colour / logit input width %u outside (64, 128]", who, pe + a->n_feat);
  DNS_REQUIRE(a->code || a->n_feat == a->hidden || a->n_feat > a->hidden, "%s: n_feat", who);
  DNS_REQUIRE(a->n_class >= 1 && a->n_class <= 64, "%s: n_class %u outside [1, 64]", who, a->n_class);
  DNS_REQUIRE(!a->code || (a->code_dim % 4u == 0 && a->hidden + a->code_dim <= a->n_feat && (((uintptr_t)a->code) & 15u) == 0),
              "%s: code width %u does not fit the feature block / is not 16-byte aligned", who, a->code_dim);
  DNS_REQUIRE((a->n_neurons == 64 && a->n_hidden_layers == 2) || (a->n_neurons == 32 && a->n_hidden_layers == 1),
              "%s: networks of %u x %u (the fused iteration is built for 64 x 2 and 32 x 1)", who, a->n_neurons, a->n_hidden_layers);
  mtl = a->n_class <= 32u ? 1u : 2u;
  return DNS_OK;
}

extern "C" uint64_t dns_track_fused_ws_floats(const DnsTrackFused* a) {
  uint32_t ld, mtl;
  if (track_shape("dns_track_fused_ws_floats", a, ld, mtl) != DNS_OK) return 0;
  const uint32_t S = a->n_uniform + a->n_surface;
  return track_ws(a->n_rays, S, ld, a->meta->n_levels, a->hidden, a->n_feat, a->n_class, rays_per_wg(S)).total;
}

extern "C" int dns_track_fused_begin(const int64_t* pix_all, uint32_t n_rays, uint32_t n_iters, const float* depth, int W, int H0, int W0,
                                     int W1, float* t_surf_all, uint32_t n_surface, uint32_t* dmax_all, void* stream) {
  if (n_iters == 0) return DNS_OK;
  DNS_REQUIRE(pix_all && depth && t_surf_all && dmax_all, "dns_track_fused_begin: NULL argument");
  DNS_REQUIRE(n_rays >= 1 && n_surface >= 1 && W1 > W0, "dns_track_fused_begin: empty draw / window");
  hipStream_t st = (hipStream_t)stream;
  const int rc = ensure_ready(st, "dns_track_fused_begin");
  if (rc != DNS_OK) return rc;
  return tf::launch_track_begin(pix_all, n_rays, n_iters, depth, W, H0, W0, W1 - W0, t_surf_all, n_surface, n_surface / 2u + 1u, dmax_all, st);
}

extern "C" int dns_track_fused_iter(const DnsTrackFused* a, void* stream) {
  uint32_t ld, mtl;
  {
    const int rc = track_shape("dns_track_fused_iter", a, ld, mtl);
    if (rc != DNS_OK) return rc;
  }
  DNS_REQUIRE(a->color && a->depth && a->label && a->cam && a->bound && a->pix && a->t_surf && a->t_zero && a->dmax && a->iter && a->quat && a->trans &&
                  a->table && a->w_coarse && a->w_color && a->w_logit && a->ws && a->adam_m && a->adam_v && a->adam_state && a->best_loss &&
                  a->best_cam && a->out && a->g_quat && a->g_trans,
              "dns_track_fused_iter: NULL argument");
  DNS_REQUIRE(a->n_uniform == 0 || a->t_uniform, "dns_track_fused_iter: t_uniform is NULL");
  DNS_REQUIRE(0 <= a->H0 && a->H0 < a->H1 && a->H1 <= a->H && 0 <= a->W0 && a->W0 < a->W1 && a->W1 <= a->W, "dns_track_fused_iter: bad window");
  DNS_REQUIRE((((uintptr_t)a->ws) & 255u) == 0 && (((uintptr_t)a->w_coarse | (uintptr_t)a->w_color | (uintptr_t)a->w_logit) & 15u) == 0,
              "dns_track_fused_iter: ws must be 256-byte aligned, the prepared images 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int rc = ensure_ready(st, "dns_track_fused_iter");
  if (rc != DNS_OK) return rc;
  const uint32_t S = a->n_uniform + a->n_surface, N = a->n_rays, pe = 3u * a->n_bins;
  const uint32_t rpw = rays_per_wg(S), n_wg = (N + rpw - 1u) / rpw;
  const TrackWs w = track_ws(N, S, ld, a->meta->n_levels, a->hidden, a->n_feat, a->n_class, rpw);
  const bool big = a->n_neurons == 64;
  const uint32_t mlp_lds = big ? tf::track_fused_mlp_lds_64_2(pe + a->n_feat, mtl) : tf::track_fused_mlp_lds_32_1(pe + a->n_feat, mtl);
  DNS_REQUIRE(mlp_lds != 0 && (size_t)mlp_lds + tf::TF_KEEP_BYTES <= (size_t)MAX_DYN_LDS, "dns_track_fused_iter: %u B of LDS for the network phases",
              mlp_lds);
  tf::TrackArgs t = {};
  t.color = a->color; t.depth = a->depth; t.label = a->label;
  t.H = a->H; t.W = a->W; t.H0 = a->H0; t.W0 = a->W0; t.wwin = a->W1 - a->W0;
  t.cam.fx = (float)a->cam[0]; t.cam.fy = (float)a->cam[1]; t.cam.cx = (float)a->cam[2]; t.cam.cy = (float)a->cam[3];
  for (int k = 0; k < 6; ++k) t.bd.b[k] = a->bound[k];
  for (int k = 0; k < 3; ++k) {
    t.b6.b0[k] = a->bound[2 * k];
    t.b6.b1[k] = a->bound[2 * k + 1];
    t.b6.inv_unused[k] = 0.0;
  }
  t.pix = a->pix; t.t_uniform = a->t_uniform; t.t_surf = a->t_surf; t.t_zero = a->t_zero; t.dmax = a->dmax;
  t.iter = a->iter; t.n_iters = a->n_iters ? a->n_iters : 1u;
  t.nu = (int)a->n_uniform; t.ns = (int)a->n_surface; t.N = N; t.S = S; t.rpw = rpw;
  t.quat = a->quat; t.trans = a->trans;
  t.table = reinterpret_cast<const float2*>(a->table);
  t.lv = to_levels(a->meta);
  t.n_bins = a->n_bins;
  t.w_coarse = reinterpret_cast<const unsigned char*>(a->w_coarse);
  t.w_color = reinterpret_cast<const unsigned char*>(a->w_color);
  t.w_logit = reinterpret_cast<const unsigned char*>(a->w_logit);
  t.fwdb_coarse = mlp_prepared_fwd_bytes(ld, a->hidden + 1u, a->n_neurons, a->n_hidden_layers);
  t.fwdb_color = mlp_prepared_fwd_bytes(pe + a->n_feat, 3u, a->n_neurons, a->n_hidden_layers);
  t.fwdb_logit = mlp_prepared_fwd_bytes(pe + a->n_feat, a->n_class, a->n_neurons, a->n_hidden_layers);
  t.hid = a->hidden; t.nf = a->n_feat; t.C = a->n_class;
  t.code = a->code; t.code_dim = a->code ? a->code_dim : 0u;
  t.lambda_p = a->lambda_p; t.lambda_d = a->lambda_d; t.lambda_l = a->lambda_l;
  float* ws = a->ws;
  t.buf = ws + w.buf; t.x3 = ws + w.x3; t.dydx = reinterpret_cast<float2*>(ws + w.dydx); t.lat = ws + w.lat; t.feat = ws + w.feat;
  t.raw = ws + w.raw; t.logit = ws + w.logit; t.d_col = ws + w.d_col; t.d_logit = ws + w.d_logit; t.d_featx = ws + w.d_featx;
  t.d_buf = ws + w.d_buf; t.z = ws + w.z; t.part = ws + w.part;
  t.keep_off = mlp_lds;
  {
    static const uint32_t dbg = [] { const char* e = getenv("DNS_TF_PHASES"); return (uint32_t)(e ? atoi(e) : 0); }();
    t.n_phases = dbg;
  }
  t.err = device_error_word();
  tf::PoseArgs p = {};
  p.part = t.part; p.n_wg = n_wg;
  p.lambda_p = a->lambda_p; p.lambda_d = a->lambda_d; p.lambda_l = a->lambda_l;
  p.quat = a->quat; p.trans = a->trans; p.m = a->adam_m; p.v = a->adam_v; p.state = a->adam_state;
  p.lr_q = a->lr_quat; p.lr_t = a->lr_trans; p.beta1 = a->beta1; p.beta2 = a->beta2; p.eps = a->eps;
  p.iter = a->iter;
  p.best_loss = a->best_loss; p.best_cam = a->best_cam; p.out = a->out; p.g_quat = a->g_quat; p.g_trans = a->g_trans;
  return big ? tf::launch_track_fused_64_2(t, p, n_wg, mtl, st) : tf::launch_track_fused_32_1(t, p, n_wg, mtl, st);
}

// Argument blocks of the fused tracker iteration (track_fused.inc): shared by its per-(n_neurons, n_hidden_layers) translation
// units and the C-ABI wrapper (track.hip).
#pragma once
#include "common.hpp"
#include "dev_sample.hpp"
#include "dev_encode.hpp"

namespace dns {
namespace tf {

struct TrackArgs {
  // the frame and the camera
  const float* color;                        // [H, W, 3]
  const float* depth;                        // [H, W]
  const float* label;                        // [H, W]
  int H, W, H0, W0, wwin;
  Cam cam;
  BoundD bd;                                 // scene bound (fp64): box clip
  Bound6 b6;                                 // the same, for the normalisation
  // this iteration's draws
  const int64_t* pix;                        // [N]
  const float* t_uniform;                    // [nu]
  const float* t_surf;                       // [ns] (mid sample already forced)
  const float* t_zero;                       // [ns]
  const uint32_t* dmax;                      // bits of max(gt_depth) over the draw
  const uint32_t* iter;                      // device counter: which of the frame's n_iters draws this launch uses (pix, t_surf,
  uint32_t n_iters;                          // t_zero, dmax are the bases of [n_iters, .] arrays); the pose kernel advances it
  int nu, ns;
  uint32_t N, S, rpw;                        // rays, samples per ray, rays per workgroup (2 or 4)
  const float* quat;                         // [4]
  const float* trans;                        // [3]
  // the frozen scene
  const float2* table;
  GridLevels lv;
  uint32_t n_bins;
  const unsigned char* w_coarse;             // prepared operand images (dns_mlp_prepare): forward part, then backward part
  const unsigned char* w_color;
  const unsigned char* w_logit;
  uint32_t fwdb_coarse, fwdb_color, fwdb_logit;   // bytes of the forward part of each blob
  uint32_t hid, nf, C;                       // latent width (32), feature block width (64), classes
  const float* code;                         // [N * S, code_dim] 2-D code of every sample, or NULL (zeros)
  uint32_t code_dim;
  float lambda_p, lambda_d, lambda_l;
  // workgroup-private rows in global scratch
  float* buf;                                // [P, ld]  OneBlob | grid
  float* x3;                                 // [P, 3]
  float2* dydx;                              // [L][3][P]
  float* lat;                                // [P, hid + 1]
  float* feat;                               // [P, nf]
  float* raw;                                // [P, 4]
  float* logit;                              // [P, C]
  float* d_col;                              // [P, 4]
  float* d_logit;                            // [P, C]
  float* d_featx;                            // [P, 4 + nf]: column 3 = d occupancy, columns 4.. = d feature block
  float* d_buf;                              // [P, ld]
  float* z;                                  // [N, S]
  float* part;                               // [workgroups][16]: dL/dR (9) | dL/dT (3) | sum p, d, l | n_valid
  uint32_t n_phases;                         // (tools: DNS_TF_PHASES=k stops the kernel behind phase k; 0 = all)
  uint32_t keep_off;                         // byte offset of the workgroup's persistent LDS region (behind the MLP phases' LDS)
  uint32_t* err;
};

constexpr uint32_t TF_RAY_FLOATS = 224;      // per ray: z[64] w[64] dsem[64] | o d dir gtc (12) gd valid label pad | loss[4] pose[12]
constexpr uint32_t TF_DX_FLOATS = 4 * 64 * 3;
constexpr uint32_t TF_KEEP_BYTES = (4 * TF_RAY_FLOATS + TF_DX_FLOATS) * 4;

struct PoseArgs {
  const float* part;                         // [n_wg][16]
  uint32_t n_wg;
  float lambda_p, lambda_d, lambda_l;
  float* quat;
  float* trans;
  float* m;                                  // Adam first moments  [8]: quat at 0, trans at 4
  float* v;                                  // second moments
  float* state;                              // [3]: step count, 1 - beta1^t, 1 - beta2^t
  float lr_q, lr_t, beta1, beta2, eps;
  float* best_loss;
  float* best_cam;                           // [7]
  float* out;                                // [8]: terms p, d, l, total, n_valid, grad quat(4)... (diagnostics: out[0..4])
  float* g_quat;                             // [4] the gradient, for callers that want it
  float* g_trans;                            // [3]
  uint32_t* iter;                            // advanced by one at the end (the next launch takes the next draws)
};

// per-(n_neurons, n_hidden_layers) launchers (track_fused_*.hip): the iteration kernel and the pose kernel, on one stream
int launch_track_fused_64_2(const TrackArgs& a, const PoseArgs& p, uint32_t n_wg, uint32_t mtl, hipStream_t st);
int launch_track_fused_32_1(const TrackArgs& a, const PoseArgs& p, uint32_t n_wg, uint32_t mtl, hipStream_t st);
// LDS bytes of the MLP phases of one workgroup (the persistent region follows), 0 = unsupported shape
uint32_t track_fused_mlp_lds_64_2(uint32_t n_in_out, uint32_t mtl);
uint32_t track_fused_mlp_lds_32_1(uint32_t n_in_out, uint32_t mtl);
int launch_track_begin(const int64_t* pix, uint32_t N, uint32_t n_iters, const float* depth, int W, int H0, int W0, int wwin,
                       float* t_surf, uint32_t ns, uint32_t half_idx, uint32_t* dmax, hipStream_t st);

}  // namespace tf
}  // namespace dns

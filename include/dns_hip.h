/*
 * dns_hip.h -- C ABI of libdns_hip.so, the MI355X (gfx950) implementation of the
 * DNS-SLAM volumetric-rendering hot path.
 *
 * Drop-in boundary.  The reference is pure Python; the only native
 * code on its path is the third-party CUDA package `tinycudann`, reached through
 *   tcnn.Encoding(...)  reference models/pos_encoding.py:16,34,50,63,76,88
 *   tcnn.Network(...)   reference models/decoder.py:58,84,101,110, slams/mapping.py:737
 * plus stock torch ops in utils/common.py.  Each entry point below names the reference
 * interface it replaces.  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked [host];
 *   - the caller owns all buffers (the library never allocates or frees device memory);
 *   - every launch goes to the caller's stream `stream` (a hipStream_t passed as void*);
 *     no entry point synchronises the device;
 *   - gradient outputs marked (+=) are ACCUMULATED into caller-initialised buffers;
 *   - return 0 on success, a negative DNS_E_* code on failure; dns_last_error() then holds
 *     a thread-local message.  No C++ exception crosses this boundary.
 *   - row-major tensors; `ld*` = leading dimension in elements.
 */
#ifndef DNS_HIP_H
#define DNS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNS_ABI_VERSION 12
#define DNS_MAX_LEVELS 32

#define DNS_OK 0
#define DNS_E_ARG (-1)      /* bad argument / unsupported shape */
#define DNS_E_LAUNCH (-2)   /* HIP launch error */
#define DNS_E_STATE (-3)    /* call refused in the current state (first use on a device inside a stream capture) */

/* Multi-resolution hash-grid level table (tcnn GridEncoding constructor; reference call site
 * models/pos_encoding.py:31-46).  Built once on the host by dns_grid_meta_init and passed by
 * pointer [host] to every grid entry point. */
typedef struct DnsGridMeta {
  uint32_t n_levels;
  uint32_t n_features;            /* features per level; this build supports 2 */
  uint32_t log2_hashmap_size;
  uint32_t base_resolution;
  uint32_t total_rows;            /* sum of size[] */
  float per_level_scale;
  float scale[DNS_MAX_LEVELS];    /* exp2f(l*log2f(pls))*base - 1 */
  uint32_t resolution[DNS_MAX_LEVELS];
  uint32_t size[DNS_MAX_LEVELS];  /* rows in the level */
  uint32_t offset[DNS_MAX_LEVELS];/* first row of the level */
  uint32_t hashed[DNS_MAX_LEVELS];/* 1 = coherent-prime hash, 0 = dense index */
} DnsGridMeta;

int dns_abi_version(void);
const char* dns_last_error(void);

/* One-time set-up of the CURRENT device: raises the dynamic-LDS limit of every kernel that stages more than 64 KB
 * (hipFuncSetAttribute -- a context-level call that is illegal while a stream is being captured and pointless per launch).
 * Idempotent and thread-safe.  Entry points call it implicitly on their first use on a device; inside a stream capture
 * that first use is refused with DNS_E_STATE instead, so a caller that captures hipGraphs (the reference's loops
 * slams/mapping.py:881-910, slams/tracking.py:313-340 replayed from a graph) calls dns_init() once after selecting the
 * device. */
int dns_init(void);

/* Measurement aid (bench.py's roofline): dns_kernel_timing(1) clears the record and brackets every kernel the library
 * launches from then on with an event pair on its launch stream (never inside a stream capture); dns_kernel_timing(0)
 * stops.  dns_kernel_timing_count() = spans recorded so far; dns_kernel_timing_get(i, ...) waits for span i and returns
 * the kernel's name and its duration in milliseconds. */
int dns_kernel_timing(int enable);
int dns_kernel_timing_count(void);
int dns_kernel_timing_get(int i, char* name, int name_cap, float* ms);

/* [host] fills *meta.  per_level_scale is the float64 value reference pos_encoding.py:33 computes. */
int dns_grid_meta_init(DnsGridMeta* meta, uint32_t n_levels, uint32_t n_features,
                       uint32_t log2_hashmap_size, uint32_t base_resolution, double per_level_scale);

/* ---- ray generation + depth-guided sampling ------------------------------------------------
 * Replaces, for K stacked target frames of n_per_frame rays each:
 *   get_rotation_from_quad  utils/common.py:447 (quat -> R, in-kernel)
 *   get_sample_uv/select_uv utils/common.py:266-293 (pixel gather; indices are an INPUT)
 *   get_rays_from_uv        utils/common.py:248-264
 *   box far clip            slams/mapping.py:519-527, slams/tracking.py:148-156 (fp64)
 *   sample_along_rays       utils/common.py:561-599 (per-frame max depth, per-ray sort)
 * pix_idx: flat index inside the window [H0,H1)x[W0,W1).  color [K,H,W,3], depth [K,H,W],
 * label [K,H,W] fp32.  quat [K,4] (w,x,y,z), trans [K,3].  cam = fx,fy,cx,cy [host].
 * bound [host] = 6 doubles b0x,b1x,b0y,b1y,b0z,b1z.  t_uniform [n_uniform] = linspace(0,1),
 * t_surf / t_zero the two jitter draws (t_surf already holds the forced 0.5): jitter_stride = 0 -> [n_surface], one
 * pair shared by the K frames; jitter_stride >= n_surface -> [K, jitter_stride], frame f reads its own row (the reference
 * calls sample_along_rays once per frame with fresh draws, slams/mapping.py:530).
 * depth_max_ws: K uint32 (bit patterns of the per-frame max sampled depth): scratch when depth_max_given == 0; when
 * depth_max_given != 0 the caller has filled it (multi-GPU: the max over ALL ranks' rays of the frame).  Outputs: rays_o,rays_d,gt_color [n,3], gt_depth [n],
 * gt_label [n] int64, inside [n] uint8 (far_bb >= depth, before the +0.01), z [n, n_uniform+n_surface]
 * ascending. */
int dns_raygen_sample(const int64_t* pix_idx, const float* color, const float* depth, const float* label,
                      const float* quat, const float* trans, const double* cam, const double* bound,
                      int H, int W, int H0, int H1, int W0, int W1, int n_frames, int n_per_frame,
                      const float* t_uniform, const float* t_surf, const float* t_zero,
                      int n_uniform, int n_surface, int jitter_stride, uint32_t* depth_max_ws, int depth_max_given,
                      float* rays_o, float* rays_d, float* gt_color, float* gt_depth, int64_t* gt_label,
                      uint8_t* inside, float* z, float* pts /* [n,S,3] = o + d*z (slams/mapping.py:531), NULL = skip */,
                      void* stream);

/* The free functions get_samples / get_samples_by_class / get_samples_by_uniq_class (utils/common.py:296-304,353-403:
 * the gather of select_uv / select_by_class :266-338 + get_rays_from_uv :248-264) and get_all_rays (:540-559), which take
 * the rotation as a MATRIX: n rays from the window [H0,H1)x[W0,W1) of ONE image.  pix_idx [n] window-flat indices (NULL:
 * ray r = pixel r, the whole-window order of get_all_rays); image [H,W,C] fp32 (C <= 8: rgb | depth | label; NULL with
 * sample NULL); R [9] row-major, T [3] on the DEVICE; cam [host] fx,fy,cx,cy.  Outputs rays_o, rays_d [n,3], sample [n,C]
 * (NULL = skip), ij [n,2] = (column, row) as fp32 like the reference's i, j (NULL = skip). */
int dns_rays_from_pixels(const int64_t* pix_idx, const float* image, int C, const float* R, const float* T,
                         const double* cam, int H, int W, int H0, int H1, int W0, int W1, int n,
                         float* rays_o, float* rays_d, float* sample, float* ij, void* stream);

/* Stand-alone sample_along_rays(gt_depth, n_samples, n_surface, far_bb, device) (utils/common.py:561-599):
 * gt_depth [n] fp32, far_bb [n] fp64 (already carrying the +0.01), batch-global max depth taken over the n
 * rays.  depth_max_ws: 1 uint32 scratch.  z [n, n_uniform+n_surface] ascending. */
int dns_sample_along_rays(const float* gt_depth, const double* far_bb, int n_rays, const float* t_uniform,
                          const float* t_surf, const float* t_zero, int n_uniform, int n_surface,
                          uint32_t* depth_max_ws, float* z, void* stream);

/* Backward of the ray generation w.r.t. the pose (autograd of pts = o + d*z, get_rays_from_uv and
 * quad2rotation, utils/common.py:248-264,406-429): d_quat [K,4] (+=), d_trans [K,3] (+=) from d_pts [n,S,3]
 * and/or direct d_rays_o / d_rays_d [n,3] (each may be NULL).  ws: dns_raygen_bwd_ws_floats(K, n_per_frame) floats of scratch
 * (one [12] partial sum per workgroup: no atomics, nothing to clear). */
int dns_raygen_bwd(const int64_t* pix_idx, const float* quat, const double* cam,
                   int H0, int H1, int W0, int W1, int n_frames, int n_per_frame, int S,
                   const float* z, const float* d_pts, const float* d_rays_o, const float* d_rays_d,
                   float* ws, float* d_quat, float* d_trans, void* stream);
uint64_t dns_raygen_bwd_ws_floats(int n_frames, int n_per_frame);

/* ---- point encoding: OneBlob + hash grid ---------------------------------------------------
 * Replaces Pos_Encoding.forward (models/decoder.py:45-48) = tcnn OneBlob (pos_encoding.py:61-71)
 * + tcnn HashGrid (pos_encoding.py:31-46), optionally preceded by the fp64 normalisation
 * (pts - b0)/(b1 - b0) of slams/mapping.py:608 / slams/tracking.py:190 when bound != NULL.
 * in [P,3]; if bound != NULL [host, 6 doubles] `in` holds world points and x_out [P,3] receives the
 * normalised fp32 coordinates (may be NULL).  pe_out [P, ld_pe] gets 3*n_bins channels (NULL = skip),
 * grid_out [P, ld_grid] gets n_levels*n_features channels (NULL = skip).  table [total_rows, F].
 * dy_dx (NULL = skip): [n_levels, 3, P, 2] floats, 8-byte aligned -- d(grid features)/d(normalised coordinate), what tcnn's
 * kernel_grid keeps when the input needs a gradient; handed to dns_encode_bwd it replaces the second gather of
 * the 8 corners per level by a streaming dot product. */
int dns_encode_fwd(const float* in, const double* bound, uint32_t P, uint32_t n_bins,
                   const float* table, const DnsGridMeta* meta,
                   float* x_out, float* pe_out, uint32_t ld_pe, float* grid_out, uint32_t ld_grid,
                   float* dy_dx, void* stream);

/* ---- split rows: the activation format between the encoders and the MLP kernels (ABI v9) ---------------------------
 * Replaces, on the fixed launch sequence of dns_slam_amd/fused_step.py, the fp32 [P, K] rows that tcnn hands from its encodings
 * to its networks (models/decoder.py:45-48 -> :93-94, :123-124).  A row of K values travels as K halfs hi[j] = f16(v[j] 2^e)
 * followed by K halfs lo[j] = f16(v[j] 2^e - hi[j]) plus ONE int32 exponent e per row (e puts max |row| into [2^13, 2^14)):
 * exactly the operand parts the MLP kernels otherwise derive from the fp32 row in every launch that reads it (row maximum,
 * scale, two conversions per value, an LDS transpose) -- the rows of a mapping iteration are read by ~15 launches.  The
 * bytes per row are those of the fp32 row; DNS_SPLIT_HI_ONLY (half-width mode, with DNS_MLP_FP16 consumers: tcnn's own fp16
 * activations, models/decoder.py:58-64,94) writes the hi plane only: half the bytes.
 *   rows: 16-byte aligned, ld halfs per row (ld % 8 == 0); hi plane at columns [0, K), lo plane at [lo_off, lo_off + K)
 *   (lo_off = 0: no lo plane); exps [rows] int32.  The kernels that take the format need K % 16 == 0. */
typedef struct DnsSplitRows {
  const void* rows;
  const int32_t* exps;
  uint32_t ld;
  uint32_t lo_off;
} DnsSplitRows;
#define DNS_SPLIT_HI_ONLY 1u
/* DNS_SPLIT_PLAIN (ABI v12): HALF ROWS -- the row is written as K plain f16 values, no scale, no exponent (xexp may be NULL), no lo
 * plane: tcnn's own activation format (its networks round their fp32 inputs to half precision, models/decoder.py:93-94,123-125),
 * what dns_mlp_fwd_half / dns_mlp_bwd_half read.  160 bytes per [OneBlob | grid] row instead of 320. */
#define DNS_SPLIT_PLAIN 2u
/* dns_encode_fwd with the (OneBlob | grid) row written in the split-row format: xs_out [P, ldxs] halfs (ldxs >= 2 K, or >= K
 * with DNS_SPLIT_HI_ONLY; K = 3 n_bins + 2 n_levels, both parts multiples of 8), xexp [P]; f32_out (NULL = skip) [P, ld32]
 * additionally receives the fp32 row (the streaming dW_in kernel dns_mlp_dwin reads fp32 rows).  x_out, dy_dx, bound as in
 * dns_encode_fwd.  The exponent is scale_exp(max |row|) -- bit for bit what the MLP kernels derive from the fp32 row, so a
 * one-segment network gives IDENTICAL results on either input form. */
int dns_encode_fwd_split(const float* in, const double* bound, uint32_t P, uint32_t n_bins, const float* table,
                         const DnsGridMeta* meta, float* x_out, float* f32_out, uint32_t ld32, void* xs_out, uint32_t ldxs,
                         int32_t* xexp, uint32_t flags, float* dy_dx, void* stream);

/* Backward.  x [P,3] normalised coordinates.  d_pe / d_grid may be NULL.  d_table (+=) [total_rows,F]
 * (NULL = skip), d_x [P,3] (overwritten; NULL = skip) = dL/dx of the NORMALISED coordinate; if
 * bound != NULL it is scaled by 1/(b1-b0) so it is dL/d(world point).  dy_dx: NULL, or what dns_encode_fwd wrote for the
 * same points and table (then the grid part of d_x needs no table access).  ws: 8-byte aligned scratch of
 * dns_encode_bwd_ws_floats(P, meta, flags, queue_cap) floats for the LDS-binned table scatter (NULL = per-corner atomics).
 * flags selects the form of the table scatter (tcnn kernel_grid_backward).  DNS_SCATTER_AUTO: HASHED levels of more than 8192 rows
 * and DENSE levels of at least six 8192-row chunks through pair lists (ABI v10, below: 64-bit fixed-point LDS bins, exact
 * order-independent sums per workgroup), the other levels through the LDS-bin sweep (FLOAT64 bins, ds_add_f64: sums good to ~1e-16
 * of the terms but dependent on their order; dense levels keep a cell's 8 x 2 sums in registers while consecutive points stay in
 * the cell).  Whatever the form, a chunk's sums reach d_table by float atomics, so d_table is not bit-reproducible from run to run.
 * _ATOMIC = one float atomic per corner (tcnn's form; order-dependent fp32 sums); _BINNED = the sweep for every level; _QUEUES =
 * per-chunk queues of {row, w g0, w g1} for every multi-chunk level (rounds 1-3's form for large tables; fixed-point bins); neither
 * uses lists unless DNS_SCATTER_LISTS is added.  queue_cap: 0, or the entry capacity of each queue and hashed list (what does
 * not fit falls back to float atomics) -- same value in both calls.  A NaN / Inf in d_grid gives a non-finite d_table in
 * every form. */
#define DNS_SCATTER_AUTO 0u
#define DNS_SCATTER_ATOMIC 1u
#define DNS_SCATTER_BINNED 2u
#define DNS_SCATTER_QUEUES 3u
#define DNS_SCATTER_MASK 3u
/* | DNS_SCATTER_REPLAY (ABI v9): hashed levels of at most 2^16 rows handled by the LDS-bin form keep the 8 corner rows of every point
 * (eight 16-bit values, written once per call beside the level-major gradient copy: 16 more bytes of workspace per point and
 * level, counted by dns_encode_bwd_ws_floats with the same flags) and the 8 chunk visits of a level replay them instead of
 * hashing the 8 corners again.  Same sums. */
#define DNS_SCATTER_REPLAY 0x10u
/* | DNS_SCATTER_LISTS (ABI v10; implied by _AUTO, may be added to _BINNED / _QUEUES): PAIR LISTS.  Pass 1 hashes each point-level
 * once and appends one 32-bit word {point, x-pair of corners} to the list of the pair's 4096-row chunk (16 bytes per point-level
 * instead of the 8 chunk visits of the sweep or the 192 bytes of the queues; workspace counted by dns_encode_bwd_ws_floats with
 * the same flags), pass 2 gives every lane one entry, re-forms its two rows and weights from the point and adds into the chunk's
 * 64-bit fixed-point LDS bins.  Hashed levels (2^14 .. 2^20 rows): fixed-capacity lists (uniform hash), a fixed number of jobs per
 * list.  Dense levels: the lists fill by where the rays are -- they are sized EXACTLY (a counting sweep, a one-workgroup scan that
 * places every chunk's list inside the level's 8 P-word region, a writing sweep) and cut into jobs of 8192 entries from the actual
 * counts.  P < 2^30. */
#define DNS_SCATTER_LISTS 0x20u
int dns_encode_bwd(const float* x, const double* bound, uint32_t P, uint32_t n_bins,
                   const float* table, const DnsGridMeta* meta,
                   const float* d_pe, uint32_t ld_dpe, const float* d_grid, uint32_t ld_dgrid,
                   float* d_table, float* d_x, const float* dy_dx, float* ws, uint32_t flags, uint32_t queue_cap,
                   void* stream);
uint64_t dns_encode_bwd_ws_floats(uint32_t P, const DnsGridMeta* meta, uint32_t flags, uint32_t queue_cap);

/* Debug / parity: absolute table rows of the 8 corners of every level, [P, n_levels, 8] uint32. */
int dns_hashgrid_indices(const float* x, uint32_t P, const DnsGridMeta* meta, uint32_t* rows, void* stream);

/* ---- bias-free ReLU MLP (tcnn CutlassMLP; decoder.py:58-64,84-90,101-116, mapping.py:737-743) ----
 * params: flat fp32 [n_neurons*n_in | (n_hidden_layers-1)*n_neurons^2 | out_pad*n_neurons], row-major
 * matrices, out_pad = next multiple of 16 of n_out.  n_neurons in {32,64}; n_hidden_layers in {1,2};
 * n_in a multiple of 8 and <= 128; n_out <= 64; x 16-byte aligned with ldx % 4 == 0.
 * Arithmetic: every fp32 operand is split into two f16 parts (power-of-two scaled per matrix / per point) and a product
 * is three v_mfma_f32_32x32x16_f16 with fp32 accumulation -- error below an fp32 fma chain's;
 * DNS_MLP_FP16 keeps the leading part only (tcnn's own precision: fp16 operands, fp32 accumulate).
 * Two-segment input (x2 != NULL): input columns [0, n_in1) are read from x, columns [n_in1, n_in) from
 * x2[row*ldx2 + (col - n_in1)] -- the torch.cat((pe, features), -1) inputs of decoder.py:73,93,123-124 without
 * the copy.  n_in1 % 4 == 0; x2 16-byte aligned, ldx2 % 4 == 0 -- dns_mlp_fwd alone takes an x2 of any 4-byte alignment and any
 * ldx2 >= n_in - n_in1 (a column slice of another matrix, e.g. the latent columns fine[:, 1:] of the [P, 33] rows of the fine
 * decoders: the forward-only frame render builds no packed copy).  x2 == NULL: one segment (ldx2, n_in1 ignored).
 * A workgroup handles 128 consecutive point SLOTS.  row_index (NULL = identity) maps slot -> row of x / x2 / y /
 * dy / d_x / d_x2, -1 = padding slot (computes on zeros, stores nothing).  tile_group (NULL = one net) gives, per
 * 128-slot tile, the weight set: params + tile_group[t]*param_stride (-1 = skip the tile) -- the per-class
 * fine decoders of slams/mapping.py:590-601 without gathering activations. */
#define DNS_MLP_FP16 0x100u   /* flags of dns_mlp_fwd; bit 8 of accumulate_dx of dns_mlp_bwd */
/* DNS_MLP_PREPARED (flags of dns_mlp_fwd, bit 9 of accumulate_dx of dns_mlp_bwd): `params` points to what dns_mlp_prepare wrote
 * for the same (n_in, n_out, n_neurons, n_hidden_layers) -- the scaled, split operand images of the kernels' LDS prologue --
 * instead of to the fp32 matrices: every workgroup then copies its weight set's images in (L2 hits) instead of loading,
 * reducing, splitting and scattering the matrices itself.  Results are bit-identical.  The weights change once per optimiser
 * step and are used by ~10 launches of hundreds of workgroups each; param_stride keeps its meaning (fp32 floats per weight
 * set: it still addresses d_params), the prepared sets are dns_mlp_prepared_floats(...) floats apart.
 * dns_mlp_prepare: params [n_sets, param_stride] -> prepared [n_sets, dns_mlp_prepared_floats] (16-byte aligned). */
#define DNS_MLP_PREPARED 0x200u
/* DNS_MLP_NO_DWIN (bit 10 of accumulate_dx of dns_mlp_bwd): the backward kernel leaves dH_1 in ws as always, but the second,
 * streaming kernel (dW_in = dH_1^T x) is NOT launched: the caller launches it with dns_mlp_dwin -- same x / ws / d_params /
 * slot arguments -- possibly on ANOTHER stream: it is memory-bound and independent of everything that follows in the
 * backward pass except d_params, so it runs beside the next network's (vector-bound) backward kernel (fused_step.MapStep). */
#define DNS_MLP_NO_DWIN 0x400u
/* DNS_MLP_DX_FIRST (bit 11 of accumulate_dx of dns_mlp_bwd, two-segment input only): only d_x -- the gradient of input columns
 * [0, n_in1) -- is produced; d_x2 may be NULL and the products of the second segment's columns are skipped (Decoder.merge: the
 * OneBlob columns carry the pose gradient, the looked-up image code has none, models/decoder.py:70-74). */
#define DNS_MLP_DX_FIRST 0x800u
/* DNS_MLP_LIVE_IN(n) (bits 16-23 of the flags of dns_mlp_fwd / dns_mlp_dwin and of accumulate_dx of dns_mlp_bwd; ABI v9): input
 * columns [n, n_in) are IDENTICALLY ZERO for every row of this call -- the reference's colour / logit networks when no 2-D feature
 * code is attached (slams/mapping.py:553-557 multiplies a zero code: BASELINE configs 1, 2, 4, 5).  The kernels then run as an
 * n-input network on the same parameter tensor (W_in keeps its row stride n_in): the zero columns are not read (x2 needs only
 * n - n_in1 columns), their K-steps are skipped, no d_x / d_x2 is produced for them and their columns of dW_in receive nothing
 * (their exact gradient is zero).  n: a multiple of 8 with n_in1 < n <= n_in.  Same results as the full-width call on rows padded
 * with zeros up to fp32 rounding (W_in's power-of-two operand scale is taken over its live columns). */
#define DNS_MLP_LIVE_IN(n) ((uint32_t)(n) << 16)
/* DNS_MLP_DX_FROM(c) (bits 24-30 of accumulate_dx of dns_mlp_bwd): the input gradient of columns [0, c) has no consumer -- it is
 * neither formed (whole 32-column tiles) nor stored.  The smoothness lattice's coarse network (slams/mapping.py:129-159) needs
 * the hash-grid columns' gradient only: its points carry no pose.  c: a multiple of 4 below n_in. */
#define DNS_MLP_DX_FROM(c) ((uint32_t)(c) << 24)
int dns_mlp_dwin(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, uint32_t n_in, uint32_t n_neurons,
                 uint32_t n_hidden_layers, float* d_params, const float* ws, uint32_t n_slots, const int32_t* row_index,
                 const int32_t* tile_group, uint32_t param_stride, uint32_t flags, void* stream);
uint64_t dns_mlp_prepared_floats(uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers);
/* (ABI v11: `flags` = DNS_MLP_LIVE_IN(n) or 0.  With a live width the images are built -- and the blob is laid out:
 * dns_mlp_prepared_floats(n, ...) floats per weight set -- for the n-input network on the same parameter tensor; dns_mlp_fwd /
 * dns_mlp_bwd then take DNS_MLP_PREPARED | DNS_MLP_LIVE_IN(n) together.) */
int dns_mlp_prepare(const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers,
                    uint32_t n_sets, uint32_t param_stride, float* prepared, uint32_t flags, void* stream);
int dns_mlp_fwd(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1,
                const float* params, uint32_t n_in, uint32_t n_out,
                uint32_t n_neurons, uint32_t n_hidden_layers, float* y, uint32_t ldy, uint32_t n_slots,
                const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride,
                float* h_save /* NULL, or [n_hidden_layers, n_slots, n_neurons]: also write the hidden activations (inspection, or
                                 to hand them to dns_mlp_bwd as h_saved; the backward does not NEED them) */,
                uint32_t flags /* 0 or DNS_MLP_FP16 */, void* stream);

/* Backward.  Hidden activations are RECOMPUTED from x, or -- with h_saved = what dns_mlp_fwd wrote to h_save for the same slots,
 * fp32-grade mode with d_params only -- read back (bit-identical results; trades the recompute's vector work for 512 B of traffic
 * per point each way).  One kernel produces d_x, dW_hidden
 * and dW_out and leaves dH_1 in ws; a second, streaming kernel forms dW_in = dH_1^T x.  d_x [rows, lddx] (columns
 * [0, n_in1)) and, with a two-segment input, d_x2 [rows, lddx2] (columns [n_in1, n_in)) are written for valid slots (d_x NULL
 * = skip both); d_params (+=) same layout as params (+ group*param_stride), NULL = skip (no weight-gradient work at all:
 * the tracker's frozen scene).  ws: 16-byte aligned workspace of dns_mlp_bwd_ws_floats(...) floats (needed with d_params).
 * accumulate_dx: bit 0 -> d_x += instead of =, bit 1 -> d_x2 += (several networks reading one input add their input
 * gradients in place instead of through separate buffers and an add); | DNS_MLP_FP16: as in the forward (weight gradients
 * then use two bf16 parts, 16 significant bits). */
int dns_mlp_bwd(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1,
                const float* dy, uint32_t lddy, const float* params,
                uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers,
                float* d_x, uint32_t lddx, float* d_x2, uint32_t lddx2, float* d_params, float* ws, uint32_t n_slots,
                const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride, const float* h_saved,
                int accumulate_dx, void* stream);
uint64_t dns_mlp_bwd_ws_floats(uint32_t n_slots, uint32_t n_neurons, uint32_t n_hidden_layers);
/* The same two entry points reading their input in the split-row format (above): x [host] describes the rows of input columns
 * [0, n_in1) (or all n_in columns when x2 == NULL), x2 [host] those of columns [n_in1, n_in); n_in, n_in1 multiples of 16.  The
 * operand fragments go from memory straight into the matrix instruction (one 16-byte load per part, lane and K-step).  With
 * DNS_MLP_FP16 only the hi planes are read (lo_off may be 0).  Everything else as in dns_mlp_fwd / dns_mlp_bwd, except:
 * dns_mlp_bwd_split never launches the streaming dW_in kernel (DNS_MLP_NO_DWIN is implied): the caller runs dns_mlp_dwin on fp32
 * rows or dns_mlp_dwin_split on the same split rows. */
int dns_mlp_fwd_split(const DnsSplitRows* x, const DnsSplitRows* x2, uint32_t n_in1, const float* params, uint32_t n_in,
                      uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, float* y, uint32_t ldy, uint32_t n_slots,
                      const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride, uint32_t flags, void* stream);
int dns_mlp_bwd_split(const DnsSplitRows* x, const DnsSplitRows* x2, uint32_t n_in1, const float* dy, uint32_t lddy,
                      const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers,
                      float* d_x, uint32_t lddx, float* d_x2, uint32_t lddx2, float* d_params, float* ws, uint32_t n_slots,
                      const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride, int accumulate_dx,
                      void* stream);

/* ---- the same network in tcnn's OWN arithmetic: HALF ROWS (ABI v12) -------------------------------------------------------
 * Replaces tcnn.Network{CutlassMLP} as the reference runs it (models/decoder.py:58-64,84-90,101-116 construct half-precision
 * networks; :94 `.float()`s their output): f16 activations, f16 weight operands, fp32 accumulation in v_mfma_f32_32x32x16_f16, a
 * STATIC loss scale on the gradients (tcnn's default: 128).  BASELINE configs[4].  No row maxima, no per-point exponents, no hi / lo
 * parts: x / x2 are plain f16 rows (DNS_SPLIT_PLAIN: dns_encode_fwd_split, dns_feature_block_split), [rows, ldx] halfs, 16-byte
 * aligned, ldx % 8 == 0; n_in (the live width) % 16 == 0, n_in1 % 8 == 0; params / d_params: the fp32 master weights in the layout
 * of dns_mlp_fwd (rounded to f16 when a workgroup builds its LDS images -- ONE image per matrix, read row-wise for W and with
 * ds_read_b64_tr_b16 for W^T).  y, dy, d_x, d_x2 stay fp32 rows (the compositing / loss / encoder-backward kernels' format).
 * row_index / tile_group / param_stride as in dns_mlp_fwd.  flags: DNS_MLP_LIVE_IN(n) (n % 16 == 0) or 0.
 * dns_mlp_bwd_half: ONE kernel produces d_x (/ d_x2) and ALL weight gradients, dW_in included (no workspace, no dns_mlp_dwin);
 * hidden activations are recomputed.  dY is multiplied by loss_scale before it is rounded to f16, every hidden gradient is f16
 * at that scale, d_x and d_params receive 1 / loss_scale times the fp32 sums.  accumulate_dx: bits 0 / 1 (d_x / d_x2 +=),
 * DNS_MLP_DX_FIRST, DNS_MLP_LIVE_IN(n), DNS_MLP_DX_FROM(c) as in dns_mlp_bwd. */
int dns_mlp_fwd_half(const void* x, uint32_t ldx, const void* x2, uint32_t ldx2, uint32_t n_in1, const float* params, uint32_t n_in,
                     uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, float* y, uint32_t ldy, uint32_t n_slots,
                     const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride, uint32_t flags, void* stream);
int dns_mlp_bwd_half(const void* x, uint32_t ldx, const void* x2, uint32_t ldx2, uint32_t n_in1, const float* dy, uint32_t lddy,
                     const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, float* d_x,
                     uint32_t lddx, float* d_x2, uint32_t lddx2, float* d_params, uint32_t n_slots, const int32_t* row_index,
                     const int32_t* tile_group, uint32_t param_stride, int accumulate_dx, float loss_scale, void* stream);

/* ---- occupancy compositing (raw2nerf_color, utils/common.py:506-537, + the logit composite of
 * slams/mapping.py:633 / slams/tracking.py:212) ----------------------------------------------
 * raw [N,S,4] (rgb, occ), z [N,S], logits [N,S,C] (C may be 0).  Outputs depth,var [N], rgb [N,3],
 * weights [N,S], sem [N,C]. */
int dns_composite_fwd(const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S, uint32_t C,
                      float* depth, float* var, float* rgb, float* weights, float* sem, void* stream);
/* d_weights may be NULL.  d_raw [N,S,4], d_logits [N,S,C] overwritten. */
int dns_composite_bwd(const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S, uint32_t C,
                      const float* d_depth, const float* d_var, const float* d_rgb, const float* d_weights,
                      const float* d_sem, float* d_raw, float* d_logits, void* stream);

/* The same pair with the colour network's output activation folded in (ABI v9): DNS_COMPOSITE_RGB_LOGITS -- raw[..., 0:3] holds
 * the colour network's LOGITS; the kernels apply sigmoid (models/decoder.py:124) on the fly, and the backward's d_raw[..., 0:3] is
 * the gradient w.r.t. those logits (d colour * s (1 - s)), i.e. directly the colour network's output gradient: no separate
 * sigmoid pass over [P,4] forward, none backward. */
#define DNS_COMPOSITE_RGB_LOGITS 1u
int dns_composite_fwd_ex(const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S, uint32_t C, float* depth,
                         float* var, float* rgb, float* weights, float* sem, uint32_t flags, void* stream);
int dns_composite_bwd_ex(const float* raw, const float* z, const float* logits, uint32_t N, uint32_t S, uint32_t C,
                         const float* d_depth, const float* d_var, const float* d_rgb, const float* d_weights, const float* d_sem,
                         float* d_raw, float* d_logits, uint32_t flags, void* stream);

/* ---- fused loss reductions --------------------------------------------------------------------
 * Mapper: photometric / depth / label / latent losses (slams/mapping.py:110-126) + get_opacity_loss
 * (utils/common.py:769-802) and their weighted sum (mapping.py:906-907); tracker = 1: the three masked losses of
 * slams/tracking.py:85-96 (depth term divided by sqrt(var + 1e-10)).
 * lambdas [host, 8 floats] = lambda_p, lambda_d, lambda_l, lambda_lt, lambda_fs, lambda_opacity, truncation, sigma.
 * Rays: pred_color [N,3], pred_depth [N], pred_var [N] (tracker), pred_logits [N,C], gt_color [N,3], gt_depth [N],
 * gt_label [N] int64, valid [N] uint8 (NULL = all rays count).  Points (mapper): fine, coarse [N*S, L], z [N,S].
 * Three calls: dns_loss_sums fills sums[0..15] (numerators and counts; multi-GPU callers all-reduce those 16 here;
 * the buffer must hold DNS_LOSS_SUMS_FLOATS floats, the rest is reduction workspace),
 * dns_loss_finalize turns it into out[16] = {p, d, l, lt, fs, op, total, -, coefficients...},
 * dns_loss_bwd writes d(total * g_total)/d(inputs): d_color, d_depth, d_var (tracker; may be NULL), d_logits,
 * d_fine, d_coarse (overwritten; both NULL in mapper mode: the ray part only, see dns_loss_bwd_points).  ldd_fine: row stride of d_fine in floats (0 = L, contiguous) -- lets the caller place the
 * fine decoder's loss gradient straight into a wider row that later kernels add to (fused_step.MapStep). */
#define DNS_LOSS_SUMS_FLOATS (32 + 5 * 1024)
/* dns_loss_rays (ABI v10) = dns_loss_sums + dns_loss_finalize + the RAY part of dns_loss_bwd where nothing has to happen between the
 * sums and the coefficients (one rank, no all-reduce): the point pass's partial sums (mapper) and ONE single-workgroup kernel that
 * walks the rays, reduces in a fixed order (no atomics: sums[0..15] are reproducible), finalises out[16] and writes d_color,
 * d_depth, d_var (tracker; may be NULL), d_logits.  1 <= N <= 2^20.  The mapper's point gradients follow with dns_loss_bwd_points.
 * For SMALL ray counts (the tracker's 500-1000 rays: one launch instead of four); at 4096 rays the single workgroup takes 63 us
 * where the three launches take 22. */
int dns_loss_rays(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                  const float* pred_color, const float* pred_depth, const float* pred_var, const float* pred_logits,
                  const float* gt_color, const float* gt_depth, const int64_t* gt_label, const uint8_t* valid,
                  const float* fine, const float* coarse, const float* z, float* sums, float* out, const float* g_total,
                  float* d_color, float* d_depth, float* d_var, float* d_logits, void* stream);
/* dns_loss_finalize_bwd (ABI v10) = dns_loss_finalize + the RAY part of dns_loss_bwd in one launch: every workgroup of the rays'
 * backward derives the terms / coefficients from sums[0..15] itself, workgroup 0 writes out[16] for the kernels that follow
 * (dns_loss_bwd_points, dns_keep_best).  Works behind an all-reduce of the sums too.  N >= 1. */
int dns_loss_finalize_bwd(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                          const float* sums, float* out, const float* g_total, const float* pred_color,
                          const float* pred_depth, const float* pred_var, const float* pred_logits, const float* gt_color,
                          const float* gt_depth, const int64_t* gt_label, const uint8_t* valid, float* d_color,
                          float* d_depth, float* d_var, float* d_logits, void* stream);
int dns_loss_sums(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                  const float* pred_color, const float* pred_depth, const float* pred_var, const float* pred_logits,
                  const float* gt_color, const float* gt_depth, const int64_t* gt_label, const uint8_t* valid,
                  const float* fine, const float* coarse, const float* z, float* sums, void* stream);
int dns_loss_finalize(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                      const float* sums, float* out, void* stream);
int dns_loss_bwd(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, int tracker,
                 const float* out, const float* g_total, const float* pred_color, const float* pred_depth,
                 const float* pred_var, const float* pred_logits, const float* gt_color, const float* gt_depth,
                 const int64_t* gt_label, const uint8_t* valid, const float* fine, const float* coarse, const float* z,
                 float* d_color, float* d_depth, float* d_var, float* d_logits, float* d_fine, float* d_coarse,
                 uint32_t ldd_fine, void* stream);

/* The point part of dns_loss_bwd alone (dns_loss_bwd with d_fine = d_coarse = NULL runs the ray part alone), with the
 * compositing's gradient of the occupancy logit added in: d_fine[p, 0] += d_occ[p * ld_occ] (d_occ NULL = nothing) -- the occupancy
 * is column 0 of the fine latents (slams/mapping.py:626-627), so the sum the reference's autograd forms needs no
 * read-modify-write pass over a strided column.  Call order: dns_loss_bwd (rays) -> dns_composite_bwd -> this. */
int dns_loss_bwd_points(const float* lambdas, uint32_t N, uint32_t S, uint32_t C, uint32_t L, const float* out,
                        const float* g_total, const float* gt_depth, const uint8_t* valid, const float* fine, const float* coarse,
                        const float* z, float* d_fine, float* d_coarse, uint32_t ldd_fine, const float* d_occ, uint32_t ld_occ,
                        void* stream);

/* ---- fused Adam --------------------------------------------------------------------------------
 * torch.optim.Adam.step with default betas / eps, no weight decay, no amsgrad (reference slams/mapping.py:464,910,
 * slams/tracking.py:120-124,339) over up to 32 parameter tensors in one launch.  tensors [host]: per tensor the
 * parameter, its gradient (NULL = skip), first / second moment buffers, element count and learning rate.
 * state: 3 device floats {step count, 1-beta1^t, 1-beta2^t}; zero it when the optimiser is created ("fresh moments
 * every optimize() call", slams/mapping.py:464); each call advances the count on the device. */
typedef struct DnsAdamTensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  uint64_t n;
  float lr;
} DnsAdamTensor;
int dns_adam_step(const DnsAdamTensor* tensors, uint32_t n_tensors, float beta1, float beta2, float eps,
                  float* state, void* stream);

/* ---- the tracker's optimise iteration as ONE kernel + a pose kernel (ABI v11) ---------------------------------------
 * Replaces the per-iteration body of Tracker.run (reference slams/tracking.py:313-340: get_target_samples :128-186,
 * renderer :188-214, the three masked losses :85-96, backward to the pose, Adam on (quat, T), keep-best :326-335) for the
 * frozen scene.  A 256-thread workgroup owns 2 (S > 32) or 4 rays from the pixel draw to their share of the pose gradient:
 * sampling, OneBlob + hash-grid encoding, coarse / colour / logit networks forward, compositing, losses, compositing and
 * network backward, encoding backward; a one-workgroup pose kernel sums the workgroups' partial results, forms the loss,
 * keeps the best pose, runs the quaternion chain rule and the Adam update.  Two launches per iteration.
 *   The draws of ALL n_iters iterations of a frame are made up front: pix [n_iters, n_rays] int64 (index into the window
 *   [H0,H1) x [W0,W1)), t_surf / t_zero [n_iters, n_surface], dmax [n_iters] (bits of max(gt_depth > 0) over each draw;
 *   dns_track_fused_begin computes it and forces the mid sample of every t_surf row in one launch: neither depends on the
 *   pose); t_uniform [n_uniform] is shared.  iter [1] uint32: device counter of the frame's iterations (zero it per frame):
 *   launch k reads draws min(iter, n_iters - 1) and the pose kernel advances it -- every launch of a frame has the SAME
 *   arguments, so one captured iteration replays n_iters times.
 *   quat [4], trans [3]: the pose, updated in place.  adam_m / adam_v [8] (quat at 0, trans at 4), adam_state [3] (step count,
 *   1 - beta1^t, 1 - beta2^t): zero them per frame.  best_loss [1] (+inf per frame), best_cam [7].
 *   w_coarse / w_color / w_logit: dns_mlp_prepare images of the three networks (inputs 3 n_bins + 2 n_levels -> hidden + 1;
 *   3 n_bins + n_feat -> 3 and -> n_class).  code [n_rays * S, code_dim] or NULL: the 2-D feature code of every sample.
 *   n_feat = hidden + code_dim; WITHOUT a code either n_feat = the networks' full second-segment width (zero code columns, images
 *   of the full networks) or n_feat = hidden with images prepared with DNS_MLP_LIVE_IN(3 n_bins + hidden): no feature block is
 *   built then, the colour / logit networks read the latent out of the coarse network's output rows.
 *   ws: dns_track_fused_ws_floats(a) floats, 256-byte aligned (rows the workgroups hand from phase to phase).
 *   out [8]: loss terms p, d, l, total, n_valid of THIS iteration (before the update); g_quat [4], g_trans [3]: its gradient.
 * Supported: S <= 64, networks 64 x 2 or 32 x 1, 64 < 3 n_bins + 2 n_levels <= 96, 96 < 3 n_bins + n_feat <= 128; otherwise
 * DNS_E_ARG (callers fall back to the launch sequence of the single entry points). */
typedef struct DnsTrackFused {
  const float* color;
  const float* depth;
  const float* label;
  int32_t H, W, H0, H1, W0, W1;
  const double* cam;                 /* fx fy cx cy */
  const double* bound;               /* [3][2] */
  const int64_t* pix;
  const float* t_uniform;
  const float* t_surf;
  const float* t_zero;
  const uint32_t* dmax;
  uint32_t* iter;
  uint32_t n_iters;
  uint32_t n_uniform, n_surface, n_rays;
  float* quat;
  float* trans;
  const float* table;
  const DnsGridMeta* meta;
  uint32_t n_bins;
  const float* w_coarse;
  const float* w_color;
  const float* w_logit;
  uint32_t n_neurons, n_hidden_layers, hidden, n_feat, n_class;
  const float* code;
  uint32_t code_dim;
  float lambda_p, lambda_d, lambda_l;
  float* ws;
  float* adam_m;
  float* adam_v;
  float* adam_state;
  float lr_quat, lr_trans, beta1, beta2, eps;
  float* best_loss;
  float* best_cam;
  float* out;
  float* g_quat;
  float* g_trans;
} DnsTrackFused;
uint64_t dns_track_fused_ws_floats(const DnsTrackFused* a);
int dns_track_fused_begin(const int64_t* pix_all, uint32_t n_rays, uint32_t n_iters, const float* depth, int W, int H0, int W0,
                          int W1, float* t_surf_all, uint32_t n_surface, uint32_t* dmax_all, void* stream);
int dns_track_fused_iter(const DnsTrackFused* a, void* stream);

/* ---- small fused helpers of the mapping iteration ----------------------------------------------
 * Total-variation smoothness of coarse[:, 0] on a lattice of nx x n x n points (x-major; nx = n: the n^3 cube of
 * slams/mapping.py:151-157), divided by sample_points^3.  lat [nx*n*n, ld] (the coarse latents, occupancy in column 0).
 * halo != 0: the last x-plane belongs to the next slab of a lattice cut along x (one slab per GPU) -- it closes the
 * x-differences of plane nx-2 and contributes no y / z differences, so the slabs' values sum to the cube's value.
 * out: 1 float.  Backward: d_lat [nx*n*n, ld] = g[0] * d loss / d lat (column 0; the other columns are zeroed). */
int dns_tv_fwd(const float* lat, uint32_t ld, uint32_t nx, uint32_t n, int halo, uint32_t sample_points, float* out,
               void* stream);
int dns_tv_bwd(const float* lat, uint32_t ld, uint32_t nx, uint32_t n, int halo, uint32_t sample_points, const float* g,
               float* d_lat, void* stream);

/* Counting sort of P points by weight-set id into 128-slot tiles for dns_mlp_fwd/bwd (the per-class dispatch of
 * Mapper.fine_fn, slams/mapping.py:590-601).  slot_of_point [P] int64 (negative = no network).  Outputs row_index
 * [n_slots] (-1 = padding), tile_group [n_slots/128] (-1 = skip; groups with fewer than min_count points are
 * skipped, mapping.py:597).  n_slots: multiple of 128, >= P + 127 * n_groups.  ws: 512 uint32 of scratch. */
int dns_group_slots(const int64_t* slot_of_point, uint32_t P, uint32_t n_groups, uint32_t min_count, uint32_t n_slots,
                    uint32_t* ws, int32_t* row_index, int32_t* tile_group, void* stream);

/* Device-side error word.  Kernels that consume DEVICE counters (the cursors of dns_group_slots: a stale or corrupted cursor
 * would otherwise be an out-of-bounds store, not an error code) clamp to their buffers' capacity and, when they had to, set a
 * bit of a sticky per-device word in pinned host memory (allocated by dns_init(); never device memory).  Every entry point
 * polls the word after its launches -- no synchronisation: the bit shows up once the faulting kernel has run -- and from
 * then on returns DNS_E_LAUNCH with the cause in dns_last_error() until the word is cleared.
 *   dns_device_error(0) -> the word (0 = none), dns_device_error(1) -> the word, then cleared.
 *   bit 0: dns_group_slots / dns_group_scatter: a group's cursor ran past n_slots
 * dns_group_scatter is the scatter step of dns_group_slots alone, on caller-supplied cursors [n_groups] (uint32, the first
 * free slot of every group; advanced): the hardening test drives it with a deliberately stale cursor. */
int dns_device_error(int clear);
int dns_group_scatter(const int64_t* slot_of_point, uint32_t P, uint32_t n_groups, uint32_t* cursor, uint32_t n_slots,
                      int32_t* row_index, void* stream);

/* ---- glue of one mapping iteration as single launches (csrc/step.hip) ---------------------------
 * Each replaces a handful of elementwise torch ops of the reference's iteration; used by dns_slam_amd/fused_step.py.
 * dns_class_slots: slot_of_point[p] = lut[label of p] (-1 when the label is outside [0, n_lut)); the label of point p is
 *   labels[p mod N] when tiled (the reference's gt_label.repeat(1, S) layout, slams/mapping.py:613), else labels[p / S].
 * dns_feature_block: feat[p, 0:hidden] = fine[p, 1:1+hidden], feat[p, hidden:hidden+C] = code[p, :] * trunc(p) with
 *   trunc = (1 - [z < 0.95 d]) (1 - [z > 1.05 d]) [d > 0], d = gt_depth[p / S] (slams/mapping.py:553-556; code NULL = zeros),
 *   and raw[p, 3] = fine[p, 0] (raw [P,4], NULL = skip) -- the colour / logit networks' feature input (models/decoder.py:123)
 *   and the occupancy column of the compositing input (slams/mapping.py:627).  hidden, C multiples of 4; feat, code 16-byte aligned.
 * dns_rgb_sigmoid: raw[p, 0:3] = sigmoid(raw[p, 0:3]) in place (models/decoder.py:124), column 3 untouched.
 * dns_raw_bwd: the backward of both: d_col[p, 0:3] = d_raw[p, 0:3] s (1 - s) with s = raw[p, 0:3], d_col[p, 3] = 0, and
 *   d_occ[p * ld_occ] = (accumulate ? += : =) d_raw[p, 3].
 * dns_lattice_points: normalised coordinates of the n^3 smoothness lattice (slams/mapping.py:133-143 as one float64 affine
 *   map): pts[i,j,k] = float(mar + r6[0:3] * off + r6[3:6] * vox + (i,j,k) * vox); consts9 [host, 9 doubles] = vox, off, mar;
 *   r6 [device, 6 floats] = the offset and jitter draws.  order (NULL = x-major): output row m holds lattice element order[m];
 *   count (0 = n^3): the number of rows / entries of order -- a rank's slab of x-planes of the ONE lattice in union-batch mode
 *   (x-major index i n^2 + j n + k) -- a Morton order of the elements makes the hash-grid gather of the lattice 2.3x faster
 *   (neighbouring rows share table lines: 112 -> 49 us at 63^3), the caller maps the network's output back for the TV kernel. */
int dns_class_slots(const int64_t* labels, uint32_t N, uint32_t S, int tiled, const int64_t* lut, uint32_t n_lut,
                    int64_t* slot_of_point, void* stream);
int dns_feature_block(const float* fine, uint32_t ld_fine, uint32_t hidden, const float* code, uint32_t C, const float* z,
                      const float* gt_depth, uint32_t N, uint32_t S, float* feat, uint32_t ld_feat, float* raw, void* stream);
/* dns_feature_block with the block written in the split-row format (above; ABI v9): xs_out [P, ldxs] halfs (hidden + C values
 * per row), xexp [P] (xs_out NULL: the fp32 block only); feat (NULL = skip) receives the fp32 block.  n_ref > 1: code is [n_frames][n_ref]
 * [pts_per_frame][C] and the n_ref slabs of a point are AVERAGED before the truncation mask -- the mean over the reference frames
 * of Decoder.merge's latents (models/decoder.py:76, utils/common.py:677).  max(hidden, C) / 4 must be a power of two <= 16. */
int dns_feature_block_split(const float* fine, uint32_t ld_fine, uint32_t hidden, const float* code, uint32_t C, uint32_t n_ref,
                            uint32_t pts_per_frame, const float* z, const float* gt_depth, uint32_t N, uint32_t S, float* feat,
                            uint32_t ld_feat, void* xs_out, uint32_t ldxs, int32_t* xexp, uint32_t flags, float* raw, void* stream);
int dns_rgb_sigmoid(float* raw, uint32_t P, void* stream);
int dns_raw_bwd(const float* d_raw, const float* raw, uint32_t P, float* d_col, float* d_occ, uint32_t ld_occ, int accumulate,
                void* stream);
int dns_lattice_points(const float* r6, const double* consts9, uint32_t n, const int32_t* order, uint32_t count, float* pts,
                       void* stream);
/* dns_draw_finish: the index arithmetic behind one iteration's pixel draws for all K frames (select_uv + select_by_class,
 * utils/common.py:274,313-328) plus what follows from the drawn pixels alone.  The RANDOM NUMBERS come from the caller's
 * generator: i1 [K, n1] int64 uniform picks in [0, HW), u [K, n2] float64 in [0, 1).  Per frame f: pix[f, 0:n1] = i1[f],
 * pix[f, n1 + s] = sorted_flat[start_flat[f, s] + min(int64(u[f, s] * count_f64[f, s]), count_m1[f, s])] (sorted_flat [K * HW]:
 * every frame's pixel indices ordered by class; start_flat already holds the frame's offset f * HW), labels[f, r] =
 * int64(label[f, pix[f, r]]), dmax[f] = bit pattern of max(0, max_r depth[f, pix[f, r]]) -- the form dns_raygen_sample takes
 * with depth_max_given = 1.  pix, labels [K * (n1 + n2)] int64; depth, label [K, HW] fp32. */
int dns_draw_finish(const int64_t* i1, const double* u, const double* count_f64, const int64_t* count_m1,
                    const int64_t* start_flat, const int64_t* sorted_flat, const float* depth, const float* label,
                    uint32_t n_frames, uint32_t n1, uint32_t n2, uint32_t HW, int64_t* pix, int64_t* labels, uint32_t* dmax,
                    void* stream);
/* Tracker glue (slams/tracking.py:171-172, 326-335; utils/common.py:572-574), used by fused_step.TrackStep:
 * dns_track_mask: valid[n] = gt_depth[n] > min_depth && inside[n] (uint8).
 * dns_keep_best: if (loss[0] < best_loss[0]) { best_loss[0] = loss[0]; best_cam[0:7] = (quat[0:4] | trans[0:3]) } -- device-side
 *   keep-best-pose of the tracking loop, no host read of the loss (a NaN loss keeps the old best).
 * dns_force_half: t[idx] = 0.5 unless some t[i] == 0.5 already (the forced mid-sample of sample_along_rays' surface draws). */
int dns_track_mask(const float* gt_depth, const uint8_t* inside, uint32_t N, float min_depth, uint8_t* valid, void* stream);
int dns_keep_best(const float* loss, const float* quat, const float* trans, float* best_loss, float* best_cam, void* stream);
int dns_force_half(float* t, uint32_t n, uint32_t idx, void* stream);

/* ---- 2-D feature lookup (feature_matching / feature_searching, utils/common.py:632-673) --------
 * pts [P,3] world points, w2c [R,16] row-major world->camera of the R reference frames, K [host, 9 floats] intrinsics,
 * feat [R,h,w,C] stem feature maps, CHANNELS LAST.  code [R,P,C] = the bilinear (align_corners) value of the map at
 * the rounded projected full-resolution pixel, zero where the projection is invalid; mask [R,P] uint8 (NULL = skip). */
int dns_feature_gather(const float* pts, const float* w2c, const float* K, const float* feat, uint32_t R, uint32_t P,
                       uint32_t C, int h, int w, int H, int W, float* code, uint8_t* mask, void* stream);

/* The same lookup for the K target frames of one optimise iteration in ONE launch (slams/mapping.py:532-551 per target frame,
 * slams/tracking.py:162-165; ABI v9): reference rr = f R + r (R views per frame) looks at the points of frame f: pts
 * [n_frames, pts_per_frame, 3], w2c [n_frames R, 16], feat [n_frames R, h, w, C] channels last.  code row (rr, p) starts at
 * code + (rr pts_per_frame + p) ld_code -- e.g. column 3 n_bins of the Merge network's input rows -- and rel_out (NULL = skip)
 * [n_frames R, pts_per_frame, 3] receives pts - origin[rr] (refer_p, utils/common.py:675), origin [n_frames R, 3].
 * dns_refer_poses: the w2c / origin of those views: src[rr] >= 0 takes target frame src[rr]'s pose as it stands in the optimiser
 * (quat [K,4] (w,x,y,z), trans [K,3]; get_rotation_from_quad, detached: slams/mapping.py:537-543), src[rr] < 0 the stored pose
 * fixed_c2w[rr] [16] (:545); w2c = its inverse (torch.inverse, :547).
 * dns_merge_dy: backward of (mean over the R views, truncation mask) -- d_lat [n_frames, R, pts_per_frame, C] = d_code[p] *
 * trunc(p) / R for every view (models/decoder.py:76, slams/mapping.py:553-557); d_code [P, ld_dcode] (the first C columns of each
 * row) is CLEARED afterwards: the colour / logit networks add their input gradients into it every iteration.
 * dns_add_ref_sum: d_pts [P,3] += sum over the R views of d_rel [n_frames, R, pts_per_frame, 3] (the points' share of Merge's
 * gradient, through refer_p = pts - refer_o). */
int dns_feature_gather_frames(const float* pts, const float* w2c, const float* origin, const float* K, const float* feat,
                              uint32_t n_frames, uint32_t R, uint32_t pts_per_frame, uint32_t C, int h, int w, int H, int W,
                              float* code, uint32_t ld_code, float* rel_out, void* stream);
int dns_refer_poses(const float* quat, const float* trans, const int32_t* src, const float* fixed_c2w, uint32_t n, float* w2c,
                    float* origin, void* stream);
int dns_merge_dy(float* d_code, uint32_t ld_dcode, uint32_t C, uint32_t R, uint32_t pts_per_frame, const float* z,
                 const float* gt_depth, uint32_t N, uint32_t S, float* d_lat, void* stream);
int dns_add_ref_sum(const float* d_rel, uint32_t R, uint32_t pts_per_frame, uint32_t P, float* d_pts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DNS_HIP_H */

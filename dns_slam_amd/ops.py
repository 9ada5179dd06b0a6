"""torch.autograd bindings of the HIP entry points (include/dns_hip.h).

PyTorch is plumbing here: device buffers, the current stream and the autograd graph.  All arithmetic
happens in libdns_hip.so; nothing in this module has a CPU or eager-torch fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import os

import numpy as np
import torch

from ._lib import DnsGridMeta, check, lib as _rawlib, ptr, require_cuda, stream_ptr


class _TimedLib:
    """Pass-through to libdns_hip.so that, when ``ops.timer`` is armed, brackets every launch-carrying entry point
    with events on the current stream (the stream the kernels are launched on) -- bench.py's per-kernel durations."""

    def __init__(self, raw):
        self._raw = raw
        self.records = None          # list of (name, start_event, end_event, units, key arguments, kernel-span range) while armed
        self.kernels = False         # also collect the library's per-kernel spans (dns_kernel_timing)

    def __getattr__(self, name):
        fn = getattr(self._raw, name)
        if self.records is None or name in ("dns_grid_meta_init", "dns_mlp_bwd_ws_floats", "dns_mlp_prepared_floats", "dns_encode_bwd_ws_floats", "dns_raygen_bwd_ws_floats", "dns_last_error",
                                            "dns_abi_version", "dns_init") or name.startswith("dns_kernel_timing"):
            return fn

        ui = self._UNITS_ARG.get(name)
        info_of = self._INFO.get(name)

        def timed(*a):
            k0 = self._raw.dns_kernel_timing_count() if self.kernels else 0
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            k1 = self._raw.dns_kernel_timing_count() if self.kernels else 0
            units = 0 if ui is None else (a[ui] if not isinstance(ui, tuple) else a[ui[0]] * a[ui[1]])
            self.records.append((name, e0, e1, int(units), info_of(a) if info_of else None, (k0, k1)))
            return rc
        return timed

    # per-call facts the roofline needs: the network shape of an MLP launch and which gradients it produces
    # (n_in = the LIVE input width: with DNS_MLP_LIVE_IN the kernels skip the identically-zero columns, and the roofline counts
    #  the work that is done, not the multiplications by zero the reference's full-width GEMM performs)
    _live = staticmethod(lambda n_in, flags: ((int(flags) >> 16) & 0xff) or n_in)
    @staticmethod
    def _grid_info(meta_ref):
        """levels of a grid and how many of them the scatter sends through pair lists (hashed levels of more than one 8192-row
        chunk, dense levels of at least six: csrc/encode.hip list_plan)"""
        m = getattr(meta_ref, "_obj", None)
        if m is None:
            return None
        L = int(m.n_levels)
        in_lists = lambda l: 8192 < m.size[l] <= (1 << 20) and (m.hashed[l] or (m.size[l] + 8191) // 8192 >= 6)
        return {"n_levels": L, "list_levels": sum(1 for l in range(L) if in_lists(l))}

    _INFO = {"dns_encode_bwd": lambda a: _TimedLib._grid_info(a[5]),
             "dns_mlp_fwd": lambda a: {"n_in": _TimedLib._live(a[6], a[17]), "n_out": a[7], "nn": a[8], "nl": a[9]},
             "dns_mlp_dwin": lambda a: {"n_in": _TimedLib._live(a[5], a[14]), "n_out": 0, "nn": a[6], "nl": a[7]},
             "dns_mlp_bwd": lambda a: {"n_in": _TimedLib._live(a[8], a[23]), "n_out": a[9], "nn": a[10], "nl": a[11], "dx": bool(a[12]),
                                       "dw": bool(a[16])},
             "dns_mlp_fwd_half": lambda a: {"n_in": _TimedLib._live(a[6], a[16]), "n_out": a[7], "nn": a[8], "nl": a[9]},
             "dns_mlp_bwd_half": lambda a: {"n_in": _TimedLib._live(a[8], a[21]), "n_out": a[9], "nn": a[10], "nl": a[11], "dx": bool(a[12]),
                                            "dw": bool(a[16]), "dx_from": (int(a[21]) >> 24) & 0x7f, "acc": int(a[21]) & 3},
             "dns_mlp_fwd_split": lambda a: {"n_in": a[4], "n_out": a[5], "nn": a[6], "nl": a[7]},
             "dns_mlp_bwd_split": lambda a: {"n_in": a[6], "n_out": a[7], "nn": a[8], "nl": a[9], "dx": bool(a[10]), "dw": bool(a[14])}}

    # argument index holding the number of units (points / slots / rays) a launch processes
    _UNITS_ARG = {"dns_encode_fwd": 2, "dns_encode_bwd": 2, "dns_mlp_fwd": 12, "dns_mlp_bwd": 18,
                  "dns_composite_fwd": 3, "dns_composite_bwd": 3, "dns_raygen_sample": (14, 15), "dns_raygen_bwd": (7, 8),
                  "dns_rays_from_pixels": 12, "dns_mlp_dwin": 10, "dns_feature_block": (7, 8), "dns_loss_sums": (1, 2),
                  "dns_loss_bwd": (1, 2), "dns_loss_rays": (1, 2), "dns_loss_finalize_bwd": (1, 2), "dns_raw_bwd": 2, "dns_rgb_sigmoid": 1, "dns_class_slots": (1, 2),
                  "dns_hashgrid_indices": 1, "dns_sample_along_rays": 2, "dns_feature_gather": (4, 5),
                  "dns_encode_fwd_split": 2, "dns_mlp_fwd_half": 12, "dns_mlp_bwd_half": 17, "dns_mlp_fwd_split": 10, "dns_mlp_bwd_split": 16, "dns_feature_block_split": (9, 10),
                  "dns_composite_fwd_ex": 3, "dns_composite_bwd_ex": 3, "dns_loss_bwd_points": (1, 2)}

    def arm(self, kernels=False):
        self.kernels = bool(kernels)
        if self.kernels:
            check(self._raw.dns_kernel_timing(1), "dns_kernel_timing")
        self.records = []

    def disarm(self):
        """-> {entry point: (calls, total_ms, total_units)}; call after a device synchronise.  With ``arm(kernels=True)``
        ``self.kernel_spans`` then holds one ``(entry point, kernel, ms, units, info)`` per kernel launch."""
        recs, self.records = self.records or [], None
        out = {}
        spans = []
        if self.kernels:
            check(self._raw.dns_kernel_timing(0), "dns_kernel_timing")
            buf, ms = C.create_string_buffer(160), C.c_float()
        for name, e0, e1, units, info, (k0, k1) in recs:
            c, t, u = out.get(name, (0, 0.0, 0))
            out[name] = (c + 1, t + e0.elapsed_time(e1), u + units)
            for k in range(k0, k1):
                check(self._raw.dns_kernel_timing_get(k, buf, 160, C.byref(ms)), "dns_kernel_timing_get")
                spans.append((name, buf.value.decode().strip("()"), float(ms.value), units, info))
        self.kernel_spans, self.kernels = spans, False
        return out


lib = _TimedLib(_rawlib)
timer = lib


# ----------------------------------------------------------------------------- grid meta
class GridMeta:
    """Host-side hash-grid level table (tcnn GridEncoding constructor; reference call site
    models/pos_encoding.py:31-46).  ``per_level_scale`` follows pos_encoding.py:33 in float64."""

    def __init__(self, log2_hashmap_size: int, desired_resolution: int, n_levels: int = 16,
                 n_features: int = 2, base_resolution: int = 16, per_level_scale: Optional[float] = None):
        if per_level_scale is None:
            per_level_scale = float(np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1)))
        self.c = DnsGridMeta()
        check(lib.dns_grid_meta_init(C.byref(self.c), n_levels, n_features, log2_hashmap_size, base_resolution,
                                     per_level_scale), "dns_grid_meta_init")
        self.n_levels = n_levels
        self.n_features = n_features
        self.total_rows = int(self.c.total_rows)
        self.out_dim = n_levels * n_features
        self._args = (log2_hashmap_size, desired_resolution, n_levels, n_features, base_resolution, per_level_scale)

    def levels(self):
        c = self.c
        return [dict(scale=np.float32(c.scale[l]), resolution=int(c.resolution[l]), size=int(c.size[l]),
                     offset=int(c.offset[l]), hashed=bool(c.hashed[l])) for l in range(self.n_levels)]

    # picklable (spawn / torch.save / deepcopy of modules that own one)
    def __getstate__(self):
        return {"args": self._args}

    def __setstate__(self, st):
        a = st["args"]
        self.__init__(a[0], a[1], a[2], a[3], a[4], a[5])


def _bound6(bound) -> Optional[C.Array]:
    """[3,2] float64 bound -> 6 host doubles b0x,b1x,b0y,b1y,b0z,b1z."""
    if bound is None:
        return None
    if isinstance(bound, C.Array):
        return bound
    cached = getattr(bound, "_dns_b6", None) if isinstance(bound, torch.Tensor) else None
    if cached is not None:
        return cached
    b = torch.as_tensor(bound, dtype=torch.float64).detach().cpu().reshape(3, 2)   # one sync, then cached on the tensor
    b6 = (C.c_double * 6)(*[float(v) for v in b.reshape(-1)])
    if isinstance(bound, torch.Tensor):
        try:
            bound._dns_b6 = b6
        except Exception:
            pass
    return b6


# ----------------------------------------------------------------------------- encoding
# (form, queue_cap) of the table-gradient scatter, include/dns_hip.h DNS_SCATTER_*: 0 auto (pair lists for the hashed levels, LDS
# sweep / queues for the dense ones), 1 per-corner atomics (tcnn's form), 2 LDS sweep for every level, 3 per-chunk queues for
# every multi-chunk level; queue_cap 0 = sized by the library
SCATTER_AUTO, SCATTER_ATOMIC, SCATTER_BINNED, SCATTER_QUEUES = 0, 1, 2, 3
SCATTER_REPLAY = 0x10       # include/dns_hip.h DNS_SCATTER_REPLAY: hashed levels' corner rows stored once, replayed by the chunk visits
SCATTER_LISTS = 0x20        # include/dns_hip.h DNS_SCATTER_LISTS (implied by auto): hashed levels through per-chunk lists of {point, corner pair} words
SCATTER_FORM = (SCATTER_AUTO | (SCATTER_REPLAY if os.environ.get("DNS_SCATTER_REPLAY", "0") == "1" else 0)
                | (SCATTER_LISTS if os.environ.get("DNS_SCATTER_LISTS", "0") == "1" else 0), 0)
SAVE_DY_DX = True           # encode forward keeps d(grid)/dx when the points need a gradient (False: the backward re-gathers)


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, table, meta: Optional[GridMeta], bound, n_bins: int, want_pe: bool, want_grid: bool):
        require_cuda(pts, table)
        P = pts.shape[0]
        pts = pts.contiguous().float()
        pe_dim = 3 * n_bins if want_pe else 0
        g_dim = meta.out_dim if want_grid else 0
        ld = pe_dim + g_dim
        out = torch.empty(P, ld, device=pts.device, dtype=torch.float32)
        b6 = _bound6(bound)
        x = torch.empty(P, 3, device=pts.device, dtype=torch.float32) if b6 is not None else pts
        pe_ptr = ptr(out) if want_pe else None
        grid_ptr = C.c_void_p(out.data_ptr() + 4 * pe_dim) if want_grid else None
        # the points need a gradient (poses, through pts): keep d(grid features)/dx like tcnn does -- 384 B per
        # point written here instead of a second 8-corner gather per level in the backward
        dydx = torch.empty(meta.n_levels * 3 * P * 2, device=pts.device, dtype=torch.float32) \
            if (want_grid and pts.requires_grad and SAVE_DY_DX) else None
        check(lib.dns_encode_fwd(ptr(pts), b6, P, n_bins, ptr(table) if want_grid else None,
                                 C.byref(meta.c) if want_grid else None,
                                 ptr(x) if b6 is not None else None, pe_ptr, ld, grid_ptr, ld, ptr(dydx), stream_ptr()),
              "dns_encode_fwd")
        ctx.save_for_backward(x, table if want_grid else None, dydx)
        ctx.meta, ctx.b6, ctx.n_bins, ctx.pe_dim, ctx.g_dim = meta, b6, n_bins, pe_dim, g_dim
        ctx.need_x = pts.requires_grad
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, table, dydx = ctx.saved_tensors
        P = x.shape[0]
        d_out = d_out.contiguous()
        ld = ctx.pe_dim + ctx.g_dim
        need_x = ctx.needs_input_grad[0]
        need_t = ctx.g_dim > 0 and ctx.needs_input_grad[1]
        d_x = torch.empty(P, 3, device=x.device, dtype=torch.float32) if need_x else None
        d_table = torch.zeros_like(table) if need_t else None
        d_pe = ptr(d_out) if ctx.pe_dim else None
        d_grid = C.c_void_p(d_out.data_ptr() + 4 * ctx.pe_dim) if ctx.g_dim else None
        form, cap = SCATTER_FORM
        ws = torch.empty(int(lib.dns_encode_bwd_ws_floats(P, C.byref(ctx.meta.c), form, cap)), device=x.device,
                         dtype=torch.float32) if need_t else None
        check(lib.dns_encode_bwd(ptr(x), ctx.b6, P, ctx.n_bins, ptr(table) if ctx.g_dim else None,
                                 C.byref(ctx.meta.c) if ctx.g_dim else None, d_pe, ld, d_grid, ld,
                                 ptr(d_table), ptr(d_x), ptr(dydx) if need_x else None, ptr(ws), form, cap, stream_ptr()),
              "dns_encode_bwd")
        return d_x, d_table, None, None, None, None, None


def encode(pts: torch.Tensor, table: Optional[torch.Tensor], meta: Optional[GridMeta], bound=None, n_bins: int = 16,
           want_pe: bool = True, want_grid: bool = True) -> torch.Tensor:
    """[P,3] -> [P, 3*n_bins (+) L*F] (OneBlob channels first).  With ``bound`` the input is world points and
    the fp64 normalisation of slams/mapping.py:608 happens in-kernel."""
    if table is None:
        table = pts.new_zeros(1)
    return _EncodeFn.apply(pts, table, meta, bound, n_bins, want_pe, want_grid)


def hashgrid_rows(x: torch.Tensor, meta: GridMeta) -> torch.Tensor:
    """Debug/parity: absolute table rows [P, L, 8] (int64) of the corners the kernel gathers."""
    require_cuda(x)
    x = x.contiguous().float()
    rows = torch.empty(x.shape[0], meta.n_levels, 8, device=x.device, dtype=torch.int32)
    check(lib.dns_hashgrid_indices(ptr(x), x.shape[0], C.byref(meta.c), ptr(rows), stream_ptr()), "dns_hashgrid_indices")
    return rows.to(torch.int64) & 0xFFFFFFFF


# ----------------------------------------------------------------------------- MLP
def mlp_out_padded(n_out: int) -> int:
    return (n_out + 15) // 16 * 16


def mlp_param_count(n_in: int, n_out: int, n_neurons: int, n_hidden_layers: int) -> int:
    return n_neurons * n_in + (n_hidden_layers - 1) * n_neurons * n_neurons + mlp_out_padded(n_out) * n_neurons


def _row_major_2d(x: torch.Tensor) -> torch.Tensor:
    if x.dim() != 2 or x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
        x = x.contiguous()
        if x.data_ptr() % 16 != 0:
            x = x.clone()
    return x


# The backward recomputes the hidden activations (two layers of f16 matrix instructions cost less than reading 512 B per
# point back from HBM): the forward keeps nothing.  True = dns_mlp_fwd also writes them (h_save; inspection only).
LOSS_SUMS_FLOATS = 32 + 5 * 1024          # include/dns_hip.h DNS_LOSS_SUMS_FLOATS
MLP_SAVE_HIDDEN = False
MLP_FP16_FLAG = 0x100                      # include/dns_hip.h DNS_MLP_FP16
MLP_PREPARED_FLAG = 0x200                  # include/dns_hip.h DNS_MLP_PREPARED
MLP_NO_DWIN_FLAG = 0x400                   # include/dns_hip.h DNS_MLP_NO_DWIN
MLP_DX_FIRST_FLAG = 0x800                  # include/dns_hip.h DNS_MLP_DX_FIRST
MLP_DX_FROM = lambda c: int(c) << 24       # include/dns_hip.h DNS_MLP_DX_FROM(c): no input gradient for columns [0, c)
MLP_LIVE_IN = lambda n: int(n) << 16       # include/dns_hip.h DNS_MLP_LIVE_IN(n): input columns [n, n_in) are identically zero


class _MlpFn(torch.autograd.Function):
    """y = MLP(x; params).  ``params`` is [G, count] (G weight sets, G=1 for a plain network)."""

    @staticmethod
    def forward(ctx, x, params, shape, row_index, tile_group, n_slots):
        n_in, n_out, nn, nl = shape[:4]
        fp16 = MLP_FP16_FLAG if (len(shape) > 4 and shape[4]) else 0
        require_cuda(params, row_index, tile_group)
        if not x.is_cuda:
            raise ValueError("dns_slam_amd ops run on the GPU only (got a non-CUDA tensor); there is no CPU fallback")
        x = _row_major_2d(x.float())
        P = x.shape[0]
        if row_index is None:
            y = torch.empty(P, n_out, device=x.device, dtype=torch.float32)
            n_slots = P
        else:
            y = torch.zeros(P, n_out, device=x.device, dtype=torch.float32)
        stride = params.shape[-1] if params.dim() == 2 else 0
        # keep the hidden activations when a backward will follow: it then skips the forward recompute
        keep = MLP_SAVE_HIDDEN and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        h_save = torch.empty(nl * n_slots * nn, device=x.device, dtype=torch.float32) if keep else None
        check(lib.dns_mlp_fwd(ptr(x), x.stride(0), None, 0, 0, ptr(params), n_in, n_out, nn, nl, ptr(y), n_out, n_slots,
                              ptr(row_index), ptr(tile_group), stride, ptr(h_save), fp16, stream_ptr()), "dns_mlp_fwd")
        ctx.save_for_backward(x, params, row_index, tile_group, h_save)
        ctx.shape, ctx.n_slots, ctx.stride, ctx.fp16 = shape[:4], n_slots, stride, fp16
        return y

    @staticmethod
    def backward(ctx, dy):
        x, params, row_index, tile_group, h_save = ctx.saved_tensors
        n_in, n_out, nn, nl = ctx.shape
        dy = dy.contiguous()
        P = x.shape[0]
        need_x, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if row_index is None:
            d_x = torch.empty(P, n_in, device=x.device, dtype=torch.float32) if need_x else None
        else:
            d_x = torch.zeros(P, n_in, device=x.device, dtype=torch.float32) if need_x else None
        d_p = torch.zeros_like(params) if need_p else None
        ws = torch.empty(int(lib.dns_mlp_bwd_ws_floats(ctx.n_slots, nn, nl)), device=x.device, dtype=torch.float32)
        check(lib.dns_mlp_bwd(ptr(x), x.stride(0), None, 0, 0, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl,
                              ptr(d_x), n_in, None, 0, ptr(d_p), ptr(ws), ctx.n_slots, ptr(row_index), ptr(tile_group),
                              ctx.stride, ptr(h_save), ctx.fp16, stream_ptr()), "dns_mlp_bwd")
        return d_x, d_p, None, None, None, None


class _MlpCatFn(torch.autograd.Function):
    """y = MLP(cat(x1, x2, -1); params) without building the concatenation: the kernels read the two column segments in
    place (dns_mlp_fwd/bwd two-segment input) and write the two input gradients separately."""

    @staticmethod
    def forward(ctx, x1, x2, params, shape):
        n_in, n_out, nn, nl = shape[:4]
        fp16 = MLP_FP16_FLAG if (len(shape) > 4 and shape[4]) else 0
        require_cuda(params)
        if not (x1.is_cuda and x2.is_cuda):
            raise ValueError("dns_slam_amd ops run on the GPU only (got a non-CUDA tensor); there is no CPU fallback")
        x1, x2 = _row_major_2d(x1.float()), _row_major_2d(x2.float())
        P, n1 = x1.shape
        y = torch.empty(P, n_out, device=x1.device, dtype=torch.float32)
        h_save = torch.empty(nl * P * nn, device=x1.device, dtype=torch.float32) if MLP_SAVE_HIDDEN else None
        check(lib.dns_mlp_fwd(ptr(x1), x1.stride(0), ptr(x2), x2.stride(0), n1, ptr(params), n_in, n_out, nn, nl, ptr(y),
                              n_out, P, None, None, 0, ptr(h_save), fp16, stream_ptr()), "dns_mlp_fwd")
        ctx.save_for_backward(x1, x2, params, h_save)
        ctx.shape, ctx.fp16 = shape[:4], fp16
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, params, h_save = ctx.saved_tensors
        n_in, n_out, nn, nl = ctx.shape
        dy = dy.contiguous()
        P, n1 = x1.shape
        need_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        d1 = torch.empty(P, n1, device=x1.device, dtype=torch.float32) if need_x else None
        d2 = torch.empty(P, n_in - n1, device=x1.device, dtype=torch.float32) if need_x else None
        d_p = torch.zeros_like(params) if ctx.needs_input_grad[2] else None
        ws = torch.empty(int(lib.dns_mlp_bwd_ws_floats(P, nn, nl)), device=x1.device, dtype=torch.float32)
        check(lib.dns_mlp_bwd(ptr(x1), x1.stride(0), ptr(x2), x2.stride(0), n1, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl,
                              ptr(d1), n1, ptr(d2), n_in - n1, ptr(d_p), ptr(ws), P, None, None, 0, ptr(h_save), ctx.fp16,
                              stream_ptr()), "dns_mlp_bwd")
        return (d1 if ctx.needs_input_grad[0] else None), (d2 if ctx.needs_input_grad[1] else None), d_p, None


def mlp_cat(x1: torch.Tensor, x2: torch.Tensor, params: torch.Tensor, n_out: int, n_neurons: int = 32,
            n_hidden_layers: int = 1, fp16: bool = False) -> torch.Tensor:
    """``mlp(torch.cat((x1, x2), -1), ...)`` with the concatenation left implicit (models/decoder.py:73,123-124).  Falls
    back to the explicit cat when a segment is not a multiple of 4 columns / 16-byte addressable or wider than 64."""
    n1, n2 = x1.shape[-1], x2.shape[-1]
    ok = (n1 % 4 == 0 and n2 % 4 == 0 and 0 < n1 <= 64 and 0 < n2 <= 64 and x1.dim() == 2 and x2.dim() == 2
          and x1.stride(-1) == 1 and x2.stride(-1) == 1 and x1.stride(0) % 4 == 0 and x2.stride(0) % 4 == 0
          and x1.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0 and x1.dtype == torch.float32 and x2.dtype == torch.float32)
    if not ok:
        return mlp(torch.cat((x1, x2), -1), params, n1 + n2, n_out, n_neurons, n_hidden_layers, fp16)
    return _MlpCatFn.apply(x1, x2, params, (n1 + n2, n_out, n_neurons, n_hidden_layers, bool(fp16)))


def mlp(x: torch.Tensor, params: torch.Tensor, n_in: int, n_out: int, n_neurons: int = 32,
        n_hidden_layers: int = 1, fp16: bool = False) -> torch.Tensor:
    """Bias-free ReLU MLP on the matrix cores (tcnn CutlassMLP replacement): exact fp32, or with ``fp16`` tcnn's own
    precision (fp16 operands, fp32 accumulate; parameters, kept activations and weight gradients stay fp32)."""
    return _MlpFn.apply(x, params, (n_in, n_out, n_neurons, n_hidden_layers, bool(fp16)), None, None, 0)


# ---- half rows (include/dns_hip.h, ABI v12): tcnn's own arithmetic -- f16 activations and weight operands, fp32 accumulation, a
# static loss scale.  Thin, non-autograd wrappers (fused_step.MapStep drives the entry points directly; tests use these).
HALF_LOSS_SCALE = 128.0                       # tcnn's default loss_scale for half-precision networks
SPLIT_PLAIN = 2                               # DNS_SPLIT_PLAIN: dns_encode_fwd_split / dns_feature_block_split write plain f16 rows


def _cuda_rows(*tensors):
    """CUDA tensors with unit column stride (row-strided views are what these entry points take: ld arguments)."""
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise ValueError("dns_slam_amd ops run on the GPU only (got a non-CUDA tensor); there is no CPU fallback")
        if t.dim() == 2 and t.stride(1) != 1:
            raise ValueError("dns_slam_amd ops need unit column stride")


def _half_rows_ok(t):
    return t is None or (t.is_cuda and t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1)


def mlp_fwd_half(x, params, n_in, n_out, n_neurons, n_hidden_layers, x2=None, row_index=None, tile_group=None, n_slots=None,
                 param_stride=0, live_in=0, out=None):
    """y = MLP(x | x2) on f16 rows (dns_mlp_fwd_half): x [rows, >= n_in1] / x2 [rows, >= n_in - n_in1] float16, params fp32 (a pool
    [G, stride] with tile_group), y fp32 [rows, n_out].  ``live_in``: DNS_MLP_LIVE_IN width (0 = n_in)."""
    _cuda_rows(x, x2, out)
    require_cuda(params)
    if not (_half_rows_ok(x) and _half_rows_ok(x2)):
        raise ValueError("mlp_fwd_half: x / x2 must be 2-D float16 CUDA tensors with unit column stride")
    rows = x.shape[0]
    n_slots = rows if n_slots is None else n_slots
    y = out if out is not None else torch.zeros(rows, n_out, device=x.device, dtype=torch.float32)
    n_in1 = x.shape[1] if x2 is not None else 0
    check(lib.dns_mlp_fwd_half(ptr(x), x.stride(0), ptr(x2), 0 if x2 is None else x2.stride(0), n_in1, ptr(params), n_in, n_out,
                               n_neurons, n_hidden_layers, ptr(y), y.stride(0), n_slots, ptr(row_index), ptr(tile_group), param_stride,
                               MLP_LIVE_IN(live_in) if live_in else 0, stream_ptr()), "dns_mlp_fwd_half")
    return y


def mlp_bwd_half(x, dy, params, n_in, n_out, n_neurons, n_hidden_layers, x2=None, d_x=None, d_x2=None, d_params=None, row_index=None,
                 tile_group=None, n_slots=None, param_stride=0, accumulate=0, loss_scale=HALF_LOSS_SCALE):
    """Input and weight gradients of the same network (dns_mlp_bwd_half; ONE kernel, dW_in included).  dy fp32 [rows, n_out];
    d_x / d_x2 fp32 (None = skip), d_params fp32 like params (+=; None = skip).  ``accumulate``: the entry point's accumulate_dx
    word (bits 0 / 1, MLP_DX_FIRST_FLAG, MLP_LIVE_IN(n), MLP_DX_FROM(c))."""
    _cuda_rows(x, x2, dy, d_x, d_x2)
    require_cuda(params, d_params)
    if not (_half_rows_ok(x) and _half_rows_ok(x2)):
        raise ValueError("mlp_bwd_half: x / x2 must be 2-D float16 CUDA tensors with unit column stride")
    n_slots = x.shape[0] if n_slots is None else n_slots
    n_in1 = x.shape[1] if x2 is not None else 0
    check(lib.dns_mlp_bwd_half(ptr(x), x.stride(0), ptr(x2), 0 if x2 is None else x2.stride(0), n_in1, ptr(dy), dy.stride(0),
                               ptr(params), n_in, n_out, n_neurons, n_hidden_layers, ptr(d_x), 0 if d_x is None else d_x.stride(0),
                               ptr(d_x2), 0 if d_x2 is None else d_x2.stride(0), ptr(d_params), n_slots, ptr(row_index),
                               ptr(tile_group), param_stride, int(accumulate), float(loss_scale), stream_ptr()), "dns_mlp_bwd_half")


def group_slots(slot_of_point: torch.Tensor, n_groups: int, min_count: int = 2):
    """Device-side (sync-free) layout for the grouped MLP: points counting-sorted by weight-set id into 128-slot tiles
    (dns_group_slots).  ``slot_of_point`` [P] int64 in [0, n_groups) (negative = no network).  Returns (row_index
    int32 [n_slots], tile_group int32 [n_slots/128], n_slots).  Groups with fewer than ``min_count`` points are
    skipped, as ``Mapper.fine_fn`` does (slams/mapping.py:597: ``if index.sum() > 1``)."""
    require_cuda(slot_of_point)
    slot_of_point = slot_of_point.contiguous()
    if slot_of_point.dtype != torch.int64:
        slot_of_point = slot_of_point.to(torch.int64)
    P = slot_of_point.shape[0]
    dev = slot_of_point.device
    n_slots = (P + 127) // 128 * 128 + 128 * n_groups
    row_index = torch.empty(n_slots, device=dev, dtype=torch.int32)
    tile_group = torch.empty(n_slots // 128, device=dev, dtype=torch.int32)
    ws = torch.empty(512, device=dev, dtype=torch.int32)
    check(lib.dns_group_slots(ptr(slot_of_point), P, n_groups, min_count, n_slots, ptr(ws), ptr(row_index), ptr(tile_group),
                              stream_ptr()), "dns_group_slots")
    return row_index, tile_group, n_slots


def mlp_grouped(x: torch.Tensor, params_pool: torch.Tensor, slot_of_point: torch.Tensor, n_in: int, n_out: int,
                n_neurons: int = 32, n_hidden_layers: int = 1, min_count: int = 2, fp16: bool = False) -> torch.Tensor:
    """Per-point weight sets (the per-class fine decoders, slams/mapping.py:590-601): point p runs through
    ``params_pool[slot_of_point[p]]``; points with no network / tiny groups get zeros."""
    G = params_pool.shape[0]
    row_index, tile_group, n_slots = group_slots(slot_of_point, G, min_count)
    return _MlpFn.apply(x, params_pool, (n_in, n_out, n_neurons, n_hidden_layers, bool(fp16)), row_index, tile_group, n_slots)


# ----------------------------------------------------------------------------- the renderer's four networks, fused glue
class _RenderNetsFn(torch.autograd.Function):
    """Coarse + per-class fine + colour + logit networks of ``Mapper.renderer`` (slams/mapping.py:616-626) as ONE
    autograd node.  Same kernels as ``mlp`` / ``mlp_grouped``; what it removes is glue traffic:
      * the colour / logit input ``cat(pe, cat(fine[:, 1:], pixel))`` (models/decoder.py:123-124) is never built: the
        networks read the pe columns of ``buf`` and the [P, 64] feature block as a two-segment input;
      * in the backward every network adds its input gradient in place (accumulate_dx) into one d_buf / d_feat pair
        instead of autograd materialising one [P, 80] / [P, 112] gradient per consumer and summing them;
      * the compositing input ``cat(sigmoid(colour), fine[:, 0:1])`` (slams/mapping.py:622-627) is an output of the node
        (the colour network writes into its rows), so neither the cat nor the [P, h+1] zero-padded gradient of the
        ``fine[:, 0:1]`` slice and its sum with the losses' gradient exist.
    buf: [P, pe_dim + grid_dim] (OneBlob | hash grid), pixel: [P, C] 2-D feature code."""

    @staticmethod
    def forward(ctx, buf, pixel, coarse_p, fine_pool, color_p, logit_p, slot_of_point, cfg):
        pe_dim, shp_c, shp_f, shp_col, shp_log, min_count, fp16, n_groups = cfg[:8]
        need_coarse = cfg[8] if len(cfg) > 8 else True
        fp16 = MLP_FP16_FLAG if fp16 else 0
        require_cuda(buf, pixel, coarse_p, fine_pool, color_p, logit_p, slot_of_point)
        buf = _row_major_2d(buf.float())
        pixel = pixel.float()
        P, dev = buf.shape[0], buf.device
        keep = MLP_SAVE_HIDDEN
        st = stream_ptr()

        def run(x, x2, n_in1, params, shape, y, ri, tg, n_slots, stride, live=0):
            n_in, n_out, nn, nl = shape
            h = torch.empty(nl * n_slots * nn, device=dev, dtype=torch.float32) if keep else None
            check(lib.dns_mlp_fwd(ptr(x), x.stride(0), ptr(x2), 0 if x2 is None else x2.stride(0), n_in1, ptr(params),
                                  n_in, n_out, nn, nl, ptr(y), y.stride(0), n_slots, ptr(ri), ptr(tg), stride, ptr(h), fp16 | live, st),
                  "dns_mlp_fwd")
            return h

        # (need_coarse = False: the forward-only frame render, slams/mapping.py:638-694 -- the coarse latents feed the latent
        #  loss alone, mapping.py:893-896, and frame_vis drops them: one of the four networks is not run)
        coarse = torch.empty(P if need_coarse else 0, shp_c[1], device=dev, dtype=torch.float32)
        h_c = run(buf, None, 0, coarse_p, shp_c, coarse, None, None, P, 0) if need_coarse else None
        ri, tg, n_slots = group_slots(slot_of_point, n_groups or fine_pool.shape[0], min_count)
        fine = torch.zeros(P, shp_f[1], device=dev, dtype=torch.float32)
        h_f = run(buf, None, 0, fine_pool, shp_f, fine, ri, tg, n_slots, fine_pool.shape[-1])
        if pixel.shape[1] == 0 and not torch.is_grad_enabled():
            # forward-only, no 2-D code: the colour / logit networks read the latent straight out of the fine rows (second input
            # segment = fine[:, 1:], row stride hidden + 1: dns_mlp_fwd takes a 4-byte aligned slice) -- no [P, hidden] copy
            # (537 MB written and read again per 65 536-ray chunk of a frame render)
            feat = fine[:, 1:]
        else:
            feat = torch.cat((fine[:, 1:], pixel), -1)                    # [P, hidden + C]
        # A code of ZERO columns (forward-only callers without a 2-D code: the reference multiplies a zero code through,
        # slams/mapping.py:553-557): the colour / logit networks run as their live (pe + hidden)-input networks, DNS_MLP_LIVE_IN
        live = 0
        if pe_dim + feat.shape[1] < shp_col[0]:
            if torch.is_grad_enabled() or (pe_dim + feat.shape[1]) % 8 != 0:
                raise ValueError("render_nets: a code narrower than the networks' input is a forward-only form (no_grad, multiple of 8)")
            live = MLP_LIVE_IN(pe_dim + feat.shape[1])
        # raw = (sigmoid(colour) | occupancy) = the compositing kernel's input (slams/mapping.py:622-627): the colour
        # network writes its three columns straight into the [P, 4] rows
        raw = torch.empty(P, 4, device=dev, dtype=torch.float32)
        logit = torch.empty(P, shp_log[1], device=dev, dtype=torch.float32)
        h_col = run(buf, feat, pe_dim, color_p, shp_col, raw, None, None, P, 0, live)
        h_log = run(buf, feat, pe_dim, logit_p, shp_log, logit, None, None, P, 0, live)
        raw.sigmoid_()                                                    # column 3 is not written yet
        raw[:, 3] = fine[:, 0]
        ctx.save_for_backward(buf, feat, coarse_p, fine_pool, color_p, logit_p, ri, tg, h_c, h_f, h_col, h_log, raw)
        ctx.cfg, ctx.n_slots, ctx.pixel_dim = cfg, n_slots, pixel.shape[1]
        ctx.set_materialize_grads(False)
        return coarse, fine, raw, logit

    @staticmethod
    def backward(ctx, d_coarse, d_fine, d_raw, d_logit):
        buf, feat, coarse_p, fine_pool, color_p, logit_p, ri, tg, h_c, h_f, h_col, h_log, raw = ctx.saved_tensors
        pe_dim, shp_c, shp_f, shp_col, shp_log, _, fp16, _ = ctx.cfg[:8]
        if len(ctx.cfg) > 8 and not ctx.cfg[8]:
            raise RuntimeError("render_nets(need_coarse=False) is forward-only")
        fp16 = MLP_FP16_FLAG if fp16 else 0
        P, dev = buf.shape[0], buf.device
        st = stream_ptr()
        need_buf = ctx.needs_input_grad[0]
        need_pix = ctx.needs_input_grad[1]
        d_buf = torch.empty_like(buf)
        # [P, 4 + hidden + C]: column 3 = d occupancy (from raw), columns 4.. = the feature-block gradient of the colour
        # and logit networks, so that columns 3..3+hidden are the fine network's output gradient in ONE strided view
        n_f = shp_f[1]
        d_featx = torch.empty(P, 4 + feat.shape[1], device=dev, dtype=torch.float32)
        d_feat = d_featx[:, 4:]

        # one zero-fill for the four parameter gradients (they are views of one buffer)
        need = [ctx.needs_input_grad[i] for i in (2, 3, 4, 5)]
        sizes = [p.numel() if n else 0 for p, n in zip((coarse_p, fine_pool, color_p, logit_p), need)]
        flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        offs = [sum(sizes[:i]) for i in range(4)]
        dps = {id(p): (flat[o:o + z].view_as(p) if n else None)
               for p, o, z, n in zip((coarse_p, fine_pool, color_p, logit_p), offs, sizes, need)}

        def run(x, x2, n_in1, dy, params, shape, d_x, d_x2, need_p, ri_, tg_, n_slots, stride, h, acc):
            n_in, n_out, nn, nl = shape
            d_p = dps[id(params)] if need_p else None
            ws = torch.empty(int(lib.dns_mlp_bwd_ws_floats(n_slots, nn, nl)), device=dev, dtype=torch.float32)
            check(lib.dns_mlp_bwd(ptr(x), x.stride(0), ptr(x2), 0 if x2 is None else x2.stride(0), n_in1,
                                  ptr(dy), dy.stride(0), ptr(params), n_in, n_out, nn, nl,
                                  ptr(d_x), d_x.stride(0), ptr(d_x2), 0 if d_x2 is None else d_x2.stride(0),
                                  ptr(d_p), ptr(ws), n_slots, ptr(ri_), ptr(tg_), stride, ptr(h), acc | fp16, st), "dns_mlp_bwd")
            return d_p

        def grad(d, like_cols):
            return torch.zeros(P, like_cols, device=dev) if d is None else _row_major_2d(d.float())

        # 1. coarse: writes all columns of d_buf
        d_cp = run(buf, None, 0, grad(d_coarse, shp_c[1]), coarse_p, shp_c, d_buf, None, ctx.needs_input_grad[2],
                   None, None, P, 0, h_c, 0)
        # 2./3. colour then logit: pe columns of d_buf (+=), feature block d_feat (=, then +=).  The colour network's
        # output gradient is the sigmoid's: d_raw * s * (1 - s) on the whole [P, 4] rows (column 3 is not read, lddy 4)
        if d_raw is None:
            d_col = torch.zeros(P, 4, device=dev)
            d_featx[:, 3] = 0
        else:
            d_raw = d_raw.contiguous().float()
            d_col = torch.ops.aten.sigmoid_backward(d_raw, raw)
            d_featx[:, 3] = d_raw[:, 3]
        d_colp = run(buf, feat, pe_dim, d_col, color_p, shp_col, d_buf, d_feat,
                     ctx.needs_input_grad[4], None, None, P, 0, h_col, 1)
        d_logp = run(buf, feat, pe_dim, grad(d_logit, shp_log[1]), logit_p, shp_log, d_buf, d_feat,
                     ctx.needs_input_grad[5], None, None, P, 0, h_log, 3)
        # 4. fine: its output gradient = the caller's + d occupancy + what colour / logit sent back through the features
        d_ft = d_featx[:, 3:3 + n_f]
        if d_fine is not None:
            d_ft = d_fine.float() + d_ft
        d_fp = run(buf, None, 0, d_ft, fine_pool, shp_f, d_buf, None, ctx.needs_input_grad[3], ri, tg, ctx.n_slots,
                   fine_pool.shape[-1], h_f, 1)
        d_pix = d_feat[:, n_f - 1:] if need_pix else None
        return (d_buf if need_buf else None), d_pix, d_cp, d_fp, d_colp, d_logp, None, None


def render_nets(buf, pixel, coarse_p, fine_pool, color_p, logit_p, slot_of_point, pe_dim, shp_coarse, shp_fine,
                shp_color, shp_logit, min_count=2, fp16=False, n_groups=None, need_coarse=True):
    """-> (coarse [P, h+1], fine [P, h+1], raw [P, 4] = (sigmoid(colour), fine[:, 0]) -- the compositing input --,
    logits [P, n_class]); shapes are (n_in, n_out, n_neurons, n_hidden_layers) tuples (colour n_out = 3).  n_groups: only
    the first n_groups weight sets of fine_pool are in use (slot_of_point < n_groups) -- pass the whole pool rather than
    a slice of it and autograd has no [capacity, n_params] slice gradient to zero-fill and copy into."""
    cfg = (int(pe_dim), tuple(shp_coarse), tuple(shp_fine), tuple(shp_color), tuple(shp_logit), int(min_count), bool(fp16),
           None if n_groups is None else int(n_groups), bool(need_coarse))
    return _RenderNetsFn.apply(buf, pixel, coarse_p, fine_pool, color_p, logit_p, slot_of_point, cfg)


# ----------------------------------------------------------------------------- compositing
class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z, logits):
        require_cuda(raw, z, logits)
        raw = raw.contiguous().float()
        z = z.contiguous().float()
        N, S = z.shape
        Cn = 0 if logits is None else logits.shape[-1]
        if logits is not None:
            logits = logits.contiguous().float()
        dev = raw.device
        depth = torch.empty(N, device=dev)
        var = torch.empty(N, device=dev)
        rgb = torch.empty(N, 3, device=dev)
        weights = torch.empty(N, S, device=dev)
        sem = torch.empty(N, Cn, device=dev) if Cn else None
        check(lib.dns_composite_fwd(ptr(raw), ptr(z), ptr(logits), N, S, Cn, ptr(depth), ptr(var), ptr(rgb),
                                    ptr(weights), ptr(sem), stream_ptr()), "dns_composite_fwd")
        ctx.save_for_backward(raw, z, logits)
        ctx.dims = (N, S, Cn)
        ctx.set_materialize_grads(False)                  # unused outputs (var, weights, ...) arrive as None, not zeros
        if sem is None:
            sem = raw.new_zeros(N, 0)
        return depth, var, rgb, weights, sem

    @staticmethod
    def backward(ctx, d_depth, d_var, d_rgb, d_weights, d_sem):
        raw, z, logits = ctx.saved_tensors
        N, S, Cn = ctx.dims
        d_raw = torch.empty_like(raw)
        d_logits = torch.empty_like(logits) if Cn else None
        c = lambda t: None if t is None else t.contiguous()
        check(lib.dns_composite_bwd(ptr(raw), ptr(z), ptr(logits), N, S, Cn, ptr(c(d_depth)), ptr(c(d_var)),
                                    ptr(c(d_rgb)), ptr(c(d_weights)), ptr(c(d_sem)) if Cn else None, ptr(d_raw),
                                    ptr(d_logits), stream_ptr()), "dns_composite_bwd")
        return d_raw, None, d_logits


def composite(raw: torch.Tensor, z_vals: torch.Tensor, logits: Optional[torch.Tensor] = None):
    """raw [N,S,4], z [N,S], logits [N,S,C] -> depth [N], var [N], rgb [N,3], weights [N,S], sem [N,C]."""
    return _CompositeFn.apply(raw, z_vals, logits)


# ----------------------------------------------------------------------------- ray generation + sampling
class _RaygenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, quat, trans, pix_idx, color, depth, label, cam, bound, window, npf, t_uniform, t_surf, t_zero,
                depth_max=None):
        require_cuda(quat, trans, pix_idx, color, depth, label, t_uniform, t_surf, t_zero)
        K, H, W = depth.shape
        H0, H1, W0, W1 = window
        n = K * npf
        nu = 0 if t_uniform is None else t_uniform.numel()
        ns = t_surf.shape[-1]
        if t_surf.dim() == 2 and (t_surf.shape != (K, ns) or t_zero.shape != (K, ns)):
            raise ValueError("per-frame jitter must be [K, n_surface] for both draws")
        jstride = ns if t_surf.dim() == 2 else 0               # [K, ns]: frame f samples with its own jitter rows
        S = nu + ns
        dev = quat.device
        quat = quat.contiguous().float()
        trans = trans.contiguous().float()
        rays_o = torch.empty(n, 3, device=dev)
        rays_d = torch.empty(n, 3, device=dev)
        gt_color = torch.empty(n, 3, device=dev)
        gt_depth = torch.empty(n, device=dev)
        gt_label = torch.empty(n, device=dev, dtype=torch.int64)
        inside = torch.empty(n, device=dev, dtype=torch.uint8)
        z = torch.empty(n, S, device=dev)
        pts = torch.empty(n, S, 3, device=dev)
        if depth_max is not None:                         # per-frame max sampled depth supplied (multi-GPU: global max)
            ws = depth_max.detach().float().contiguous().view(torch.int32)
        else:
            ws = torch.empty(K, device=dev, dtype=torch.int32)
        camv = (C.c_double * 4)(*[float(v) for v in cam])
        b6 = _bound6(bound)
        check(lib.dns_raygen_sample(ptr(pix_idx), ptr(color), ptr(depth), ptr(label), ptr(quat), ptr(trans), camv, b6,
                                    H, W, H0, H1, W0, W1, K, npf, ptr(t_uniform), ptr(t_surf), ptr(t_zero), nu, ns, jstride,
                                    ptr(ws), 0 if depth_max is None else 1, ptr(rays_o), ptr(rays_d), ptr(gt_color),
                                    ptr(gt_depth), ptr(gt_label),
                                    ptr(inside), ptr(z), ptr(pts), stream_ptr()), "dns_raygen_sample")
        ctx.save_for_backward(pix_idx, quat, z)
        ctx.misc = (camv, window, K, npf, S)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(gt_color, gt_depth, gt_label, inside, z)
        return rays_o, rays_d, pts, gt_color, gt_depth, gt_label, inside, z

    @staticmethod
    def backward(ctx, d_ro, d_rd, d_pts, *_):
        pix_idx, quat, z = ctx.saved_tensors
        camv, (H0, H1, W0, W1), K, npf, S = ctx.misc
        dev = quat.device
        d_quat = torch.zeros(K, 4, device=dev)
        d_trans = torch.zeros(K, 3, device=dev)
        ws = torch.empty(max(int(lib.dns_raygen_bwd_ws_floats(K, npf)), 1), device=dev)
        c = lambda t: None if t is None else t.contiguous()
        check(lib.dns_raygen_bwd(ptr(pix_idx), ptr(quat), camv, H0, H1, W0, W1, K, npf, S, ptr(z), ptr(c(d_pts)),
                                 ptr(c(d_ro)), ptr(c(d_rd)), ptr(ws), ptr(d_quat), ptr(d_trans), stream_ptr()),
              "dns_raygen_bwd")
        return (d_quat, d_trans) + (None,) * 12


def raygen_sample(quat, trans, pix_idx, color, depth, label, cam, bound, window, n_per_frame, t_uniform, t_surf, t_zero,
                  depth_max=None):
    """K stacked frames -> (rays_o, rays_d, pts, gt_color, gt_depth, gt_label, inside, z).  quat [K,4], trans [K,3],
    pix_idx [K*n_per_frame] int64 window-flat indices, color [K,H,W,3], depth/label [K,H,W] fp32,
    cam=(fx,fy,cx,cy), bound [3,2] fp64, window=(H0,H1,W0,W1)."""
    return _RaygenFn.apply(quat, trans, pix_idx, color, depth, label, tuple(cam), bound, tuple(window), n_per_frame,
                           t_uniform, t_surf, t_zero, depth_max)


class _RaysFromPixelsFn(torch.autograd.Function):
    """rays_o, rays_d, sample rows for given pixel indices and a rotation MATRIX (get_rays_from_uv, utils/common.py:248-264).
    The backward (dL/dR = sum_n dL/dd_n (x) dir_n, dL/dT = sum_n dL/do_n: SURVEY A1) is two [n,3]-sized reductions; they are
    plain torch on purpose -- this node only serves callers of the reference's free functions (Mesher, eval_2d: no
    gradient); the optimise steps differentiate the pose through dns_raygen_bwd."""

    @staticmethod
    def forward(ctx, R, T, pix_idx, image, cam, hw, window):
        require_cuda(R, T, pix_idx, image)
        H, W = hw
        H0, H1, W0, W1 = window
        dev = R.device
        n = (H1 - H0) * (W1 - W0) if pix_idx is None else pix_idx.numel()
        Rc, Tc = R.detach().contiguous().float(), T.detach().contiguous().float()
        Cn = 0 if image is None else image.shape[-1]
        rays_o = torch.empty(n, 3, device=dev)
        rays_d = torch.empty(n, 3, device=dev)
        sample = torch.empty(n, Cn, device=dev) if Cn else None
        ij = torch.empty(n, 2, device=dev)
        camv = (C.c_double * 4)(*[float(v) for v in cam])
        check(lib.dns_rays_from_pixels(ptr(pix_idx), ptr(image), Cn, ptr(Rc), ptr(Tc), camv, H, W, H0, H1, W0, W1, n,
                                       ptr(rays_o), ptr(rays_d), ptr(sample), ptr(ij), stream_ptr()), "dns_rays_from_pixels")
        ctx.save_for_backward(ij)
        ctx.cam = tuple(float(v) for v in cam)
        ctx.mark_non_differentiable(ij)
        if sample is None:
            sample = rays_o.new_zeros(n, 0)
        ctx.mark_non_differentiable(sample)
        ctx.set_materialize_grads(False)
        return rays_o, rays_d, sample, ij

    @staticmethod
    def backward(ctx, d_ro, d_rd, _ds, _dij):
        ij, = ctx.saved_tensors
        fx, fy, cx, cy = ctx.cam
        d_R = d_T = None
        if d_rd is not None and ctx.needs_input_grad[0]:
            dirs = torch.stack(((ij[:, 0] - cx) / fx, -(ij[:, 1] - cy) / fy, -torch.ones_like(ij[:, 0])), -1)
            d_R = d_rd.t() @ dirs
        if d_ro is not None and ctx.needs_input_grad[1]:
            d_T = d_ro.sum(0)
        return d_R, d_T, None, None, None, None, None


def rays_from_pixels(R, T, pix_idx, image, cam, hw, window):
    """R [3,3], T [3] (device), pix_idx [n] int64 window-flat (None = every pixel of the window in row-major order),
    image [H,W,C] or None, cam=(fx,fy,cx,cy), hw=(H,W), window=(H0,H1,W0,W1) -> rays_o, rays_d [n,3], sample [n,C], ij [n,2]."""
    if image is not None:
        image = image.contiguous().float()
    if pix_idx is not None:
        pix_idx = pix_idx.contiguous()
    return _RaysFromPixelsFn.apply(R.contiguous(), T.contiguous(), pix_idx, image, tuple(cam), tuple(hw), tuple(window))


def sample_along_rays(gt_depth, far_bb, t_uniform, t_surf, t_zero):
    """Stand-alone depth-guided sampling (utils/common.py:561-599) with explicit jitter: gt_depth [n] fp32,
    far_bb [n] or [n,1] fp64 (+0.01 already applied) -> z [n, S] ascending fp32."""
    require_cuda(gt_depth, far_bb, t_uniform, t_surf, t_zero)
    gt_depth = gt_depth.reshape(-1).contiguous().float()
    far_bb = far_bb.reshape(-1).contiguous().double()
    n = gt_depth.numel()
    nu = 0 if t_uniform is None else t_uniform.numel()
    ns = t_surf.numel()
    z = torch.empty(n, nu + ns, device=gt_depth.device)
    ws = torch.empty(1, device=gt_depth.device, dtype=torch.int32)
    check(lib.dns_sample_along_rays(ptr(gt_depth), ptr(far_bb), n, ptr(t_uniform), ptr(t_surf), ptr(t_zero), nu, ns,
                                    ptr(ws), ptr(z), stream_ptr()), "dns_sample_along_rays")
    return z


# ----------------------------------------------------------------------------- fused losses
class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_color, pred_depth, pred_var, pred_logits, fine, coarse, gt_color, gt_depth, gt_label, valid, z,
                lambdas, tracker, reduce_sums):
        require_cuda(pred_color, pred_depth, pred_var, pred_logits, fine, coarse, gt_color, gt_depth, gt_label, valid, z)
        c = lambda t: None if t is None else t.contiguous()
        pred_color, pred_depth, pred_var, pred_logits = c(pred_color), c(pred_depth), c(pred_var), c(pred_logits)
        fine, coarse, gt_color, gt_depth, gt_label, z = c(fine), c(coarse), c(gt_color), c(gt_depth), c(gt_label), c(z)
        N = pred_depth.shape[0]
        Cn = 0 if pred_logits is None else pred_logits.shape[-1]
        S = 1 if z is None else z.shape[1]
        L = 1 if fine is None else fine.shape[-1]
        dev = pred_depth.device
        lam = (C.c_float * 8)(*[float(v) for v in lambdas])
        sums_ws = torch.empty(LOSS_SUMS_FLOATS, device=dev)     # [0:16] results, the rest reduction workspace
        sums = sums_ws[:16]
        out = torch.empty(16, device=dev)
        check(lib.dns_loss_sums(lam, N, S, Cn, L, int(tracker), ptr(pred_color), ptr(pred_depth), ptr(pred_var),
                                ptr(pred_logits), ptr(gt_color), ptr(gt_depth), ptr(gt_label), ptr(valid), ptr(fine),
                                ptr(coarse), ptr(z), ptr(sums_ws), stream_ptr()), "dns_loss_sums")
        if reduce_sums is not None:
            reduce_sums(sums)                      # multi-GPU: global numerators / counts (dns_slam_amd.dist)
        check(lib.dns_loss_finalize(lam, N, S, Cn, L, int(tracker), ptr(sums), ptr(out), stream_ptr()), "dns_loss_finalize")
        ctx.save_for_backward(pred_color, pred_depth, pred_var, pred_logits, fine, coarse, gt_color, gt_depth, gt_label,
                              valid, z, out)
        ctx.misc = (lam, N, S, Cn, L, int(tracker))
        total = out[6]
        terms = out[:6]
        ctx.mark_non_differentiable(terms)
        return total, terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        (pred_color, pred_depth, pred_var, pred_logits, fine, coarse, gt_color, gt_depth, gt_label, valid, z,
         out) = ctx.saved_tensors
        lam, N, S, Cn, L, tracker = ctx.misc
        dev = pred_depth.device
        g = g_total.reshape(1).contiguous().float()
        d_color = torch.empty_like(pred_color)
        d_depth = torch.empty_like(pred_depth)
        d_var = torch.empty_like(pred_var) if pred_var is not None else None
        d_logits = torch.empty_like(pred_logits) if Cn else None
        d_fine = torch.empty_like(fine) if fine is not None else None
        d_coarse = torch.empty_like(coarse) if coarse is not None else None
        check(lib.dns_loss_bwd(lam, N, S, Cn, L, tracker, ptr(out), ptr(g), ptr(pred_color), ptr(pred_depth), ptr(pred_var),
                               ptr(pred_logits), ptr(gt_color), ptr(gt_depth), ptr(gt_label), ptr(valid), ptr(fine),
                               ptr(coarse), ptr(z), ptr(d_color), ptr(d_depth), ptr(d_var), ptr(d_logits), ptr(d_fine),
                               ptr(d_coarse), 0, stream_ptr()), "dns_loss_bwd")
        return (d_color, d_depth, d_var, d_logits, d_fine, d_coarse) + (None,) * 8


def mapping_losses(pred_color, pred_depth, pred_logits, fine, coarse, gt_color, gt_depth, gt_label, z_vals, lambdas,
                   valid=None, reduce_sums=None):
    """The six ray-batch loss terms of one mapping iteration, fused (slams/mapping.py:887-907 minus smoothness).
    lambdas = (lambda_p, lambda_d, lambda_l, lambda_lt, lambda_fs, lambda_opacity, truncation, sigma).
    -> (weighted total [scalar, differentiable], terms [6] = p, d, l, lt, fs, opacity)."""
    return _LossFn.apply(pred_color, pred_depth, None, pred_logits, fine, coarse, gt_color, gt_depth, gt_label, valid,
                         z_vals, lambdas, False, reduce_sums)


def tracking_losses(pred_color, pred_depth, pred_var, pred_logits, gt_color, gt_depth, gt_label, mask, lambdas):
    """The three masked tracking losses, fused (slams/tracking.py:85-96, 326-329). mask: bool / uint8 [N]."""
    valid = mask.to(torch.uint8).contiguous()
    lam = tuple(lambdas) + (0.0,) * (8 - len(lambdas))
    lam = (lam[0], lam[1], lam[2], 0.0, 0.0, 0.0, 0.0, 1.0)
    return _LossFn.apply(pred_color, pred_depth, pred_var, pred_logits, None, None, gt_color, gt_depth, gt_label, valid,
                         None, lam, True, None)



# ----------------------------------------------------------------------------- smoothness (TV)
class _TvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lat, n, sample_points, nx, halo):
        require_cuda(lat)
        lat = lat.contiguous().float()
        if lat.shape[0] != nx * n * n:
            raise ValueError(f"tv_smoothness: {lat.shape[0]} lattice rows for a {nx} x {n} x {n} slab")
        out = torch.empty(1, device=lat.device)
        check(lib.dns_tv_fwd(ptr(lat), lat.shape[1], nx, n, int(halo), sample_points, ptr(out), stream_ptr()), "dns_tv_fwd")
        ctx.save_for_backward(lat)
        ctx.misc = (n, sample_points, nx, int(halo))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lat, = ctx.saved_tensors
        n, sp, nx, halo = ctx.misc
        d = torch.empty_like(lat)
        check(lib.dns_tv_bwd(ptr(lat), lat.shape[1], nx, n, halo, sp, ptr(g.reshape(1).contiguous().float()), ptr(d), stream_ptr()),
              "dns_tv_bwd")
        return d, None, None, None, None


def tv_smoothness(latents: torch.Tensor, n: int, sample_points: int, nx: Optional[int] = None, halo: bool = False) -> torch.Tensor:
    """Total variation of latents[:, 0] on an n^3 lattice / sample_points^3 (slams/mapping.py:151-157).
    latents [n^3, L] = the coarse decoder's output on the lattice points (x-major).  ``nx`` / ``halo``: the rows are a slab of
    nx x-planes of a lattice cut along x, the last plane being the next slab's first (include/dns_hip.h): the slabs' values
    sum to the cube's."""
    return _TvFn.apply(latents, n, sample_points, n if nx is None else nx, halo)


# ----------------------------------------------------------------------------- 2-D feature lookup
def feature_gather(pts: torch.Tensor, refer_w2c: torch.Tensor, K, feat_nhwc: torch.Tensor, H: int, W: int):
    """pts [P,3], refer_w2c [R,4,4], K 3x3 (host values), feat_nhwc [R,h,w,C] -> (code [R,P,C], mask [R,P] bool).
    No gradient (frozen stem features, rounded pixel)."""
    require_cuda(pts, refer_w2c, feat_nhwc)
    pts = pts.detach().contiguous().float()
    w2c = refer_w2c.detach().contiguous().float()
    R, h, w, Cc = feat_nhwc.shape
    P = pts.shape[0]
    Kh = (C.c_float * 9)(*[float(v) for v in torch.as_tensor(K).reshape(-1).tolist()])
    code = torch.empty(R, P, Cc, device=pts.device)
    mask = torch.empty(R, P, device=pts.device, dtype=torch.uint8)
    check(lib.dns_feature_gather(ptr(pts), ptr(w2c), Kh, ptr(feat_nhwc), R, P, Cc, h, w, H, W, ptr(code), ptr(mask),
                                 stream_ptr()), "dns_feature_gather")
    return code, mask.bool()

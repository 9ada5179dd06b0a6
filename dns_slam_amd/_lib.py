"""ctypes binding of libdns_hip.so (C ABI declared in include/dns_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C dns_slam_amd/csrc``.  Loading fails
loudly: there is no Python or CPU fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DNS_HIP_LIB") or os.path.join(_HERE, "libdns_hip.so")   # override: A/B of two builds
DNS_MAX_LEVELS = 32
ABI_VERSION = 12


class DnsGridMeta(C.Structure):
    _fields_ = [
        ("n_levels", C.c_uint32), ("n_features", C.c_uint32), ("log2_hashmap_size", C.c_uint32),
        ("base_resolution", C.c_uint32), ("total_rows", C.c_uint32), ("per_level_scale", C.c_float),
        ("scale", C.c_float * DNS_MAX_LEVELS), ("resolution", C.c_uint32 * DNS_MAX_LEVELS),
        ("size", C.c_uint32 * DNS_MAX_LEVELS), ("offset", C.c_uint32 * DNS_MAX_LEVELS),
        ("hashed", C.c_uint32 * DNS_MAX_LEVELS),
    ]


class DnsSplitRows(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("exps", C.c_void_p), ("ld", C.c_uint32), ("lo_off", C.c_uint32)]


class DnsAdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_uint64),
                ("lr", C.c_float)]


_P = C.c_void_p
_U = C.c_uint32
_I = C.c_int


class DnsTrackFused(C.Structure):
    """include/dns_hip.h: the argument block of dns_track_fused_iter (the tracker's iteration as one kernel + a pose kernel)."""
    _fields_ = [("color", _P), ("depth", _P), ("label", _P),
                ("H", C.c_int32), ("W", C.c_int32), ("H0", C.c_int32), ("H1", C.c_int32), ("W0", C.c_int32), ("W1", C.c_int32),
                ("cam", _P), ("bound", _P),
                ("pix", _P), ("t_uniform", _P), ("t_surf", _P), ("t_zero", _P), ("dmax", _P), ("iter", _P), ("n_iters", _U),
                ("n_uniform", _U), ("n_surface", _U), ("n_rays", _U),
                ("quat", _P), ("trans", _P),
                ("table", _P), ("meta", C.POINTER(DnsGridMeta)), ("n_bins", _U),
                ("w_coarse", _P), ("w_color", _P), ("w_logit", _P),
                ("n_neurons", _U), ("n_hidden_layers", _U), ("hidden", _U), ("n_feat", _U), ("n_class", _U),
                ("code", _P), ("code_dim", _U),
                ("lambda_p", C.c_float), ("lambda_d", C.c_float), ("lambda_l", C.c_float),
                ("ws", _P), ("adam_m", _P), ("adam_v", _P), ("adam_state", _P),
                ("lr_quat", C.c_float), ("lr_trans", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("best_loss", _P), ("best_cam", _P), ("out", _P), ("g_quat", _P), ("g_trans", _P)]

# name -> (restype, argtypes); must list every symbol include/dns_hip.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "dns_abi_version": (C.c_int, []),
    "dns_last_error": (C.c_char_p, []),
    "dns_init": (C.c_int, []),
    "dns_kernel_timing": (C.c_int, [C.c_int]),
    "dns_kernel_timing_count": (C.c_int, []),
    "dns_kernel_timing_get": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_float)]),
    "dns_grid_meta_init": (C.c_int, [C.POINTER(DnsGridMeta), _U, _U, _U, _U, C.c_double]),
    "dns_raygen_sample": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I,
                                    _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dns_rays_from_pixels": (C.c_int, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "dns_sample_along_rays": (C.c_int, [_P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "dns_raygen_bwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dns_raygen_bwd_ws_floats": (C.c_uint64, [_I, _I]),
    "dns_encode_fwd": (C.c_int, [_P, _P, _U, _U, _P, C.POINTER(DnsGridMeta), _P, _P, _U, _P, _U, _P, _P]),
    "dns_encode_bwd": (C.c_int, [_P, _P, _U, _U, _P, C.POINTER(DnsGridMeta), _P, _U, _P, _U, _P, _P, _P, _P, _U, _U, _P]),
    "dns_encode_bwd_ws_floats": (C.c_uint64, [_U, C.POINTER(DnsGridMeta), _U, _U]),
    "dns_hashgrid_indices": (C.c_int, [_P, _U, C.POINTER(DnsGridMeta), _P, _P]),
    "dns_mlp_fwd": (C.c_int, [_P, _U, _P, _U, _U, _P, _U, _U, _U, _U, _P, _U, _U, _P, _P, _U, _P, _U, _P]),
    "dns_mlp_bwd": (C.c_int, [_P, _U, _P, _U, _U, _P, _U, _P, _U, _U, _U, _U, _P, _U, _P, _U, _P, _P, _U, _P, _P, _U, _P, _I, _P]),
    "dns_mlp_bwd_ws_floats": (C.c_uint64, [_U, _U, _U]),
    "dns_mlp_dwin": (C.c_int, [_P, _U, _P, _U, _U, _U, _U, _U, _P, _P, _U, _P, _P, _U, _U, _P]),
    "dns_mlp_prepared_floats": (C.c_uint64, [_U, _U, _U, _U]),
    "dns_mlp_prepare": (C.c_int, [_P, _U, _U, _U, _U, _U, _U, _P, _U, _P]),
    "dns_encode_fwd_split": (C.c_int, [_P, _P, _U, _U, _P, C.POINTER(DnsGridMeta), _P, _P, _U, _P, _U, _P, _U, _P, _P]),
    "dns_mlp_fwd_split": (C.c_int, [C.POINTER(DnsSplitRows), C.POINTER(DnsSplitRows), _U, _P, _U, _U, _U, _U, _P, _U, _U, _P, _P, _U,
                                    _U, _P]),
    "dns_mlp_bwd_split": (C.c_int, [C.POINTER(DnsSplitRows), C.POINTER(DnsSplitRows), _U, _P, _U, _P, _U, _U, _U, _U, _P, _U, _P, _U,
                                    _P, _P, _U, _P, _P, _U, _I, _P]),
    "dns_mlp_fwd_half": (C.c_int, [_P, _U, _P, _U, _U, _P, _U, _U, _U, _U, _P, _U, _U, _P, _P, _U, _U, _P]),
    "dns_mlp_bwd_half": (C.c_int, [_P, _U, _P, _U, _U, _P, _U, _P, _U, _U, _U, _U, _P, _U, _P, _U, _P, _U, _P, _P, _U, _I, C.c_float, _P]),
    "dns_feature_block_split": (C.c_int, [_P, _U, _U, _P, _U, _U, _U, _P, _P, _U, _U, _P, _U, _P, _U, _P, _U, _P, _P]),
    "dns_loss_sums": (C.c_int, [_P, _U, _U, _U, _U, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dns_loss_rays": (C.c_int, [_P, _U, _U, _U, _U, _I] + [_P] * 19),
    "dns_loss_finalize_bwd": (C.c_int, [_P, _U, _U, _U, _U, _I] + [_P] * 16),
    "dns_loss_finalize": (C.c_int, [_P, _U, _U, _U, _U, _I, _P, _P, _P]),
    "dns_loss_bwd": (C.c_int, [_P, _U, _U, _U, _U, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                _P, _P, _U, _P]),
    "dns_loss_bwd_points": (C.c_int, [_P, _U, _U, _U, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _P, _U, _P]),
    "dns_composite_fwd_ex": (C.c_int, [_P, _P, _P, _U, _U, _U, _P, _P, _P, _P, _P, _U, _P]),
    "dns_composite_bwd_ex": (C.c_int, [_P, _P, _P, _U, _U, _U, _P, _P, _P, _P, _P, _P, _P, _U, _P]),
    "dns_class_slots": (C.c_int, [_P, _U, _U, _I, _P, _U, _P, _P]),
    "dns_feature_block": (C.c_int, [_P, _U, _U, _P, _U, _P, _P, _U, _U, _P, _U, _P, _P]),
    "dns_rgb_sigmoid": (C.c_int, [_P, _U, _P]),
    "dns_raw_bwd": (C.c_int, [_P, _P, _U, _P, _P, _U, _I, _P]),
    "dns_lattice_points": (C.c_int, [_P, _P, _U, _P, _U, _P, _P]),
    "dns_draw_finish": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _U, _U, _U, _U, _P, _P, _P, _P]),
    "dns_track_mask": (C.c_int, [_P, _P, _U, C.c_float, _P, _P]),
    "dns_keep_best": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "dns_force_half": (C.c_int, [_P, _U, _U, _P]),
    "dns_track_fused_ws_floats": (C.c_uint64, [C.POINTER(DnsTrackFused)]),
    "dns_track_fused_begin": (C.c_int, [_P, _U, _U, _P, _I, _I, _I, _I, _P, _U, _P, _P]),
    "dns_track_fused_iter": (C.c_int, [C.POINTER(DnsTrackFused), _P]),
    "dns_feature_gather": (C.c_int, [_P, _P, _P, _P, _U, _U, _U, _I, _I, _I, _I, _P, _P, _P]),
    "dns_feature_gather_frames": (C.c_int, [_P, _P, _P, _P, _P, _U, _U, _U, _U, _I, _I, _I, _I, _P, _U, _P, _P]),
    "dns_refer_poses": (C.c_int, [_P, _P, _P, _P, _U, _P, _P, _P]),
    "dns_merge_dy": (C.c_int, [_P, _U, _U, _U, _U, _P, _P, _U, _U, _P, _P]),
    "dns_add_ref_sum": (C.c_int, [_P, _U, _U, _U, _P, _P]),
    "dns_tv_fwd": (C.c_int, [_P, _U, _U, _U, _I, _U, _P, _P]),
    "dns_tv_bwd": (C.c_int, [_P, _U, _U, _U, _I, _U, _P, _P, _P]),
    "dns_group_slots": (C.c_int, [_P, _U, _U, _U, _U, _P, _P, _P, _P]),
    "dns_group_scatter": (C.c_int, [_P, _U, _U, _P, _U, _P, _P]),
    "dns_device_error": (C.c_int, [C.c_int]),
    "dns_adam_step": (C.c_int, [C.POINTER(DnsAdamTensor), _U, C.c_float, C.c_float, C.c_float, _P, _P]),
    "dns_composite_fwd": (C.c_int, [_P, _P, _P, _U, _U, _U, _P, _P, _P, _P, _P, _P]),
    "dns_composite_bwd": (C.c_int, [_P, _P, _P, _U, _U, _U, _P, _P, _P, _P, _P, _P, _P, _P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP hot-path library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C dns_slam_amd/csrc`). "
            "dns_slam_amd has no CPU fallback.")
    # torch first: libdns_hip.so has no HIP runtime of its own and binds to the one PyTorch-ROCm loaded
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.dns_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"libdns_hip.so ABI version {v} != expected {ABI_VERSION}; rebuild it")
    return lib


lib = _load()


_ready_devices = set()


def ensure_init() -> None:
    """dns_init() once per device, outside any stream capture (include/dns_hip.h): called by stream_ptr(), i.e. ahead of
    every launch, so that no kernel attribute is ever set lazily inside a hipGraph capture."""
    import torch
    d = torch.cuda.current_device()
    if d not in _ready_devices:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("dns_slam_amd: first use on this device inside a hipGraph capture; run one op (or "
                               "dns_slam_amd._lib.ensure_init()) before capturing")
        check(lib.dns_init(), "dns_init")
        _ready_devices.add(d)


def check(rc: int, what: str = "") -> None:
    """Raise on a negative return code: -1 -> ValueError (argument), others -> RuntimeError (launch / state)."""
    if rc == 0:
        return
    msg = lib.dns_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(f"{what}: {msg}")
    if rc == -2:
        # DNS_E_LAUNCH can be the STICKY device error word (include/dns_hip.h: set by a kernel that had to clamp a device
        # counter, reported by whichever entry point polls next).  Read and clear it here, so that ONE fault raises ONE
        # exception -- naming the word -- instead of failing every later call on the device.
        word = lib.dns_device_error(1)
        if word:
            raise RuntimeError(f"{what}: {msg} (code {rc}; device error word {word:#x} was set by an EARLIER kernel -- bit 0: "
                               f"dns_group_slots / dns_group_scatter cursor past n_slots -- and has been cleared)")
    raise RuntimeError(f"{what}: {msg} (code {rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    ensure_init()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    import torch
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise ValueError("dns_slam_amd ops run on the GPU only (got a non-CUDA tensor); there is no CPU fallback")
        if not t.is_contiguous():
            raise ValueError("dns_slam_amd ops need contiguous tensors")

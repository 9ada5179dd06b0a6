"""CPU: the C-ABI library loads and exports every symbol include/dns_hip.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dns_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dns_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from dns_slam_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 12
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(cdll, s), f"{s} declared in dns_hip.h but not exported by libdns_hip.so"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in dns_slam_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(syms)


def test_abi_version_and_error_slot():
    from dns_slam_amd import _lib
    assert _lib.lib.dns_abi_version() == _lib.ABI_VERSION
    m = _lib.DnsGridMeta()
    rc = _lib.lib.dns_grid_meta_init(ctypes.byref(m), 99, 2, 16, 16, 1.3)     # too many levels
    assert rc == -1
    assert b"n_levels" in _lib.lib.dns_last_error()
    try:
        _lib.check(rc, "dns_grid_meta_init")
        raise AssertionError("check() must raise")
    except ValueError:
        pass


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "dns_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "from .. import oracle" not in src


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from dns_slam_amd import ops
    with pytest.raises(ValueError):
        ops.composite(torch.zeros(2, 4, 4), torch.zeros(2, 4), None)
    with pytest.raises(ValueError):
        ops.mlp(torch.zeros(4, 80), torch.zeros(ops.mlp_param_count(80, 33, 32, 1)), 80, 33)

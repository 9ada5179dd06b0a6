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


def test_scatter_workspace_sizes_per_form():
    """dns_encode_bwd_ws_floats is host arithmetic (no GPU): the workspace of the table scatter per form.  Every form holds the
    level-major gradient copy (P x L x 2 floats) + the max words; the pair lists (DNS_SCATTER_LISTS, implied by _AUTO) add
    16 bytes per point and hashed level (+ slack) and, for large dense levels, an exact 8 P-word region each; the queue form adds
    its 24-byte-per-corner queues; a caller-chosen capacity shrinks lists and queues alike; the replayed rows add 4 floats per
    point and level."""
    from dns_slam_amd import _lib, ops
    ws = lambda meta, P, flags, cap=0: int(_lib.lib.dns_encode_bwd_ws_floats(P, ctypes.byref(meta.c), flags, cap))
    P = 262144
    small, big = ops.GridMeta(16, 592), ops.GridMeta(20, 231)         # T = 2^16: 4 dense + 12 hashed levels; T = 2^20: 11 dense + 5 hashed
    base = P * 16 * 2 + 4
    assert ws(small, P, ops.SCATTER_BINNED) == base                   # sweep only: no lists, no queues (levels of < 16 chunks)
    auto, lists = ws(small, P, ops.SCATTER_AUTO), ws(small, P, ops.SCATTER_BINNED | ops.SCATTER_LISTS)
    assert auto == lists > base
    n_hashed = sum(1 for l in range(16) if small.c.hashed[l])
    per_level = (auto - base) / n_hashed / P                          # words per point and hashed level: 4 entries + 1/8 slack
    assert 4.0 <= per_level <= 5.0, per_level
    assert ws(small, P, ops.SCATTER_AUTO, 64) < auto                  # 64-entry lists
    assert ws(small, P, ops.SCATTER_AUTO | ops.SCATTER_REPLAY) == ((auto + 3) // 4) * 4 + P * 16 * 4
    q, a = ws(big, P, ops.SCATTER_QUEUES), ws(big, P, ops.SCATTER_AUTO)
    assert q > base and a > base
    n_dense_big = sum(1 for l in range(16) if not big.c.hashed[l] and big.c.size[l] >= 6 * 8192 - 8191)
    assert a - base >= n_dense_big * 8 * P                            # every large dense level owns an 8 P-word region
    assert ws(small, 0, ops.SCATTER_AUTO) <= 4 + 4 * 8192 + 8         # no points: counters only

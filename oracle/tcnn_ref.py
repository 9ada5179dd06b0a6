"""Oracle: restatement of the three tiny-cuda-nn components the reference calls.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED: ``tinycudann`` is a
third-party CUDA package (reference ``requirements.txt:35``, unpinned HEAD; README
fallback commit 91ee479d) that is absent from ``/root/reference`` and cannot be built
here; the reference holds no golden vectors for it.  The functions below restate its
published algorithm (Mueller et al., "Instant Neural Graphics Primitives", 2022, sec. 3;
tiny-cuda-nn ``encodings/grid.h``, ``encodings/oneblob.h``, ``networks/cutlass_mlp``) at
the reference's call sites:

* HashGrid   ``models/pos_encoding.py:31-46``  (n_levels 16, 2 features, base 16,
  ``per_level_scale = exp2(log2(res/16)/15)`` computed in float64 at :33)
* OneBlob    ``models/pos_encoding.py:61-71``  (n_bins 16)
* CutlassMLP ``models/decoder.py:58-64,84-90,101-116``, ``slams/mapping.py:737-743``
  (ReLU, no output activation, no bias)

Everything is plain differentiable PyTorch so autograd supplies d/dparams and d/dx.
Two deliberate, documented choices where tcnn's own arithmetic cannot be reproduced
bit-for-bit without the package: ``pos = x*scale + 0.5`` is two IEEE roundings (tcnn
uses one fused multiply-add); MLPs compute in fp32 (tcnn stores fp16) -- SURVEY D11.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)  # tcnn coherent_prime_hash, 3-D


@dataclass
class Level:
    scale: np.float32
    resolution: int
    size: int      # rows in this level
    offset: int    # first row
    hashed: bool


@dataclass
class GridMeta:
    n_levels: int
    n_features: int
    base_resolution: int
    log2_hashmap_size: int
    per_level_scale: float
    levels: List[Level]
    total_rows: int


def desired_resolution(bound: torch.Tensor, voxel_size: float) -> int:
    """models/decoder.py:38-39: int(max extent / voxel_size) on the fp64 bound."""
    dim_max = (bound[:, 1] - bound[:, 0]).max()
    return int(dim_max / voxel_size)


def per_level_scale(desired_res: int, base_resolution: int = 16, n_levels: int = 16) -> float:
    """models/pos_encoding.py:33 (float64 numpy)."""
    return float(np.exp2(np.log2(desired_res / base_resolution) / (n_levels - 1)))


def grid_meta(log2_hashmap_size: int, desired_res: int, n_levels: int = 16, n_features: int = 2,
              base_resolution: int = 16, per_level_scale=None) -> GridMeta:
    """tcnn GridEncoding constructor: per level
    scale = exp2(l * log2(pls)) * base - 1, evaluated in float64 and rounded once to float32 (tcnn uses CUDA's
    float32 exp2f, which cannot be reproduced off-CUDA and is 1-ulp-sensitive at the finest level -- SURVEY
    Appendix A7; the float64 definition makes the finest scale exactly desired_res - 1), res = ceilf(scale) + 1,
    size = min(next_multiple(res^3, 8), 2^log2_hashmap_size); a level is hashed iff the
    dense stride product exceeds its size."""
    # ``per_level_scale`` given: the float64 value models/pos_encoding.py:33 hands to tcnn.Encoding (desired_res is unused)
    pls = float(per_level_scale) if per_level_scale is not None else globals()["per_level_scale"](desired_res, base_resolution, n_levels)
    log2_pls = np.log2(np.float64(pls))
    levels = []
    offset = 0
    T = 1 << log2_hashmap_size
    for l in range(n_levels):
        scale = np.float32(np.exp2(np.float64(l) * log2_pls) * np.float64(base_resolution) - 1.0)
        res = int(np.ceil(scale)) + 1
        dense = res ** 3
        size = min(((dense + 7) // 8) * 8, T)
        # tcnn loop: for (dim = 0; dim < 3 && stride <= size; ++dim) stride *= res (uint32)
        stride = 1
        for _ in range(3):
            if stride > size:
                break
            stride = (stride * res) & 0xFFFFFFFF
        hashed = size < stride
        levels.append(Level(scale, res, size, offset, hashed))
        offset += size
    return GridMeta(n_levels, n_features, base_resolution, log2_hashmap_size, pls, levels, offset)


def grid_index(gx, gy, gz, lvl: Level):
    """tcnn ``grid_index``: dense x + y*res + z*res^2 while the running stride stays
    <= level size, else the coherent prime hash; then ``% size``.  uint32 wrap-around is
    emulated in int64."""
    M = 0xFFFFFFFF
    size = lvl.size
    res = lvl.resolution
    stride = 1
    index = torch.zeros_like(gx)
    for g in (gx, gy, gz):
        if stride > size:
            break
        index = (index + g * stride) & M
        stride = (stride * res) & M
    if size < stride:
        index = ((gx * PRIMES[0]) & M) ^ ((gy * PRIMES[1]) & M) ^ ((gz * PRIMES[2]) & M)
    return index % size


def hashgrid_indices(x: torch.Tensor, meta: GridMeta):
    """[P,3] fp32 -> int64 [P, L, 8] absolute table rows and fp32 [P, L, 3] fractions.
    Corner c has offset bit d of c along dim d (tcnn corner loop)."""
    P = x.shape[0]
    rows = torch.empty(P, meta.n_levels, 8, dtype=torch.int64)
    fracs = torch.empty(P, meta.n_levels, 3, dtype=torch.float32)
    for l, lvl in enumerate(meta.levels):
        pos = x * torch.tensor(lvl.scale) + 0.5          # two fp32 roundings
        fl = torch.floor(pos)
        # (uint32)(int)floorf(pos): negatives wrap
        g = fl.to(torch.int32).to(torch.int64) & 0xFFFFFFFF
        fracs[:, l] = pos - fl
        for c in range(8):
            gx = (g[:, 0] + (c & 1)) & 0xFFFFFFFF
            gy = (g[:, 1] + ((c >> 1) & 1)) & 0xFFFFFFFF
            gz = (g[:, 2] + ((c >> 2) & 1)) & 0xFFFFFFFF
            rows[:, l, c] = grid_index(gx, gy, gz, lvl) + lvl.offset
    return rows, fracs


def hashgrid_forward(x: torch.Tensor, table: torch.Tensor, meta: GridMeta) -> torch.Tensor:
    """Trilinear multi-resolution gather: [P,3] fp32, table [rows, F] fp32 -> [P, L*F].
    Differentiable w.r.t. ``table`` (scatter-add) and ``x`` (through the fractions)."""
    outs = []
    for l, lvl in enumerate(meta.levels):
        pos = x * torch.tensor(lvl.scale) + 0.5
        fl = torch.floor(pos).detach()
        g = fl.to(torch.int32).to(torch.int64) & 0xFFFFFFFF
        f = pos - fl
        acc = 0
        for c in range(8):
            bx, by, bz = c & 1, (c >> 1) & 1, (c >> 2) & 1
            idx = grid_index((g[:, 0] + bx) & 0xFFFFFFFF, (g[:, 1] + by) & 0xFFFFFFFF,
                             (g[:, 2] + bz) & 0xFFFFFFFF, lvl) + lvl.offset
            w = (f[:, 0] if bx else 1 - f[:, 0]) * (f[:, 1] if by else 1 - f[:, 1]) \
                * (f[:, 2] if bz else 1 - f[:, 2])
            acc = acc + w[:, None] * table[idx]
        outs.append(acc)
    # a float64 ``table`` (tests that need the scatter-add of the table gradient free of fp32 summation error) promotes the
    # products; the features leave in x's dtype either way
    return torch.cat(outs, -1).to(x.dtype)


def _quartic_cdf(v, n_bins):
    u = v * n_bins
    u2 = u * u
    u4 = u2 * u2
    return torch.clamp((15.0 / 16.0) * u * (1 - (2.0 / 3.0) * u2 + (1.0 / 5.0) * u4) + 0.5, 0.0, 1.0)


def oneblob_forward(x: torch.Tensor, n_bins: int = 16) -> torch.Tensor:
    """tcnn ``kernel_one_blob``: per coordinate, bin b gets G(b+1) - G(b) with
    G(b) = cdf(b/n - x) + cdf(b/n - x - 1) + cdf(b/n - x + 1), the last bin taking
    G(0) + 1 as its right edge (wrap); quartic kernel of radius 1/n.  Channel order
    dim*n_bins + bin.  [P,D] -> [P, D*n_bins]."""
    P, D = x.shape
    b = torch.arange(n_bins, dtype=x.dtype) / n_bins                  # left boundaries
    diff = b[None, None, :] - x[:, :, None]                              # [P,D,n]
    left = _quartic_cdf(diff, n_bins) + _quartic_cdf(diff - 1.0, n_bins) + _quartic_cdf(diff + 1.0, n_bins)
    right = torch.cat([left[..., 1:], left[..., :1] + 1.0], -1)
    return (right - left).reshape(P, D * n_bins)


def mlp_out_padded(n_out: int) -> int:
    """CutlassMLP pads the output width to a multiple of 16 (rows beyond n_out are
    storage only; the torch binding slices them away)."""
    return ((n_out + 15) // 16) * 16


def mlp_param_count(n_in: int, n_out: int, n_neurons: int, n_hidden_layers: int) -> int:
    return n_neurons * n_in + (n_hidden_layers - 1) * n_neurons * n_neurons + mlp_out_padded(n_out) * n_neurons


def mlp_split(params: torch.Tensor, n_in: int, n_out: int, n_neurons: int, n_hidden_layers: int):
    """Flat fp32 ``params`` -> [W_in [n,in], W_h [n,n] x (layers-1), W_out [out_pad, n]],
    all row-major, concatenated in that order (tcnn ``m_weight_matrices``)."""
    mats = []
    o = 0
    mats.append(params[o:o + n_neurons * n_in].reshape(n_neurons, n_in)); o += n_neurons * n_in
    for _ in range(n_hidden_layers - 1):
        mats.append(params[o:o + n_neurons * n_neurons].reshape(n_neurons, n_neurons)); o += n_neurons * n_neurons
    op = mlp_out_padded(n_out)
    mats.append(params[o:o + op * n_neurons].reshape(op, n_neurons))
    return mats


def mlp_forward(x: torch.Tensor, params: torch.Tensor, n_in: int, n_out: int, n_neurons: int = 32,
                n_hidden_layers: int = 1) -> torch.Tensor:
    """y = W_out relu(... relu(W_in x)); no bias; fp32."""
    mats = mlp_split(params, n_in, n_out, n_neurons, n_hidden_layers)
    h = x
    for W in mats[:-1]:
        h = torch.relu(h @ W.t())
    return (h @ mats[-1].t())[:, :n_out]


def mlp_init(n_in: int, n_out: int, n_neurons: int, n_hidden_layers: int, generator=None) -> torch.Tensor:
    """Xavier-uniform per matrix (tcnn default initialisation), padded output rows
    included; returns the flat fp32 params."""
    chunks = []
    shapes = [(n_neurons, n_in)] + [(n_neurons, n_neurons)] * (n_hidden_layers - 1) + [(mlp_out_padded(n_out), n_neurons)]
    for (r, c) in shapes:
        s = float(np.sqrt(6.0 / (r + c)))
        chunks.append(((torch.rand(r * c, generator=generator) * 2 - 1) * s).float())
    return torch.cat(chunks)


def grid_init(meta: GridMeta, generator=None) -> torch.Tensor:
    """tcnn grid initialisation U(-1e-4, 1e-4); [rows, F] fp32."""
    return ((torch.rand(meta.total_rows, meta.n_features, generator=generator) * 2 - 1) * 1e-4).float()

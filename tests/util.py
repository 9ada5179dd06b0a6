"""Shared helpers for the parity tests: build a product Decoder/Mapper and an oracle model with IDENTICAL
parameters, and compare tensors with the tolerance BASELINE.json states (1e-4 relative, fp32)."""
import numpy as np
import torch

from oracle import slam_ref as sr
from oracle import tcnn_ref as tr

RTOL = 1e-4   # BASELINE.json north_star: "within 1e-4 rel fp32"

# every comparison made through assert_close: (what, scale-relative error, worst element-wise ratio); conftest.py writes it
# to gpurun_out/parity_report.json at the end of a session so that both figures are on record for every tensor
REPORT = []


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max|b|  (relative to the tensor's scale, robust to zeros)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    scale = b.abs().max().item()
    if scale == 0.0:
        return (a - b).abs().max().item()
    return ((a - b).abs().max() / scale).item()


def _group_scales(b: torch.Tensor, groups) -> torch.Tensor:
    """Per-element magnitude scale: the RMS of the NON-ZERO reference entries of the element's group.  groups: None (the
    whole tensor is one group), an int d (one group per index along dimension d: per output column of a [P, C] tensor,
    per level of a [L, rows, F] view, ...) or a list of slices of the flattened tensor (per matrix of a flat parameter
    vector)."""
    def rms_nz(t):
        nz = t[t != 0]
        return nz.pow(2).mean().sqrt() if nz.numel() else torch.zeros((), dtype=t.dtype)
    s = torch.empty_like(b)
    if groups is None:
        s.fill_(rms_nz(b))
    elif isinstance(groups, int):
        bm = b.movedim(groups, 0)
        sm = s.movedim(groups, 0)
        for i in range(bm.shape[0]):
            sm[i] = rms_nz(bm[i])
    else:
        fb, fs = b.reshape(-1), s.reshape(-1)
        fs.fill_(rms_nz(fb))
        for sl in groups:
            fs[sl] = rms_nz(fb[sl])
    return s


def elem_err(a: torch.Tensor, b: torch.Tensor, rtol=RTOL, groups=None, atol_factor=1.0, atol=None, outlier_frac=0.0) -> float:
    """Worst element-wise ratio |a-b| / (rtol*|b| + atol); <= 1 passes.  atol = atol_factor * rtol * (RMS of the non-zero
    reference entries of the element's group): an entry much smaller than its group's typical magnitude is a sum with
    cancellation, whose fp32 error scales with the terms (~ the group's magnitude), not with the result -- everything else
    is held to rtol of ITS OWN magnitude, so small entries (fine levels next to coarse ones, padded rows) are checked too."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    if b.numel() == 0:
        return 0.0
    if not torch.isfinite(b).all():                   # NaN / Inf must match in place (D9: NaN propagates)
        same = (torch.isnan(a) == torch.isnan(b)).all() and (torch.isinf(a) == torch.isinf(b)).all()
        if not same:
            return float("inf")
        m = torch.isfinite(b)
        a, b = torch.where(m, a, torch.zeros_like(a)), torch.where(m, b, torch.zeros_like(b))
    if atol is not None:                              # explicit per-element (or scalar) absolute term, e.g. from sum |terms|
        tol = rtol * b.abs() + (atol.detach().double().cpu() if torch.is_tensor(atol) else float(atol))
    else:
        tol = rtol * b.abs() + atol_factor * rtol * _group_scales(b, groups)
    d = (a - b).abs()
    ok0 = (tol == 0) & (d == 0)
    ratio = torch.where(ok0, torch.zeros_like(d), d / tol.clamp_min(1e-300)).reshape(-1)
    i = int(torch.argmax(ratio))
    elem_err.worst = (i, a.reshape(-1)[i].item(), b.reshape(-1)[i].item(), tol.reshape(-1)[i].item())
    if outlier_frac > 0.0 and ratio.numel() > 1:
        # a hidden unit whose pre-activation is within rounding of zero falls on either side of the ReLU in two fp32
        # implementations; the affected point's contribution then differs by a finite amount.  Such elements are allowed as a
        # stated FRACTION of the tensor (they still obey the scale-relative criterion); the figure returned is the worst ratio
        # among the rest.
        k = max(int(ratio.numel() * (1.0 - outlier_frac)), 1)
        return torch.kthvalue(ratio, k).values.item()
    return ratio.max().item()


def assert_close(a, b, rtol=RTOL, what="", groups=None, atol_factor=1.0, elementwise=True, atol=None, outlier_frac=0.0):
    """Two criteria, both reported: (i) max|a-b| <= rtol * max|b| (the tensor's scale); (ii) element-wise
    |a-b| <= rtol*|b| + atol with the per-group atol of ``elem_err``."""
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b)
    r = elem_err(a, b, rtol, groups, atol_factor, atol, outlier_frac) if elementwise else float("nan")
    REPORT.append((what, e, r, rtol))
    assert e <= rtol, f"{what}: scale-relative error {e:.3e} > {rtol:.1e} (element-wise ratio {r:.2f})"
    assert not (r > 1.0), (f"{what}: element-wise |a-b| <= {rtol:.0e}*|b| + atol violated, worst ratio {r:.2f} (scale-relative "
                           f"{e:.3e}); worst element (index, got, want, tolerance) = {getattr(elem_err, 'worst', None)}")


def assert_pose_grad_close(q, gq, gq_ref, gT, gT_ref, what=""):
    """Pose gradients.  The rotation depends on q only through q/|q| (utils/common.py quad2rotation divides by |q|^2), so the
    exact gradient has NO component along q: whatever either fp32 implementation returns there is rounding residue of terms
    that cancel analytically, and it moves nothing (a step along q leaves the rotation unchanged).  The component in the
    tangent space of the unit sphere -- the part that rotates the camera -- is held to 1e-4; the radial residue of each side
    must be small against the gradient itself; the translation gradient is held to 1e-4."""
    q = q.detach().double().cpu()
    n = q / q.norm()
    tang = lambda g: (g.detach().double().cpu() - (g.detach().double().cpu() @ n) * n).float()
    assert_close(tang(gq), tang(gq_ref), rtol=1e-4, what=f"{what} d quat (tangential)", elementwise=False)
    for g, side in ((gq, "product"), (gq_ref, "oracle")):
        rad = abs(float(g.detach().double().cpu() @ n))
        REPORT.append((f"{what} d quat radial residue / |g| ({side})", rad / float(g.norm()), rad / float(g.norm()) / 1e-3, 1e-3))
        assert rad <= 1e-3 * float(g.norm()), f"{what}: radial component {rad:.3e} of |g| = {float(g.norm()):.3e} ({side})"
    assert_close(gT.detach().cpu(), gT_ref, rtol=1e-4, what=f"{what} d T", elementwise=False)


def table_level_groups(meta, n_features=2):
    """Slices of the flattened [total_rows, F] hash table (or its gradient), one per level: a fine level's entries are
    compared against the magnitude of THEIR level, not the coarse levels'.  meta: oracle or product grid meta."""
    if callable(getattr(meta, "levels", None)):                      # product ops.GridMeta
        lv = [(d["offset"], d["size"]) for d in meta.levels()]
    else:                                                            # oracle tcnn_ref.GridMeta (list of Level)
        lv = [(int(l.offset), int(l.size)) for l in meta.levels]
    return [slice(o * n_features, (o + z) * n_features) for o, z in lv]


def mlp_param_groups(n_in, n_out, nn, nl):
    """Slices of a flat tcnn-layout parameter vector, one per weight matrix (W_in | W_hidden... | W_out)."""
    out, o = [], 0
    for r, c in [(nn, n_in)] + [(nn, nn)] * (nl - 1) + [((n_out + 15) // 16 * 16, nn)]:
        out.append(slice(o, o + r * c))
        o += r * c
    return out


def oracle_cfg_from(cfg: dict, n_class: int) -> sr.ModelCfg:
    m = cfg["model"]
    return sr.ModelCfg(n_bins=m["pos"]["n_bins"], hash_size=m["grid"]["hash_size"], voxel_size=m["grid"]["voxel_size"],
                       hidden_dim=m["hidden_dim"], n_neurons=m["mlp"]["n_neurons"],
                       n_hidden_layers=m["mlp"]["n_hidden_layers"], pixel_dim=m["pixel_dim"], n_class=n_class)


def oracle_from_product(cfg, bound, decoder, mapper=None, n_class=8, table64=False) -> sr.OracleModel:
    """OracleModel carrying copies of the product's parameters (flat fp32 layouts are identical by design).  table64: the
    hash table is a float64 leaf, so autograd's scatter-add of its gradient (index_add over up to thousands of
    contributions per cell at full size) is free of the ORACLE's own fp32 summation error; values are unchanged."""
    om = sr.OracleModel(oracle_cfg_from(cfg, n_class), bound, fine_classes=())
    cp = lambda p: p.detach().cpu().clone().float().requires_grad_(True)
    om.table = cp(decoder.pe_fn.grid_fn.params).reshape(om.meta.total_rows, 2).detach()
    om.table = (om.table.double() if table64 else om.table).requires_grad_(True)
    om.coarse = cp(decoder.coarse_fn.decoder.params)
    om.color = cp(decoder.out_fn.color_decoder.params)
    om.logit = cp(decoder.out_fn.logit_decoder.params)
    if mapper is not None:
        om.fine = {c: cp(mapper.fine_decoders.params_of(c)) for c in mapper.fine_decoders.keys()}
    return om


def randomise_(module_or_params, seed, scale=1.0):
    """Give every parameter a distinct random value (tcnn's default seed makes equal-shaped nets identical)."""
    g = torch.Generator().manual_seed(seed)
    ps = list(module_or_params.parameters()) if hasattr(module_or_params, "parameters") else list(module_or_params)
    with torch.no_grad():
        for p in ps:
            if p.numel() == 0:
                continue
            r = (torch.rand(p.shape, generator=g) * 2 - 1) * scale * max(float(p.detach().abs().max()), 1e-3)
            p.copy_(r.to(p.device))

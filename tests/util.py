"""Shared helpers for the parity tests: build a product Decoder/Mapper and an oracle model with IDENTICAL
parameters, and compare tensors with the tolerance BASELINE.json states (1e-4 relative, fp32)."""
import numpy as np
import torch

from oracle import slam_ref as sr
from oracle import tcnn_ref as tr

RTOL = 1e-4   # BASELINE.json north_star: "within 1e-4 rel fp32"


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max|b|  (relative to the tensor's scale, robust to zeros)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    scale = b.abs().max().item()
    if scale == 0.0:
        return (a - b).abs().max().item()
    return ((a - b).abs().max() / scale).item()


def assert_close(a, b, rtol=RTOL, what=""):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b)
    assert e <= rtol, f"{what}: relative error {e:.3e} > {rtol:.1e}"


def oracle_cfg_from(cfg: dict, n_class: int) -> sr.ModelCfg:
    m = cfg["model"]
    return sr.ModelCfg(n_bins=m["pos"]["n_bins"], hash_size=m["grid"]["hash_size"], voxel_size=m["grid"]["voxel_size"],
                       hidden_dim=m["hidden_dim"], n_neurons=m["mlp"]["n_neurons"],
                       n_hidden_layers=m["mlp"]["n_hidden_layers"], pixel_dim=m["pixel_dim"], n_class=n_class)


def oracle_from_product(cfg, bound, decoder, mapper=None, n_class=8) -> sr.OracleModel:
    """OracleModel carrying copies of the product's parameters (flat fp32 layouts are identical by design)."""
    om = sr.OracleModel(oracle_cfg_from(cfg, n_class), bound, fine_classes=())
    cp = lambda p: p.detach().cpu().clone().float().requires_grad_(True)
    om.table = cp(decoder.pe_fn.grid_fn.params).reshape(om.meta.total_rows, 2).detach().requires_grad_(True)
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

"""SURVEY 8f rank 4 on the GPU: a ``model.pt`` in the reference's layout -- tinycudann CutlassMLP tensors with their output
rows padded to 8 (models/decoder.py:60, tcnn CutlassMLP), ``fine_decoders`` pickled as tinycudann module OBJECTS
(slams/mapping.py:1121) -- is loaded into a fresh GPU ``Decoder`` + per-class pool, and ``Mapper.renderer`` on those weights
equals the oracle built INDEPENDENTLY from the same 8-padded tensors (the test slices W_in / W_out out of the flat tensors
itself; the product's ``repack_mlp_params`` is not involved on the oracle side).  A tcnn-written file cannot exist here
(SURVEY 8c): the padding facts are tcnn's published ones."""
import os

import pytest
import torch

from oracle import slam_ref as sr
from test_host_logic import write_reference_style_checkpoint
from util import assert_close, oracle_cfg_from

pytestmark = pytest.mark.gpu
DEV = "cuda"
N_CLASS = 6


def _pad8(n):
    return (n + 7) // 8 * 8


def _flat8(n_in, n_out, g):
    """A CutlassMLP flat tensor: [W_in 32 x n_in | W_out pad8(n_out) x 32]; the padded rows hold junk (tcnn initialises them
    like any other row) that no consumer may read."""
    return torch.cat(((torch.rand(32 * n_in, generator=g) * 2 - 1) * 0.4, (torch.rand(_pad8(n_out) * 32, generator=g) * 2 - 1) * 0.4))


def _to16(flat8, n_in, n_out):
    """The oracle's 16-row layout from the 8-row file tensor, by hand."""
    body = 32 * n_in
    w_out = flat8[body:body + n_out * 32]
    return torch.cat((flat8[:body], w_out, torch.zeros(((n_out + 15) // 16 * 16 - n_out) * 32)))


def test_reference_layout_checkpoint_renders_like_the_oracle(tmp_path):
    from dns_slam_amd import synthetic
    from dns_slam_amd.checkpoint import load_reference_decoder
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.mapping import Mapper
    g = torch.Generator().manual_seed(77)
    bound = synthetic.load_bound(synthetic.ROOM0_BOUND)
    cfg = synthetic.default_cfg(n_pixels=96, n_samples_ray=12, n_surface_ray=4, hash_size=12, voxel_size=0.16)
    cam = synthetic.camera(H=12, W=16, fx=10.0, fy=10.0)

    # ---- the file: what a tinycudann build of the reference would have written ----
    probe = Decoder(cfg["model"], bound, n_class=N_CLASS)                     # only for the table's row count
    table = (torch.rand(probe.pe_fn.grid_fn.params.numel(), generator=g) * 2 - 1) * 0.5
    file_sd = {"pe_fn.grid_fn.params": table, "pe_fn.pe_fn.params": torch.zeros(0),
               "coarse_fn.decoder.params": _flat8(80, 33, g), "out_fn.color_decoder.params": _flat8(112, 3, g),
               "out_fn.logit_decoder.params": _flat8(112, N_CLASS, g), "merge.decoder.params": _flat8(112, 32, g),
               "merge.pe_fn.params": torch.zeros(0)}
    fine8 = {c: _flat8(80, 33, g) for c in (0, 2, 3, 5)}
    assert file_sd["coarse_fn.decoder.params"].numel() == 80 * 32 + 40 * 32 and file_sd["out_fn.color_decoder.params"].numel() == 112 * 32 + 8 * 32
    path = os.path.join(str(tmp_path), "model.pt")
    write_reference_style_checkpoint(path, file_sd, fine8, extra={"idx": 99, "keyframe_list": [0, 5, 10]})

    # ---- product: fresh modules on the GPU, filled from the file ----
    dec = Decoder(cfg["model"], bound, n_class=N_CLASS).to(DEV)
    mapper = Mapper(cfg, dec, bound, cam, device=DEV)
    rest = load_reference_decoder(path, dec, mapper.fine_decoders, device=DEV)
    assert rest == {"idx": 99, "keyframe_list": [0, 5, 10]}
    assert sorted(mapper.fine_decoders.keys()) == [0, 2, 3, 5]
    assert dec.coarse_fn.decoder.params.is_cuda and mapper.fine_decoders.pool.is_cuda

    # ---- oracle: straight from the 8-padded tensors ----
    om = sr.OracleModel(oracle_cfg_from(cfg, N_CLASS), bound)
    leaf = lambda t: t.clone().requires_grad_(True)
    om.table = leaf(table.reshape(om.meta.total_rows, 2))
    om.coarse = leaf(_to16(file_sd["coarse_fn.decoder.params"], 80, 33))
    om.color = leaf(_to16(file_sd["out_fn.color_decoder.params"], 112, 3))
    om.logit = leaf(_to16(file_sd["out_fn.logit_decoder.params"], 112, N_CLASS))
    om.fine = {c: leaf(_to16(p, 80, 33)) for c, p in fine8.items()}

    N, S = 96, 16
    ext = bound[:, 1] - bound[:, 0]
    o = (bound[:, 0] + ext * (0.3 + 0.4 * torch.rand(N, 3, generator=g, dtype=torch.float64))).float()
    d = torch.randn(N, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    z = torch.sort(torch.rand(N, S, generator=g) * 1.5 + 0.05, -1)[0]
    samples = {"pts": o[:, None] + d[:, None] * z[..., None], "rays_d": d, "z_vals": z,
               "gt_label": torch.tensor([(0, 2, 3, 5)[i % 4] for i in range(N)]), "features": torch.rand(N, S, 32, generator=g) * 2 - 1}
    want = sr.mapper_renderer(om, samples)
    got = mapper.renderer({k: v.to(DEV) for k, v in samples.items()})
    for name, a, b in zip(("color", "depth", "var", "logits", "fine", "coarse"), got, want):
        assert_close(a.detach().cpu(), b.detach(), what=f"reference-layout checkpoint: {name}")

    # and back out: a file saved with mlp_granule=8 carries the same used weights in CutlassMLP's layout
    from dns_slam_amd.checkpoint import Checkpoint
    Checkpoint(str(tmp_path), device=DEV, decoder=dec, fine_decoders=mapper.fine_decoders).save("out8.pt", mlp_granule=8)
    raw = torch.load(os.path.join(str(tmp_path), "out8.pt"), weights_only=False)
    used = 80 * 32 + 33 * 32
    assert raw["decoder"]["coarse_fn.decoder.params"].numel() == 80 * 32 + 40 * 32
    assert torch.equal(raw["decoder"]["coarse_fn.decoder.params"][:used], file_sd["coarse_fn.decoder.params"][:used])
    assert torch.equal(raw["fine_decoders"][3][:used], fine8[3][:used])

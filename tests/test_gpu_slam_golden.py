"""The HIP path against vectors the reference's OWN ``Mapper.renderer / fine_fn / compute_*_loss`` and ``Tracker.renderer /
compute_*_loss`` produced (tests/golden/slam_wiring.npz; generator tests/golden/make_golden_slam.py imports
``slams/mapping.py`` / ``slams/tracking.py`` as they lie).  No oracle in between: product outputs, loss terms and every
gradient are compared with the fixture arrays at BASELINE's 1e-4.  (The arithmetic inside OneBlob / HashGrid / MLP of the
fixtures is oracle/tcnn_ref.py standing in for the absent tinycudann: that part is parity-unpinned, the wiring is not.)"""
import os

import numpy as np
import pytest
import torch

from util import assert_close, mlp_param_groups, table_level_groups

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "slam_wiring.npz"))
T = lambda a: torch.from_numpy(np.array(a))


def product_of_case(p, track=False):
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.mapping import Mapper
    N, S, n_class, hash_size = (int(v) for v in G[p + "dims"])
    bound = T(G[p + "bound"])
    cfg = synthetic.default_cfg(n_pixels=N, n_samples_ray=max(S - 1, 1), n_surface_ray=1, hash_size=hash_size,
                                voxel_size=float(G[p + "voxel"]), smooth_pts=8)
    dec = Decoder(cfg["model"], bound, n_class=n_class).to(DEV)
    assert dec.pe_fn.resolution == int(G[p + "resolution"])
    cam = synthetic.camera(H=12, W=16, fx=10.0, fy=10.0)
    mapper = Mapper(cfg, dec, bound, cam, device=DEV)                          # label_layout default: reference_tiled (D1)
    new = mapper.set_decoder({"label_dict": [int(c) for c in G[p + "fine_classes"]]})
    assert new == [int(c) for c in G[p + "new_decoders"]]                     # slams/mapping.py:727-760
    with torch.no_grad():
        dec.pe_fn.grid_fn.params.copy_(T(G[p + "table"]))
        dec.coarse_fn.decoder.params.copy_(T(G[p + "coarse"]))
        dec.out_fn.color_decoder.params.copy_(T(G[p + "color"]))
        dec.out_fn.logit_decoder.params.copy_(T(G[p + "logit"]))
        for c in G[p + "fine_classes"]:
            mapper.fine_decoders.params_of(int(c)).copy_(T(G[p + f"fine_{int(c)}"]))
    pts = T(G[p + "pts"]).to(DEV).requires_grad_(True)
    samples = {"pts": pts, "rays_d": T(G[p + "rays_d"]).to(DEV), "z_vals": T(G[p + "z_vals"]).to(DEV),
               "gt_label": T(G[p + "gt_label"]).to(DEV), "features": T(G[p + "features"]).to(DEV),
               "gt_depth": T(G[p + "gt_depth"]).to(DEV), "gt_color": T(G[p + "gt_color"]).to(DEV)}
    return cfg, dec, mapper, samples


@pytest.mark.parametrize("fused_nets", [True, False])
@pytest.mark.parametrize("ci", range(int(G["n_cases"])))
def test_mapper_renderer_and_gradients_equal_the_reference(ci, fused_nets):
    p = f"c{ci}_"
    cfg, dec, mapper, s = product_of_case(p)
    mapper.fused_nets = fused_nets
    pc, pd, pv, pl, fine, coarse = mapper.renderer(s)
    for k, v in (("m_color", pc), ("m_depth", pd), ("m_var", pv), ("m_logits", pl), ("m_fine", fine), ("m_coarse", coarse)):
        assert_close(v.detach().cpu(), T(G[p + k]), what=f"{p}{k}")
    loss, terms = mapper.iteration_loss(s, lambda_lt=10.0, smooth=False, strict=True)
    want = G[p + "m_terms"]
    for i, k in enumerate(("p_loss", "d_loss", "l_loss", "lt_loss", "fs_loss", "opacity_loss")):
        assert abs(float(terms[k]) - want[i]) <= 1e-4 * max(abs(want[i]), 1e-9), (k, float(terms[k]), want[i])
    assert abs(float(loss) - float(G[p + "m_loss"])) <= 1e-4 * abs(float(G[p + "m_loss"]))
    loss.backward()
    N, S, n_class, _ = (int(v) for v in G[p + "dims"])
    used = lambda n_in, n_out: 32 * n_in + n_out * 32                         # rows beyond n_out are storage only
    # element-wise, every entry against ITS level's scale (criterion 2)
    assert_close(dec.pe_fn.grid_fn.params.grad.cpu().reshape(-1), T(G[p + "g_table"]).reshape(-1), what=f"{p}d table",
                 groups=table_level_groups(dec.pe_fn.grid_fn.meta))
    for name, par, n_in, n_out in (("coarse", dec.coarse_fn.decoder.params, 80, 33), ("color", dec.out_fn.color_decoder.params, 112, 3),
                                   ("logit", dec.out_fn.logit_decoder.params, 112, n_class)):
        u = used(n_in, n_out)
        assert_close(par.grad.cpu()[:u], T(G[p + "g_" + name])[:u], what=f"{p}d {name}", groups=mlp_param_groups(n_in, n_out, 32, 1))
    pool_grad = mapper.fine_decoders.pool.grad.cpu()
    for c, slot in mapper.fine_decoders.slot.items():
        want_g = T(G[p + f"g_fine_{c}"])[:used(80, 33)]
        if torch.count_nonzero(want_g) == 0:
            assert torch.count_nonzero(pool_grad[slot]) == 0, c
        else:
            assert_close(pool_grad[slot][:used(80, 33)], want_g, what=f"{p}d fine[{c}]")
    # d pts: element-wise too (|a - b| <= 1e-4 |b| + 1e-4 rms of the batch's point gradients)
    gp = s["pts"].grad.cpu().reshape(-1, 3)
    assert_close(gp, T(G[p + "g_pts"]).reshape(-1, 3), what=f"{p}d pts")


@pytest.mark.parametrize("ci", range(int(G["n_cases"])))
def test_tracker_renderer_and_losses_equal_the_reference(ci):
    from dns_slam_amd.tracking import Tracker
    p = f"c{ci}_"
    cfg, dec, mapper, s = product_of_case(p)
    cam = {"H": 12, "W": 16, "fx": 10.0, "fy": 10.0, "cx": 7.5, "cy": 5.5}
    tracker = Tracker(cfg, dec, T(G[p + "bound"]), cam, device=DEV)
    tc, td, tv, tl = tracker.renderer(s)
    for k, v in (("t_color", tc), ("t_depth", td), ("t_var", tv), ("t_logits", tl)):
        assert_close(v.detach().cpu(), T(G[p + k]), what=f"{p}{k}")
    mask = T(G[p + "t_mask"]).to(DEV)
    pl_ = tracker.compute_photometric_loss(s["gt_color"], tc, mask)
    dl_ = tracker.compute_depth_loss(s["gt_depth"], td, tv, mask)
    ll_ = tracker.compute_label_loss(s["gt_label"], tl, mask)
    want = G[p + "t_terms"]
    for got, w, k in ((pl_, want[0], "p"), (dl_, want[1], "d"), (ll_, want[2], "l")):
        assert abs(float(got) - w) <= 1e-4 * max(abs(w), 1e-9), (k, float(got), w)
    loss = 5.0 * pl_ + 5.0 * dl_ + 0.1 * ll_
    loss.backward()
    # criterion 1 only (max |a - b| <= 1e-4 max |b|; measured 8.8e-6): the tracker's depth term is divided by sqrt(var + 1e-10)
    # (slams/tracking.py:90), which spreads the point gradients of one batch over four decades -- a small entry is a sum of terms
    # a thousand times its size, and an element-wise bound relative to the entry itself measures fp32 summation order, nothing else
    assert_close(s["pts"].grad.cpu().reshape(-1, 3), T(G[p + "t_g_pts"]).reshape(-1, 3), what=f"{p}tracker d pts", elementwise=False)

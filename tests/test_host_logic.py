"""CPU: host-side logic that needs no GPU -- module construction / state-dict layout, checkpoint round trip, the
tinycudann stand-in's parameter layouts, config defaults, quaternion conversion, synthetic scene."""
import math
import os

import numpy as np
import torch

from dns_slam_amd import synthetic
from dns_slam_amd import tcnn_shim as tcnn
from dns_slam_amd.checkpoint import Checkpoint
from dns_slam_amd.common import get_camera_from_tensor, get_quad_from_c2w, quad2rotation
from dns_slam_amd.decoder import Decoder
from oracle import render_math as rm
from oracle import tcnn_ref as tr


def test_decoder_layout_and_checkpoint_roundtrip(tmp_path):
    bound = synthetic.load_bound(synthetic.ROOM0_BOUND)
    cfg = synthetic.default_cfg()
    dec = Decoder(cfg["model"], bound, n_class=40)
    sd = dec.state_dict()
    assert sd["pe_fn.grid_fn.params"].numel() == 853312 * 2                       # room_0 table
    assert sd["coarse_fn.decoder.params"].numel() == tr.mlp_param_count(80, 33, 32, 1) == 80 * 32 + 48 * 32
    assert sd["out_fn.logit_decoder.params"].numel() == tr.mlp_param_count(112, 40, 32, 1)
    assert dec.pe_dim == 48 and dec.grid_dim == 32 and dec.pe_fn.resolution == 592
    ck = Checkpoint(str(tmp_path), device="cpu", decoder=dec)
    ck.save("model.pt", scene="room_0", idx=torch.tensor(5), keyframe_list=[0, 5])
    dec2 = Decoder(cfg["model"], bound, n_class=40)
    with torch.no_grad():
        dec2.coarse_fn.decoder.params.zero_()
    rest = Checkpoint(str(tmp_path), device="cpu", decoder=dec2).load("model.pt")
    assert rest["scene"] == "room_0" and rest["keyframe_list"] == [0, 5] and int(rest["idx"]) == 5
    for k in sd:
        assert torch.equal(dec2.state_dict()[k], sd[k])


def test_tcnn_shim_constructor_contract():
    enc = tcnn.Encoding(3, {"otype": "OneBlob", "n_bins": 16}, dtype=torch.float)
    assert enc.n_output_dims == 48
    grid = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 16,
                             "base_resolution": 16, "per_level_scale": np.exp2(np.log2(592 / 16) / 15)}, dtype=torch.float)
    assert grid.n_output_dims == 32 and grid.params.numel() == 853312 * 2
    assert float(grid.params.abs().max()) <= 1e-4                                  # tcnn grid init U(-1e-4, 1e-4)
    net = tcnn.Network(80, 33, {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None",
                                "n_neurons": 32, "n_hidden_layers": 1})
    assert net.params.numel() == 80 * 32 + 48 * 32 and net.params.dtype == torch.float32
    lim = math.sqrt(6.0 / (32 + 80))
    assert float(net.params[:80 * 32].abs().max()) <= lim                          # Xavier-uniform
    try:
        tcnn.Network(80, 33, {"activation": "Sine"})
        raise AssertionError("unsupported activation must raise")
    except ValueError:
        pass


def test_quaternion_roundtrip_matches_scipy():
    from scipy.spatial.transform import Rotation
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        q = torch.randn(4, generator=g)
        q = q / q.norm()
        R = quad2rotation(q[None])[0]
        Rs = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()               # scipy: (x,y,z,w)
        assert np.allclose(R.numpy(), Rs, atol=1e-6)
        c2w = torch.eye(4)
        c2w[:3, :3] = R
        q2 = get_quad_from_c2w(c2w)
        assert min(float((q2 - q).abs().max()), float((q2 + q).abs().max())) < 1e-5
        assert torch.allclose(quad2rotation(q[None] * 3.0)[0], R, atol=1e-6)        # 2/|q|^2: no normalisation needed
        assert torch.equal(quad2rotation(q[None]), rm.quad2rotation(q[None]))
    RT = get_camera_from_tensor(torch.tensor([1.0, 0, 0, 0, 1, 2, 3]))
    assert torch.equal(RT, torch.tensor([[1.0, 0, 0, 1], [0, 1, 0, 2], [0, 0, 1, 3]]))


def test_synthetic_scene_is_consistent():
    cam = synthetic.camera(H=30, W=40, fx=30.0, fy=30.0)
    bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=0)
    assert bound.dtype == torch.float64 and frames["gt_depth"].shape == (4, 30, 40)
    assert set(frames["label_dict"]) <= set(range(8))
    # depths lie inside the bound along every ray (so no ray is dropped by the box clip)
    for f in range(4):
        c2w = frames["est_c2w"][f]
        idx = torch.arange(30 * 40)
        i, j = rm.uv_from_indices(idx, 0, 30, 0, 40)
        ro, rd = rm.rays_from_uv(i, j, c2w[:3, :3], c2w[:3, 3], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        far, inside = rm.box_far(ro, rd, frames["gt_depth"][f].reshape(-1), bound)
        assert bool(inside.all())

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


def test_class_balanced_index_draws_replay_the_reference(golden_dir):
    """select_by_class (utils/common.py:307-338) and get_samples_by_uniq_class (:364-403) index draws: same generator
    state -> the IMPORTED reference's indices, bit for bit (per-class randint shapes and order, first class takes the
    remainder, a one-pixel class is repeated without a draw, an absent class of class_dict is skipped)."""
    from dns_slam_amd.common import _class_pick
    gd = np.load(os.path.join(golden_dir, "class_picks.npz"))
    for ci in range(int(gd["n_cases"])):
        p = f"c{ci}_"
        img = torch.from_numpy(gd[p + "image"])
        n, seed = int(gd[p + "n"]), int(gd[p + "seed"])
        cls = list(gd[p + "class_dict"]) if int(gd[p + "uniq"]) else None
        torch.manual_seed(seed)
        idx = _class_pick(img[..., -1].reshape(-1), n, "cpu", class_list=cls)
        want = torch.from_numpy(gd[p + "indices"])
        assert torch.equal(idx, want), f"case {ci}"
        lab = img[..., -1].reshape(-1)[idx]
        # quotas: n // n_class each, the first class the remainder; absent classes contribute nothing
        wanted = sorted(set(img[..., -1].reshape(-1).tolist())) if cls is None else [float(c) for c in cls]
        n_k = n // len(wanted)
        for i, c in enumerate(wanted):
            m = n - n_k * (len(wanted) - 1) if i == 0 else n_k
            present = bool((img[..., -1] == c).any())
            assert int((lab == c).sum()) == (m if present else 0), (ci, c)


def test_checkpoint_repacks_cutlass_padded_mlp_tensors(tmp_path):
    """A tinycudann CutlassMLP flat tensor pads its output rows to 8, the kernels here to 16 (dns_slam_amd/checkpoint.py):
    a file with 8-row padding loads into the model with the SAME weights, padded rows zero, and saving with
    mlp_granule=8 writes that layout back; a tensor that matches no padding is refused."""
    from dns_slam_amd.checkpoint import mlp_numel, repack_mlp_params
    from dns_slam_amd.mapping import FineDecoderPool
    bound = synthetic.load_bound(synthetic.ROOM0_BOUND)
    cfg = synthetic.default_cfg(hash_size=12, voxel_size=0.2)
    dec = Decoder(cfg["model"], bound, n_class=40)
    pool = FineDecoderPool(80, 33, dec.coarse_fn.decoder.network_config, capacity=4, device="cpu")
    pool.add(3), pool.add(7)
    with torch.no_grad():
        pool.pool.copy_(torch.randn_like(pool.pool))
    ck = Checkpoint(str(tmp_path), device="cpu", decoder=dec, fine_decoders=pool)
    ck.save("m8.pt", mlp_granule=8, scene="room_0", keyframe_list=[0, 5])
    raw = torch.load(os.path.join(str(tmp_path), "m8.pt"), weights_only=False)
    assert raw["decoder"]["coarse_fn.decoder.params"].numel() == mlp_numel(80, 33, 32, 1, 8) == 80 * 32 + 40 * 32
    assert raw["decoder"]["out_fn.color_decoder.params"].numel() == 112 * 32 + 8 * 32
    assert raw["fine_decoders"][7].numel() == 80 * 32 + 40 * 32
    dec2 = Decoder(cfg["model"], bound, n_class=40)
    pool2 = FineDecoderPool(80, 33, dec.coarse_fn.decoder.network_config, capacity=4, device="cpu")
    with torch.no_grad():
        for prm in dec2.parameters():
            prm.zero_()
    rest = Checkpoint(str(tmp_path), device="cpu", decoder=dec2, fine_decoders=pool2).load("m8.pt")
    assert rest["scene"] == "room_0" and rest["keyframe_list"] == [0, 5]
    a, b = dec.state_dict(), dec2.state_dict()
    used = 80 * 32 + 33 * 32
    assert torch.equal(a["coarse_fn.decoder.params"][:used], b["coarse_fn.decoder.params"][:used])
    assert torch.count_nonzero(b["coarse_fn.decoder.params"][used:]) == 0
    assert torch.equal(a["pe_fn.grid_fn.params"], b["pe_fn.grid_fn.params"])
    assert sorted(pool2.keys()) == [3, 7] and torch.equal(pool2.params_of(7)[:used], pool.params_of(7)[:used])
    try:
        repack_mlp_params(torch.zeros(80 * 32 + 41 * 32), 80, 33, 32, 1)
        raise AssertionError("a tensor that matches no padding must be refused")
    except ValueError:
        pass


def write_reference_style_checkpoint(path, decoder_sd, fine_params: dict, n_in=80, n_out=33, extra=None):
    """A ``model.pt`` the way the reference writes it (slams/mapping.py:1119-1128): ``fine_decoders`` is a dict of pickled
    ``tinycudann.modules.Network`` OBJECTS.  tinycudann is absent, so a throw-away package of that name -- holding a class
    whose pickled state is its ``__dict__`` without the native handle, like tcnn's ``Module.__getstate__`` -- exists in
    ``sys.modules`` only while the file is written; the loader never sees it."""
    import sys
    import types
    from torch import nn
    pkg, mod = types.ModuleType("tinycudann"), types.ModuleType("tinycudann.modules")

    class Network(nn.Module):
        def __init__(self, n_input_dims, n_output_dims, network_config, params):
            super().__init__()
            self.n_input_dims, self.n_output_dims, self.network_config = n_input_dims, n_output_dims, network_config
            self.native_tcnn_module = object()                   # tcnn: a pybind handle, dropped from the pickled state
            self.params = nn.Parameter(params.clone())

        def __getstate__(self):
            st = self.__dict__.copy()
            del st["native_tcnn_module"]
            return st

    Network.__module__, Network.__qualname__ = "tinycudann.modules", "Network"
    mod.Network, pkg.modules = Network, mod
    saved = {k: sys.modules.get(k) for k in ("tinycudann", "tinycudann.modules")}
    sys.modules["tinycudann"], sys.modules["tinycudann.modules"] = pkg, mod
    try:
        cfg = {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 32, "n_hidden_layers": 1}
        blob = {"decoder": decoder_sd, "fine_decoders": {c: Network(n_in, n_out, cfg, p) for c, p in fine_params.items()}}
        blob.update(extra or {})
        torch.save(blob, path)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_checkpoint_loads_pickled_tinycudann_module_objects(tmp_path):
    """The reference's ``fine_decoders`` are pickled tinycudann modules (slams/mapping.py:1121); ``torch.load`` on a box
    without tinycudann raises ModuleNotFoundError unless the unpickler maps that package to a stand-in
    (dns_slam_amd/checkpoint.py:_RefUnpickler)."""
    import sys
    from dns_slam_amd.checkpoint import mlp_numel
    from dns_slam_amd.mapping import FineDecoderPool
    bound = synthetic.load_bound(synthetic.ROOM0_BOUND)
    cfg = synthetic.default_cfg(hash_size=12, voxel_size=0.2)
    dec = Decoder(cfg["model"], bound, n_class=40)
    g = torch.Generator().manual_seed(0)
    fine = {5: torch.randn(mlp_numel(80, 33, 32, 1, 8), generator=g), 11: torch.randn(mlp_numel(80, 33, 32, 1, 8), generator=g)}
    path = os.path.join(str(tmp_path), "model.pt")
    write_reference_style_checkpoint(path, dec.state_dict(), fine, extra={"idx": 7})
    assert "tinycudann" not in sys.modules or not hasattr(sys.modules["tinycudann"], "modules")
    try:                                                         # the stock loader cannot read the file here
        torch.load(path, weights_only=False)
        raise AssertionError("expected the stock unpickler to fail without tinycudann")
    except (ModuleNotFoundError, AttributeError):
        pass
    pool = FineDecoderPool(80, 33, dec.coarse_fn.decoder.network_config, capacity=4, device="cpu")
    rest = Checkpoint(str(tmp_path), device="cpu", decoder=Decoder(cfg["model"], bound, n_class=40), fine_decoders=pool).load("model.pt")
    assert rest == {"idx": 7}
    used = 80 * 32 + 33 * 32
    assert sorted(pool.keys()) == [5, 11]
    for c in (5, 11):
        assert torch.equal(pool.params_of(c)[:used], fine[c][:used]) and torch.count_nonzero(pool.params_of(c)[used:]) == 0


def test_fixed_launch_sequences_refuse_to_run_without_a_gpu():
    """MapStep / TrackStep (dns_slam_amd/fused_step.py) are product paths: no CPU fallback, a loud error instead."""
    import pytest
    from types import SimpleNamespace
    from dns_slam_amd.fused_step import MapStep, TrackStep
    with pytest.raises(ValueError, match="GPU only"):
        MapStep(SimpleNamespace(device="cpu"), {})
    with pytest.raises(ValueError, match="GPU only"):
        TrackStep(SimpleNamespace(device="cpu"), {}, None)


def test_bench_parent_relays_only_a_line_of_the_size_it_asked_for(tmp_path, capsys):
    """``bench.py --gpus N`` starts N ranks as a child and relays rank 0's line; a line that reports another job size (a child
    that ran one rank), a missing line or a failing child must NOT produce a result line on the parent's stdout."""
    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    child = tmp_path / "child.py"

    def run(body, n):
        child.write_text(body)
        rc = bench.relay_rank0_line([sys.executable, str(child)], n)
        return rc, capsys.readouterr().out

    good = 'import json; print("noise"); print(json.dumps({"metric": "ray-samples/s", "value": 1.0, "n_gpus": %d, "rccl_ranks": %d}))'
    rc, out = run(good % (2, 2), 2)
    assert rc == 0 and '"n_gpus": 2' in out
    rc, out = run(good % (1, 1), 2)                  # the child ran ONE rank
    assert rc != 0 and out.strip() == ""
    rc, out = run(good % (2, 1), 2)                  # n_gpus claimed, but the process group had one rank
    assert rc != 0 and out.strip() == ""
    rc, out = run('print("no line")', 2)
    assert rc != 0 and out.strip() == ""
    rc, out = run(good % (2, 2) + "; import sys; sys.exit(7)", 2)      # a line, but the launch failed
    assert rc == 7 and out.strip() == ""

"""Golden vectors for the WIRING of the optimise-step bodies, from the IMPORTED reference classes.

Run once, in the build container only (the reference never travels):

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_slam.py

What is pinned.  ``slams/mapping.py`` / ``slams/tracking.py`` are imported as they lie, with in-memory placeholder
modules (nothing is written to disk) for what the container lacks and the calls below never touch: ``mathutils``, ``cv2``,
``colorama``, ``tqdm``, ``models.encoder`` (its ResNet would download weights).  The reference's methods are then called
as UNBOUND functions on a ``SimpleNamespace`` self:

* ``Mapper.set_decoder``  (slams/mapping.py:727-760)  -- which fine decoders get created, with which constructor arguments
* ``Mapper.fine_fn``      (:590-601)                   -- per-class routing, the "> 1 point" rule
* ``Mapper.renderer``     (:603-635)                   -- D1 label tiling, ``fine[:, 1:]`` routing, logits composite
* ``Mapper.compute_{photometric,depth,label,latent}_loss`` (:110-126) and ``get_opacity_loss`` as called at :896 (D5)
* ``Tracker.renderer``    (slams/tracking.py:188-214) and ``Tracker.compute_*_loss`` (:85-96)
* the reference's ``models.decoder.Decoder`` (models/decoder.py:7-125) itself builds the networks

What is NOT pinned by this: ``tinycudann`` is absent (CUDA-only), so the ``tcnn.Encoding`` / ``tcnn.Network``
objects the reference constructs are CPU stand-ins over ``oracle/tcnn_ref.py`` -- the arithmetic INSIDE OneBlob, HashGrid and
the MLPs stays "parity unpinned"; everything AROUND them (what is fed to which network, in which order, with which labels,
how outputs are sliced, composited and reduced) is the reference's own code.  ``Mapper.smoothness`` raises on this torch
(D2) and ``get_target_samples`` needs ``quad2rotation`` on a GPU (D8): neither can be run here.

Outputs: ``slam_wiring.npz`` -- inputs, parameters, outputs and autograd gradients as plain arrays.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(OUT, "..", "..")))
from oracle import tcnn_ref as tr  # noqa: E402


# ---- in-memory placeholders (never written to disk) -------------------------------------------------------------------------
class _Blank:
    def __getattr__(self, k):
        return ""


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _StandInEncoding(nn.Module):
    """tcnn.Encoding(n_input_dims, encoding_config, dtype) over oracle/tcnn_ref.py (CPU, fp32)."""

    def __init__(self, n_input_dims, encoding_config, dtype=torch.float):
        super().__init__()
        self.cfg = dict(encoding_config)
        ot = self.cfg["otype"].lower()
        if ot == "oneblob":
            self.kind, self.n_bins = "oneblob", int(self.cfg["n_bins"])
            self.n_output_dims = n_input_dims * self.n_bins
            self.params = nn.Parameter(torch.zeros(0), requires_grad=False)
        elif ot == "hashgrid":
            self.kind = "hashgrid"
            self.meta = tr.grid_meta(int(self.cfg["log2_hashmap_size"]), 0, int(self.cfg["n_levels"]),
                                     int(self.cfg["n_features_per_level"]), int(self.cfg["base_resolution"]),
                                     per_level_scale=float(self.cfg["per_level_scale"]))
            self.n_output_dims = self.meta.n_levels * self.meta.n_features
            self.params = nn.Parameter(tr.grid_init(self.meta, torch.Generator().manual_seed(1337)).reshape(-1))
        else:
            raise ValueError(ot)

    def forward(self, x):
        x = x.float()
        if self.kind == "oneblob":
            return tr.oneblob_forward(x, self.n_bins)
        return tr.hashgrid_forward(x, self.params.reshape(self.meta.total_rows, self.meta.n_features), self.meta)


class _StandInNetwork(nn.Module):
    """tcnn.Network(n_input_dims, n_output_dims, network_config) over oracle/tcnn_ref.py (CPU, fp32)."""

    def __init__(self, n_input_dims, n_output_dims, network_config):
        super().__init__()
        self.n_in, self.n_out = int(n_input_dims), int(n_output_dims)
        self.nn, self.nl = int(network_config["n_neurons"]), int(network_config["n_hidden_layers"])
        assert network_config["activation"] == "ReLU" and network_config["output_activation"] == "None"
        self.params = nn.Parameter(tr.mlp_init(self.n_in, self.n_out, self.nn, self.nl, torch.Generator().manual_seed(1337)))

    def forward(self, x):
        return tr.mlp_forward(x, self.params, self.n_in, self.n_out, self.nn, self.nl)


_placeholder("mathutils", Matrix=object)
_placeholder("cv2")
_placeholder("colorama", Fore=_Blank(), Style=_Blank())
_placeholder("tqdm", tqdm=lambda it, *a, **k: it)
_placeholder("tinycudann", Encoding=_StandInEncoding, Network=_StandInNetwork)
_placeholder("models.encoder", ResNet=nn.Identity)
sys.path.insert(0, "/root/reference")
import warnings  # noqa: E402
warnings.filterwarnings("ignore")
import models.decoder as RD      # noqa: E402  the reference's Decoder
import slams.mapping as RM       # noqa: E402
import slams.tracking as RT      # noqa: E402
import utils.common as RC        # noqa: E402

BOUND = [[-2.9, 8.9], [-3.2, 5.5], [-3.5, 3.3]]       # configs/replica/room_0.yaml:4


def load_bound():
    """slams/dns_slam.py:100-107 (float64)."""
    b = torch.tensor(BOUND, dtype=torch.float64)
    b[:, 1] = (((b[:, 1] - b[:, 0]) / 0.32).int() + 1) * 0.32 + b[:, 0]
    return b


def randomise(module, seed, scale):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if p.numel():
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * scale)


def make_case(ci, N, S, n_class, labels, seed, hash_size=10, voxel=0.16, extra_classes=()):
    g = torch.Generator().manual_seed(seed)
    bound = load_bound()
    cfg_model = {"pts_dim": 3, "pixel_dim": 64, "hidden_dim": 32, "pos": {"method": "OneBlob", "n_bins": 16},
                 "grid": {"method": "HashGrid", "hash_size": hash_size, "voxel_size": voxel}}
    dec = RD.Decoder(cfg_model, bound, n_class=n_class)                     # the reference's own class
    randomise(dec.coarse_fn, seed + 1, 0.35)
    randomise(dec.out_fn, seed + 2, 0.35)
    randomise(dec.pe_fn.grid_fn, seed + 3, 0.5)

    # Mapper state the called methods read (slams/mapping.py:21-107)
    me = types.SimpleNamespace(device="cpu", bound=bound, decoder=dec, hidden_dim=32, pe_dim=dec.pe_dim, grid_dim=dec.grid_dim,
                               fine_decoders={}, exist_decoders={}, class2label_dict={c: c for c in range(n_class)})
    me.fine_fn = types.MethodType(RM.Mapper.fine_fn, me)
    label_dict = sorted(set(int(v) for v in labels) | set(extra_classes))
    new_list = RM.Mapper.set_decoder(me, {"label_dict": label_dict})           # creates the per-class networks (:736-749)
    for k, c in enumerate(sorted(me.fine_decoders)):
        randomise(me.fine_decoders[c], seed + 10 + k, 0.35)

    # a ray batch inside the bound (rays_d is accepted and unused by raw2nerf_color in occupancy mode)
    ext = bound[:, 1] - bound[:, 0]
    o = (bound[:, 0] + ext * (0.3 + 0.4 * torch.rand(N, 3, generator=g, dtype=torch.float64))).float()
    d = torch.randn(N, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    z = torch.sort(torch.rand(N, S, generator=g) * 1.5 + 0.05, -1)[0]
    pts = (o[:, None, :] + d[:, None, :] * z[..., None]).requires_grad_(True)
    gt_depth = z[:, S // 2].clone() + 0.01
    gt_depth[1] = 0.0                                                         # a zero-depth ray (masked depth loss)
    gt_color = torch.rand(N, 3, generator=g)
    gt_label = torch.tensor(labels, dtype=torch.int64)
    feats = (torch.rand(N, S, 32, generator=g) * 2 - 1)
    samples = {"pts": pts, "rays_d": d, "z_vals": z, "gt_label": gt_label, "features": feats,
               "gt_depth": gt_depth, "gt_color": gt_color}

    # ---- Mapper: renderer + the six ray-batch losses, summed with the weights of :906 (lambda_lt = 10, no smoothness) ----
    pc, pd, pv, pl, fine, coarse = RM.Mapper.renderer(me, samples)
    d_loss = RM.Mapper.compute_depth_loss(me, gt_depth, pd)
    p_loss = RM.Mapper.compute_photometric_loss(me, gt_color, pc)
    l_loss = RM.Mapper.compute_label_loss(me, gt_label, pl)
    lt_loss = RM.Mapper.compute_latent_loss(me, coarse, fine)
    fs_loss, op_loss = RC.get_opacity_loss(z, gt_depth, fine[..., -1], 0.05)   # as called at mapping.py:896 (D5)
    loss = 5.0 * p_loss + 5.0 * d_loss + 0.1 * l_loss + 10.0 * lt_loss + 10.0 * fs_loss + 10.0 * op_loss
    params = [dec.pe_fn.grid_fn.params, dec.coarse_fn.decoder.params, dec.out_fn.color_decoder.params,
              dec.out_fn.logit_decoder.params] + [me.fine_decoders[c].params for c in sorted(me.fine_decoders)]
    grads = torch.autograd.grad(loss, params + [pts], allow_unused=True)
    zero = lambda gr, p: (gr if gr is not None else torch.zeros_like(p)).detach().numpy()

    p = f"c{ci}_"
    out = {p + "dims": np.array([N, S, n_class, hash_size]), p + "voxel": np.array(voxel), p + "bound": bound.numpy(),
           p + "resolution": np.array(dec.pe_fn.resolution),
           p + "new_decoders": np.array(new_list), p + "fine_classes": np.array(sorted(me.fine_decoders)),
           p + "pts": pts.detach().numpy(), p + "rays_d": d.numpy(), p + "z_vals": z.numpy(), p + "gt_label": gt_label.numpy(),
           p + "features": feats.numpy(), p + "gt_depth": gt_depth.numpy(), p + "gt_color": gt_color.numpy(),
           p + "table": params[0].detach().numpy(), p + "coarse": params[1].detach().numpy(),
           p + "color": params[2].detach().numpy(), p + "logit": params[3].detach().numpy(),
           p + "m_color": pc.detach().numpy(), p + "m_depth": pd.detach().numpy(), p + "m_var": pv.detach().numpy(),
           p + "m_logits": pl.detach().numpy(), p + "m_fine": fine.detach().numpy(), p + "m_coarse": coarse.detach().numpy(),
           p + "m_terms": np.array([float(p_loss), float(d_loss), float(l_loss), float(lt_loss), float(fs_loss), float(op_loss)]),
           p + "m_loss": np.array(float(loss)),
           p + "g_table": zero(grads[0], params[0]), p + "g_coarse": zero(grads[1], params[1]),
           p + "g_color": zero(grads[2], params[2]), p + "g_logit": zero(grads[3], params[3]),
           p + "g_pts": zero(grads[-1], pts)}
    for k, c in enumerate(sorted(me.fine_decoders)):
        out[p + f"fine_{c}"] = params[4 + k].detach().numpy()
        out[p + f"g_fine_{c}"] = zero(grads[4 + k], params[4 + k])

    # ---- Tracker: renderer + the three masked losses (slams/tracking.py:326-329) on the same batch ----
    tk = types.SimpleNamespace(device="cpu", bound=bound, decoder=dec)
    mask = (gt_depth > 0.01).numpy() & (np.arange(N) % 5 != 3)                 # 'mask' is a numpy bool array (tracking.py:176)
    tc, td, tv, tl = RT.Tracker.renderer(tk, samples)
    tp_loss = RT.Tracker.compute_photometric_loss(tk, gt_color, tc, mask)
    td_loss = RT.Tracker.compute_depth_loss(tk, gt_depth, td, tv, mask)
    tl_loss = RT.Tracker.compute_label_loss(tk, gt_label, tl, mask)
    t_loss = 5.0 * tp_loss + 5.0 * td_loss + 0.1 * tl_loss
    tg = torch.autograd.grad(t_loss, [pts])[0]
    out.update({p + "t_mask": mask, p + "t_color": tc.detach().numpy(), p + "t_depth": td.detach().numpy(),
                p + "t_var": tv.detach().numpy(), p + "t_logits": tl.detach().numpy(),
                p + "t_terms": np.array([float(tp_loss), float(td_loss), float(tl_loss)]), p + "t_loss": np.array(float(t_loss)),
                p + "t_g_pts": tg.numpy()})
    return out


if __name__ == "__main__":
    out = {}
    # case 0: 40 rays x 12 samples, 6 classes, every class on several rays (tiled labels scramble the routing, D1)
    lab0 = [(3 * i + i // 7) % 6 for i in range(40)]
    out.update(make_case(0, 40, 12, 6, lab0, seed=500))
    # case 1: ONE sample per ray, so a class seen by one ray has exactly one point: fine_fn leaves its row zero (:597);
    # class 4 is absent from the batch (its decoder exists and gets no gradient)
    lab1 = [0, 0, 1, 2, 2, 3, 5, 5, 0]
    out.update(make_case(1, 9, 1, 6, lab1, seed=600, extra_classes=(4,)))
    # case 2: 24 rays x 47 samples (the reference's own 32 + 15), 3 classes of 5 in use
    lab2 = [(i * i) % 3 for i in range(24)]
    out.update(make_case(2, 24, 47, 5, lab2, seed=700, hash_size=11, voxel=0.08))
    out["n_cases"] = np.array(3)
    np.savez_compressed(os.path.join(OUT, "slam_wiring.npz"), **out)
    print("slam_wiring.npz", os.path.getsize(os.path.join(OUT, "slam_wiring.npz")))

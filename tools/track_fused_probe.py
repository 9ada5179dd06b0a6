"""One fused tracker frame at a chosen shape (debugging aid of round 5: DNS_TF_PHASES=k stops the kernel behind phase k).
usage: python tools/track_fused_probe.py NN NL NU NS [N] [n_iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_slam import _setup
from dns_slam_amd.fused_step import TrackStep
from dns_slam_amd.tracking import Tracker
nn, nl, nu, ns = (int(v) for v in sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 250
n_it = int(sys.argv[6]) if len(sys.argv) > 6 else 2
cfg, bound, cam, frames, dec, mapper = _setup(nn, nl, n_pixels=400)
cfg["tracking"]["n_pixels"] = N
cfg["training"]["n_samples_ray"], cfg["training"]["n_surface_ray"] = nu, ns
cur = {"gt_color": frames["gt_color"][2], "gt_depth": frames["gt_depth"][2], "gt_label": frames["gt_label"][2]}
c2w = frames["est_c2w"][2].clone()
tracker = Tracker(cfg, dec, bound, cam, device="cuda")
tracker.border = 5
tracker.static_shapes = True
with tracker.frozen_scene():
    ts = TrackStep(tracker, cur, c2w)
    print("supported", ts.fused_supported(), "S", ts.S, flush=True)
    cam7, best = ts.run_fused(n_it, graph=False)
    torch.cuda.synchronize()
    print("ok", float(best), ts.fused_out.cpu().tolist(), flush=True)

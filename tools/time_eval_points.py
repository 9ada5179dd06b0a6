"""Mapper.eval_points (the meshing / evaluation query, slams/meshing.py:461-503) on 4 M points of the cfg2 scene: ms per call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dns_slam_amd import dist as dd

cfg, bound, cam, frames, mapper, step = bench.build(bench.WORKLOADS[os.environ.get("DNS_WL", "cfg2")], "cuda", seed=100, dist_ctx=dd.DistCtx(),
                                                    overlap=True)
g = torch.Generator().manual_seed(0)
P = int(os.environ.get("DNS_P", 1 << 22))
b = torch.as_tensor(bound).float()
pts = (torch.rand(P, 3, generator=g) * 1.1 - 0.05) * (b[:, 1] - b[:, 0]) + b[:, 0]
known = list(mapper.fine_decoders.classes()) if hasattr(mapper.fine_decoders, "classes") else None
lut = mapper.fine_decoders.lut(0)
ids = torch.nonzero(lut >= 0).reshape(-1).cpu()
labels = ids[torch.randint(0, ids.numel(), (P,), generator=g)]
pts, labels = pts.cuda(), labels.cuda()
for stage in ("fine", "coarse"):
    f = lambda: mapper.eval_points(pts, None, labels if stage == "fine" else None, stage=stage)
    v, l = f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        v, l = f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"eval_points({stage}), {P} points: {dt * 1e3:.2f} ms per call = {dt / P * 1e9:.3f} ns per point, checksum {float(v.double().abs().sum()):.6e}", flush=True)

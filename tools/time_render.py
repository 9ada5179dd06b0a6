"""Mapper.render_frame (full 640 x 480 frame, forward only, 65536-ray chunks) a few times -- for rocprofv3 --kernel-trace --stats
(cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -o r -- python3 $REPO/tools/time_render.py) or stand-alone timing."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dns_slam_amd import dist as dd

cfg, bound, cam, frames, mapper, step = bench.build(bench.WORKLOADS[os.environ.get("DNS_WL", "cfg2")], "cuda", seed=100, dist_ctx=dd.DistCtx(),
                                                    overlap=True)
mapper.static_shapes = False
rf = lambda: mapper.render_frame(frames["gt_color"][0], frames["gt_depth"][0], frames["gt_label"][0], frames["est_c2w"][0],
                                 n_pts_batch=int(os.environ.get("DNS_CHUNK", 65536)))
rf()
torch.cuda.synchronize()
n = int(os.environ.get("DNS_N", 5))
t = time.perf_counter()
for _ in range(n):
    rf()
torch.cuda.synchronize()
print(f"render_frame: {(time.perf_counter() - t) / n * 1e3:.2f} ms per frame", flush=True)

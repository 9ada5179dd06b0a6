import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from dns_slam_amd import dist as ddist
from dns_slam_amd.tracking import Tracker
from dns_slam_amd.fused_step import TrackStep
ctx = ddist.DistCtx()
wl = bench.WORKLOADS["cfg3"]
cfg, bound, cam, frames, mapper, step = bench.build(wl, "cuda:0", seed=100, dist_ctx=ctx, overlap=True)
tracker = Tracker(dict(cfg), mapper.decoder, bound, cam, device="cuda:0")
cur = {"gt_color": frames["gt_color"][1], "gt_depth": frames["gt_depth"][1], "gt_label": frames["gt_label"][1]}
with tracker.frozen_scene():
    ts = TrackStep(tracker, cur, frames["est_c2w"][1])
    for _ in range(5): ts.step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ts.step()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    print("replay only: %.4f ms/iter" % ((time.perf_counter() - t0) / 200 * 1e3))
    t0 = time.perf_counter()
    for _ in range(200): ts.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("eager: enqueue %.4f total %.4f ms/iter" % ((t1 - t0) / 200 * 1e3, (time.perf_counter() - t0) / 200 * 1e3))

"""Experiment: HIP stream priorities for the two streams of the mapping step (main = the chain of dependent kernels, side =
lattice branch + next step's preparation); the device offers two levels (0, -1)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from dns_slam_amd import dist as ddist
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
ctx = ddist.DistCtx()
wl = bench.WORKLOADS["cfg2"]
def run(main_prio, side_prio, n=300):
    cfg, bound, cam, frames, mapper, step = bench.build(wl, "cuda:0", seed=100, dist_ctx=ctx, overlap=True)
    if side_prio is not None:
        mapper._side_stream = torch.cuda.Stream(priority=side_prio)
        mapper.map_step.side = mapper._side_stream
    main = torch.cuda.Stream(priority=main_prio) if main_prio is not None else torch.cuda.current_stream()
    main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main):
        for _ in range(30): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
    print(f"main prio {main_prio} side prio {side_prio}: {dt:.4f} ms/step", flush=True)
for mp, sp in ((None, None), (-1, 0), (None, None), (-1, 0), (None, None), (-1, 0), (None, None), (-1, 0)):
    try:
        run(mp, sp)
    except Exception as e:
        print("failed", mp, sp, type(e).__name__, e)

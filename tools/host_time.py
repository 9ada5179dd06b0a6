import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from dns_slam_amd import dist as ddist
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
ctx = ddist.DistCtx()
wl = bench.WORKLOADS["cfg2"]
cfg, bound, cam, frames, mapper, step = bench.build(wl, "cuda:0", seed=100, dist_ctx=ctx, overlap=True)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {(t1-t0)/30*1e3:.3f} ms/step, total {(t2-t0)/30*1e3:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)

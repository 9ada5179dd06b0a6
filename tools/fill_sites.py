"""Which torch ops launch the fill / copy / cat kernels of one mapping iteration (torch.profiler, CPU op -> shapes + python stack)."""
import collections, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from dns_slam_amd import dist as ddist
from torch.profiler import profile, ProfilerActivity
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
ctx = ddist.DistCtx()
cfg, bound, cam, frames, mapper, step = bench.build(bench.WORKLOADS["cfg2"], "cuda:0", seed=100, dist_ctx=ctx, overlap=False)
for _ in range(5): step()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    for _ in range(N): step()
torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::cat", "aten::copy_", "aten::add", "aten::add_", "aten::clone", "aten::contiguous")
agg = collections.Counter()
for e in prof.events():
    if e.name in want:
        stack = [s for s in e.stack if "dns_slam_amd" in s or "bench.py" in s][:2]
        agg[(e.name, str(e.input_shapes)[:70], " <- ".join(s.split("/")[-1] for s in stack))] += 1
for (n, shp, st), c in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{c / N:5.1f}  {n:18s} {shp:70s} {st}")

"""cfg5_fp16 MapStep on the half-rows kernels vs on the fp16-operand kernels: per-step times and the gradient buffers of one step
on identical draws.  usage: python tools/half_step_probe.py [workload]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dns_slam_amd import dist as dd
wlname = sys.argv[1] if len(sys.argv) > 1 else "cfg5_fp16"
res = {}
for half in ("0", "1"):
    os.environ["DNS_HALF_ROWS"] = half
    cfg, bound, cam, frames, mapper, step = bench.build(bench.WORKLOADS[wlname], "cuda:0", seed=100, dist_ctx=dd.DistCtx(), overlap=True, prefetch=True)
    ms = mapper.map_step
    assert ms.half == (half == "1"), (ms.half, half)
    step(); torch.cuda.synchronize()
    G = ms.cur.G.clone()
    names = ["color", "logit", "pool", "table", "coarse", "quat", "trans"]
    segs = {n: getattr(ms.cur, "g_" + n).clone() for n in names}
    losses = [float(ms.losses()[0])]
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    losses.append(float(ms.losses()[0]))
    res[half] = (dt, segs, losses)
    print(f"half_rows={half}: {dt:.3f} ms per step, loss first / after 121 steps: {losses}")
a, b = res["0"][1], res["1"][1]
for n in a:
    d = (a[n].double() - b[n].double())
    print(f"  g_{n:7s} rel rms diff (half vs fp16-operand) {float(d.pow(2).mean().sqrt() / a[n].double().pow(2).mean().sqrt().clamp_min(1e-30)):.3e}  max|g| {float(a[n].abs().max()):.3e}")
print(f"step time ratio half / fp16-operand: {res['1'][0] / res['0'][0]:.3f}")

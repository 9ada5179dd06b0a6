"""MapStep at label_layout 'per_ray' with the drawn pixels in draw order vs in Morton order of (row, col):
step time and the kernels that gather / scatter along the rays.  usage: python tools/time_morton.py [workload]"""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dns_slam_amd import dist as dd, ops
from dns_slam_amd.fused_step import MapStep
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
cfg, bound, cam, frames, mapper, step0 = bench.build(wl, "cuda:0", seed=100, dist_ctx=dd.DistCtx(), overlap=True, prefetch=True)
ms0 = mapper.map_step
mapper.label_layout = "per_ray"
for morton in (False, True, False, True):
    mapper.morton_draws = morton
    ms = MapStep(mapper, frames, ms0.quad_list, ms0.T_list, prep=ms0.prep, features=mapper.bench_code, lambda_lt=10.0, smooth=True)
    assert ms.morton == morton
    for _ in range(30):
        ms.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        ms.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    ops.timer.arm(kernels=True)
    for _ in range(20):
        ms.step()
    torch.cuda.synchronize()
    ops.timer.disarm()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for entry, kernel, msk, units, info in ops.timer.kernel_spans:
        k = kernel.split("<")[0]
        agg[k][0] += 1
        agg[k][1] += msk
    keys = ["encode_fwd_kernel", "encode_bwd_kernel", "hashgrid_bwd_pairlist_kernel", "hashgrid_bwd_pairbins_kernel", "hashgrid_bwd_binned_kernel",
            "dgrid_transpose_kernel", "mlp_fwd_kernel", "mlp_bwd_kernel"]
    print(f"morton={morton}: {dt:.3f} ms per step (loss {float(ms.losses()[0]):.4f}) | " +
          ", ".join(f"{k.replace('_kernel', '')} {agg[k][1] / 20 * 1e3:.0f} us/step" for k in keys if k in agg))

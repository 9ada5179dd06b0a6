"""Experiment: dns_encode_fwd as one launch (OneBlob + hash grid through one LDS tile) against two launches (OneBlob only, grid
only: the grid-only launch needs a 33-float tile row instead of 49 -> 9 instead of 6 workgroups per CU)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from dns_slam_amd import dist as ddist, ops
from dns_slam_amd._lib import check, ptr, stream_ptr
ctx = ddist.DistCtx()
cfg, bound, cam, frames, mapper, step = bench.build(bench.WORKLOADS["cfg2"], "cuda:0", seed=100, dist_ctx=ctx, overlap=True)
for _ in range(3): step()
torch.cuda.synchronize()
ms = mapper.map_step
lib = ops.lib._raw
meta = C.byref(ms.meta.c)
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for name, pts, P, b6, wd in (("ray", ms.pts.reshape(-1, 3), ms.P, ms.b6, True), ("lattice", ms.pts_l, ms.Pl, None, False)):
    buf = torch.empty(P, 80, device="cuda"); x3 = torch.empty(P, 3, device="cuda")
    dydx = torch.empty(16 * 3 * P * 2, device="cuda") if wd else None
    grid = C.c_void_p(buf.data_ptr() + 4 * 48)
    full = lambda: check(lib.dns_encode_fwd(ptr(pts), b6, P, 16, ptr(ms.p_table), meta, ptr(x3) if b6 else None, ptr(buf), 80, grid, 80, ptr(dydx), stream_ptr()))
    pe = lambda: check(lib.dns_encode_fwd(ptr(pts), b6, P, 16, None, None, ptr(x3) if b6 else None, ptr(buf), 80, None, 80, None, stream_ptr()))
    gr = lambda: check(lib.dns_encode_fwd(ptr(pts), b6, P, 16, ptr(ms.p_table), meta, None, None, 80, grid, 80, ptr(dydx), stream_ptr()))
    print(f"{name}: one launch {timeit(full):.1f} us; OneBlob only {timeit(pe):.1f} us + grid only {timeit(gr):.1f} us")

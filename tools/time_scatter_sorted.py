"""Experiment: the table scatter (dns_encode_bwd, d_table only) on the lattice points in x-major order against Morton order, and on
a step's ray samples in ray order against Morton order."""
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
def part1by2(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v
def morton(x, bits=8):
    q = (x.clamp(0, 1) * ((1 << bits) - 1)).long()
    return torch.argsort(part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2))
def timeit(x, dbuf):
    P = x.shape[0]
    g = torch.zeros_like(ms.p_table)
    ws = torch.empty(int(lib.dns_encode_bwd_ws_floats(P, meta, 0, 0)), device="cuda")
    dgrid = C.c_void_p(dbuf.data_ptr() + 4 * 48)
    fn = lambda: check(lib.dns_encode_bwd(ptr(x), None, P, 16, ptr(ms.p_table), meta, None, 80, dgrid, 80, ptr(g), None, None, ptr(ws), 0, 0, stream_ptr()))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for name, x, d in (("lattice", ms.pts_l.clone(), ms.d_bufl.clone()), ("rays", ms.x3.clone(), ms.d_buf.clone())):
    perm = morton(x)
    print(f"{name}: given order {timeit(x, d):.1f} us, Morton order {timeit(x[perm].contiguous(), d[perm].contiguous()):.1f} us (transpose + scatter)")
x, d = ms.x3.clone(), ms.d_buf.clone()
perm = torch.randperm(x.shape[0], device="cuda")
print(f"rays: random order {timeit(x[perm].contiguous(), d[perm].contiguous()):.1f} us")
S = ms.S
perm = torch.arange(x.shape[0], device="cuda").reshape(-1, S).t().reshape(-1)          # sample-major: point (s, ray)
print(f"rays: sample-major order {timeit(x[perm].contiguous(), d[perm].contiguous()):.1f} us")
x, d = ms.pts_l.clone(), ms.d_bufl.clone()
perm = torch.randperm(x.shape[0], device="cuda")
print(f"lattice (rows as the step has them): {timeit(x, d):.1f} us; random order {timeit(x[perm].contiguous(), d[perm].contiguous()):.1f} us")

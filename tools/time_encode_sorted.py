"""Experiment: would a spatially coherent point order help the hash-grid gather?  dns_encode_fwd on the ray-ordered sample points
of a real cfg2 step against the same points sorted by a Morton code of their normalised coordinates (at several cell sizes)."""
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
P = ms.P
x = ms.x3.clone()                      # normalised coordinates of the step's samples
def part1by2(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v
def timeit(pts, with_dydx):
    buf = torch.empty(P, 80, device="cuda")
    dydx = torch.empty(16 * 3 * P * 2, device="cuda") if with_dydx else None
    grid = C.c_void_p(buf.data_ptr() + 4 * 48)
    fn = lambda: check(lib.dns_encode_fwd(ptr(pts), None, P, 16, ptr(ms.p_table), meta, None, ptr(buf), 80, grid, 80, ptr(dydx), stream_ptr()))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
print(f"ray order: {timeit(x, True):.1f} us (with dy_dx), {timeit(x, False):.1f} us (without)")
for bits in (4, 6, 8, 10):
    q = (x.clamp(0, 1) * ((1 << bits) - 1)).long()
    code = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
    perm = torch.argsort(code)
    xs = x[perm].contiguous()
    print(f"Morton order, {1 << bits} cells per axis: {timeit(xs, True):.1f} us (with dy_dx), {timeit(xs, False):.1f} us (without)")
xr = x[torch.randperm(P, device="cuda")].contiguous()
print(f"random order: {timeit(xr, True):.1f} us, {timeit(xr, False):.1f} us")
# the smoothness lattice: x-major regular grid (63^3) against its Morton order
Pl = ms.Pl
xl = ms.pts_l.clone()
P = Pl
print(f"lattice, x-major order: {timeit(xl, False):.1f} us")
for bits in (6, 8, 10):
    q = (xl.clamp(0, 1) * ((1 << bits) - 1)).long()
    code = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
    xs = xl[torch.argsort(code)].contiguous()
    print(f"lattice, Morton order, {1 << bits} cells per axis: {timeit(xs, False):.1f} us")
n = ms.n_lat
idx = torch.arange(n, device="cuda")
ii, jj, kk = torch.meshgrid(idx, idx, idx, indexing="ij")
code = part1by2(ii.reshape(-1)) | (part1by2(jj.reshape(-1)) << 1) | (part1by2(kk.reshape(-1)) << 2)
xs = xl[torch.argsort(code)].contiguous()
print(f"lattice, Morton order of the lattice indices: {timeit(xs, False):.1f} us")
# rays sorted by (frame, 2-D Morton code of the pixel), points taken sample-major inside tiles of 64 neighbouring rays: an order
# that needs only the drawn pixel indices (known a step ahead), not the sample positions
P = ms.P
N, S, K, npf = ms.N, ms.S, ms.K, ms.npf
pix = ms.cur.draws["pix"].reshape(K, npf)
W = mapper.W
row, col = pix // W, pix % W
def part1by1(v):
    v = v & 0xffff
    v = (v | (v << 8)) & 0x00ff00ff
    v = (v | (v << 4)) & 0x0f0f0f0f
    v = (v | (v << 2)) & 0x33333333
    return (v | (v << 1)) & 0x55555555
key = (torch.arange(K, device="cuda")[:, None] << 40) | (part1by1(row) << 1) | part1by1(col)
rays = torch.argsort(key.reshape(-1))                         # [N] ray ids in sorted order
for tile in (64, 32, 16):
    r = rays.reshape(-1, tile)                                # [N / tile, tile]
    perm = (r[:, None, :] * S + torch.arange(S, device="cuda")[None, :, None]).reshape(-1)     # tile t, sample s, lane l
    xs = x[perm].contiguous()
    print(f"rays by pixel Morton code, sample-major tiles of {tile} rays: {timeit(xs, True):.1f} us (with dy_dx), {timeit(xs, False):.1f} us")
perm = (rays[:, None] * S + torch.arange(S, device="cuda")[None, :]).reshape(-1)
print(f"rays by pixel Morton code, ray-major: {timeit(x[perm].contiguous(), True):.1f} us")

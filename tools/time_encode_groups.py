"""Experiment: dns_encode_fwd over level GROUPS (n launches of 16/n levels each, level-group-major) against one launch of all 16
levels, on the ray-ordered sample points of a real cfg2 step -- does keeping the working set of the table inside one XCD's L2
(4 MB; 16 levels x 512 KB = 8 MB) pay for the partial-row writes?"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from dns_slam_amd import dist as ddist, ops
from dns_slam_amd._lib import DnsGridMeta, check, ptr, stream_ptr
ctx = ddist.DistCtx()
wl = bench.WORKLOADS["cfg2"]
cfg, bound, cam, frames, mapper, step = bench.build(wl, "cuda:0", seed=100, dist_ctx=ctx, overlap=True)
for _ in range(3): step()
torch.cuda.synchronize()
ms = mapper.map_step
lib = ops.lib._raw
meta = ms.meta.c
L = meta.n_levels

def sub(l0, n):
    m = DnsGridMeta()
    C.memmove(C.byref(m), C.byref(meta), C.sizeof(DnsGridMeta))
    m.n_levels = n
    for i in range(n):
        m.scale[i], m.resolution[i], m.size[i], m.offset[i], m.hashed[i] = meta.scale[l0 + i], meta.resolution[l0 + i], meta.size[l0 + i], meta.offset[l0 + i], meta.hashed[l0 + i]
    return m

def run(groups, pts, P, with_dydx):
    n = L // groups
    buf = torch.empty(P, 80, device="cuda")
    x3 = torch.empty(P, 3, device="cuda")
    dydx = torch.empty(L * 3 * P * 2, device="cuda") if with_dydx else None
    metas = [sub(g * n, n) for g in range(groups)]
    def go():
        for g in range(groups):
            grid = C.c_void_p(buf.data_ptr() + 4 * (48 + 2 * g * n))
            dy = C.c_void_p(dydx.data_ptr() + 4 * (g * n * 3 * P * 2)) if with_dydx else None
            check(lib.dns_encode_fwd(ptr(pts), ms.b6 if with_dydx else None, P, 16, ptr(ms.p_table), C.byref(metas[g]), ptr(x3) if (g == 0 and with_dydx) else None,
                                     ptr(buf) if g == 0 else None, 80, grid, 80, dy, stream_ptr()), "enc")
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3, buf

for name, pts, P, wd in (("ray", ms.pts.reshape(-1, 3), ms.P, True), ("lattice", ms.pts_l, ms.Pl, False)):
    ref = None
    for g in (1, 2, 4, 8):
        t, buf = run(g, pts, P, wd)
        if ref is None: ref = buf
        print(f"{name}: {g} launch(es) of {L // g} levels: {t:.1f} us  equal={torch.equal(buf, ref)}")

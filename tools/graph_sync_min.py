"""Smallest graphs of the sampling step under the replay / synchronise pattern (see tools/graph_sync_bisect.py)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import ops, synthetic
dev = "cuda"
cam = synthetic.camera(H=60, W=80, fx=60.0, fy=60.0)
bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=0)
K, npf, nu, ns = 4, 100, 32, 15
g = torch.Generator().manual_seed(0)
color = frames["gt_color"].float().to(dev).contiguous()
depth = frames["gt_depth"].float().to(dev).contiguous()
label = frames["gt_label"].float().to(dev).contiguous()
quat = torch.tensor([[1.0, 0.0, 0.0, 0.0]] * K, device=dev)
trans = frames["est_c2w"][:, :3, 3].float().to(dev).contiguous()
pix = torch.randint(60 * 80, (K * npf,), generator=g).to(dev)
tu = torch.linspace(0.0, 1.0, nu, device=dev)
ts, tz = torch.rand(K, ns, generator=g).to(dev), torch.rand(K, ns, generator=g).to(dev)
camt = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
dmax = torch.full((K,), 5.0, device=dev)
ql = [quat[i].clone() for i in range(K)]


def raygen(depth_max=None):
    return ops.raygen_sample(quat, trans, pix, color, depth, label, camt, bound, (0, 60, 0, 80), npf, tu, ts, tz, depth_max=depth_max)


def vA():
    return torch.stack(ql).sum() + torch.stack([trans[i] for i in range(K)]).sum()


def vB():
    r = raygen()
    return r[7].sum() + r[2].sum()


def vC():
    r = raygen(dmax)
    return r[7].sum() + r[2].sum()


def vD():
    return raygen()[7].sum()


def vE():                      # the kernels' outputs only copied, no reduction over them
    return raygen()[7][0, :4].clone().sum()


def run(name, fn, sync):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = fn()
    vals = []
    for _ in range(3):
        gr.replay()
    vals.append(float(out.detach()))
    for r in range(4):
        for _ in range(3):
            gr.replay()
        if sync == "device":
            torch.cuda.synchronize()
        elif sync == "stream":
            torch.cuda.current_stream().synchronize()
        vals.append(float(out.detach()))
    print(name, sync, "SAME" if len(set(vals)) == 1 else "DIFFERENT", vals, flush=True)


want = sys.argv[1:] or ["A", "B", "C", "D", "E"]
for name, fn in (("A", vA), ("B", vB), ("C", vC), ("D", vD), ("E", vE)):
    if name in want:
        for sync in ("item", "device"):
            run(name, fn, sync)

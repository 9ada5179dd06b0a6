"""Which part of the path, captured in a hipGraph, survives 'replay -> device synchronize -> replay'?  One graph per stage;
a fault kills the process at the failing stage (run ONCE, read the last 'ok' line).  usage: graph_sync_probe.py [stage ...]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import ops
from dns_slam_amd.optim import FusedAdam
dev = "cuda"
torch.manual_seed(0)


def stage(name, fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    vals = []
    for _ in range(3):
        g.replay()
    vals.append(float(out.detach().float().sum()))          # .item(): stream synchronisation only
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()                                   # device-wide synchronisation
    vals.append(float(out.detach().float().sum()))
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    vals.append(float(out.detach().float().sum()))
    print(f"stage {name}", "SAME" if len(set(vals)) == 1 else "DIFFERENT", vals, flush=True)


P = 65536
x = torch.rand(P, 3, device=dev)
meta = ops.GridMeta(16, 592)
table = (torch.rand(meta.total_rows * 2, device=dev) * 2 - 1).requires_grad_(True)
nn_, nl = 64, 2
numel = lambda n_in, n_out: nn_ * n_in + (nl - 1) * nn_ * nn_ + ((n_out + 15) // 16 * 16) * nn_
w = (torch.randn(numel(80, 33), device=dev) * 0.1).requires_grad_(True)
pool = (torch.randn(8, numel(80, 33), device=dev) * 0.1).requires_grad_(True)
colw = (torch.randn(numel(112, 3), device=dev) * 0.1).requires_grad_(True)
logw = (torch.randn(numel(112, 8), device=dev) * 0.1).requires_grad_(True)
buf = torch.randn(P, 80, device=dev)
pix = torch.randn(P, 32, device=dev)
slot = torch.randint(0, 8, (P,), device=dev)
opt = FusedAdam([{"params": [w, table], "lr": 1e-3}])


def st_torch():
    return (x * 2).sum()


def st_encode_fwd():
    return ops.encode(x, table.detach(), meta, None, 16, True, True).sum()


def st_encode_bwd():
    table.grad = None
    ops.encode(x, table, meta, None, 16, True, True).sum().backward()
    return table.grad.sum()


def st_mlp():
    w.grad = None
    b = buf.clone().requires_grad_(True)
    ops.mlp(b, w, 80, 33, nn_, nl).sum().backward()
    return w.grad.sum() + b.grad.sum()


def st_render_nets():
    for p in (w, pool, colw, logw):
        p.grad = None
    b = buf.clone().requires_grad_(True)
    outs = ops.render_nets(b, pix, w, pool, colw, logw, slot, 48, (80, 33, nn_, nl), (80, 33, nn_, nl), (112, 3, nn_, nl), (112, 8, nn_, nl))
    sum(o.sum() for o in outs).backward()
    return pool.grad.sum() + b.grad.sum()


def st_composite():
    raw = torch.rand(1024, 64, 4, device=dev, requires_grad=True)
    z = torch.rand(1024, 64, device=dev).sort(-1)[0]
    lg = torch.rand(1024, 64, 8, device=dev, requires_grad=True)
    d, v, rgb, wt, sem = ops.composite(raw, z, lg)
    (d.sum() + rgb.sum() + sem.sum()).backward()
    return raw.grad.sum()


def st_tv():
    lat = torch.randn(15 * 15 * 15, 1, device=dev, requires_grad=True)
    ops.tv_smoothness(lat, 15, 16).backward()
    return lat.grad.sum()


def st_adam():
    w.grad = torch.ones_like(w)
    table.grad = torch.ones_like(table)
    opt.step()
    return opt._dev_state[0:1].clone()


stages = {"torch": st_torch, "encode_fwd": st_encode_fwd, "encode_bwd": st_encode_bwd, "mlp": st_mlp, "render_nets": st_render_nets,
          "composite": st_composite, "tv": st_tv, "adam": st_adam}
for name in (sys.argv[1:] or list(stages)):
    stage(name, stages[name])
print("all stages ok", flush=True)

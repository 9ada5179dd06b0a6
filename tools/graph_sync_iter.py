"""replay -> device synchronize -> replay of a whole captured Mapper iteration at a small size.
usage: graph_sync_iter.py [nosmooth] [noba] [noadam] [big]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import synthetic
from dns_slam_amd.decoder import Decoder
from dns_slam_amd.mapping import Mapper
fl = set(sys.argv[1:])
dev = "cuda"
big = "big" in fl
cam = synthetic.camera() if big else synthetic.camera(H=60, W=80, fx=60.0, fy=60.0)
bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=0)
cfg = synthetic.default_cfg(n_pixels=4096 if big else 400, n_samples_ray=48 if big else 32, n_surface_ray=16 if big else 15, n_frames=4,
                            hash_size=16 if big else 14, voxel_size=0.02 if big else 0.08, n_neurons=64, n_hidden_layers=2,
                            smooth_pts=64 if big else 12)
torch.manual_seed(1234)
dec = Decoder(cfg["model"], bound, n_class=8).to(dev)
mapper = Mapper(cfg, dec, bound, cam, device=dev)
mapper.static_shapes = True
mapper.is_BA = "noba" not in fl
mapper.set_decoder(frames)
opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
for grp, lr in zip(opt.param_groups, (mapper.lr, mapper.BA_cam_lr, mapper.BA_cam_lr)):
    grp["lr"] = lr
prep = mapper.prepare_frames(frames)


if "rng" in fl:                      # torch's graph-safe generator alone
    g = torch.cuda.CUDAGraph()
    torch.rand(4, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        r = torch.rand(4, device=dev)
    seq = []
    for i in range(3):
        g.replay(); seq.append(r[0].item())
    for i in range(3):
        g.replay()
    torch.cuda.synchronize(); seq.append(r[0].item())
    for i in range(3):
        g.replay()
    torch.cuda.synchronize(); seq.append(r[0].item())
    print("rng-only", seq, flush=True)
    sys.exit(0)

fixed = None
if "fixed" in fl:                   # no generator call inside the graph: every replay must give the SAME loss
    torch.manual_seed(5)
    fixed = (mapper.draw_pixels(prep), mapper.draw_jitter(), torch.rand(3, device=dev), torch.rand((1, 1, 1, 3), device=dev))


def step():
    opt.zero_grad(set_to_none=True)
    if fixed is not None:
        s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=fixed[0], jitter=fixed[1])
        loss, _ = mapper.iteration_loss(s, smooth="nosmooth" not in fl, u_offset=fixed[2], u_jitter=fixed[3])
    else:
        s = mapper.get_target_samples(frames, ql, Tl, prep=prep)
        loss, _ = mapper.iteration_loss(s, smooth="nosmooth" not in fl)
    loss.backward()
    if "noadam" not in fl:
        opt.step()
    return loss


st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(st)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print("captured", flush=True)
if "item" in fl:
    vals = []
    for r in range(15):
        g.replay()
        vals.append(float(out.detach()))
    print("item-sync losses", ["%.4f" % v for v in vals], flush=True)
else:
    for r in range(3):
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        print("round", r, "ok", float(out.detach()), flush=True)
print("all ok", sorted(fl), flush=True)

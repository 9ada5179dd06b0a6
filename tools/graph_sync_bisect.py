"""Sub-steps of one Mapper iteration (fixed draws, no optimiser: every replay must give the same checksum), each captured as
its own hipGraph and replayed: 3x with a stream sync, 3x + DEVICE sync, 3x + device sync."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import synthetic
from dns_slam_amd.decoder import Decoder
from dns_slam_amd.mapping import Mapper
dev = "cuda"
cam = synthetic.camera(H=60, W=80, fx=60.0, fy=60.0)
bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=0)
cfg = synthetic.default_cfg(n_pixels=400, n_samples_ray=32, n_surface_ray=15, n_frames=4, hash_size=14, voxel_size=0.08, n_neurons=64,
                            n_hidden_layers=2, smooth_pts=12)
torch.manual_seed(1234)
dec = Decoder(cfg["model"], bound, n_class=8).to(dev)
mapper = Mapper(cfg, dec, bound, cam, device=dev)
mapper.static_shapes = True
mapper.is_BA = True
mapper.set_decoder(frames)
opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
prep = mapper.prepare_frames(frames)
torch.manual_seed(5)
fixed = (mapper.draw_pixels(prep), mapper.draw_jitter(), torch.rand(3, device=dev), torch.rand((1, 1, 1, 3), device=dev))
params = [p for g in opt.param_groups for p in g["params"]]


def samples():
    return mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=fixed[0], jitter=fixed[1])


def s1():
    s = samples()
    return s["z_vals"].sum() + s["pts"].sum() + s["gt_depth"].sum()


def s2():
    with torch.no_grad():
        pc, pd, pv, pl, fine, coarse = mapper.renderer(samples(), strict=False)
    return pc.sum() + pd.sum() + pl.sum() + fine.sum() + coarse.sum()


def s3(smooth=False):
    loss, _ = mapper.iteration_loss(samples(), smooth=smooth, u_offset=fixed[2], u_jitter=fixed[3])
    return loss


def s4(smooth=False):
    for p in params:
        p.grad = None
    loss = s3(smooth)
    loss.backward()
    return sum(p.grad.abs().sum() for p in params if p.grad is not None)


def s5():
    return s4(True)


def s6():
    return s3(True)


def run(name, fn):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    vals = []
    for _ in range(3):
        g.replay()
    vals.append(float(out.detach()))
    sync = torch.cuda.current_stream().synchronize if os.environ.get("STREAM_SYNC") else torch.cuda.synchronize
    for r in range(4):
        for _ in range(3):
            g.replay()
        sync()
        vals.append(float(out.detach()))
    print(name, "SAME" if len(set(vals)) == 1 else "DIFFERENT", vals, flush=True)


for name, fn in (("samples", s1), ("renderer fwd", s2), ("loss fwd", s3), ("loss fwd + smooth", s6), ("fwd+bwd", s4), ("fwd+bwd + smooth", s5)):
    if len(sys.argv) > 1 and name.split()[0] not in sys.argv[1:]:
        continue
    run(name, fn)

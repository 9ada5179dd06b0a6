"""Child process of tests/test_gpu_graph_capture.py: hipGraph capture of the tracker and mapper iterations in a FRESH
process, with no eager iteration before the capture (the library's kernel attributes are set by dns_init(), never inside
a capture -- DESIGN.md section 4, "stream capture").  Prints one line `OK ...` and exits 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dns_slam_amd import synthetic  # noqa: E402
from dns_slam_amd.decoder import Decoder  # noqa: E402
from dns_slam_amd.mapping import Mapper  # noqa: E402
from dns_slam_amd.tracking import Tracker  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    dev = "cuda"
    cam = synthetic.camera(H=60, W=80, fx=60.0, fy=60.0)
    bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=0)
    cfg = synthetic.default_cfg(n_pixels=360, n_samples_ray=32, n_surface_ray=15, hash_size=14, voxel_size=0.08, smooth_pts=12,
                                track_pixels=300)
    torch.manual_seed(0)
    dec = Decoder(cfg["model"], bound, n_class=8).to(dev)
    with torch.no_grad():
        dec.pe_fn.grid_fn.params.mul_(2000.0)
    out = []
    if which == "state":
        # first use of an entry point on this device INSIDE a capture, without dns_init(): refused with DNS_E_STATE
        import ctypes as C
        from dns_slam_amd import _lib
        from dns_slam_amd.ops import mlp_param_count
        x = torch.zeros(128, 80, device=dev)
        y = torch.zeros(128, 33, device=dev)
        w = torch.zeros(mlp_param_count(80, 33, 32, 1), device=dev)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            g.capture_begin()
            rc = _lib.lib.dns_mlp_fwd(_lib.ptr(x), 80, None, 0, 0, _lib.ptr(w), 80, 33, 32, 1, _lib.ptr(y), 33, 128, None, None, 0,
                                      None, 0, C.c_void_p(s.cuda_stream))
            msg = _lib.lib.dns_last_error().decode()
            y.add_(1.0)                                   # something to capture
            g.capture_end()
        assert rc == -3 and "dns_init" in msg, (rc, msg)
        assert _lib.lib.dns_init() == 0
        rc = _lib.lib.dns_mlp_fwd(_lib.ptr(x), 80, None, 0, 0, _lib.ptr(w), 80, 33, 32, 1, _lib.ptr(y), 33, 128, None, None, 0,
                                  None, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert rc == 0
        print(f"OK first use inside a capture refused: rc {rc}, '{msg}'", flush=True)
        return
    if which in ("tracker", "both"):
        tracker = Tracker(cfg, dec, bound, cam, device=dev)
        tracker.border = 5
        cur = {"gt_color": frames["gt_color"][2], "gt_depth": frames["gt_depth"][2], "gt_label": frames["gt_label"][2]}
        c2w = frames["est_c2w"][2].clone()
        c2w[:3, 3] += torch.tensor([0.03, -0.02, 0.02])
        n_it = 12
        cam7, best = tracker.track_frame(cur, c2w, n_iters=n_it, graph=True, graph_warmup=0)   # capture is the first use
        steps_graph = float(tracker.last_optimizer._dev_state[0])
        cam7e, beste = tracker.track_frame(cur, c2w, n_iters=n_it, fused=True, graph=False)
        steps_eager = float(tracker.last_optimizer._dev_state[0])
        torch.cuda.synchronize()
        assert bool(torch.isfinite(cam7).all()) and float(best) == float(best)
        assert steps_graph == steps_eager == n_it, (steps_graph, steps_eager)
        out.append(f"tracker graph {steps_graph:.0f} steps, best {float(best):.5f} (eager {float(beste):.5f})")
    if which in ("mapper", "both"):
        mapper = Mapper(cfg, dec, bound, cam, device=dev, label_layout="per_ray")
        mapper.static_shapes = True
        mapper.set_decoder(frames)
        optimizer, ql, Tl = mapper.set_optimizer(frames, fused=True)
        for grp, lr in zip(optimizer.param_groups, (mapper.lr, mapper.BA_cam_lr, mapper.BA_cam_lr)):
            grp["lr"] = lr
        prep = mapper.prepare_frames(frames)
        losses = torch.zeros(8, device=dev)
        k = torch.zeros((), dtype=torch.long, device=dev)

        def step():
            optimizer.zero_grad(set_to_none=True)
            s = mapper.get_target_samples(frames, ql, Tl, prep=prep)
            loss, _ = mapper.iteration_loss(s, smooth=True)
            loss.backward()
            optimizer.step()
            with torch.no_grad():
                losses.index_copy_(0, k.reshape(1), loss.detach().reshape(1))
                k.add_(1)

        from dns_slam_amd._lib import ensure_init
        ensure_init()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(6):
            g.replay()
        torch.cuda.synchronize()
        l = losses[:6].tolist()
        assert all(v == v and v > 0 for v in l), l
        assert float(optimizer._dev_state[0]) == 6
        assert l[-1] < l[0], l                           # six Adam steps lower the loss
        out.append("mapper graph losses " + " ".join(f"{v:.4f}" for v in l))
    print("OK " + "; ".join(out), flush=True)


if __name__ == "__main__":
    main()

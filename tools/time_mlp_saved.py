"""dns_mlp_bwd recomputing the hidden activations from x against reading back what dns_mlp_fwd kept (h_save -> h_saved), and what
keeping them costs the forward: per-launch time at P points."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import ops
from dns_slam_amd._lib import check, ptr, stream_ptr
lib = ops.lib._raw
P = int(os.environ.get("DNS_P", 204800))
for n_in, n_out, nn, nl in ((80, 33, 64, 2), (112, 8, 64, 2)):
    count = ops.mlp_param_count(n_in, n_out, nn, nl)
    params = (torch.randn(count, device="cuda") * 0.2)
    x = torch.randn(P, n_in, device="cuda"); dy = torch.randn(P, n_out, device="cuda")
    y = torch.empty(P, n_out, device="cuda"); dx = torch.empty(P, n_in, device="cuda"); dp = torch.zeros(count, device="cuda")
    ws = torch.empty(int(lib.dns_mlp_bwd_ws_floats(P, nn, nl)), device="cuda")
    hs = torch.empty(nl * P * nn, device="cuda")
    res = {}
    for name, h in (("recompute", None), ("saved", hs)):
        def fwd(): check(lib.dns_mlp_fwd(ptr(x), n_in, None, 0, 0, ptr(params), n_in, n_out, nn, nl, ptr(y), n_out, P, None, None, 0, ptr(h), 0, stream_ptr()))
        def bwd(): check(lib.dns_mlp_bwd(ptr(x), n_in, None, 0, 0, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(dx), n_in, None, 0, ptr(dp), ptr(ws), P, None, None, 0, ptr(h), ops.MLP_NO_DWIN_FLAG, stream_ptr()))
        out = []
        for fn in (fwd, bwd):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[name] = (out, dx.clone(), dp.clone())
        print(f"{n_in}->{nn}x{nl}->{n_out} {name:10s}: fwd {out[0]:.1f} us, bwd (without dW_in) {out[1]:.1f} us")
    print("   dx equal:", torch.equal(res["recompute"][1], res["saved"][1]))

"""dns_mlp_fwd / dns_mlp_bwd with the fp32 weights (the kernels build their operand images per workgroup) against
DNS_MLP_PREPARED (the images copied in from what dns_mlp_prepare wrote): per-launch time at P points."""
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
    prep = torch.empty(int(lib.dns_mlp_prepared_floats(n_in, n_out, nn, nl)), device="cuda")
    check(lib.dns_mlp_prepare(ptr(params), n_in, n_out, nn, nl, 1, 0, ptr(prep), 0, stream_ptr()))
    for name, w, flag in (("fp32 weights", params, 0), ("prepared", prep, ops.MLP_PREPARED_FLAG)):
        def fwd(): check(lib.dns_mlp_fwd(ptr(x), n_in, None, 0, 0, ptr(w), n_in, n_out, nn, nl, ptr(y), n_out, P, None, None, 0, None, flag, stream_ptr()))
        def bwd(): check(lib.dns_mlp_bwd(ptr(x), n_in, None, 0, 0, ptr(dy), n_out, ptr(w), n_in, n_out, nn, nl, ptr(dx), n_in, None, 0, ptr(dp), ptr(ws), P, None, None, 0, None, flag, stream_ptr()))
        res = []
        for fn in (fwd, bwd):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"{n_in}->{nn}x{nl}->{n_out} {name:13s}: fwd {res[0]:.1f} us, bwd (+dW_in) {res[1]:.1f} us")

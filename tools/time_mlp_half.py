"""Time the half-rows MLP entry points (dns_mlp_fwd_half / dns_mlp_bwd_half) beside the fp32-grade and fp16-operand kernels on
the same shapes (event pairs over 20 launches).  usage: time_mlp_half.py [n_in n_out nn nl]   env DNS_P = points"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops
from dns_slam_amd._lib import check, ptr, stream_ptr
P = int(os.environ.get("DNS_P", 262144))
n_in, n_out, nn, nl = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (80, 33, 64, 2))]
dev = "cuda"
w = torch.randn(ops.mlp_param_count(n_in, n_out, nn, nl), device=dev) * 0.1
x32 = torch.randn(P, n_in, device=dev)
x16 = x32.half()
dy = torch.randn(P, n_out, device=dev) * 1e-3
y = torch.empty(P, n_out, device=dev)
dx = torch.zeros(P, n_in, device=dev)
dw = torch.zeros_like(w)
ws = torch.empty(max(int(ops.lib._raw.dns_mlp_bwd_ws_floats(P, nn, nl)), 4), device=dev)

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def fwd32(flag=0):
    check(ops.lib._raw.dns_mlp_fwd(ptr(x32), n_in, None, 0, 0, ptr(w), n_in, n_out, nn, nl, ptr(y), n_out, P, None, None, 0, None, flag, stream_ptr()), "fwd")
def bwd32(flag=0, dwp=True, acc=0):
    check(ops.lib._raw.dns_mlp_bwd(ptr(x32), n_in, None, 0, 0, ptr(dy), n_out, ptr(w), n_in, n_out, nn, nl, ptr(dx), n_in, None, 0,
                                   ptr(dw) if dwp else None, ptr(ws), P, None, None, 0, None, flag | acc, stream_ptr()), "bwd")
res = {
    "fwd fp32-grade": timeit(lambda: fwd32(0)),
    "fwd fp16-operand (PREC 1)": timeit(lambda: fwd32(ops.MLP_FP16_FLAG)),
    "fwd HALF": timeit(lambda: ops.mlp_fwd_half(x16, w, n_in, n_out, nn, nl, out=y)),
    "bwd+dwin fp32-grade": timeit(lambda: bwd32(0)),
    "bwd+dwin fp16-operand": timeit(lambda: bwd32(ops.MLP_FP16_FLAG)),
    "bwd HALF (all gradients)": timeit(lambda: ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_x=dx, d_params=dw)),
    "bwd HALF (all gradients, dx +=)": timeit(lambda: ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_x=dx, d_params=dw, accumulate=1)),
    "bwd HALF (dW only)": timeit(lambda: ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_params=dw)),
    "bwd fp32-grade frozen (dx only)": timeit(lambda: bwd32(0, False)),
    "bwd HALF frozen (dx only)": timeit(lambda: ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_x=dx)),
}
print(f"{n_in}->{nn}x{nl}->{n_out}, {P} points:")
for k, v in res.items():
    print(f"  {k:36s} {v:8.1f} us")

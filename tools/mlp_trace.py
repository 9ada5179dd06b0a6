"""Developer tool: per-phase issue-time stamps of one wave of mlp_fwd_kernel (library built with -DDNS_TRACE).

    make -C dns_slam_amd/csrc trace        # -> dns_slam_amd/libdns_hip_trace.so
    DNS_HIP_LIB=$PWD/dns_slam_amd/libdns_hip_trace.so python tools/mlp_trace.py [n_in n_out nn nl]
"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops, _lib
P = int(os.environ.get("DNS_P", 262144))
n_in, n_out, nn, nl = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (112, 8, 64, 2))]
x = torch.randn(P, n_in, device="cuda", requires_grad=True)
w = (torch.randn(ops.mlp_param_count(n_in, n_out, nn, nl), device="cuda") * 0.1).requires_grad_(True)
raw = ctypes.CDLL(os.environ["DNS_HIP_LIB"])
buf = (ctypes.c_ulonglong * 128)()
which = os.environ.get("DNS_TRACE_KERNEL", "fwd")
gy = torch.randn(P, n_out, device="cuda")
for _ in range(3):
    y = ops.mlp(x, w, n_in, n_out, nn, nl)
    if which == "bwd":
        torch.cuda.synchronize()
        y.backward(gy)        # the backward kernel's stamps overwrite the forward's
torch.cuda.synchronize()
assert raw.dns_trace_read(buf) == 0
if which == "fwd":
    names = ["start", "images"] + ["x issue", "commit0", "mfma0", "commit1", "mfma1", "commit2", "mfma2", "commit3", "mfma3", "layer_in(prefetch issue)", "relu+save", "layer_h+save", "out mfma", ] * 12
else:
    names = ["start", "images"] + ["h loads issued", "dl = Wout^T dy", "relu' + chain", "ws stores", "dX chain 0 (prev stores)", "dX chain 1 (+dx stores 0)"] * 16
for blk in range(2):
    t = [buf[blk * 64 + i] for i in range(64)]
    print(f"workgroup {'0' if blk == 0 else '300'}: (10 ns ticks since kernel start of this workgroup)")
    last = t[0]
    for i in range(1, 40):
        if t[i] == 0 or t[i] < t[0]:
            break
        print(f"  {i:2d} {names[i]:14s} +{(t[i] - last) * 10:6d} ns   (t={(t[i] - t[0]) * 10} ns)")
        last = t[i]

"""Phase times of the MLP backward kernel from in-kernel s_memtime stamps (tools build: make -C dns_slam_amd/csrc trace).
usage: DNS_HIP_LIB=dns_slam_amd/libdns_hip_trace.so python tools/bwd_trace.py [n_in n_out accumulate]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import ops
from dns_slam_amd._lib import lib
n_in, n_out, acc = (int(v) for v in (sys.argv[1:4] + ["80", "33", "0"][len(sys.argv) - 1:]))
P, nn, nl = 262144, 64, 2
dev = "cuda"
numel = nn * n_in + nn * nn + ((n_out + 15) // 16 * 16) * nn
w = (torch.randn(numel, device=dev) * 0.1).requires_grad_(os.environ.get("DNS_NO_DW") is None)
x = torch.randn(P, n_in, device=dev, requires_grad=True)
gy = torch.randn(P, n_out, device=dev)
trace = torch.zeros(256 * 4 * 8 * 8, device=dev, dtype=torch.int64)
raw = lib._raw if hasattr(lib, "_raw") else lib
raw.dns_debug_bwd_trace.argtypes = [C.c_void_p]
raw.dns_debug_bwd_trace.restype = None
for it in range(3):
    x.grad = None
    w.grad = None
    if acc:
        x.grad = torch.zeros_like(x)          # accumulate path is exercised through render_nets in the product; here: plain
    y = ops.mlp(x, w, n_in, n_out, nn, nl)
    if it == 2:
        trace.zero_()
        raw.dns_debug_bwd_trace(C.c_void_p(trace.data_ptr()))
    y.backward(gy)
torch.cuda.synchronize()
raw.dns_debug_bwd_trace(None)
t = trace.cpu().reshape(256 * 4, 8, 8).double()
ok = (t[:, :, 0] > 0) & (t[:, :, 6] > 0)
d = t[:, :, 1:7] - t[:, :, 0:6]
names = ["A recompute", "B dH_last", "C dW_out", "D dH1 + dW_h", "E dW_in", "E dX tiles"]
print(f"{n_in}->{nn}x{nl}->{n_out}: cycles (s_memtime, 100 MHz ticks x ?) per phase, mean over {int(ok.sum())} tiles (tiles 1..5 of each wave)")
sel = ok.clone(); sel[:, 0] = False
for k, nme in enumerate(names):
    print(f"  {nme:14s} {float(d[:, :, k][sel].mean()):10.1f}")
print(f"  tile total     {float((t[:, :, 6] - t[:, :, 0])[sel].mean()):10.1f}")
run = t[:, 7, :4]
okr = (run[:, 0] > 0) & (run[:, 3] > 0)
for k, nme in enumerate(["images (prologue)", "tile loop", "flush + barrier"]):
    print(f"  run: {nme:18s} {float((run[:, k + 1] - run[:, k])[okr].mean()):10.1f}")
gap = t[:, 1:6, 0] - t[:, 0:5, 6]
print(f"  between tiles  {float(gap[ok[:, 1:6] & ok[:, 0:5]].mean()):10.1f}")

"""Time dns_encode_bwd's table scatter alone (torch events), per scatter form (ops.SCATTER_FORM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops
P = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
if len(sys.argv) > 2:
    ops.SCATTER_FORM = (int(sys.argv[2]), 0)
pm = ops.GridMeta(16, 592)
dev = "cuda"
g = torch.Generator().manual_seed(0)
# ray-like points: 4096 rays x 64 samples through the unit cube
o = torch.rand(P // 64, 1, 3, generator=g) * 0.3 + 0.35
d = torch.randn(P // 64, 1, 3, generator=g) * 0.3
t = torch.linspace(0, 1, 64)[None, :, None]
x = (o + d * t).reshape(-1, 3).clamp(0, 1).to(dev)
table = torch.rand(pm.total_rows * 2, device=dev).requires_grad_(True)
y = ops.encode(x, table, pm, None, 16, True, True)
gy = torch.randn_like(y)
for _ in range(3):
    table.grad = None
    y.backward(gy, retain_graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    table.grad = None
    y.backward(gy, retain_graph=True)
e1.record()
torch.cuda.synchronize()
print(f"P={P} form={ops.SCATTER_FORM[0]}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per backward (incl. zeros_like + transpose)")

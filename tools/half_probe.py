"""Diagnostics of the half-rows MLP kernels against the f16 emulation of tests/test_gpu_mlp_half.py: rms errors and where the worst rows are."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from oracle import tcnn_ref as tr
from dns_slam_amd import ops
import test_gpu_mlp_half as T
DEV = "cuda:0"
for (n_in, n_out, nn, nl, P) in [(80, 33, 64, 2, 5000), (80, 1, 64, 2, 5000), (80, 1, 64, 2, 40000), (80, 33, 64, 2, 40000), (80, 33, 32, 1, 70000)]:
    g = torch.Generator().manual_seed(7)
    w = tr.mlp_init(n_in, n_out, nn, nl, g).to(DEV) * 3.0
    x16 = torch.randn(P, n_in, generator=g).to(DEV).half()
    dy = (torch.randn(P, n_out, generator=g) * 1e-3).to(DEV)
    y = ops.mlp_fwd_half(x16, w, n_in, n_out, nn, nl)
    dx = torch.full((P, n_in), float("nan"), device=DEV)
    dw = torch.zeros_like(w)
    ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_x=dx, d_params=dw)
    ye, dxe, dWe = T._emulate(x16, w, dy, n_in, n_out, nn, nl)
    want = torch.cat([t.reshape(-1) for t in dWe])
    err = (dx.double() - dxe.double()).abs().max(1)[0]
    sc = float(dxe.abs().max())
    worst = torch.topk(err, 8)
    print(f"{(n_in, n_out, nn, nl, P)}: y {T._rms_rel(y, ye):.2e} dx {T._rms_rel(dx, dxe):.2e} dW {T._rms_rel(dw[:want.numel()], want):.2e}"
          f" | rows off by > 1e-2 scale: {int((err > 1e-2 * sc).sum())} worst rows {worst.indices.tolist()} err/scale {[round(float(v) / sc, 4) for v in worst.values]}")
    # per-matrix dW errors
    o = 0
    for t in dWe:
        n = t.numel()
        print("    dW block", tuple(t.shape), f"{T._rms_rel(dw[o:o + n], t.reshape(-1)):.2e}")
        o += n

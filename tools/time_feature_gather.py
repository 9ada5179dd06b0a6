"""dns_feature_gather (the 2-D code lookup of the feature branch) per launch at a cfg3 iteration's size: 3 reference views x
262 144 points, 64-channel half-resolution stem maps of a 640 x 480 frame; DNS_FEATURE_GATHER_LANES=1 = the one-lane-per-channel
kernel (a wave per pair), default = four channels per lane (four pairs per wave).  Prints time and a checksum of the code."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops

g = torch.Generator().manual_seed(0)
R, C, h, w, H, W, P = 3, 64, 240, 320, 480, 640, 262144
feats = torch.randn(R, h, w, C, generator=g).cuda()
K = torch.tensor([[600.0, 0, (W - 1) / 2], [0, 600.0, (H - 1) / 2], [0, 0, 1.0]])
w2c = torch.eye(4).repeat(R, 1, 1)
w2c[:, :3, 3] = torch.randn(R, 3, generator=g) * 0.2
pts = (torch.randn(P, 3, generator=g) * torch.tensor([1.2, 0.9, 1.0]) + torch.tensor([0.0, 0.0, -3.0])).cuda()
w2c = w2c.cuda()
for _ in range(3):
    code, mask = ops.feature_gather(pts, w2c, K, feats, H, W)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    code, mask = ops.feature_gather(pts, w2c, K, feats, H, W)
e1.record()
torch.cuda.synchronize()
print(f"feature gather, {R} x {P} pairs x {C} channels: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (incl. two allocations), "
      f"valid {float(mask.float().mean()):.3f}, checksum {float(code.double().abs().sum()):.9e}")

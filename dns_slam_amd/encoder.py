"""Mirror of reference models/encoder.py:4-17 (``ResNet``) -- the frozen image stem of the 2-D feature branch.

Only the stem the reference actually executes (models/layers.py:56-58,95-98: conv 7x7 stride 2 + batch-norm + ReLU; the
residual stages are commented out there).  It runs ONCE per ``optimize()`` / per tracked frame, outside the iteration
loop (slams/mapping.py:846, slams/tracking.py:293-296), so it stays on stock PyTorch-ROCm convolution.  The reference
downloads ImageNet ResNet-18 weights at construction (layers.py:125); there is no network here: pass ``weights`` (a
state dict or a path holding ``conv1.weight``, ``bn1.*``) or get the reference's own fallback initialisation
(layers.py:70-76: He-normal conv, unit batch-norm)."""
import math

import torch
import torch.nn as nn


class _Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        n = 7 * 7 * 64
        self.conv1.weight.data.normal_(0, math.sqrt(2.0 / n))
        self.bn1.weight.data.fill_(1)
        self.bn1.bias.data.zero_()

    def forward(self, x):
        return self.relu(self.bn1(self.conv1(x)))


class ResNet(nn.Module):
    def __init__(self, weights=None, seed=0):
        super().__init__()
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.conv_blocks = _Stem()
        torch.random.set_rng_state(g)
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            own = self.conv_blocks.state_dict()
            own.update({k: v for k, v in sd.items() if k in own})
            self.conv_blocks.load_state_dict(own)
        self.eval()
        for p in self.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def forward(self, images):
        """images [B, N, H, W, 3] -> features [B, N, 64, H/2, W/2] (models/encoder.py:9-17)."""
        B, N = images.shape[:2]
        x = images.flatten(0, 1).permute(0, 3, 1, 2)
        f = self.conv_blocks(x)
        return f.reshape(B, N, *f.shape[1:])

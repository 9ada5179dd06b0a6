"""Fused Adam (csrc/adam.hip): the reference's ``torch.optim.Adam`` step (slams/mapping.py:464,910,
slams/tracking.py:120-124,339) over all parameter tensors in one launch, with the step count on the device so the update
replays from a hipGraph.  Same constructor shape as ``torch.optim.Adam`` for what the reference uses: a list of
``{'params': [...], 'lr': ...}`` groups, default betas (0.9, 0.999) and eps 1e-8, no weight decay, no amsgrad."""
from __future__ import annotations

from typing import List

import torch

from ._lib import DnsAdamTensor, check, ptr, require_cuda, stream_ptr
from .ops import lib


class FusedAdam:
    def __init__(self, param_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if isinstance(param_groups, (list, tuple)) and param_groups and not isinstance(param_groups[0], dict):
            param_groups = [{"params": list(param_groups)}]
        self.param_groups: List[dict] = []
        for g in param_groups:
            g = dict(g)
            g["params"] = [p for p in g["params"] if p.numel() > 0]
            g.setdefault("lr", lr)
            self.param_groups.append(g)
        self.betas, self.eps = betas, eps
        self.state = {}
        dev = None
        for g in self.param_groups:
            for p in g["params"]:
                require_cuda(p)
                dev = p.device
                self.state[p] = (torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32))
        self._dev_state = torch.zeros(3, device=dev) if dev is not None else None     # fresh moments, step 0

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    @torch.no_grad()
    def step(self):
        items = []
        for g in self.param_groups:
            lr = float(g["lr"])
            for p in g["params"]:
                if not p.requires_grad or p.grad is None:
                    continue
                gr = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                m, v = self.state[p]
                items.append((p, gr, m, v, lr))
        for i in range(0, len(items), 32):
            chunk = items[i:i + 32]
            arr = (DnsAdamTensor * len(chunk))()
            for k, (p, gr, m, v, lr) in enumerate(chunk):
                arr[k].p, arr[k].g, arr[k].m, arr[k].v = p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr()
                arr[k].n, arr[k].lr = p.numel(), lr
            # one step count for all chunks: only the first chunk of a step ticks it
            st = self._dev_state if i == 0 else self._dev_state_view()
            check(lib.dns_adam_step(arr, len(chunk), self.betas[0], self.betas[1], self.eps, ptr(st), stream_ptr()),
                  "dns_adam_step")

    def _dev_state_view(self):
        raise ValueError("FusedAdam: more than 32 parameter tensors per step are not supported")

"""Fused AdamW on the GPU (replaces the reference's DeepSpeed CPU-offloaded Adam,
configs/deepspeed_config.json:2-13,24-27): fp32 master weights and moments, model-dtype
(bf16) parameters and gradients, one mk_adamw launch per parameter tensor."""
from __future__ import annotations

import torch

from . import ops


class FusedAdamW:
    def __init__(self, params, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = {}

    def _state(self, p):
        st = self.state.get(p)
        if st is None:
            master = ops.cast(p.detach().contiguous(), torch.float32) if p.dtype != torch.float32 \
                else p.detach().clone()
            m = torch.empty_like(master)
            v = torch.empty_like(master)
            ops.fill_(m, 0.0)
            ops.fill_(v, 0.0)
            st = self.state[p] = (master, m, v)
        return st

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step_param(self, p, grad_scale: float = 1.0):
        """update ONE parameter on the current stream with the current step_count
        (used by train.OverlappedStep from gradient hooks)"""
        if p.grad is None:
            return
        b1, b2 = self.betas
        master, m, v = self._state(p)
        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        ops.adamw_(p.data, master, m, v, g, self.lr, b1, b2, self.eps, self.weight_decay,
                   self.step_count, grad_scale)

    # ---- ZeRO-1 style shard (train.OverlappedStep, world > 1) ---------------------------------
    def _shard_state(self, key, w):
        st = self.state.get(key)
        if st is None:
            master = ops.cast(w.contiguous(), torch.float32) if w.dtype != torch.float32 else w.clone()
            m = torch.empty_like(master)
            v = torch.empty_like(master)
            ops.fill_(m, 0.0)
            ops.fill_(v, 0.0)
            st = self.state[key] = (master, m, v)
        return st

    @torch.no_grad()
    def step_shard(self, key, w, grad_shard, grad_scale: float = 1.0):
        """update the flat parameter slice `w` (a view into one or several adjacent parameters)
        from `grad_shard`, the rank-averaged gradient of exactly those elements.  Optimizer
        state (fp32 master / m / v, 12 B per owned element) exists only for the slice and is
        keyed by `key` (stable across steps: the slice a rank owns never changes)."""
        b1, b2 = self.betas
        master, m, v = self._shard_state(key, w)
        ops.adamw_(w, master, m, v, grad_shard, self.lr, b1, b2, self.eps, self.weight_decay,
                   self.step_count, grad_scale)

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        self.step_count += 1
        for p in self.params:
            self.step_param(p, grad_scale)

"""Replicated data parallelism over the GPUs of one node: one process per GPU
(torch.distributed, backend "nccl" = RCCL over xGMI), gradients averaged once per step.

The reference trains through DeepSpeed ZeRO-3 with CPU offload (train.sh:16,
configs/deepspeed_config.json); samples are independent and the loss is a per-rank mean, so
plain replicated DP with a mean all-reduce of the gradients is exactly equivalent (SURVEY
§8e).  Gradients are reduced from post-accumulate hooks WHILE the backward of earlier layers
is still running: each LLaMA layer's weight gradients (~400 MB bf16) become ready together
when that layer's backward block returns, and go out as large per-tensor collectives on
RCCL's own stream; the many tiny tensors (norm weights, biases) are coalesced into one flat
buffer reduced at the end.  Only parameters that actually receive gradients are reduced
(frozen encoders / unused towers are skipped — no find_unused_parameters pass).
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None,
                 small_threshold: int = 1 << 20, average: bool = True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        # RCCL reduces with AVG; gloo only has SUM (also for CUDA tensors staged through the host):
        # the choice follows the BACKEND, not where the tensor lives
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.small_threshold = small_threshold
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self._handles = []
        self._small: List[torch.nn.Parameter] = []
        self._hooks = []
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # called by autograd right after p.grad has been written for this step
    def _on_grad(self, p: torch.nn.Parameter):
        if p.grad is None:
            return
        if p.grad.numel() < self.small_threshold:
            self._small.append(p)
            return
        self._launch(p.grad)

    def _launch(self, t: torch.Tensor):
        op = dist.ReduceOp.AVG if (self.average and self._avg) else dist.ReduceOp.SUM
        h = dist.all_reduce(t, op=op, group=self.group, async_op=True)
        self._handles.append((h, t, op))

    def finish(self):
        """Flush the coalesced small gradients and wait for every collective."""
        if self.world <= 1:
            return
        if self._small:
            flat = torch.cat([p.grad.reshape(-1) for p in self._small])
            self._launch(flat)
        for h, t, op in self._handles:
            h.wait()
            if self.average and op == dist.ReduceOp.SUM:
                t.div_(self.world)   # gloo (CPU tests) has no AVG
        if self._small:
            flat = self._handles[-1][1]
            off = 0
            for p in self._small:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._handles.clear()
        self._small.clear()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()


def shard_batch(global_batch: int, rank: int, world: int):
    """Even split of the global batch; returns (start, stop) of this rank's samples."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per

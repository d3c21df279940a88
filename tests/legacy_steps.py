"""Reference implementations of the per-tensor data-parallel steps of rounds 1-2, kept ONLY as
test yardsticks for macaw_llm_amd.bucketed.BucketedStep (the package's one step runtime since round
3): GradSync (hook-driven per-tensor all-reduce) and OverlappedStep (per-tensor all-reduce or
ZeRO-1 reduce-scatter / shard AdamW / all-gather behind the backward).  Not imported by the
package, bench.py or __graft_entry__."""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from macaw_llm_amd import ops  # noqa: F401
from macaw_llm_amd.optim import FusedAdamW


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


class OverlappedStep:
    def __init__(self, params: Iterable[torch.nn.Parameter], opt: FusedAdamW, process_group=None,
                 small_threshold: int = 1 << 20, overlap: bool = True,
                 overlap_optimizer: bool = False, shard_optimizer: Optional[bool] = None,
                 force_collectives: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.opt = opt
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # RCCL reduces with AVG; gloo (CPU tests, and CUDA tensors staged through the host) only
        # has SUM: the mean is then finished with a division
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        # force_collectives: issue the collectives even with one rank (exercises the RCCL call
        # path on a single-GPU box)
        self.collective = self.world > 1 or (force_collectives and dist.is_initialized())
        self.shard = (self.collective if shard_optimizer is None else bool(shard_optimizer)) \
            and self.collective and hasattr(opt, "step_shard")
        self.small_threshold = small_threshold
        self.overlap = overlap
        # Measured on MI355X (1 GPU, cfg 3): running AdamW beside the backward GEMMs slows those
        # GEMMs by exactly what it saves (they are memory-latency sensitive: 293 -> 313 ms of GEMM
        # time, step time unchanged), so by default only the COLLECTIVES overlap the backward
        # and the optimizer runs after it.
        self.overlap_optimizer = overlap_optimizer
        dev = self.params[0].device
        self.side = (torch.cuda.Stream(device=dev)
                     if (overlap and (overlap_optimizer or self.shard) and dev.type == "cuda") else None)
        self._small: List[torch.nn.Parameter] = []
        self._pending = []  # (handle, param) for the non-overlapped / CPU path
        self._shards = []   # (reduce-scatter handle, param, lo, n, grad shard) not yet updated
        self._gathers = []  # all-gather handles of this step
        self._run = None    # open run of memory-adjacent parameters (fused q|k|v, gate|up)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def begin(self):
        """call once per step before backward: advances Adam's bias-correction step"""
        self.opt.step_count += 1
        for p in self.params:
            p.grad = None

    # ---- autograd hook: p.grad has just been written on the compute stream
    def _on_grad(self, p):
        g = p.grad
        if g is None:
            return
        if self.collective and g.numel() < self.small_threshold:
            self._small.append(p)
            return
        if self.shard:
            # q|k|v (gate|up) live back to back in one fused buffer and so do their gradients
            # (modeling.LlamaDecoderLayer.fuse_projections): extend the open run instead of
            # issuing three (two) collectives
            if self._run is not None and self._extends_run(p, g):
                self._run["params"].append(p)
                self._run["n"] += g.numel()
                return
            self._flush_run()
            if g.is_contiguous() and p.data.is_contiguous():
                self._run = dict(params=[p], n=g.numel(), g0=g, w0=p.data)
                return
        handle = None
        if self.collective:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            handle = dist.all_reduce(g, op=op, group=self.group, async_op=self.overlap)
        if self.side is not None:
            cur = torch.cuda.current_stream(g.device)
            ev = torch.cuda.Event()
            ev.record(cur)
            g.record_stream(self.side)
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                if handle is not None:
                    handle.wait()          # stream-ordered wait on the collective
                    if not self._avg:
                        self._div(g)
                self.opt.step_param(p)
        else:
            self._pending.append((handle, p))

    def _div(self, t):
        """finish a SUM-reduced mean (gloo only; RCCL reduces with AVG)"""
        t.div_(self.world)

    # ---- ZeRO-1 path ------------------------------------------------------------------------
    def _extends_run(self, p, g) -> bool:
        r = self._run
        es = g.element_size()
        return (g.is_contiguous() and p.data.is_contiguous() and g.dtype == r["g0"].dtype
                and p.data.dtype == r["w0"].dtype
                and g.data_ptr() == r["g0"].data_ptr() + r["n"] * es
                and p.data.data_ptr() == r["w0"].data_ptr() + r["n"] * p.data.element_size()
                and g.untyped_storage().data_ptr() == r["g0"].untyped_storage().data_ptr()
                and p.data.untyped_storage().data_ptr() == r["w0"].untyped_storage().data_ptr())

    def _flush_run(self):
        """issue the collective(s) of the open run of adjacent parameters"""
        r, self._run = self._run, None
        if r is None:
            return
        n_all = r["n"]
        g = r["g0"].as_strided((n_all,), (1,))
        w = r["w0"].as_strided((n_all,), (1,))
        if n_all % (8 * self.world) == 0:       # 16-byte aligned slices for the vector AdamW kernel
            self._reduce_scatter(r["params"][0], g, w)
            return
        for p in r["params"]:                   # not divisible: replicated update after an all-reduce
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            h = dist.all_reduce(p.grad, op=op, group=self.group, async_op=self.overlap)
            self._pending.append((h, p))

    def _reduce_scatter(self, key_param, g, w):
        n = g.numel() // self.world
        lo = self.rank * n
        gs = torch.empty(n, dtype=g.dtype, device=g.device)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        h = dist.reduce_scatter_tensor(gs, g, op=op, group=self.group, async_op=self.overlap)
        if self.side is not None:
            gs.record_stream(self.side)
            with torch.cuda.stream(self.side):
                if h is not None:
                    h.wait()               # side stream waits for the collective, not the host
                self._update_and_gather(key_param, w, lo, n, gs)
        else:
            self._shards.append((h, key_param, w, lo, n, gs))

    def _update_and_gather(self, key_param, w, lo, n, gs):
        if not self._avg:
            self._div(gs)                  # gloo has no AVG
        self.opt.step_shard((key_param, lo, n), w[lo:lo + n], gs)
        h = dist.all_gather_into_tensor(w, w[lo:lo + n], group=self.group, async_op=self.overlap)
        if h is not None:
            self._gathers.append(h)

    def finish(self):
        """flush small tensors, run whatever was not overlapped, join the side stream"""
        self._flush_run()
        for h, kp, w, lo, n, gs in self._shards:
            if h is not None:
                h.wait()
            self._update_and_gather(kp, w, lo, n, gs)
        self._shards.clear()
        if self.collective and self._small:
            flat = torch.cat([p.grad.reshape(-1) for p in self._small])
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            dist.all_reduce(flat, op=op, group=self.group)
            if op == dist.ReduceOp.SUM:
                flat.div_(self.world)
            off = 0
            for p in self._small:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        todo = []
        for handle, p in self._pending:
            if handle is not None:
                if self.overlap:
                    handle.wait()
                if not self._avg:
                    self._div(p.grad)
            todo.append(p)
        todo.extend(self._small)
        if hasattr(self.opt, "step_params") and not os.environ.get("MACAW_ADAMW_SINGLE"):
            self.opt.step_params(todo)      # one multi-tensor launch for everything replicated
        else:
            for p in todo:
                self.opt.step_param(p)
        self._pending.clear()
        self._small.clear()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        for h in self._gathers:            # the next forward reads the gathered parameters
            h.wait()
        self._gathers.clear()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()



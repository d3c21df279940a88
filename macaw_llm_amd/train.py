"""Training-step runtime: gradient averaging (RCCL) and the fused AdamW update both run
*behind* the backward pass instead of after it.

Backward on this chip is MFMA-bound (the GEMMs), the optimizer is HBM-bound (28 B per
parameter) and the gradient all-reduce is xGMI-bound: three different resources.  As soon as
autograd has written a parameter's gradient (post-accumulate hook — for a LLaMA layer all nine
weight gradients appear together when that layer's backward block returns) we
  1. (N > 1) launch its asynchronous all-reduce (RCCL runs on its own stream),
  2. enqueue its AdamW update on a side HIP stream that waits for (1) — or, for N = 1, for an
     event recorded on the compute stream —
so collectives and optimizer traffic overlap the remaining backward GEMMs.  `finish()` flushes
the coalesced small tensors and joins the streams.  Numerically identical to
backward -> all-reduce -> optimizer.step().

`shard_optimizer=True` (the default when N > 1) is the ZeRO-1 form of the same step -- the
reference itself trains under DeepSpeed ZeRO (train.sh:16, configs/deepspeed_config.json): each
large gradient is REDUCE-SCATTERED instead of all-reduced, every rank runs AdamW only on the
1/N slice it owns (fp32 master / moments exist only for that slice: 12 B per owned element),
and the updated bf16 slices are ALL-GATHERED in place into the parameter.  Same bytes on the
xGMI links as a ring all-reduce, but the optimizer's HBM traffic (28 B per parameter, 38 ms per
step at 7B, 13 % of a 1-GPU step) and its state shrink by N.  Slice update + all-gather run on a
side stream / RCCL's stream behind the remaining backward; `finish()` joins them.  The
parameters are bit-identical to the replicated update (the mean gradient of an element is the
same number whichever collective produced it).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .optim import FusedAdamW


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


class GraphedStep:
    """Single-GPU training step replayed from ONE hipGraph: zero grads -> forward -> backward -> fused
    AdamW, ~1,160 kernels at cfg 3, captured once and launched with a single call per step.  (At cfg
    3 it buys 0.2 %: 249.1 vs 249.6 ms -- the ~6 us between consecutive kernels in the rocprofv3
    traces are the tracer's; the host is far ahead of the GPU in the eager step.  It matters where
    kernels are short, e.g. small models.)  What changes from step to step lives in device memory: the
    optimizer's {lr, bias corrections, grad scale} (FusedAdamW.update_hyper) and the dropout seed
    offset (ops.set_dropout_seed_offset; MM_LLMs._dropout_seed strides by SEED_STRIDE per step), so
    a replay is bit-identical to the eager step it stands for.

    loss_fn() must read its batch from tensors whose ADDRESSES stay the same (copy each batch into
    them before step()) and must not synchronise with the host.  The first call runs eagerly (it
    materialises optimizer state, fused storage and allocator pools), the second captures and
    replays, later calls replay.  eager_step() runs one step kernel by kernel at any time (e.g. to
    time individual launches); the sequence of steps stays the same."""

    def __init__(self, model, loss_fn, params, opt: FusedAdamW, grad_scale: float = 1.0):
        self.model, self.loss_fn, self.opt, self.grad_scale = model, loss_fn, opt, grad_scale
        self.params = [p for p in params if p.requires_grad]
        self.graph = None
        self.loss = None
        self._warm = False
        self._seed_dev = None
        self._graph_steps = 0       # steps executed from the graph
        self._step0 = 0             # model._step of the captured step
        self._eager_since_capture = 0

    def eager_step(self):
        self.opt.step_count += 1
        for p in self.params:
            p.grad = None
        if self.graph is not None:
            # the graph left the host-side dropout step counter behind: bring it up to date, and
            # advance the device offset past the step that runs eagerly now
            self.model._step = self._step0 + self._graph_steps + self._eager_since_capture - 1
            self._eager_since_capture += 1
            self._seed_dev.add_(self.model.SEED_STRIDE)
        loss = self.loss_fn()
        loss.backward()
        self.opt.step_params(self.params, self.grad_scale)
        self._warm = True
        return loss.detach()

    def step(self):
        if not self._warm:
            return self.eager_step()
        dev = self.params[0].device
        if self.graph is None:
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()                  # the eager pools: the graph brings its own
            self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
            self.opt.step_count += 1
            self.opt.update_hyper(dev, self.grad_scale)
            self.opt.prepare_graph(self.params)
            for p in self.params:
                p.grad = None
            torch.cuda.synchronize(dev)
            ops.set_dropout_seed_offset(self._seed_dev)
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    loss = self.loss_fn()
                    loss.backward()
                    self.opt.step_params_dev(self.params)
                    self._seed_dev.add_(self.model.SEED_STRIDE)
            finally:
                ops.set_dropout_seed_offset(None)     # eager launches carry their seed as an argument
            self.graph, self.loss = graph, loss.detach()
            self._step0 = getattr(self.model, "_step", 0)
            self._eager_since_capture = 0
        else:
            self.opt.step_count += 1
            self.opt.update_hyper(dev, self.grad_scale)
        self.graph.replay()
        self._graph_steps += 1
        return self.loss

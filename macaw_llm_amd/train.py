"""Whole-step hipGraph replay on top of the step runtime (bucketed.BucketedStep).

The step runtime itself -- flat buckets, ZeRO-1 collectives behind the backward for N > 1, one fused
AdamW launch for N = 1, accumulation, clipping, schedule -- lives in macaw_llm_amd/bucketed.py and is
what bench.py runs at every N.  The per-tensor runtimes of rounds 1-2 (dp.GradSync,
train.OverlappedStep) are test references now (tests/legacy_steps.py)."""
from __future__ import annotations

import torch

from . import ops
from .bucketed import BucketedStep


class GraphedStep:
    """Single-GPU training step replayed from ONE hipGraph: forward -> backward (weight gradients
    stored straight into the flat gradient buckets) -> one fused AdamW launch over the buckets,
    ~1,150 kernels at cfg 3, captured once and launched with a single call per step.  (At cfg 3 it
    buys 0.2 %: the host is far ahead of the GPU in the eager step.  It matters where kernels are
    short, e.g. small models.)  What changes from step to step lives in device memory: the
    optimizer's {lr, bias corrections, grad scale} (FusedAdamW.update_hyper) and the dropout seed
    offset (ops.set_dropout_seed_offset; MM_LLMs._dropout_seed strides by SEED_STRIDE per step), so
    a replay is bit-identical to the eager step it stands for.

    runtime: a BucketedStep WITHOUT collectives (one rank, no max_grad_norm, accumulate_steps 1).
    loss_fn() must read its batch from tensors whose ADDRESSES stay the same (copy each batch into
    them before step()) and must not synchronise with the host.  The first call runs eagerly (it
    materialises optimizer state, the frozen bucket order and allocator pools), the second captures
    and replays, later calls replay.  eager_step() runs one step kernel by kernel at any time (e.g.
    to time individual launches); the sequence of steps stays the same."""

    def __init__(self, model, loss_fn, runtime: BucketedStep, grad_scale: float = 1.0):
        if (runtime.collective or runtime.max_grad_norm is not None or runtime.accumulate_steps != 1
                or getattr(runtime, "loss_scaler", None) is not None):
            raise ValueError("GraphedStep: needs a single-rank BucketedStep without clipping / accumulation / dynamic "
                             "loss scale (collectives and the host reads of the clip factor / overflow verdict "
                             "cannot be captured)")
        self.model, self.loss_fn, self.rt, self.grad_scale = model, loss_fn, runtime, grad_scale
        # the eager steps (warm-up, eager_step()) take their scale from the runtime, the replays from
        # update_hyper(): ONE value, or a replay is not the eager step it stands for
        runtime.grad_scale = float(grad_scale)
        self.opt = runtime.opt
        self.graph = None
        self.loss = None
        self._warm = False
        self._seed_dev = None
        self._graph_steps = 0       # steps executed from the graph
        self._step0 = 0             # model._step of the captured step
        self._eager_since_capture = 0

    def _one_step(self):
        self.rt.begin()
        loss = self.loss_fn()
        loss.backward()
        self.rt.finish()
        return loss

    def eager_step(self):
        if self.graph is not None:
            # the graph left the host-side dropout step counter behind: bring it up to date, and
            # advance the device offset past the step that runs eagerly now
            self.model._step = self._step0 + self._graph_steps + self._eager_since_capture - 1
            self._eager_since_capture += 1
            self._seed_dev.add_(self.model.SEED_STRIDE)
        loss = self._one_step()
        self._warm = True
        return loss.detach()

    def step(self):
        if not self._warm:
            return self.eager_step()
        dev = self.rt.params[0].device
        if self.graph is None:
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()                  # the eager pools: the graph brings its own
            self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
            # (begin() inside the capture advances step_count; the scalars of THAT step go up first)
            self.opt.step_count += 1
            self.opt.update_hyper(dev, self.grad_scale)
            self.opt.step_count -= 1
            torch.cuda.synchronize(dev)
            ops.set_dropout_seed_offset(self._seed_dev)
            self.rt.dev_hyper = True
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    loss = self._one_step()
                    self._seed_dev.add_(self.model.SEED_STRIDE)
            finally:
                self.rt.dev_hyper = False
                ops.set_dropout_seed_offset(None)     # eager launches carry their seed as an argument
            self.graph, self.loss = graph, loss.detach()
            self._step0 = getattr(self.model, "_step", 0)
            self._eager_since_capture = 0
        else:
            self.opt.step_count += 1
            self.opt.update_hyper(dev, self.grad_scale)
        self.graph.replay()
        ops.bump_weight_version()      # the weights changed inside the replay: cached e4m3 copies are stale
        self._graph_steps += 1
        return self.loss

"""Bucketed data-parallel training step (one process per GPU, torch.distributed "nccl" = RCCL over
xGMI; gloo for the CPU tests).

The reference trains under DeepSpeed ZeRO (train.sh:14-16, configs/deepspeed_config.json:22-41:
fp16 parameter all-gathers and gradient reduce-scatters in `hidden^2`-element buckets, gradient
accumulation 3, gradient clipping, cosine schedule with 3 % warm-up).  On MI355X everything fits in
288 GB, so the equivalent here is ZeRO-1 over FLAT BUCKETS:

  * at construction every trainable parameter is re-homed into one of a few large flat buffers
    (`bucket_bytes`, default 768 MiB -> ~18 buckets at 7B), ordered so that a bucket fills up in
    backward order; parameters that already sit back to back (fused q|k|v, gate|up) stay in that
    order.  State-dict keys do not change (the parameters become views, as with
    LlamaDecoderLayer.fuse_projections).
  * a second set of flat buffers of the same layout receives the gradients.  The grad-weight GEMMs
    of the decoder layers and lm_head write STRAIGHT into them (ops.GRAD_DST); any other gradient
    is copied in by its post-accumulate hook.
  * when the last gradient of a bucket has arrived, ONE collective goes out for the whole bucket:
    `reduce_scatter_tensor` into a preallocated shard buffer (each rank receives the mean of its
    1/N slice over all 7 xGMI links at once), then -- on a side stream, behind the remaining
    backward -- fused AdamW on that slice (fp32 master / moments exist only for the slice) and
    `all_gather_into_tensor` of the updated bf16 slice in place into the parameter bucket.
    2 collectives per bucket, <= 40 per step, no allocation inside the step.
  * gradient accumulation: `accumulate_steps` micro-batches add into the gradient buckets; the
    collectives and the update run on the last one only (DDP's no_sync()).
  * `max_grad_norm`: global-norm clipping as HF Trainer / DeepSpeed do it: the reduce-scatters still
    overlap the backward, the updates wait for the global norm (sum of the shard norms^2 +
    one scalar all-reduce) and take the clip factor as AdamW's grad_scale.
  * learning-rate schedule: `set_lr()` per step; `cosine_with_warmup()` is HF's
    get_cosine_schedule_with_warmup (train.sh: --lr_scheduler_type cosine --warmup_ratio 0.03).

With one rank there are no collectives and a shard is the whole bucket; the arithmetic is the
same kernel (mk_adamw) on the same values, so N = 1 and N > 1 agree bit for bit on equal
gradients.  After a step `p.grad` views the LOCAL (unreduced) gradient bucket; the rank-mean
exists only as shards (use `grad_norm` for the global norm).
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import ops


def cosine_with_warmup(step: int, total_steps: int, warmup_ratio: float = 0.03, base_lr: float = 3e-5,
                       num_cycles: float = 0.5) -> float:
    """transformers.get_cosine_schedule_with_warmup with HF's warm-up step count
    (ceil(total * ratio)); `step` = optimizer steps already taken."""
    warm = math.ceil(total_steps * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total_steps - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * prog)))


def _dev(t):
    return t.is_cuda


def _copy_(dst, src):
    if _dev(dst):
        ops.copy2d(src, dst, 1, src.numel(), src.numel(), src.numel())
    else:                      # gloo CPU tests only: the HIP kernels do not exist there
        dst.copy_(src)


def _add_(dst, src):
    if _dev(dst):
        ops.add(dst, src, out=dst)
    else:
        dst.add_(src)


def _zero_(t):
    if _dev(t):
        ops.fill_(t, 0.0)
    else:
        t.zero_()


class _Bucket:
    __slots__ = ("idx", "w", "g", "shard_g", "items", "n", "arrived", "launched", "rs", "missing")

    def __init__(self, idx):
        self.idx = idx
        self.items = []        # (param, offset, numel)
        self.n = 0
        self.arrived = 0
        self.launched = False
        self.rs = None
        self.missing = None


class BucketedStep:
    ALIGN = 64                 # elements: every parameter starts 128-byte aligned inside its bucket

    def __init__(self, params: Iterable[torch.nn.Parameter], opt, process_group=None,
                 bucket_bytes: int = 768 << 20, accumulate_steps: int = 1,
                 max_grad_norm: Optional[float] = None, overlap: bool = True,
                 force_collectives: bool = False, direct_grads: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.opt = opt
        self.group = process_group
        init = dist.is_initialized()
        self.world = dist.get_world_size(process_group) if init else 1
        self.rank = dist.get_rank(process_group) if init else 0
        self._avg = init and dist.get_backend(process_group) == "nccl"   # gloo has no AVG
        self.collective = self.world > 1 or (force_collectives and init)
        self.accumulate_steps = max(1, int(accumulate_steps))
        self.max_grad_norm = max_grad_norm
        self.overlap = overlap
        self.direct_grads = direct_grads
        self.grad_norm = None          # device scalar (fp32) of the last clipped step
        self._micro = 0
        dev = self.params[0].device
        self.side = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and overlap) else None
        self._build(bucket_bytes)
        self._gathers = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    # ------------------------------------------------------------------ layout ---
    def _runs(self):
        """maximal chains of parameters lying back to back in memory (fused q|k|v, gate|up: ascending
        address order, whatever their registration order), single parameters otherwise; chains in
        backward order = descending position of their last-registered member"""
        order = {id(p): i for i, p in enumerate(self.params)}
        by_ptr = {}
        for p in self.params:
            if p.data.is_contiguous():
                by_ptr.setdefault(p.data.data_ptr(), p)
        nxt, has_prev = {}, set()
        for p in self.params:
            if not p.data.is_contiguous():
                continue
            q = by_ptr.get(p.data.data_ptr() + p.numel() * p.element_size())
            if (q is not None and q is not p and q.dtype == p.dtype and q.device == p.device
                    and q.data.untyped_storage().data_ptr() == p.data.untyped_storage().data_ptr()):
                nxt[id(p)] = q
                has_prev.add(id(q))
        runs = []
        for p in self.params:
            if id(p) in has_prev:
                continue
            run = [p]
            while id(run[-1]) in nxt:
                run.append(nxt[id(run[-1])])
            runs.append(run)
        runs.sort(key=lambda r: -max(order[id(x)] for x in r))
        return runs

    @torch.no_grad()
    def _build(self, bucket_bytes):
        quantum = 8 * self.world               # shard boundaries 16-byte aligned in any dtype
        self.buckets: List[_Bucket] = []
        self._where = {}
        by_dtype = {}
        self._run_list = self._runs()
        for run in self._run_list:
            by_dtype.setdefault((run[0].dtype, run[0].device), []).append(run)
        for (dtype, dev), runs in by_dtype.items():
            es = torch.empty((), dtype=dtype).element_size()
            cap = max(1, bucket_bytes // es)
            cur = None
            for run in runs:
                size = sum(p.numel() for p in run)
                if cur is None or (cur.n > 0 and cur.n + size > cap):
                    cur = _Bucket(len(self.buckets))
                    self.buckets.append(cur)
                for j, p in enumerate(run):
                    # keep a fused run gap-free: only the FIRST parameter of a run is aligned
                    if j == 0:
                        cur.n = (cur.n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
                    cur.items.append((p, cur.n, p.numel()))
                    self._where[p] = (cur, cur.n)
                    cur.n += p.numel()
        for b in self.buckets:
            p0 = b.items[0][0]
            total = (b.n + quantum - 1) // quantum * quantum
            b.w = torch.empty(total, dtype=p0.dtype, device=p0.device)
            b.g = torch.empty(total, dtype=p0.dtype, device=p0.device)
            _zero_(b.w)
            _zero_(b.g)
            for p, off, n in b.items:
                src = p.data.contiguous().view(-1)
                _copy_(b.w[off:off + n], src)
                p.data = b.w[off:off + n].view(p.shape)
            b.n = total
            b.shard_g = (torch.empty(total // self.world, dtype=p0.dtype, device=p0.device)
                         if self.collective else None)
        # destinations of the grad-weight GEMMs, keyed (data_ptr, numel): every parameter, and every
        # fused run of 2-D weights with equal inner dimension under its first member's pointer
        self._dst = {}
        for p in self.params:
            bk, off = self._where[p]
            self._dst[(p.data.data_ptr(), p.numel())] = bk.g[off:off + p.numel()].view(p.shape)
        for run in self._run_list:
            if len(run) > 1 and all(x.dim() == 2 and x.shape[1] == run[0].shape[1] for x in run):
                bk, off = self._where[run[0]]
                tot = sum(x.numel() for x in run)
                self._dst[(run[0].data.data_ptr(), tot)] = bk.g[off:off + tot].view(-1, run[0].shape[1])

    def _install_dst(self, on: bool):
        if not (self.direct_grads and self.params[0].is_cuda):
            return
        ops.GRAD_DST.clear()
        if on:
            ops.GRAD_DST.update(self._dst)

    # --------------------------------------------------------------------- step ---
    def set_lr(self, lr: float):
        self.opt.lr = float(lr)

    def begin(self):
        """call before the backward of every micro-batch"""
        first = self._micro == 0
        if first:
            self.opt.step_count += 1
            for b in self.buckets:
                b.arrived, b.launched, b.rs, b.missing = 0, False, None, set()
        else:
            for b in self.buckets:
                b.arrived = 0
        for p in self.params:
            p.grad = None
        # straight-into-the-bucket GEMM stores only on the first micro-batch (later ones ADD)
        self._install_dst(first)

    def _last_micro(self):
        return self._micro == self.accumulate_steps - 1

    def _on_grad(self, p):
        g = p.grad
        if g is None:
            return
        b, off = self._where[p]
        n = p.numel()
        dst = b.g[off:off + n]
        first_time = id(p) not in b.missing      # `missing` doubles as the set of SEEN parameters
        if g.data_ptr() != dst.data_ptr():
            src = g if g.is_contiguous() else g.contiguous()
            (_copy_ if first_time else _add_)(dst, src.view(-1))
        elif not first_time:
            raise RuntimeError("gradient written in place on an accumulation micro-step")
        b.missing.add(id(p))
        p.grad = dst.view(p.shape)
        b.arrived += 1
        if b.arrived == len(b.items) and self._last_micro():
            self._launch(b)

    def _launch(self, b: _Bucket):
        b.launched = True
        if not self.collective:
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        b.rs = (dist.reduce_scatter_tensor(b.shard_g, b.g, op=op, group=self.group, async_op=self.overlap), False)
        if self.max_grad_norm is None and self.side is not None:
            b.shard_g.record_stream(self.side)
            with torch.cuda.stream(self.side):
                self._finish_bucket(b, 1.0)

    def _shard(self, b):
        n = b.n // self.world if self.collective else b.n
        lo = self.rank * n if self.collective else 0
        return lo, n

    def _reduced(self, b: _Bucket):
        """the rank-mean gradient of this rank's slice (waits for the reduce-scatter, stream-ordered)"""
        if not self.collective:
            return b.g
        h, meaned = b.rs
        if h is not None:
            h.wait()
        if not meaned and not self._avg:
            b.shard_g.div_(self.world)        # gloo (tests): SUM -> mean
        b.rs = (None, True)
        return b.shard_g

    def _finish_bucket(self, b: _Bucket, scale: float):
        """update the owned slice from the reduced gradient, gather the updated slices in place"""
        lo, n = self._shard(b)
        gs = self._reduced(b)
        self.opt.step_shard((b.idx, lo, n), b.w[lo:lo + n], gs, scale)
        if self.collective:
            h = dist.all_gather_into_tensor(b.w, b.w[lo:lo + n], group=self.group, async_op=self.overlap)
            if h is not None:
                self._gathers.append(h)

    def finish(self):
        """call after the backward of every micro-batch; on the last one of a window it completes
        the step: buckets whose parameters did not all receive a gradient (an absent modality) are
        flushed with zeros for the missing ones, the global norm is formed if clipping is on, every
        update and all-gather is joined."""
        if not self._last_micro():
            self._micro += 1
            return
        self._micro = 0
        self._install_dst(False)
        queued = self.collective and self.max_grad_norm is None and self.side is not None
        for b in self.buckets:
            if not b.launched:
                if not b.missing:
                    continue                    # nothing in this bucket was used: no update at all
                for p, off, n in b.items:
                    if id(p) not in b.missing:
                        _zero_(b.g[off:off + n])
                self._launch(b)
        active = [b for b in self.buckets if b.launched]
        if not queued:
            scale = self._clip_scale(active) if self.max_grad_norm is not None else 1.0
            for b in active:
                self._finish_bucket(b, scale)
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        for h in self._gathers:                 # the next forward reads the gathered parameters
            h.wait()
        self._gathers.clear()

    def _clip_scale(self, active):
        """grad_scale = min(1, max_norm / (||g|| + 1e-6)) as torch.nn.utils.clip_grad_norm_;
        ||g|| over the rank-mean gradient = sqrt(sum over ranks of the owned shards' norms^2)"""
        dev = self.params[0].device
        tot = torch.zeros(1, dtype=torch.float32, device=dev)
        for b in active:
            gs = self._reduced(b)
            if _dev(gs):
                ops.sumsq(gs, out=tot, accumulate=True)
            else:
                tot += gs.float().pow(2).sum()
        if self.collective:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        self.grad_norm = tot.sqrt()
        # the clip factor is a host scalar of AdamW's launch: one small D2H sync per step, as the
        # reference's trainers pay for their overflow / norm checks
        return min(1.0, float(self.max_grad_norm) / (float(self.grad_norm) + 1e-6))

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        self._install_dst(False)

    # -------------------------------------------------------------- introspection ---
    def describe(self) -> str:
        nb = len(self.buckets)
        mb = sum(b.n * b.w.element_size() for b in self.buckets) / 2 ** 20
        mode = "ZeRO-1 reduce-scatter / shard AdamW / all-gather" if self.collective else "local"
        return (f"{nb} flat buckets, {mb:.0f} MiB of parameters, {mode}"
                + (f", {2 * nb} collectives per step" if self.collective else "")
                + (f", grad accumulation x{self.accumulate_steps}" if self.accumulate_steps > 1 else "")
                + (f", clip {self.max_grad_norm}" if self.max_grad_norm is not None else ""))

"""The training-step runtime: flat parameter / gradient buckets, ZeRO-1 over RCCL for N > 1 ranks,
ONE fused AdamW launch for N = 1 (one process per GPU, torch.distributed "nccl" = RCCL over xGMI;
gloo for the CPU tests).  Since round 3 this is the only step runtime of the package: bench.py runs
it at every N (the per-tensor steps of rounds 1-2 live on as test references, tests/legacy_steps.py).

The reference trains under DeepSpeed ZeRO (train.sh:14-16, configs/deepspeed_config.json:22-41:
fp16 parameter all-gathers and gradient reduce-scatters in `hidden^2`-element buckets, gradient
accumulation 3, gradient clipping, cosine schedule with 3 % warm-up).  On MI355X everything fits in
288 GB, so the equivalent here is ZeRO-1 over FLAT BUCKETS:

  * at construction every trainable parameter is re-homed into one of a few large flat buffers
    (`bucket_bytes`, default 768 MiB -> ~18 buckets at 7B), ordered so that a bucket fills up in
    backward order; parameters that already sit back to back (fused q|k|v, gate|up) stay in that
    order.  State-dict keys do not change (the parameters become views).  Pass `model=` and the
    q|k|v / gate|up projections are fused FIRST (modeling.fuse_model); once a parameter lives in a
    bucket the lazy fusion of modeling.py refuses to move it (ops.PINNED_STORAGE) and begin()
    verifies every step that each parameter still views its bucket slot.
  * a second set of flat buffers of the same layout receives the gradients.  The grad-weight GEMMs,
    the norm-weight / bias reductions and the embedding-table gradient write STRAIGHT into them
    (ops.GRAD_DST); any other gradient is copied in by its post-accumulate hook.
  * N > 1: when the last gradient of a bucket has arrived, ONE collective goes out for the whole
    bucket: `reduce_scatter_tensor` into a preallocated shard buffer (each rank receives the mean of
    its 1/N slice over all 7 xGMI links at once), then -- on a side stream, behind the remaining
    backward -- fused AdamW on that slice (fp32 master / moments exist only for the slice) and
    `all_gather_into_tensor` of the updated bf16 slice in place into the parameter bucket.
    2 collectives per bucket, <= 40 per step, no allocation inside the step.
  * RANK-INVARIANT SCHEDULE.  Collectives must be issued in the same order by every rank, whatever
    gradients each rank's batch produced (a modality absent on one rank leaves that rank's bucket
    incomplete until the end of its backward).  Buckets therefore go out in ONE fixed order: the
    first step is a discovery step (nothing is launched during its backward; the order in which
    rank 0's buckets completed is broadcast and frozen), afterwards bucket k of that order is
    launched as soon as it AND all its predecessors are complete, the rest in finish().  Every
    bucket is launched every step; parameters without a gradient contribute zeros (the ZeRO flat
    partition semantics: their moments decay, weight decay applies).
  * N = 1: no collectives; finish() updates all buckets with ONE multi-tensor AdamW launch over a
    static pointer table (FusedAdamW.step_buckets).
  * gradient accumulation: `accumulate_steps` micro-batches add into the gradient buckets; the
    collectives and the update run on the last one only (DDP's no_sync()) and the SUM is divided
    by accumulate_steps inside AdamW's grad_scale (`average_accumulated`, default on: HF Trainer /
    DeepSpeed divide the loss by gradient_accumulation_steps, train.sh:29).
  * `max_grad_norm`: global-norm clipping as HF Trainer / DeepSpeed do it: the reduce-scatters still
    overlap the backward, the updates wait for the global norm (sum of the shard norms^2 +
    one scalar all-reduce) and take the clip factor as AdamW's grad_scale.
  * `loss_scaler=DynamicLossScaler()`: fp16 training as the reference's DeepSpeed fp16 block does it
    (configs/deepspeed_config.json:14-21: loss_scale 0 = dynamic, initial 2^16, window 1000, hysteresis 2,
    min 1): multiply the loss by `loss_scale` (or use `scale_loss()`), the gradients are unscaled in fp32
    inside AdamW; a step whose rank-mean gradients are not finite is SKIPPED on every rank (same global
    test: the sum of the shard norms^2) and the scale backs off.
  * learning-rate schedule: `set_lr()` per step; `cosine_with_warmup()` is HF's
    get_cosine_schedule_with_warmup (train.sh: --lr_scheduler_type cosine --warmup_ratio 0.03).
  * `zero1=False`: all-reduce of each bucket + replicated update (the plain-DDP form; bench.py's
    fallback if the ZeRO-1 collectives fail on a machine).

With one rank a shard is the whole bucket; the arithmetic is the same kernel body on the same
values, so N = 1 and N > 1 agree bit for bit on equal gradients.  After a step `p.grad` views the
LOCAL (unreduced) gradient bucket; the rank-mean exists only as shards (`grad_norm` = global norm).
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


def shard_batch(global_batch: int, rank: int, world: int):
    """Even split of the global batch (train.sh: per_device_train_batch_size x 8 ranks); returns
    (start, stop) of this rank's samples."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def default_comm_cus() -> int:
    """CUs the GEMMs leave to the collective kernels while a window's collectives are in flight: MACAW_COMM_CUS if
    set, else the RCCL channel cap NCCL_MAX_NCHANNELS (one resident workgroup = one CU per channel), else 16 -- the
    cap bench.py and hf.py put into the environment.  The value is chosen by profiles/r05_overlap_1rank.txt."""
    import os
    for k in ("MACAW_COMM_CUS", "NCCL_MAX_NCHANNELS"):
        v = os.environ.get(k)
        if v:
            try:
                return max(0, int(v))
            except ValueError:
                pass
    return 16


_OVERLAP_GROUP = {}


def overlap_group():
    """The process group the bucket collectives run on when the caller passes none: all ranks, RCCL, and its
    kernels on a HIGH-PRIORITY stream.  Why (profiles/r05_overlap_1rank.txt): HIP maps streams to a handful of
    hardware queues; the stream torch hands the default group came out on the SAME hardware queue as the compute
    stream on the boxes measured, so every collective kernel was serialised with the GEMMs submitted around it --
    "overlapped 0.00 ms" whatever the GEMMs planned for.  Streams of another priority live in another pool of
    hardware queues: the collective's 16 resident workgroups are dispatched beside the GEMM that left them 16 CUs
    (BucketedStep(comm_cus)), and first when both wait for a CU.  MACAW_COMM_NORMAL_PRIORITY=1 keeps the default
    group (A/B).  Returns (group or None, note for the bench line)."""
    import os
    if os.environ.get("MACAW_COMM_NORMAL_PRIORITY"):
        return None, "default process group (normal-priority stream: MACAW_COMM_NORMAL_PRIORITY)"
    # NOTE dist.new_group() is a collective over the WORLD: every rank must construct its BucketedStep (with the same
    # MACAW_COMM_NORMAL_PRIORITY) -- or pass process_group= explicitly.  The cache is keyed by the LIVE default group
    # object: after destroy_process_group() / re-init in one process (tests, notebooks) a stale group is never handed out.
    world = dist.distributed_c10d._get_default_group()
    if _OVERLAP_GROUP.get("world") is not world:
        _OVERLAP_GROUP.clear()
        _OVERLAP_GROUP["world"] = world
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except (TypeError, AttributeError) as e:      # a torch without the option: the default group works
            _OVERLAP_GROUP["group"] = (None, f"default process group (no high-priority option: {e!r})"[:160])
        else:
            _OVERLAP_GROUP["group"] = (dist.new_group(backend="nccl", pg_options=opts),
                                       "own RCCL group on a high-priority stream")
    return _OVERLAP_GROUP["group"]


class DynamicLossScaler:
    """deepspeed.runtime.fp16.loss_scaler.DynamicLossScaler with the reference's settings as defaults
    (configs/deepspeed_config.json:14-21).  update(overflow) after every optimizer step attempt:
    overflow -> the step was skipped; the scale is halved once the hysteresis is used up (floor min_scale);
    otherwise every `window` steps since the last overflow the scale doubles."""

    def __init__(self, init_scale: float = 2.0 ** 16, scale_factor: float = 2.0, window: int = 1000,
                 hysteresis: int = 2, min_scale: float = 1.0, consecutive_hysteresis: bool = False):
        self.scale = float(init_scale)
        self.factor = float(scale_factor)
        self.window = int(window)
        self.hysteresis = int(hysteresis)
        self.min_scale = float(min_scale)
        self.consecutive_hysteresis = bool(consecutive_hysteresis)
        self.cur_hysteresis = int(hysteresis)
        self.cur_iter = 0
        self.last_overflow_iter = -1
        self.skipped = 0

    def update(self, overflow: bool):
        if overflow:
            if self.hysteresis == 1 or self.cur_hysteresis == 1:
                self.scale = max(self.scale / self.factor, self.min_scale)
            else:
                self.cur_hysteresis -= 1
            self.last_overflow_iter = self.cur_iter
            self.skipped += 1
        else:
            if self.consecutive_hysteresis:
                self.cur_hysteresis = self.hysteresis
            if (self.cur_iter - self.last_overflow_iter) % self.window == 0:
                if not self.consecutive_hysteresis:
                    self.cur_hysteresis = self.hysteresis
                self.scale *= self.factor
        self.cur_iter += 1

    def state_dict(self):
        """DeepSpeed saves the loss scaler with the optimizer: scale, window position, hysteresis left"""
        return {k: getattr(self, k) for k in ("scale", "factor", "window", "hysteresis", "min_scale",
                                              "consecutive_hysteresis", "cur_hysteresis", "cur_iter",
                                              "last_overflow_iter", "skipped")}

    def load_state_dict(self, sd):
        for k, v in sd.items():
            setattr(self, k, v)


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
    __slots__ = ("idx", "w", "g", "shard_g", "items", "n", "arrived", "launched", "ready", "rs", "seen", "dirty",
                 "updated")

    def __init__(self, idx):
        self.idx = idx
        self.items = []        # (param, offset, numel)
        self.n = 0
        self.arrived = 0
        self.launched = False
        self.ready = False
        self.rs = None
        self.updated = False   # its AdamW launch went out behind the backward (one rank, local_overlap)
        self.seen = set()      # ids of the parameters that received a gradient in this window
        self.dirty = set()     # ids whose gradient slot may be non-zero (written since it was zeroed)


class BucketedStep:
    ALIGN = 64                 # elements: every parameter starts 128-byte aligned inside its bucket

    def __init__(self, params: Optional[Iterable[torch.nn.Parameter]], opt, process_group=None,
                 bucket_bytes: int = 768 << 20, accumulate_steps: int = 1,
                 max_grad_norm: Optional[float] = None, overlap: bool = True,
                 force_collectives: bool = False, direct_grads: bool = True, model=None,
                 zero1: bool = True, average_accumulated: bool = True,
                 loss_scaler: Optional[DynamicLossScaler] = None, comm_cus: int = 0,
                 local_overlap: Optional[bool] = None):
        if model is not None:
            # fuse q|k|v / gate|up BEFORE the parameters are pinned into buckets (the lazy fusion at
            # the first forward must never re-home a bucket view: round-2 advisor finding)
            from . import modeling as _m
            _m.fuse_model(model)
            if params is None:
                params = model.parameters()
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.opt = opt
        init = dist.is_initialized()
        self.world = dist.get_world_size(process_group) if init else 1
        self.rank = dist.get_rank(process_group) if init else 0
        self._avg = init and dist.get_backend(process_group) == "nccl"   # gloo has no AVG
        self.collective = self.world > 1 or (force_collectives and init)
        self.group = process_group
        self.group_note = "caller's process group"
        if process_group is None and self.collective and overlap and self._avg:
            self.group, self.group_note = overlap_group()
        self.zero1 = bool(zero1)
        self.accumulate_steps = max(1, int(accumulate_steps))
        self.average_accumulated = bool(average_accumulated)
        self.max_grad_norm = max_grad_norm
        self.overlap = overlap
        self.direct_grads = direct_grads
        self.dev_hyper = False         # True while train.GraphedStep captures: AdamW scalars from device memory
        self.serial_update = False     # True: this window updates behind the backward (one launch) even with local_overlap
        self._local_live = False
        self.grad_scale = 1.0          # multiplied into every gradient inside AdamW (set to 1 / loss_scale for
                                       # fp16 training with a scaled loss; set it before the backward)
        self.grad_norm = None          # device scalar (fp32) of the last clipped step
        self.loss_scaler = loss_scaler # dynamic fp16 loss scale: see scale_loss() / loss_scale
        # CUs the collective kernels hold while they overlap the backward (= RCCL channels, NCCL_MAX_NCHANNELS): the
        # 256 x 256 GEMM needs whole CUs, so from the first collective of a step to finish() the tile kernels plan
        # their rounds for (device CUs - comm_cus) -- scripts/probe/cu_hold.cpp: with 16 CUs held a 288-tile dx
        # GEMM runs at 1039 TFLOP/s planned for 256 CUs, 1196 planned for 240 (1270 alone)
        self.comm_cus = int(comm_cus) if (self.collective and overlap) else 0
        self.local_cus = 0             # one rank, local_overlap: CUs the per-bucket AdamW launches may hold (below)
        self._cus_reserved = False
        self.last_step_skipped = False
        self._micro = 0
        self._order = None             # frozen launch order (bucket indices); None until the discovery step ran
        self._cursor = 0
        self._arrival: List[int] = []
        dev = self.params[0].device
        # ONE rank, no collectives (opt-in: local_overlap=True / MACAW_LOCAL_OVERLAP=1): the AdamW launch of a bucket goes
        # out on a HIGH-PRIORITY side stream (another pool of hardware queues: see overlap_group()) as soon as the
        # bucket's gradients are complete, instead of as one launch behind the backward (34 ms = 15 % of the cfg-3 step).
        # MEASURED and therefore OFF by default (profiles/r05_local_overlap_ab.txt, two alternations on one box):
        # 231.5 / 232.3 ms per step overlapped against 229.5 / 230.1 serial -- AdamW is HBM-bound at 5.6 TB/s and takes
        # every CU a finishing GEMM workgroup frees, so the backward's GEMMs lose what the tail gains; the holes a
        # 288-tile GEMM leaves are too short for it.  (With N ranks the same launches update 1/N of a bucket each and
        # the question does not arise.)  Same kernels, same arithmetic, same optimizer-state keys either way
        # (bit-identical weights: tests/test_train_gpu.py).
        #
        # Round 6: the CONFINED form, measured WORSE (profiles/r06_local_overlap_confined.txt) and therefore off
        # (MACAW_LOCAL_CUS = 0).  With `local_cus` = c > 0 the launches behind the backward are capped at 8 c blocks
        # (mk_adamw_set_max_blocks: the grid-stride kernel becomes a persistent update) and the GEMMs plan their rounds
        # for 256 - c CUs (mk_gemm_set_cus, as for RCCL channels).  Step 222 ms serial / 224 unconfined / 241 (c = 8) /
        # 287 (16) / 278 (24) / 260 (32) / 259 (48): the dispatcher does not PACK the blocks -- one resident 4-wave block
        # is enough to keep a 256 x 256 GEMM workgroup (all 512 VGPRs per SIMD, 160 KiB LDS) off a CU, so 8 c blocks
        # poison up to 8 c CUs -- and a CU streams ~22 GB/s (5.65 TB/s / 256), so an update confined to c CUs would need
        # c ~ 75 to finish inside the backward.  AdamW on one rank costs 35 ms x 256 CUs wherever it is put.
        #
        import os as _os
        if local_overlap is None:
            local_overlap = bool(_os.environ.get("MACAW_LOCAL_OVERLAP"))
        self.local_overlap = bool(local_overlap and overlap and not self.collective and dev.type == "cuda")
        if self.local_overlap:
            self.local_cus = max(0, int(_os.environ.get("MACAW_LOCAL_CUS", "0")))
            self.comm_cus = self.local_cus
            self.local_blocks = max(1, int(_os.environ.get("MACAW_LOCAL_BLOCKS", str(8 * self.local_cus)))) if self.local_cus else 0
        self._in_finish = False
        if dev.type == "cuda" and overlap and self.collective:
            self.side = torch.cuda.Stream(device=dev)
        elif self.local_overlap:
            self.side = torch.cuda.Stream(device=dev, priority=-1)
        else:
            self.side = None
        self._build(bucket_bytes)
        self._gathers = []
        self._comm = None              # communication profile of the step in flight (profile_comm())
        self.digests = None            # trace_digests(): per step and bucket, sha1 of each stage of the update
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
                         if (self.collective and self.zero1) else None)
            ops.PINNED_STORAGE.add(b.w.untyped_storage().data_ptr())
        # destinations of the gradient-producing kernels, keyed (data_ptr, numel): every parameter, and
        # every fused run of 2-D weights with equal inner dimension under its first member's pointer
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
        """the table of direct gradient destinations is process-global (the kernels' callers look a weight up by
        address): it belongs to ONE runtime from begin() to finish()"""
        if not (self.direct_grads and self.params[0].is_cuda):
            return
        owner = ops.GRAD_DST_OWNER[0]
        if owner is not None and owner is not self:
            if on:
                raise RuntimeError("BucketedStep.begin(): another BucketedStep is between begin() and finish() "
                                   "(two runtimes cannot share one backward window)")
            return                                   # not ours: leave the other runtime's table alone
        ops.GRAD_DST.clear()
        ops.GRAD_DST_TAKEN.clear()
        ops.GRAD_DST_OWNER[0] = None
        if on:
            ops.GRAD_DST.update(self._dst)
            ops.GRAD_DST_OWNER[0] = self

    def _check_homes(self):
        """every parameter must still view its bucket slot (something re-materialised it otherwise --
        .to(), a manual p.data = ..., a fusion pass -- and the optimizer would update a stale copy)"""
        for b in self.buckets:
            base, es = b.w.data_ptr(), b.w.element_size()
            for p, off, n in b.items:
                if p.data.data_ptr() != base + off * es or p.numel() != n:
                    raise RuntimeError(
                        "BucketedStep: a parameter no longer views its bucket slot (it was moved or re-homed "
                        "after the runtime was built: build BucketedStep AFTER .to() / fusion, or pass model=)")

    # --------------------------------------------------------------------- step ---
    def set_lr(self, lr: float):
        self.opt.lr = float(lr)

    @property
    def loss_scale(self) -> float:
        """what the loss has to be multiplied by before backward() (1 without a scaler)"""
        return self.loss_scaler.scale if self.loss_scaler is not None else 1.0

    def scale_loss(self, loss):
        return loss * self.loss_scale if self.loss_scaler is not None else loss

    def _needs_global(self) -> bool:
        """the updates wait for a statistic of ALL gradients (clip factor / overflow verdict)"""
        return self.max_grad_norm is not None or self.loss_scaler is not None

    def begin(self):
        """call before the backward of every micro-batch"""
        first = self._micro == 0
        if first:
            # a backward / finish() that raised (OOM, a refused optimizer state, a retried step) must not leave the
            # process-global GEMM planning limit behind: every window starts with the whole chip
            self._reserve_cus(False)
            if hasattr(self.opt, "uniform_hyper") and not self.opt.uniform_hyper():
                raise ValueError("BucketedStep: the optimizer's parameter groups differ in lr / betas / eps / "
                                 "weight_decay; the bucket runtime updates every parameter with one launch "
                                 "(build FusedAdamW with one setting, as train.sh:27-31 does)")
            self.opt.step_count += 1
            self._check_homes()
            for b in self.buckets:
                b.arrived, b.launched, b.ready, b.rs = 0, False, False, None
                b.updated = False
                b.seen = set()
            self._cursor = 0
            self._arrival = []
            # (the updates go out behind the backward, on the side stream, unless this window is being captured in a
            #  hipGraph or profiled kernel by kernel: `serial_update`)
            self._local_live = self.local_overlap and not self.dev_hyper and not self.serial_update
        else:
            for b in self.buckets:
                b.arrived = 0
        for p in self.params:
            p.grad = None
        # straight-into-the-bucket stores only on the first micro-batch (later ones ADD)
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
        first_time = id(p) not in b.seen
        if g.data_ptr() != dst.data_ptr():
            src = g if g.is_contiguous() else g.contiguous()
            (_copy_ if first_time else _add_)(dst, src.view(-1))
        elif not first_time:
            raise RuntimeError("gradient written in place on an accumulation micro-step")
        b.seen.add(id(p))
        b.dirty.add(id(p))
        p.grad = dst.view(p.shape)
        b.arrived += 1
        if b.arrived == len(b.items) and self._last_micro():
            b.ready = True
            self._arrival.append(b.idx)
            self._advance()

    def _advance(self):
        """launch the longest prefix of the frozen order whose buckets are complete"""
        if self._order is None:
            return                     # discovery step: everything goes out in finish()
        while self._cursor < len(self._order):
            b = self.buckets[self._order[self._cursor]]
            if not b.ready:
                return
            self._launch(b)
            self._cursor += 1

    def _acc_scale(self) -> float:
        acc = 1.0 / self.accumulate_steps if (self.average_accumulated and self.accumulate_steps > 1) else 1.0
        return acc * float(self.grad_scale) / self.loss_scale

    def _reserve_cus(self, on: bool):
        if not (self.comm_cus > 0 and self.params[0].is_cuda) or on == self._cus_reserved:
            return
        from . import lib as _L
        total = torch.cuda.get_device_properties(self.params[0].device).multi_processor_count
        _L.load().mk_gemm_set_cus(max(8, total - self.comm_cus) if on else 0)
        self._cus_reserved = on

    def _adamw_cap(self, n_blocks: int) -> int:
        """grid cap of the per-shard AdamW launches (process-global, read by the launcher on the host); returns the
        previous cap"""
        from . import lib as _L
        return int(_L.load().mk_adamw_set_max_blocks(int(n_blocks)))

    def _launch(self, b: _Bucket):
        b.launched = True
        if not self.collective:
            if self.local_overlap and self._local_live and not self._needs_global():
                confined = self.local_cus > 0 and not self._in_finish
                if confined:
                    self._reserve_cus(True)                             # the GEMMs that follow plan for 256 - c CUs
                cap = self._adamw_cap(self.local_blocks if confined else 0)
                self.side.wait_stream(torch.cuda.current_stream())      # the bucket's gradients are complete
                try:
                    with torch.cuda.stream(self.side):
                        self._finish_bucket(b, self._acc_scale())
                finally:
                    self._adamw_cap(cap)
                b.updated = True
            return
        self._reserve_cus(True)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self.zero1:
            h = dist.reduce_scatter_tensor(b.shard_g, b.g, op=op, group=self.group, async_op=self.overlap)
        else:
            h = dist.all_reduce(b.g, op=op, group=self.group, async_op=self.overlap)
        b.rs = (h, False)
        if self._comm is not None:
            self._comm["rs"].append((b.idx, h, b.g.numel() * b.g.element_size()))
        if not self._needs_global() and self.side is not None:
            (b.shard_g if self.zero1 else b.g).record_stream(self.side)
            with torch.cuda.stream(self.side):
                self._finish_bucket(b, self._acc_scale())

    def _shard(self, b):
        if self.collective and self.zero1:
            n = b.n // self.world
            return self.rank * n, n
        return 0, b.n

    def _reduced(self, b: _Bucket):
        """the rank-mean gradient of this rank's slice (waits for the collective, stream-ordered)"""
        if not self.collective:
            return b.g
        h, meaned = b.rs
        out = b.shard_g if self.zero1 else b.g
        if h is not None:
            h.wait()
        if not meaned and not self._avg:
            out.div_(self.world)              # gloo (tests): SUM -> mean
        b.rs = (None, True)
        return out

    def _finish_bucket(self, b: _Bucket, scale: float):
        """update the owned slice from the reduced gradient, gather the updated slices in place"""
        lo, n = self._shard(b)
        gs = self._reduced(b)
        self.opt.step_shard((b.idx, lo, n), b.w[lo:lo + n], gs, scale)
        if self.collective and self.zero1:
            h = dist.all_gather_into_tensor(b.w, b.w[lo:lo + n], group=self.group, async_op=self.overlap)
            if h is not None:
                self._gathers.append(h)
            if self._comm is not None:
                self._comm["ag"].append((b.idx, h, b.w.numel() * b.w.element_size()))

    def _zero_missing(self, b: _Bucket):
        """gradient slots of parameters that got no gradient in this window must read as zeros; a slot
        is re-zeroed only if something was written to it since the last time (never-used parameters
        cost nothing per step)"""
        run_lo = run_hi = None
        for p, off, n in b.items:
            if id(p) in b.seen or id(p) not in b.dirty:
                continue
            b.dirty.discard(id(p))
            if run_hi == off:                  # coalesce neighbours into one fill
                run_hi = off + n
                continue
            if run_lo is not None:
                _zero_(b.g[run_lo:run_hi])
            run_lo, run_hi = off, off + n
        if run_lo is not None:
            _zero_(b.g[run_lo:run_hi])

    def _freeze_order(self):
        nb = len(self.buckets)
        done = set(self._arrival)
        order = list(self._arrival) + [i for i in range(nb) if i not in done]
        if self.collective:
            dev = self.params[0].device
            t = torch.tensor(order, dtype=torch.int64, device=dev)
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(t, src=src, group=self.group)
            order = [int(x) for x in t.tolist()]
        if sorted(order) != list(range(nb)):
            raise RuntimeError("BucketedStep: inconsistent bucket order across ranks (different models?)")
        self._order = order

    def finish(self):
        """call after the backward of every micro-batch; on the last one of a window it completes
        the step: the remaining buckets go out in the frozen order (zeros for parameters without a
        gradient), the global norm is formed if clipping is on, every update and all-gather is
        joined."""
        if not self._last_micro():
            self._micro += 1
            return
        self._micro = 0
        try:
            self._finish_window()
        finally:
            # also when the window raised: eval / generate / the next step plan for the whole chip again
            self._reserve_cus(False)

    def abort(self):
        """drop a window whose forward / backward raised (the caller will not reach finish()): the direct
        gradient destinations and the GEMM planning limit are process-global and must not outlive it"""
        self._micro = 0
        self._install_dst(False)
        self._reserve_cus(False)

    def _finish_window(self):
        self._install_dst(False)
        if self._comm is not None and self.params[0].is_cuda:
            self._comm["ev0"] = torch.cuda.Event(enable_timing=True)
            self._comm["ev0"].record()          # behind the last kernel of the backward on the compute stream
        if self._order is None:
            self._freeze_order()
        self._in_finish = True                  # (local_overlap: the backward is over, these updates get the whole chip)
        try:
            while self._cursor < len(self._order):
                b = self.buckets[self._order[self._cursor]]
                self._zero_missing(b)
                self._launch(b)
                self._cursor += 1
        finally:
            self._in_finish = False
        queued = self.collective and not self._needs_global() and self.side is not None
        self.last_step_skipped = False
        if not queued:
            scale = self._acc_scale()
            if self._needs_global():
                scale = self._clip_scale(scale)
            if scale is None:                  # non-finite gradients under a dynamic loss scale: no update on any rank
                self.last_step_skipped = True
                self.opt.step_count -= 1       # (begin() counted this attempt; Adam's bias correction must not)
                for b in self.buckets:
                    if b.rs is not None and b.rs[0] is not None:
                        b.rs[0].wait()
            elif not self.collective and hasattr(self.opt, "step_buckets") and self.buckets[0].w.is_cuda:
                rest = [b for b in self.buckets if not getattr(b, "updated", False)]
                if len(rest) == len(self.buckets):
                    self.opt.step_buckets([((b.idx, 0, b.n), b.w, b.g) for b in self.buckets], scale,
                                          dev_hyper=self.dev_hyper)
                else:
                    for b in rest:                 # (buckets whose gradients completed in finish(): zeros / stragglers)
                        self._finish_bucket(b, scale)
            else:
                for i in self._order:
                    self._finish_bucket(self.buckets[i], scale)
            if self.loss_scaler is not None:
                self.loss_scaler.update(self.last_step_skipped)
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        for h in self._gathers:                 # the next forward reads the gathered parameters
            h.wait()
        self._gathers.clear()
        self._reserve_cus(False)                # the forward has the whole chip again
                                                # (finish() repeats this in its `finally`)
        if self._comm is not None and self.params[0].is_cuda:
            self._comm["ev1"] = torch.cuda.Event(enable_timing=True)
            self._comm["ev1"].record()          # the step is complete on the compute stream
        ops.bump_weight_version()               # cached fp8 copies of the weights are stale now
        if self.digests is not None:
            self._record_digests()
        if getattr(self.opt, "_loaded_keys", None) is not None and not self.last_step_skipped:
            self.opt.assert_restored()          # every checkpoint entry found its slot (else: wrong layout)

    # ------------------------------------------------------------ diagnosis ---
    def trace_digests(self, on: bool = True):
        """DEBUGGING aid (synchronises and copies every bucket to the host each step): after every completed step
        `self.digests` gains one entry per bucket -- sha1 of (a) the LOCAL gradient bucket as the backward left it,
        (b) this rank's reduced slice (behind the reduce-scatter / all-reduce), (c) the bucket after AdamW and the
        all-gather.  Two runs that should be bit-identical are compared stage by stage: the first stage that differs
        names the culprit -- (a) a backward kernel of one of the bucket's parameters, (b) the collective, (c) the
        optimizer kernel or the gather (tests/test_train_gpu.py prints this on a replica / runtime mismatch)."""
        self.digests = [] if on else None

    def _record_digests(self):
        import hashlib

        def dg(t):
            return hashlib.sha1(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]

        if self.params[0].is_cuda:
            torch.cuda.synchronize()
        step = len(self.digests)
        entry = []
        for b in self.buckets:
            lo, n = self._shard(b)
            red = b.shard_g if (self.collective and self.zero1) else b.g[lo:lo + n]
            entry.append({"step": step, "bucket": b.idx, "elements": b.n, "owned": (lo, n),
                          "local_grad": dg(b.g) if (self.collective and self.zero1) else None,
                          "reduced": dg(red), "updated": dg(b.w)})
        self.digests.append(entry)

    def bucket_of(self, named_params) -> dict:
        """parameter name -> bucket index (for messages), given `model.named_parameters()`"""
        where = {id(p): b.idx for b in self.buckets for p, _, _ in b.items}
        return {n: where[id(p)] for n, p in named_params if id(p) in where}

    def _clip_scale(self, acc_scale: float):
        """the global statistic of a step: with `max_grad_norm`,
        grad_scale = acc_scale * min(1, max_norm / (||g|| + 1e-6)) as torch.nn.utils.clip_grad_norm_; with a
        dynamic loss scaler, None when ||g|| is not finite (DeepSpeed's has_overflow: the step is skipped);
        ||g|| over the rank-mean (and micro-batch-mean) gradient = acc_scale * sqrt(sum over ranks of
        the owned shards' norms^2)"""
        dev = self.params[0].device
        tot = torch.zeros(1, dtype=torch.float32, device=dev)
        for i in self._order:
            gs = self._reduced(self.buckets[i])
            if _dev(gs):
                ops.sumsq(gs, out=tot, accumulate=True)
            else:
                tot += gs.float().pow(2).sum()
        if self.collective and self.zero1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        self.grad_norm = tot.sqrt() * acc_scale
        # the clip factor / overflow verdict is a host scalar of AdamW's launch: one small D2H sync per
        # step, as the reference's trainers pay for their overflow / norm checks
        norm = float(self.grad_norm)
        if self.loss_scaler is not None and not math.isfinite(norm):
            return None                        # identical on every rank: `tot` was all-reduced
        if self.max_grad_norm is None:
            return acc_scale
        return acc_scale * min(1.0, float(self.max_grad_norm) / (norm + 1e-6))

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        self._install_dst(False)
        self._reserve_cus(False)
        for b in self.buckets:
            ops.PINNED_STORAGE.discard(b.w.untyped_storage().data_ptr())
        ops.clear_fp8_cache()                   # e4m3 copies of weights that lived in these buckets

    # ---------------------------------------------------------- checkpoint / resume ---
    def layout(self) -> dict:
        """what the optimizer's shard keys depend on: a checkpoint resumes only into the same layout"""
        # the shard keys depend on the rank only under ZeRO-1; in all-reduce mode the state is REPLICATED (HF's stock
        # save writes rank 0's file and every rank reloads it: ADVICE r5) -- rank 0 there
        sharded = bool(self.zero1 and self.collective)
        return {"world": self.world if self.collective else 1, "rank": self.rank if sharded else 0,
                "zero1": sharded, "bucket_elems": [int(b.n) for b in self.buckets],
                "dtypes": [str(b.w.dtype) for b in self.buckets]}

    def state_dict(self) -> dict:
        """optimizer shards of THIS rank (fp32 master + moments per bucket slot, step counter, hyper-parameters),
        the bucket layout they are keyed by and the dynamic loss scaler -- what DeepSpeed writes per rank
        (`*_optim_states.pt`).  The 16-bit parameters are the model's own state dict."""
        return {"optimizer": self.opt.state_dict(layout=self.layout()),
                "loss_scaler": self.loss_scaler.state_dict() if self.loss_scaler is not None else None,
                "grad_scale": float(self.grad_scale)}

    def load_state_dict(self, sd: dict):
        """raises if the checkpoint was written with another world size / rank / bucket size (the shard keys
        would match nothing and the moments would silently restart at zero under a restored step counter)"""
        self.opt.load_state_dict(sd["optimizer"], layout=self.layout())
        if sd.get("loss_scaler") is not None:
            if self.loss_scaler is None:
                raise ValueError("BucketedStep.load_state_dict: the checkpoint carries a dynamic loss scaler, this "
                                 "runtime has none (pass loss_scaler=DynamicLossScaler())")
            self.loss_scaler.load_state_dict(sd["loss_scaler"])
        self.grad_scale = float(sd.get("grad_scale", self.grad_scale))

    # -------------------------------------------------------------- introspection ---
    def profile_comm(self, on: bool = True):
        """arm / disarm the communication profile of the NEXT step (bench.py arms it for its last timed step;
        read it with comm_report() after a synchronize)"""
        self._comm = {"rs": [], "ag": [], "ev0": None, "ev1": None} if on else None

    def comm_report(self) -> Optional[dict]:
        """what the first multi-GPU run needs to be read: the un-overlapped tail of the step
        -- compute-stream time from the end of the backward to the end of finish(): the remaining collectives, the
        shard updates and the all-gathers that did not fit behind the backward (at N = 1 it is the fused AdamW
        launch: compare) -- and per bucket, in launch order, the duration and the bytes of its reduce-scatter
        (all-reduce with zero1=False) and its all-gather.  Durations are the process group's own device-side
        timing of each collective (ProcessGroupNCCL with TORCH_NCCL_ENABLE_TIMING=1, which bench.py sets); None
        where the backend does not time its work objects (gloo)."""
        c = self._comm
        if c is None:
            return None

        def dur(h):
            try:
                d = h._get_duration() if h is not None else None       # ms (NCCL work, timing enabled)
                return round(float(d), 3) if d is not None else None
            except Exception:                                          # noqa: BLE001  (backend without timing)
                return None

        tail = None
        if c["ev0"] is not None and c["ev1"] is not None:
            c["ev1"].synchronize()
            tail = round(c["ev0"].elapsed_time(c["ev1"]), 3)
        rs = [(i, dur(h), nbytes) for i, h, nbytes in c["rs"]]
        ag = [(i, dur(h), nbytes) for i, h, nbytes in c["ag"]]

        def tot(xs):
            ds = [d for _, d, _ in xs if d is not None]
            return round(sum(ds), 3) if ds else None

        return {"world": self.world, "collective": ("reduce_scatter+all_gather" if self.zero1 else "all_reduce")
                if self.collective else "none",
                "buckets": len(self.buckets), "tail_after_backward_ms": tail,
                "rs_ms": [d for _, d, _ in rs], "ag_ms": [d for _, d, _ in ag],
                "rs_total_ms": tot(rs), "ag_total_ms": tot(ag),
                "rs_bytes": sum(n for _, _, n in rs), "ag_bytes": sum(n for _, _, n in ag),
                "bucket_order": [i for i, _, _ in rs],
                "comm_cus": self.comm_cus, "group": self.group_note if self.collective else None,
                "timing": "per-collective device time from the process group (TORCH_NCCL_ENABLE_TIMING)"
                if any(d is not None for _, d, _ in rs + ag) else "collectives not timed by this backend"}

    def describe(self) -> str:
        nb = len(self.buckets)
        mb = sum(b.n * b.w.element_size() for b in self.buckets) / 2 ** 20
        if not self.collective:
            mode = ("local (per-bucket AdamW launches behind the backward on a high-priority side stream"
                    + (f", confined to {self.local_blocks} blocks, GEMMs leave {self.local_cus} CUs" if self.local_cus else "") + ")"
                    if self.local_overlap and not self._needs_global() else "local (one fused AdamW launch)")
        elif self.zero1:
            mode = "ZeRO-1 reduce-scatter / shard AdamW / all-gather"
        else:
            mode = "all-reduce / replicated AdamW"
        ncoll = (2 * nb if self.zero1 else nb) if self.collective else 0
        return (f"{nb} flat buckets, {mb:.0f} MiB of parameters, {mode}"
                + (f", {ncoll} collectives per step in a fixed rank-invariant order" if ncoll else "")
                + (f", grad accumulation x{self.accumulate_steps}"
                   + (" (mean)" if self.average_accumulated else " (sum)") if self.accumulate_steps > 1 else "")
                + (f", clip {self.max_grad_norm}" if self.max_grad_norm is not None else "")
                + (f", dynamic loss scale {self.loss_scale:g}" if self.loss_scaler is not None else "")
                + (f", GEMMs plan for {self.comm_cus} fewer CUs while collectives are in flight" if self.comm_cus else "")
                + (f", collectives on {self.group_note}" if self.collective else ""))

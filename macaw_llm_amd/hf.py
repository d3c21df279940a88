"""`transformers.Trainer` on the MI355X step runtime.

The reference trains with `LLMTrainer(Trainer)` (llm_trainer.py:183-188, built in run_clm_llms.py:541-552 and
driven by `trainer.train()` / `trainer.save_model()`, :561-563) under DeepSpeed (train.sh:16).  The model of
this package drops into that trainer as it is; THIS module makes the measured step runtime drop in as well:

    class LLMTrainer(MacawTrainerMixin, Trainer):        # the reference's class + one base
        def compute_loss(self, model, inputs, return_outputs=False, **kw):
            inputs = self.get_self_inputs(inputs)
            loss = model(**inputs)[0]
            return loss

What the mixin changes, and nothing else of the Trainer (data loading, logging, callbacks, LR scheduler,
checkpoint cadence, `save_model`, `resume_from_checkpoint` stay HF's):

  * `create_optimizer()` builds `optim.FusedAdamW` (a `torch.optim.Optimizer`) from the TrainingArguments
    (`learning_rate`, `adam_beta1/2`, `adam_epsilon`, `weight_decay`) -- or keeps the one passed as
    `Trainer(optimizers=(opt, sched))` if it is a FusedAdamW.
  * `training_step()` runs forward + backward between `BucketedStep.begin()` / `finish()`: the gradients are
    written straight into the flat buckets, ZeRO-1 reduce-scatter / shard AdamW / all-gather overlap the
    backward for N > 1 ranks, one fused AdamW launch for N = 1; `gradient_accumulation_steps` is the runtime's
    accumulation window (the collectives and the update run on the window's last micro-batch, as DDP's
    `no_sync()`), `max_grad_norm` its global-norm clip.  The update has happened when `training_step` of the
    last micro-batch returns, so `optimizer.step()` of HF's loop is a no-op and `_clip_grad_norm()` reports the
    norm the runtime already clipped with (clipping twice would be wrong).
  * the model is NOT wrapped in `DistributedDataParallel` (the runtime reduces the gradients itself -- a
    second all-reduce per parameter would double the xGMI traffic): `_wrap_model()` hands back a
    pass-through module when more than one rank trains.
  * checkpoints: `optimizer.state_dict()` carries this rank's fp32 master / moment shards, the bucket layout they
    are keyed by and the dynamic loss scaler (fp16); `load_state_dict` refuses another world size / rank /
    bucket size instead of silently restarting the moments.  With N > 1 ranks under ZeRO-1 every rank owns a
    different slice of the optimizer state, so `_save_optimizer_and_scheduler` / `_load_optimizer_and_scheduler`
    write and read ONE FILE PER RANK (`optimizer_rank{r}-of-{N}.pt`, as DeepSpeed's `*_optim_states.pt`); HF's
    stock code would save rank 0's shards only and hand them to every rank on resume.  N = 1 keeps HF's
    `optimizer.pt`.
  * the GEMMs plan for `comm_cus` fewer CUs while collectives are in flight (`bucketed.default_comm_cus()`: the
    RCCL channel cap, 16 unless NCCL_MAX_NCHANNELS / MACAW_COMM_CUS say otherwise -- the same default `bench.py`
    uses), and the LR scheduler does not advance on a step the dynamic loss scaler skipped (DeepSpeed's rule).

`--deepspeed configs/deepspeed_config.json` (train.sh:16) is NOT used with this mixin: the runtime IS the ZeRO-1
equivalent ("DeepSpeed").
"""
from __future__ import annotations

import os

import torch

from .bucketed import BucketedStep, DynamicLossScaler, default_comm_cus
from .optim import FusedAdamW

# the channel cap has to be in the environment before RCCL builds its communicator (the first collective)
if ("NCCL_MAX_NCHANNELS" not in os.environ and torch.distributed.is_available()
        and torch.distributed.is_initialized()):
    import warnings
    warnings.warn("macaw_llm_amd.hf imported after init_process_group() without NCCL_MAX_NCHANNELS in the environment: "
                  "if RCCL has already built its communicator the 16-channel cap set here does not apply and "
                  "comm_cus (16 CUs left to the collectives) may not match the channels in use; export "
                  "NCCL_MAX_NCHANNELS=16 in the launcher or pass macaw_comm_cus", stacklevel=2)
os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")


class _PassThrough(torch.nn.Module):
    """what `_wrap_model` returns for N > 1: forwards everything to the model, adds no hooks, so that
    `accelerator.prepare` does not wrap the model in DistributedDataParallel"""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)


def unwrap_optimizer(opt):
    """accelerate's AcceleratedOptimizer (and torch's wrappers) keep the real one in `.optimizer`"""
    seen = 0
    while not isinstance(opt, FusedAdamW) and hasattr(opt, "optimizer") and seen < 4:
        opt, seen = opt.optimizer, seen + 1
    return opt


class MacawTrainerMixin:
    macaw_bucket_bytes = 768 << 20      # flat bucket size (bucketed.BucketedStep)
    macaw_zero1 = True                  # ZeRO-1 (reduce-scatter / shard AdamW / all-gather); False: all-reduce
    macaw_dynamic_loss_scale = None     # None: on iff the parameters are fp16 (configs/deepspeed_config.json:14-21)
    macaw_comm_cus = None               # CUs left to the collective kernels; None: bucketed.default_comm_cus()

    # ---- optimizer -------------------------------------------------------------------------------------------
    def create_optimizer(self, model=None):
        if self.optimizer is None:
            m = model if model is not None else self.model
            a = self.args
            self.optimizer = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=a.learning_rate,
                                        betas=(a.adam_beta1, a.adam_beta2), eps=a.adam_epsilon,
                                        weight_decay=a.weight_decay)
        elif not isinstance(unwrap_optimizer(self.optimizer), FusedAdamW):
            raise TypeError("MacawTrainerMixin: the bucket runtime updates through optim.FusedAdamW; pass "
                            "optimizers=(FusedAdamW(...), scheduler) or leave the optimizer to the trainer")
        return self.optimizer

    # ---- runtime ---------------------------------------------------------------------------------------------
    def macaw_runtime(self) -> BucketedStep:
        rt = getattr(self, "_macaw_rt", None)
        if rt is None:
            opt = unwrap_optimizer(self.optimizer)
            a = self.args
            scaler = None
            dyn = self.macaw_dynamic_loss_scale
            p0 = next(p for p in self.model.parameters() if p.requires_grad)
            if dyn or (dyn is None and p0.dtype == torch.float16):
                scaler = DynamicLossScaler()
            rt = BucketedStep(None, opt, model=self.model, bucket_bytes=self.macaw_bucket_bytes,
                              accumulate_steps=a.gradient_accumulation_steps,
                              max_grad_norm=a.max_grad_norm if (a.max_grad_norm or 0) > 0 else None,
                              zero1=self.macaw_zero1, loss_scaler=scaler,
                              comm_cus=default_comm_cus() if self.macaw_comm_cus is None else int(self.macaw_comm_cus))
            opt.attach_runtime(rt)
            self._macaw_rt = rt
        return rt

    # ---- LR schedule: a step the loss scaler skipped does not count (DeepSpeed) --------------------
    def create_scheduler(self, num_training_steps, optimizer=None):
        sched = super().create_scheduler(num_training_steps, optimizer=optimizer)
        sched = sched if sched is not None else self.lr_scheduler
        if sched is not None and not getattr(type(sched), "_macaw_wrapped", False):
            trainer, base = self, type(sched)

            def step(sched_self, *a, **kw):
                rt = getattr(trainer, "_macaw_rt", None)
                if rt is not None and rt.last_step_skipped:
                    return None
                return base.step(sched_self, *a, **kw)

            # a subclass, not an instance attribute: LRScheduler.state_dict() is the instance __dict__
            sched.__class__ = type(base.__name__, (base,), {"step": step, "_macaw_wrapped": True})
        return sched

    # ---- checkpoints: one optimizer file per rank under ZeRO-1 ------------------------------------
    def _macaw_sharded(self) -> bool:
        return (torch.distributed.is_available() and torch.distributed.is_initialized()
                and torch.distributed.get_world_size() > 1 and self.macaw_zero1)

    @staticmethod
    def _macaw_opt_file(rank: int, world: int) -> str:
        return f"optimizer_rank{rank}-of-{world}.pt"

    def _save_optimizer_and_scheduler(self, output_dir):
        if not self._macaw_sharded():
            return super()._save_optimizer_and_scheduler(output_dir)
        r, n = torch.distributed.get_rank(), torch.distributed.get_world_size()
        os.makedirs(output_dir, exist_ok=True)
        torch.save(unwrap_optimizer(self.optimizer).state_dict(), os.path.join(output_dir, self._macaw_opt_file(r, n)))
        if self.args.should_save:
            torch.save(self.lr_scheduler.state_dict(), os.path.join(output_dir, "scheduler.pt"))
        torch.distributed.barrier()          # rotation / the next step must not race a rank that is still writing

    def _load_optimizer_and_scheduler(self, checkpoint):
        if checkpoint is None or not self._macaw_sharded():
            return super()._load_optimizer_and_scheduler(checkpoint)
        r, n = torch.distributed.get_rank(), torch.distributed.get_world_size()
        f = os.path.join(checkpoint, self._macaw_opt_file(r, n))
        if not os.path.isfile(f):
            others = sorted(x for x in os.listdir(checkpoint) if x.startswith("optimizer_rank"))
            raise FileNotFoundError(
                f"MacawTrainerMixin: {f} not found (the checkpoint holds {others or 'no per-rank optimizer files'}): "
                "ZeRO-1 optimizer shards resume into the world size they were written with")
        unwrap_optimizer(self.optimizer).load_state_dict(torch.load(f, map_location=self.args.device, weights_only=True))
        sf = os.path.join(checkpoint, "scheduler.pt")
        if os.path.isfile(sf):
            self.lr_scheduler.load_state_dict(torch.load(sf, map_location="cpu", weights_only=True))

    def _wrap_model(self, model, training=True, dataloader=None):
        if training and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            if not isinstance(model, _PassThrough):
                model = _PassThrough(model)
            return model
        return super()._wrap_model(model, training=training, dataloader=dataloader)

    # ---- one micro-batch -------------------------------------------------------------------------------------
    def training_step(self, model, inputs, num_items_in_batch=None):
        rt = self.macaw_runtime()
        model.train()
        inputs = self._prepare_inputs(inputs)
        if rt._micro == 0:
            # HF's last window of an epoch may hold fewer micro-batches (Trainer._run_epoch)
            rt.accumulate_steps = max(1, int(getattr(self, "current_gradient_accumulation_steps",
                                                     self.args.gradient_accumulation_steps)))
        try:
            rt.begin()
            with self.compute_loss_context_manager():
                loss = self.compute_loss(model, inputs)
            rt.scale_loss(loss).backward()
            rt.finish()
        except BaseException:
            # a window whose forward / backward raised at micro-step k > 0 must not leave _micro = k and the direct
            # gradient destinations installed: the next begin() would accumulate onto stale bucket contents
            rt.abort()
            raise
        # the runtime averages the window inside AdamW; HF sums what training_step returns over the window
        return loss.detach() / rt.accumulate_steps

    def _clip_grad_norm(self, model):
        """the runtime clipped with the global norm inside finish(): report it, do not clip again"""
        return self.macaw_runtime().grad_norm

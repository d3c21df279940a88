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
    bucket size instead of silently restarting the moments.

`--deepspeed configs/deepspeed_config.json` (train.sh:16) is NOT used with this mixin: the runtime IS the ZeRO-1
equivalent ("DeepSpeed").
"""
from __future__ import annotations

import torch

from .bucketed import BucketedStep, DynamicLossScaler
from .optim import FusedAdamW


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
                              zero1=self.macaw_zero1, loss_scaler=scaler)
            opt.attach_runtime(rt)
            self._macaw_rt = rt
        return rt

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
        rt.begin()
        with self.compute_loss_context_manager():
            loss = self.compute_loss(model, inputs)
        rt.scale_loss(loss).backward()
        rt.finish()
        # the runtime averages the window inside AdamW; HF sums what training_step returns over the window
        return loss.detach() / rt.accumulate_steps

    def _clip_grad_norm(self, model):
        """the runtime clipped with the global norm inside finish(): report it, do not clip again"""
        return self.macaw_runtime().grad_norm

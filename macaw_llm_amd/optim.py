"""Fused AdamW on the GPU (replaces the reference's DeepSpeed CPU-offloaded Adam,
configs/deepspeed_config.json:2-13,24-27): fp32 master weights and moments, model-dtype
(bf16) parameters and gradients, one mk_adamw launch per parameter tensor."""
from __future__ import annotations

import torch

from . import ops


class _Slot(tuple):
    """(master, exp_avg, exp_avg_sq) of one parameter / bucket shard.  A tuple for the kernels' callers, a mapping
    for code that walks `optimizer.state` the torch way (accelerate / HF move or inspect `state.values()` items)."""
    _NAMES = ("master", "exp_avg", "exp_avg_sq")

    def items(self):
        return zip(self._NAMES, self)

    def keys(self):
        return iter(self._NAMES)

    def values(self):
        return iter(self)

    def __getitem__(self, k):
        return tuple.__getitem__(self, self._NAMES.index(k) if isinstance(k, str) else k)


class FusedAdamW(torch.optim.Optimizer):
    """A `torch.optim.Optimizer`: `param_groups`, `step(closure)`, `zero_grad()`, `state_dict()` /
    `load_state_dict()`, `add_param_group()`, LR schedulers (`torch.optim.lr_scheduler.*`, HF's
    `get_cosine_schedule_with_warmup`) and `transformers.Trainer(optimizers=(opt, sched))` work on it.  The
    hyper-parameters live in `param_groups` (the single source of truth: `opt.lr` etc. are views of group 0);
    the bucket runtime (bucketed.BucketedStep) updates every parameter with ONE launch and therefore needs the
    groups to agree on lr / betas / eps / weight_decay at step time (the reference trains with one setting:
    train.sh:27-31 `--weight_decay 0.`); the per-parameter `step()` honours each group's own values."""

    def __init__(self, params, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        if params and isinstance(params[0], dict):
            groups = [dict(g, params=[p for p in g["params"] if p.requires_grad]) for g in params]
            groups = [g for g in groups if g["params"]]
        else:
            groups = [p for p in params if p.requires_grad]
        if not groups:
            raise ValueError("FusedAdamW: no trainable parameters")
        super().__init__(groups, dict(lr=float(lr), betas=tuple(betas), eps=float(eps),
                                      weight_decay=float(weight_decay)))
        self.step_count = 0
        self._hp_override = None  # the group step() is updating (its lr / betas / eps / weight_decay): see _hp
        self._index = {}          # id(parameter) -> position in the checkpoint's numbering (stable: never derived
        self._reindex()           # from a temporarily narrowed param_groups -- ADVICE r4)
        self._stashed = None      # loss-scaler state / layout of a checkpoint loaded before the runtime existed
        self._pending = {}        # loaded state waiting for its (lazily created) slot: see load_state_dict
        self._loaded_keys = None  # key names of the last load_state_dict (None: nothing was loaded)
        self._loaded_layout = None

    # ---- hyper-parameters: views of param_groups[0] (schedulers write group["lr"]) --------------------------
    @property
    def params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def _reindex(self):
        self._index = {id(p): i for i, p in enumerate(p for g in self.param_groups for p in g["params"])}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_index"):
            self._reindex()

    def _hp(self, name):
        g = self._hp_override if self._hp_override is not None else self.param_groups[0]
        return g[name]

    def _set_hp(self, name, v):
        for g in self.param_groups:
            g[name] = v

    lr = property(lambda self: float(self._hp("lr")), lambda self, v: self._set_hp("lr", float(v)))
    betas = property(lambda self: tuple(self._hp("betas")), lambda self, v: self._set_hp("betas", tuple(v)))
    eps = property(lambda self: float(self._hp("eps")), lambda self, v: self._set_hp("eps", float(v)))
    weight_decay = property(lambda self: float(self._hp("weight_decay")),
                            lambda self, v: self._set_hp("weight_decay", float(v)))

    def attach_runtime(self, runtime):
        """the bucket runtime that owns the update from now on (hf.MacawTrainerMixin): `step()` becomes a no-op
        (the update happens in runtime.finish()), `state_dict()` / `load_state_dict()` carry and check the
        runtime's bucket layout and its dynamic loss scaler.  A checkpoint loaded BEFORE the runtime existed (HF's
        Trainer restores the optimizer ahead of the first training_step, which is where the mixin builds the
        runtime) left its loss-scaler state and layout stashed: they are applied / verified here."""
        import weakref
        self._runtime_steps = True
        self._runtime_ref = weakref.ref(runtime)
        st, self._stashed = self._stashed, None
        if st is not None:
            saved, scaler = st
            layout = runtime.layout()
            if saved is not None and saved != layout:
                raise ValueError(f"FusedAdamW: the checkpoint loaded before the runtime was built has layout {saved}, "
                                 f"this runtime has {layout} (ZeRO-1 shards are per world size / rank / bucket size)")
            if scaler is not None and runtime.loss_scaler is not None:
                runtime.loss_scaler.load_state_dict(scaler)

    def _runtime(self):
        ref = getattr(self, "_runtime_ref", None)
        return ref() if ref is not None else None

    def uniform_hyper(self) -> bool:
        """True if every group has group 0's lr / betas / eps / weight_decay (what a one-launch update needs)"""
        g0 = self.param_groups[0]
        return all(float(g["lr"]) == float(g0["lr"]) and tuple(g["betas"]) == tuple(g0["betas"])
                   and float(g["eps"]) == float(g0["eps"]) and float(g["weight_decay"]) == float(g0["weight_decay"])
                   for g in self.param_groups[1:])

    def _state(self, p):
        st = self.state.get(p)
        if st is None:
            master = ops.cast(p.detach().contiguous(), torch.float32) if p.dtype != torch.float32 \
                else p.detach().clone()
            m = torch.empty_like(master)
            v = torch.empty_like(master)
            ops.fill_(m, 0.0)
            ops.fill_(v, 0.0)
            st = self.state[p] = _Slot((master, m, v))
            self._restore(p, st)
        return st

    # ---- checkpoint / resume (HF Trainer saves `optimizer.state_dict()` every save_steps, train.sh:24-26) ----
    def _key_name(self, key):
        if isinstance(key, tuple):
            return "shard:%d:%d:%d" % key             # (bucket, first element, elements) of bucketed.BucketedStep
        i = self._index.get(id(key))
        if i is None:
            self._reindex()                           # (param_groups edited in place by the caller)
            i = self._index.get(id(key))
        if i is None:
            raise KeyError("FusedAdamW: state of a parameter that is not in self.params")
        return "param:%d" % i

    def _restore(self, key, st):
        if self._loaded_keys is None:
            return
        name = self._key_name(key)
        src = self._pending.pop(name, None)
        if src is None:
            if name in self._loaded_keys:
                return                                # restored before (the slot was re-created)
            raise KeyError(f"FusedAdamW: optimizer slot {name} has no entry in the loaded checkpoint: it was saved "
                           f"with a different bucket layout / world size / rank ({self._loaded_layout}); a silent "
                           "restart of the moments with the restored step counter would mis-scale the first updates")
        for dst, t in zip(st, src):
            if dst.shape != t.shape:
                raise ValueError(f"FusedAdamW.load_state_dict: {name} has {tuple(t.shape)} elements "
                                 f"in the checkpoint, {tuple(dst.shape)} here (different bucket layout / world size)")
            dst.copy_(t.to(dst.device))

    def assert_restored(self):
        """after the first completed step that follows load_state_dict(): every checkpoint entry must have found
        its slot (leftovers = entries keyed by another world size / rank / bucket size, which would otherwise be
        dropped silently)"""
        if self._pending:
            left = sorted(self._pending)
            self._pending = {}
            raise RuntimeError(f"FusedAdamW.load_state_dict: {len(left)} checkpoint entries matched no optimizer slot "
                               f"(first: {left[:3]}; saved layout {self._loaded_layout}): the checkpoint belongs to a "
                               "different world size / rank / bucket layout")

    def state_dict(self, layout=None):
        """fp32 master weights and both moments of every slot this rank owns (under ZeRO-1: its shards only, as
        DeepSpeed's per-rank optimizer files), the step counter, the hyper-parameters of every group and -- from
        the bucket runtime -- the LAYOUT the shard keys depend on (world, rank, elements per bucket).  Tensors are
        the live ones: clone (or torch.save) before training on."""
        rt = self._runtime()
        if layout is None and rt is not None:
            layout = rt.layout()
        scaler = rt.loss_scaler.state_dict() if (rt is not None and rt.loss_scaler is not None) else None
        return {"step_count": self.step_count, "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps,
                "weight_decay": self.weight_decay, "loss_scaler": scaler,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} | {"params": len(g["params"])}
                                 for g in self.param_groups],
                "layout": layout,
                "state": {self._key_name(k): {"master": st[0], "exp_avg": st[1], "exp_avg_sq": st[2]}
                          for k, st in self.state.items()}}

    def load_state_dict(self, sd, layout=None):
        """resume: slots that exist are overwritten now, the others when their first step creates them (state is
        created lazily); the bf16 / fp16 parameters themselves come from the model's own state dict.  A slot
        created later that finds no entry raises, and so do entries nobody claimed (assert_restored(), called by
        the bucket runtime after the first step): shard keys depend on world size, rank and bucket size.
        `layout`: the current runtime's layout, compared with the saved one up front."""
        saved = sd.get("layout")
        rt = self._runtime()
        if layout is None and rt is not None:
            layout = rt.layout()
        if rt is not None and rt.loss_scaler is not None and sd.get("loss_scaler") is not None:
            rt.loss_scaler.load_state_dict(sd["loss_scaler"])
        if rt is None and layout is None and (saved is not None or sd.get("loss_scaler") is not None):
            self._stashed = (saved, sd.get("loss_scaler"))      # applied / verified in attach_runtime()
        if layout is not None and saved is not None and saved != layout:
            raise ValueError(f"FusedAdamW.load_state_dict: the checkpoint was written with layout {saved}, this "
                             f"runtime has {layout} (ZeRO-1 shards are per world size / rank / bucket size; re-shard "
                             "offline or resume with the saved configuration)")
        names = set(sd["state"])
        for k in names:
            if k.startswith("param:") and int(k.split(":")[1]) >= len(self.params):
                raise ValueError(f"FusedAdamW.load_state_dict: entry {k} but only {len(self.params)} parameters")
        self.step_count = int(sd["step_count"])
        groups = sd.get("param_groups")
        if groups is not None and len(groups) == len(self.param_groups):
            for g, sg in zip(self.param_groups, groups):
                if sg.get("params") != len(g["params"]):
                    raise ValueError("FusedAdamW.load_state_dict: parameter groups differ from the checkpoint's")
                g.update({k: (tuple(v) if k == "betas" else v) for k, v in sg.items() if k != "params"})
        else:
            self.lr, self.betas, self.eps = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"])
            self.weight_decay = float(sd["weight_decay"])
        self._pending = {k: (t["master"], t["exp_avg"], t["exp_avg_sq"]) for k, t in sd["state"].items()}
        # (an EMPTY state at step 0 is not a checkpoint: accelerate round-trips a fresh optimizer's state_dict
        # through load_state_dict when it wraps it)
        self._loaded_keys = names if (names or self.step_count) else None
        self._loaded_layout = saved
        for key, st in self.state.items():
            self._restore(key, st)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None                # (the gradient buckets are re-zeroed by the runtime where needed)

    @torch.no_grad()
    def step_param(self, p, grad_scale: float = 1.0):
        """update ONE parameter on the current stream with the current step_count
        (used by train.OverlappedStep from gradient hooks)"""
        if p.grad is None:
            return
        ops.bump_weight_version()
        b1, b2 = self.betas
        master, m, v = self._state(p)
        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        w = p.data
        if not w.is_contiguous():
            raise ValueError("FusedAdamW: parameter storage must be contiguous (a strided view would be "
                             "updated at the wrong elements)")
        if (w.data_ptr() | g.data_ptr()) & 15:
            # mk_adamw moves 16-byte vectors: an odd-offset slice of a fused buffer is updated through
            # an aligned staging copy (rare: tiny ragged tensors only)
            wa = torch.empty_like(w)
            ops.copy2d(w, wa, 1, w.numel(), w.numel(), w.numel())
            ga = torch.empty_like(g)
            ops.copy2d(g, ga, 1, g.numel(), g.numel(), g.numel())
            ops.adamw_(wa, master, m, v, ga, self.lr, b1, b2, self.eps, self.weight_decay,
                       self.step_count, grad_scale)
            ops.copy2d(wa, w, 1, w.numel(), w.numel(), w.numel())
            return
        ops.adamw_(w, master, m, v, g, self.lr, b1, b2, self.eps, self.weight_decay,
                   self.step_count, grad_scale)

    # ---- multi-tensor form ------------------------------------------------------------------
    @property
    def _CHUNK(self) -> int:
        """elements per workgroup slice of the multi-tensor kernel: the library says (mk_adamw_chunk, csrc/softmax.hip)"""
        c = FusedAdamW._chunk_cache
        if c is None:
            from . import lib as _L
            c = FusedAdamW._chunk_cache = int(_L.load().mk_adamw_chunk())
        return c

    _chunk_cache = None

    @torch.no_grad()
    def step_params(self, params, grad_scale: float = 1.0):
        """update every parameter of `params` that has a gradient with ONE mk_adamw_multi launch
        per dtype (same arithmetic as step_param; the pointer table is rebuilt each step because
        autograd allocates fresh gradients)."""
        from . import lib as _L
        ops.bump_weight_version()
        groups = {}
        keep = []
        for p in params:
            if p.grad is None:
                continue
            master, m, v = self._state(p)
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            ptrs = (p.data.data_ptr(), master.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr())
            if not p.data.is_contiguous() or any(x & 15 for x in ptrs):
                self.step_param(p, grad_scale)          # unaligned view: single-tensor kernel
                continue
            keep.append(g)
            groups.setdefault((p.dtype, p.device), []).append(ptrs + (p.numel(),))
        b1, b2 = self.betas
        lib = _L.load()
        for (dtype, dev), items in groups.items():
            n = len(items)
            host = self._host_table(n)
            t = host[: 7 * n + 1]
            flat = []
            starts = [0]
            for it in items:
                flat.extend(it)
                starts.append(starts[-1] + (it[5] + self._CHUNK - 1) // self._CHUNK)
            t[: 6 * n] = torch.tensor(flat, dtype=torch.int64)
            t[6 * n:] = torch.tensor(starts, dtype=torch.int64)
            devt = self._dev_table(dev, 7 * n + 1)
            devt[: 7 * n + 1].copy_(t, non_blocking=True)
            self._table_event.record(torch.cuda.current_stream(dev))
            _L.check(lib.mk_adamw_multi(devt.data_ptr(), devt.data_ptr() + 6 * n * 8, n, starts[-1], self.lr,
                                        b1, b2, self.eps, self.weight_decay, self.step_count, grad_scale,
                                        ops._DT[dtype], torch.cuda.current_stream(dev).cuda_stream),
                     "mk_adamw_multi")
        del keep

    def _host_table(self, n):
        """pinned staging buffer, two of them used alternately; the H2D copy of the one written two
        calls ago has long completed (its event is checked anyway)"""
        need = 7 * n + 1
        st = getattr(self, "_tables", None)
        if st is None or st[0].numel() < need:
            st = self._tables = [torch.empty(max(need, 4096), dtype=torch.int64, pin_memory=True) for _ in range(2)]
            self._table_events = [torch.cuda.Event(), torch.cuda.Event()]
            self._table_flip = 0
        self._table_flip ^= 1
        self._table_event = self._table_events[self._table_flip]
        self._table_event.synchronize()
        return st[self._table_flip]

    def _dev_table(self, dev, need):
        tabs = self.__dict__.setdefault("_dev_tables", {})
        t = tabs.get(dev)
        if t is None or t.numel() < need:
            t = tabs[dev] = torch.empty(max(need, 4096), dtype=torch.int64, device=dev)
        return t

    # ---- flat buckets (bucketed.BucketedStep, one rank) -----------------------------------------
    @torch.no_grad()
    def step_buckets(self, items, grad_scale: float = 1.0, dev_hyper: bool = False):
        """update every flat bucket with ONE launch.  items: [(key, w, g)] with w / g the whole flat
        parameter / gradient buffer of a bucket (fixed addresses), key as in step_shard.  The
        pointer table is STATIC: built and uploaded once per set of buckets, so a step costs one
        launch and no host-to-device traffic (dev_hyper: the per-step scalars come from the device
        buffer of update_hyper(), for graph capture)."""
        from . import lib as _L
        ck = tuple(k for k, _, _ in items)
        cache = self.__dict__.setdefault("_bucket_tables", {})
        ent = cache.get(ck)
        if ent is None:
            rows, starts = [], [0]
            for key, w, g in items:
                master, m, v = self._shard_state(key, w)
                ptrs = (w.data_ptr(), master.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr())
                if any(x & 15 for x in ptrs) or w.dtype != items[0][1].dtype:
                    raise ValueError("FusedAdamW.step_buckets: buckets must be 16-byte aligned and of one dtype")
                rows.extend(ptrs + (w.numel(),))
                starts.append(starts[-1] + (w.numel() + self._CHUNK - 1) // self._CHUNK)
            n = len(items)
            dev = items[0][1].device
            devt = torch.tensor(rows + starts, dtype=torch.int64).to(dev)     # setup time, once
            torch.cuda.current_stream(dev).synchronize()
            ent = cache[ck] = (devt, n, starts[-1], items[0][1].dtype, dev)
        devt, n, chunks, dtype, dev = ent
        b1, b2 = self.betas
        st = torch.cuda.current_stream(dev).cuda_stream
        if dev_hyper:
            _L.check(_L.load().mk_adamw_multi_dev(devt.data_ptr(), devt.data_ptr() + 6 * n * 8, n, chunks, b1, b2,
                                                  self.eps, self.weight_decay, self._hyper_dev.data_ptr(),
                                                  ops._DT[dtype], st), "mk_adamw_multi_dev")
        else:
            _L.check(_L.load().mk_adamw_multi(devt.data_ptr(), devt.data_ptr() + 6 * n * 8, n, chunks, self.lr,
                                              b1, b2, self.eps, self.weight_decay, self.step_count, grad_scale,
                                              ops._DT[dtype], st), "mk_adamw_multi")

    # ---- hipGraph form (train.GraphedStep) ------------------------------------------------------
    _HYPER_SLOTS = 4

    def update_hyper(self, device, grad_scale: float = 1.0):
        """write {lr, 1 - beta1^step, 1 - beta2^step, grad_scale} of the CURRENT step_count to the
        device buffer the captured optimizer launch reads (stream-ordered H2D from pinned memory).
        The copy reads the pinned slot when it EXECUTES, and graph launches are asynchronous (a
        caller that never reads the loss runs many steps ahead of the GPU): each slot carries an
        event recorded behind its copy and is not rewritten before that event has completed."""
        import ctypes as C
        from . import lib as _L
        if getattr(self, "_hyper_dev", None) is None:
            self._hyper_dev = torch.zeros(4, dtype=torch.float32, device=device)
            self._hyper_host = [torch.zeros(4, dtype=torch.float32, pin_memory=True) for _ in range(self._HYPER_SLOTS)]
            self._hyper_events = [None] * self._HYPER_SLOTS
            self._hyper_flip = 0
        bc = (C.c_float * 2)()
        _L.check(_L.load().mk_adamw_bias_correction(self.betas[0], self.betas[1], self.step_count, bc),
                 "mk_adamw_bias_correction")
        self._hyper_flip = (self._hyper_flip + 1) % self._HYPER_SLOTS
        ev = self._hyper_events[self._hyper_flip]
        if ev is not None:
            ev.synchronize()          # the copy that last read this slot has executed
        h = self._hyper_host[self._hyper_flip]
        h[0], h[1], h[2], h[3] = float(self.lr), float(bc[0]), float(bc[1]), float(grad_scale)
        self._hyper_dev.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._hyper_events[self._hyper_flip] = ev

    def prepare_graph(self, params):
        """allocations step_params_dev needs, made BEFORE the capture starts (pinned host memory
        cannot be allocated while a stream is capturing)"""
        n = len(params)
        dev = params[0].device
        self._graph_host = torch.zeros(7 * n + 1, dtype=torch.int64, pin_memory=True)
        self._graph_dev = torch.zeros(7 * n + 1, dtype=torch.int64, device=dev)

    @torch.no_grad()
    def step_params_dev(self, params):
        """step_params for a hipGraph capture: the per-step scalars come from the device buffer of
        update_hyper(), the pointer table is built ONCE (the captured gradients keep their
        addresses) and uploaded by a captured copy from pinned memory that stays alive."""
        from . import lib as _L
        items = []
        for p in params:
            if p.grad is None:
                continue
            master, m, v = self._state(p)
            g = p.grad
            ptrs = (p.data.data_ptr(), master.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr())
            if not p.data.is_contiguous() or not g.is_contiguous() or any(x & 15 for x in ptrs):
                raise ValueError("FusedAdamW.step_params_dev: unaligned / strided parameter or gradient "
                                 "(use the eager step for this model)")
            if p.dtype != params[0].dtype or p.device != params[0].device:
                raise ValueError("FusedAdamW.step_params_dev: one dtype and one device per graph")
            items.append(ptrs + (p.numel(),))
        n = len(items)
        flat, starts = [], [0]
        for it in items:
            flat.extend(it)
            starts.append(starts[-1] + (it[5] + self._CHUNK - 1) // self._CHUNK)
        host, devt = self._graph_host, self._graph_dev
        host[: 7 * n + 1] = torch.tensor(flat + starts, dtype=torch.int64)
        devt.copy_(host, non_blocking=True)
        b1, b2 = self.betas
        dev = params[0].device
        _L.check(_L.load().mk_adamw_multi_dev(devt.data_ptr(), devt.data_ptr() + 6 * n * 8, n, starts[-1], b1, b2,
                                              self.eps, self.weight_decay, self._hyper_dev.data_ptr(),
                                              ops._DT[params[0].dtype], torch.cuda.current_stream(dev).cuda_stream),
                 "mk_adamw_multi_dev")

    # ---- ZeRO-1 style shard (train.OverlappedStep, world > 1) ---------------------------------
    def _shard_state(self, key, w):
        st = self.state.get(key)
        if st is None:
            master = ops.cast(w.contiguous(), torch.float32) if w.dtype != torch.float32 else w.clone()
            m = torch.empty_like(master)
            v = torch.empty_like(master)
            ops.fill_(m, 0.0)
            ops.fill_(v, 0.0)
            st = self.state[key] = _Slot((master, m, v))
            self._restore(key, st)
        return st

    @torch.no_grad()
    def step_shard(self, key, w, grad_shard, grad_scale: float = 1.0):
        """update the flat parameter slice `w` (a view into one or several adjacent parameters)
        from `grad_shard`, the rank-averaged gradient of exactly those elements.  Optimizer
        state (fp32 master / m / v, 12 B per owned element) exists only for the slice and is
        keyed by `key` (stable across steps: the slice a rank owns never changes)."""
        ops.bump_weight_version()
        b1, b2 = self.betas
        master, m, v = self._shard_state(key, w)
        ops.adamw_(w, master, m, v, grad_shard, self.lr, b1, b2, self.eps, self.weight_decay,
                   self.step_count, grad_scale)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        """torch.optim.Optimizer.step: one multi-tensor launch per parameter group (each with its own lr /
        weight_decay / betas / eps).  Under a BucketedStep runtime the update already happened in finish() and
        the runtime's `optimizer_step_is_noop` flag makes this a no-op (hf.MacawTrainerMixin)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if getattr(self, "_runtime_steps", False):
            return loss
        self.step_count += 1
        if len(self.param_groups) == 1:
            self.step_params(self.param_groups[0]["params"], grad_scale)
        else:
            try:
                for g in self.param_groups:           # one group at a time, each with its OWN hyper-parameters
                    self._hp_override = g             # (param_groups itself is never narrowed: the checkpoint
                    self.step_params(g["params"], grad_scale)   # key of a parameter is its global position)
            finally:
                self._hp_override = None
        return loss

"""CPU tests of the round-4 advisor findings: the optimizer's checkpoint keys under several parameter
groups, the checkpoint loaded before the runtime exists, one optimizer file per rank under ZeRO-1 in the HF mixin
(2 gloo ranks), the process-global GEMM planning limit after a window that raised."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _two_groups():
    from macaw_llm_amd.optim import FusedAdamW
    ps = [torch.nn.Parameter(torch.zeros(8, 8)) for _ in range(4)]          # identical shapes, as transformer layers
    opt = FusedAdamW([dict(params=ps[:2], lr=1e-3, weight_decay=0.0), dict(params=ps[2:], lr=5e-4, weight_decay=0.1)])
    return ps, opt


def test_multi_group_step_keys_every_parameter_by_its_global_position_and_uses_each_groups_hyper():
    """step() used to narrow param_groups to one group at a time, and _key_name() numbered the
    parameters through it: group 1's parameters were keyed 'param:0', 'param:1' while they were stepped."""
    ps, opt = _two_groups()
    seen = []
    opt.step_params = lambda params, gs=1.0: seen.append(([opt._key_name(p) for p in params], opt.lr, opt.weight_decay))
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()
    assert seen == [(["param:0", "param:1"], 1e-3, 0.0), (["param:2", "param:3"], 5e-4, 0.1)]
    assert len(opt.param_groups) == 2 and opt._hp_override is None and opt.lr == 1e-3
    # a group added later extends the numbering
    extra = torch.nn.Parameter(torch.zeros(8, 8))
    opt.add_param_group(dict(params=[extra]))
    assert opt._key_name(extra) == "param:4" and opt._key_name(ps[3]) == "param:3"


def test_restore_after_load_goes_to_the_right_slot_in_every_group(monkeypatch):
    """the silent failure the finding describes: with identical shapes a group-relative key pops ANOTHER
    parameter's moments.  Slots are created lazily inside step(); each must receive its own entry."""
    from macaw_llm_amd import ops
    monkeypatch.setattr(ops, "cast", lambda t, dt: t.to(dt))
    monkeypatch.setattr(ops, "fill_", lambda t, v: t.fill_(v))
    ps, opt = _two_groups()
    sd = {"step_count": 3, "lr": 1e-3, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0.0, "layout": None,
          "loss_scaler": None, "param_groups": None,
          "state": {f"param:{i}": {"master": torch.full((8, 8), float(i)), "exp_avg": torch.full((8, 8), 10.0 + i),
                                   "exp_avg_sq": torch.full((8, 8), 20.0 + i)} for i in range(4)}}
    opt.load_state_dict(sd)
    touched = []
    opt.step_params = lambda params, gs=1.0: touched.extend(opt._state(p) for p in params)
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()
    for i, st in enumerate(touched):
        assert float(st[0][0, 0]) == i and float(st[1][0, 0]) == 10.0 + i and float(st[2][0, 0]) == 20.0 + i
    opt.assert_restored()


class _FakeRuntime:
    def __init__(self, layout, scaler):
        self._layout, self.loss_scaler = layout, scaler

    def layout(self):
        return self._layout


def test_checkpoint_loaded_before_the_runtime_exists_is_applied_when_it_attaches():
    """under the HF mixin the runtime is built in the first training_step, AFTER
    Trainer._load_optimizer_and_scheduler: the saved loss-scaler state was dropped and the layout not compared."""
    from macaw_llm_amd.bucketed import DynamicLossScaler
    _, opt = _two_groups()
    saved_scaler = DynamicLossScaler()
    saved_scaler.update(True)
    saved_scaler.update(True)                                       # hysteresis used up: the scale has halved
    layout = {"world": 2, "rank": 1, "zero1": True, "bucket_elems": [128], "dtypes": ["torch.float32"]}
    sd = {"step_count": 0, "lr": 1e-3, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0.0, "layout": layout,
          "loss_scaler": saved_scaler.state_dict(), "param_groups": None, "state": {}}
    opt.load_state_dict(sd)
    assert opt._stashed is not None
    fresh = DynamicLossScaler()
    opt.attach_runtime(rt := _FakeRuntime(dict(layout), fresh))
    assert fresh.state_dict() == saved_scaler.state_dict() and opt._stashed is None
    del rt
    _, other = _two_groups()
    other.load_state_dict(sd)
    with pytest.raises(ValueError, match="layout"):
        other.attach_runtime(_FakeRuntime(dict(layout, rank=0), DynamicLossScaler()))


# ---- one optimizer file per rank (hf.MacawTrainerMixin) ---------------------------------------------------------
class _Args:
    def __init__(self, rank):
        self.should_save = rank == 0
        self.device = torch.device("cpu")


class _Opt:
    """stands in for FusedAdamW: the state of rank r is recognisable"""

    def __init__(self, rank):
        self.rank, self.loaded = rank, None

    def state_dict(self):
        return {"layout": {"world": 2, "rank": self.rank}, "state": {"shard:0:0:4": torch.full((4,), float(self.rank))}}

    def load_state_dict(self, sd):
        self.loaded = sd


def _ckpt_worker(rank, world, port, out_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from macaw_llm_amd.hf import MacawTrainerMixin

        class T(MacawTrainerMixin):
            pass

        lin = torch.nn.Linear(2, 2)
        sched_opt = torch.optim.SGD(lin.parameters(), lr=0.1)
        t = T()
        t.args, t.optimizer = _Args(rank), _Opt(rank)
        t.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(sched_opt, lambda s: 1.0 / (1 + s))
        sched_opt.step()
        t.lr_scheduler.step()
        t._save_optimizer_and_scheduler(out_dir)
        files = sorted(os.listdir(out_dir))
        # resume: fresh objects, every rank must get ITS shards back
        t2 = T()
        t2.args, t2.optimizer = _Args(rank), _Opt(rank)
        t2.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(torch.optim.SGD(lin.parameters(), lr=0.1), lambda s: 1.0 / (1 + s))
        t2._load_optimizer_and_scheduler(out_dir)
        ok = (t2.optimizer.loaded["layout"]["rank"] == rank
              and float(t2.optimizer.loaded["state"]["shard:0:0:4"][0]) == float(rank)
              and t2.lr_scheduler.last_epoch == 1)
        q.put((rank, files, ok))
    finally:
        dist.destroy_process_group()


def test_hf_mixin_writes_and_reads_one_optimizer_file_per_rank_under_zero1(tmp_path):
    """HF saves optimizer.state_dict() on rank 0 only; under ZeRO-1 that is rank 0's shards, and
    every rank would load them on resume (layout mismatch -> raise; before round 4: silent restart)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out = str(tmp_path / "checkpoint-1")
    procs = [ctx.Process(target=_ckpt_worker, args=(r, world, port, out, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for rank, files, ok in res:
        assert ok, res
        assert "optimizer_rank0-of-2.pt" in files and "optimizer_rank1-of-2.pt" in files and "scheduler.pt" in files
    # another world size finds no file of its own and says so
    from macaw_llm_amd.hf import MacawTrainerMixin
    assert MacawTrainerMixin._macaw_opt_file(3, 8) == "optimizer_rank3-of-8.pt"


def test_lr_scheduler_does_not_advance_on_a_skipped_step():
    from macaw_llm_amd.hf import MacawTrainerMixin

    class Base:
        lr_scheduler = None

        def create_scheduler(self, num_training_steps, optimizer=None):
            if self.lr_scheduler is None:          # (as transformers.Trainer.create_scheduler)
                self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: 1.0 / (1 + s))
            return self.lr_scheduler

    class T(MacawTrainerMixin, Base):
        pass

    class RT:
        last_step_skipped = False

    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    t = T()
    t._macaw_rt = RT()
    s = t.create_scheduler(10, optimizer=opt)
    opt.step()
    s.step()
    assert s.last_epoch == 1
    t._macaw_rt.last_step_skipped = True
    s.step()
    assert s.last_epoch == 1                       # DeepSpeed's rule: an overflow step is not a step
    t._macaw_rt.last_step_skipped = False
    s.step()
    assert s.last_epoch == 2
    assert "step" not in s.state_dict() and t.create_scheduler(10, optimizer=opt) is s     # state_dict stays picklable


def test_comm_cus_default_follows_the_channel_cap(monkeypatch):
    from macaw_llm_amd.bucketed import default_comm_cus
    monkeypatch.delenv("MACAW_COMM_CUS", raising=False)
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "12")
    assert default_comm_cus() == 12
    monkeypatch.setenv("MACAW_COMM_CUS", "0")
    assert default_comm_cus() == 0
    monkeypatch.delenv("MACAW_COMM_CUS")
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    assert default_comm_cus() == 16


def test_planning_limit_is_released_when_a_window_raises(monkeypatch):
    """_reserve_cus(True) is set from backward hooks; a finish() that raised left every later GEMM
    of the process planned for fewer CUs."""
    from macaw_llm_amd.bucketed import BucketedStep
    calls = []

    class Rt(BucketedStep):
        def __init__(self):                    # only what finish() / begin() / abort() touch
            self.accumulate_steps, self._micro, self._cus_reserved, self.comm_cus = 1, 0, True, 16
            self.direct_grads = False
            self.params = [torch.nn.Parameter(torch.zeros(1))]

        def _reserve_cus(self, on):
            calls.append(on)
            self._cus_reserved = on

        def _finish_window(self):
            raise RuntimeError("out of memory")

    rt = Rt()
    with pytest.raises(RuntimeError, match="out of memory"):
        rt.finish()
    assert calls == [False] and rt._cus_reserved is False
    rt._cus_reserved = True
    rt.abort()
    assert calls[-1] is False

"""`transformers.Trainer` (the reference's training driver: llm_trainer.py:183-188 `LLMTrainer(Trainer)`,
run_clm_llms.py:541-563 `trainer.train()` / `trainer.save_model()`) running the MEASURED step runtime
(bucketed.BucketedStep + optim.FusedAdamW) through hf.MacawTrainerMixin -- on the micro configuration with the
golden inputs, on the GPU through the HIP kernels."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_case  # noqa: E402
from oracle import configs  # noqa: E402
from test_model_gpu import build_model, to_dev  # noqa: E402


class _Samples(torch.utils.data.Dataset):
    """the golden batch cut into samples, repeated with rolled token windows so that steps see different data"""

    def __init__(self, inputs, n):
        self.rows = []
        B = inputs["input_ids"].shape[0]
        for i in range(n):
            b, roll = i % B, i // B
            row = {}
            for k, v in inputs.items():
                if v is None:
                    continue
                t = v[b].clone()
                if k in ("input_ids", "labels") and roll:
                    t[3:] = torch.roll(t[3:], roll)
                row[k] = t
            self.rows.append(row)

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        return self.rows[i]


def _trainer_cls():
    from transformers import Trainer
    from macaw_llm_amd.hf import MacawTrainerMixin

    class LLMTrainer(MacawTrainerMixin, Trainer):
        macaw_bucket_bytes = 64 << 10                # several buckets on the micro model

        def get_self_inputs(self, batch):            # (the reference loads frames / audio from disk here)
            return {"inputs": dict(batch)}

        def compute_loss(self, model, inputs, return_outputs=False, **kw):      # llm_trainer.py:184-188
            inputs = self.get_self_inputs(inputs)
            # forward pass
            loss = model(**inputs)[0]
            return loss

        def _get_train_sampler(self, *a, **kw):      # fixed order: the manual loop below sees the same batches
            return torch.utils.data.SequentialSampler(self.train_dataset)

    return LLMTrainer


@pytest.mark.parametrize("accum", [1, 2])
def test_hf_trainer_runs_the_bucket_runtime_and_round_trips_through_save_model(dev, tmp_path, accum):
    from transformers import TrainingArguments, default_data_collator
    from macaw_llm_amd import modeling as M
    from macaw_llm_amd.bucketed import BucketedStep, cosine_with_warmup
    from macaw_llm_amd.optim import FusedAdamW
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    steps, bs, lr = 3, 2, 1e-3
    data = _Samples(fx["inputs"], steps * bs * accum)

    # ---- the reference's driver: LLMTrainer(...).train(); save_model()
    torch.manual_seed(0)              # (parameters outside the golden state dict keep their random init: same seed below)
    model = build_model(cfg, fx["state"], torch.float32, dev, fuse=True)
    model.llm.gradient_checkpointing_enable()                     # train.sh:38 `--gradient_checkpointing` surface
    args = TrainingArguments(output_dir=str(tmp_path / "out"), per_device_train_batch_size=bs,
                             gradient_accumulation_steps=accum, max_steps=steps, learning_rate=lr,
                             weight_decay=0.0, warmup_steps=2, lr_scheduler_type="cosine", max_grad_norm=1.0,
                             logging_steps=1, save_strategy="no", report_to=[], remove_unused_columns=False,
                             dataloader_pin_memory=False, seed=1)
    tr = _trainer_cls()(model=model, args=args, train_dataset=data, data_collator=default_data_collator)
    before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    out = tr.train()
    assert tr.state.global_step == steps and math.isfinite(out.training_loss)
    from macaw_llm_amd.hf import unwrap_optimizer
    opt = unwrap_optimizer(tr.optimizer)
    assert isinstance(opt, FusedAdamW) and isinstance(opt, torch.optim.Optimizer)
    assert opt.step_count == steps                                # optimizer.step() of HF's loop did not step again
    rt = tr.macaw_runtime()
    assert len(rt.buckets) > 1 and rt.max_grad_norm == 1.0 and rt.accumulate_steps == accum
    assert all(not torch.equal(p.detach(), before[n]) for n, p in model.named_parameters()
               if p.requires_grad and n.endswith("q_proj.weight"))
    logged = [h for h in tr.state.log_history if "grad_norm" in h]
    assert logged and all(math.isfinite(h["grad_norm"]) and h["grad_norm"] > 0 for h in logged)

    # ---- the same three steps by hand on the same runtime pieces: bit-identical weights
    torch.manual_seed(0)
    ref = build_model(cfg, fx["state"], torch.float32, dev, fuse=True)
    ref.llm.gradient_checkpointing_enable()
    ref.train()
    ropt = FusedAdamW([p for p in ref.parameters() if p.requires_grad], lr=lr, weight_decay=0.0)
    rrt = BucketedStep(None, ropt, model=ref, bucket_bytes=64 << 10, accumulate_steps=accum, max_grad_norm=1.0)
    it = iter(torch.utils.data.DataLoader(data, batch_size=bs, collate_fn=default_data_collator))
    for s in range(steps):
        rrt.set_lr(cosine_with_warmup(s, steps, warmup_ratio=0.34, base_lr=lr))
        for _ in range(accum):
            batch = to_dev(next(it), dev)
            rrt.begin()
            ref(inputs=batch)[0].backward()
            rrt.finish()
    torch.cuda.synchronize()
    got = dict(model.named_parameters())
    for n, p in ref.named_parameters():
        if p.requires_grad:
            assert torch.equal(p.detach(), got[n].detach()), n

    # ---- trainer.save_model() -> from_pretrained -> identical logits (run_clm_llms.py:563, inference script)
    tr.save_model(str(tmp_path / "saved"))
    M.AUTO_FUSE = True
    from macaw_llm_amd.factory import make_config
    loaded = M.MM_LLMs.from_pretrained(str(tmp_path / "saved"), config=make_config(cfg)).to(dev).eval()   # run_clm_llms_inference.py:455
    model.eval()
    inp = to_dev(fx["inputs"], dev)
    with torch.no_grad():
        a, b = model(inputs=inp).logits, loaded(inputs=inp).logits
    assert torch.equal(a, b)
    rt.remove()
    rrt.remove()


def test_trainer_optimizer_checkpoint_carries_the_layout_and_refuses_another(dev):
    """ZeRO-1 shard keys depend on world size / rank / bucket size; loading a checkpoint written with
    another layout used to match nothing and restart the moments at zero under a restored step counter."""
    from macaw_llm_amd.bucketed import BucketedStep
    from macaw_llm_amd.optim import FusedAdamW
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    inp = to_dev(fx["inputs"], dev)

    def fresh(bucket_bytes):
        model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
        opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
        rt = BucketedStep(None, opt, model=model, bucket_bytes=bucket_bytes)
        opt.attach_runtime(rt)
        return model, opt, rt

    def step(model, rt):
        rt.begin()
        model(inputs=inp).loss.backward()
        rt.finish()

    A, oa, ra = fresh(64 << 10)
    step(A, ra)
    sd = oa.state_dict()
    assert sd["layout"]["world"] == 1 and sd["layout"]["bucket_elems"] == [b.n for b in ra.buckets]
    ra.remove()
    B, ob, rb = fresh(32 << 10)                                   # other bucket size: other shard keys
    with pytest.raises(ValueError, match="layout"):
        ob.load_state_dict(sd)
    rb.remove()
    # a checkpoint WITHOUT layout information (rounds <= 3) whose keys match nothing: caught at the first step
    C, oc, rc = fresh(32 << 10)
    legacy = dict(sd, layout=None)
    oc.load_state_dict(legacy)
    with pytest.raises((KeyError, RuntimeError)):
        step(C, rc)
    rc.remove()
    D, od, rd = fresh(64 << 10)                                   # same layout: loads and steps
    od.load_state_dict(sd)
    step(D, rd)
    assert od.step_count == 2 and not od._pending
    rd.remove()


def test_mixin_runtime_reserves_cus_for_the_collectives_like_bench_py(dev, monkeypatch, tmp_path):
    """`hf.py` built BucketedStep without `comm_cus`, so the reference's own driver route would
    have run N > 1 with every GEMM planned for all 256 CUs beside the resident RCCL channels."""
    from transformers import TrainingArguments, default_data_collator
    from macaw_llm_amd.bucketed import default_comm_cus
    monkeypatch.delenv("MACAW_COMM_CUS", raising=False)
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.float32, dev, fuse=True)
    args = TrainingArguments(output_dir=str(tmp_path / "o"), per_device_train_batch_size=2, max_steps=1,
                             save_strategy="no", report_to=[], remove_unused_columns=False, dataloader_pin_memory=False)
    tr = _trainer_cls()(model=model, args=args, train_dataset=_Samples(fx["inputs"], 2), data_collator=default_data_collator)
    tr.create_optimizer()
    rt = tr.macaw_runtime()
    # one rank: no collective, nothing to reserve -- the value is carried and takes effect as soon as world > 1
    assert rt.comm_cus == 0 and not rt.collective
    assert default_comm_cus() == int(__import__("os").environ["NCCL_MAX_NCHANNELS"]) > 0
    rt.remove()
    cls = _trainer_cls()
    cls.macaw_comm_cus = 24
    import macaw_llm_amd.hf as H
    seen = {}
    real = H.BucketedStep

    def spy(*a, **kw):
        seen.update(kw)
        return real(*a, **kw)

    monkeypatch.setattr(H, "BucketedStep", spy)
    tr2 = cls(model=model, args=args, train_dataset=_Samples(fx["inputs"], 2), data_collator=default_data_collator)
    tr2.create_optimizer()
    tr2.macaw_runtime().remove()
    assert seen["comm_cus"] == 24


def test_two_group_optimizer_resumes_every_slot_from_its_own_entry(dev):
    """ADVICE r4 (medium), on the real kernels: two parameter groups of identically shaped tensors, save after two
    steps, load into a fresh optimizer, step both: bit-identical to the optimizer that never stopped."""
    from macaw_llm_amd.optim import FusedAdamW
    torch.manual_seed(3)

    def make():
        g = torch.Generator().manual_seed(5)
        ps = [torch.nn.Parameter(torch.randn(64, 64, generator=g).to(dev, torch.bfloat16)) for _ in range(4)]
        opt = FusedAdamW([dict(params=ps[:2], lr=1e-2), dict(params=ps[2:], lr=3e-3, weight_decay=0.1)])
        return ps, opt

    def grads(ps, k):
        g = torch.Generator().manual_seed(100 + k)
        for i, p in enumerate(ps):
            p.grad = (torch.randn(64, 64, generator=g) * (i + 1)).to(dev, torch.bfloat16)

    A, oa = make()
    for k in range(2):
        grads(A, k)
        oa.step()
    sd = {k: (v if k != "state" else {n: {m: t.clone() for m, t in e.items()} for n, e in v.items()})
          for k, v in oa.state_dict().items()}
    assert sorted(sd["state"]) == ["param:0", "param:1", "param:2", "param:3"]
    B, ob = make()
    with torch.no_grad():
        for a, b in zip(A, B):
            b.copy_(a)
    ob.load_state_dict(sd)
    for k in range(2, 4):
        grads(A, k)
        oa.step()
        grads(B, k)
        ob.step()
    torch.cuda.synchronize()
    ob.assert_restored()
    for i, (a, b) in enumerate(zip(A, B)):
        assert torch.equal(a, b), i
        for x, y in zip(oa.state[a], ob.state[b]):
            assert torch.equal(x, y), i

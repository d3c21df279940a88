"""End-to-end GPU parity of macaw_llm_amd.modeling.MM_LLMs (hand-written HIP path, through the
C ABI) against golden vectors produced by the reference itself (tests/golden/, see
oracle/make_golden.py).  Tolerances:
  fp32 engine : logits within 1e-3 abs of the reference (north_star's bound; measured ~1e-5),
                integer outputs (attention_mask, labels, prefix layout) bit-exact.
  bf16 engine : bf16 storage of every activation => ~2^-8 relative per tensor; on this
                2-layer micro model logits (|x| <= ~1) must be within 3e-2 abs / loss within 2e-2.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_case  # noqa: E402
from oracle import configs  # noqa: E402


@pytest.fixture(autouse=True)
def _restore_auto_fuse():
    from macaw_llm_amd import modeling as M
    yield
    M.AUTO_FUSE = True


def build_model(cfg, state, dtype, dev, freeze_encoders=True, fuse=False):
    """fuse=False pins the reference FORMULATION of the engine (three q/k/v GEMMs, two gate/up
    GEMMs, unfused encoder projections): lazy fusion is switched off for the test."""
    from transformers import CLIPConfig, LlamaConfig, WhisperConfig
    from macaw_llm_amd import modeling as M
    M.AUTO_FUSE = bool(fuse)
    mm = M.MM_LLMs_Config(clip_config=CLIPConfig(**cfg["clip"]), whisper_config=WhisperConfig(**cfg["whisper"]),
                          llm_config=LlamaConfig(**cfg["llama"]), **cfg["mm"])
    model = M.MM_LLMs(mm)
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    model = model.to(dev).to(dtype)
    if fuse:  # q|k|v and gate|up weights re-homed in contiguous storage -> single GEMMs
        for layer in model.llm.model.layers:
            layer.fuse_projections()
    if freeze_encoders:  # run_clm_llms.py:390-393
        for n, p in model.named_parameters():
            p.requires_grad_("encoder" not in n)
    return model


def to_dev(inputs, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inputs.items()}


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("case", ["micro_all", "micro_image"])
def test_forward_backward_fp32_matches_reference(dev, case, fuse):
    fx = load_case(case)
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.float32, dev, fuse=fuse).eval()
    if fuse:
        l0 = model.llm.model.layers[0]
        assert l0._fused_view((l0.mlp.gate_proj.weight, l0.mlp.up_proj.weight)) is not None
    inp = to_dev(fx["inputs"], dev)
    emb, am, lab = model.prepare_inputs_for_generation(inp)
    assert torch.equal(am.cpu(), fx["attention_mask"])          # INT: bit exact
    assert torch.equal(lab.cpu(), fx["labels"])                 # INT: bit exact
    assert (emb.float().cpu() - fx["inputs_embeds"]).abs().max().item() < 1e-4
    out = model(inputs=inp)
    err = (out.logits.float().cpu() - fx["logits"]).abs().max().item()
    assert err < 1e-3, f"logits max abs err {err}"
    assert abs(out.loss.item() - fx["loss"].item()) < 1e-4
    out.loss.backward()
    named = dict(model.named_parameters())
    for name, g in fx["grads"].items():
        got = named[name].grad
        assert got is not None, name
        e = (got.float().cpu() - g).abs().max().item()
        assert e <= 2e-4 * max(1.0, g.abs().max().item()), (name, e)
    for name, n in fx["grad_norms"].items():
        got = named[name].grad
        assert got is not None, name
        assert abs(got.float().norm().item() - n) <= 2e-3 * max(n, 1e-3), (name, got.float().norm().item(), n)
    # parameters the reference leaves without gradient also stay without gradient
    for name in fx["no_grad_params"]:
        assert named[name].grad is None, name


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("case", ["micro_all", "micro_image"])
def test_forward_backward_bf16(dev, case, fuse):
    fx = load_case(case)
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=fuse).eval()
    inp = to_dev(fx["inputs"], dev)
    inp = {k: (v.half() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}  # llm_trainer.py:366-368
    out = model(inputs=inp)
    err = (out.logits.float().cpu() - fx["logits"]).abs().max().item()
    assert err < 3e-2, f"bf16 logits max abs err {err}"
    assert abs(out.loss.item() - fx["loss"].item()) < 2e-2
    out.loss.backward()
    named = dict(model.named_parameters())
    for name, n in fx["grad_norms"].items():
        got = named[name].grad
        assert got is not None and torch.isfinite(got).all(), name
        assert abs(got.float().norm().item() - n) <= 0.08 * max(n, 1e-3), (name, got.float().norm().item(), n)


def test_train_mode_dropout_and_text_only(dev):
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.float32, dev).train()
    inp = to_dev(fx["inputs"], dev)
    l1 = model(inputs=inp).loss
    l1.backward()
    assert torch.isfinite(l1)
    assert abs(l1.item() - fx["loss"].item()) < 0.5      # dropout perturbs, does not break
    # text-only (the reference itself cannot run this case with labels; see test_oracle.py)
    t = dict(inp, images=None, audios=None, videos=None)
    out = model.eval()(inputs=t)
    assert out.logits.shape[1] == inp["input_ids"].shape[1] and torch.isfinite(out.loss)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fuse", [False, True])
def test_gradient_checkpointing_is_bit_identical(dev, dtype, fuse):
    """modeling.py:474-489 (checkpoint every decoder layer while training): our recompute path
    re-runs the deterministic layer forward inside its backward, so loss, logits and every
    gradient must equal the non-checkpointed run bit for bit."""
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], dtype, dev, fuse=fuse)
    llm = model.llm.train()
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, cfg["llama"]["vocab_size"], (3, 24), generator=g).to(dev)
    am = torch.ones_like(ids)
    am[1, 19:] = 0
    res = []
    for flag in (False, True):
        llm.model.gradient_checkpointing = flag
        llm.zero_grad(set_to_none=True)
        out = llm(input_ids=ids, attention_mask=am, labels=ids)
        out.loss.backward()
        res.append((out.loss.detach().clone(), out.logits.detach().clone(),
                    {n: p.grad.clone() for n, p in llm.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][2].keys() == res[1][2].keys() and len(res[0][2]) > 10
    for n in res[0][2]:
        assert torch.equal(res[0][2][n], res[1][2][n]), n
    # eval mode ignores the flag (reference: `self.gradient_checkpointing and self.training`)
    llm.eval()
    assert torch.equal(llm(input_ids=ids, attention_mask=am).logits, res[0][1])


@pytest.mark.parametrize("use_cache", [True, False])
def test_generate_matches_restated_greedy(dev, use_cache):
    """greedy token ids bit-exact vs the restated HF greedy loop (golden), with the KV-cache
    decode path and with full-prefix recompute"""
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.float32, dev).eval()
    emb = fx["inputs_embeds"].to(dev)
    ids = model.llm.generate(inputs_embeds=emb, max_new_tokens=8, eos_token_id=2, bos_token_id=1,
                             pad_token_id=cfg["tags"]["pad"], use_cache=use_cache)
    assert torch.equal(ids.cpu(), fx["generate_ids"])   # token ids: bit exact


def test_generate_kv_cache_bf16_and_inference_flag(dev):
    """bf16 (fused attention with Lq=1 against the cache): the cached decode must reproduce the
    ids of the full-recompute decode, and MM_LLMs.forward(inference=True) returns ids."""
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
    emb = fx["inputs_embeds"].to(dev).to(torch.bfloat16)
    a = model.llm.generate(inputs_embeds=emb, max_new_tokens=12, eos_token_id=2, pad_token_id=106)
    b = model.llm.generate(inputs_embeds=emb, max_new_tokens=12, eos_token_id=2, pad_token_id=106,
                           use_cache=False)
    assert a.shape[0] == emb.shape[0] and a.dtype == torch.long
    agree = (a[:, : b.shape[1]] == b[:, : a.shape[1]]).float().mean().item()
    assert agree >= 0.9, agree          # bf16 near-ties may flip an argmax; ids must otherwise agree
    inp = to_dev(fx["inputs"], dev)
    inp["inference"] = True
    gen = model(inputs=inp)
    assert gen.dim() == 2 and gen.shape[0] == emb.shape[0] and gen.shape[1] <= 128


def test_generate_hipgraph_decode_matches_eager_loop(dev):
    """the decode loop replayed from ONE captured graph per token (device-side position, token,
    finished flags) must emit the ids of the eager per-kernel loop, including the early stop when
    every sample has produced eos and pad ids for samples that finished earlier"""
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
    emb = fx["inputs_embeds"].to(dev).to(torch.bfloat16)
    for eos in (2, -1):
        g = model.llm.generate(inputs_embeds=emb, max_new_tokens=24, eos_token_id=eos, pad_token_id=106)
        e = model.llm.generate(inputs_embeds=emb, max_new_tokens=24, eos_token_id=eos, pad_token_id=106,
                               decode_graph=False)
        assert g.dtype == torch.long and g.shape == e.shape, (g.shape, e.shape)
        agree = (g == e).float().mean().item()
        assert agree >= 0.9, agree      # different attention kernel: a bf16 near-tie may flip an argmax
    # force an early stop: use the most frequent greedy token as eos
    e = model.llm.generate(inputs_embeds=emb, max_new_tokens=24, eos_token_id=-1, pad_token_id=106, decode_graph=False)
    eos = int(e[:, 2:].flatten().mode().values)
    g = model.llm.generate(inputs_embeds=emb, max_new_tokens=24, eos_token_id=eos, pad_token_id=106)
    e = model.llm.generate(inputs_embeds=emb, max_new_tokens=24, eos_token_id=eos, pad_token_id=106,
                           decode_graph=False)
    assert g.shape == e.shape and (g == e).float().mean().item() >= 0.9
    # unfused q / k / v and gate / up storage: the graph path takes the separate projections
    mu = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=False).eval()
    g = mu.llm.generate(inputs_embeds=emb, max_new_tokens=16, eos_token_id=-1, pad_token_id=106)
    e = mu.llm.generate(inputs_embeds=emb, max_new_tokens=16, eos_token_id=-1, pad_token_id=106, decode_graph=False)
    assert g.shape == e.shape and (g == e).float().mean().item() >= 0.9
    # max_new_tokens 1 / 2 / 3 (eager token 0, eager token 1, first replay)
    for n in (1, 2, 3):
        g = model.llm.generate(inputs_embeds=emb, max_new_tokens=n, eos_token_id=-1, pad_token_id=106)
        e = model.llm.generate(inputs_embeds=emb, max_new_tokens=n, eos_token_id=-1, pad_token_id=106, decode_graph=False)
        assert g.shape == e.shape == (emb.shape[0], n) and (g == e).float().mean().item() >= 0.9
    # and in fp32 parameters (no fused decode attention: the graph path must step aside)
    m32 = build_model(cfg, fx["state"], torch.float32, dev).eval()
    ids = m32.llm.generate(inputs_embeds=fx["inputs_embeds"].to(dev), max_new_tokens=8, eos_token_id=2,
                           bos_token_id=1, pad_token_id=cfg["tags"]["pad"])
    assert torch.equal(ids.cpu(), fx["generate_ids"])


def test_reference_construction_path_runs_the_fused_kernels(dev):
    """run_clm_llms.py:478-497 builds the model as MM_LLMs(config) -> resize_token_embeddings ->
    freeze -> Trainer .to(device / dtype).  That path must hit the SAME launches as
    factory.build_model(fuse=True): q|k|v / gate|up (and the towers' q|k|v) are fused lazily at the
    first forward and again after a later .to(); a reloaded state dict lands in the fused storage."""
    from transformers import CLIPConfig, LlamaConfig, WhisperConfig
    from macaw_llm_amd import modeling as M, ops
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    mm = M.MM_LLMs_Config(clip_config=CLIPConfig(**cfg["clip"]), whisper_config=WhisperConfig(**cfg["whisper"]),
                          llm_config=LlamaConfig(**cfg["llama"]), **cfg["mm"])
    inp = to_dev(fx["inputs"], dev)

    def launches(model):
        ops.prof_begin()
        out = model(inputs=inp)
        out.loss.backward()
        _, _, n = ops.prof_end()
        model.zero_grad(set_to_none=True)
        return n, out

    model = M.MM_LLMs(mm)                                    # CPU, fp32, as the driver does
    model.load_state_dict(fx["state"], strict=False)
    model.llm.resize_token_embeddings(cfg["llama"]["vocab_size"])
    for n, p in model.named_parameters():                    # run_clm_llms.py:390-393
        p.requires_grad_("encoder" not in n)
    model = model.to(dev).to(torch.bfloat16).eval()
    l0 = model.llm.model.layers[0]
    a0 = model.image_encoder.vision_model.encoder.layers[0].self_attn
    assert l0._fused_view((l0.self_attn.q_proj.weight, l0.self_attn.k_proj.weight, l0.self_attn.v_proj.weight)) is None
    n_auto, out_auto = launches(model)
    assert all(v is not None for v in l0.fused_weights())
    assert all(v is not None for v in M.fused_encoder_qkv(a0))
    ref = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
    n_fused, out_fused = launches(ref)
    assert n_auto == n_fused
    assert torch.equal(out_auto.logits, out_fused.logits)    # same kernels, same bits
    M.AUTO_FUSE = False
    n_unfused, _ = launches(build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=False).eval())
    M.AUTO_FUSE = True
    assert n_unfused > n_fused
    # a later .to() gives every parameter its own storage again: fused again on the next forward
    model = model.to(torch.float32).to(torch.bfloat16)
    assert l0._fused_view((l0.mlp.gate_proj.weight, l0.mlp.up_proj.weight)) is None
    n_again, out_again = launches(model)
    assert n_again == n_fused and torch.equal(out_again.logits, out_fused.logits)
    # reloading a checkpoint writes INTO the fused storage (no stale copies anywhere)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    key = "image_encoder.vision_model.encoder.layers.0.self_attn.k_proj.weight"
    sd[key] = sd[key] * 0.5
    model.load_state_dict(sd)
    W3, _ = M.fused_encoder_qkv(a0)
    E = a0.q_proj.weight.shape[0]
    assert torch.equal(W3[E:2 * E], sd[key].to(dev))
    _, out_mod = launches(model)
    assert not torch.equal(out_mod.logits, out_fused.logits)


@pytest.mark.parametrize("case", ["micro_all", "micro_image"])
def test_fp16_parameters_match_the_oracle(dev, case):
    """The reference's scripts run in fp16 (train.sh:36 `--fp16 True`; llm_trainer.py:411-412
    `.to(torch.float16)` for inference): a model built the reference's way and cast with
    `.to(torch.float16)` runs on the f16 instantiation of every kernel (GEMMs, fused attention, norms,
    RoPE, softmax, loss; csrc/common.h E16<>).  fp16 keeps 10 mantissa bits (bf16: 7), so the bound
    against the fp32 oracle is TIGHTER than the bf16 one: logits 8e-3 abs (bf16 test: 3e-2), loss
    5e-3; gradients finite and within 2 % in norm; integer outputs bit-exact."""
    fx = load_case(case)
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.float16, dev, fuse=True).eval()
    assert model.llm.lm_head.weight.dtype == torch.float16
    inp = to_dev(fx["inputs"], dev)
    emb, am, lab = model.prepare_inputs_for_generation(inp)
    assert emb.dtype == torch.float16
    assert torch.equal(am.cpu(), fx["attention_mask"]) and torch.equal(lab.cpu(), fx["labels"])
    out = model(inputs=inp)
    assert out.logits.dtype == torch.float16
    err = (out.logits.float().cpu() - fx["logits"]).abs().max().item()
    assert err <= 8e-3, err
    assert abs(out.loss.item() - fx["loss"].item()) <= 5e-3
    out.loss.backward()
    for name, n in fx["grad_norms"].items():
        got = dict(model.named_parameters())[name].grad
        assert got is not None and torch.isfinite(got).all(), name
        assert abs(got.float().norm().item() - n) <= 0.02 * max(n, 1e-3), (name, got.float().norm().item(), n)


def test_fp16_generate_ids_match_the_restated_greedy_loop(dev):
    """run_clm_llms_inference.py's dtype: greedy decode with fp16 parameters through the KV-cache /
    hipGraph decode kernels; the golden ids come from the restated HF greedy loop in fp32 -- the
    micro model's logit gaps are far above fp16 resolution, so the ids must be identical."""
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.float16, dev).eval()
    emb = fx["inputs_embeds"].to(dev).to(torch.float16)
    for kw in (dict(use_cache=True), dict(use_cache=True, decode_graph=False), dict(use_cache=False)):
        ids = model.llm.generate(inputs_embeds=emb, max_new_tokens=8, eos_token_id=2, bos_token_id=1,
                                 pad_token_id=cfg["tags"]["pad"], **kw)
        assert torch.equal(ids.cpu(), fx["generate_ids"]), kw


def test_fp16_training_step_with_static_loss_scale(dev):
    """fp16 gradients need loss scaling (configs/deepspeed_config.json: fp16 with a loss scaler): scale
    the loss by 2^k before backward and set BucketedStep.grad_scale = 2^-k -- FusedAdamW unscales in
    fp32 inside the update.  Yardstick = the same three steps with fp32 parameters.  Adam normalises
    every gradient element to ~lr per step, so an element whose tiny gradient rounds differently may
    move the other way: the elementwise bound is 2 * lr * steps; the MEAN deviation is what shows the
    scaling at work (gradients that underflow in unscaled fp16 get no update at all)."""
    from macaw_llm_amd.optim import FusedAdamW
    from macaw_llm_amd.bucketed import BucketedStep
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    inp = to_dev(fx["inputs"], dev)
    lr, steps = 1e-3, 3

    def run(dtype, scale):
        model = build_model(cfg, fx["state"], dtype, dev, fuse=True).eval()
        params = [p for p in model.parameters() if p.requires_grad]
        opt = FusedAdamW(params, lr=lr, weight_decay=0.0)
        rt = BucketedStep(params, opt, bucket_bytes=64 << 10)
        losses = []
        for _ in range(steps):
            rt.begin()
            rt.grad_scale = 1.0 / scale
            loss = model(inputs=inp).loss
            (loss * scale).backward()
            rt.finish()
            losses.append(loss.item())
        torch.cuda.synchronize()
        rt.remove()
        return losses, {n: p.detach().float().clone() for n, p in model.named_parameters()
                        if p.requires_grad and n in fx["state"]}

    l32, p32 = run(torch.float32, 1.0)
    l16, p16 = run(torch.float16, 1024.0)
    assert l32[-1] < l32[0] and l16[-1] < l16[0]
    assert all(abs(a - b) <= 1e-2 * abs(a) + 2e-3 for a, b in zip(l32, l16)), (l32, l16)
    mean_dev = []
    for n in p32:
        d = (p32[n] - p16[n]).abs()
        assert d.max().item() <= 2 * lr * steps + 2e-3 * p32[n].abs().max().item() + 1e-4, (n, d.max().item())
        mean_dev.append(d.mean().item())
    assert sum(mean_dev) / len(mean_dev) <= 0.5 * lr * steps, sum(mean_dev) / len(mean_dev)


def test_fp16_training_with_the_dynamic_loss_scale_of_the_reference_recipe(dev):
    """configs/deepspeed_config.json:14-21 (fp16: loss_scale 0 = dynamic): BucketedStep(loss_scaler=
    DynamicLossScaler(...)).  Started far too high (2^24: the scaled loss is not representable in fp16), the
    first steps overflow -- no update, Adam's step counter does not advance, the scale halves (hysteresis 1) --
    until the gradients are finite; from then on every step updates and the loss falls as in the static-scale
    test above.  The overflow verdict is the global gradient norm (mk_sumsq over the buckets)."""
    from macaw_llm_amd.optim import FusedAdamW
    from macaw_llm_amd.bucketed import BucketedStep, DynamicLossScaler
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    inp = to_dev(fx["inputs"], dev)
    model = build_model(cfg, fx["state"], torch.float16, dev, fuse=True).eval()
    params = [p for p in model.parameters() if p.requires_grad]
    before = [p.detach().clone() for p in params]
    opt = FusedAdamW(params, lr=1e-3, weight_decay=0.0)
    sc = DynamicLossScaler(init_scale=2.0 ** 24, window=1000, hysteresis=1)
    rt = BucketedStep(params, opt, bucket_bytes=64 << 10, loss_scaler=sc)
    losses, skipped = [], []
    for it in range(20):
        rt.begin()
        loss = model(inputs=inp).loss
        rt.scale_loss(loss).backward()
        rt.finish()
        losses.append(loss.item())
        skipped.append(rt.last_step_skipped)
        if it == 0:      # the very first attempt must have been rejected without touching anything
            assert rt.last_step_skipped and opt.step_count == 0
            assert all(torch.equal(a, b.detach()) for a, b in zip(before, params))
    torch.cuda.synchronize()
    rt.remove()
    n_skip = sum(skipped)
    assert 1 <= n_skip <= 16 and not any(skipped[n_skip:]), skipped        # a run of overflows, then clean steps only
    assert sc.scale == 2.0 ** (24 - n_skip) and sc.skipped == n_skip
    assert opt.step_count == 20 - n_skip >= 4
    assert all(math.isfinite(x) for x in losses) and losses[-1] < losses[n_skip] - 0.05, losses
    assert all(torch.isfinite(p).all() for p in params)



@pytest.mark.parametrize("fuse", [False, True])
def test_audio_tower_on_its_own_stream_changes_no_bit(dev, fuse):
    """engine.ENC_SIDE (a switch, off by default: measured +0.4 % / +-0): from the SECOND forward of a model on, the frozen audio tower runs on a second stream
    beside the image / video towers (the first forward stays on one stream: it re-homes the towers' q / k / v parameters).
    Same kernels on the same operands: logits, loss and every gradient bit-identical to the one-stream forward, in eval and
    in train mode (dropout seeds advance per step: compared step by step)."""
    from macaw_llm_amd import engine
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    inp = to_dev(fx["inputs"], dev)
    res = {}
    old = engine.ENC_SIDE["on"]
    try:
        for on in (False, True):
            engine.ENC_SIDE["on"] = on
            engine.ENC_SIDE["launches"] = 0
            model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=fuse).eval()
            outs = []
            for step in range(3):
                model.zero_grad(set_to_none=True)
                out = model(inputs=inp)
                out.loss.backward()
                grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
                outs.append((out.logits.detach().clone(), out.loss.detach().clone(), grads))
            torch.cuda.synchronize()
            res[on] = outs
            if on:
                assert engine.ENC_SIDE["launches"] == 2, "the audio tower left the compute stream in forwards 2 and 3 only"
                assert getattr(model.audio_encoder, "_macaw_side_warm", False)
            else:
                assert engine.ENC_SIDE["launches"] == 0
    finally:
        engine.ENC_SIDE["on"] = old
    for (l0, s0, g0), (l1, s1, g1) in zip(res[False], res[True]):
        assert torch.equal(l0, l1) and torch.equal(s0, s1)
        assert g0.keys() == g1.keys() and all(torch.equal(g0[n], g1[n]) for n in g0)

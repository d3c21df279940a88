"""TEST INFRASTRUCTURE ONLY. Generates the committed golden vectors under tests/golden/ by running
the REFERENCE ITSELF (/root/reference/modeling.py, imported read-only through oracle/ref_loader.py)
on seeded synthetic inputs, CPU fp32, eval mode (dropout off, grads on — SURVEY §7 parity protocol).

    python -m oracle.make_golden          # rewrites tests/golden/*.pt

Contents of each fixture: the hot-path weights, the inputs, and the reference's outputs
(inputs_embeds, extended attention_mask / labels, logits, loss, gradients of the trainable hot-path
parameters: full tensors for a representative subset, L2 norms for all).  Only this script needs
/root/reference; the tests that consume the fixtures do not.
"""
from __future__ import annotations

import os
import sys

import torch

from . import configs, inputs as oin, ref_loader, restate

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

FULL_GRAD_KEYS = (
    "llm.model.embed_tokens.weight", "llm.lm_head.weight", "llm.model.norm.weight",
    "llm.model.layers.0.self_attn.q_proj.weight", "llm.model.layers.0.self_attn.v_proj.weight",
    "llm.model.layers.0.mlp.down_proj.weight", "llm.model.layers.1.input_layernorm.weight",
    "image_align_attention.in_proj_weight", "image_align_attention.bias_k",
    "audio_align_attention.out_proj.bias", "video_align_attention.in_proj_bias",
    "project_image.weight", "project_audio.bias", "transform_video_to_hidden.weight",
    "video_long_self_attention.in_proj_weight", "video_long_self_attention.bias_v",
)


def reference_greedy(llm, inputs_embeds, max_new_tokens, eos, pad):
    """greedy decode through the reference's own cached forward; returns new token ids [B, <= max_new_tokens]"""
    out = llm(inputs_embeds=inputs_embeds, use_cache=True)
    B = inputs_embeds.shape[0]
    done = torch.zeros(B, dtype=torch.bool)
    ids = []
    for _ in range(max_new_tokens):
        nxt = out.logits[:, -1, :].argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
        ids.append(nxt)
        done = done | (nxt == eos)
        if bool(done.all()):
            break
        step = llm.prepare_inputs_for_generation(nxt.unsqueeze(1), past_key_values=out.past_key_values, use_cache=True)
        out = llm(**step)
    return torch.stack(ids, dim=1)


def make(name, cfg_name, batch, text_len, modalities, pad_tail, seed=1234, with_generate=False):
    cfg = configs.get(cfg_name)
    model = ref_loader.build_reference_model(cfg, seed=seed)
    # the reference's training driver freezes every '*encoder*' parameter (run_clm_llms.py:390-393)
    for n, p in model.named_parameters():
        p.requires_grad_("encoder" not in n)
    inp = oin.make_inputs(cfg, batch, text_len, modalities=modalities, seed=1, pad_tail=pad_tail)
    emb, am, lab = model.prepare_inputs_for_generation(inp)
    out = model(inputs=inp)
    out.loss.backward()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    hot = restate.hot_path_state(sd)
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    state_path = os.path.join(GOLDEN_DIR, f"{cfg_name}_state_seed{seed}.pt")
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    if not os.path.exists(state_path):
        torch.save(hot, state_path)   # weights shared by every case of this (config, seed)
    fx = dict(
        config_name=cfg_name, seed=seed, state_file=os.path.basename(state_path), inputs=inp,
        inputs_embeds=emb.detach(), attention_mask=am, labels=lab,
        logits=out.logits.detach(), loss=out.loss.detach(),
        grad_norms={n: g.norm().item() for n, g in grads.items()},
        grads={n: grads[n].clone() for n in FULL_GRAD_KEYS if n in grads},
        no_grad_params=sorted(n for n, p in model.named_parameters() if p.requires_grad and p.grad is None),
    )
    if with_generate:
        # HF 5.x has no GenerationMixin on the reference's LlamaForCausalLM, so the greedy LOOP
        # (argmax, pad after eos: transformers 4.29 `greedy_search`) is ours -- but every step is the
        # REFERENCE's own arithmetic: its forward with its KV cache (modeling.py:183-195) and its
        # `prepare_inputs_for_generation` (:624-659).  The restated full-recompute loop must give the same ids.
        with torch.no_grad():
            ids_ref = reference_greedy(model.llm, emb.detach(), max_new_tokens=8, eos=2, pad=cfg["tags"]["pad"])
            ids_restated = restate.greedy_generate(sd, emb.detach(), cfg, max_new_tokens=8,
                                                   eos=2, pad=cfg["tags"]["pad"])
        if not torch.equal(ids_ref, ids_restated):
            raise SystemExit(f"restated greedy loop disagrees with the reference's cached decode:\n{ids_ref}\n{ids_restated}")
        fx["generate_ids"] = ids_ref
        fx["generate_ids_source"] = "reference LlamaForCausalLM.forward + past_key_values, greedy loop of make_golden.reference_greedy"
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(fx, path)
    print(f"{name}: S={emb.shape[1]} loss={out.loss.item():.6f} "
          f"{os.path.getsize(path)/1e6:.2f} MB, {len(grads)} grads", file=sys.stderr)


# ---- real-dimension fixture: ONE LLaMA-7B decoder layer + final norm + a slice of lm_head, run by
# the reference's own classes.  The 203 M weights are not stored: both sides regenerate them from
# the seeded recipe in oracle/inputs.py (real7b_layer_weights), bf16-exact values.
def make_real7b_layer(name="real7b_layer", seed=4242, B=2, S=24, head_rows=512):
    mod = ref_loader.load_reference_modeling()
    from transformers import LlamaConfig
    lcfg = LlamaConfig(**configs.get("real_7b")["llama"])
    lcfg._attn_implementation = "eager"
    layer = mod.LlamaDecoderLayer(lcfg).eval()
    norm = mod.LlamaRMSNorm(lcfg.hidden_size, eps=lcfg.rms_norm_eps)
    w = oin.real7b_layer_weights(seed, head_rows)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            p.copy_(w["layer." + n])
        norm.weight.copy_(w["norm.weight"])
    head = w["lm_head.weight"]
    x, am = oin.real7b_layer_inputs(seed, B, S)
    x = x.clone().requires_grad_(True)
    # the mask / position ids exactly as LlamaModel.forward builds them (modeling.py:434-450)
    mask = restate.decoder_mask(am, B, S, torch.float32, x.device)
    pos = torch.arange(S).unsqueeze(0)
    h = layer(x, attention_mask=mask, position_ids=pos)[0]
    logits = torch.nn.functional.linear(norm(h), head)
    # a scalar functional of the logits drives the backward (fixed cotangent)
    cot = oin.real7b_layer_cotangent(seed, B, S, head_rows)
    (logits * cot).sum().backward()
    named = dict(layer.named_parameters())
    fx = dict(seed=seed, B=B, S=S, head_rows=head_rows, layer_out=h.detach(), logits=logits.detach(),
              dx=x.grad.detach(),
              dq_rows=named["self_attn.q_proj.weight"].grad[:8].clone(),
              ddown_rows=named["mlp.down_proj.weight"].grad[:8].clone(),
              dgate_rows=named["mlp.gate_proj.weight"].grad[5000:5008].clone(),
              dnorm1=named["input_layernorm.weight"].grad.clone(),
              grad_norms={n: p.grad.norm().item() for n, p in named.items()})
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(fx, path)
    print(f"{name}: |out| {h.abs().max().item():.3f} |logits| {logits.abs().max().item():.3f} "
          f"{os.path.getsize(path)/1e6:.2f} MB", file=sys.stderr)


def main():
    if not ref_loader.reference_available():
        raise SystemExit("needs /root/reference")
    if "--real7b-only" in sys.argv:
        make_real7b_layer()
        return
    make("micro_all", "micro", batch=2, text_len=12, modalities=("images", "audios", "videos"),
         pad_tail=3, with_generate=True)
    make("micro_image", "micro", batch=1, text_len=9, modalities=("images",), pad_tail=0)
    make_real7b_layer()


if __name__ == "__main__":
    main()

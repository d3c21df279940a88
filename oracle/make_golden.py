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
        with torch.no_grad():
            fx["generate_ids"] = restate.greedy_generate(sd, emb.detach(), cfg, max_new_tokens=8,
                                                         eos=2, pad=cfg["tags"]["pad"])
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(fx, path)
    print(f"{name}: S={emb.shape[1]} loss={out.loss.item():.6f} "
          f"{os.path.getsize(path)/1e6:.2f} MB, {len(grads)} grads", file=sys.stderr)


def main():
    if not ref_loader.reference_available():
        raise SystemExit("needs /root/reference")
    make("micro_all", "micro", batch=2, text_len=12, modalities=("images", "audios", "videos"),
         pad_tail=3, with_generate=True)
    make("micro_image", "micro", batch=1, text_len=9, modalities=("images",), pad_tail=0)


if __name__ == "__main__":
    main()

"""TEST INFRASTRUCTURE ONLY. BASELINE cfg 1 at FULL size: runs the REFERENCE ITSELF
(/root/reference/modeling.py through oracle/ref_loader.py) with CLIP-ViT-L/14 + alignment attention +
the 32-layer LLaMA-7B, image-only, batch 1, fp32, on this container's CPU (8 B parameters = 32 GB of the
62 GB), and a second fixture with the real Whisper-base + 6-frame CLIP video path (audio + video, LLaMA
truncated to 2 layers: the stack is pinned by the first fixture), and a third with the BACKWARD (image + audio, B = 2,
2-layer LLaMA: weights + gradients fit): the reference's own gradients at V = 32,007 / D = 4096 as row subsets + norms.

    python -m oracle.make_golden_cfg1            # writes tests/golden/cfg1_full.pt, real_av_trunc.pt, real_grad_trunc.pt

Weights and inputs are NOT stored: every hot-path parameter is a pure integer-hash function of its name
(oracle/hashweights.py), identical on CPU and GPU, so tests regenerate them.  Stored: the reference's
parameter shapes (pins the key set), its intermediate and final outputs (encoder states, aligned
features, inputs_embeds, INT mask / labels, decoder hidden states at 8 positions after 5 depths, logits at
8 positions x all 32,007 columns, argmax ids at every position, loss).  Follows modeling.py:941-963
(forward), :965-1048 (prefix), :1070-1093 (encoders).
"""
from __future__ import annotations

import copy
import os
import sys
import time

import torch

from . import configs, hashweights as hw, ref_loader, restate

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _positions(S, n=8):
    return sorted({int(round(i * (S - 1) / (n - 1))) for i in range(n)})


def build_hashed_reference(cfg):
    t0 = time.time()
    model = ref_loader.build_reference_model(cfg, seed=0)
    t1 = time.time()
    hot = set(restate.hot_path_state(dict(model.named_parameters())).keys())
    shapes = {}
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n in hot:
                shapes[n] = tuple(p.shape)
                p.copy_(hw.hash_tensor(n, p.shape))
    print(f"  reference built in {t1 - t0:.0f} s, {len(shapes)} hot-path tensors "
          f"({sum(torch.Size(s).numel() for s in shapes.values()) / 1e9:.2f} B values) hashed in {time.time() - t1:.0f} s",
          file=sys.stderr)
    return model.eval(), shapes


GRAD_ROWS = {   # reference gradients stored as row subsets (the full tensors are 64-525 MB each) + the L2 norm of every gradient
    "llm.model.embed_tokens.weight": [0, 1, 2, 3, 777, 12345, 31999, 32000, 32001, 32002, 32003, 32004, 32005, 32006],
    "llm.lm_head.weight": [0, 1, 5, 4242, 32006],
    "llm.model.norm.weight": None,                                     # None = the whole tensor
    "llm.model.layers.1.mlp.down_proj.weight": [0, 1, 2047, 4095],
    "llm.model.layers.1.self_attn.q_proj.weight": [0, 129, 4095],
    "llm.model.layers.0.mlp.gate_proj.weight": [0, 5000, 11007],
    "llm.model.layers.0.self_attn.v_proj.weight": [0, 2048, 4095],
    "llm.model.layers.0.input_layernorm.weight": None,
    "image_align_attention.in_proj_weight": [0, 1, 4096, 4097, 8192, 12287],   # rows of the q / k / v thirds
    "image_align_attention.in_proj_bias": None,
    "image_align_attention.bias_k": None,
    "image_align_attention.out_proj.weight": [0, 4095],
    "audio_align_attention.in_proj_weight": [5, 4101, 8197],
    "audio_align_attention.bias_v": None,
    "audio_align_attention.out_proj.bias": None,
    "project_image.weight": [0, 767],
    "project_image.bias": None,
    "project_audio.weight": [0, 511],
    "transform_image_to_hidden.weight": [0, 4095],
    "transform_audio_to_hidden.bias": None,
}


def run_with_grads(name, cfg, batch, text_len, modalities):
    """forward + BACKWARD of the reference at real dimensions (LLaMA truncated so that weights + gradients fit): the
    training recipe's freeze (run_clm_llms.py:390-393), eval mode (dropout off), loss.backward()"""
    model, shapes = build_hashed_reference(cfg)
    for n, p in model.named_parameters():
        p.requires_grad_("encoder" not in n)
    inp = hw.make_inputs(cfg, batch, text_len, modalities, tag=name, n_prompt=8)
    t0 = time.time()
    out = model(inputs=inp)
    out.loss.backward()
    with torch.no_grad():
        emb, am, lab = model.prepare_inputs_for_generation(inp)
    S = emb.shape[1]
    pos = _positions(S)
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    fx = dict(
        name=name, config_name="real_7b", llama_layers=cfg["llama"]["num_hidden_layers"], text_len=text_len, batch=batch,
        modalities=tuple(modalities), shapes=shapes, positions=pos, n_prompt=8,
        inputs_embeds=emb.detach().clone(), attention_mask=am, labels=lab,
        logits_at=out.logits[:, pos].detach().clone(), argmax_ids=out.logits.argmax(-1),
        logit_absmax=out.logits.abs().max().item(), loss=out.loss.detach().clone(),
        # (float64: a float32 sum of 45 M squares on the CPU is itself off by ~0.5 %)
        grad_norms={n: g.double().norm().item() for n, g in grads.items()},
        grad_absmax={n: g.abs().max().item() for n, g in grads.items()},
        grad_rows={n: (grads[n].detach().clone() if rows is None else grads[n][rows].detach().clone())
                   for n, rows in GRAD_ROWS.items()},
        grad_row_index=GRAD_ROWS,
        no_grad_params=sorted(n for n, p in model.named_parameters() if p.requires_grad and p.grad is None),
        source="reference MM_LLMs.forward + loss.backward() (modeling.py:941-1048, 555-622), CPU fp32, "
               f"torch {torch.__version__}",
    )
    # greedy decode through the REFERENCE's own cached forward (modeling.py:183-195 KV cache, :624-659
    # prepare_inputs_for_generation) from this multimodal prefix, at real width: ids + the top-1 / top-2 margin of every step
    # (a consumer compares ids only where the margin exceeds its own logit error)
    from .make_golden import reference_greedy
    with torch.no_grad():
        ids = reference_greedy(model.llm, emb.detach(), max_new_tokens=12, eos=2, pad=cfg["tags"]["pad"])
        full = torch.cat([emb.detach(), torch.nn.functional.embedding(ids, model.llm.model.embed_tokens.weight)], 1)
        z = model.llm(inputs_embeds=full).logits[:, S - 1:S - 1 + ids.shape[1]]
        top2 = z.topk(2, dim=-1).values
    fx["generate_ids"], fx["generate_margin"] = ids, (top2[..., 0] - top2[..., 1])
    fx["generate_ids_source"] = "reference LlamaForCausalLM.forward + past_key_values, greedy loop of make_golden.reference_greedy"
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(fx, path)
    print(f"  generate ids {ids.tolist()} margins min {fx['generate_margin'].min().item():.4f}", file=sys.stderr)
    print(f"{name}: S={S} loss={out.loss.item():.6f} {len(grads)} gradients, forward + backward {time.time() - t0:.0f} s, "
          f"{os.path.getsize(path) / 1e6:.2f} MB", file=sys.stderr)
    del model
    return fx


def run(name, cfg, text_len, modalities, depths):
    model, shapes = build_hashed_reference(cfg)
    inp = hw.make_inputs(cfg, 1, text_len, modalities, tag=name)
    grabbed = {}
    hooks = []

    def grab(key, pick=lambda o: o):
        def f(_m, _a, out):
            grabbed[key] = pick(out).detach().clone()
        return f

    first = lambda o: o[0]
    hooks.append(model.image_encoder.vision_model.register_forward_hook(grab("clip_last_hidden", first)))
    hooks.append(model.video_encoder.vision_model.register_forward_hook(grab("video_clip_last_hidden", first)))
    hooks.append(model.audio_encoder.encoder.register_forward_hook(grab("whisper_last_hidden", first)))
    hooks.append(model.video_long_self_attention.register_forward_hook(grab("video_long_attn", first)))
    for m in ("image", "audio", "video"):
        hooks.append(getattr(model, f"{m}_align_attention").register_forward_hook(grab(f"{m}_aligned", first)))
    for d in depths:
        hooks.append(model.llm.model.layers[d - 1].register_forward_hook(grab(f"hidden_after_{d}", first)))
    t0 = time.time()
    with torch.no_grad():
        emb, am, lab = model.prepare_inputs_for_generation(inp)
        grabbed_prefix = dict(grabbed)
        out = model(inputs=inp)
    for h in hooks:
        h.remove()
    S = emb.shape[1]
    pos = _positions(S)
    fx = dict(
        name=name, config_name="real_7b", llama_layers=cfg["llama"]["num_hidden_layers"], text_len=text_len,
        modalities=tuple(modalities), shapes=shapes, positions=pos,
        inputs_embeds=emb.detach().clone(), attention_mask=am, labels=lab,
        logits_at=out.logits[0, pos].detach().clone(), argmax_ids=out.logits[0].argmax(-1),
        logit_absmax=out.logits.abs().max().item(), loss=out.loss.detach().clone(),
        hidden_at={d: grabbed[f"hidden_after_{d}"][0, pos] for d in depths},
        source="reference MM_LLMs.forward / prepare_inputs_for_generation (modeling.py:941-1048), CPU fp32, "
               f"torch {torch.__version__}",
    )
    # encoder-side intermediates (row subsets where the tensor is large)
    if "images" in modalities:
        fx["clip_last_hidden"] = grabbed_prefix["clip_last_hidden"][0]                       # [257, 1024]
        fx["image_aligned"] = grabbed_prefix["image_aligned"].transpose(0, 1).contiguous()   # [1, 6, 4096]
    if "audios" in modalities:
        fx["whisper_rows"] = list(range(0, 1500, 25))
        fx["whisper_last_hidden"] = grabbed_prefix["whisper_last_hidden"][0, fx["whisper_rows"]]
        fx["audio_aligned"] = grabbed_prefix["audio_aligned"].transpose(0, 1).contiguous()
    if "videos" in modalities:
        v = grabbed_prefix["video_long_attn"].transpose(0, 1).contiguous()                   # [1, 1536, 768]
        fx["video_rows"] = list(range(0, v.shape[1], 16))
        fx["video_long_attn"] = v[0, fx["video_rows"]]
        fx["video_aligned"] = grabbed_prefix["video_aligned"].transpose(0, 1).contiguous()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(fx, path)
    print(f"{name}: S={S} loss={out.loss.item():.6f} max|logit|={fx['logit_absmax']:.3f} "
          f"forward {time.time() - t0:.0f} s, {os.path.getsize(path) / 1e6:.2f} MB", file=sys.stderr)
    del model
    return fx


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    which = sys.argv[1:] or ["real_av_trunc", "real_grad_trunc", "cfg1_full"]
    if "real_grad_trunc" in which:
        cfg = configs.get("real_7b")
        cfg["llama"]["num_hidden_layers"] = 2
        run_with_grads("real_grad_trunc", cfg, 2, 32, ("images", "audios"))
    if "real_av_trunc" in which:
        cfg = configs.get("real_7b")
        cfg["llama"]["num_hidden_layers"] = 2
        run("real_av_trunc", cfg, 128, ("audios", "videos"), depths=(1, 2))
    if "cfg1_full" in which:
        run("cfg1_full", copy.deepcopy(configs.get("real_7b")), 128, ("images",), depths=(1, 8, 16, 24, 32))


if __name__ == "__main__":
    main()

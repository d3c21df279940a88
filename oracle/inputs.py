"""TEST INFRASTRUCTURE ONLY. Seeded synthetic inputs following SURVEY.md §8(d): images / video
frames ~ N(0,1) (post-normalisation statistics, llm_trainer.py:157), log-mel ~ N(0,1)*0.5 clipped
to [-1,1.5], input_ids uniform over the text vocabulary with BOS first, labels = input_ids with
the first `n_prompt` positions set to -100, optional right padding, and the modality tag ids."""
from __future__ import annotations

import torch


def make_inputs(cfg: dict, batch: int, text_len: int, modalities=("images", "audios", "videos"),
                seed: int = 1, pad_tail: int = 0, n_prompt: int = 4, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    v = cfg["clip"]["vision_config"]
    w = cfg["whisper"]
    tags = cfg["tags"]
    img = v["image_size"]
    n_text_vocab = tags["image"][0]  # ids below the first tag are ordinary tokens
    inputs = {}
    inputs["images"] = (torch.randn(batch, 3, img, img, generator=g).to(dtype)
                        if "images" in modalities else None)
    mel_len = w["max_source_positions"] * 2
    inputs["audios"] = ((torch.randn(batch, w["num_mel_bins"], mel_len, generator=g) * 0.5)
                        .clamp(-1.0, 1.5).to(dtype) if "audios" in modalities else None)
    inputs["videos"] = (torch.randn(batch, cfg["mm"]["n_frames"], 3, img, img, generator=g).to(dtype)
                        if "videos" in modalities else None)
    ids = torch.randint(3, n_text_vocab, (batch, text_len), generator=g, dtype=torch.int64)
    ids[:, 0] = 1
    am = torch.ones(batch, text_len, dtype=torch.int64)
    labels = ids.clone()
    labels[:, :n_prompt] = -100
    if pad_tail > 0:  # right padding on the odd rows, like the tokenizer's padding to 256
        for b in range(1, batch, 2):
            ids[b, -pad_tail:] = tags["pad"]
            am[b, -pad_tail:] = 0
            labels[b, -pad_tail:] = -100
    inputs.update(input_ids=ids, attention_mask=am, labels=labels)
    for name in ("image", "audio", "video"):
        s, e = tags[name]
        inputs[f"{name}_starts"] = torch.full((batch,), s, dtype=torch.int32)
        inputs[f"{name}_ends"] = torch.full((batch,), e, dtype=torch.int32)
    return inputs


# ---- seeded recipe of the real-dimension single-layer fixture (oracle/make_golden.py
# make_real7b_layer and tests/test_fullsize_gpu.py regenerate identical tensors from it)
def _bf16_exact(t):
    return t.to(torch.bfloat16).to(torch.float32)


def real7b_layer_weights(seed: int, head_rows: int = 512):
    """weights of one LLaMA-7B decoder layer (reference parameter names under 'layer.'), the
    final norm and `head_rows` rows of lm_head; N(0, 0.02) like HF init, norm weights around 1,
    every value exactly representable in bf16."""
    g = torch.Generator().manual_seed(seed)
    D, FF = 4096, 11008
    shapes = [("layer.self_attn.q_proj.weight", (D, D)), ("layer.self_attn.k_proj.weight", (D, D)),
              ("layer.self_attn.v_proj.weight", (D, D)), ("layer.self_attn.o_proj.weight", (D, D)),
              ("layer.mlp.gate_proj.weight", (FF, D)), ("layer.mlp.down_proj.weight", (D, FF)),
              ("layer.mlp.up_proj.weight", (FF, D))]
    w = {n: _bf16_exact(torch.randn(s, generator=g) * 0.02) for n, s in shapes}
    for n in ("layer.input_layernorm.weight", "layer.post_attention_layernorm.weight", "norm.weight"):
        w[n] = _bf16_exact(1.0 + 0.1 * torch.randn(D, generator=g))
    w["lm_head.weight"] = _bf16_exact(torch.randn(head_rows, D, generator=g) * 0.02)
    return w


def real7b_layer_inputs(seed: int, B: int, S: int):
    g = torch.Generator().manual_seed(seed + 1)
    x = _bf16_exact(torch.randn(B, S, 4096, generator=g))
    am = torch.ones(B, S, dtype=torch.long)
    am[-1, S - 5:] = 0                      # right padding on the last sample
    return x, am


def real7b_layer_cotangent(seed: int, B: int, S: int, head_rows: int):
    g = torch.Generator().manual_seed(seed + 2)
    return _bf16_exact(torch.randn(B, S, head_rows, generator=g) / 64)

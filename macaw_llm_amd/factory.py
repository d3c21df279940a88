"""Model construction helpers (setup time, not the hot path): build MM_LLMs from plain config
dicts directly on the HIP device in the training dtype, random HF-default init."""
from __future__ import annotations

import contextlib

import torch
from transformers import CLIPConfig, LlamaConfig, WhisperConfig

from . import modeling as M


@contextlib.contextmanager
def _default_dtype(dtype):
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def make_config(cfg: dict) -> M.MM_LLMs_Config:
    return M.MM_LLMs_Config(clip_config=CLIPConfig(**cfg["clip"]),
                            whisper_config=WhisperConfig(**cfg["whisper"]),
                            llm_config=LlamaConfig(**cfg["llama"]), **cfg["mm"])


def build_model(cfg: dict, dtype=torch.bfloat16, device="cuda", seed=1234, freeze_encoders=True,
                fuse=True):
    """Random-init model of the given architecture, parameters created on `device` in `dtype`.
    freeze_encoders mirrors run_clm_llms.py:390-393 (every '*encoder*' parameter frozen)."""
    torch.manual_seed(seed)
    with _default_dtype(dtype), torch.device(device):
        model = M.MM_LLMs(make_config(cfg))
    model = model.to(device=device, dtype=dtype)
    if fuse:   # (a model built without this fuses lazily at its first forward: modeling.AUTO_FUSE)
        for layer in model.llm.model.layers:
            layer.fuse_projections()
    if freeze_encoders:
        for n, p in model.named_parameters():
            p.requires_grad_("encoder" not in n)
    return model


# BASELINE.json configurations
def baseline_config(name: str = "real_7b") -> dict:
    v = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
             image_size=224, patch_size=14, projection_dim=768)
    t = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
             projection_dim=768)
    whisper = dict(d_model=512, encoder_layers=6, encoder_attention_heads=8, encoder_ffn_dim=2048,
                   decoder_layers=6, decoder_attention_heads=8, decoder_ffn_dim=2048,
                   max_source_positions=1500, num_mel_bins=80, vocab_size=51865)
    if name == "real_7b":
        ll = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                  num_attention_heads=32, num_key_value_heads=32)
    elif name == "real_13b":
        ll = dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40,
                  num_attention_heads=40, num_key_value_heads=40)
    else:
        raise KeyError(name)
    ll.update(vocab_size=32007, max_position_embeddings=2048, rms_norm_eps=1e-6, hidden_act="silu",
              pad_token_id=0, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
    return dict(clip=dict(vision_config=v, text_config=t, projection_dim=768), whisper=whisper,
                llama=ll,
                mm=dict(n_frames=6, attention_heads=8, image_conv_kernel=48, image_conv_stride=36,
                        video_conv_kernel=36, video_conv_stride=30, audio_conv_kernel=240,
                        audio_conv_stride=220),
                tags=dict(image=(32000, 32001), audio=(32002, 32003), video=(32004, 32005), pad=32006))


def synthetic_inputs(cfg: dict, batch: int, text_len: int, modalities=("images", "audios"),
                     seed: int = 1, device="cuda", dtype=torch.float16, n_prompt: int = 32):
    """Seeded synthetic batch of the shape BASELINE.json names: N(0,1) images /
    frames, clipped N(0,1)*0.5 log-mel, uniform token ids with BOS first, labels = ids with the
    first n_prompt positions ignored.  Float inputs arrive as fp16 like llm_trainer.py:366-368."""
    g = torch.Generator().manual_seed(seed)
    v, w, tags = cfg["clip"]["vision_config"], cfg["whisper"], cfg["tags"]
    img = v["image_size"]
    out = dict(images=None, audios=None, videos=None)
    if "images" in modalities:
        out["images"] = torch.randn(batch, 3, img, img, generator=g).to(dtype)
    if "audios" in modalities:
        out["audios"] = (torch.randn(batch, w["num_mel_bins"], w["max_source_positions"] * 2, generator=g) * 0.5
                         ).clamp(-1.0, 1.5).to(dtype)
    if "videos" in modalities:
        out["videos"] = torch.randn(batch, cfg["mm"]["n_frames"], 3, img, img, generator=g).to(dtype)
    ids = torch.randint(3, tags["image"][0], (batch, text_len), generator=g, dtype=torch.int64)
    ids[:, 0] = 1
    labels = ids.clone()
    labels[:, :n_prompt] = -100
    out.update(input_ids=ids, attention_mask=torch.ones(batch, text_len, dtype=torch.int64), labels=labels)
    for name in ("image", "audio", "video"):
        s, e = tags[name]
        out[f"{name}_starts"] = torch.full((batch,), s, dtype=torch.int32)
        out[f"{name}_ends"] = torch.full((batch,), e, dtype=torch.int32)
    return {k: (t.to(device) if torch.is_tensor(t) else t) for k, t in out.items()}

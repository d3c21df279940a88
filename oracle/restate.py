"""TEST INFRASTRUCTURE ONLY — the CPU oracle for Macaw-LLM's multimodal forward path.

A plain-PyTorch (CPU, fp32 or fp64) *restatement* of the algorithm the reference executes
on its hot path, written functionally over a state dict (reference parameter names), so
that it can run where /root/reference does not exist (the GPU box).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module; the
product path (macaw_llm_amd/) never does.

Pinned against the reference itself: tests/test_oracle_vs_reference.py imports the real
`/root/reference/modeling.py` (oracle/ref_loader.py) and requires every function here to
reproduce it, and oracle/make_golden.py stores reference outputs under tests/golden/.
The reference ships no tests or golden vectors of its own, and the
arithmetic of CLIP / Whisper / nn.MultiheadAttention lives in un-vendored packages
(transformers==4.29.0, torch==2.0.0 in requirements.txt:1,24); here they are restated from
the versions installed in this image (transformers 5.x, torch 2.10) — see SURVEY §8(c).

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# Test hook (tests/fp8_ref.py): the "what does the e4m3 FORMAT itself cost" yardstick for the fp8
# path of BASELINE cfg 5 swaps a fake-quantised linear in at the sites the fp8 MFMA path covers --
# "qkv" (modeling.py:179-181), "align" (the K/V projection of the token table, :882-910), "mlp"
# (:139-140).  None / empty = plain F.linear everywhere, i.e. the reference's arithmetic.
FP8_LINEAR = None
FP8_SITES = ()


def _lin(x, W, b=None, site=""):
    if FP8_LINEAR is not None and site in FP8_SITES:
        return FP8_LINEAR(x, W, b)
    return F.linear(x, W, b)


# ------------------------------------------------------------------ LLaMA ---
def rms_norm(x, w, eps):
    """modeling.py:311-319 (LlamaRMSNorm.forward)."""
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    h = x * torch.rsqrt(var + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        h = h.to(w.dtype)
    return w * h


def rotary_tables(hd, max_pos, base=10000.0, device=None):
    """modeling.py:95-107 (LlamaRotaryEmbedding.__init__): cos/sin [max_pos, hd] fp32."""
    inv_freq = 1.0 / (base ** (torch.arange(0, hd, 2, device=device).float() / hd))
    t = torch.arange(max_pos, device=device, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    """modeling.py:76-80."""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin, position_ids):
    """modeling.py:83-91; q,k [B,H,S,hd]; cos/sin [max_pos,hd] already in q.dtype."""
    c = cos[position_ids].unsqueeze(1)
    s = sin[position_ids].unsqueeze(1)
    return (q * c) + (rotate_half(q) * s), (k * c) + (rotate_half(k) * s)


def decoder_mask(attention_mask, bsz, tgt, dtype, device):
    """modeling.py:44-73,373-394: causal(finfo.min) + expanded padding mask, [B,1,S,S]."""
    minv = torch.finfo(dtype).min
    mask = torch.full((tgt, tgt), minv, device=device, dtype=torch.float32)
    cond = torch.arange(tgt, device=device)
    mask.masked_fill_(cond < (cond + 1).view(tgt, 1), 0)
    mask = mask.to(dtype)[None, None].expand(bsz, 1, tgt, tgt)
    if attention_mask is not None:
        exp = attention_mask[:, None, None, :].expand(bsz, 1, tgt, tgt).to(dtype)
        inv = 1.0 - exp
        inv = inv.masked_fill(inv.to(torch.bool), minv)
        mask = inv + mask
    return mask


def llama_attention(sd: SD, p: str, x, mask, position_ids, n_heads, cos, sin):
    """modeling.py:168-231 (LlamaAttention.forward, no KV cache)."""
    B, S, D = x.shape
    hd = D // n_heads
    q = _lin(x, sd[p + "q_proj.weight"], site="qkv").view(B, S, n_heads, hd).transpose(1, 2)
    k = _lin(x, sd[p + "k_proj.weight"], site="qkv").view(B, S, n_heads, hd).transpose(1, 2)
    v = _lin(x, sd[p + "v_proj.weight"], site="qkv").view(B, S, n_heads, hd).transpose(1, 2)
    q, k = apply_rope(q, k, cos.to(x.dtype), sin.to(x.dtype), position_ids)
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd)
    if mask is not None:
        w = w + mask
        w = torch.max(w, torch.tensor(torch.finfo(w.dtype).min))
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, S, D)
    return F.linear(o, sd[p + "o_proj.weight"])


def llama_mlp(sd: SD, p: str, x):
    """modeling.py:139-140."""
    return _lin(F.silu(_lin(x, sd[p + "gate_proj.weight"], site="mlp")) * _lin(x, sd[p + "up_proj.weight"], site="mlp"),
                sd[p + "down_proj.weight"], site="mlp")


def llama_layer(sd: SD, p: str, x, mask, position_ids, n_heads, eps, cos, sin):
    """modeling.py:247-299 (LlamaDecoderLayer.forward)."""
    h = x + llama_attention(sd, p + "self_attn.", rms_norm(x, sd[p + "input_layernorm.weight"], eps),
                            mask, position_ids, n_heads, cos, sin)
    return h + llama_mlp(sd, p + "mlp.", rms_norm(h, sd[p + "post_attention_layernorm.weight"], eps))


def llama_forward(sd: SD, p: str, inputs_embeds, attention_mask, cfg, labels=None,
                  position_ids=None, hidden_states=None):
    """modeling.py:397-522 (LlamaModel.forward) + 555-622 (LlamaForCausalLM.forward).
    p = 'llm.'; returns (loss|None, logits).  hidden_states: optional dict, filled with {depth: output of
    decoder layer `depth` (1-based)} -- what `output_hidden_states` (:452-456,491-493) exposes."""
    B, S, D = inputs_embeds.shape
    n_layers, n_heads, eps = cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["rms_norm_eps"]
    if position_ids is None:
        position_ids = torch.arange(S, device=inputs_embeds.device).unsqueeze(0)  # :434-439
    if attention_mask is None:
        attention_mask = torch.ones((B, S), dtype=torch.bool, device=inputs_embeds.device)
    mask = decoder_mask(attention_mask, B, S, inputs_embeds.dtype, inputs_embeds.device)
    cos, sin = rotary_tables(D // n_heads, max(cfg.get("max_position_embeddings", 2048), S),
                             device=inputs_embeds.device)
    h = inputs_embeds
    for i in range(n_layers):
        h = llama_layer(sd, f"{p}model.layers.{i}.", h, mask, position_ids, n_heads, eps, cos, sin)
        if hidden_states is not None:
            hidden_states[i + 1] = h
    h = rms_norm(h, sd[p + "model.norm.weight"], eps)
    logits = F.linear(h, sd[p + "lm_head.weight"])
    loss = None
    if labels is not None:  # :600-610
        V = logits.shape[-1]
        loss = F.cross_entropy(logits[..., :-1, :].contiguous().view(-1, V),
                               labels[..., 1:].contiguous().view(-1))
    return loss, logits


# ------------------------------------------------- nn.MultiheadAttention ---
def mha_forward(sd: SD, p: str, query, key, value, n_heads, share_kv_proj=None):
    """torch.nn.MultiheadAttention(add_bias_kv=True, add_zero_attn=True), eval mode,
    seq-first [L,B,E] (modeling.py:882-910; called :986,1007,1025,1078).  Restates
    torch.nn.functional.multi_head_attention_forward: packed in-proj, append bias_k/bias_v
    row, append a zero row, softmax(q k^T / sqrt(hd)), out-proj.  Returns [L,B,E]."""
    L, B, E = query.shape
    S = key.shape[0]
    hd = E // n_heads
    W, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(query, W[:E], b[:E])
    k = F.linear(key, W[E:2 * E], b[E:2 * E])
    v = F.linear(value, W[2 * E:], b[2 * E:])
    k = torch.cat([k, sd[p + "bias_k"].repeat(1, B, 1)], dim=0)
    v = torch.cat([v, sd[p + "bias_v"].repeat(1, B, 1)], dim=0)
    q = q.reshape(L, B * n_heads, hd).transpose(0, 1)
    k = k.reshape(S + 1, B * n_heads, hd).transpose(0, 1)
    v = v.reshape(S + 1, B * n_heads, hd).transpose(0, 1)
    zeros = torch.zeros((B * n_heads, 1, hd), dtype=k.dtype, device=k.device)
    k = torch.cat([k, zeros], dim=1)
    v = torch.cat([v, zeros], dim=1)
    att = torch.bmm(q * math.sqrt(1.0 / hd), k.transpose(1, 2))
    att = F.softmax(att, dim=-1)
    o = torch.bmm(att, v).transpose(0, 1).reshape(L * B, E)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"]).view(L, B, E)


def mha_forward_hoisted(sd: SD, p: str, query, table, n_heads):
    """Same result as mha_forward(query, table.repeat(B), table.repeat(B)) but with the K/V
    projection of the (batch-independent) embedding table done once (bit-exact).
    query [L,B,E], table [V,E]."""
    L, B, E = query.shape
    hd = E // n_heads
    W, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(query, W[:E], b[:E])
    k = torch.cat([_lin(table, W[E:2 * E], b[E:2 * E], site="align"), sd[p + "bias_k"].view(1, E),
                   torch.zeros(1, E, dtype=table.dtype, device=table.device)], dim=0)
    v = torch.cat([_lin(table, W[2 * E:], b[2 * E:], site="align"), sd[p + "bias_v"].view(1, E),
                   torch.zeros(1, E, dtype=table.dtype, device=table.device)], dim=0)
    S2 = k.shape[0]
    qh = q.reshape(L * B, n_heads, hd).transpose(0, 1)        # [H, L*B, hd]
    kh = k.view(S2, n_heads, hd).transpose(0, 1)               # [H, S2, hd]
    vh = v.view(S2, n_heads, hd).transpose(0, 1)
    att = F.softmax(torch.bmm(qh * math.sqrt(1.0 / hd), kh.transpose(1, 2)), dim=-1)
    o = torch.bmm(att, vh).transpose(0, 1).reshape(L * B, E)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"]).view(L, B, E)


# ------------------------------------------------------------ encoders ------
def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def _act(name):
    return {"quick_gelu": quick_gelu, "gelu": F.gelu}[name]


def encoder_self_attn(sd: SD, p: str, x, n_heads, k_bias=True):
    """HF CLIPAttention / WhisperAttention (eager): softmax(q k^T * hd^-0.5) v, fp32 softmax."""
    B, T, E = x.shape
    hd = E // n_heads
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"] if k_bias else None)
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    q, k, v = (t.view(B, T, n_heads, hd).transpose(1, 2) for t in (q, k, v))
    w = torch.matmul(q, k.transpose(-1, -2)) * hd ** -0.5
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, T, E)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def clip_vision_forward(sd: SD, p: str, pixel_values, vcfg):
    """HF CLIPModel.vision_model(...)[0] as used at modeling.py:1073,1092: patch-embed conv
    (no bias) -> [CLS]+pos -> pre_layrnorm -> N x pre-LN encoder layers (quick_gelu) ->
    last_hidden_state (NO post_layernorm).  p = 'image_encoder.vision_model.'"""
    eps = vcfg.get("layer_norm_eps", 1e-5)
    heads = vcfg["num_attention_heads"]
    act = _act(vcfg.get("hidden_act", "quick_gelu"))
    B = pixel_values.shape[0]
    pe = F.conv2d(pixel_values, sd[p + "embeddings.patch_embedding.weight"], stride=vcfg["patch_size"])
    pe = pe.flatten(2).transpose(1, 2)
    cls = sd[p + "embeddings.class_embedding"].expand(B, 1, -1)
    h = torch.cat([cls, pe], dim=1) + sd[p + "embeddings.position_embedding.weight"][None]
    E = h.shape[-1]
    h = F.layer_norm(h, (E,), sd[p + "pre_layrnorm.weight"], sd[p + "pre_layrnorm.bias"], eps)
    for i in range(vcfg["num_hidden_layers"]):
        lp = f"{p}encoder.layers.{i}."
        r = h
        h = F.layer_norm(h, (E,), sd[lp + "layer_norm1.weight"], sd[lp + "layer_norm1.bias"], eps)
        h = r + encoder_self_attn(sd, lp + "self_attn.", h, heads)
        r = h
        h = F.layer_norm(h, (E,), sd[lp + "layer_norm2.weight"], sd[lp + "layer_norm2.bias"], eps)
        h = F.linear(act(F.linear(h, sd[lp + "mlp.fc1.weight"], sd[lp + "mlp.fc1.bias"])),
                     sd[lp + "mlp.fc2.weight"], sd[lp + "mlp.fc2.bias"])
        h = r + h
    return h


def encode_image(sd: SD, enc: str, images, vcfg):
    """modeling.py:1085-1093 (encode_image): visual_projection(vision_model(x)[0])[:,1:,:]."""
    h = clip_vision_forward(sd, enc + "vision_model.", images, vcfg)
    return F.linear(h, sd[enc + "visual_projection.weight"])[:, 1:, :]


def whisper_encoder_forward(sd: SD, p: str, mel, wcfg):
    """HF WhisperModel.encoder(mel)[0] (modeling.py:1081-1083). p='audio_encoder.encoder.'"""
    heads = wcfg["encoder_attention_heads"]
    act = _act(wcfg.get("activation_function", "gelu"))
    h = F.gelu(F.conv1d(mel, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1))
    h = F.gelu(F.conv1d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], stride=2, padding=1))
    h = h.permute(0, 2, 1) + sd[p + "embed_positions.weight"]
    E = h.shape[-1]
    for i in range(wcfg["encoder_layers"]):
        lp = f"{p}layers.{i}."
        r = h
        h = F.layer_norm(h, (E,), sd[lp + "self_attn_layer_norm.weight"], sd[lp + "self_attn_layer_norm.bias"], 1e-5)
        h = r + encoder_self_attn(sd, lp + "self_attn.", h, heads, k_bias=False)
        r = h
        h = F.layer_norm(h, (E,), sd[lp + "final_layer_norm.weight"], sd[lp + "final_layer_norm.bias"], 1e-5)
        h = F.linear(act(F.linear(h, sd[lp + "fc1.weight"], sd[lp + "fc1.bias"])),
                     sd[lp + "fc2.weight"], sd[lp + "fc2.bias"])
        h = r + h
    return F.layer_norm(h, (E,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)


def positional_encoding(L, h, device=None):
    """modeling.py:1095-1106 (create_positional_encoding), vectorised.  NOTE the reference's
    non-textbook exponent: i already steps by 2 and is doubled again (2*i/h)."""
    i = torch.arange(0, h, 2, dtype=torch.float32, device=device)
    div = torch.exp(-(math.log(10000.0) / h * (2 * i)))
    pos = torch.arange(L, dtype=torch.float32, device=device)[:, None]
    pe = torch.zeros(L, h, device=device)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def positional_encoding_loop(L, h):
    """modeling.py:1095-1106 literally (python double loop) — for small L only."""
    pe = torch.zeros(L, h)
    for pos in range(L):
        for i in range(0, h, 2):
            d = torch.exp(torch.tensor(-(math.log(10000.0) / h * (2 * i))))
            pe[pos, i] = torch.sin(pos * d)
            pe[pos, i + 1] = torch.cos(pos * d)
    return pe


def encode_video_long(sd: SD, videos, vcfg, n_frames, heads):
    """modeling.py:1070-1079."""
    fr = videos.view(-1, videos.size(-3), videos.size(-2), videos.size(-1))
    f = encode_image(sd, "video_encoder.", fr, vcfg)
    f = f.reshape(fr.size(0) // n_frames, n_frames * f.size(1), -1).contiguous()
    f = f + positional_encoding(f.size(1), f.size(2), f.device).to(f.dtype)[None]
    f = f.transpose(0, 1).contiguous()
    return mha_forward(sd, "video_long_self_attention.", f, f, f, heads).transpose(0, 1).contiguous()


# --------------------------------------------------------- MM_LLMs forward --
def prepare_inputs(sd: SD, inputs: dict, cfg: dict, hoist: bool = True):
    """modeling.py:965-1048 (MM_LLMs.prepare_inputs_for_generation).
    Returns (inputs_embeds [B,S,D], attention_mask|None, labels|None, aux dict)."""
    mm, vcfg, wcfg = cfg["mm"], cfg["clip"]["vision_config"], cfg["whisper"]
    heads = mm["attention_heads"]
    E = sd["llm.model.embed_tokens.weight"]
    aux = {}
    image_f = encode_image(sd, "image_encoder.", inputs["images"], vcfg) if inputs.get("images") is not None else None
    audio_f = whisper_encoder_forward(sd, "audio_encoder.encoder.", inputs["audios"], wcfg) if inputs.get("audios") is not None else None
    video_f = encode_video_long(sd, inputs["videos"], vcfg, mm["n_frames"], heads) if inputs.get("videos") is not None else None
    aux.update(image_features=image_f, audio_features=audio_f, video_features=video_f)
    text = F.embedding(inputs["input_ids"].long(), E)
    B = text.size(0)
    ignore = 0

    def align(name, feats, kernel, stride):
        nonlocal text, ignore
        starts = F.embedding(inputs[f"{name}_starts"].long(), E).unsqueeze(1)
        ends = F.embedding(inputs[f"{name}_ends"].long(), E).unsqueeze(1)
        f = F.conv1d(feats.transpose(1, 2).contiguous(), sd[f"project_{name}.weight"],
                     sd[f"project_{name}.bias"], stride=stride).transpose(1, 2).contiguous()
        f = F.linear(f, sd[f"transform_{name}_to_hidden.weight"], sd[f"transform_{name}_to_hidden.bias"])
        q = f.transpose(0, 1)
        if hoist:
            a = mha_forward_hoisted(sd, f"{name}_align_attention.", q, E, heads * 2)
        else:
            tok = E.unsqueeze(0).repeat(B, 1, 1).transpose(0, 1)  # :974-975
            a = mha_forward(sd, f"{name}_align_attention.", q, tok, tok, heads * 2)
        a = a.transpose(0, 1).contiguous()
        aux[f"{name}_aligned"] = a
        block = torch.cat([starts, a, ends], dim=1)
        text = torch.cat([text[:, 0:1], block, text[:, 1:]], dim=1)
        ignore += block.size(1)

    if video_f is not None:
        align("video", video_f, mm["video_conv_kernel"], mm["video_conv_stride"])
    if audio_f is not None:
        align("audio", audio_f, mm["audio_conv_kernel"], mm["audio_conv_stride"])
    if image_f is not None:
        align("image", image_f, mm["image_conv_kernel"], mm["image_conv_stride"])

    am = None
    if "attention_mask" in inputs and inputs["attention_mask"] is not None:
        am = torch.cat([torch.ones(B, ignore, dtype=inputs["attention_mask"].dtype,
                                   device=inputs["attention_mask"].device), inputs["attention_mask"]], dim=1)
    lab = None
    if inputs.get("labels") is not None:
        lab = torch.cat([torch.full((B, ignore), -100, dtype=inputs["labels"].dtype,
                                    device=inputs["labels"].device), inputs["labels"]], dim=1)
    return text, am, lab, aux


def mm_forward(sd: SD, inputs: dict, cfg: dict, hoist: bool = True, hidden_states=None):
    """modeling.py:941-963 (MM_LLMs.forward, training branch). Returns dict(loss, logits, ...)."""
    emb, am, lab, aux = prepare_inputs(sd, inputs, cfg, hoist)
    loss, logits = llama_forward(sd, "llm.", emb, am, cfg["llama"], labels=lab, hidden_states=hidden_states)
    aux.update(inputs_embeds=emb, attention_mask=am, labels=lab, loss=loss, logits=logits)
    return aux


def greedy_generate(sd: SD, inputs_embeds, cfg, max_new_tokens=128, eos=2, pad=32006):
    """Restated greedy decode for modeling.py:954-960 (`llm.generate(inputs_embeds=...,
    max_new_tokens=128, eos_token_id=2, bos_token_id=1, pad_token_id=32006)`, no attention
    mask).  HF GenerationMixin is unavailable for the reference under transformers 5.x
    (SURVEY §8c), so this is a full-recompute loop over llama_forward (no KV cache): same
    arithmetic as the cached path.  Returns new token ids [B, <=max_new_tokens]."""
    E = sd["llm.model.embed_tokens.weight"]
    B = inputs_embeds.shape[0]
    emb = inputs_embeds
    out = []
    done = torch.zeros(B, dtype=torch.bool, device=inputs_embeds.device)
    for _ in range(max_new_tokens):
        _, logits = llama_forward(sd, "llm.", emb, None, cfg["llama"])
        nxt = logits[:, -1, :].argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
        out.append(nxt)
        done = done | (nxt == eos)
        if bool(done.all()):
            break
        emb = torch.cat([emb, F.embedding(nxt, E).unsqueeze(1)], dim=1)
    return torch.stack(out, dim=1)


HOT_PATH_PREFIXES = (
    "image_encoder.vision_model.", "image_encoder.visual_projection.",
    "video_encoder.vision_model.", "video_encoder.visual_projection.",
    "audio_encoder.encoder.", "llm.", "image_align_attention.", "audio_align_attention.",
    "video_align_attention.", "video_long_self_attention.", "transform_", "project_",
)


def hot_path_state(sd: SD) -> SD:
    """Subset of a reference state dict that the forward path reads (drops CLIP text towers,
    Whisper decoder, temporal_*, logit_scale, layer_norm, post_layernorm — SURVEY Q14)."""
    return {k: v for k, v in sd.items()
            if k.startswith(HOT_PATH_PREFIXES) and ".post_layernorm." not in k}

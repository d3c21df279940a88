"""Drop-in mirror of the reference's `modeling.py` surface on the MI355X-native engine.

Same class names, constructor kwargs, attribute names and state-dict keys as
/root/reference/modeling.py (MM_LLMs_Config :807-861, MM_LLMs :863-1093, the vendored
LLaMA classes :44-659), so `from modeling import MM_LLMs, MM_LLMs_Config` in
run_clm_llms.py:95 / llm_trainer.py:114 keeps working (a `modeling.py` shim at the repo root
re-exports this module).  Every arithmetic op of forward and backward runs in the
hand-written gfx950 kernels of csrc/ through macaw_llm_amd.engine; there is no eager
fallback — calling the model on CPU tensors raises MacawHipError.

The CLIP / Whisper towers are instantiated from the installed `transformers` classes purely
as PARAMETER CONTAINERS (identical module tree => identical checkpoint keys, including the
never-used text tower / decoder); their HF `forward` is never called.
"""
from __future__ import annotations

import copy
import math
import os
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn
from torch.nn import CrossEntropyLoss
from transformers import (CLIPConfig, CLIPModel, LlamaConfig, PretrainedConfig, PreTrainedModel,
                          WhisperConfig, WhisperModel)
from transformers.modeling_outputs import BaseModelOutputWithPast, CausalLMOutputWithPast

from . import engine as eng
from . import ops
from .lib import MacawHipError


def _dev_check(t: torch.Tensor):
    if not t.is_cuda:
        raise MacawHipError("macaw_llm_amd runs on the HIP device only (model and inputs must be on "
                            "'cuda'); there is no CPU path")


def _dtype_check(dtype):
    """bf16, fp16 (the reference's scripts: train.sh `--fp16 True`, llm_trainer.py:411-412
    `.to(torch.float16)`) and fp32 (parity mode) parameters are implemented: the same kernels
    instantiated per element type (csrc/common.h E16<>), fp32 accumulation everywhere."""
    if dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise MacawHipError(f"macaw_llm_amd: model parameters are {dtype}; the gfx950 kernels implement bfloat16, "
                            "float16 and float32")


# Parameters that feed one GEMM (q|k|v, gate|up) are re-homed back to back in ONE buffer the first
# time a layer runs on the device (and again after .to() / .cuda() / any re-materialisation gave
# them separate storage), so a model built the reference's way -- MM_LLMs(config), .to(), Trainer --
# runs the same fused GEMMs as factory.build_model.  MACAW_NO_AUTO_FUSE=1 disables it.
AUTO_FUSE = os.environ.get("MACAW_NO_AUTO_FUSE") is None


def _rows_view(ts):
    """tensor [sum rows, ...] aliasing `ts` if they lie back to back in one storage, else None"""
    t0 = ts[0]
    ptr, es = t0.data_ptr(), t0.element_size()
    inner = t0[0].numel() if t0.dim() > 1 else 1
    rows = 0
    for t in ts:
        if (not t.is_contiguous() or t.data_ptr() != ptr + rows * inner * es or t.dtype != t0.dtype
                or t.shape[1:] != t0.shape[1:] or t.device != t0.device):
            return None
        rows += t.shape[0]
    try:
        return t0.data.as_strided((rows,) + tuple(t0.shape[1:]), t0.stride())
    except RuntimeError:   # not inside one storage
        return None


_PIN_WARNED = [False]


@torch.no_grad()
def _rehome_rows(ts):
    """concatenate along dim 0 into one new buffer and make every tensor a view of it.  Refused
    (returns None) for parameters that live in a BucketedStep's flat buckets: moving them would
    leave the optimizer and the collectives updating the bucket while the model computes with the
    copy (build the runtime with model= / call fuse_model() first to get the fused GEMMs)."""
    if any(ops.is_pinned(t.data) for t in ts):
        if not _PIN_WARNED[0]:
            _PIN_WARNED[0] = True
            import warnings
            warnings.warn("macaw_llm_amd: q|k|v / gate|up parameters already live in BucketedStep buckets and "
                          "were not fused; the layer runs one GEMM per projection (correct, slower).  Pass "
                          "model= to BucketedStep or call modeling.fuse_model(model) before building it.")
        return None
    fused = torch.cat([t.data for t in ts], dim=0).contiguous()
    off = 0
    for t in ts:
        t.data = fused[off:off + t.shape[0]]
        off += t.shape[0]
    return fused


def fused_encoder_qkv(attn):
    """([3E, E] weight, [3E] bias) aliasing the q/k/v projections of a HF CLIPAttention /
    WhisperAttention module (one GEMM with N = 3E instead of three).  The parameters themselves
    are re-homed (same state-dict keys), so there is no copy that could go stale when the towers
    are trained or reloaded; a missing k bias (Whisper) is a zero slot of the bias buffer.
    Returns (None, None) when fusing is off or the module is not on the device."""
    q, k, v = attn.q_proj, attn.k_proj, attn.v_proj
    ws = (q.weight, k.weight, v.weight)
    E = q.weight.shape[0]
    W3 = _rows_view(ws)
    b3 = getattr(attn, "_macaw_b3", None)
    es = q.weight.element_size()

    def bias_ok():
        if b3 is None or b3.dtype != q.weight.dtype or b3.device != q.weight.device or b3.numel() != 3 * E:
            return False
        if q.bias is None or v.bias is None:
            return False
        ok = q.bias.data_ptr() == b3.data_ptr() and v.bias.data_ptr() == b3.data_ptr() + 2 * E * es
        return ok and (k.bias is None or k.bias.data_ptr() == b3.data_ptr() + E * es)

    if W3 is not None and bias_ok():
        return W3, b3
    if not (AUTO_FUSE and q.weight.is_cuda) or q.bias is None or v.bias is None:
        return None, None
    with torch.no_grad():
        if W3 is None:
            if _rehome_rows(ws) is None:
                return None, None
            W3 = _rows_view(ws)
        if any(lin.bias is not None and ops.is_pinned(lin.bias.data) for lin in (q, k, v)):
            return None, None          # biases pinned in an optimizer bucket: keep the per-projection GEMMs
        b3 = torch.empty(3 * E, dtype=q.weight.dtype, device=q.weight.device)
        ops.fill_(b3, 0.0)
        for i, lin in enumerate((q, k, v)):
            if lin.bias is not None:
                b3[i * E:(i + 1) * E].copy_(lin.bias.data)
                lin.bias.data = b3[i * E:(i + 1) * E]
        attn._macaw_b3 = b3
    return W3, b3


def fuse_model(model):
    """fuse every q|k|v / gate|up projection of `model` NOW (decoder layers, and the CLIP / Whisper
    attention modules when they are on the device) instead of lazily at the first forward.  Call it
    -- or pass model= to bucketed.BucketedStep -- before parameters are handed to an optimizer
    runtime that re-homes them.  Idempotent."""
    for mod in model.modules():
        if isinstance(mod, LlamaDecoderLayer):
            mod.fuse_projections()
        elif (all(hasattr(mod, n) for n in ("q_proj", "k_proj", "v_proj", "out_proj"))
              and isinstance(getattr(mod, "q_proj"), nn.Linear) and mod.q_proj.weight.is_cuda):
            fused_encoder_qkv(mod)
    return model


# ---------------------------------------------------------------- LLaMA -----
class LlamaRotaryEmbedding(nn.Module):
    """modeling.py:94-123.  Tables are built once in fp32 and cast to the activation dtype."""

    def __init__(self, dim, max_position_embeddings=2048, base=10000, device=None):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float().to(device) / dim))
        self.register_buffer("inv_freq", inv_freq)
        self.dim, self.base = dim, base
        self.max_seq_len_cached = max_position_embeddings
        self._tables = {}

    def tables(self, seq_len, dtype, device):
        if seq_len > self.max_seq_len_cached:
            self.max_seq_len_cached = seq_len
            self._tables = {}
        key = (dtype, str(device))
        if key not in self._tables:
            inv = 1.0 / (self.base ** (torch.arange(0, self.dim, 2).float() / self.dim))
            t = torch.arange(self.max_seq_len_cached, dtype=inv.dtype)
            emb = torch.cat((torch.einsum("i,j->ij", t, inv),) * 2, dim=-1)
            cos, sin = emb.cos().to(device), emb.sin().to(device)  # init-time only
            self._tables[key] = (ops.cast(cos, dtype), ops.cast(sin, dtype))
        return self._tables[key]


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        _dev_check(hidden_states)
        return eng.RMSNormFn.apply(hidden_states, self.weight, self.variance_epsilon)


class LlamaMLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str):
        super().__init__()
        if hidden_act != "silu":
            raise ValueError("LlamaMLP: only hidden_act='silu' is implemented")
        self.gate_proj = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.down_proj = nn.Linear(intermediate_size, hidden_size, bias=False)
        self.up_proj = nn.Linear(hidden_size, intermediate_size, bias=False)


class LlamaAttention(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.max_position_embeddings = config.max_position_embeddings
        if (self.head_dim * self.num_heads) != self.hidden_size:
            raise ValueError(
                f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                f" and `num_heads`: {self.num_heads}).")
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.k_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.v_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)
        self.rotary_emb = LlamaRotaryEmbedding(self.head_dim,
                                               max_position_embeddings=self.max_position_embeddings)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = LlamaAttention(config=config)
        self.mlp = LlamaMLP(hidden_size=self.hidden_size, intermediate_size=config.intermediate_size,
                            hidden_act=config.hidden_act)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    # ---- optional fused storage for q|k|v and gate|up ------------------------------------
    @torch.no_grad()
    def fuse_projections(self):
        """Re-home q/k/v (and gate/up) weights in ONE contiguous [3D, D] ([2FF, D]) buffer each;
        the nn.Linear parameters become row-slice views of it, so state-dict keys, optimizers
        and gradients are unchanged while the engine runs one GEMM instead of three (two)."""
        a, m = self.self_attn, self.mlp
        for mods in ((a.q_proj, a.k_proj, a.v_proj), (m.gate_proj, m.up_proj)):
            ws = [x.weight for x in mods]
            if _rows_view(ws) is None:
                _rehome_rows(ws)
        return self

    @staticmethod
    def _fused_view(ws):
        """[sum rows, D] tensor aliasing the parameters if they are laid out back to back"""
        return _rows_view(ws)

    def fused_weights(self):
        """(wqkv, wgu) views; fuses lazily on the device (AUTO_FUSE) so that the reference's own
        construction path -- MM_LLMs(config) -> resize_token_embeddings -> .to(dtype / device),
        run_clm_llms.py:478-497 -- gets the one-GEMM projections too."""
        a, m = self.self_attn, self.mlp
        qkv = (a.q_proj.weight, a.k_proj.weight, a.v_proj.weight)
        gu = (m.gate_proj.weight, m.up_proj.weight)
        wqkv, wgu = self._fused_view(qkv), self._fused_view(gu)
        if (wqkv is None or wgu is None) and AUTO_FUSE and qkv[0].is_cuda:
            self.fuse_projections()
            wqkv, wgu = self._fused_view(qkv), self._fused_view(gu)
        return wqkv, wgu

    def forward(self, hidden_states, kmask=None, pos=None, past_key_value=None, use_cache=False,
                recompute=False):
        """hidden_states [B,S,D]; kmask int32 [B,S] (0 = padding) or None; pos int32 [B*S];
        recompute = activation checkpointing for this layer."""
        if past_key_value is not None or use_cache:
            raise NotImplementedError("KV-cache decode goes through LlamaForCausalLM.generate")
        a, m = self.self_attn, self.mlp
        wqkv, wgu = self.fused_weights()
        cos, sin = a.rotary_emb.tables(hidden_states.shape[1], hidden_states.dtype,
                                       hidden_states.device)
        out = eng.LlamaLayerFn.apply(
            hidden_states, kmask, pos, cos, sin, a.num_heads, self.input_layernorm.variance_epsilon,
            a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.o_proj.weight, m.gate_proj.weight,
            m.up_proj.weight, m.down_proj.weight, self.input_layernorm.weight,
            self.post_attention_layernorm.weight, wqkv, wgu, recompute)
        return (out,)


class LlamaPreTrainedModel(PreTrainedModel):
    config_class = LlamaConfig
    base_model_prefix = "model"
    supports_gradient_checkpointing = True
    _no_split_modules = ["LlamaDecoderLayer"]

    def _init_weights(self, module):
        std = self.config.initializer_range
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()

    def _set_gradient_checkpointing(self, module, value=False):
        if isinstance(module, LlamaModel):
            module.gradient_checkpointing = value


class LlamaModel(LlamaPreTrainedModel):
    def __init__(self, config: LlamaConfig):
        super().__init__(config)
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.gradient_checkpointing = False
        self._pos_cache = {}
        self.post_init()

    def get_input_embeddings(self):
        return self.embed_tokens

    def set_input_embeddings(self, value):
        self.embed_tokens = value

    def _positions(self, B, S, device):
        key = (B, S, str(device))
        if key not in self._pos_cache:
            self._pos_cache = {key: torch.arange(S, dtype=torch.int32, device=device).repeat(B)}
        return self._pos_cache[key]

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both decoder_input_ids and decoder_inputs_embeds at the same time")
        if input_ids is None and inputs_embeds is None:
            raise ValueError("You have to specify either decoder_input_ids or decoder_inputs_embeds")
        if output_attentions:
            raise NotImplementedError("output_attentions is not supported by the fused engine")
        if past_key_values is not None or use_cache:
            raise NotImplementedError("KV-cache decode goes through LlamaForCausalLM.generate")
        if inputs_embeds is None:
            _dev_check(input_ids)
            B, S = input_ids.shape
            inputs_embeds = EmbeddingFn.apply(self.embed_tokens.weight, input_ids.long().reshape(-1),
                                              self.padding_idx).view(B, S, -1)
        _dev_check(inputs_embeds)
        B, S, _ = inputs_embeds.shape
        dev = inputs_embeds.device
        if position_ids is None:  # modeling.py:434-439: arange(S) irrespective of padding
            pos = self._positions(B, S, dev)
        else:
            pos = position_ids.to(torch.int32).expand(B, S).reshape(-1).contiguous()
        kmask = None
        if attention_mask is not None:
            if attention_mask.shape != (B, S):
                raise ValueError(f"Attention mask should be of size {(B, S)}, but is {tuple(attention_mask.shape)}")
            kmask = attention_mask.to(torch.int32).contiguous()
        h = inputs_embeds
        all_h = () if output_hidden_states else None
        for layer in self.layers:
            if output_hidden_states:
                all_h += (h,)
            # modeling.py:474-489: checkpoint every decoder layer when training with the flag on
            h = layer(h, kmask=kmask, pos=pos,
                      recompute=self.gradient_checkpointing and self.training)[0]
        # NOTE: the final RMSNorm is fused with lm_head in LlamaForCausalLM; standalone
        # LlamaModel.forward applies it here.
        if getattr(self, "_defer_final_norm", False):
            return h
        h = self.norm(h)
        if output_hidden_states:
            all_h += (h,)
        return BaseModelOutputWithPast(last_hidden_state=h, past_key_values=None, hidden_states=all_h,
                                       attentions=None)


class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids, padding_idx):
        out = ops.embedding_fwd(table, ids)
        ctx.save_for_backward(ids)
        ctx.shape, ctx.padding_idx = table.shape, padding_idx
        ctx.dtype = table.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        dt = torch.empty(ctx.shape, dtype=ctx.dtype, device=dout.device)
        ops.fill_(dt, 0.0)
        ops.embedding_bwd_(dt, dout.contiguous(), ids,
                           -1 if ctx.padding_idx is None else ctx.padding_idx)
        return dt, None, None


class LlamaForCausalLM(LlamaPreTrainedModel):
    def __init__(self, config):
        super().__init__(config)
        self.model = LlamaModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.post_init()

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def set_output_embeddings(self, new_embeddings):
        self.lm_head = new_embeddings

    def set_decoder(self, decoder):
        self.model = decoder

    def get_decoder(self):
        return self.model

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        self.model._defer_final_norm = True
        try:
            h = self.model(input_ids=input_ids, attention_mask=attention_mask,
                           position_ids=position_ids, past_key_values=past_key_values,
                           inputs_embeds=inputs_embeds, use_cache=use_cache,
                           output_attentions=output_attentions)
        finally:
            self.model._defer_final_norm = False
        shift = None
        if labels is not None:
            # Shift so that tokens < n predict n (modeling.py:601-603): row (b,s) is scored
            # against labels[b, s+1]; the last position of every sample is ignored.
            labels = labels.to(h.device).long()
            shift = torch.cat([labels[:, 1:], torch.full_like(labels[:, :1], -100)], dim=1)
            shift = shift.reshape(-1).contiguous()
        loss, logits = eng.LMHeadLossFn.apply(h, self.model.norm.weight, self.lm_head.weight, shift,
                                              self.model.norm.variance_epsilon)
        loss = loss[0] if labels is not None else None
        if return_dict is False:
            return ((loss, logits) if loss is not None else (logits,))
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=None,
                                      hidden_states=None, attentions=None)

    @torch.no_grad()
    def _generate_graph(self, run, logits, last, emb_w, B, S0, max_new_tokens, eos, pad, dev):
        """Decode loop with ONE hipGraph launch per token.  Everything a step needs lives in device
        memory -- the position (int32 counter read by the fused RoPE + cache append + attention
        kernel), the last token ids, the per-sample finished flags and the output matrix (all advanced
        by ops.decode_emit) -- so the launches of a step (5 per layer + 3) are captured once, after
        one eager step that also serves as the warm-up, and replayed; the host only looks at the
        finished flags every few tokens."""
        state = torch.tensor([S0, 0, 0, 0], dtype=torch.int32, device=dev)   # position fed, output column, 0
        t_dev = state[:1]
        tok = torch.zeros(B, dtype=torch.long, device=dev)
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        out_buf = torch.full((B, max_new_tokens), pad, dtype=torch.long, device=dev)
        V = self.lm_head.weight.shape[0]

        def emit(h_last):                # argmax, pad / eos handling, output column, position += 1
            ops.decode_emit(logits(h_last), V, pad, eos, tok, done, out_buf, state)

        def step():
            emit(run(ops.embedding_fwd(emb_w, tok), 1, 0, pos=t_dev, t_dev=t_dev))

        emit(last)                       # token 0 (from the prefill) ...
        state[0] = S0                    # ... is the one fed at position S0
        emitted = 1
        if not bool(done.all()):
            step()                       # token 1: eager (kernel attributes, allocator pools, workspace)
            emitted += 1
        remaining = max_new_tokens - emitted
        if remaining > 0 and not bool(done.all()):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            for i in range(remaining):
                graph.replay()
                emitted += 1
                if (i & 7) == 7 and bool(done.all()):
                    break
            del graph
        res = out_buf[:, :emitted]
        # the eager loop stops right after the token at which every sample has finished
        fin = ((res == eos).cumsum(1) > 0).all(0)
        if bool(fin.any()):
            res = res[:, : int(torch.nonzero(fin)[0]) + 1]
        return res.clone()

    @torch.no_grad()
    def generate(self, inputs_embeds=None, input_ids=None, max_new_tokens=128, eos_token_id=2,
                 bos_token_id=1, pad_token_id=None, use_cache=True, decode_graph=True, **_):
        """Greedy decode — the only mode the reference uses (modeling.py:959:
        `llm.generate(inputs_embeds=…, max_new_tokens=128, eos_token_id=2, bos_token_id=1,
        pad_token_id=32006)`, no attention mask).  Prefill runs the prompt once and fills a
        preallocated per-layer KV cache [B, T_max, D]; every decode step runs one position
        through the layers against the cache (fused attention, Lq = 1) instead of the
        reference's per-step `torch.cat` of the cache (modeling.py:190-195).  Finished samples
        emit pad_token_id.  Returns the NEW token ids [B, <= max_new_tokens].
        use_cache=False recomputes the whole prefix each step (same ids; test reference)."""
        emb_w = self.model.embed_tokens.weight
        if inputs_embeds is None:
            inputs_embeds = ops.embedding_fwd(emb_w, input_ids.long().reshape(-1)).view(*input_ids.shape, -1)
        _dev_check(inputs_embeds)
        B, S0, D = inputs_embeds.shape
        dev, dtype = inputs_embeds.device, inputs_embeds.dtype
        pad = pad_token_id if pad_token_id is not None else (eos_token_id or 0)
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        eps = self.model.norm.variance_epsilon
        V = self.lm_head.weight.shape[0]
        out = []

        def logits(h_last):       # h_last [B, D] -> [B, V] (final norm folded into the lm_head stream)
            if h_last.is_contiguous() and ops.decode_linear_ok(h_last, self.lm_head.weight, 1):
                return ops.decode_linear(h_last, self.lm_head.weight, 1, self.model.norm.weight, eps)
            _, y, _ = ops.rmsnorm_fwd(h_last, self.model.norm.weight, eps)
            return ops.linear_fwd(y, self.lm_head.weight)

        def select(h_last):       # h_last [B, D] -> next token ids [B]
            return ops.argmax_rows(logits(h_last), V)

        if not use_cache:
            emb = inputs_embeds.contiguous()
            for _ in range(max_new_tokens):
                S = emb.shape[1]
                self.model._defer_final_norm = True
                try:
                    h = self.model(inputs_embeds=emb)
                finally:
                    self.model._defer_final_norm = False
                last = torch.empty((B, D), dtype=dtype, device=dev)
                ops.copy2d(h.contiguous(), last, 1, D, D, D, batch=B, s_src=S * D, s_dst=D,
                           src_off=(S - 1) * D)
                nxt = torch.where(done, torch.full((B,), pad, dtype=torch.long, device=dev), select(last))
                out.append(nxt)
                done = done | (nxt == eos_token_id)
                if bool(done.all()):
                    break
                nemb = torch.empty((B, S + 1, D), dtype=dtype, device=dev)
                ops.copy2d(emb, nemb, S, D, D, D, batch=B, s_src=S * D, s_dst=(S + 1) * D)
                ops.copy2d(ops.embedding_fwd(emb_w, nxt.contiguous()), nemb, 1, D, D, D, batch=B,
                           s_src=D, s_dst=(S + 1) * D, dst_off=S * D)
                emb = nemb
            return torch.stack(out, dim=1)

        Tmax = S0 + max_new_tokens
        layers = self.model.layers
        rot = layers[0].self_attn.rotary_emb
        cos, sin = rot.tables(Tmax, dtype, dev)
        kvc = [torch.empty((B, Tmax, 2 * D), dtype=dtype, device=dev) for _ in layers]   # [keys | values]

        def run(x2, Sn, t0, pos=None, t_dev=None):
            if pos is None:
                pos = (torch.arange(t0, t0 + Sn, dtype=torch.int32, device=dev)).repeat(B)
            for i, lyr in enumerate(layers):
                a, m = lyr.self_attn, lyr.mlp
                x2 = eng.llama_layer_cached(
                    x2, B, Sn, t0, kvc[i], Tmax, pos, cos, sin, a.num_heads,
                    lyr.input_layernorm.variance_epsilon, a.q_proj.weight, a.k_proj.weight,
                    a.v_proj.weight, a.o_proj.weight, m.gate_proj.weight, m.up_proj.weight,
                    m.down_proj.weight, lyr.input_layernorm.weight,
                    lyr.post_attention_layernorm.weight, *lyr.fused_weights(), t_dev=t_dev)
            return x2

        h = run(eng._c2(inputs_embeds, B * S0, D), S0, 0)                 # prefill
        last = torch.empty((B, D), dtype=dtype, device=dev)
        ops.copy2d(h, last, 1, D, D, D, batch=B, s_src=S0 * D, s_dst=D, src_off=(S0 - 1) * D)
        hd = D // layers[0].self_attn.num_heads
        if (decode_graph and max_new_tokens > 2 and ops.decode_attn_ok(dtype, hd, Tmax)
                and not os.environ.get("MACAW_NO_DECODE_GRAPH")):
            return self._generate_graph(run, logits, last, emb_w, B, S0, max_new_tokens, eos_token_id, pad, dev)
        for t in range(max_new_tokens):
            nxt = torch.where(done, torch.full((B,), pad, dtype=torch.long, device=dev), select(last))
            out.append(nxt)
            done = done | (nxt == eos_token_id)
            if t + 1 == max_new_tokens or bool(done.all()):
                break
            x = ops.embedding_fwd(emb_w, nxt.contiguous())                 # [B, D]
            last = run(x, 1, S0 + t)                                       # decode step
        return torch.stack(out, dim=1)


# ---------------------------------------------------------- multimodal ------
class MM_LLMs_Config(PretrainedConfig):
    model_type = "mm_llms"
    is_composition = True

    def __init__(self, n_frames=6, attention_heads=8, image_conv_kernel=48, image_conv_stride=36,
                 video_conv_kernel=36, video_conv_stride=30, audio_conv_kernel=240,
                 audio_conv_stride=220, clip_config=None, whisper_config=None, llm_config=None,
                 **kwargs):
        # transformers >= 4.3x instantiates `cls()` with no arguments inside save_pretrained /
        # to_diff_dict; the reference class would raise there.  Default sub-configs keep
        # checkpoint round trips (run_clm_llms_inference.py:455) working.
        clip_config = CLIPConfig() if clip_config is None else clip_config
        whisper_config = WhisperConfig() if whisper_config is None else whisper_config
        llm_config = LlamaConfig() if llm_config is None else llm_config
        for name, sub, klass in (("clip_config", clip_config, CLIPConfig),
                                 ("whisper_config", whisper_config, WhisperConfig),
                                 ("llm_config", llm_config, LlamaConfig)):
            if isinstance(sub, dict):
                sub = klass.from_dict(sub)
                if name == "clip_config":
                    clip_config = sub
                elif name == "whisper_config":
                    whisper_config = sub
                else:
                    llm_config = sub
        self.image_config = clip_config
        self.audio_config = whisper_config
        self.llm_config = llm_config
        self.n_frames = n_frames
        self.attention_heads = attention_heads
        self.image_conv_kernel = image_conv_kernel
        self.image_conv_stride = image_conv_stride
        self.video_conv_kernel = video_conv_kernel
        self.video_conv_stride = video_conv_stride
        self.audio_conv_kernel = audio_conv_kernel
        self.audio_conv_stride = audio_conv_stride
        self.hidden_size = max(llm_config.hidden_size, clip_config.projection_dim,
                               whisper_config.d_model, clip_config.projection_dim)
        super().__init__(**kwargs)

    def to_dict(self):
        output = copy.deepcopy(self.__dict__)
        output["image_config"] = self.image_config.to_dict()
        output["audio_config"] = self.audio_config.to_dict()
        output["llm_config"] = self.llm_config.to_dict()
        for k in ("n_frames", "attention_heads", "image_conv_kernel", "image_conv_stride",
                  "video_conv_kernel", "video_conv_stride", "audio_conv_kernel", "audio_conv_stride",
                  "hidden_size"):
            output[k] = getattr(self, k)
        output["model_type"] = self.__class__.model_type
        return output

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        config_dict, kwargs = cls.get_config_dict(pretrained_model_name_or_path, **kwargs)
        clip_config = CLIPConfig.from_dict(config_dict["image_config"])
        whisper_config = WhisperConfig.from_dict(config_dict["audio_config"])
        llm_config = LlamaConfig.from_dict(config_dict["llm_config"])
        return cls(clip_config=clip_config, whisper_config=whisper_config, llm_config=llm_config,
                   **kwargs)


def build_prefix_layout(inputs: dict, lq: dict):
    """Integer plumbing of modeling.py:977-1046 (bit-exact by construction, no float math).

    lq maps each PRESENT modality to its number of aligned prefix tokens.  Returns
      ids_full [B,S] int64 : token id at every position of the spliced sequence, -1 where the
                             aligned modal features go.  Final order:
                             [BOS][<image> f.. </image>][<audio> f.. </audio>][<video> f.. </video>][text 1:]
      slots {name: (first feature position, Lq)}
      attention_mask       : ones for the whole prefix PREPENDED to the text mask   (:1036-1040)
      labels               : -100 for the whole prefix PREPENDED to the text labels (:1042-1046)
    Unlike the reference this keeps integer dtypes when no modality is present (the reference's
    empty `torch.tensor([])` is float and breaks the loss; tests/test_oracle.py)."""
    ids = inputs["input_ids"].long()
    B = ids.shape[0]
    cols = [ids[:, :1]]
    slots, pos, ignore = {}, 1, 0
    for name in eng.MODALITIES:
        if name not in lq:
            continue
        n = lq[name]
        cols += [inputs[f"{name}_starts"].long().view(B, 1),
                 torch.full((B, n), -1, dtype=torch.long, device=ids.device),
                 inputs[f"{name}_ends"].long().view(B, 1)]
        slots[name] = (pos + 1, n)
        pos += n + 2
        ignore += n + 2
    cols.append(ids[:, 1:])
    ids_full = torch.cat(cols, dim=1).contiguous()
    attention_mask = labels = None
    if inputs.get("attention_mask") is not None:
        am = inputs["attention_mask"]
        attention_mask = torch.cat([torch.ones((B, ignore), dtype=am.dtype, device=am.device), am], dim=1)
    if inputs.get("labels") is not None:
        lb = inputs["labels"]
        labels = torch.cat([torch.full((B, ignore), -100, dtype=lb.dtype, device=lb.device), lb], dim=1)
    return ids_full, slots, attention_mask, labels


def _mha_params(m: nn.MultiheadAttention):
    return (m.in_proj_weight, m.in_proj_bias, m.bias_k, m.bias_v, m.out_proj.weight, m.out_proj.bias)


class MM_LLMs(PreTrainedModel):
    config_class = MM_LLMs_Config
    supports_gradient_checkpointing = True

    def __init__(self, config):
        super().__init__(config)
        self.config = config
        self.temporal_position_embeddings = nn.Embedding(config.n_frames, config.image_config.projection_dim)
        self.image_encoder = CLIPModel(config.image_config)
        self.video_encoder = CLIPModel(config.image_config)
        self.audio_encoder = WhisperModel(config.audio_config)
        self.llm = LlamaForCausalLM(config.llm_config)
        attn_dropout, kv, za = 0.1, True, True
        pd, D = config.image_config.projection_dim, config.llm_config.hidden_size
        mk = lambda e, h: nn.MultiheadAttention(e, h, dropout=attn_dropout, add_bias_kv=kv,  # noqa: E731
                                                add_zero_attn=za)
        self.temporal_self_attention = mk(pd, config.attention_heads)
        self.video_align_attention = mk(D, config.attention_heads * 2)
        self.audio_align_attention = mk(D, config.attention_heads * 2)
        self.image_align_attention = mk(D, config.attention_heads * 2)
        self.video_long_self_attention = mk(pd, config.attention_heads)
        self.transform_video_to_hidden = nn.Linear(pd, D)
        self.transform_audio_to_hidden = nn.Linear(config.audio_config.d_model, D)
        self.transform_image_to_hidden = nn.Linear(pd, D)
        self.project_image = nn.Conv1d(pd, pd, kernel_size=config.image_conv_kernel,
                                       stride=config.image_conv_stride)
        self.project_video = nn.Conv1d(pd, pd, kernel_size=config.video_conv_kernel,
                                       stride=config.video_conv_stride)
        self.project_audio = nn.Conv1d(config.audio_config.d_model, config.audio_config.d_model,
                                       kernel_size=config.audio_conv_kernel,
                                       stride=config.audio_conv_stride)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.layer_norm = nn.LayerNorm(pd)
        self.softmax = nn.Softmax(dim=-1)
        self.relu = nn.ReLU()
        self.gelu = nn.GELU()
        self.elu = nn.ELU()
        self.sigmoid = nn.Sigmoid()
        self.loss_fct = CrossEntropyLoss()
        self._pe_cache = {}
        self._step = 0
        self.dropout_seed = 0x5EED
        self.post_init()

    SEED_STRIDE = 64        # seeds of one step: base + 64 * step + module slot (slot < 64)

    def _dropout_seed(self, slot):
        """seed of dropout site `slot` in training step self._step.  One stride per step, so a step
        replayed from a hipGraph gets the same seeds from base + (device counter += SEED_STRIDE)
        (train.GraphedStep, ops.set_dropout_seed_offset)."""
        return (self.dropout_seed * 1000003 + self._step * self.SEED_STRIDE + slot) & 0x7FFFFFFFFFFF

    # ------------------------------------------------------------ forward ---
    def forward(self, inputs=None):
        _dtype_check(self._param_dtype())
        text_embeddings, attention_mask, labels = self.prepare_inputs_for_generation(inputs)
        if "inference" in inputs and inputs["inference"] is True:
            return self.llm.generate(inputs_embeds=text_embeddings, max_new_tokens=128,
                                     eos_token_id=2, bos_token_id=1, pad_token_id=32006)
        return self.llm(inputs_embeds=text_embeddings, attention_mask=attention_mask, labels=labels)

    @staticmethod
    def set_fp8(qkv: bool = True, align: bool = True, mlp: bool = False):
        """BASELINE cfg 5: run the forward AND grad-input GEMMs of the fused q|k|v projections
        (modeling.py:159-162) and of the alignment K/V projection of the token table (:882-910) on
        the fp8 MFMA path: e4m3, one scale per token row for activations / gradients, one per
        channel for the weights, which are quantised once per optimizer step (engine.FP8).
        mlp=True extends it to gate|up / down (not part of cfg 5's wording).  Needs bf16
        parameters and fused projections; layers that do not qualify keep the bf16 GEMM.
        Process-wide switch."""
        eng.FP8["qkv"], eng.FP8["align"], eng.FP8["mlp"] = bool(qkv), bool(align), bool(mlp)
        ops.clear_fp8_cache()

    def prepare_inputs_for_generation(self, inputs):
        """modeling.py:965-1048 — same outputs (inputs_embeds, attention_mask, labels)."""
        cfg = self.config
        audio_f, audio_side = None, None
        if inputs.get("audios") is not None:
            # The towers are independent of each other: with frozen encoders (run_clm_llms.py:390-393) the audio tower
            # goes out on a second stream beside the image / video tower (engine.ENC_SIDE; their short-K GEMMs and 4-wave
            # attention leave CUs idle between rounds: +0.4 % of the cfg-3 step)
            audio_side = eng.tower_side_stream(inputs["audios"], self.audio_encoder,
                                               other=inputs.get("images") is not None or inputs.get("videos") is not None)
            if audio_side is not None:
                audio_side.wait_stream(torch.cuda.current_stream(inputs["audios"].device))
                with torch.cuda.stream(audio_side):
                    audio_f = self.encode_audio(inputs["audios"])
            else:
                audio_f = self.encode_audio(inputs["audios"])
        image_f = self.encode_image(inputs["images"]) if inputs.get("images") is not None else None
        video_f = self.encode_video_long(inputs["videos"]) if inputs.get("videos") is not None else None
        if audio_side is not None:
            main = torch.cuda.current_stream(audio_f.device)
            main.wait_stream(audio_side)
            audio_f.record_stream(main)
        E = self.llm.model.embed_tokens.weight
        ids = inputs["input_ids"]
        _dev_check(ids)
        ids = ids.long()
        B, L = ids.shape
        feats = dict(image=image_f, audio=audio_f, video=video_f)
        geom = dict(image=(cfg.image_conv_kernel, cfg.image_conv_stride),
                    audio=(cfg.audio_conv_kernel, cfg.audio_conv_stride),
                    video=(cfg.video_conv_kernel, cfg.video_conv_stride))
        lq = {}
        for name in eng.MODALITIES:
            if feats[name] is not None:
                kw, st = geom[name]
                lq[name] = (feats[name].shape[1] - kw) // st + 1
        ids_full, slots, attention_mask, labels = build_prefix_layout(inputs, lq)
        self._step += 1
        p = 0.1 if self.training else 0.0
        meta = dict(ids_full=ids_full, slots=slots, heads=cfg.attention_heads * 2, geom=geom, p=p,
                    seeds={n: self._dropout_seed(i) for i, n in enumerate(eng.MODALITIES)},
                    padding_idx=-1 if self.llm.model.padding_idx is None else self.llm.model.padding_idx)
        params = []
        for name in eng.MODALITIES:
            conv = getattr(self, f"project_{name}")
            lin = getattr(self, f"transform_{name}_to_hidden")
            params += [conv.weight, conv.bias, lin.weight, lin.bias,
                       *_mha_params(getattr(self, f"{name}_align_attention"))]
        text_embeddings = eng.PrefixAssembleFn.apply(E, meta, image_f, audio_f, video_f, *params)

        return text_embeddings, attention_mask, labels

    # ----------------------------------------------------------- encoders ---
    def _param_dtype(self):
        return self.llm.model.embed_tokens.weight.dtype

    def _clip_tokens(self, clip: CLIPModel, images):
        """visual_projection(vision_model(x)[0])[:, 1:, :]  (modeling.py:1073,1092)."""
        _dev_check(images)
        vm = clip.vision_model
        vcfg = clip.config.vision_config
        dtype = vm.embeddings.patch_embedding.weight.dtype
        x = ops.cast(images.contiguous(), dtype)
        B = x.shape[0]
        P, Ed = vcfg.patch_size, vcfg.hidden_size
        g = vcfg.image_size // P
        T = g * g + 1
        if x.shape[-1] != vcfg.image_size or x.shape[-2] != vcfg.image_size:
            raise ValueError(f"Input image size ({x.shape[-2]}*{x.shape[-1]}) doesn't match model "
                             f"({vcfg.image_size}*{vcfg.image_size}).")
        h = eng.ClipEmbedFn.apply(x, vm.embeddings.patch_embedding.weight,
                                  vm.embeddings.class_embedding,
                                  vm.embeddings.position_embedding.weight, P)
        h = eng.LayerNormFn.apply(h, vm.pre_layrnorm.weight, vm.pre_layrnorm.bias, vcfg.layer_norm_eps)
        act = eng.ACT_CODE[vcfg.hidden_act]
        for lyr in vm.encoder.layers:
            a, m = lyr.self_attn, lyr.mlp
            h = eng.EncoderLayerFn.apply(
                h, vcfg.num_attention_heads, vcfg.layer_norm_eps, act, lyr.layer_norm1.weight,
                lyr.layer_norm1.bias, a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, a.k_proj.bias,
                a.v_proj.weight, a.v_proj.bias, a.out_proj.weight, a.out_proj.bias,
                lyr.layer_norm2.weight, lyr.layer_norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight,
                m.fc2.bias, *fused_encoder_qkv(a))
        return eng.DropClsProjectFn.apply(h, clip.visual_projection.weight)

    def encode_image(self, images):
        return self._clip_tokens(self.image_encoder, images)

    def encode_audio(self, audios):
        """audio_encoder.encoder(audios)[0]  (modeling.py:1081-1083)."""
        _dev_check(audios)
        enc = self.audio_encoder.encoder
        wcfg = self.audio_encoder.config
        dtype = enc.conv1.weight.dtype
        x = ops.cast(audios.contiguous(), dtype)
        expected = wcfg.max_source_positions * 2
        if x.shape[-1] != expected:
            raise ValueError(f"Whisper expects the mel input features to be of length {expected}, "
                             f"but found {x.shape[-1]}.")
        h = eng.WhisperStemFn.apply(x, enc.conv1.weight, enc.conv1.bias, enc.conv2.weight,
                                    enc.conv2.bias, enc.embed_positions.weight)
        act = eng.ACT_CODE[wcfg.activation_function]
        for lyr in enc.layers:
            a = lyr.self_attn
            h = eng.EncoderLayerFn.apply(
                h, wcfg.encoder_attention_heads, 1e-5, act, lyr.self_attn_layer_norm.weight,
                lyr.self_attn_layer_norm.bias, a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, None,
                a.v_proj.weight, a.v_proj.bias, a.out_proj.weight, a.out_proj.bias,
                lyr.final_layer_norm.weight, lyr.final_layer_norm.bias, lyr.fc1.weight, lyr.fc1.bias,
                lyr.fc2.weight, lyr.fc2.bias, *fused_encoder_qkv(a))
        return eng.LayerNormFn.apply(h, enc.layer_norm.weight, enc.layer_norm.bias, 1e-5)

    def encode_video_long(self, videos):
        """modeling.py:1070-1079."""
        _dev_check(videos)
        nf = self.config.n_frames
        frames = videos.reshape(-1, videos.size(-3), videos.size(-2), videos.size(-1))
        f = self._clip_tokens(self.video_encoder, frames)           # [B*nf, T, pd]
        Bv = frames.size(0) // nf
        f = f.reshape(Bv, nf * f.size(1), f.size(2))
        pe = self._positional_encoding(f.size(1), f.size(2), f.dtype, f.device)
        f = eng.AddBroadcastFn.apply(f, pe)
        m = self.video_long_self_attention
        p = m.dropout if self.training else 0.0
        return eng.MHASelfFn.apply(f, m.num_heads, p, self._dropout_seed(40), *_mha_params(m))

    def _positional_encoding(self, L, h, dtype, device):
        """create_positional_encoding (modeling.py:1095-1106) is input independent: built once
        per (L, h) instead of by a 590k-iteration Python loop every forward."""
        key = (L, h, dtype, str(device))
        if key not in self._pe_cache:
            self._pe_cache[key] = ops.cast(create_positional_encoding(L, h).to(device), dtype)
        return self._pe_cache[key]

    def encode_video(self, videos):
        raise NotImplementedError("encode_video is dead code in the reference (modeling.py:969 uses "
                                  "encode_video_long); not on the hot path")


def create_positional_encoding(L, h):
    """modeling.py:1095-1106, vectorised (the reference fills the [L, h] table with a Python double loop):
    pe[pos, i] = sin(pos * w_i), pe[pos, i + 1] = cos(pos * w_i), w_i = exp(-(ln 10000 / h) * 2 i) for EVEN i --
    the reference's exponent uses 2 i with i already stepping by 2, kept as it is.  fp32 on
    the CPU like the reference's; pinned to it by tests/test_oracle.py::test_reference_positional_encoding."""
    i = torch.arange(0, h, 2, dtype=torch.float32)
    div = torch.exp(-(math.log(10000.0) / h * (2 * i)))
    posn = torch.arange(L, dtype=torch.float32)[:, None]
    pe = torch.zeros(L, h)
    pe[:, 0::2] = torch.sin(posn * div)
    pe[:, 1::2] = torch.cos(posn * div)
    return pe


def add_positional_encoding(tensor):
    """modeling.py:1108-1118 on the device path."""
    N, L, h = tensor.size()
    pe = create_positional_encoding(L, h)
    return eng.AddBroadcastFn.apply(tensor, ops.cast(pe.to(tensor.device), tensor.dtype))

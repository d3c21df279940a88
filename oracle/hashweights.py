"""TEST INFRASTRUCTURE ONLY. Device-independent, version-independent weights and inputs for the
FULL-SIZE reference fixtures (oracle/make_golden_cfg1.py -> tests/golden/cfg1_full.pt, real_av_trunc.pt, real_grad_trunc.pt).

The reference at BASELINE cfg 1 (CLIP-L/14 + alignment + 32-layer LLaMA-7B, fp32) has 8 B parameters:
32 GB cannot be committed, and `torch.randn` streams differ between CPU and GPU generators.  Every
tensor is therefore a pure function of (parameter name, shape) through an INTEGER hash evaluated in
int64 without overflow (all intermediates < 2^59), so the build container (where the reference runs),
the CPU restatement test and the GPU box regenerate bit-identical tensors:

    x  = (flat_index + crc32(name)) mod 2^32
    x  = two rounds of  x = ((x ^ (x >> 16)) * 0x45d9f3b) mod 2^32 ;  x ^= x >> 16      (integer mixing)
    k  = x >> 24                                   8-bit level, 0 .. 255
    w  = (k - 128) * step                          step a power of two  => every value bf16-EXACT

8-bit levels keep all values exactly representable in bf16, so the fp32 reference, the fp32 engine and
the bf16 engine see the SAME weights (no weight-rounding term in the bf16 comparison).  Standard
deviation of (k-128)/256 is 0.2887: matrices use step 2^-12 (sigma 0.018, HF init is 0.02), norm
weights 1 + (k>>4 - 8)/64, inputs step 2^-6 (sigma 1.15).
"""
from __future__ import annotations

import zlib
from collections.abc import Mapping

import torch

_M32 = 0xFFFFFFFF
_CHUNK = 1 << 18          # cache-resident int64 temporaries: 220 M values/s on 8 cores (16 M-chunks: 25)


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode()) & _M32


def hash_levels(n: int, seed: int, device=None, start: int = 0) -> torch.Tensor:
    """int64 tensor of n 8-bit levels (0..255) for flat indices start .. start+n-1."""
    x = (torch.arange(start, start + n, dtype=torch.int64, device=device) + int(seed)) & _M32
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & _M32
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & _M32
    x = x ^ (x >> 16)
    return x >> 24


def _filled(shape, seed, device, fn):
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=torch.float32, device=device)
    chunk = _CHUNK if out.device.type == "cpu" else _CHUNK << 6     # values do not depend on the chunking
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        out[a:a + m] = fn(hash_levels(m, seed, device, a))
    return out.view(*shape)


def kind_of(name: str, shape) -> str:
    """which value law a reference parameter gets: LayerNorm / RMSNorm scales sit around 1, the rest around 0"""
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "weight" and len(shape) == 1 and "norm" in name:      # incl. CLIP's 'pre_layrnorm'
        return "norm"
    return "matrix"


def hash_tensor(name: str, shape, device=None) -> torch.Tensor:
    """fp32 tensor for reference parameter `name` (bf16-exact values; see the module docstring)."""
    seed = name_seed(name)
    if kind_of(name, shape) == "norm":
        return _filled(shape, seed, device, lambda k: ((k >> 4) - 8).to(torch.float32) * (1.0 / 64) + 1.0)
    return _filled(shape, seed, device, lambda k: (k - 128).to(torch.float32) * (1.0 / 4096))


def hash_input(name: str, shape, device=None, step: float = 1.0 / 64) -> torch.Tensor:
    """synthetic float input (pixels, log-mel): levels * step, sigma = 73.9 * step"""
    return _filled(shape, name_seed("input:" + name), device, lambda k: (k - 128).to(torch.float32) * step)


def hash_ids(name: str, shape, lo: int, hi: int, device=None) -> torch.Tensor:
    """int64 ids uniform-ish in [lo, hi)"""
    n = 1
    for s in shape:
        n *= int(s)
    x = (torch.arange(n, dtype=torch.int64, device=device) + name_seed("ids:" + name)) & _M32
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & _M32
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & _M32
    x = x ^ (x >> 16)
    return (lo + x % (hi - lo)).view(*shape)


class HashState(Mapping):
    """A state dict whose tensors are generated on access and not kept: `restate.mm_forward(HashState(...))`
    streams a 7B model through a few hundred MB.  `shapes` = {reference key: shape} (stored in the fixture,
    taken from the reference's own state_dict, so the key set is pinned to the reference too)."""

    def __init__(self, shapes: dict, device=None, dtype=torch.float32, keep=()):
        self.shapes, self.device, self.dtype = dict(shapes), device, dtype
        self._keep = {k: None for k in keep}

    def __getitem__(self, key):
        if key in self._keep and self._keep[key] is not None:
            return self._keep[key]
        t = hash_tensor(key, self.shapes[key], self.device).to(self.dtype)
        if key in self._keep:
            self._keep[key] = t
        return t

    def __iter__(self):
        return iter(self.shapes)

    def __len__(self):
        return len(self.shapes)


def make_inputs(cfg: dict, batch: int, text_len: int, modalities, tag: str, device=None, n_prompt: int = 32,
                dtype=torch.float32) -> dict:
    """the input dict of MM_LLMs.forward (llm_trainer.py:365-378) with hash-generated contents"""
    v, w, tags = cfg["clip"]["vision_config"], cfg["whisper"], cfg["tags"]
    img = v["image_size"]
    out = dict(images=None, audios=None, videos=None)
    if "images" in modalities:
        out["images"] = hash_input(tag + ".images", (batch, 3, img, img), device).to(dtype)
    if "audios" in modalities:
        out["audios"] = hash_input(tag + ".audios", (batch, w["num_mel_bins"], w["max_source_positions"] * 2),
                                   device, step=1.0 / 128).to(dtype)
    if "videos" in modalities:
        out["videos"] = hash_input(tag + ".videos", (batch, cfg["mm"]["n_frames"], 3, img, img), device).to(dtype)
    ids = hash_ids(tag + ".input_ids", (batch, text_len), 3, tags["image"][0], device)
    ids[:, 0] = 1
    labels = ids.clone()
    labels[:, :n_prompt] = -100
    out.update(input_ids=ids, attention_mask=torch.ones(batch, text_len, dtype=torch.int64, device=device),
               labels=labels)
    for name in ("image", "audio", "video"):
        s, e = tags[name]
        out[f"{name}_starts"] = torch.full((batch,), s, dtype=torch.int32, device=device)
        out[f"{name}_ends"] = torch.full((batch,), e, dtype=torch.int32, device=device)
    return out

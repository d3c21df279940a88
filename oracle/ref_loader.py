"""TEST INFRASTRUCTURE ONLY — never imported by the product path.

Imports the reference's own `modeling.py` from /root/reference *unmodified*, through the
3-part in-memory shim described in SURVEY.md §8(c) (needed because this container has
transformers 5.x / torch 2.10 rather than the pinned 4.29 / 2.0):

  1. `transformers.modeling_utils.PretrainedConfig` (moved)      -> modeling.py:25
  2. `transformers.models.clip.modeling_clip.CLIPVisionTransformer` (renamed, unused) -> :39
  3. `MM_LLMs.init_weights` needs `post_init()` first under 5.x  -> modeling.py:939

Only usable where /root/reference exists (this build container).  On the GPU box the
tests use the committed fixtures in tests/golden/ instead.
"""
from __future__ import annotations

import os
import sys

REFERENCE_DIR = os.environ.get("MACAW_REFERENCE_DIR", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "modeling.py"))


_ref_module = None


def load_reference_modeling():
    """Return the reference's `modeling` module (shimmed import, no file edits)."""
    global _ref_module
    if _ref_module is not None:
        return _ref_module
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_DIR}")
    import importlib.util

    import transformers
    import transformers.modeling_utils as mu
    import transformers.models.clip.modeling_clip as mc

    if not hasattr(mu, "PretrainedConfig"):
        mu.PretrainedConfig = transformers.PretrainedConfig
    if not hasattr(mc, "CLIPVisionTransformer"):
        mc.CLIPVisionTransformer = mc.CLIPVisionModel
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True  # never write into the read-only reference tree
    try:
        spec = importlib.util.spec_from_file_location(
            "macaw_reference_modeling", os.path.join(REFERENCE_DIR, "modeling.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["macaw_reference_modeling"] = mod
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old

    _orig = transformers.PreTrainedModel.init_weights

    def _init_weights(self):
        if not hasattr(self, "all_tied_weights_keys"):
            return transformers.PreTrainedModel.post_init(self)
        return _orig(self)

    mod.MM_LLMs.init_weights = _init_weights
    _ref_module = mod
    return mod


def build_reference_model(cfg: dict, seed: int = 1234):
    """Construct the reference MM_LLMs for a config dict produced by oracle.configs."""
    import torch
    from transformers import CLIPConfig, LlamaConfig, WhisperConfig

    mod = load_reference_modeling()
    clip = CLIPConfig(**cfg["clip"])
    whisper = WhisperConfig(**cfg["whisper"])
    llama = LlamaConfig(**cfg["llama"])
    for c in (clip, clip.vision_config, clip.text_config, whisper, llama):
        c._attn_implementation = "eager"
    mm = mod.MM_LLMs_Config(clip_config=clip, whisper_config=whisper, llm_config=llama,
                            **cfg["mm"])
    torch.manual_seed(seed)
    model = mod.MM_LLMs(mm)
    model.eval()
    return model

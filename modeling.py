"""Shim so that the reference drivers' `from modeling import MM_LLMs, MM_LLMs_Config`
(run_clm_llms.py:95, run_clm_llms_inference.py, llm_trainer.py:114) resolve to the
MI355X-native implementation unchanged."""
from macaw_llm_amd.modeling import *  # noqa: F401,F403
from macaw_llm_amd.modeling import (LlamaAttention, LlamaDecoderLayer, LlamaForCausalLM,  # noqa: F401
                                    LlamaMLP, LlamaModel, LlamaPreTrainedModel, LlamaRMSNorm,
                                    LlamaRotaryEmbedding, MM_LLMs, MM_LLMs_Config,
                                    add_positional_encoding)

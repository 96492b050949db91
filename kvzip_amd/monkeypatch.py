"""Swap the attention forward of the supported HF model families (reference model/monkeypatch.py:5-22)."""
from __future__ import annotations

from .attn import llama_qwen_attn_forward, register_attention_interface


def replace_attn(model_id: str = "") -> None:
    """Patch Llama / Qwen2 / Qwen3 attention classes (the families of the reference's EvictCache path;
    Gemma3 and the QServe W8A8KV4 model are retain-only in the reference and out of scope here)."""
    import transformers
    register_attention_interface()
    mid = model_id.lower()
    if "gemma" in mid or "w8a8kv4" in mid:
        raise NotImplementedError(f"{model_id}: retain-only model family, outside the eviction hot path")
    transformers.models.llama.modeling_llama.LlamaAttention.forward = llama_qwen_attn_forward
    transformers.models.qwen2.modeling_qwen2.Qwen2Attention.forward = llama_qwen_attn_forward
    try:
        transformers.models.qwen3.modeling_qwen3.Qwen3Attention.forward = llama_qwen_attn_forward
    except AttributeError:  # transformers build without Qwen3
        pass

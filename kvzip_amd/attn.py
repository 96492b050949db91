"""Attention forward with the three cache hooks — counterpart of the reference's ``llama_qwen_attn_forward``
(reference attention/attn.py:19-96) written against transformers 5.x (``past_key_values`` keyword, attention
interface registry).  Hooks on the cache object, in this order (attention/attn.py:44-73):

    update()      append the new K,V                                   (always)
    _get_score()  KV importance of the current scoring chunk           (if cache.get_score)
    prepare()+attend()  variable-length attention over the pruned KV   (if cache.pruned)
    update_attend()     the three of them in one launch                (generation step on a pruned slack-layout cache)

The dense (pre-prune) attention — flash-attn's dense kernel in the reference (attention/attn.py:75-89) — runs on the
library's multi-row kernel (``ops.flash_fwd`` -> ``kvz_flash_fwd``), registered as the attention implementation
``"kvzip_hip"`` so that transformers skips its own mask construction (the kernel aligns the causal mask bottom-right).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def dense_causal_attention(module, query, key, value, attention_mask=None, dropout: float = 0.0,
                           scaling: Optional[float] = None, **kwargs):
    """[b, H, q, D] x [b, Hkv, k, D] -> ([b, q, H, D], None); causal mask aligned to the bottom-right corner
    (query i sees keys j <= i + k - q), like flash-attn's dense kernel used by the reference (attention/attn.py:75-89).
    Device tensors run on the library's own kernels (``kvz_flash_fwd``) straight off the dense cache views - and ONLY there: what
    those kernels do not take (fp32, head dims other than 64 / 128, batch > 1, sliding-window layers, dropout) raises instead of
    silently falling back to a generic implementation; CPU tensors raise as well."""
    k_len = key.shape[-2]
    window = kwargs.get("sliding_window", None) or getattr(module, "sliding_window", None)
    if window is not None and k_len > window:
        raise NotImplementedError("sliding-window attention layers (Gemma3 local layers) are outside this path: the dense kernel "
                                  "attends globally (reference attention/attn.py:99-190 handles them with a hybrid cache)")
    from . import ops
    if not query.is_cuda:
        raise ops.KvzError("dense attention runs on the library's HIP kernels only: the product path has no CPU implementation")
    if dropout:
        raise ops.KvzError("kvz_flash_fwd has no dropout (inference path)")
    if not (query.dtype in (torch.float16, torch.bfloat16) and query.shape[0] == 1 and query.shape[-1] in (64, 128)
            and key.shape[1] <= 64):
        raise ops.KvzError(f"dense attention on the device needs fp16 / bf16, batch 1, head_dim 64 or 128 and <= 64 KV heads "
                           f"(got {query.dtype}, batch {query.shape[0]}, head_dim {query.shape[-1]}, {key.shape[1]} KV heads): "
                           "there is no generic fallback in the product path")
    return ops.flash_fwd(query, key, value, causal=True, softmax_scale=scaling), None


def register_attention_interface():
    from transformers import AttentionInterface
    AttentionInterface.register("kvzip_hip", dense_causal_attention)


def llama_qwen_attn_forward(self, hidden_states: torch.Tensor,
                            position_embeddings: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                            attention_mask: Optional[torch.Tensor] = None, past_key_values=None, **kwargs):
    """Replacement for ``LlamaAttention / Qwen2Attention / Qwen3Attention.forward`` (transformers 5.x signature)."""
    from transformers.models.llama.modeling_llama import apply_rotary_pos_emb

    bsz, q_len = hidden_states.shape[:2]
    hidden_shape = (bsz, q_len, -1, self.head_dim)
    if hasattr(self, "q_norm"):  # Qwen3 (reference attention/attn.py:33-35)
        query_states = self.q_norm(self.q_proj(hidden_states).view(hidden_shape)).transpose(1, 2)
        key_states = self.k_norm(self.k_proj(hidden_states).view(hidden_shape)).transpose(1, 2)
    else:
        query_states = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)
        key_states = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    value_states = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)

    cos, sin = position_embeddings
    query_states, key_states = apply_rotary_pos_emb(query_states, key_states, cos, sin)

    kv = past_key_values
    # generation step on a pruned slack-layout cache: update + prepare + attend in one launch (bit-identical result)
    if (q_len == 1 and getattr(kv, "pruned", None) and not getattr(kv, "get_score", None)
            and getattr(kv, "layout", None) == "slack" and hasattr(kv, "update_attend")):
        attn = kv.update_attend(query_states, key_states, value_states, self.layer_idx, softmax_scale=self.scaling)
        n_kv = self.config.num_key_value_heads
        attn_output = attn.view(bsz, n_kv, q_len, -1, self.head_dim).transpose(1, 2).reshape(bsz, q_len, -1).contiguous()
        return self.o_proj(attn_output), None
    if kv is not None:
        key_states, value_states = kv.update(key_states, value_states, self.layer_idx)

    attn_output = None
    if getattr(kv, "get_score", None):  # calculate KV importance (attention/attn.py:53-54)
        query_states = query_states.contiguous()
        if getattr(kv, "fuse_forward_score", False) and not getattr(kv, "pruned", None):
            # f2: ONE attention kernel produces the layer's output and the row statistics of the scores (None: shape not taken)
            attn_output = kv._score_forward(query_states, key_states, value_states, self.layer_idx, softmax_scale=self.scaling)
        if attn_output is None:
            kv._get_score(query_states, key_states, self.layer_idx)

    if attn_output is not None:
        pass
    elif getattr(kv, "pruned", None):     # attention with the pruned cache (attention/attn.py:56-73)
        q, k, v, info = kv.prepare(query_states.contiguous(), key_states, value_states, self.layer_idx)
        attn = kv.attend(q, k, v, info, causal=True, softmax_scale=self.scaling)      # [Hkv*q_len, G, D]
        n_kv = self.config.num_key_value_heads
        attn_output = attn.view(bsz, n_kv, q_len, -1, self.head_dim).transpose(1, 2)  # [b, q, Hkv, G, D]
    else:
        attn_output, _ = dense_causal_attention(self, query_states, key_states, value_states, None,
                                                scaling=self.scaling)

    attn_output = attn_output.reshape(bsz, q_len, -1).contiguous()
    return self.o_proj(attn_output), None

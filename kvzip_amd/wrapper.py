"""ModelKVzip — the model-level API of the reference (reference model/wrapper.py:61-306), same method names and
argument meaning: ``prefill / scoring / self_task / generate / apply_template / __call__ / _prob``.

    model = ModelKVzip("Qwen/Qwen2.5-7B-Instruct-1M")     # or ModelKVzip(hf_model, tokenizer=tok)
    kv = model.prefill(context, load_score=False)          # prefill KV cache + importance scoring
    kv.prune(ratio=0.3)                                     # eviction (score -> select -> compact on the GPU)
    answer = model.generate(query, kv=kv, update_cache=False)

What differs from the reference (deliberately):
  * the model may be passed as an object (random-init models from a config work: there is no network here);
  * token ids can be used everywhere text is accepted, so the eviction path runs without a tokenizer;
  * generation is a small greedy loop over the patched HF decoder (the reference calls ``model.generate`` of
    transformers 4.51.3, whose cache plumbing changed in 5.x); semantics kept: greedy, ``max_new_tokens``,
    the cache is sliced back to the context unless ``update_cache=True``;
  * head-level scores (``load_score=True``) are read from ``head_score_dir`` (the reference hard-codes
    ./utils/head_score/, model/wrapper.py:40-58).
"""
from __future__ import annotations

import glob
import os
from typing import List, Optional, Tuple, Union

import torch

from .kvcache import EvictCache, RetainCache
from .monkeypatch import replace_attn
from .ops import KvzError
from .template import template


def chunk_fn(ctx_ids: torch.Tensor, chunk_size: int) -> List[torch.Tensor]:
    """Split ``[1, n]`` token ids into chunks of ``chunk_size`` (reference model/wrapper.py:18-37)."""
    n = ctx_ids.shape[1]
    if n <= chunk_size:
        return [ctx_ids]
    return [ctx_ids[:, s:s + chunk_size] for s in range(0, n, chunk_size)]


def head_score_name(model_name: str) -> str:
    """File-name stem of a model's head-score files (reference model/wrapper.py:41-46)."""
    for prefix, short in (("Qwen2.5-7B", "qwen2.5-7b"), ("Qwen2.5-14B", "qwen2.5-14b"), ("Llama-3.1-8B", "llama3.1-8b")):
        if model_name.startswith(prefix):
            return short
    return model_name


def load_head_score(model_name: str, ctx_len: int, head_score_dir: str, device) -> torch.Tensor:
    """Head-level scores expanded over the context: ``[L, 1, Hkv, ctx_len]`` (reference model/wrapper.py:40-58).
    The expansion is a stride-0 VIEW of the ``[L, Hkv]`` maxima: ``prune`` recognises it and selects / compacts whole
    heads without ever materialising anything of size ``ctx_len``."""
    name = head_score_name(model_name)
    paths = sorted(glob.glob(os.path.join(head_score_dir, f"{name}-*.pt")))
    if not paths:
        raise FileNotFoundError(f"no head-score file {name}-*.pt under {head_score_dir}")
    attn = torch.stack([torch.load(p, map_location="cpu").squeeze() for p in paths], dim=0).amax(0).to(device)
    return attn.unsqueeze(-1).expand(-1, -1, ctx_len).unsqueeze(1)


def _on_model_device(fn):
    """Run a ModelKVzip method with the model's GPU as the current device: the library launches on the CURRENT HIP device, and a
    model that lives on cuda:1 while cuda:0 is current must work the same way (side streams, events, workspaces of the cache)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        dev = self.device   # (always a GPU: __init__ refuses anything else)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)
    return wrapped


class ModelKVzip:

    def __init__(self, model: Union[str, torch.nn.Module], kv_type: str = "evict", tokenizer=None,
                 name: Optional[str] = None, head_score_dir: str = "./utils/head_score", max_new_tokens: int = 512,
                 eos_token_id: Optional[Union[int, List[int]]] = None, cache_kwargs: Optional[dict] = None):
        if isinstance(model, str):  # reference model/load.py:58-64 (needs the HF hub or a local path)
            from transformers import AutoModelForCausalLM, AutoTokenizer
            replace_attn(model)
            name = name or model.split("/")[-1]
            tokenizer = tokenizer or AutoTokenizer.from_pretrained(model)
            # ONE device: a cache object (side streams, events, workspaces) lives on one GPU, and the multi-GPU scheme of this path is
            # one context per GPU (kvzip_amd/dist.py), not one model sharded over several
            if not torch.cuda.is_available():
                raise KvzError("ModelKVzip needs a GPU: the hot path (dense and post-prune attention, scoring, selection, compaction) "
                               "runs on the library's HIP kernels only, there is no CPU implementation")
            dev = f"cuda:{torch.cuda.current_device()}"
            model = AutoModelForCausalLM.from_pretrained(model, torch_dtype="auto", device_map=dev).eval()
        else:
            replace_attn(name or type(model).__name__)
        self.model, self.tokenizer = model, tokenizer
        devices = {p.device for p in model.parameters()}
        if len(devices) > 1:
            raise ValueError(f"the model is sharded over {sorted(map(str, devices))}: ModelKVzip needs it on ONE device (one context "
                             "per GPU is the multi-GPU scheme of this path, kvzip_amd/dist.py)")
        self.name = name or type(model).__name__
        self.dtype = next(model.parameters()).dtype
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":   # (fail HERE, not at the first forward inside kvzip_amd.attn)
            raise KvzError(f"ModelKVzip: the model lives on {self.device}; the product path has no CPU implementation - move it to the GPU "
                           "(model.to('cuda')) before wrapping it")
        self.config = model.config
        self.config._attn_implementation = "kvzip_hip"  # dense path through kvzip_amd.attn, no HF mask construction
        self.kv_type = kv_type
        # f2: scoring passes can take the row statistics of the scores from the forward's own attention kernel (one QK^T over the
        # window instead of two).  Built, parity-tested - and OFF by default: the row-statistics pass is bound by its VALU rounding
        # chain, not by the matrix cores, so sharing the QK^T saves nothing, and inside the forward that chain sits on the critical
        # path (+17 % on a 2 026 x 35 k forward, profiles/r3_flash_probe.txt) instead of on a side stream beside the MLP GEMMs
        self.fuse_forward_score = False
        self.head_score_dir = head_score_dir
        self.cache_kwargs = dict(cache_kwargs or {})
        # the reference's generation settings (model/wrapper.py:81-94): greedy by default, the sampling knobs of HF generate under
        # the same names; Qwen3 stops at <|im_end|> = 151645 like there
        self.gen_kwargs = {"do_sample": False, "temperature": 1.0, "top_p": 1, "top_k": None, "max_new_tokens": max_new_tokens}
        if eos_token_id is None and getattr(self.config, "model_type", "") == "qwen3":
            eos_token_id = 151645
        if eos_token_id is None:
            eos_token_id = getattr(model.generation_config, "eos_token_id", None) if hasattr(model, "generation_config") else None
        if eos_token_id is not None:
            self.gen_kwargs["eos_token_id"] = eos_token_id
        self.sample_generator: Optional[torch.Generator] = None   # seed source of do_sample=True (None: torch's global generator)
        self.sys_prompt_ids = torch.zeros((1, 0), dtype=torch.long, device=self.device)
        self.postfix_ids = torch.zeros((1, 0), dtype=torch.long, device=self.device)
        if tokenizer is not None:
            self.set_chat_template()

    # ---- text <-> ids (reference model/wrapper.py:99-118) -----------------------------------------------
    def encode(self, text: str) -> torch.Tensor:
        if self.tokenizer is None:
            raise ValueError("no tokenizer: pass token-id tensors instead of text")
        return self.tokenizer.encode(text, add_special_tokens=False, return_tensors="pt").to(self.device)

    def decode(self, input_ids: torch.Tensor) -> str:
        if input_ids.dim() == 2:
            input_ids = input_ids[0]
        if self.tokenizer is None:
            return " ".join(str(int(t)) for t in input_ids)
        return self.tokenizer.decode(input_ids)

    def set_chat_template(self, task: str = "qa"):
        prefix, postfix = template(self.name, task)
        self.sys_prompt_ids, self.postfix_ids = self.encode(prefix), self.encode(postfix)

    def set_prompt_ids(self, sys_prompt_ids: torch.Tensor, postfix_ids: torch.Tensor):
        """Tokenizer-free variant of ``set_chat_template``."""
        self.sys_prompt_ids, self.postfix_ids = sys_prompt_ids.to(self.device), postfix_ids.to(self.device)

    def apply_template(self, query: Union[str, torch.Tensor]) -> torch.Tensor:
        q_ids = self.encode(f"\n\n{query.strip()}") if isinstance(query, str) else query.to(self.device)
        return torch.cat([q_ids, self.postfix_ids], dim=1)

    # ---- forward (reference model/wrapper.py:120-146) ---------------------------------------------------------
    @_on_model_device
    @torch.inference_mode()
    def __call__(self, input_ids: torch.Tensor, kv, update_cache: bool = False, return_logits: bool = False,
                 *args, **kwargs):
        """Transformer forward pass.  By default the KV of ``input_ids`` is NOT kept (``update_cache=False``)."""
        seen_token_prev = kv._seen_tokens
        if return_logits:
            outputs = self.model(input_ids, past_key_values=kv, use_cache=True, *args, **kwargs)
        else:
            _ = self.model.model(input_ids, past_key_values=kv, use_cache=True, *args, **kwargs)
            outputs = None
        if not update_cache:
            kv.slice(seen_token_prev)
        return outputs

    def _init_kv(self, kv=None, evict_range: Tuple[int, int] = (0, 0)):
        if kv is None:
            if self.kv_type == "retain":
                kv = RetainCache(self.model, evict_range, **self.cache_kwargs)
            elif self.kv_type == "evict":
                kv = EvictCache(self.model, evict_range, **self.cache_kwargs)
            else:
                raise NotImplementedError(f"type {self.kv_type} is not implemented on this path "
                                          "(int4static / hybrid_static are retain-only in the reference)")
        return kv

    # ---- prefill + scoring (reference model/wrapper.py:169-249) ------------------------------------------------
    @_on_model_device
    @torch.inference_mode()
    def prefill(self, ctx_ids: Union[str, torch.Tensor], prefill_chunk_size: int = 16000, load_score: bool = False,
                do_score: bool = True):
        """Chunked prefill of the KV cache, then KV importance scoring."""
        if isinstance(ctx_ids, str):
            ctx_ids = self.encode(ctx_ids)
        ctx_ids = ctx_ids.to(self.device)
        prefill_ids = torch.cat([self.sys_prompt_ids, ctx_ids], dim=1)
        evict_range = (self.sys_prompt_ids.shape[1], prefill_ids.shape[1])
        kv = self._init_kv(evict_range=evict_range)  # the system prompt is never evicted
        kv.ctx_ids = ctx_ids
        kv.prefill_ids = prefill_ids
        for input_ids in chunk_fn(prefill_ids, prefill_chunk_size):
            self.__call__(input_ids, kv, update_cache=True)
        if do_score:
            self.scoring(kv, ctx_ids, load_score=load_score)
        return kv

    def self_task(self, ctx_ids: torch.Tensor, chunk_size: int = 2000, prev_postfix_size: int = 8,
                  repeat_prompt_ids: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """Chunked "repeat the context" inputs: list of (chunk ids, repeat prompt ++ postfix ++ chunk ids)."""
        chunked = chunk_fn(ctx_ids.to(self.device), chunk_size)
        if repeat_prompt_ids is None:
            first = self.encode("\n\nRepeat the previous context exactly.")
            later = self.encode("\n\nRepeat the part of the previous context exactly, starting with ")
        else:
            first, later = (t.to(self.device) for t in repeat_prompt_ids)
        inputs = []
        for i, a_ids in enumerate(chunked):
            if i == 0:
                q_ids = first
            else:
                q_ids = torch.cat([later, chunked[i - 1][:, -prev_postfix_size:]], dim=1)
            inputs.append((a_ids, torch.cat([q_ids, self.postfix_ids, a_ids], dim=1)))
        return inputs

    @_on_model_device
    @torch.inference_mode()
    def scoring(self, kv, ctx_ids: torch.Tensor, load_score: bool = False, chunk_size: int = 2000,
                repeat_prompt_ids=None):
        """KV importance scoring (fills ``kv.score``)."""
        if not load_score:
            kv.init_score()
            fused = hasattr(kv, "fuse_update_score")
            if fused:
                kv.fuse_update_score = True  # the forward pass is kvzip_amd.attn: update() is always followed by _get_score()
                kv.fuse_forward_score = self.fuse_forward_score  # ... and its attention kernel emits the row statistics (f2)
            start_idx_tmp = kv.start_idx
            kv.end_idx = 0
            try:
                for prefill_ids_p, repeat_ids_p in self.self_task(ctx_ids, chunk_size=chunk_size,
                                                                  repeat_prompt_ids=repeat_prompt_ids):
                    kv.end_idx = kv.start_idx + prefill_ids_p.shape[1]      # window of this chunk
                    self.__call__(repeat_ids_p, kv, update_cache=False)     # the patched attention calls kv._get_score
                    kv.start_idx = kv.end_idx
            finally:
                # also when a forward pass raised (out of memory ...): the cache must not stay in the mode in which update()
                # returns views whose new rows are still to be written, nor keep the chunk's K,V pinned
                kv.start_idx = start_idx_tmp
                if fused:
                    kv.fuse_update_score = False
                    kv.fuse_forward_score = False
                    kv._flush_append()
            assert kv.score[0].shape[-1] == kv.ctx_len
        else:
            kv.score = load_head_score(self.name, kv.ctx_len, self.head_score_dir, self.device)
        kv.get_score = False

    @property
    def eos_token_ids(self) -> List[int]:
        e = self.gen_kwargs.get("eos_token_id")
        return [e] if isinstance(e, int) else list(e or [])

    @staticmethod
    def next_token(logits: torch.Tensor, gen_kwargs: dict, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """Next token ``[b, 1]`` from last-position logits ``[b, V]`` under the reference's ``gen_kwargs`` (model/wrapper.py:81-87,
        passed to HF ``generate`` there): ``do_sample=False`` -> argmax; otherwise temperature, then top-k, then top-p (nucleus: the
        smallest set of most probable tokens whose mass reaches ``top_p``, at least one token), then one multinomial draw - the
        order and the semantics of transformers' logits warpers."""
        if not gen_kwargs.get("do_sample", False):
            return logits.argmax(-1, keepdim=True)
        x = logits.float()
        t = gen_kwargs.get("temperature", 1.0)
        if t is not None and t != 1.0:
            x = x / float(t)
        k = gen_kwargs.get("top_k")
        if k:
            kth = torch.topk(x, min(int(k), x.shape[-1]), dim=-1).values[..., -1:]
            x = x.masked_fill(x < kth, float("-inf"))
        p = gen_kwargs.get("top_p", 1)
        if p is not None and p < 1:
            sx, si = torch.sort(x, dim=-1, descending=False)
            cum = torch.softmax(sx, dim=-1).cumsum(-1)
            drop = cum <= (1 - float(p))          # the low-probability tail whose mass stays below 1 - top_p
            drop[..., -1:] = False                # (at least the most probable token survives)
            x = x.masked_fill(torch.zeros_like(drop).scatter(-1, si, drop), float("-inf"))
        return torch.multinomial(torch.softmax(x, dim=-1), 1, generator=generator)

    # ---- generation (reference model/wrapper.py:251-284) ---------------------------------------------------------
    @_on_model_device
    @torch.inference_mode()
    def generate(self, query: Union[str, torch.Tensor], kv=None, update_cache: bool = False,
                 return_ids: bool = False):
        """Response to ``query`` under ``self.gen_kwargs`` (the reference's knobs, model/wrapper.py:81-94: greedy by default;
        ``do_sample / temperature / top_k / top_p / max_new_tokens / eos_token_id`` as HF ``generate`` takes them there).  The
        reference hands the loop to ``model.generate(input_ids, past_key_values=kv, **gen_kwargs)`` (model/wrapper.py:276); the
        token loop lives here because the cache object is not a ``transformers.Cache`` subclass in transformers 5.x - same tokens
        for the same knobs, and like there the prompt is only what is new (HF slices ``input_ids[:, -new:]`` off the full
        sequence).  The KV of the query and of the answer is evicted afterwards (``kv.slice``) unless ``update_cache=True``
        (multi-turn)."""
        kv = self._init_kv(kv=kv)
        seen_token_prev = kv._seen_tokens
        input_ids = self.encode(query) if isinstance(query, str) else query.to(self.device)
        out_ids = []
        cur = input_ids
        for _ in range(self.gen_kwargs["max_new_tokens"]):
            logits = self.model(cur, past_key_values=kv, use_cache=True).logits[:, -1]
            nxt = self.next_token(logits, self.gen_kwargs, self.sample_generator)
            out_ids.append(nxt)
            if self.eos_token_ids and int(nxt) in self.eos_token_ids:
                break
            cur = nxt
        # the reference drops the last generated token (model/wrapper.py:277: output[:, len(input_ids):-1]): it is the
        # terminating EOS (or the token that hit max_new_tokens) and its KV was never written to the cache
        a_ids = torch.cat(out_ids, dim=1)[:, :-1]
        if not update_cache:
            kv.slice(seen_token_prev)
        elif kv.prefill_ids is not None:
            kv.prefill_ids = torch.cat([kv.prefill_ids, input_ids, a_ids], dim=1)
        return a_ids if return_ids else self.decode(a_ids)

    def head_score(self, kv) -> torch.Tensor:
        """Per-(layer, KV head) maximum score ``[L, Hkv]`` — what the reference saves with ``--save_head_score``
        (test.py:22-25: ``torch.stack(kv.score, dim=0).squeeze().amax(-1)``) for context-independent, head-level eviction."""
        from . import ops
        score = kv._stacked_score(kv.score)
        if not score.is_cuda:  # (host-side scores, e.g. loaded from a file: no kernel needed)
            return score.reshape(score.shape[0], score.shape[-2], score.shape[-1]).amax(-1)
        return ops.rowmax(score.reshape(score.shape[0], score.shape[-2], score.shape[-1]))

    def save_head_score(self, kv, data: str, idx: int, head_score_dir: Optional[str] = None) -> str:
        """Write the head scores of ``kv`` where ``load_score=True`` looks for them: ``<dir>/<name>-<data>-<idx>.pt`` holding
        a ``[L, Hkv]`` tensor in the model dtype (reference test.py:22-25).  Several files of one model are combined by
        ``load_head_score`` with an element-wise maximum."""
        d = head_score_dir or self.head_score_dir
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"{head_score_name(self.name)}-{data}-{idx}.pt")
        torch.save(self.head_score(kv).cpu(), path)
        return path

    def eval_ratios(self, kv, ratios, fn, level: str = "pair"):
        """Multi-ratio evaluation from ONE prefill (reference eval.py:30-36): for every ratio ``kv.prune(ratio, level)`` on a
        non-evicting cache (``kv_type="retain"``: the mask is re-applied at every attention call) and ``fn(kv)``.
        Returns ``[([ratio, round(real_ratio, 4), round(thres, 4)], fn(kv)), ...]`` — the record layout of eval.py:34."""
        assert isinstance(kv, RetainCache), "several ratios from one prefill need the non-evicting cache (kv_type='retain')"
        out = []
        for ratio in ratios:
            thres, ratio_true = kv.prune(ratio, level)
            out.append(([ratio, round(ratio_true, 4), round(thres, 4)], fn(kv)))
        return out

    @_on_model_device
    @torch.inference_mode()
    def _prob(self, input_ids: torch.Tensor, kv=None, device: str = "cuda") -> torch.Tensor:
        """Next-token probabilities (reference model/wrapper.py:286-306)."""
        kv = self._init_kv(kv=kv)
        output = self.__call__(input_ids.to(self.device), kv, update_cache=False, return_logits=True)
        probs = torch.softmax(output.logits[0].float(), dim=-1).squeeze()
        return probs.cpu() if device == "cpu" else probs

"""System-prompt / postfix strings per chat format (role of the reference's model/template.py:1-36).
Only their token COUNT matters to the eviction path: ``sink = len(sys_prompt_ids)`` is never evicted
(reference model/wrapper.py:182)."""
from __future__ import annotations

from typing import Tuple


def template(model_name: str, task: str = "qa") -> Tuple[str, str]:
    name = model_name.lower()
    if "llama" in name or "duo" in name:
        prefix = ("<|begin_of_text|><|start_header_id|>system<|end_header_id|>\n\nYou are a helpful assistant."
                  "<|eot_id|><|start_header_id|>user<|end_header_id|>\n\n")
        postfix = "<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n"
    elif "qwen" in name:
        prefix = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n"
        postfix = "<|im_end|>\n<|im_start|>assistant\n"
        if "qwen3" in name:
            postfix += "<think>\n\n</think>\n\n"
    else:
        prefix, postfix = "", "\n"
    if task.startswith("gsm"):
        prefix += "Solve the problem step by step.\n\n"
    return prefix, postfix

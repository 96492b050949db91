"""System-prompt / postfix strings per chat format — the strings of the reference's model/template.py:5-33, because with a
real tokenizer they decide ``sink = len(sys_prompt_ids)`` (never evicted, reference model/wrapper.py:182) and the prompts the
model sees.  Table-driven: (match rule, prefix parts, postfix)."""
from __future__ import annotations

from typing import Tuple

_TASK_LINE = {
    True: "Given the context, answer to the following reasoning question.\n\n",                       # gsm*
    False: "Given the context, answer to the following question or request without explanation.\n\n",
}

_FORMATS = (
    # Llama-3.1 card format; "duo" is a Llama-3 derivative
    (lambda n: "llama" in n or n == "duo",
     "<|begin_of_text|><|start_header_id|>system<|end_header_id|>\n\n"
     "You are a helpful assistant<|eot_id|><|start_header_id|>user<|end_header_id|>\n\n",
     lambda n: "\n\n<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n"),
    # ChatML (Qwen2.5 / Qwen3; Qwen3 gets an empty thinking block)
    (lambda n: n.startswith("qwen"),
     "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n",
     lambda n: "<|im_end|>\n<|im_start|>assistant\n" + ("<think>\n\n</think>\n\n" if "qwen3-" in n else "")),
    (lambda n: n.startswith("gemma3") or n.startswith("gemma-3"),
     "<bos><start_of_turn>user\nYou are a helpful assistant.\n\n",
     lambda n: "<end_of_turn>\n<start_of_turn>model\n"),
)


def template(model_name: str, task: str = "qa") -> Tuple[str, str]:
    name = model_name.lower()
    for match, prefix, postfix in _FORMATS:
        if match(name):
            break
    else:  # unknown family: the reference's fallback
        prefix, postfix = "<|begin_of_text|>", lambda n: "\n\nAnswer: "
    return prefix + _TASK_LINE[task.startswith("gsm")], postfix(name)

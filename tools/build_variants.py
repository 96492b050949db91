#!/usr/bin/env python
"""Build A/B variants of the library into tools/ab/lib_<name>.so (same sources, different -D flags):
   python tools/build_variants.py name:-DKVZ_FOO=1,-DKVZ_BAR=2 name2: ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "kvzip_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "ab")
os.makedirs(OUT, exist_ok=True)
SRCS = ["kvz_api.hip", "kvz_select.hip", "kvz_compact.hip", "kvz_score.hip", "kvz_attn.hip", "kvz_flash.hip", "kvz_flash2.hip"]
base_objs = {s: os.path.join(CS, s.replace(".hip", ".o")) for s in SRCS}
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    flags = [f for f in flags.split(",") if f]
    objs = []
    for s in SRCS:
        # only the files whose macros are touched are recompiled (KVZ_P*/KVZ_SCHED/KVZ_KSPLIT* live in kvz_score.hip, KVZ_ATTN* in kvz_attn.hip)
        need = any(("ATTN" in f) == (s == "kvz_attn.hip") for f in flags) and s in ("kvz_score.hip", "kvz_attn.hip")
        if not need:
            objs.append(base_objs[s])
            continue
        obj = os.path.join(OUT, f"{name}_{s.replace('.hip', '.o')}")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "-c",
                               os.path.join(CS, s), "-o", obj])
        objs.append(obj)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o",
                           os.path.join(OUT, f"lib_{name}.so")])
    for o in objs:
        if o.startswith(OUT):
            os.remove(o)
    print("built", name, flags)

#!/usr/bin/env python
"""Instruction mix of the MFMA-bearing basic blocks of one kernel (device ISA from hipcc -S):
   python tools/isa_blocks.py <kernel-name-substring> [-Dflags ...]      (source: kvzip_amd/csrc/kvz_score.hip, or SRC=...)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.environ.get("SRC", os.path.join(ROOT, "kvzip_amd", "csrc", "kvz_score.hip"))
pat, flags = sys.argv[1], sys.argv[2:]
out = "/tmp/isa_blocks.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *flags, "-o", out, src],
                      stderr=subprocess.DEVNULL)
lines, on = [], False
for l in open(out):
    if re.match(r"^_Z.*" + re.escape(pat) + r".*:", l):
        on = True
    if on:
        lines.append(l)
        if ".end_amdhsa_kernel" in l:
            break
blocks, cur, name = [], [], "entry"
for l in lines:
    if re.match(r"^\.LBB", l):
        blocks.append((name, cur)); name, cur = l.split(":")[0], []
    else:
        m = re.match(r"^\s+([vsdg][a-z0-9_]+)", l)
        if m:
            cur.append(m.group(1))
blocks.append((name, cur))
tot = collections.Counter()
for n, b in blocks:
    c = collections.Counter(b)
    nm = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    tot.update(c)
    if nm >= 8:
        nv = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        print(f"{n:12s} instr {len(b):4d} mfma {nm:3d} valu {nv:4d} ({nv / nm:5.2f}/mfma) accvgpr {sum(v for k, v in c.items() if 'accvgpr' in k):3d} "
              f"ds {sum(v for k, v in c.items() if k.startswith('ds_')):3d} nop {c['s_nop']:3d} waitcnt {c['s_waitcnt']:3d} mov {c['v_mov_b32_e32']:3d} "
              f"scratch {sum(v for k, v in c.items() if k.startswith('scratch')):2d}")
print("kernel total:", sum(tot.values()), "instr;", "accvgpr", sum(v for k, v in tot.items() if "accvgpr" in k), "scratch",
      sum(v for k, v in tot.items() if k.startswith("scratch")))

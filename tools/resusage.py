#!/usr/bin/env python
"""Register / scratch usage of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage, parsed).
   python tools/resusage.py kvz_score.hip [-DFOO=1 ...]"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
       *sys.argv[2:], "-o", "/tmp/" + src.replace(".hip", ".s"), "kvzip_amd/csrc/" + src]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:90]}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPRs Spill", "SGPRs Spill", "SGPRs", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
for r in rows:
    print(f"{r['name']:<92} vgpr {r.get('VGPRs'):>3} agpr {r.get('AGPRs', 0):>3} scratch {r.get('ScratchSize [bytes/lane]'):>3} "
          f"spill {r.get('VGPRs Spill')}/{r.get('SGPRs Spill')} occ {r.get('Occupancy [waves/SIMD]')} lds {r.get('LDS Size [bytes/block]')}")

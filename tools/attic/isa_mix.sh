#!/bin/bash
# instruction histogram of the MFMA-bearing basic blocks of one kernel:  tools/isa_mix.sh <mangled-name-substring>
set -e
SRC=${SRC:-kvzip_amd/csrc/kvz_score.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I kvzip_amd/csrc -S --cuda-device-only "$SRC" -o /tmp/isa_mix.s 2>/dev/null
awk -v pat="$1" '$0 ~ "^_ZN.*" pat ".*:" {f=1} f{print} /\.end_amdhsa_kernel/{if(f){exit}}' /tmp/isa_mix.s > /tmp/isa_mix_kernel.s
# split into basic blocks at labels / branches; print histogram for blocks with >= 8 MFMAs
awk '
function flush() { if (mf >= 8) { printf("---- block ending line %d: %d instr, %d mfma\n", NR, n, mf); for (k in h) printf("%6d %s\n", h[k], k) | "sort -rn"; close("sort -rn") } delete h; n=0; mf=0 }
/^\.LBB/ { flush(); next }
/^[ \t]+[vsdg][a-z0-9_]+/ { op=$1; h[op]++; n++; if (op ~ /v_mfma/) mf++; if (op ~ /s_cbranch|s_branch/) flush() }
END { flush() }' /tmp/isa_mix_kernel.s

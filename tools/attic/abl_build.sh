#!/bin/bash
# time-attribution builds of the scoring kernels: tools/abl_build.sh <mask> [<mask> ...]  ->  tools/ab/lib_abl_<mask>.so
# The product source carries no ablation code: tools/abl_score.patch (KVZ_ABL / KVZ_MSCHED macros) is applied to a COPY of
# kvz_score.hip; with KVZ_ABL = 0 the patched file compiles to the same ISA as the product file (checked with hipcc -S).
# KVZ_ABL bits: 1 no hand-over barrier, 2 no DMA in the loops, 4 no fragment reads, 8 no MFMA, 16 no VALU epilogue,
# 32 no cold-path check, 64 no s_setprio.  Results of these builds are garbage, their times are the measurement.
# (masks without bit 32 or 16 but with 8: the stale accumulators send every step through the cold path - use 40, not 8)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab/obj
cp kvzip_amd/csrc/kvz_score.hip tools/ab/obj/kvz_score_abl.hip
patch -s tools/ab/obj/kvz_score_abl.hip < tools/abl_score.patch
for m in "$@"; do
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Ikvzip_amd/csrc -DKVZ_ABL=$m $EXTRA -c tools/ab/obj/kvz_score_abl.hip -o tools/ab/obj/score_abl_$m.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/ab/obj/score_abl_$m.o kvzip_amd/csrc/kvz_api.o kvzip_amd/csrc/kvz_select.o kvzip_amd/csrc/kvz_compact.o \
        kvzip_amd/csrc/kvz_attn.o kvzip_amd/csrc/kvz_flash.o kvzip_amd/csrc/kvz_flash2.o -o tools/ab/lib_abl_$m.so
    echo built $m
  ) &
done
wait

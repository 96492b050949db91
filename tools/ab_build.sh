#!/bin/bash
# A/B builds of the library from the CURRENT tree with extra -D flags for kvz_score.hip: tools/ab_build.sh <name> [-DFLAG=V ...] -> tools/ab/lib_<name>.so
# (the other objects are the product build's: run `make -C kvzip_amd/csrc` first)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/ab/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Ikvzip_amd/csrc "$@" -c kvzip_amd/csrc/kvz_score.hip -o tools/ab/obj/score_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/ab/obj/score_$name.o kvzip_amd/csrc/kvz_api.o kvzip_amd/csrc/kvz_select.o kvzip_amd/csrc/kvz_compact.o \
    kvzip_amd/csrc/kvz_attn.o kvzip_amd/csrc/kvz_flash.o kvzip_amd/csrc/kvz_flash2.o -o tools/ab/lib_$name.so
echo built tools/ab/lib_$name.so

#!/bin/bash
# device ISA of one source file: tools/isa.sh kvz_score.hip [extra -D flags] -> /tmp/<name>.s
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o /tmp/$(basename $f .hip).s kvzip_amd/csrc/$f

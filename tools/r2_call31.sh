#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/abl_time.py
bash tools/r2_call21.sh "$@"

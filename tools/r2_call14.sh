#!/bin/bash
# full GPU suite after the multi-row kernel + smoke
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c14_tests.log 2>&1
tail -8 gpurun_out/c14_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c14_smoke.log 2>&1
tail -3 gpurun_out/c14_smoke.log

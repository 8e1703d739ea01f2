#!/bin/bash
# Round 6, run H: mlp_bwd without LDS transposes -- tests, stand-alone timings
set -u
out=gpurun_out/r06_h; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_field_gpu.py tests/test_native_step_gpu.py -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -15 > $out/tests.txt; cat $out/tests.txt
timeout 300 python tools/mlp_bench.py > $out/mlp_bench.txt 2>&1; tail -2 $out/mlp_bench.txt

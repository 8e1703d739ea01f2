#!/bin/bash
# round 5, run B: the fused-MLP forward without the conflicted LDS gather (+ input prefetch, 8-byte stores), the fused density
# pass of the evaluation image; tests of the files they touch, both kernels timed alone, one evaluation image priced
set -u
out=gpurun_out/r05_b; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_field_gpu.py tests/test_training_gpu.py tests/test_image_parity_gpu.py tests/test_eval_metrics.py tests/test_boundary.py -q -m gpu 2>&1 | tail -25 > $out/tests.txt
tail -4 $out/tests.txt
timeout 200 python tools/mlp_bench.py > $out/mlp_bench.json 2> $out/mlp_bench.err; cat $out/mlp_bench.json
timeout 400 python tools/eval_bench.py --price > $out/eval_bench.txt 2> $out/eval_bench.err; grep -a "preblend=" $out/eval_bench.txt; tail -1 $out/eval_bench.txt | cut -c1-1500

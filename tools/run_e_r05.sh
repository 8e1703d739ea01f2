#!/bin/bash
# round 5, run E: the table-indexed deformation forward with merged weight stages (5 barriers per tile, bias folded into the row
# terms, one v_sin per input element): operator tests, image-level parity, both forwards timed against each other
set -u
out=gpurun_out/r05_e; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_deform_gpu.py tests/test_image_parity_gpu.py tests/test_native_step_gpu.py tests/test_field_gpu.py -q -m gpu 2>&1 | tail -30 > $out/tests.txt
tail -6 $out/tests.txt
timeout 300 python tools/deform_fwd_ab.py --rows 24,1,64,100 > $out/deform_fwd_ab.json 2> $out/deform_fwd_ab.err; cat $out/deform_fwd_ab.json
timeout 300 python tools/mfma_bench.py --only deform_fwd,deform_fwd_general,deform_bwd > $out/mfma_bench.json 2>/dev/null; cat $out/mfma_bench.json

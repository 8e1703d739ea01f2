#!/bin/bash
# round 5, run F: the backward chain kernel recomputes its forward through the row terms (the forward kernel's pre-activations
# bit for bit; 64 fewer MFMAs, 192 instead of 256 KB of weight stages per 256 samples, no code-row loads)
set -u
out=gpurun_out/r05_f; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_deform_gpu.py tests/test_training_gpu.py tests/test_native_step_gpu.py tests/test_full_size_gpu.py -q -m gpu 2>&1 | tail -30 > $out/tests.txt
tail -6 $out/tests.txt
timeout 300 python tools/mfma_bench.py --only deform_fwd,deform_bwd > $out/mfma_bench.json 2>/dev/null; cat $out/mfma_bench.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-first-grid-phase --no-open-window --no-with-datamanager > $out/bench.json 2> $out/bench.err
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r05_f/bench.json") if l.startswith("{")][-1])
    print(round(d["ms_per_step"],3), "steady", round(d["steady_state"]["ms_per_step"],3))
    print({k:(round(v["avg_ms"],4), v["calls"]) for k,v in d["native_kernel_ms"].items()})
    for k,v in d["kernels_alone"]["kernels"].items(): print(k, v.get("ms"), v.get("frac"))
except Exception as e:
    print("ERR", e); print(open("gpurun_out/r05_f/bench.err").read()[-3000:])
P

#!/bin/bash
# Round 6, run A: the level-parallel exchange v2 (fused payloads, native step split at the exchange) -- the multi-rank tests,
# the emulated rank through RCCL, and the first line of `bench.py --level-parallel-one-rank 8`.
set -u
out=gpurun_out/r06_a; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sharded_gpu.py tests/test_native_step_gpu.py -x -q -m gpu 2>&1 | tail -25 > $out/tests.txt
cat $out/tests.txt
timeout 600 python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone > $out/lp8_rank7.json 2> $out/lp8_rank7.err
tail -5 $out/lp8_rank7.err
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06_a/lp8_rank7.json") if l.startswith("{")][-1])
    print("window ms/step", round(d["ms_per_step"],3), "steady", json.dumps(d.get("steady_state"))[:1500])
    print("comm", json.dumps(d.get("comm"))[:1200])
    print({k: v["avg_ms"] for k, v in d["native_kernel_ms"].items()})
except Exception as e:
    print("ERR", e)
P

#!/bin/bash
# Round 6, run A: the level-parallel exchange v2 (fused payloads, native step split at the exchange) -- the multi-rank tests,
# the emulated rank through RCCL, and the first line of `bench.py --level-parallel-one-rank 8`.
set -u
out=gpurun_out/r06_a; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sharded_gpu.py tests/test_native_step_gpu.py -q -m gpu 2>&1 | tail -40 > $out/tests.txt
cat $out/tests.txt
timeout 600 python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone > $out/lp8_rank7.json 2> $out/lp8_rank7.err
tail -5 $out/lp8_rank7.err
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06_a/lp8_rank7.json") if l.startswith("{")][-1])
    ss=d.get("steady_state") or {}
    print("window ms/step", round(d["ms_per_step"],3), "steady", {k: ss.get(k) for k in ("ms_per_step","host_issue_ms_per_step","samples_per_step_min_max","psnr","protocol")})
    print("steady comm", json.dumps(ss.get("comm"))[:1200])
    print("steady rooflines", json.dumps(ss.get("rooflines"))[:1500])
    print("comm", json.dumps(d.get("comm"))[:800])
    print({k: v["avg_ms"] for k, v in d["native_kernel_ms"].items()})
except Exception as e:
    print("ERR", e)
P
timeout 400 python tools/host_sections.py --level-parallel-one-rank 8 > $out/host_sections_lp8.txt 2>&1; tail -30 $out/host_sections_lp8.txt

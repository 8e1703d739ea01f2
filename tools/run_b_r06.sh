#!/bin/bash
set -u
out=gpurun_out/r06_b; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sharded_gpu.py -q -m gpu -k "marches_nothing or two_rank_training or two_ranks_on_half or emulated_rank_7" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $out/tests_full.txt
grep -n "^E \|Error\|FAILED\|passed\|failed" $out/tests_full.txt | head -60
timeout 600 python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone > $out/lp8_rank7.json 2> $out/lp8_rank7.err
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06_b/lp8_rank7.json") if l.startswith("{")][-1])
    ss=d.get("steady_state") or {}
    print("window ms/step", round(d["ms_per_step"],3), "steady", {k: ss.get(k) for k in ("ms_per_step","host_issue_ms_per_step","samples_per_step_min_max","psnr")})
    print("steady comm", json.dumps(ss.get("comm"))[:1200])
except Exception as e:
    print("ERR", e)
P
timeout 400 python tools/host_sections.py --level-parallel-one-rank 8 > $out/host_sections_lp8.txt 2>&1; tail -16 $out/host_sections_lp8.txt

#!/bin/bash
# Round 6, run B: the level-parallel tests + the emulated ranks of an 8-rank job (balanced level assignment)
set -u
out=gpurun_out/r06_b; mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_sharded_gpu.py -q -m gpu -k "level_parallel or emulated or marches_nothing or handed or eight_level" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $out/tests_full.txt
grep -n "^E  \|FAILED\|passed\|failed" $out/tests_full.txt | head -40
LP="python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
for r in 7 0 3; do timeout 600 $LP --rank $r > $out/lp8_rank$r.json 2> $out/lp8_rank$r.err; done
python - <<'P'
import json
for r in (7, 0, 3):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r06_b/lp8_rank{r}.json") if l.startswith("{")][-1])
        ss=d.get("steady_state") or {}
        c=ss.get("comm") or {}
        print("rank", r, "levels", c.get("levels"), "window ms/step", round(d["ms_per_step"],3), "steady", {k: ss.get(k) for k in ("ms_per_step","host_issue_ms_per_step","host_issue_ms_per_step_min")}, "shadow", c.get("shadow_fwd_ms"), "adam", c.get("shard_adam_ms"))
        print("   window kernels", {k: v["avg_ms"] for k, v in list(d["native_kernel_ms"].items())[:6]})
    except Exception as e:
        print("ERR", r, e)
P

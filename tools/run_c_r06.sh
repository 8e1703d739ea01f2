#!/bin/bash
# Round 6, run C: data-parallel ranks in the compact layouts -- the sharded tests, and one rank of a DP job priced on one GPU
set -u
out=gpurun_out/r06_c; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sharded_gpu.py tests/test_bench_launch_gpu.py -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $out/tests_full.txt
grep -n "^E  \|FAILED\|passed\|failed" $out/tests_full.txt | head -40
for mode in full compact; do
  flags=""; [ $mode = compact ] && flags="--compact-first-grid"
  timeout 600 python bench.py --sharded-one-rank $flags --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager > $out/dp_one_rank_$mode.json 2> $out/dp_one_rank_$mode.err
done
timeout 600 python bench.py --compact-first-grid --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone > $out/single_compact.json 2> $out/single_compact.err
python - <<'P'
import json
for f in ("dp_one_rank_full", "dp_one_rank_compact", "single_compact"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r06_c/{f}.json") if l.startswith("{")][-1])
        ss=d.get("steady_state") or {}
        print(f, "window", round(d["ms_per_step"],3), "steady", round(ss.get("ms_per_step", 0),3), "compact", d["config"]["compact_first_grid"], d["config"]["table_step"][:40])
        print("   comm", json.dumps(d.get("comm"))[:700])
        print("   kernels", {k: v["avg_ms"] for k, v in list(d["native_kernel_ms"].items())[:8]})
    except Exception as e:
        print(f, "ERR", e)
P

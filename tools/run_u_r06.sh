#!/bin/bash
# Round 6, run U: all source ranks of the level-parallel exchange in ONE launch (grid.y = source rank) against one launch per
# source rank: the new parity test, the level-parallel tests, emulated ranks 7 and 0 of 8 with both settings on one box.
set -u
out=gpurun_out/r06_u; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_boundary.py -q -m gpu -k "one_launch or emulated or level_parallel or marches_nothing or options" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -8 | tee $out/tests.txt
LP="python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
for r in 7 0; do
  timeout 400 $LP --rank $r > $out/lp8_rank${r}_one_launch.json 2> $out/lp8_rank${r}_one_launch.err
  timeout 400 $LP --rank $r --lp-launch-per-source > $out/lp8_rank${r}_per_source.json 2> $out/lp8_rank${r}_per_source.err
done
python - <<'P'
import json
for r in (7, 0):
    for tag in ("one_launch", "per_source"):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r06_u/lp8_rank{r}_{tag}.json") if l.startswith("{")][-1])
            ss = d.get("steady_state") or {}; k = d["native_kernel_ms"]; sk = {}
            print(r, tag, "window", round(d["ms_per_step"], 3), "fwd_run", k.get("nsx_lp_fwd_run", {}).get("avg_ms"), "bwd_run", k.get("nsx_lp_bwd_run", {}).get("avg_ms"),
                  "| steady", round(ss.get("ms_per_step", 0), 3), "host", round(ss.get("host_issue_ms_per_step", 0), 3),
                  "fwd_run", sk.get("nsx_lp_fwd_run", {}).get("avg_ms"), "bwd_run", sk.get("nsx_lp_bwd_run", {}).get("avg_ms"))
        except Exception as e:
            print(r, tag, "ERR", repr(e))
P

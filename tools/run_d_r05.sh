#!/bin/bash
# round 5, run D: data-parallel tests (compact moments of the narrow phases, level-parallel), fused MLPs (head input from two
# 16-byte loads), and the driver's bench command with the steady-state rooflines
set -u
out=gpurun_out/r05_d; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_mlp_gpu.py tests/test_bench_launch_gpu.py -q -m gpu 2>&1 | tail -30 > $out/tests.txt
tail -6 $out/tests.txt
timeout 200 python tools/mlp_bench.py > $out/mlp_bench.json 2> $out/mlp_bench.err; cat $out/mlp_bench.json
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r05_d/bench.json") if l.startswith("{")][-1])
    print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "roofline", json.dumps(d["roofline"])[:700])
    ss=d["steady_state"]; print("steady", round(ss["ms_per_step"],3), json.dumps(ss.get("roofline"))[:600]); print(json.dumps(ss.get("rooflines"))[:900])
    for k in ("first_grid_phase","open_window","with_datamanager"):
        v=d.get(k,{}); s2=(v.get("steady_state") or {}); print(k, v.get("ms_per_step"), s2.get("ms_per_step"), json.dumps(s2.get("roofline"))[:400], v.get("error"))
    ka=d.get("kernels_alone",{}); print({k:(v["ms"],v["frac"]) for k,v in ka.items()})
    print("cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e:
    print("ERR", e); print(open("gpurun_out/r05_d/bench.err").read()[-3000:])
P

#!/bin/bash
# Round 5, last state of the tree on ONE box: smoke, the whole -m gpu suite, the driver's bench line, the 2-rank control flows
# (gloo, both ranks on cuda:0) of the narrow exchange (one reduce-scatter call at width 1) and of the level-parallel exchange.
set -u
out=gpurun_out/r05_i; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -12 > $out/full_suite.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
DPC="python bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --reserve-gb 2"
timeout 600 $DPC --steady-after 0 > $out/dp2_narrow.json 2> $out/dp2_narrow.err
timeout 900 $DPC --window-hash 0 1 --steady-after 200 > $out/dp2_level.json 2> $out/dp2_level.err
tail -2 $out/smoke.txt; tail -4 $out/full_suite.txt
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r05_i/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], [c["kernel"] for c in d["roofline"].get("co_dominant", [])])
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("error"))
for f in ("dp2_narrow", "dp2_level"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r05_i/{f}.json") if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"], 2), json.dumps(d.get("comm"))[:1300])
        print(f, "steady", json.dumps(d.get("steady_state"))[:800])
    except Exception as e:
        print(f, "ERR", e)
P
tail -3 $out/*.err | cut -c1-300 | tail -30

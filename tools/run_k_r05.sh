#!/bin/bash
# Round 5, the tree after the ray-wise sampler tail: smoke, the whole -m gpu suite with its slowest tests listed, the driver's
# bench line, rocprofv3 kernel statistics + HBM counter passes of the same command (tools/collect_profiles.sh).
set -u
out=gpurun_out/r05_k; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu --durations=40 2>&1 | grep -a "passed\|failed\|FAILED\|Error\|s call\|s setup" | tail -60 > $out/full_suite.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
bash tools/collect_profiles.sh r05k > $out/collect.log 2>&1
tail -2 $out/smoke.txt; tail -45 $out/full_suite.txt
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r05_k/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], [c["kernel"] for c in d["roofline"].get("co_dominant", [])])
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("error"))
P
tail -12 $out/collect.log | cut -c1-250

#!/bin/bash
# Round 6, the very last tree: smoke, the whole -m gpu suite, the driver's bench line (no profiler passes: the single-GPU step's kernels
# are those of tools/final_run_r06.sh A).
set -u
out=gpurun_out/final_check_r06; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -12 > $out/full_suite.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -2 $out/smoke.txt; tail -4 $out/full_suite.txt
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/final_check_r06/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("error"))
P

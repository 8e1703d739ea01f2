#!/bin/bash
# Round 6: the table optimizer's placement calibration with 14 candidates (was 6): three fresh processes on one box
set -u
out=gpurun_out/r06_pl; mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
for i in 1 2 3; do timeout 400 $B > $out/bench_$i.json 2> $out/bench_$i.err; done
python - <<'P'
import json
for i in (1, 2, 3):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r06_pl/bench_{i}.json") if l.startswith("{")][-1])
        print(i, round(d["ms_per_step"], 3), round(d["value"] / 1e6, 2), "steady", round(d["steady_state"]["ms_per_step"], 3), "adam", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("table_placement"))
    except Exception as e:
        print(i, "ERR", repr(e))
P

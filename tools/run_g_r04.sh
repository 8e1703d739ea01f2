#!/bin/bash
# round 4, last pass: the exchange that follows the window (kernels, world-1 RCCL, world-2 gloo on one GPU, the bench's comm
# block), then the whole suite and the driver's line on the final state
set -u
export TMPDIR=/tmp
out=gpurun_out/g_r04; mkdir -p $out
python -m pytest tests/test_sharded_gpu.py tests/test_adam_gpu.py tests/test_bench_launch_gpu.py -q -m gpu 2>&1 | tail -12 > $out/tests_exchange.txt
cat $out/tests_exchange.txt
DPC="python bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --steady-after 0 --reserve-gb 2"
timeout 600 $DPC > $out/dp2_weak.json 2> $out/dp2_weak.err
timeout 600 $DPC --window-hash 0 1 > $out/dp2_weak_open_window.json 2> $out/dp2_weak_open_window.err
python - <<'P'
import json
for f in ("dp2_weak", "dp2_weak_open_window"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/g_r04/{f}.json") if l.startswith("{")][-1]); print(f, round(d["ms_per_step"],2), d.get("comm"))
    except Exception as e: print(f, "ERR", e)
P
python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -12 > $out/full_suite.txt
cat $out/full_suite.txt
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/g_r04/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("error"))
P

#!/bin/bash
# round 4: the steady state un-traced through the timeline's harness, the bench line without any kernel events, the priced
# evaluation image, the one test of run D that failed on its own filter
set -u
export TMPDIR=/tmp
out=gpurun_out/e_r04; mkdir -p $out
python -m pytest tests/test_native_step_gpu.py -q -m gpu 2>&1 | tail -3 > $out/tests.txt
python tools/host_profile.py --plain --steps 100 > $out/untraced_full.txt 2>&1
python tools/host_profile.py --plain --steps 100 --compact > $out/untraced_compact.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager --no-kernel-events > $out/bench_no_events.json 2> $out/bench_no_events.err
python tools/eval_bench.py --price > $out/eval_price.txt 2> $out/eval_price.err
cat $out/tests.txt; tail -1 $out/untraced_full.txt; tail -1 $out/untraced_compact.txt
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/e_r04/bench_no_events.json") if l.startswith("{")][-1])
print("no events:", round(d["ms_per_step"],3), d["steady_state"]["ms_per_step"])
P
tail -6 $out/eval_price.txt | cut -c1-3000

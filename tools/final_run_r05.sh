#!/bin/bash
# Round-5 evidence on ONE box.  part A: smoke, the whole -m gpu suite, the driver's bench line, rocprofv3 kernel statistics + the
# HBM counter passes of the headline and of the three steady states (full layout, open window, compact first grid).
# part B: SQ counters of the MFMA kernels alone, one evaluation image priced, the other BASELINE configurations, the steady-state
# device timeline, host issue time, the level-parallel kernels of one rank, the 2-rank control flows (gloo, both ranks on cuda:0).
#   usage: bash tools/final_run_r05.sh A|B        results: gpurun_out/final_r05/ (+ gpurun_out/prof_r05/, gpurun_out/sq_r05/)
set -u
part=${1:-A}
out=gpurun_out/final_r05; mkdir -p $out
export TMPDIR=/tmp
if [ "$part" = A ]; then
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -12 > $out/full_suite.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
bash tools/collect_profiles.sh r05 > $out/collect.log 2>&1
# counter passes of the steady states: the run settles for 600 steps first; only the dispatches of the last 8 steps count
p=gpurun_out/prof_r05
PMCS="python bench.py --preroll 600 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --steady-after 0 --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
for state in steady_full steady_open_window steady_compact; do
  flags=""; [ $state = steady_open_window ] && flags="--window-hash 0 1"; [ $state = steady_compact ] && flags="--compact-first-grid"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $p/${state}_fetch -o $state -- $PMCS $flags > /dev/null 2> $p/${state}_fetch.err
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $p/${state}_write -o $state -- $PMCS $flags > /dev/null 2> $p/${state}_write.err
  PMC_LAST_DISPATCHES=8 python tools/pmc_to_json.py $p/${state}_fetch $p/${state}_write $p/r05_$state.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), the dispatches of the last 8 steps -- $PMCS $flags" > $p/${state}_summary.txt 2>&1
  find $p/${state}_fetch $p/${state}_write \( -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
done
tail -2 $out/smoke.txt; tail -4 $out/full_suite.txt
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/final_r05/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], [c["kernel"] for c in d["roofline"].get("co_dominant", [])])
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("error"))
P
for s in steady_full steady_open_window steady_compact; do echo $s; head -12 gpurun_out/prof_r05/${s}_summary.txt; done
else
# (part A's suite: 335 passed, 1 failed on a tolerance that sat at one fp16 ulp of the gradient -- tests/test_sharded_gpu.py::
# test_sharded_step_follows_the_window; the file again with the bound restated)
timeout 600 python -m pytest tests/test_sharded_gpu.py -q -m gpu 2>&1 | tail -3 > $out/sharded_again.txt; cat $out/sharded_again.txt
bash tools/sq_counters.sh r05 > $out/sq.log 2>&1; tail -30 $out/sq.log | cut -c1-400
timeout 400 python tools/eval_bench.py --price > $out/eval_bench.txt 2> $out/eval.err; grep -a "preblend=" $out/eval_bench.txt; tail -1 $out/eval_bench.txt | cut -c1-1200
bash tools/config_lines.sh > $out/config_lines.txt 2>&1; cat $out/config_lines.txt | tail -5
timeout 300 python tools/level_parallel_bench.py > $out/level_parallel_bench.json 2> $out/level_parallel_bench.err; cut -c1-3000 $out/level_parallel_bench.json
tl=$out/tl; mkdir -p $tl
for mode in full compact; do
  flags=""; [ $mode = compact ] && flags="--compact"
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $tl/$mode -o tl -- python tools/host_profile.py --plain --steps 30 $flags > $tl/$mode.out 2> $tl/$mode.err
  python tools/timeline.py $tl/$mode 20 > $out/timeline_steady_$mode.txt 2>&1
  find $tl/$mode \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*memory_copy_trace.csv" \) -delete
done
head -30 $out/timeline_steady_full.txt
python tools/host_sections.py > $out/host_sections_compact.txt 2>&1; head -8 $out/host_sections_compact.txt
python tools/host_sections.py --full-layout > $out/host_sections_full.txt 2>&1; head -8 $out/host_sections_full.txt
DPC="python bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --reserve-gb 2"
timeout 600 $DPC --steady-after 0 > $out/dp2_narrow.json 2> $out/dp2_narrow.err
timeout 900 $DPC --window-hash 0 1 --steady-after 200 > $out/dp2_level.json 2> $out/dp2_level.err
python - <<'P'
import json
for f in ("dp2_narrow", "dp2_level"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/final_r05/{f}.json") if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"], 2), json.dumps(d.get("comm"))[:1300])
        print(f, "steady", json.dumps(d.get("steady_state"))[:1500])
    except Exception as e:
        print(f, "ERR", e)
P
fi

#!/bin/bash
# Round-6 evidence on ONE box.
#  part A: smoke, the whole -m gpu suite, the driver's bench line, rocprofv3 kernel statistics + the HBM counter passes of the headline.
#  part B: one rank of an 8-rank level-parallel job (emulated; ranks 7 and 0) and of a data-parallel job (full / compact layouts) priced
#          on one GPU, host issue time by section, the 2-rank control flows (gloo, both ranks on cuda:0), counters of the
#          level-parallel rank's optimizer pass.
#   usage: bash tools/final_run_r06.sh A|B        results: gpurun_out/final_r06/ (+ gpurun_out/prof_r06/)
set -u
part=${1:-A}
out=gpurun_out/final_r06; mkdir -p $out
export TMPDIR=/tmp
if [ "$part" = A ]; then
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -12 > $out/full_suite.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
bash tools/collect_profiles.sh r06 > $out/collect.log 2>&1
tail -2 $out/smoke.txt; tail -4 $out/full_suite.txt
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/final_r06/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], [c["kernel"] for c in d["roofline"].get("co_dominant", [])])
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("error"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:700])
print("kernels_alone", {k: (v["ms"], v["frac"]) for k, v in d["kernels_alone"]["kernels"].items()})
P
head -14 gpurun_out/prof_r06/pmc_summary.txt
else
LP="python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
timeout 600 $LP > $out/lp8_rank7.json 2> $out/lp8_rank7.err
timeout 600 $LP --rank 0 > $out/lp8_rank0.json 2> $out/lp8_rank0.err
for mode in full compact; do
  flags=""; [ $mode = compact ] && flags="--compact-first-grid"
  timeout 600 python bench.py --sharded-one-rank $flags --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager > $out/dp_one_rank_$mode.json 2> $out/dp_one_rank_$mode.err
done
timeout 400 python tools/host_sections.py --level-parallel-one-rank 8 > $out/host_sections_lp8.txt 2>&1
timeout 400 python tools/host_sections.py > $out/host_sections_compact.txt 2>&1
timeout 400 python tools/host_sections.py --full-layout --window-open > $out/host_sections_open_window.txt 2>&1
DPC="python bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --reserve-gb 2"
timeout 600 $DPC --compact-first-grid --steady-after 0 > $out/dp2_narrow_compact.json 2> $out/dp2_narrow_compact.err
timeout 900 $DPC --window-hash 0 1 --steady-after 200 > $out/dp2_level.json 2> $out/dp2_level.err
# HBM counters of the level-parallel rank's step (the optimizer pass over 192 planes, the per-source-rank backward)
p=gpurun_out/prof_r06; mkdir -p $p
PM="$LP --steady-after 0 --no-kernel-events --steps 6 --warmup 2"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $p/lp_fetch -o lp -- $PM > /dev/null 2> $p/lp_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $p/lp_write -o lp -- $PM > /dev/null 2> $p/lp_write.err
PMC_LAST_DISPATCHES=6 python tools/pmc_to_json.py $p/lp_fetch $p/lp_write $p/r06_level_parallel_rank7.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $PM" > $p/lp_summary.txt 2>&1
find $p/lp_fetch $p/lp_write \( -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
python - <<'P'
import json
def line(f):
    return json.loads([l for l in open(f"gpurun_out/final_r06/{f}.json") if l.startswith("{")][-1])
for f in ("lp8_rank7", "lp8_rank0"):
    try:
        d=line(f); ss=d.get("steady_state") or {}
        print(f, "window", round(d["ms_per_step"],3), "steady", {k: ss.get(k) for k in ("ms_per_step","host_issue_ms_per_step","host_issue_ms_per_step_min","samples_per_step_min_max")})
        print("   steady comm", json.dumps(ss.get("comm"))[:900])
        print("   window kernels", {k: v["avg_ms"] for k, v in list(d["native_kernel_ms"].items())[:8]})
    except Exception as e: print(f, "ERR", e)
for f in ("dp_one_rank_full", "dp_one_rank_compact"):
    try:
        d=line(f); ss=d.get("steady_state") or {}
        print(f, "window", round(d["ms_per_step"],3), "steady", round(ss.get("ms_per_step", 0),3))
    except Exception as e: print(f, "ERR", e)
for f in ("dp2_narrow_compact", "dp2_level"):
    try:
        d=line(f)
        print(f, round(d["ms_per_step"], 2), json.dumps(d.get("comm"))[:900]); print(f, "steady", json.dumps(d.get("steady_state"))[:900])
    except Exception as e: print(f, "ERR", e)
P
head -14 $out/host_sections_lp8.txt; head -6 $out/host_sections_compact.txt; head -6 $out/host_sections_open_window.txt; head -14 $p/lp_summary.txt
fi

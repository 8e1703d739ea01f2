# Round-4 evidence on ONE box.  part A: smoke, the whole -m gpu suite, the driver's bench line, rocprofv3 kernel statistics
# + the two HBM counter passes of it.  part B: steady-state device timelines (full layout, compact first-grid phase, with
# the datamanager in the loop), kernel statistics of one evaluation image, the other BASELINE configurations, host issue
# time by section, the 2-rank control flow with its comm block (gloo, both ranks on cuda:0), the window-ramp layouts.
#   usage: bash tools/final_run_r04.sh A|B        results: gpurun_out/final_r04/ (+ gpurun_out/prof_r04/)
set -u
part=${1:-A}
out=gpurun_out/final_r04; mkdir -p $out
export TMPDIR=/tmp
if [ "$part" = A ]; then
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -12 > $out/full_suite.txt
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
bash tools/collect_profiles.sh r04 > $out/collect.log 2>&1
tail -2 $out/smoke.txt; tail -4 $out/full_suite.txt
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/final_r04/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("datamanager"), v.get("error"))
P
else
python -m pytest tests/test_training_gpu.py -q -k "ramp or handed" 2>&1 | tail -2 > $out/ramp_tests.txt
DM="python bench.py --with-datamanager --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
$DM > $out/bench_with_datamanager.json 2>/dev/null
$DM --compact-first-grid > $out/bench_with_datamanager_compact.json 2>/dev/null
tl=$out/tl; mkdir -p $tl
for mode in full compact datamanager; do
  flags=""; [ $mode = compact ] && flags="--compact"; [ $mode = datamanager ] && flags="--compact --datamanager"
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $tl/$mode -o tl -- python tools/host_profile.py --plain --steps 30 $flags > $tl/$mode.out 2> $tl/$mode.err
  python tools/timeline.py $tl/$mode 20 > $out/timeline_steady_$mode.txt 2>&1
  find $tl/$mode \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*memory_copy_trace.csv" \) -delete
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/eval -o eval -- python tools/eval_bench.py > $out/eval_bench.txt 2> $out/eval.err
find $out/eval -name "*kernel_stats.csv" -exec cp {} $out/eval_image_kernel_stats.csv \;
find $out/eval \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
bash tools/config_lines.sh > $out/config_lines.txt 2>&1
python tools/host_sections.py > $out/host_sections_compact.txt 2>&1
python tools/host_sections.py --full-layout > $out/host_sections_full.txt 2>&1
DPC="python bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --steady-after 0 --reserve-gb 2"
timeout 600 $DPC > $out/dp2_weak.json 2> $out/dp2_weak.err
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
for w in "-7000 80000" "-44000 80000"; do
  n=$(echo $w | tr " -" "_m")
  $B --window-hash $w --compact-first-grid > $out/ramp_compact_$n.json 2>/dev/null
  $B --window-hash $w > $out/ramp_full_$n.json 2>/dev/null
done
cat $out/ramp_tests.txt
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/final_r04/ramp_*.json")) + sorted(glob.glob("gpurun_out/final_r04/bench_with_datamanager*.json")) + ["gpurun_out/final_r04/dp2_weak.json"]:
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, round(d["ms_per_step"],3), (d.get("steady_state") or {}).get("ms_per_step"), d.get("comm"), {k: v for k, v in (d.get("datamanager") or {}).items() if k.startswith("next_train_host")})
    except Exception as e: print(f, "ERR", e)
P
cat $out/config_lines.txt | tail -5; cat $out/host_sections_compact.txt | head -8; tail -3 $out/eval_bench.txt
fi

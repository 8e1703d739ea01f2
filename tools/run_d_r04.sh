#!/bin/bash
# round 4, last evidence pass (final state of the round) on ONE box: smoke, the whole -m gpu suite, the driver's bench line,
# rocprofv3 kernel statistics + the two HBM counter passes of it, steady-state timelines (full layout / compact first-grid
# phase), host issue time by section, the window-ramp lines.      results: gpurun_out/final_r04d/ (+ gpurun_out/prof_r04d/)
set -u
export TMPDIR=/tmp
out=gpurun_out/final_r04d; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -12 > $out/full_suite.txt
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
bash tools/collect_profiles.sh r04d > $out/collect.log 2>&1
tl=$out/tl; mkdir -p $tl
for mode in full compact; do
  flags=""; [ $mode = compact ] && flags="--compact"
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $tl/$mode -o tl -- python tools/host_profile.py --plain --steps 30 $flags > $tl/$mode.out 2> $tl/$mode.err
  python tools/timeline.py $tl/$mode 20 > $out/timeline_steady_$mode.txt 2>&1
  find $tl/$mode \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*memory_copy_trace.csv" \) -delete
done
python tools/host_sections.py > $out/host_sections_compact.txt 2>&1
python tools/host_sections.py --full-layout > $out/host_sections_full.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
$B --window-hash -7000 80000 --compact-first-grid > $out/ramp_compact_m7000_80000.json 2>/dev/null
$B --workload static_h1 > $out/static_h1.json 2>/dev/null
tail -2 $out/smoke.txt; tail -4 $out/full_suite.txt
python - <<'P'
import json, glob
d=json.loads([l for l in open("gpurun_out/final_r04d/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("error"))
for k, v in ((d.get("kernels_alone") or {}).get("kernels") or {}).items():
    if "deform" in k: print("   alone", k, v["ms"], v["frac"])
print({k: v["avg_ms"] for k, v in (d.get("native_kernel_ms") or {}).items()})
for f in ("ramp_compact_m7000_80000", "static_h1"):
    try:
        e=json.loads([l for l in open(f"gpurun_out/final_r04d/{f}.json") if l.startswith("{")][-1]); print(f, round(e["ms_per_step"],3), (e.get("steady_state") or {}).get("ms_per_step"))
    except Exception as ex: print(f, "ERR", ex)
P
head -4 $out/timeline_steady_full.txt; head -4 $out/timeline_steady_compact.txt; head -8 $out/host_sections_compact.txt; head -4 $out/host_sections_full.txt; cat gpurun_out/prof_r04d/pmc_summary.txt; grep -E "deform_bwd|deform_wgrad|deform_code|adam_hash" gpurun_out/prof_r04d/r04d_kernel_stats.csv | awk -F, '{print substr($1,1,60), $2, $4}'

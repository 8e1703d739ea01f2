#!/bin/bash
# Round 5: the driver's bench line and the steady-state device timelines after the sampler's tail became ray-wise.
set -u
out=gpurun_out/r05_j; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=.
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone > $out/bench.json 2> $out/bench.err
tl=$out/tl; mkdir -p $tl
for mode in full compact; do
  flags=""; [ $mode = compact ] && flags="--compact"
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $tl/$mode -o tl -- python tools/host_profile.py --plain --steps 30 $flags > $tl/$mode.out 2> $tl/$mode.err
  python tools/timeline.py $tl/$mode 20 > $out/timeline_steady_$mode.txt 2>&1
  find $tl/$mode \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*memory_copy_trace.csv" \) -delete
done
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r05_j/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3))
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("error"))
P
head -3 $out/timeline_steady_full.txt; head -3 $out/timeline_steady_compact.txt

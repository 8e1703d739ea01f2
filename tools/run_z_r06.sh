#!/bin/bash
# Round 6, run Z (last tree): one rank of a data-parallel job (closed window; full / compact layouts) after nsx_multi_adam_present
# entered the step, the 2-rank control flows through gloo on one GPU (narrow compact exchange; level-parallel hand-over), and the
# single GPU in the compact layout on the same box.
set -u
out=gpurun_out/r06_z; mkdir -p $out
export TMPDIR=/tmp
for mode in full compact; do
  flags=""; [ $mode = compact ] && flags="--compact-first-grid"
  timeout 600 python bench.py --sharded-one-rank $flags --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager > $out/dp_one_rank_$mode.json 2> $out/dp_one_rank_$mode.err
done
timeout 600 python bench.py --compact-first-grid --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager > $out/single_compact.json 2> $out/single_compact.err
DPC="python bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --reserve-gb 2"
timeout 600 $DPC --compact-first-grid --steady-after 0 > $out/dp2_narrow_compact.json 2> $out/dp2_narrow_compact.err
timeout 900 $DPC --window-hash 0 1 --steady-after 200 > $out/dp2_level.json 2> $out/dp2_level.err
python - <<'P'
import json
def line(f):
    return json.loads([l for l in open(f"gpurun_out/r06_z/{f}.json") if l.startswith("{")][-1])
for f in ("dp_one_rank_full", "dp_one_rank_compact", "single_compact", "dp2_narrow_compact", "dp2_level"):
    try:
        d = line(f); ss = d.get("steady_state") or {}
        if isinstance(ss, list): ss = ss[0] if ss else {}
        print(f, "window", round(d["ms_per_step"], 3), "steady", ss.get("ms_per_step"), "host", ss.get("host_issue_ms_per_step"), "psnr", ss.get("psnr"), "comm", json.dumps(d.get("comm") or ss.get("comm"))[:300])
    except Exception as e:
        print(f, "ERR", repr(e))
P

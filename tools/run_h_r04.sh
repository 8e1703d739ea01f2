#!/bin/bash
# round 4, the matched evidence set of the LAST code state on one box: the driver's line, rocprofv3 kernel statistics and the
# two HBM counter passes of the same command, the SQ counters of the MFMA kernels
set -u
export TMPDIR=/tmp
out=gpurun_out/h_r04; mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
bash tools/collect_profiles.sh r04h > $out/collect.log 2>&1
bash tools/sq_counters.sh r04h > $out/sq.log 2>&1
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/h_r04/bench.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k in ("first_grid_phase","open_window","with_datamanager"):
    v=d.get(k,{}); print(k, v.get("ms_per_step"), (v.get("steady_state") or {}).get("ms_per_step"), v.get("error"))
for k, v in d["kernels_alone"]["kernels"].items(): print("  alone", k, v["ms"], v["frac"])
P
cat gpurun_out/prof_r04h/pmc_summary.txt; grep -E "deform_bwd|deform_wgrad|deform_code|adam_hash" gpurun_out/prof_r04h/r04h_kernel_stats.csv | awk -F, '{print substr($1,1,60), $2, $4}'
grep -E "mfma_busy_frac|^  \"nsx" gpurun_out/sq_r04h/r04h_sq_mfma_kernels.json | head -40

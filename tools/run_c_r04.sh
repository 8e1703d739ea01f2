#!/bin/bash
# round 4, third evidence pass: head gradients of the deformation field formed in the chain kernel (92 KB scratch per tile),
# separate scatter for <= 4 grids, one load per corner in the pre-blended eval lookup
set -u
export TMPDIR=/tmp
out=gpurun_out/c_r04; mkdir -p $out
python -m pytest tests/test_deform_gpu.py tests/test_native_step_gpu.py tests/test_training_gpu.py tests/test_hash_ensemble_gpu.py tests/test_field_gpu.py tests/test_image_parity_gpu.py -q -m gpu -x 2>&1 | tail -8 > $out/tests.txt
cat $out/tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-first-grid-phase --no-open-window --no-with-datamanager > $out/bench.json 2> $out/bench.err
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
$B --window-hash -7000 80000 --compact-first-grid > $out/ramp_compact_m7000_80000.json 2>/dev/null
$B --window-hash -2500 80000 --compact-first-grid > $out/ramp_compact_m2500_80000.json 2>/dev/null
$B --window-hash -2500 80000 > $out/ramp_full_m2500_80000.json 2>/dev/null
python tools/eval_bench.py > $out/eval_bench.txt 2> $out/eval.err
bash tools/collect_profiles.sh r04c > $out/collect.log 2>&1
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/c_r04/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"],3), (d.get("steady_state") or {}).get("ms_per_step"), {k: v["avg_ms"] for k, v in (d.get("native_kernel_ms") or {}).items() if "deform_bwd" in k or "bwd_" in k})
        ka = (d.get("kernels_alone") or {}).get("kernels") or {}
        for k, v in ka.items():
            if "deform" in k: print("   alone", k, v["ms"], v["frac"])
    except Exception as e: print(f, "ERR", e)
P
tail -4 $out/eval_bench.txt; cat gpurun_out/prof_r04c/pmc_summary.txt

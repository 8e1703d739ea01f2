# On the GPU box: one bench line per BASELINE.json configuration that runs on one GPU (configs[1], [3], [4] + the static one),
# without the blocks that belong to the headline workload.  Results: gpurun_out/configs/<workload>.json
set -u
out=gpurun_out/configs; mkdir -p $out
for w in p030_h16 p097_dense p124_dp static_h1; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --steady-after 0 > $out/$w.json 2> $out/$w.err
  python - "$out/$w.json" "$w" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], round(d["ms_per_step"], 3), "ms/step", round(d["value"] / 1e6, 2), "M samples/s", d["samples_per_step_min_max"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
done

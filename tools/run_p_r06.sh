#!/bin/bash
# Round 6, run P: gradient planes of the compact first-grid phase (HashEnsemble.first_grid_planes): parity test + a sweep
set -u
out=gpurun_out/r06_p; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_training_gpu.py -q -m gpu -x -k "first_grid" 2>&1 | grep -v "amdgpu.ids" | tail -5 > $out/tests.txt
cat $out/tests.txt
B="--compact-first-grid --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
for P in 0 1 2 4 8; do
  export NSX_FIRST_GRID_PLANES=$P
  timeout 600 python bench.py $B > $out/single_P$P.json 2> $out/single_P$P.err
  timeout 600 python bench.py --sharded-one-rank $B > $out/dp_P$P.json 2> $out/dp_P$P.err
done
python - <<'P'
import json
for P in (0, 1, 2, 4, 8):
    for f in ("single", "dp"):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r06_p/{f}_P{P}.json") if l.startswith("{")][-1])
            k = d["native_kernel_ms"]; ks = d["steady_state"].get("native_kernel_ms") or {}
            print(P, f, "window", round(d["ms_per_step"], 3), "steady", round(d["steady_state"]["ms_per_step"], 3),
                  {n: round(v["avg_ms"], 3) for n, v in k.items() if "scatter" in n or "adam" in n or "bwd_factored" in n})
        except Exception as e:
            print(P, f, "failed", repr(e))
P

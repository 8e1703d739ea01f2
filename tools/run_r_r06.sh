#!/bin/bash
# Round 6, run R: the counting pass keeps the samples' starts (nsx_march_count_stash): parity tests, then A/B on one box
set -u
out=gpurun_out/r06_r; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_march_gpu.py tests/test_native_step_gpu.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids" | tail -5 > $out/tests.txt
cat $out/tests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
for st in 0 1; do
  export NSX_MARCH_STASH=$st
  timeout 600 python bench.py $B > $out/full_stash$st.json 2> $out/full_stash$st.err
  timeout 600 python bench.py --compact-first-grid $B > $out/compact_stash$st.json 2> $out/compact_stash$st.err
done
python - <<'P'
import json
for st in (0, 1):
    for f in ("full", "compact"):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r06_r/{f}_stash{st}.json") if l.startswith("{")][-1])
            k = d["native_kernel_ms"]
            print(st, f, "window", round(d["ms_per_step"], 3), "steady", round(d["steady_state"]["ms_per_step"], 3),
                  {n: (v["calls"], round(v["avg_ms"], 3)) for n, v in k.items() if "march" in n}, "psnr", d.get("psnr_last"), d["steady_state"].get("psnr"))
        except Exception as e:
            print(st, f, "failed", repr(e))
P

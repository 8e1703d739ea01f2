#!/bin/bash
# Round 6, run U: the level-parallel optimizer pass leaves some CUs free (NSX_ADAM_MFMA_CUS_FREE): emulated rank 7 / rank 5, A/B on one box
set -u
out=gpurun_out/r06_u; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_adam_gpu.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -3 > $out/tests.txt; cat $out/tests.txt
LP="python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
for free in 0 24 48 96; do
  export NSX_ADAM_MFMA_CUS_FREE=$free
  for r in 7 5; do
    timeout 400 $LP --rank $r > $out/lp8_rank${r}_free$free.json 2> $out/lp8_rank${r}_free$free.err
  done
done
python - <<'P'
import json
for free in (0, 24, 48, 96):
    for r in (7, 5):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r06_u/lp8_rank{r}_free{free}.json") if l.startswith("{")][-1])
            ss = d["steady_state"]; c = ss["comm"]
            print(free, r, "window", round(d["ms_per_step"], 3), "steady", round(ss["ms_per_step"], 3), "adam", round(c["shard_adam_ms"], 3),
                  "window adam", d["native_kernel_ms"].get("nsx_adam_hash_factored", {}).get("avg_ms"))
        except Exception as e:
            print(free, r, "failed", repr(e))
P

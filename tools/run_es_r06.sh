#!/bin/bash
# Round 6 experiment: the level-parallel table optimizer pass started right after the backward exchange (timing stand-in,
# tools/experiments/lp_early_table_step.py) against the step as it is; emulated ranks 7 and 1 of 8, interleaved on one box.
set -u
out=gpurun_out/r06_es; mkdir -p $out
export TMPDIR=/tmp
A="--level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
for r in 7 1; do for i in 1 2; do
  timeout 300 python bench.py $A --rank $r > $out/base_r${r}_$i.json 2> $out/base_r${r}_$i.err
  timeout 300 python tools/experiments/lp_early_table_step.py $A --rank $r > $out/early_r${r}_$i.json 2> $out/early_r${r}_$i.err
done; done
python - <<'P'
import json
for r in (7, 1):
    for f in ("base", "early"):
        for i in (1, 2):
            try:
                d = json.loads([l for l in open(f"gpurun_out/r06_es/{f}_r{r}_{i}.json") if l.startswith("{")][-1]); ss = d["steady_state"]
                print(r, f, i, "window", round(d["ms_per_step"], 3), "steady", round(ss["ms_per_step"], 3), "host", round(ss.get("host_issue_ms_per_step") or 0, 3), "adam", round((ss.get("comm") or {}).get("shard_adam_ms", 0), 3))
            except Exception as e:
                print(r, f, i, "ERR", repr(e))
P
tail -3 $out/early_r7_1.err

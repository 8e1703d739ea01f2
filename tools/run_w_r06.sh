#!/bin/bash
# Round 6, run W: run-to-run spread of the headline line on ONE box, alternating NSX_MARCH_STASH=0/1 (does keeping the samples cost
# the full-layout window anything?)
set -u
out=gpurun_out/r06_w; mkdir -p $out
export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager --steady-after 0"
for i in 1 2 3 4; do
  for st in 0 1; do
    NSX_MARCH_STASH=$st timeout 300 python bench.py $B > $out/full_stash${st}_$i.json 2> $out/full_stash${st}_$i.err
  done
done
python - <<'P'
import json
for st in (0, 1):
    v = []
    for i in (1, 2, 3, 4):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r06_w/full_stash{st}_{i}.json") if l.startswith("{")][-1])
            v.append((round(d["ms_per_step"], 3), d["roofline"]["frac"]))
        except Exception as e:
            v.append(repr(e))
    print("NSX_MARCH_STASH =", st, v)
P

#!/bin/bash
# round 5, run C2: the level-parallel tests + the 2-rank bench line again (after the gloo all-gather shape fix)
set -u
out=gpurun_out/r05_c; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gpu.py -q -m gpu -k "level or handed" 2>&1 | tail -40 > $out/tests_lp.txt
tail -25 $out/tests_lp.txt
DPC="python bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --reserve-gb 2 --window-hash 0 1"
timeout 900 $DPC --steady-after 200 > $out/dp2_level.json 2> $out/dp2_level.err
python - <<'P'
import json
for f in ("dp2_level",):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r05_c/{f}.json") if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"], 2), json.dumps(d.get("comm"))[:1500])
        print(f, "steady", json.dumps(d.get("steady_state"))[:1800])
    except Exception as e:
        print(f, "ERR", e); print(open(f"gpurun_out/r05_c/{f}.err").read()[-3000:])
P

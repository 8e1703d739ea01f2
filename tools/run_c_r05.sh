#!/bin/bash
# round 5, run C: the level-parallel exchange -- two ranks on one GPU (gloo) against the single process on the union batch, the
# hand-over from the reduce-scatter exchange, the other data-parallel tests, and the 2-rank bench line with its comm block
set -u
out=gpurun_out/r05_c; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_bench_launch_gpu.py -q -m gpu -x 2>&1 | tail -40 > $out/tests.txt
tail -15 $out/tests.txt
DPC="python bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --reserve-gb 2 --window-hash 0 1"
timeout 900 $DPC --steady-after 200 > $out/dp2_level.json 2> $out/dp2_level.err
timeout 600 $DPC --steady-after 0 --table-parallel shard > $out/dp2_shard.json 2> $out/dp2_shard.err
python - <<'P'
import json
for f in ("dp2_level", "dp2_shard"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r05_c/{f}.json") if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"], 2), json.dumps(d.get("comm"))[:900])
        print(f, "steady", json.dumps(d.get("steady_state"))[:1200])
    except Exception as e:
        print(f, "ERR", e); print(open(f"gpurun_out/r05_c/{f}.err").read()[-2500:])
P

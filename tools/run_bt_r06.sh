#!/bin/bash
# Round 6: the small gradients' bucket tail in one launch each way (nsx_bucket_pack / _unpack) against one copy per piece, same box:
# the emulated level-parallel rank 7 of 8 and one rank of a data-parallel job in the compact layout, two runs each, interleaved.
set -u
out=gpurun_out/r06_bt; mkdir -p $out
export TMPDIR=/tmp
LP="python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
DP="python bench.py --sharded-one-rank --compact-first-grid --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
for i in 1 2; do
  timeout 300 $LP > $out/lp_native_$i.json 2> $out/lp_native_$i.err
  timeout 300 $LP --bucket-tail-copies > $out/lp_copies_$i.json 2> $out/lp_copies_$i.err
  timeout 300 $DP > $out/dp_native_$i.json 2> $out/dp_native_$i.err
  timeout 300 $DP --bucket-tail-copies > $out/dp_copies_$i.json 2> $out/dp_copies_$i.err
done
python - <<'P'
import json
for f in ("lp_native", "lp_copies", "dp_native", "dp_copies"):
    for i in (1, 2):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r06_bt/{f}_{i}.json") if l.startswith("{")][-1]); ss = d["steady_state"]
            print(f, i, "window", round(d["ms_per_step"], 3), "steady", round(ss["ms_per_step"], 3), "host", ss.get("host_issue_ms_per_step"))
        except Exception as e:
            print(f, i, "ERR", repr(e))
P

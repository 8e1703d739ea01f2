#!/bin/bash
# Round 6, run N: the data-parallel rank after its expansion moved to the optimizer stream -- tests, bench lines, timeline
set -u
out=gpurun_out/r06_n; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gpu.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -5 > $out/tests.txt
for mode in compact full; do
  flags=""; [ $mode = compact ] && flags="--compact-first-grid"
  timeout 600 python bench.py --sharded-one-rank $flags --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager > $out/dp_one_rank_$mode.json 2> $out/dp_one_rank_$mode.err
done
timeout 600 python bench.py --compact-first-grid --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager > $out/single_compact.json 2> $out/single_compact.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/dp -o tl -- python $GRAFT_REPO_ROOT/tools/host_profile.py --plain --steps 30 --compact --sharded-one-rank > $GRAFT_REPO_ROOT/$out/dp.out 2> $GRAFT_REPO_ROOT/$out/dp.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $out/dp 16 nsx::adam_dense_f16grad_kernel > $out/timeline_dp_compact.txt 2>&1
rm -rf $out/dp
cat $out/tests.txt
python - <<'P'
import json
for f in ("dp_one_rank_compact", "dp_one_rank_full", "single_compact"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r06_n/{f}.json") if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"], 3), round(d["steady_state"]["ms_per_step"], 3), (d["steady_state"].get("comm") or d.get("comm") or {}).get("reduce_scatter_exposed_ms"))
    except Exception as e:
        print(f, "failed", repr(e))
P
head -3 $out/timeline_dp_compact.txt

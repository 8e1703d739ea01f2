#!/bin/bash
# Round 6, run O: GPU_MAX_HW_QUEUES -- does the data-parallel rank's expansion get a hardware queue of its own with 8?  A/B on one box
set -u
out=gpurun_out/r06_o; mkdir -p $out
export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-with-datamanager"
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  timeout 600 python bench.py --sharded-one-rank --compact-first-grid $B > $out/dp_compact_q$q.json 2> $out/dp_compact_q$q.err
  timeout 600 python bench.py --compact-first-grid $B > $out/single_compact_q$q.json 2> $out/single_compact_q$q.err
  timeout 600 python bench.py $B > $out/single_full_q$q.json 2> $out/single_full_q$q.err
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/dp$q -o tl -- python $GRAFT_REPO_ROOT/tools/host_profile.py --plain --steps 30 --compact --sharded-one-rank > $GRAFT_REPO_ROOT/$out/dp$q.out 2> $GRAFT_REPO_ROOT/$out/dp$q.err
  cd $GRAFT_REPO_ROOT
  python tools/timeline.py $out/dp$q 16 nsx::adam_dense_f16grad_kernel > $out/timeline_dp_compact_q$q.txt 2>&1
  rm -rf $out/dp$q
done
python - <<'P'
import json
for q in (4, 8):
    for f in ("dp_compact", "single_compact", "single_full"):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r06_o/{f}_q{q}.json") if l.startswith("{")][-1])
            print(q, f, round(d["ms_per_step"], 3), round(d["steady_state"]["ms_per_step"], 3))
        except Exception as e:
            print(q, f, "failed", repr(e))
P
head -2 $out/timeline_dp_compact_q4.txt; head -2 $out/timeline_dp_compact_q8.txt; grep "expand_f16_narrow\|adam_dense" $out/timeline_dp_compact_q8.txt | head -4

#!/bin/bash
# Round 6, run M: device timelines of the compact first-grid phase in steady state -- a data-parallel rank against the single GPU
set -u
out=gpurun_out/r06_m; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for mode in dp single; do
  flags="--compact"; [ $mode = dp ] && flags="--compact --sharded-one-rank"
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/$mode -o tl -- python $GRAFT_REPO_ROOT/tools/host_profile.py --plain --steps 30 $flags > $GRAFT_REPO_ROOT/$out/$mode.out 2> $GRAFT_REPO_ROOT/$out/$mode.err
done
cd $GRAFT_REPO_ROOT
python tools/timeline.py $out/dp 16 nsx::adam_dense_f16grad_kernel > $out/timeline_dp_compact.txt 2>&1
python tools/timeline.py $out/single 16 > $out/timeline_single_compact.txt 2>&1
rm -rf $out/dp $out/single
head -4 $out/timeline_dp_compact.txt; head -4 $out/timeline_single_compact.txt; tail -3 $out/dp.out $out/single.out 2>/dev/null | grep -v amdgpu

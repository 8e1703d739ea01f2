#!/bin/bash
# Round 6, run S (last state of the tree, one box): final run part B, every emulated level-parallel rank, and the compact phase's timeline
set -u
export TMPDIR=/tmp
bash tools/final_run_r06.sh B > gpurun_out/final_r06_b.log 2>&1
bash tools/run_i_r06.sh > gpurun_out/run_i.log 2>&1
out=gpurun_out/r06_s; mkdir -p $out
cd /tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/single -o tl -- python $GRAFT_REPO_ROOT/tools/host_profile.py --plain --steps 30 --compact > $GRAFT_REPO_ROOT/$out/single.out 2> $GRAFT_REPO_ROOT/$out/single.err
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/dp -o tl -- python $GRAFT_REPO_ROOT/tools/host_profile.py --plain --steps 30 --compact --sharded-one-rank > $GRAFT_REPO_ROOT/$out/dp.out 2> $GRAFT_REPO_ROOT/$out/dp.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $out/single 16 > $out/timeline_single_compact.txt 2>&1
python tools/timeline.py $out/dp 16 nsx::adam_dense_f16grad_kernel > $out/timeline_dp_compact.txt 2>&1
rm -rf $out/single $out/dp
tail -30 gpurun_out/final_r06_b.log | cut -c1-400; tail -12 gpurun_out/run_i.log | cut -c1-600; head -3 $out/timeline_single_compact.txt; head -3 $out/timeline_dp_compact.txt

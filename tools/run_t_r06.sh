#!/bin/bash
# Round 6, run T: device timeline of an emulated level-parallel rank's steady-state step (rank 7 of 8)
set -u
out=gpurun_out/r06_t; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/lp -o tl -- python $GRAFT_REPO_ROOT/tools/host_profile.py --plain --steps 30 --level-parallel-one-rank 8 > $GRAFT_REPO_ROOT/$out/lp.out 2> $GRAFT_REPO_ROOT/$out/lp.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $out/lp 16 nsx::adam_hash_factored_mfma_kernel > $out/timeline_lp8_rank7.txt 2>&1
rm -rf $out/lp
head -90 $out/timeline_lp8_rank7.txt | cut -c1-120; tail -3 $out/lp.out

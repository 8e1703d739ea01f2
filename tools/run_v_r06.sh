#!/bin/bash
# Round 6, run V: after "all source ranks in one launch": device timeline of the emulated level-parallel rank's steady-state step
# (rank 7 of 8) and the host's issue time by section
set -u
out=gpurun_out/${RUN_V_OUT:-r06_v}; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/lp -o tl -- python $GRAFT_REPO_ROOT/tools/host_profile.py --plain --steps 30 --level-parallel-one-rank 8 > $GRAFT_REPO_ROOT/$out/lp.out 2> $GRAFT_REPO_ROOT/$out/lp.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $out/lp 16 nsx::adam_hash_factored_mfma_kernel > $out/timeline_lp8_rank7.txt 2>&1
rm -rf $out/lp
timeout 400 python tools/host_sections.py --level-parallel-one-rank 8 --fine > $out/host_sections_fine_lp8.txt 2>&1
head -100 $out/timeline_lp8_rank7.txt | cut -c1-120; tail -3 $out/lp.out; head -40 $out/host_sections_fine_lp8.txt

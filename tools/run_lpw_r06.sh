#!/bin/bash
# Round 6: kernel statistics of the emulated level-parallel rank's BENCHMARK WINDOW (fresh model, ~350 k marched samples per rank)
set -u
out=gpurun_out/${RUN_LPW_OUT:-r06_lpw}; mkdir -p $out
export TMPDIR=/tmp
R=${RUN_LPW_RANK:-7}
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/tr -o lpw -- python $GRAFT_REPO_ROOT/bench.py --level-parallel-one-rank 8 --rank $R --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone --steady-after 0 --no-kernel-events ${RUN_LPW_FLAGS:-} > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/bench.err
cd $GRAFT_REPO_ROOT
find $out/tr -name "*kernel_stats.csv" -exec cp {} $out/lpw_rank${R}_kernel_stats.csv \;
rm -rf $out/tr
head -14 $out/lpw_rank${R}_kernel_stats.csv | cut -c1-200

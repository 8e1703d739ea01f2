#!/bin/bash
# Round 6, run Q: HBM counters of an emulated level-parallel rank's step (the part of final run B that lacked its directory)
set -u
export TMPDIR=/tmp
p=gpurun_out/${RUN_Q_OUT:-prof_r06}; mkdir -p $p
LP="python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
PM="$LP --steady-after 0 --no-kernel-events --steps 6 --warmup 2"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $p/lp_fetch -o lp -- $PM > /dev/null 2> $p/lp_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $p/lp_write -o lp -- $PM > /dev/null 2> $p/lp_write.err
PMC_LAST_DISPATCHES=6 python tools/pmc_to_json.py $p/lp_fetch $p/lp_write $p/r06_level_parallel_rank7.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $PM" > $p/lp_summary.txt 2>&1
find $p/lp_fetch $p/lp_write \( -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
head -20 $p/lp_summary.txt

#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace statistics of the default bench command + the two HBM counter passes
# (separate runs: gpurun refuses --pmc together with tracing).  Results land in gpurun_out/prof_<tag>/ ; copy the
# summaries into profiles/ afterwards.   usage: tools/collect_profiles.sh r02
set -u
tag=${1:-r05}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-after 0 --no-kernels-alone --no-first-grid-phase --no-open-window"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o $tag -- $BENCH > $out/bench_trace.json 2> $out/trace.err
PMC="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --steady-after 0 --no-kernels-alone --no-first-grid-phase --no-open-window"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o $tag -- $PMC > /dev/null 2> $out/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o $tag -- $PMC > /dev/null 2> $out/write.err
python tools/pmc_to_json.py $out/fetch $out/write $out/pmc_$tag.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $PMC" > $out/pmc_summary.txt 2>&1
find $out -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
find $out \( -name "*counter_collection.csv" -o -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
ls -la $out; cat $out/pmc_summary.txt | head -20; head -12 $out/${tag}_kernel_stats.csv

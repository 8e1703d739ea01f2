# On the GPU box: device timelines (tools/timeline.py) of the steady state (--preroll 600) and of the first steps from
# rocprofv3 kernel traces, an A/B of the early table step, and the two HBM counter passes.  Results: gpurun_out/tl/, gpurun_out/prof_r02/
set -u
export TMPDIR=/tmp
out=gpurun_out/tl
mkdir -p $out
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --steady-after 0 --no-kernels-alone"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/steady -o tl -- $B --preroll 600 > $out/steady_bench.json 2> $out/steady.err
python tools/timeline.py $out/steady 10 > $out/timeline_steady.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/early -o tl -- $B > $out/early_bench.json 2> $out/early.err
python tools/timeline.py $out/early 10 > $out/timeline_early.txt 2>&1
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
NSX_EARLY_TABLE_STEP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone > $out/bench_early_step.json 2> $out/bench_early_step.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone > $out/bench_default.json 2> $out/bench_default.err
PMC="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --steady-after 0 --no-kernels-alone"
p=gpurun_out/prof_r02; mkdir -p $p
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $p/fetch -o r02 -- $PMC > /dev/null 2> $p/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $p/write -o r02 -- $PMC > /dev/null 2> $p/write.err
python tools/pmc_to_json.py $p/fetch $p/write $p/pmc_r02.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $PMC" > $p/pmc_summary.txt 2>&1
find $p \( -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
head -5 $out/timeline_steady.txt; cat $p/pmc_summary.txt

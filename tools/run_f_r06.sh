#!/bin/bash
# Round 6, run F: the evaluation image un-chunked (nsx_density_fused_fwd slices its launches itself) + counter passes that say
# what bounds nsx_density_fused_fwd (rocprofv3 --pmc, counters only, one pass per block of counters)
set -u
out=gpurun_out/r06_f; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python tools/eval_bench.py --price > $out/eval_bench.txt 2> $out/eval.err; grep -a "preblend=" $out/eval_bench.txt; tail -1 $out/eval_bench.txt | cut -c1-900
RUN="python tools/eval_bench.py"
timeout 500 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $out/passA -o e -- $RUN > $out/passA.out 2> $out/passA.err
timeout 500 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum --output-format csv -d $out/passB -o e -- $RUN > $out/passB.out 2> $out/passB.err
timeout 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/passC -o e -- $RUN > $out/passC.out 2> $out/passC.err
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU --output-format csv -d $out/passD -o e -- $RUN > $out/passD.out 2> $out/passD.err
python tools/eval_counters.py $out $out/r06_density_fused_counters.json
tail -2 $out/passA.err $out/passB.err $out/passC.err $out/passD.err | cut -c1-300
find $out \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*counter_collection.csv" \) -delete

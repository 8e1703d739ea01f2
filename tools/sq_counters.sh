#!/bin/bash
# On the GPU box: SQ counter passes (rocprofv3 --pmc, counters only -- no tracing beside them) of the MFMA kernels alone at
# S = 2^20 (tools/mfma_bench.py, one warm-up + one measured launch each).  Results: gpurun_out/sq_<tag>/ ; the JSON summary
# is what gets copied to profiles/pmc/.      usage: tools/sq_counters.sh r04
set -u
tag=${1:-r05}
out=gpurun_out/sq_$tag
mkdir -p $out
export TMPDIR=/tmp
RUN="python tools/mfma_bench.py --iters 2 --warmup 1"
python tools/mfma_bench.py --iters 10 --warmup 3 > $out/mfma_bench.json 2> $out/mfma_bench.err
# pass A: matrix-core busy, SQ busy, wave cycles and their three disjoint buckets, VALU instructions, LDS bank conflicts (8 SQ slots)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $out/passA -o $tag -- $RUN > $out/passA.out 2> $out/passA.err
# pass B: LDS side + MFMA instruction count + the GRBM clock (independent block)
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE \
    --output-format csv -d $out/passB -o $tag -- $RUN > $out/passB.out 2> $out/passB.err
python tools/sq_to_json.py $out $out/${tag}_sq_mfma_kernels.json > $out/summary.txt 2>&1
find $out \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
cat $out/summary.txt | head -60; tail -3 $out/passA.err $out/passB.err

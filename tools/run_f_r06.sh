#!/bin/bash
# Round 6, run F: the evaluation image un-chunked + a counter pass that says what bounds nsx_density_fused_fwd
set -u
out=gpurun_out/r06_f; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python tools/eval_bench.py --price > $out/eval_bench.txt 2> $out/eval.err; grep -a "preblend=" $out/eval_bench.txt; tail -1 $out/eval_bench.txt | cut -c1-1500
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*\|TCC_HIT[A-Z_a-z0-9]*\|TCC_MISS[A-Z_a-z0-9]*\|TCC_REQ[A-Z_a-z0-9]*\|SQ_INSTS_VMEM[A-Z_a-z0-9]*\|SQ_WAIT[A-Z_a-z0-9]*\|TD_[A-Z_0-9a-z]*" | sort -u > $out/counters_available.txt
wc -l $out/counters_available.txt; head -100 $out/counters_available.txt | tr '\n' ' '

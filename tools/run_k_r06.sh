#!/bin/bash
# Round 6, run K: where the data-parallel rank's step differs from the single GPU's -- host sections and a steady-state timeline
set -u
out=gpurun_out/r06_k; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python tools/host_sections.py --sharded-one-rank 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|ProcessGroupNCCL" > $out/host_sections_dp_compact.txt
timeout 400 python tools/host_sections.py 2>&1 | grep -v "amdgpu.ids" > $out/host_sections_single_compact.txt
timeout 400 python tools/host_sections.py --sharded-one-rank --full-layout 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|ProcessGroupNCCL" > $out/host_sections_dp_full.txt
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --compact-first-grid --no-cpu-baseline --no-kernels-alone --no-first-grid-phase --no-open-window --no-kernel-events --preroll 600 --steps 20 --warmup 5 --steady-after 0"
timeout 500 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/trace_dp -o dp -- $B --sharded-one-rank > $GRAFT_REPO_ROOT/$out/trace_dp.json 2> $GRAFT_REPO_ROOT/$out/trace_dp.err
timeout 500 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/trace_single -o single -- $B > $GRAFT_REPO_ROOT/$out/trace_single.json 2> $GRAFT_REPO_ROOT/$out/trace_single.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $out/trace_dp 16 > $out/timeline_dp_compact.txt 2>&1
python tools/timeline.py $out/trace_single 16 > $out/timeline_single_compact.txt 2>&1
rm -rf $out/trace_dp $out/trace_single
head -30 $out/host_sections_dp_compact.txt; head -12 $out/host_sections_single_compact.txt; tail -6 $out/timeline_dp_compact.txt; tail -6 $out/timeline_single_compact.txt

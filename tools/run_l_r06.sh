#!/bin/bash
# Round 6, run L: the host side of a data-parallel rank's step (compact phase) and of an emulated level-parallel rank's, finely
set -u
out=gpurun_out/r06_l; mkdir -p $out
export TMPDIR=/tmp
F='amdgpu.ids\|socket.cpp\|Gloo\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|ProcessGroupNCCL'
timeout 400 python tools/host_sections.py --sharded-one-rank --fine 2>&1 | grep -v "$F" > $out/fine_dp_compact.txt
timeout 400 python tools/host_sections.py --level-parallel-one-rank 8 --fine 2>&1 | grep -v "$F" > $out/fine_lp8.txt
timeout 400 python tools/host_sections.py --fine 2>&1 | grep -v "$F" > $out/fine_single_compact.txt
cat $out/fine_dp_compact.txt

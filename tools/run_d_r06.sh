#!/bin/bash
# Round 6, run D: the whole -m gpu suite + smoke
set -u
out=gpurun_out/r06_d; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $out/suite.txt
grep -n "^E  \|FAILED\|passed\|failed" $out/suite.txt | head -40

#!/bin/bash
# Round 6, run E: owner-computes scatter experiment (tools/micro/owner_scatter.hip), mlp_bwd with the next tile's input prefetched
set -u
out=gpurun_out/r06_e; mkdir -p $out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$PWD/nersemble_amd/csrc:${LD_LIBRARY_PATH:-}
for args in "20 24 0" "20 24 1" "17 24 1"; do echo "== $args"; timeout 300 tools/micro/owner_scatter $args; done > $out/owner_scatter.txt 2>&1
cat $out/owner_scatter.txt
timeout 300 python tools/mlp_bench.py > $out/mlp_bench.txt 2>&1; tail -12 $out/mlp_bench.txt

#!/bin/bash
# Round 6, run E: owner-computes scatter experiment (tools/micro/owner_scatter.hip: 128 KB and 64 KB tiles)
set -u
out=gpurun_out/r06_e; mkdir -p $out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$PWD/nersemble_amd/csrc:${LD_LIBRARY_PATH:-}
for t in 14 13; do for args in "20 24 0" "20 24 1"; do echo "== tile 2^$t: $args"; timeout 300 tools/micro/owner_scatter_$t $args | tail -5; done; done > $out/owner_scatter_v3.txt 2>&1
cat $out/owner_scatter_v3.txt

#!/bin/bash
# Round 6, run E: scatter experiments without memory-side atomics (tools/micro/owner_scatter.hip, binned_scatter.hip)
set -u
out=gpurun_out/r06_e; mkdir -p $out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$PWD/nersemble_amd/csrc:${LD_LIBRARY_PATH:-}
for args in "20 24 0" "20 24 1" "17 24 1"; do echo "== binned: $args"; timeout 300 tools/micro/binned_scatter $args | tail -8; done > $out/binned_scatter.txt 2>&1
cat $out/binned_scatter.txt

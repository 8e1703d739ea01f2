#!/bin/bash
# round 5, run A: the whole -m gpu suite on the state after the environment knobs / probe kernels left libnsx.so and the
# slot-terms deformation forward became the table route; + the two deformation forwards timed against each other
set -u
out=gpurun_out/r05_a; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
tail -1 $out/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu -x -q 2>&1 | tail -40 > $out/suite.txt
tail -5 $out/suite.txt
timeout 300 python tools/deform_fwd_ab.py > $out/deform_fwd_ab.json 2> $out/deform_fwd_ab.err
cat $out/deform_fwd_ab.json

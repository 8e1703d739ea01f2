#!/bin/bash
# round 4: the scale update as one native launch -- its test, the training tests that go through GradScaler semantics, host
# issue time by section, the compact-phase steady state un-traced
set -u
export TMPDIR=/tmp
out=gpurun_out/i_r04; mkdir -p $out
python -m pytest tests/test_adam_gpu.py tests/test_training_gpu.py tests/test_native_step_gpu.py -q -m gpu -x 2>&1 | tail -6 > $out/tests.txt
cat $out/tests.txt
python tools/host_sections.py > $out/host_sections_compact.txt 2>&1
python tools/host_sections.py --full-layout > $out/host_sections_full.txt 2>&1
python tools/host_profile.py --plain --steps 200 --compact 2>/dev/null | tail -1 > $out/untraced_compact.txt
head -9 $out/host_sections_compact.txt; head -5 $out/host_sections_full.txt; cat $out/untraced_compact.txt

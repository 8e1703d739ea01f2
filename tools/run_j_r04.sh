#!/bin/bash
# round 4, A/B on one box: GradScaler.update as one native launch (NSX_NATIVE_SCALER=1, default) vs torch's route (=0)
set -u
export TMPDIR=/tmp
out=gpurun_out/j_r04; mkdir -p $out
for v in 1 0 1 0; do
  NSX_NATIVE_SCALER=$v python tools/host_sections.py 2>/dev/null | sed -n 2,5p | tr '\n' ' ' | sed "s/^/native_scaler=$v: /" >> $out/ab_scaler.txt; echo >> $out/ab_scaler.txt
done
for v in 1 0; do
  NSX_NATIVE_SCALER=$v python tools/host_profile.py --plain --steps 200 --compact 2>/dev/null | tail -1 | sed "s/^/native_scaler=$v compact: /" >> $out/ab_scaler.txt
done
cat $out/ab_scaler.txt

#!/bin/bash
# round 4, A/B on ONE box (alternating): leaf gradients deposited in place vs cloned by autograd (steady state, both
# layouts); the pre-blended lookup with one lane vs a lane pair per (sample, level) (one evaluation image)
set -u
export TMPDIR=/tmp
out=gpurun_out/f_r04; mkdir -p $out
python -m pytest tests/test_hash_ensemble_gpu.py tests/test_image_parity_gpu.py tests/test_field_gpu.py -q -m gpu -x 2>&1 | tail -3 > $out/tests.txt
for rep in 1; do
  for dep in 1 0; do
    NSX_GRAD_DEPOSIT=$dep python tools/host_profile.py --plain --steps 200 2>/dev/null | tail -1 | sed "s/^/full deposit=$dep: /" >> $out/ab_deposit.txt
    NSX_GRAD_DEPOSIT=$dep python tools/host_profile.py --plain --steps 200 --compact 2>/dev/null | tail -1 | sed "s/^/compact deposit=$dep: /" >> $out/ab_deposit.txt
  done
done
for v in 1 2 1; do
  NSX_HASHGRID_FWD=$v python tools/eval_bench.py 2>/dev/null | grep "preblend=True" | tail -1 | sed "s/^/lookup variant $v: /" >> $out/ab_lookup.txt
done
NSX_HASHGRID_FWD=2 python tools/eval_bench.py --price 2>/dev/null | tail -1 > $out/eval_priced.json
cat $out/tests.txt $out/ab_deposit.txt $out/ab_lookup.txt; cut -c1-1500 $out/eval_priced.json

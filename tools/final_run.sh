# End-of-round evidence on one box: smoke, the whole -m gpu suite, the default bench line (with its first_grid_phase block),
# optionally (PROFILES=1) rocprofv3 kernel statistics + the two HBM counter passes and (DP=1) the 2-rank control flow
# (gloo, both ranks on cuda:0).  Results: gpurun_out/final/ and gpurun_out/prof_<tag>/
set -u
tag=${1:-r03}
out=gpurun_out/final; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
if [ "${DP:-0}" = 1 ]; then
DPC="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --steady-after 0 --reserve-gb 2"
timeout 600 $DPC > $out/dp2_weak.json 2> $out/dp2_weak.err
timeout 600 $DPC --scaling strong > $out/dp2_strong.json 2> $out/dp2_strong.err
fi
python -m pytest tests -q -m gpu -x 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8 > $out/full_suite.txt
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
[ "${PROFILES:-0}" = 1 ] && bash tools/collect_profiles.sh $tag > $out/collect.log 2>&1
tail -2 $out/smoke.txt; tail -3 $out/full_suite.txt; python - <<'P'
import json
d=json.load(open("gpurun_out/final/bench.json"))
print(round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3))
print("first_grid_phase:", {k:(v if not isinstance(v,dict) else {kk:v[kk] for kk in list(v)[:4]}) for k,v in d.get("first_grid_phase",{}).items() if k!="native_kernel_avg_ms"})
P

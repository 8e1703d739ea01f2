# tests of the kernels touched last + one bench line (no CPU baseline / stand-alone kernels): results in gpurun_out/chk/
set -u
out=gpurun_out/chk; mkdir -p $out
python -m pytest tests/test_hash_ensemble_gpu.py tests/test_adam_gpu.py tests/test_training_gpu.py -x -q 2>&1 | tail -25 > $out/tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone > $out/bench.json 2> $out/bench.err
tail -5 $out/tests.txt
python - <<'P'
import json
d=json.load(open("gpurun_out/chk/bench.json"))
print("open:", round(d["ms_per_step"],3), round(d["value"]/1e6,2), "steady", round(d["steady_state"]["ms_per_step"],3), d["config"]["hash_grids_switched_on"])
print({k:v["avg_ms"] for k,v in d["native_kernel_ms"].items() if v["avg_ms"]>0.2})
s=d.get("schedule_start")
if s: print("schedule:", s["grids_switched_on"], round(s["ms_per_step"],3), round(s["value"]/1e6,2), "steady", round(s["steady_state"]["ms_per_step"],3), s["steady_state"]["psnr"])
print(d["roofline"])
P

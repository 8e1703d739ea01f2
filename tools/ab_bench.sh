# A/B of the step-level options on one box: tools/ab_bench.sh  (results in gpurun_out/ab/)
set -u
out=gpurun_out/ab; mkdir -p $out
[ "${AB_TESTS:-0}" = 1 ] && python -m pytest tests/test_adam_gpu.py tests/test_training_gpu.py -x -q 2>&1 | tail -5 > $out/tests.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
$B > $out/both.json 2> $out/both.err
NSX_PREFETCH_MARCH=0 $B > $out/no_prefetch.json 2> $out/no_prefetch.err
NSX_ADAM_CONSUMES_G=0 $B > $out/no_gclear.json 2> $out/no_gclear.err
NSX_PREFETCH_MARCH=0 NSX_ADAM_CONSUMES_G=0 $B > $out/neither.json 2> $out/neither.err
[ -f $out/tests.txt ] && cat $out/tests.txt
python - <<'P'
import json
for n in ("both","no_prefetch","no_gclear","neither"):
    try:
        d=json.load(open(f"gpurun_out/ab/{n}.json"))
        print(n, round(d["ms_per_step"],3), round(d["value"]/1e6,2), round(d["steady_state"]["ms_per_step"],3), d["steady_state"]["psnr"])
    except Exception as e:
        print(n, "failed", e)
P

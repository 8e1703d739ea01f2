#!/bin/bash
# Round 6, run W: the level-parallel exchange's collectives issued by the library (csrc/comm.hip) against torch.distributed: the
# parity test, the level-parallel tests, emulated rank 7 of 8 with both settings on one box, host issue time by section.
set -u
out=gpurun_out/r06_w; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_boundary.py -q -m gpu -x -k "collectives_issued or one_launch or emulated or level_parallel or marches_nothing" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -15 | tee $out/tests.txt
LP="python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
for r in 7 0; do
  timeout 400 $LP --rank $r > $out/lp8_rank${r}_library.json 2> $out/lp8_rank${r}_library.err
  timeout 400 $LP --rank $r --lp-torch-collectives > $out/lp8_rank${r}_torch.json 2> $out/lp8_rank${r}_torch.err
done
timeout 400 python tools/host_sections.py --level-parallel-one-rank 8 --fine > $out/host_sections_fine_lp8.txt 2>&1
python - <<'P'
import json
for r in (7, 0):
    for tag in ("library", "torch"):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r06_w/lp8_rank{r}_{tag}.json") if l.startswith("{")][-1])
            ss = d.get("steady_state") or {}; k = d["native_kernel_ms"]; c = ss.get("comm") or {}
            print(r, tag, "window", round(d["ms_per_step"], 3), "| steady", round(ss.get("ms_per_step", 0), 3), "host", round(ss.get("host_issue_ms_per_step", 0), 3),
                  "min", ss.get("host_issue_ms_per_step_min"), "shadow", round(c.get("shadow_fwd_ms", 0), 3), "coll", c.get("collectives_per_step"), "psnr", ss.get("psnr"))
        except Exception as e:
            print(r, tag, "ERR", repr(e))
P
head -30 $out/host_sections_fine_lp8.txt
tail -3 $out/lp8_rank7_library.err

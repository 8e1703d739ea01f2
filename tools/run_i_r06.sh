#!/bin/bash
# Round 6, run I: every rank of an 8-rank level-parallel job, one after the other, emulated on ONE GPU (same box) + the new test
set -u
out=gpurun_out/${RUN_I_OUT:-r06_i}; mkdir -p $out
export TMPDIR=/tmp
[ -n "${RUN_I_SKIP_TESTS:-}" ] || timeout 600 python -m pytest tests/test_sharded_gpu.py -q -m gpu -k "frozen_emulated or emulated" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -5
LP="python bench.py --level-parallel-one-rank 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels-alone"
for r in 0 1 2 3 4 5 6 7; do timeout 400 $LP --rank $r > $out/lp8_rank$r.json 2> $out/lp8_rank$r.err; done
python - <<'P'
import json, os
OUT = os.environ.get("RUN_I_OUT", "r06_i")
rows = []
for r in range(8):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{OUT}/lp8_rank{r}.json") if l.startswith("{")][-1])
        ss = d.get("steady_state") or {}
        c = ss.get("comm") or {}
        k = d["native_kernel_ms"]
        rows.append({"rank": r, "levels": c.get("levels"), "window_ms_per_step": round(d["ms_per_step"], 3),
                     "window_fwd_run_ms": (k.get("nsx_lp_fwd_run") or k.get("nsx_lp_forward") or {}).get("avg_ms"), "window_bwd_run_ms": (k.get("nsx_lp_bwd_run") or k.get("nsx_lp_backward") or {}).get("avg_ms"),
                     "window_adam_ms": k.get("nsx_adam_hash_factored", {}).get("avg_ms"),
                     "steady_ms_per_step": round(ss.get("ms_per_step", 0), 3), "steady_shadow_fwd_ms": round(c.get("shadow_fwd_ms", 0), 3),
                     "steady_adam_ms": round(c.get("shard_adam_ms", 0), 3), "host_issue_ms": round(ss.get("host_issue_ms_per_step", 0), 3),
                     "steady_samples_min_max": ss.get("samples_per_step_min_max"), "bytes_arriving_per_step": c.get("bytes_per_rank"),
                     "collectives_per_step": c.get("collectives_per_step")})
    except Exception as e:
        rows.append({"rank": r, "error": repr(e)})
ok = [x for x in rows if "error" not in x]
doc = {"what": "bench.py --level-parallel-one-rank 8 --rank r for r = 0 .. 7 on ONE GPU, one process after the other (same box): every rank of an "
               "8-rank level-parallel job, the other ranks emulated as replicas; steady state = a trained model, frozen, the replicas' "
               "feature columns replaced by one full-geometry forward per pass (steady_shadow_fwd_ms is inside steady_ms_per_step)",
       "ranks": rows}
if ok:
    w = [x["window_ms_per_step"] for x in ok]; s = [x["steady_ms_per_step"] for x in ok]
    doc["window_max_over_min"] = round(max(w) / min(w), 3); doc["steady_max_over_min"] = round(max(s) / min(s), 3)
    doc["window_max_ms"], doc["steady_max_ms"] = max(w), max(s)
json.dump(doc, open(f"gpurun_out/{OUT}/r06_level_parallel_all_ranks.json", "w"), indent=1)
for x in rows: print(x)
print({k: v for k, v in doc.items() if k not in ("ranks", "what")})
P

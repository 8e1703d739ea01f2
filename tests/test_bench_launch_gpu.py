"""bench.py's launch contract: `python bench.py --gpus N` starts its own ranks when no launcher did (the driver's
multi-GPU line goes through torch.distributed.run and sets WORLD_SIZE itself)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_launches_its_own_ranks(cuda):
    """Two ranks on the one GPU of the test box (gloo carries the collectives; RCCL needs a GPU per rank): the whole
    multi-rank control flow of bench.py -- rendezvous on 127.0.0.1, sharded table optimizer, barrier + MAX-over-ranks
    timing, rank 0 prints ONE JSON line with the aggregate over both ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--ranks-share-gpu0",
           "--workload", "p030_h16", "--steps", "3", "--warmup", "2", "--steady-after", "0", "--no-cpu-baseline",
           "--reserve-gb", "2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    assert d["config"]["rays_per_gpu"] == 4096 and d["scaling"] == "weak"
    # the line diagnoses its own exchange: durations of the phases of the sharded table step on the streams they run on
    comm = d["comm"]
    assert comm["world_size"] == 2 and comm["buckets"] >= 2 and comm["steps"] == 3 and comm["backend"] == "gloo"
    for key in ("expand_f16_ms", "reduce_scatter_ms", "shard_adam_ms", "all_gather_ms"):
        assert comm[key] > 0, (key, comm)
    assert comm["reduce_scatter_exposed_ms"] >= 0 and comm["reduce_scatter_bytes_per_rank"] > 0
    assert comm["reduce_scatter_bus_GBps"] > 0 and len(comm["ms_per_step_per_rank_min_max"]) == 2
    # the window keeps one of the 16 grids on in this configuration: the exchange carries that grid only
    assert comm["exchange_width"] == 1 and comm["grids"] == 16
    assert comm["reduce_scatter_bytes_per_rank"] < 0.07 * 2 * 16 * 2 * 6.3e6

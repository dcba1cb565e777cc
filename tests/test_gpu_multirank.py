"""The N > 1 path of bench.py with real kernels on a one-GPU box: `LQER_BENCH_REHEARSE=1 python bench.py --gpus 2 ...` starts its
own two ranks (child processes; this process only waits for them), both on cuda:0, gloo carrying the collectives RCCL would carry
on a node - layer partition, broadcast of x, barrier-bracketed timed regions, gathers, rank 0's JSON line and its oracle check.
The figures of such a run mean nothing (the ranks share the device); the plumbing is what is tested.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,extra,scaling,layers", [("c3", ["--layers", "3"], "strong", [2, 1]), ("c2", [], "weak", [1, 1])])
def test_two_ranks_on_one_gpu(workload, extra, scaling, layers):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, LQER_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-module"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["config"]["layers_per_rank"] == layers
    assert len(d["rank_ms_per_step"]) == 2 and all(t > 0 for t in d["rank_ms_per_step"])
    assert d["parity_rel_l2"] is not None and d["parity_rel_l2"] <= 1e-3
    assert d["value"] > 0 and d["two_streams"] is None and d["model_shared_inputs"] is None  # (secondary regions: single rank only)

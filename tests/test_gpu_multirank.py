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


@pytest.mark.parametrize("workload,extra,scaling,layers", [("c3", ["--layers", "3"], "strong", [2, 1]), ("c2", [], "weak", [1, 1]),
                                                           ("c2", ["--shard", "n"], "strong", [1, 1])])
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
    if "--shard" in extra:  # column-parallel single Linear: rank g owns the columns [n0, n1); the all-gathered y meets the oracle
        assert d["config"]["column_shard"] == {"4096": [[0, 2048], [2048, 4096]]}


def test_column_parallel_linear_is_bit_identical_to_the_unsharded_forward():
    """SURVEY.md §8e, optional row: a Linear split column-parallel at multiples of 16 needs no reduction - W's blocks run along
    K, B_out blocks are 16 columns - so every shard's y[:, n0:n1] carries the BITS of the unsharded forward.  Three uneven
    shards (lqer_amd.sweep.column_partition, 16-column granules) on one GPU, bias included."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import lqer_amd
    from bench import OPT_Q, make_case
    from lqer_amd import sweep

    dev = "cuda:0"
    M, K, N, r = 300, 512, 1000, 32
    x, W, A, B, b = make_case(M, K, N, r, seed=21, bias=True)

    def build(n0, n1):
        mod = lqer_amd.LinearFlexibleLqer(K, n1 - n0, bias=True, q_config=OPT_Q, l_config={"rank": r})
        mod.load_state_dict({"weight": W[n0:n1], "A": A, "B": B[:, n0:n1].contiguous(), "bias": b[n0:n1]})
        return mod.to(dev).half()

    xd = x.half().to(dev)
    full = build(0, N)(xd)
    ranges = sweep.column_partition(N, 3)
    assert ranges == [(0, 336), (336, 672), (672, 1000)] and all(n0 % 16 == 0 for n0, _ in ranges)
    for n0, n1 in ranges:
        assert torch.equal(build(n0, n1)(xd), full[:, n0:n1]), (n0, n1)
    # (OPT_Q's bias blocks are 16 wide: a shard boundary at a multiple of 16 keeps them whole as well)


def _bench(cmd_extra, env_extra, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + cmd_extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_default_multi_gpu_line_is_c4_with_c2_replicas_rehearsed():
    """`bench.py --gpus 2` with NO --workload (the driver's multi-GPU command), rehearsed on one GPU over gloo: the headline is
    BASELINE's multi-GPU configuration (c4: layers split over the ranks, strong scaling; --layers 2 keeps the rehearsal short)
    and `replicas_c2` carries the N independent C2 Linears (weak scaling), both oracle-checked."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(["--gpus", "2", "--layers", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-module"], {"LQER_BENCH_REHEARSE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["layers_per_rank"] == [1, 1] and "Llama-13B" in d["config"]["workload"]
    assert d["dtype"] == "int8" and d["parity_rel_l2"] <= 1e-3 and d["value"] > 0
    r = d["replicas_c2"]
    assert r["scaling"] == "weak" and r["value"] > 0 and r["parity_rel_l2"] <= 1e-3 and len(r["rank_ms_per_step"]) == 2


def test_two_gpus_over_rccl():
    """The first box with two GPUs that runs this suite exercises RCCL: `bench.py --gpus 2 --workload c4 --layers 4` over the nccl
    backend - broadcast of the token batch over xGMI, barrier-bracketed timed region, gathers.  Skipped on the
    one-GPU boxes of this pool."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL)")
    d = _bench(["--gpus", "2", "--workload", "c4", "--layers", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-module"],
               {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["layers_per_rank"] == [2, 2]
    assert d["parity_rel_l2"] <= 1e-3 and d["value"] > 0 and d["broadcast_ms"] > 0
    assert len(d["rank_ms_per_step"]) == 2 and all(t > 0 for t in d["rank_ms_per_step"])

"""The N > 1 host logic on CPU: two gloo ranks exercise the timing reduction, the checksum gather and
the aggregate computed from them; plus the reference's layer -> device partition rule."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lqer_amd import sweep


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        elapsed = sweep.max_over_ranks(1.0 + rank, dev)  # rank 1 is the slow one
        sums = sweep.gather_checksums(10.0 * (rank + 1), dev)
        dist.barrier()
        q.put((rank, elapsed, sums, sweep.aggregate_throughput(2e12, 5, world, elapsed)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, elapsed, sums, agg in res:
        assert elapsed == 2.0  # max over ranks
        assert sums == [10.0, 20.0]
        assert agg == pytest.approx(2e12 * 2 * 5 / 2.0 / 1e12)


def _sweep_worker(rank, world, port, q, n_layers):
    """The strong-scaling sweep's host logic as bench.py runs it: rank 0 makes the token batch, it is broadcast once
    per distinct K, every rank owns ceil(L / G) consecutive layers (uneven when G does not divide L), per-rank rows are
    gathered and the whole-job FLOPs are summed."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bench import flops, make_x

        dev = torch.device("cpu")
        shapes = [(64, 48, 2), (48, 64, 1)]  # (K, N, count per layer)
        M, r = 8, 16
        xs = {}
        for K in sorted({K for K, _, _ in shapes}):
            x = make_x(M, K, seed=0)[0].half() if rank == 0 else torch.full((M, K), float("nan"), dtype=torch.float16)
            xs[K] = sweep.broadcast_activation(x, src=0)
        mine = sweep.layer_partition(n_layers, world)[rank]
        units = sweep.projection_units(shapes, mine)
        fl = float(sum(flops(M, K, N, r) for _, K, N, _ in units))
        total = sweep.sum_over_ranks(fl, dev)
        rows = sweep.gather_rows([0.5 * (rank + 1), float(xs[64].float().sum()), float(len(mine))], dev)
        q.put((rank, list(mine), len(units), total, rows, {K: float(v.float().abs().sum()) for K, v in xs.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_layers", [5, 4, 1])
def test_layer_partitioned_sweep_two_ranks_gloo(n_layers):
    from bench import flops, make_x

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() + 7 * n_layers) % 2000
    procs = [ctx.Process(target=_sweep_worker, args=(r, 2, port, q, n_layers)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per = -(-n_layers // 2)
    assert res[0][1] == list(range(0, per)) and res[1][1] == list(range(per, n_layers))  # 5 -> 3 + 2, 4 -> 2 + 2, 1 -> 1 + 0
    assert res[0][2] == 3 * per and res[1][2] == 3 * (n_layers - per)
    want = n_layers * (2 * flops(8, 64, 48, 16) + flops(8, 48, 64, 16))
    assert res[0][3] == res[1][3] == want  # a fixed model: total work does not depend on the split
    assert res[0][4] == res[1][4] and [row[2] for row in res[0][4]] == [float(per), float(n_layers - per)]
    assert [row[0] for row in res[0][4]] == [0.5, 1.0]
    ref = {K: float(make_x(8, K, seed=0)[0].half().float().abs().sum()) for K in (48, 64)}
    assert res[0][5] == res[1][5] == ref  # rank 1 received rank 0's batch (its own buffer was NaN)


def test_single_process_is_identity():
    x = torch.ones(2, 3)
    assert sweep.broadcast_activation(x) is x
    assert sweep.gather_rows([1.0, 2.0], torch.device("cpu")) == [[1.0, 2.0]] and sweep.sum_over_ranks(3.0, torch.device("cpu")) == 3.0
    assert sweep.max_over_ranks(0.25, torch.device("cpu")) == 0.25
    assert sweep.gather_checksums(3.0, torch.device("cpu")) == [3.0]


def test_layer_partition_matches_reference_rule():
    # reference experiments/infer_device_map.py:29-37: ceil(L / G) consecutive layers per device
    assert [list(r) for r in sweep.layer_partition(32, 8)] == [list(range(4 * g, 4 * g + 4)) for g in range(8)]
    parts = sweep.layer_partition(40, 8)  # Llama-13B on 8 GPUs: 5 layers per rank
    assert [len(r) for r in parts] == [5] * 8
    parts = sweep.layer_partition(10, 4)  # ceil -> 3,3,3,1
    assert [len(r) for r in parts] == [3, 3, 3, 1] and sum(len(r) for r in parts) == 10
    assert [len(r) for r in sweep.layer_partition(2, 4)] == [1, 1, 0, 0]
    with pytest.raises(ValueError):
        sweep.layer_partition(4, 0)
    units = sweep.projection_units([(4096, 4096, 4), (4096, 11008, 2), (11008, 4096, 1)], parts[0] if False else range(2))
    assert len(units) == 14 and units[0] == (0, 4096, 4096, 0)
    assert sweep.unit_seed(0, 1) != sweep.unit_seed(1, 1) != sweep.unit_seed(1, 0)


def _approx_worker(rank, world, port, q):
    """The sharded approximator's ownership + exchange on CPU: a stand-in for the (GPU-only) factorization writes
    rank-tagged values, so the result shows who computed what and that everybody received everything."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn as nn

        from bench import MXINT_Q
        from lqer_amd import LinearFlexibleLqer
        from lqer_amd.approximate import approximate_model

        class Layer(nn.Module):
            def __init__(self):
                super().__init__()
                self.q_proj = LinearFlexibleLqer(32, 48, bias=False, q_config=MXINT_Q, l_config={"rank": 16})
                self.o_proj = LinearFlexibleLqer(48, 32, bias=False, q_config=MXINT_Q, l_config={"rank": 16})

        class Model(nn.Module):
            def __init__(self):
                super().__init__()
                self.layers = nn.ModuleList([Layer() for _ in range(3)])

        calls = []

        def fake_factors(W, w_cfg, r, a_cfg, b_cfg, scale):
            calls.append(tuple(W.shape))
            return torch.full((W.shape[1], r), float(rank + 1)), torch.full((r, W.shape[0]), float(-(rank + 1)))

        model = Model()
        out = approximate_model(model, factors_fn=fake_factors)
        q.put((rank, len(calls), {k: float(v.flatten()[0]) for k, v in out.items()}))
    finally:
        dist.destroy_process_group()


def test_sharded_approximator_two_ranks_gloo():
    from lqer_amd.approximate import module_owners

    names = [f"model.layers.{l}.self_attn.{p}" for l in range(5) for p in ("q_proj", "k_proj")] + ["lm_head"]
    own = module_owners(names, 2)  # ceil(5 / 2) = 3 consecutive layers on rank 0, the rest on rank 1
    assert [own[f"model.layers.{l}.self_attn.q_proj"] for l in range(5)] == [0, 0, 0, 1, 1] and own["lm_head"] == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_approx_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [n for _, n, _ in res] == [4, 2]  # 3 layers: two on rank 0 (4 modules), one on rank 1 (2 modules)
    assert res[0][2] == res[1][2]  # every rank ends with the same full dictionary
    d = res[0][2]
    assert d["layers.0.q_proj.A"] == 1.0 and d["layers.1.o_proj.B"] == -1.0 and d["layers.2.q_proj.A"] == 2.0 and d["layers.2.o_proj.B"] == -2.0


@pytest.mark.parametrize("workload,layers_per_rank,scaling", [("c3", [16, 16], "strong"), ("c2", [1, 1], "weak")])
def test_bench_starts_its_own_ranks_dry_run(workload, layers_per_rank, scaling):
    """`python bench.py --gpus 2` with no launcher around it (the form of the driver's 1-GPU command): the process
    starts the two ranks itself as a child (torch.distributed.run), which partition the layers by the reference's rule
    (experiments/infer_device_map.py:29-37), broadcast, time, gather - and stdout carries exactly ONE JSON line.
    --dry-run-cpu keeps it on gloo with no kernel, so it runs in a container without a GPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--workload", workload],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["value"] is None and "dry_run" in out
    assert out["config"]["layers_per_rank"] == layers_per_rank
    assert len(out["rank_ms_per_step"]) == 2
    assert out["rank_checksums"][0] == out["rank_checksums"][1]  # rank 1 holds rank 0's broadcast batch


def test_bench_launcher_relays_the_childs_exit_code():
    """A failing rank must fail the bench: --sweep strong on a single-Linear workload is refused by every rank."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--workload", "c2",
                          "--sweep", "strong"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_column_partition_rule():
    """Column-parallel split of one Linear: cuts at multiples of 16, ceil(blocks / G) blocks per rank, trailing
    ranks may own fewer or no columns, the ranges tile [0, N) exactly."""
    assert sweep.column_partition(4096, 2) == [(0, 2048), (2048, 4096)]
    assert sweep.column_partition(4096, 8) == [(512 * g, 512 * (g + 1)) for g in range(8)]
    assert sweep.column_partition(11008, 8) == [(1376 * g, 1376 * (g + 1)) for g in range(8)]
    assert sweep.column_partition(1000, 3) == [(0, 336), (336, 672), (672, 1000)]  # 63 blocks -> 21 per rank, last one ragged
    assert sweep.column_partition(40, 4) == [(0, 16), (16, 32), (32, 40), (40, 40)]  # the last rank owns nothing
    for N, G in ((4096, 3), (11008, 5), (50, 8), (16, 2)):
        rg = sweep.column_partition(N, G)
        assert rg[0][0] == 0 and rg[-1][1] == N and all(a[1] == b[0] for a, b in zip(rg, rg[1:]))
        assert all(n0 % 16 == 0 for n0, n1 in rg if n1 > n0)  # (every non-empty shard starts on a 16-column granule)
    with pytest.raises(ValueError):
        sweep.column_partition(16, 0)


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        M, N = 5, 40
        ranges = sweep.column_partition(N, world)
        full = torch.arange(M * N, dtype=torch.float32).reshape(M, N)
        n0, n1 = ranges[rank]
        got = sweep.all_gather_columns(full[:, n0:n1].contiguous(), ranges, N)
        q.put((rank, bool(torch.equal(got, full))))
    finally:
        dist.destroy_process_group()


def test_all_gather_columns_two_ranks_uneven():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]  # 40 columns over 2 ranks: 32 + 8, padded to the widest slice for the collective


def test_dry_run_column_shard_line():
    """`bench.py --gpus 2 --shard n --dry-run-cpu`: the launcher, the column split and the JSON line of the column-parallel run."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--shard", "n", "--dry-run-cpu"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["column_shard"] == {"4096": [0, 2048]}


def test_bench_default_workload_on_several_gpus_is_the_baseline_multi_gpu_config():
    """`python bench.py --gpus 2` with no --workload (the driver's multi-GPU command): the headline is BASELINE's multi-GPU
    configuration - c4, Llama-13B's 40 layers split 20 / 20 by the reference's rule, strong scaling; with
    `--gpus 1` the default stays c2 (BASELINE configs[1])."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-cpu"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["layers_per_rank"] == [20, 20]
    assert "Llama-13B" in out["config"]["workload"]
    res1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run-cpu"], env=env, capture_output=True,
                          text=True, timeout=300)
    assert res1.returncode == 0, res1.stderr[-2000:]
    out1 = json.loads([ln for ln in res1.stdout.splitlines() if ln.startswith("{")][0])
    assert out1["n_gpus"] == 1 and "4096x4096" in out1["config"]["workload"]

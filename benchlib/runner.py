"""One workload on this rank's GPU: setup (token batch, packed operands, launch plans), the timed regions, and the record
bench.py prints.  Everything that is timed goes through the C ABI (lqer_quantize_act_xa + lqer_linear_gemm, or
lqer_linear_forward for the one-launch decode route); the CPU oracle is touched only after the timed regions (parity of the
buffers just written) and in the cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import sys
import time
from dataclasses import dataclass
from typing import Optional

import torch

from . import roofline as RL
from .cpu_baseline import cpu_baseline
from .hipevents import HipEvent
from .workloads import MXINT_Q, UNQUANTIZED_AB, WORKLOADS, check_rows, flops, make_weights, make_x

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@dataclass
class Ctx:
    """Process-wide facts of a bench run: this rank, the job, its device, the process group (or None)."""
    rank: int
    world: int
    dev: torch.device
    dist: object  # torch.distributed or None


@dataclass
class Opts:
    workload: str
    steps: Optional[int] = None
    warmup: Optional[int] = None
    layers: int = 0              # override of the model's decoder-layer count (0 = the model's)
    sweep: str = "auto"          # auto | weak | strong
    shard: str = "none"          # none | n (column-parallel single Linear)
    check: bool = True
    module: bool = True          # secondary figures through the nn.Module (and shared inputs for model workloads)
    two_streams: bool = True
    cpu_base: bool = True
    prewarm_ms: float = 300.0
    graph: int = 0
    rotate: Optional[int] = None
    shared_weights: bool = False
    uninstrumented: bool = True  # a second timed region without event pairs
    tuning: int = 0              # lqer_linear_desc_t.tuning of every module (LQER_TUNE_*: kernel-variant A/B on one box, same bits)


def timed_region(ctx: Ctx, fn, steps):
    """EXACTLY `steps` steps between barrier + synchronize on both sides (driver contract)."""
    if ctx.dist is not None:
        ctx.dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    if ctx.dist is not None:
        ctx.dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def run_workload(ctx: Ctx, o: Opts) -> Optional[dict]:
    """Runs the workload; returns the record on rank 0 (None elsewhere)."""
    import lqer_amd
    from lqer_amd import _lib, ops, sweep

    rank, world, dev, dist = ctx.rank, ctx.world, ctx.dev, ctx.dist
    if o.workload == "d1layer":
        return _decode_layer(ctx, o)
    desc_txt, M, r, has_bias, qc, shapes, layers = WORKLOADS[o.workload]
    if o.layers > 0:
        layers = o.layers
        desc_txt += f" [--layers {layers}]"
    big = M >= 8192
    steps = o.steps if o.steps is not None else (4 if big else (400 if M <= 64 else 50))
    # (decode sizes: a 50-step region is 0.4 ms, of which the first launch's latency and the closing synchronize are ~10 %)
    warmup = o.warmup if o.warmup is not None else (1 if big else 10)
    shard_n = o.shard == "n"
    strong = shard_n or (o.sweep == "strong") or (o.sweep == "auto" and layers > 1)
    if shard_n and (layers > 1 or len(shapes) > 1):
        sys.exit("--shard n is the column-parallel split of ONE Linear (c2 / d1 / d16): model sweeps split by layer")
    if strong and not shard_n and layers == 1 and world > 1:
        sys.exit("--sweep strong needs a model workload (c3/c4/c5): a single Linear has no layers to split")
    my_layers = range(layers) if shard_n else (sweep.layer_partition(layers, world)[rank] if strong else range(layers))
    layers_here = len(my_layers)
    quantize_ab = not any(qc is c for c in UNQUANTIZED_AB)

    # ---- the token batch: generated on rank 0, broadcast once per distinct K (RCCL over xGMI), outside the timed region
    xs, broadcast_ms = {}, 0.0
    for K in sorted({K for K, _, _ in shapes}):
        if rank == 0:
            xd = make_x(M, K, seed=0)[0].half().to(dev)
        else:
            xd = torch.empty(M, K, dtype=torch.float16, device=dev)
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            sweep.broadcast_activation(xd, src=0)
            torch.cuda.synchronize()
            broadcast_ms += (time.perf_counter() - t0) * 1e3
        xs[K] = xd

    # ---- this rank's units: one module per distinct projection shape, weights from a per-rank seed; a model sweep re-runs
    # each shape `count x layers owned` times per step (layers differ in values, not in cost).  --shard n: every rank
    # builds the SAME Linear from one seed and keeps its columns [n0, n1) of W, B and the bias.
    mods = []
    col_ranges = {}
    for i, (K, N, cnt) in enumerate(shapes):
        g = torch.Generator().manual_seed(sweep.unit_seed(0 if shard_n else rank, i))
        wts = make_weights(g, K, N, r, bias=has_bias, quantize_ab=quantize_ab)
        n0, n1 = 0, N
        if shard_n:
            bo = qc.get("B_out_quantizer", qc["x_quantizer"])
            if bo.get("name") == "block_fp" and bo["block_size"][-1] != 16:
                sys.exit("--shard n: B_out blocks must be 16 columns (a block that spans several ranks' columns would need a reduction)")
            col_ranges[N] = sweep.column_partition(N, world)
            n0, n1 = col_ranges[N][rank]
        Nl = n1 - n0
        if Nl == 0:
            mods.append((None, xs[K], K, 0, 0, torch.empty(M, 0, dtype=torch.float16, device=dev), wts, N))
            continue
        mod = lqer_amd.LinearFlexibleLqer(K, Nl, bias=has_bias, q_config=qc, l_config={"rank": r})
        sd = {"weight": wts[0][n0:n1], "A": wts[1], "B": wts[2][:, n0:n1].contiguous()}
        if has_bias:
            sd["bias"] = wts[3][n0:n1]
        mod.load_state_dict(sd)
        mod = mod.to(dev).half()
        mod.tuning = o.tuning
        y = mod(xs[K])  # packs the operands (one-time, like the reference's first forward)
        mods.append((mod, xs[K], K, Nl, cnt * layers_here, y, wts, N))
    torch.cuda.synchronize()
    live = [m for m in mods if m[0] is not None]
    if not live:
        layers_here = 0

    # decode sizes: a model walks ~3.6 GB of DISTINCT weights per token, so one 9.4 MB image re-run from the Infinity Cache
    # says little - R copies of the Linear (own packed images; same values, same output buffer) are walked round robin
    rotate = o.rotate if o.rotate is not None else (48 if (M <= 64 and len(shapes) == 1 and layers == 1 and not shard_n) else 0)
    if rotate and not (M <= 64 and len(live) == 1):
        sys.exit("--rotate is for the single-Linear decode workloads")
    rot_mods = list(live)
    if rotate > 1:
        import copy

        rot_mods += [(copy.deepcopy(live[0][0]),) + live[0][1:] for _ in range(rotate - 1)]
        for m in rot_mods[1:]:
            m[0](m[1])  # (its launch cache)
        torch.cuda.synchronize()

    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    gemm_events = []
    launch_no = [0]
    # bracket every n-th launch of the dominant kernel with HIP events: at least 8 samples inside the timed region
    # whatever --steps is (the driver's 20-step C2 run: 8 evenly spread), at most every 10th (a pair costs ~12 us of gaps)
    EV_TOTAL = steps * sum(m[4] for m in live)  # timed launches of the dominant kernel
    EV_EVERY = max(1, min(10, EV_TOTAL // 8))
    ev_on = [True]
    ev_sample = (lambda i: i % EV_EVERY == 0) if EV_EVERY >= 10 or EV_TOTAL < 8 else \
        (lambda i: i == 0 or (i * 8) // EV_TOTAL != ((i - 1) * 8) // EV_TOTAL)
    # several projection shapes: every shape gets its own samples (>= 4 where it has that many timed launches; per_shape of the
    # roofline object) - the global rule above could leave a rare shape without one
    shape_total = {(m[2], m[3]): steps * m[4] for m in live}
    shape_seen = {k: 0 for k in shape_total}

    def ev_sample_shape(K, N):
        tot, i = shape_total[(K, N)], shape_seen[(K, N)]
        shape_seen[(K, N)] = i + 1
        every = max(1, min(10, tot // 4))
        return i % every == 0

    # per-module launch constants (descriptor, workspace carving), built once: decode-size steps are host-bound
    plans = []
    distinct = layers > 1 and not o.shared_weights and rotate <= 1
    distinct_keep = []  # (the cloned images stay alive for the run)
    ws = ops.workspace(dev, max([ops.linear_sizes(mod._desc(), M).workspace for mod, *_ in live] + [256]))  # one buffer for all
    for mod, xd, K, N, reps, y, _, _ in rot_mods:
        desc = mod._desc()
        if mod._x_i8 and L.lqer_gemm_route(C.byref(desc), M, _lib.F16) != _lib.ROUTE_I8:
            desc = mod._desc(plain=True)  # token counts the int8 tile kernel does not serve: the bf16 kernels, same buffers
        p = mod._packed
        Kp, Mp = L.lqer_padded_k(K), L.lqer_padded_m(M)
        xl, al = ops.desc_limbs(desc)  # bf16 limbs of the activation / x A images (1, 1 unless pass-through)
        xq = ws.data_ptr()
        xaq = xq + L.lqer_act_image_bytes(C.byref(desc), M)  # (the image buffer: wider for weights of 5..8 bits)
        if mod._x_f16 and K % 64 == 0 and (M % 256 == 0 or M <= 64):
            xq = xd.data_ptr()  # fp16 route: a dense, aligned fp16 tensor is its own activation image (include/lqer_hip.h)
        rp = L.lqer_padded_r(r)
        xscr = xaq + ((Mp * rp * 2 * al + 255) // 256) * 256
        nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
        gscr = L.lqer_linear_gemm_scratch_bytes(C.byref(desc), M)
        if L.lqer_decode_partials(C.byref(desc), M):
            xaq, gscr = None, nscr  # decode route: the GEMM reduces the partial tiles of x A left in the scratch itself
        a_t, a_limbs = mod._side_image(M, desc, _lib.F16)  # (the image the module itself passes at this token count)
        # model sweeps: every Linear of the model owns its packed operands (same values, distinct addresses - layers differ
        # in values, not in cost, but a weight that is re-read from the Infinity Cache 32 times is not what a model does)
        copies = []
        if distinct and reps > 1:
            a_key = {-1: "a_t_f16", -2: "a_t_b16"}.get(a_limbs, "a_t")
            for _ in range(reps - 1):
                cw, ca, cb = p["w"].clone(), p[a_key].clone(), p["b_t"].clone()
                cbias = p["bias"].clone() if p.get("bias") is not None else None
                distinct_keep.append((cw, ca, cb, cbias))
                copies.append((cw.data_ptr(), ca.data_ptr(), cb.data_ptr(), ops._ptr(cbias)))
        plans.append(dict(desc=desc, dref=C.byref(desc), x=xd.data_ptr(), a_t=a_t, a_limbs=a_limbs, xq=xq, xaq=xaq, copies=copies,
                          ws=ws.data_ptr(), ws_bytes=ws.numel(),
                          xscr=xscr, nscr=nscr, w=p["w"].data_ptr(),
                          b_t=p["b_t"].data_ptr(), b_limbs=p["b_limbs"], bias=ops._ptr(p.get("bias")), y=y.data_ptr(),
                          gscr=gscr, K=K, N=N, reps=reps, route=L.lqer_gemm_route(C.byref(desc), M, _lib.F16),
                          # 128-row tiles: lqer_linear_forward skips the reduce launch (xaq == NULL: the GEMM sums the partial tiles)
                          part=bool(r > 0 and a_limbs == 1 and L.lqer_tile_partials(C.byref(desc), M, _lib.F16))))

    # M <= 8 with block_fp activations in blocks of 16: lqer_linear_forward issues ONE launch
    one_launch = (M <= 8 and r > 0 and bool(plans) and
                  all(L.lqer_decode_partials(pl["dref"], M) and pl["a_limbs"] == 1 for pl in plans))

    # the C-ABI calls of a step with their arguments bound once per stream (the launch stream, or the capture stream of
    # --graph): at decode sizes the Python that assembles 16 arguments per call costs as much as the kernel it launches
    # (the split calls are the ABI-13 pair: the hand-over of the GEMM pre-pass's zero fill that lqer_linear_forward does internally)
    fwd, qxa, gemm = L.lqer_linear_forward, L.lqer_quantize_act_xa_prep, L.lqer_linear_gemm_prepared
    ready = C.c_size_t(0)  # written by every activation call, read by the GEMM call right behind it (one host thread)
    ready_ref = C.byref(ready)
    no_prep = bool(os.environ.get("LQER_BENCH_NO_PREP"))  # (A/B: the GEMM zero-fills its cells itself, as lqer_linear_gemm does)
    gemm_call = lambda ga, rdy=0: gemm(*ga[:-1], 0 if no_prep else rdy, ga[-1])
    bound = {}
    # event pairs for the sampled launches, created ahead of the timed region (creating one costs more host time than a
    # decode-size kernel runs)
    ev_flags = int(os.environ.get("LQER_BENCH_EVENT_FLAGS", "0x20000000"), 0)
    new_pair = lambda: (HipEvent(ev_flags), HipEvent(ev_flags))
    ev_pool = [new_pair() for _ in range(64)]

    def calls_for(st, pls=None):
        pls = plans if pls is None else pls
        if (st, id(pls)) not in bound:
            rows = []
            for pl in pls:
                K, N = pl["K"], pl["N"]
                fa = (pl["dref"], pl["x"], _lib.F16, M, K, pl["w"], pl["a_t"], pl["b_t"], pl["a_limbs"], pl["b_limbs"], pl["bias"],
                      pl["y"], N, pl["ws"], pl["ws_bytes"], st)
                xaq, gscr = (None, pl["nscr"]) if pl["part"] else (pl["xaq"], pl["gscr"])
                qa = (pl["dref"], pl["x"], _lib.F16, M, K, pl["a_t"], pl["a_limbs"], pl["xq"], xaq, pl["xscr"], pl["nscr"], pl["xscr"], ready_ref, st)
                ga = (pl["dref"], pl["xq"], M, pl["w"], xaq, pl["b_t"], pl["b_limbs"], pl["bias"], pl["y"], _lib.F16, N,
                      pl["xscr"], gscr, st)
                per_unit = [(fa, qa, ga)]
                for cw, ca, cb, cbias in pl["copies"]:  # the other Linears of this shape: own weight / A / B / bias images
                    per_unit.append(((pl["dref"], pl["x"], _lib.F16, M, K, cw, ca, cb, pl["a_limbs"], pl["b_limbs"], cbias,
                                      pl["y"], N, pl["ws"], pl["ws_bytes"], st),
                                     (pl["dref"], pl["x"], _lib.F16, M, K, ca, pl["a_limbs"], pl["xq"], xaq, pl["xscr"],
                                      pl["nscr"], pl["xscr"], ready_ref, st),
                                     (pl["dref"], pl["xq"], M, cw, xaq, cb, pl["b_limbs"], cbias, pl["y"], _lib.F16, N,
                                      pl["xscr"], gscr, st)))
                rows.append((pl["reps"], K, N, per_unit))
            bound[(st, id(pls))] = rows
        return bound[(st, id(pls))]

    rot_no = [0]
    resident = [False]  # True: every step re-runs plan 0 (the weight stays in the Infinity Cache)

    def step(timed: bool, stream=stream):
        rows = calls_for(stream)
        if rotate > 1:
            rows = rows[:1] if resident[0] else rows[rot_no[0] % rotate: rot_no[0] % rotate + 1]
            rot_no[0] += 1
        for reps, K, N, per_unit in rows:
            for u in range(reps):
                fa, qa, ga = per_unit[u % len(per_unit)]
                # (counts timed launches only: the first one is always sampled)
                ev = timed and ev_on[0] and (ev_sample(launch_no[0]) if len(shape_total) == 1 else ev_sample_shape(K, N))
                if timed:
                    launch_no[0] += 1
                if one_launch:
                    # up to 8 tokens the whole forward is ONE launch (csrc/decode1.hip) behind lqer_linear_forward - the entry
                    # point of INTEGRATION.md; the events bracket that launch
                    if ev:
                        e0, e1 = ev_pool.pop() if ev_pool else new_pair()
                        e0.record(stream)
                    rc = fwd(*fa)
                    if rc:
                        _lib.check(rc, "linear_forward")
                    if ev:
                        e1.record(stream)
                        gemm_events.append((e0, e1, K, N))
                    continue
                # the two calls of lqer_linear_forward, issued separately so that the dominant kernel can be
                # bracketed with HIP events on the launch stream
                rc = qxa(*qa)
                if rc:
                    _lib.check(rc, "quantize_act_xa")
                if ev:
                    e0, e1 = ev_pool.pop() if ev_pool else new_pair()
                    e0.record(stream)
                rc = gemm_call(ga, ready.value)
                if rc:
                    _lib.check(rc, "linear_gemm")
                if ev:
                    e1.record(stream)
                    gemm_events.append((e0, e1, K, N))

    mrot_no = [0]

    def step_module():
        if rotate > 1:
            mod, xd = rot_mods[mrot_no[0] % rotate][:2]
            mrot_no[0] += 1
            return mod(xd)
        for mod, xd, K, N, reps, _, _, _ in live:
            for _ in range(reps):
                mod(xd)

    # Device clock ramp (setup, like the packing above): after idling the GPU needs tens of milliseconds of load to
    # reach the clocks it then holds - a 60-step run (5 ms) would measure the ramp, not the kernels (C2: 850 vs 970
    # TFLOP/s-equiv on the same box).  Untimed; the W warm-up steps and the K timed steps follow unchanged.
    if layers_here > 0:
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < o.prewarm_ms * 1e-3:
            for _ in range(1 if big else 20):
                step(False)
            torch.cuda.synchronize()
    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    graph = None
    if o.graph:
        # launch-bound steps (decode sizes: three ~3 us kernels): capture one step in a hipGraph and replay it.  The
        # kernels cannot be bracketed with events inside a graph, so the roofline sample is taken from ungraphed
        # launches after the timed region.
        if steps % o.graph:
            sys.exit("--steps must be a multiple of --graph")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(o.graph):
                step(False, torch.cuda.current_stream(dev).cuda_stream)
        graph.replay()
        torch.cuda.synchronize()

    def run_abi(n):
        if graph is not None:
            for _ in range(n // o.graph):
                graph.replay()
        else:
            for _ in range(n):
                step(True)

    elapsed_rank = timed_region(ctx, run_abi, steps)
    if graph is not None:
        for _ in range(min(steps, 4 * EV_EVERY)):
            step(True)
        torch.cuda.synchronize()
    elapsed = sweep.max_over_ranks(elapsed_rank, dev)

    # second timed region: the same K steps through the drop-in module (torch.empty, descriptor cache, ctypes marshalling
    # included) - the boundary the reference's callers use.  Right behind the first region, in the same thermal state: behind the clock probe's
    # seconds of back-to-back GEMMs (below) it read 7-11 % slower than the C ABI at the single-Linear workloads, where the three doors -
    # split calls, lqer_linear_forward, the module - take the same time to a tenth of a microsecond (tools/fwd_paths_time.py)
    module = None
    if o.module and live:
        for _ in range(max(1, warmup // 2)):
            step_module()
        el_mod = sweep.max_over_ranks(timed_region(ctx, lambda n: [step_module() for _ in range(n)], steps), dev)
        module = {"ms_per_step": round(el_mod / steps * 1e3, 4), "vs_c_abi": round(el_mod / elapsed, 4)}
        if M <= 64 and graph is None:
            # decode sizes are host-bound through the module (torch.empty + ctypes per call ~8 us on a ~8 us kernel): the way a
            # serving loop runs them is ONE captured graph per token step - here G module forwards (one per rotated weight)
            # captured by lqer_amd.graph.GraphedCallable and replayed; the one-launch decode route is capturable
            from lqer_amd.graph import GraphedCallable

            G = max(rotate, 1) * max(1, 48 // max(rotate, 1))
            while steps % G:
                G -= 1
            gm = GraphedCallable(lambda: [step_module() for _ in range(G)][-1], warmup=2)
            gm()
            el_g = sweep.max_over_ranks(timed_region(ctx, lambda n: [gm() for _ in range(n // G)], steps), dev)
            module.update(graph_ms_per_step=round(el_g / steps * 1e3, 4), graph_vs_c_abi=round(el_g / elapsed, 4),
                          graph_forwards_per_replay=G)


    # the same K steps once more WITHOUT the event pairs (the instrumented region above is what the contract asks
    # `value` and the roofline sample to share; this one says what the instrumentation costs)
    uninstrumented = None
    if o.uninstrumented and graph is None:
        ev_on[0] = False
        el_u = sweep.max_over_ranks(timed_region(ctx, run_abi, steps), dev)
        ev_on[0] = True
        uninstrumented = el_u

    # the shader clock the chip holds under this very load: a few one-wave probe workgroups on a side stream
    # (lqer_clock_probe: s_memtime / s_memrealtime) in the shadow of the same steps, right behind the timed regions
    # Round 6: the load beside the probe is the DOMINANT KERNEL alone, launched back to back on the images the last
    # step left (a whole step averages the quantizer launches and the gaps in: it read 2.0-2.1 GHz where the int8 GEMM's own stamps read
    # 1.63-1.75); a probe window that did not lie inside the load is reported as null, never as a clock.
    def run_gemm_only(n):
        rows = calls_for(stream)
        for _ in range(n):
            for reps, K, N, per_unit in rows:
                for u in range(reps):
                    rc = gemm_call(per_unit[u % len(per_unit)][2])
                    if rc:
                        _lib.check(rc, "linear_gemm")

    sustained_mhz = None
    if M > 64 and layers_here > 0 and graph is None and not one_launch:
        gemm_s = max(sum(e0.elapsed_time(e1) for e0, e1, _, _ in gemm_events[:64]) / max(len(gemm_events[:64]), 1) * 1e-3, 1e-6) \
            if gemm_events else elapsed / steps
        per_step_launches = sum(reps for reps, _, _, _ in calls_for(stream))
        sustained_mhz = _sustained_clock(ctx, L, run_gemm_only, steps, gemm_s * per_step_launches)
        # (the GEMM-only launches multiplied whatever images the last quantizer call of a shape left - Linears that share a workspace got
        # each other's x A: one whole step puts every output buffer back to what the parity check below expects)
        step(False)
        torch.cuda.synchronize()

    # decode workloads: the same steps once more on ONE resident weight (what rounds 1-2 reported: an upper bound)
    resident_fig = None
    if rotate > 1 and graph is None:
        rot_events, gemm_events = gemm_events, []
        resident[0] = True
        for _ in range(warmup):
            step(False)
        el_res = sweep.max_over_ranks(timed_region(ctx, run_abi, steps), dev)
        resident[0] = False
        res_events, gemm_events = gemm_events, rot_events
        resident_fig = {"ms_per_step": round(el_res / steps * 1e3, 4), "events": res_events}

    # third timed region (model workloads): the same Linears the way the model runs them - q/k/v and
    # gate/up receive ONE tensor, so its activation image and one side GEMM over the members' concatenated A are made once
    # per group (lqer_amd.linear.SharedActivation; same quantizers and GEMM kernels, results as member by member).  The
    # headline `value` stays the conservative one: every Linear quantizes its own input, as the reference's modules do.
    model_shared = None
    # (secondary figures are single-rank only: a rank that fails or owns no layer would leave the others in a collective)
    if o.module and layers > 1 and layers_here > 0 and world == 1:
        model_shared = _shared_inputs_region(ctx, live, layers_here, warmup, steps, elapsed)

    # secondary figure: the forwards of this workload are INDEPENDENT units - issued alternately on two HIP streams
    # (own activation / x A images, scratch and output per stream) the second queue's quantizer, side GEMM and store phases
    # run under the other forward's main loop.  A throughput figure for sweeps and serving batches; `value` stays the
    # one-stream figure (a model's Linears form a chain: 8d sums their times)
    two_streams = None
    if o.two_streams and layers_here > 0 and M > 64 and graph is None and world == 1:
        two_streams = _two_streams_region(ctx, live, plans, ws, calls_for, qxa, gemm_call, ready, M, warmup, steps, elapsed)

    # decode workloads (M <= 8): the Linears a model hands the SAME token to - q/k/v - as ONE launch (lqer_linear_forward_group)
    group_fig = None
    if one_launch and layers == 1 and len(live) == 1 and world == 1 and graph is None and not shard_n:
        group_fig = _decode_group_region(ctx, live[0], M, r, has_bias, rotate, warmup, steps, ev_flags,
                                         n_members=int(os.environ.get("LQER_BENCH_GROUP_MEMBERS", "3")))  # (2..4: probe of the group size)

    # ---- gather (outside the timed regions): per-rank elapsed time, a checksum of the first unit's output
    ysum = float(live[0][5].float().sum().item()) if live else 0.0
    gathered = sweep.gather_rows([elapsed_rank * 1e3 / steps, ysum, float(layers_here)], dev)

    if shard_n:  # `value` of a column-parallel Linear: the UNSHARDED Linear's multiplies over the slowest rank's time
        flops_all = float(sum(flops(M, K, N, r) * cnt * layers for K, N, cnt in shapes))
    else:
        flops_rank = sum(flops(M, K, N, r) * reps for _, _, K, N, reps, _, _, _ in live)
        flops_all = sweep.sum_over_ranks(float(flops_rank), dev)
    ms_per_step = elapsed / steps * 1e3
    value = flops_all * steps / elapsed / 1e12

    # ---- column-parallel: assemble y from every rank's columns (all-gather, outside the timed regions) for the check
    y_full = {}
    if shard_n:
        for mod, xd, K, Nl, reps, y, wts, N in mods:
            y_full[N] = sweep.all_gather_columns(y, col_ranges[N], N)

    # ---- parity of what was just timed: row slices of every unit's output buffer against the CPU oracle (rank 0)
    parity = None
    parity_rows = None
    if o.check and rank == 0:
        from oracle import lqer_oracle as O  # the checker - after the timed regions, never inside them

        idx = check_rows(M, every=len(mods) == 1)
        parity_rows = int(len(idx))
        worst = 0.0
        h = lambda t: None if t is None else t.half().float()
        for mod, xd, K, Nl, reps, y, wts, N in mods:
            if not shard_n and mod is None:
                continue
            ref = O.lqer_linear_forward(xd[idx.to(dev)].float().cpu(), h(wts[0]), h(wts[3]) if has_bias else None, h(wts[1]), h(wts[2]), qc)
            got = (y_full[N] if shard_n else y)[idx.to(dev)].float().cpu()
            worst = max(worst, float((got - ref).norm() / ref.norm()))
        parity = worst
        print(f"# [{o.workload}] parity vs CPU oracle ({len(idx)} rows x {len(mods)} shapes): rel-L2 {parity:.3e}", file=sys.stderr)

    if rank != 0:
        return None

    # ---- roofline of the dominant kernel (benchlib/roofline.py) from the event samples taken inside the timed region
    cal_x = (live[0][1], live[0][2]) if live else None
    ev_overhead_ms = RL.event_pair_overhead_ms(L, _lib, ops, dev, stream, cal_x, new_pair, M) if live else 0.0
    routes = sorted({pl["route"] for pl in plans})
    int8 = routes == [_lib.ROUTE_I8]  # every GEMM of the step ran the int8 MFMA main loop
    rl = RL.mfma_roofline(gemm_events, ev_overhead_ms, M, r, routes, int8, one_launch, _lib, o.workload, ev_flags)
    if sustained_mhz is not None and rank == 0:
        # peaks are priced at the 2.4 GHz the data sheet quotes; at the clock the chip actually holds under this load the same
        # kernel can reach peak x sustained / 2400 at most: frac_at_sustained_clock = loop efficiency x everything but the clock
        rl["sustained_clock"] = sustained_mhz
        rl["inside_load"] = bool(sustained_mhz["inside_load"])
        if sustained_mhz["inside_load"]:
            rl["sustained_mhz"] = round(sustained_mhz["median_mhz"], 1)
            rl["frac_at_sustained_clock"] = round(rl["frac"] * RL.NOMINAL_MHZ / sustained_mhz["median_mhz"], 4)
            for ps in rl.get("per_shape", []):
                ps["frac_at_sustained_clock"] = round(ps["frac"] * RL.NOMINAL_MHZ / sustained_mhz["median_mhz"], 4)
        else:  # (the probe window missed the load: whatever it read is not this kernel's clock)
            rl["sustained_mhz"] = None
            rl["frac_at_sustained_clock"] = None
    if M <= 64:
        rl = RL.hbm_roofline(gemm_events, ev_overhead_ms, M, r, has_bias, routes, one_launch, _lib, o.workload, ev_flags, rotate,
                             ms_per_step, resident_fig)
    f16x = bool(live) and live[0][0]._x_f16
    out = {
        "metric": "W4A8+rank-r Linear GEMM TFLOPS-equiv",
        "value": round(value, 2),
        "unit": "TFLOP/s-equiv",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        # the arithmetic type of the main loop's MFMA operands
        "dtype": "int8" if (M > 64 and int8) else ("f16" if f16x else "bf16"),
        "data": "synthetic",
        "config": {"workload": desc_txt + (f" [{rotate} distinct packed weights walked round robin: {rotate * 9.4:.0f} MB > the 256 MB "
                                            "Infinity Cache]" if rotate > 1 else ""), "tokens_per_step": M, "rank": r,
                   "formats": ("x %%s, W MXINT%d/%%s, A_out,B_out as x, y fp16" % qc["w_quantizer"].get("width", 4)) % (
                       "MXINT8/%s" % qc["x_quantizer"]["block_size"][-1] if qc["x_quantizer"]["name"] == "block_fp"
                       else ("fp16 pass-through (fp16 MFMA main loop)" if f16x else "fp16 pass-through (2 bf16 limbs)"),
                       qc["w_quantizer"]["block_size"][-1]),
                   "boundary": ("C ABI (lqer_linear_forward per Linear: one launch)" if one_launch else
                                "C ABI (lqer_quantize_act_xa + lqer_linear_gemm per Linear, pre-built plans)") +
                               "; the nn.Module figure is in `module`",
                   "sharding": ("ONE Linear column-parallel: rank g owns W[n0:n1], B[:, n0:n1] (cuts at multiples of 16 - exact, no "
                                "reduction), y[:, n0:n1] all-gathered outside the timed region" if shard_n else
                                ("decoder layers split over the ranks, ceil(L/G) consecutive layers each (infer_device_map.py:29-37)"
                                 if strong else "every rank runs its own Linear unit(s) of the workload")) +
                               "; x broadcast from rank 0 and per-rank results gathered outside the timed region, no data-path collective",
                   "layers_per_rank": [int(row[2]) for row in gathered],
                   "column_shard": {str(N): [list(c) for c in rg] for N, rg in col_ranges.items()} if shard_n else None,
                   "weights": ("one packed image set per Linear of the model: %.2f GB walked per step on rank 0" % (
                       sum(sum(t.numel() * t.element_size() for t in c if t is not None) for c in distinct_keep) / 1e9
                       + sum(m[0]._packed["w"].numel() for m in live) / 1e9) if distinct else
                       ("one packed image set per projection shape, re-run for every layer" if layers > 1 else "one Linear"))},
        "tokens_per_s": round(M * steps / elapsed * (1 if strong else world), 1),
        "launch": ("hipGraph replay, %d steps per graph" % o.graph) if graph is not None else "direct launches",
        "prewarm_ms": o.prewarm_ms,
        "broadcast_ms": round(broadcast_ms, 3),
        "roofline": rl,
        # the same K steps without the roofline's event pairs (what the instrumentation inside the timed region costs)
        "uninstrumented": None if uninstrumented is None else {
            "ms_per_step": round(uninstrumented / steps * 1e3, 4), "value": round(flops_all * steps / uninstrumented / 1e12, 2)},
        "module": module,
        # (decode) three Linears that share their input - q/k/v - as one launch: per-Linear time, the launch against HBM, and the
        # same through the eager nn.Modules behind a SharedActivation
        "group": group_fig,
        # (model workloads) q/k/v and gate/up sharing one quantized input, as the model runs them; `value` does not use it
        "model_shared_inputs": model_shared if model_shared is None or "error" in model_shared else dict(
            model_shared, value=round(flops_all / (model_shared["ms_per_step"] * 1e-3) / 1e12, 2)),
        # independent forwards alternating on two HIP streams (throughput of sweeps / serving batches; not `value`)
        "two_streams": two_streams if two_streams is None or "error" in two_streams else dict(
            two_streams, value=round(flops_all / (two_streams["ms_per_step"] * 1e-3) / 1e12, 2)),
        "parity_rel_l2": None if parity is None else float(f"{parity:.3e}"),
        "parity_rows": parity_rows,
        "rank_ms_per_step": [round(row[0], 4) for row in gathered],
        "rank_checksums": [round(row[1], 3) for row in gathered],
    }
    if world == 1 and o.cpu_base:
        K0, N0, _ = shapes[0]
        out["cpu_baseline"] = cpu_baseline(M, K0, N0, r, qc)
    if parity is not None:
        assert parity <= 1e-3, f"[{o.workload}] parity of the timed outputs vs the CPU oracle: rel-L2 {parity:.3e} > 1e-3"
    return out


def _sustained_clock(ctx, L, run_abi, steps, s_per_step):
    """Median shader clock (MHz) over 4 probe waves while `run_abi` keeps the chip under the workload's own load.  The probe goes
    FIRST, on a side stream (a workgroup that comes second would wait for a free register file until the load has drained), the
    steps follow at once on the launch stream and outlast it; the probe counts the last three quarters of its window.  Events on
    both streams say whether the window lay inside the load (`inside_load`)."""
    dev = ctx.dev
    try:
        n = max(2, min(steps, int(4e-3 / max(s_per_step, 1e-6)) + 1))
        dur_us = int(min(max(0.6 * n * s_per_step * 1e6, 200.0), 20000.0))
        n = max(n, int(dur_us * 1e-6 / max(s_per_step, 1e-6) * 1.4) + 1)  # (the load outlasts the probe)
        main = torch.cuda.current_stream(dev)
        buf = torch.zeros(8, dtype=torch.int64, device=dev)
        run_abi(min(n, 4))  # (the clock the load holds, not the ramp from idle)
        rec, sides = None, []
        # Two streams can share a hardware queue: the load then waits BEHIND the probe, which reads the idle clock (2404 MHz in configs.c5
        # of round 5) while every other sign says "inside".  The load's start event tells - it completes at once when the streams run
        # side by side, only after the probe when they are serialized -, and a fresh side stream lands on another queue: up to 4 attempts.
        for attempt in range(4):
            side = torch.cuda.Stream(dev)
            sides.append(side)  # (kept alive: a destroyed stream's queue slot would be handed out again)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(side)
            rc = L.lqer_clock_probe(buf.data_ptr(), 4, dur_us, side.cuda_stream)
            ev[1].record(side)
            if rc:
                return None
            ev[2].record(main)
            run_abi(n)
            ev[3].record(main)
            torch.cuda.synchronize()
            pairs = buf.cpu().view(4, 2).double()
            mhz = sorted(float(c / t * 100.0) for c, t in pairs.tolist() if t > 0)
            if not mhz:
                return None
            load_ms, probe_end_to_load_end = ev[2].elapsed_time(ev[3]), ev[1].elapsed_time(ev[3])
            start_lag_ms = ev[0].elapsed_time(ev[2])
            side_by_side = start_lag_ms < 0.25 * dur_us * 1e-3
            rec = {"median_mhz": round((mhz[(len(mhz) - 1) // 2] + mhz[len(mhz) // 2]) / 2, 1), "min_mhz": round(mhz[0], 1),
                   "max_mhz": round(mhz[-1], 1), "probe_us": dur_us, "steps_under_probe": n, "load_ms": round(load_ms, 3),
                   "inside_load": bool(side_by_side and probe_end_to_load_end >= 0.0 and load_ms * 1e3 >= dur_us),
                   "load_start_lag_ms": round(start_lag_ms, 3), "attempts": attempt + 1,
                   "how": "lqer_clock_probe: 4 one-wave workgroups on a side stream, d(s_memtime) / d(s_memrealtime) x 100 MHz over the last "
                          "3/4 of the probe window, back-to-back launches of the dominant GEMM alone (no quantizer launches, the images of "
                          "the last timed step) running beside it right behind the timed region; a fresh side stream per attempt until "
                          "the load starts beside the probe; null when the window missed the load"}
            if rec["inside_load"]:
                break
        return rec
    except Exception as e:  # noqa: BLE001 (a diagnostic must not cost the bench line)
        print(f"# sustained clock probe failed: {type(e).__name__}: {e}", file=sys.stderr)
        return None


def _shared_inputs_region(ctx, live, layers_here, warmup, steps, elapsed):
    import copy

    from lqer_amd import sweep
    from lqer_amd.linear import SharedActivation

    try:
        units = []  # per shape: (group members, solo module or None, solo calls per layer, x)
        refused = False  # a shape with q/k/v or gate/up whose group is disabled
        for mod, xd, K, N, reps, _, _, _ in live:
            cnt = reps // layers_here
            gsz = 3 if (cnt >= 3 and K == N) else (2 if cnt == 2 else 0)
            solo_n = cnt - gsz
            copies = [copy.deepcopy(mod) for _ in range(max(gsz - 1, 0) + (1 if solo_n else 0))]
            solo = copies.pop() if solo_n else None
            members = []
            if gsz:
                grp = SharedActivation([mod] + copies)
                members = grp.members if grp.enabled else []
                if not grp.enabled:
                    solo, solo_n = mod, cnt
                    refused = True
            units.append((members, solo, solo_n, xd))
        if not any(members for members, _, _, _ in units) and not refused:
            return None
        if not any(members for members, _, _, _ in units):
            # (C5: q/k/v at rank 128 concatenate to a padded rank of 384 > the side GEMM's 256.  Measured in round 4 with the
            # group cut into chunks that share one quantized image: 1.613 vs 1.581 ms per two layers - the side GEMM on the
            # image costs what the fused quantizer + side GEMM costs, so such groups stay disabled)
            return {"error": "no group applies (padded ranks of q/k/v sum to more than 256: sharing measured slower, NOTEBOOK.md section 9.2)"}

        def step_shared():
            for _ in range(layers_here):
                for members, solo, solo_n, xd in units:
                    for m in members:
                        m(xd)
                    for _ in range(solo_n):
                        solo(xd)

        for _ in range(max(1, warmup // 2)):
            step_shared()
        el_sh = sweep.max_over_ranks(timed_region(ctx, lambda n: [step_shared() for _ in range(n)], steps), ctx.dev)
        return {"ms_per_step": round(el_sh / steps * 1e3, 4), "vs_c_abi": round(el_sh / elapsed, 4),
                "groups_per_layer": [len(members) for members, _, _, _ in units if members]}
    except Exception as e:  # (a secondary figure must not cost the bench line)
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def _two_streams_region(ctx, live, plans, ws, calls_for, qxa, gemm_call, ready, M, warmup, steps, elapsed):
    from lqer_amd import _lib, sweep

    dev = ctx.dev
    try:
        # (two explicit streams: the legacy default stream this script otherwise launches on serialises with every other stream)
        s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ws2 = torch.empty(ws.numel(), dtype=torch.uint8, device=dev)
        plans2, ys2 = [], []
        for pl in plans:
            q = dict(pl)
            for k in ("xq", "xaq", "xscr", "ws"):
                if q[k] is not None and ws.data_ptr() <= q[k] < ws.data_ptr() + ws.numel():
                    q[k] = q[k] - ws.data_ptr() + ws2.data_ptr()
            ys2.append(torch.empty(M, q["N"], dtype=torch.float16, device=dev))
            q["y"] = ys2[-1].data_ptr()
            plans2.append(q)
        unit_no, on_b = [0], set()

        def step_two():
            ra, rb = calls_for(s1.cuda_stream), calls_for(s2.cuda_stream, plans2)
            for i, ((reps, K, N, pa), (_, _, _, pb)) in enumerate(zip(ra, rb)):
                for u in range(reps):
                    if unit_no[0] % 2:
                        on_b.add((i, u % len(pa)))
                    _, qa, ga = (pa if unit_no[0] % 2 == 0 else pb)[u % len(pa)]
                    unit_no[0] += 1
                    rc = qxa(*qa) or gemm_call(ga, ready.value)
                    if rc:
                        _lib.check(rc, "two-stream step")

        s1.wait_stream(torch.cuda.current_stream(dev))
        s2.wait_stream(torch.cuda.current_stream(dev))
        for _ in range(max(2, warmup // 2)):
            step_two()
        torch.cuda.synchronize()
        for i, ((mod, xd, K, N, reps, y, _, _), y2) in enumerate(zip(live, ys2)):
            # both queues produce the bits of the one-stream run (units of a plan that share one image set)
            if any(pi == i for pi, _ in on_b):  # (every image set of a plan holds the same values)
                assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), "two-stream outputs differ"
        el_two = sweep.max_over_ranks(timed_region(ctx, lambda n: [step_two() for _ in range(n)], steps), dev)
        return {"ms_per_step": round(el_two / steps * 1e3, 4), "vs_one_stream": round(elapsed / el_two, 4)}
    except Exception as e:  # (a secondary figure must not cost the bench line)
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def _decode_group_region(ctx, unit, M, r, has_bias, rotate, warmup, steps, ev_flags, n_members=3):
    """q/k/v at decode sizes: `n_members` Linears of the workload's shape (own packed images each) that are handed the same
    tokens, run as ONE launch through lqer_linear_forward_group; `rotate // n_members` such groups are walked round robin so
    that the weights stream from HBM.  Reports per-Linear step time, the launch's HIP-event time against the HBM peak
    (algorithmic bytes: every member's packed W, B^T, bias and y, the concatenated A^T, x) and the same groups through the
    eager nn.Modules (SharedActivation: one launch + two cached outputs per token step)."""
    import copy

    from lqer_amd import _lib, ops, sweep
    from lqer_amd.linear import SharedActivation

    from .workloads import HBM_PEAK_GBS

    dev = ctx.dev
    try:
        mod, xd, K, N = unit[0], unit[1], unit[2], unit[3]
        L = _lib.lib()
        n_groups = max(1, (rotate or n_members) // n_members)
        groups = []
        for _ in range(n_groups):
            members = [copy.deepcopy(mod) for _ in range(n_members)]
            grp = SharedActivation(members)
            if not grp.enabled:
                return {"error": "the workload's Linears do not form a SharedActivation group"}
            outs = [m(xd) for m in members]  # packs, builds the group plan
            groups.append((grp, members))
        torch.cuda.synchronize()
        dtc = ops.dtype_code(xd)
        plans = [g._dplans.get((M, dtc)) for g, _ in groups]
        if any(p is None for p in plans):
            return {"error": "lqer_linear_forward_group refused the group"}
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = ops.workspace(dev, max(p["ws"] for p in plans))
        ybuf = torch.empty(n_members, M, N, dtype=torch.float16, device=dev)
        calls = []
        for p in plans:
            for i in range(n_members):
                p["tab"][i].y = ybuf[i].data_ptr()
            calls.append((p["tab"], p["n"], xd.data_ptr(), dtc, M, K, p["a_t"], 1, ws.data_ptr(), ws.numel(), stream))
        fwd = L.lqer_linear_forward_group
        no, events = [0], []
        new_pair = lambda: (HipEvent(ev_flags), HipEvent(ev_flags))
        pool = [new_pair() for _ in range(40)]

        def step_abi(timed):
            args = calls[no[0] % n_groups]
            ev = timed and no[0] % 10 == 0 and pool
            no[0] += 1
            if ev:
                e0, e1 = pool.pop()
                e0.record(stream)
            rc = fwd(*args)
            if rc:
                _lib.check(rc, "linear_forward_group")
            if ev:
                e1.record(stream)
                events.append((e0, e1))

        for _ in range(warmup):
            step_abi(False)
        el = sweep.max_over_ranks(timed_region(ctx, lambda n: [step_abi(True) for _ in range(n)], steps), dev)
        mno = [0]

        def step_mod():
            _, members = groups[mno[0] % n_groups]
            mno[0] += 1
            for m in members:
                m(xd)

        for _ in range(max(1, warmup // 2)):
            step_mod()
        el_m = sweep.max_over_ranks(timed_region(ctx, lambda n: [step_mod() for _ in range(n)], steps), dev)
        # the pair overhead: the headline sample's calibration (an empty pair right behind a small kernel)
        ovh = RL.event_pair_overhead_ms(L, _lib, ops, dev, stream, (xd, K), new_pair, M)
        tot_ms = sum(max(e0.elapsed_time(e1) - ovh, 1e-6) for e0, e1 in events)
        nl = max(len(events), 1)
        Kp, Np, rp = -(-K // 64) * 64, -(-N // 256) * 256, -(-r // 16) * 16
        by = n_members * (Np * Kp * 0.5625 + Np * rp * 2 + M * N * 2 + (Np * 4 if has_bias else 0)) + M * K * 2 + n_members * rp * Kp * 2
        gbs = by / (tot_ms / nl * 1e-3) / 1e9 if tot_ms > 0 else 0.0
        for g, members in groups:
            for m in members:
                m._group = None
        return {"members": n_members, "groups_walked": n_groups, "weights_mb_walked": round(n_groups * n_members * Np * Kp * 0.5625 / 1e6, 1),
                "ms_per_step": round(el / steps * 1e3, 4), "ms_per_linear": round(el / steps / n_members * 1e3, 4),
                "kernel": "k_decode1 (one launch for the group)", "avg_launch_us": round(tot_ms / nl * 1e3, 2), "launches": nl,
                "algorithmic_bytes_per_launch": int(by), "achieved": round(gbs, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "frac": round(gbs / HBM_PEAK_GBS, 4), "event_pair_overhead_us": round(ovh * 1e3, 2),
                "module": {"ms_per_step": round(el_m / steps * 1e3, 4), "ms_per_linear": round(el_m / steps / n_members * 1e3, 4),
                           "vs_c_abi": round(el_m / el, 4)}}
    except Exception as e:  # (a secondary figure must not cost the bench line)
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def _decode_layer(ctx, o):
    """Workload d1layer: a token step through Llama-7B decoder layers at M = 1 the way the model issues it - q/k/v as ONE launch
    (lqer_linear_forward_group), o, gate/up as one launch, down - over `layers` distinct layers (own packed images: 113.5 MB each,
    four of them do not fit the Infinity Cache).  Four launches per layer through the C ABI with pre-built plans; every launch type
    is sampled with HIP events; algorithmic bytes per layer = the seven packed weights + B^T + A^T images + x + y."""
    import copy

    import lqer_amd
    from lqer_amd import _lib, ops, sweep
    from lqer_amd.linear import SharedActivation

    from .workloads import HBM_PEAK_GBS

    dev = ctx.dev
    desc_txt, M, r, has_bias, qc, shapes, layers = WORKLOADS["d1layer"]
    if o.layers > 0:
        layers = o.layers
    if ctx.world > 1:
        sys.exit("d1layer is a single-GPU workload")
    steps = o.steps if o.steps is not None else 200
    warmup = o.warmup if o.warmup is not None else 20
    L = _lib.lib()
    H, I = 4096, 11008
    protos = {}
    for i, (K, N) in enumerate(((H, H), (H, I), (I, H))):
        g = torch.Generator().manual_seed(sweep.unit_seed(0, i))
        W, A, B = make_weights(g, K, N, r, bias=False, quantize_ab=True)
        m = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
        m.load_state_dict({"weight": W, "A": A, "B": B})
        protos[(K, N)] = (m.to(dev).half(), (W, A, B))
    xh = make_x(M, H, seed=0)[0].half().to(dev)
    xi = make_x(M, I, seed=1)[0].half().to(dev)
    for (K, N), (m, _) in protos.items():
        m(xh if K == H else xi)  # packs
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev).cuda_stream
    dtc = _lib.F16
    lay = []
    keep = []
    for _ in range(layers):
        q, k, v, op = (copy.deepcopy(protos[(H, H)][0]) for _ in range(4))
        gt, up = (copy.deepcopy(protos[(H, I)][0]) for _ in range(2))
        dn = copy.deepcopy(protos[(I, H)][0])
        g_qkv, g_gu = SharedActivation([q, k, v]), SharedActivation([gt, up])
        assert g_qkv.enabled and g_gu.enabled
        for m in (q, k, v, op, gt, up):
            m(xh)
        dn(xi)
        torch.cuda.synchronize()
        p_qkv, p_gu = g_qkv._dplans.get((M, dtc)), g_gu._dplans.get((M, dtc))
        if p_qkv is None or p_gu is None:
            sys.exit("d1layer: lqer_linear_forward_group refused a group")
        ys = [torch.empty(M, n, dtype=torch.float16, device=dev) for n in (H, H, H, I, I)]
        for i in range(3):
            p_qkv["tab"][i].y = ys[i].data_ptr()
        for i in range(2):
            p_gu["tab"][i].y = ys[3 + i].data_ptr()
        singles = []
        for m, x in ((op, xh), (dn, xi)):
            ent = m._fw_cache[(M, x.dtype, 0)]
            desc, ws_bytes, dt, consts = ent
            y = torch.empty(M, m.out_features, dtype=torch.float16, device=dev)
            singles.append((C.byref(desc), x.data_ptr(), dt, M, m.in_features, consts, y, ws_bytes))
            keep.append((desc, y))
        lay.append((p_qkv, p_gu, singles))
        keep.append((q, k, v, op, gt, up, dn, g_qkv, g_gu, ys))
    ws = ops.workspace(dev, max([pl["ws"] for la in lay for pl in la[:2]] + [sg[7] for la in lay for sg in la[2]] + [256]))
    fwd_g, fwd = L.lqer_linear_forward_group, L.lqer_linear_forward
    ev_flags = int(os.environ.get("LQER_BENCH_EVENT_FLAGS", "0x20000000"), 0)
    new_pair = lambda: (HipEvent(ev_flags), HipEvent(ev_flags))
    pool = [new_pair() for _ in range(96)]
    events = {"qkv": [], "o": [], "gate_up": [], "down": []}
    no = [0]

    def launch(kind, fn, args, timed):
        ev = timed and no[0] % 13 == 0 and pool
        no[0] += 1
        if ev:
            e0, e1 = pool.pop()
            e0.record(stream)
        rc = fn(*args)
        if rc:
            _lib.check(rc, "d1layer " + kind)
        if ev:
            e1.record(stream)
            events[kind].append((e0, e1))

    def step(timed):
        for p_qkv, p_gu, singles in lay:
            launch("qkv", fwd_g, (p_qkv["tab"], 3, xh.data_ptr(), dtc, M, H, p_qkv["a_t"], 1, ws.data_ptr(), ws.numel(), stream), timed)
            d, x, dt, m_, k_, consts, y, _ = singles[0]
            launch("o", fwd, (d, x, dt, m_, k_, *consts, y.data_ptr(), H, ws.data_ptr(), ws.numel(), stream), timed)
            launch("gate_up", fwd_g, (p_gu["tab"], 2, xh.data_ptr(), dtc, M, H, p_gu["a_t"], 1, ws.data_ptr(), ws.numel(), stream), timed)
            d, x, dt, m_, k_, consts, y, _ = singles[1]
            launch("down", fwd, (d, x, dt, m_, k_, *consts, y.data_ptr(), H, ws.data_ptr(), ws.numel(), stream), timed)

    for _ in range(warmup):
        step(False)
    el = timed_region(ctx, lambda n: [step(True) for _ in range(n)], steps)
    torch.cuda.synchronize()
    ovh = RL.event_pair_overhead_ms(L, _lib, ops, dev, stream, (xh, H), new_pair, M)
    # algorithmic bytes (SURVEY 8d's decode model): packed W 0.5625 B per weight, B^T and A^T one bf16 limb, x and y fp16
    rp = -(-r // 16) * 16
    lin = lambda K, N: N * K * 0.5625 + N * rp * 2 + rp * K * 2 + M * K * 2 + M * N * 2
    by = {"qkv": 3 * lin(H, H) - 2 * M * H * 2, "o": lin(H, H), "gate_up": 2 * lin(H, I) - M * H * 2, "down": lin(I, H)}
    by_layer = sum(by.values())
    per = {}
    for kind, evs in events.items():
        ms = sorted(max(e0.elapsed_time(e1) - ovh, 1e-6) for e0, e1 in evs)
        if ms:
            med = ms[len(ms) // 2]
            per[kind] = {"launch_us": round(med * 1e3, 2), "samples": len(ms), "algorithmic_mb": round(by[kind] / 1e6, 2),
                         "gb_s": round(by[kind] / (med * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(by[kind] / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    us_layer = el / steps / layers * 1e6
    gbs = by_layer / (us_layer * 1e-6) / 1e9
    # parity: q of the first layer against the oracle
    parity = None
    if o.check:
        from oracle import lqer_oracle as O

        W, A, B = protos[(H, H)][1]
        h = lambda t: t.half().float()
        ref = O.lqer_linear_forward(xh.float().cpu(), h(W), None, h(A), h(B), qc)
        yq = torch.empty(M, H, dtype=torch.float16, device=dev)
        lay[0][0]["tab"][0].y = yq.data_ptr()
        step(False)
        torch.cuda.synchronize()
        parity = float((yq.float().cpu() - ref).norm() / ref.norm())
        assert parity <= 1e-3, parity
    fl = sum(flops(M, K, N, r) * c for K, N, c in shapes)
    return {"metric": "W4A8+rank-r Linear GEMM TFLOPS-equiv", "value": round(fl / (us_layer * 1e-6) / 1e12, 3), "unit": "TFLOP/s-equiv", "n_gpus": 1,
            "steps": steps, "warmup": warmup, "ms_per_step": round(el / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": desc_txt + f" [{layers} distinct layers: {layers * 113.5:.0f} MB of packed weights walked per token step]",
                       "tokens_per_step": M, "rank": r, "layers_per_rank": [layers],
                       "boundary": "C ABI: lqer_linear_forward_group (q/k/v, gate/up) + lqer_linear_forward (o, down), pre-built plans"},
            "tokens_per_s": round(M * steps / el, 1), "us_per_layer": round(us_layer, 2), "us_per_linear": round(us_layer / 7, 2),
            "launches_per_layer": 4,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "traffic": None, "kernel": "k_decode1 (four launches per layer, launch gaps included: the whole layer step)",
                         "algorithmic_bytes_per_layer": int(by_layer), "per_launch": per, "event_pair_overhead_us": round(ovh * 1e3, 2)},
            "parity_rel_l2": parity}

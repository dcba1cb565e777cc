"""Re-entrancy ("no global mutable state; re-entrant; stream-ordered"): forwards issued alternately on two HIP streams
- kernels of both queues overlapping on the device, each stream with its own workspace (ops.workspace is keyed by device and
stream) - must produce the bits of the same forwards run alone.  One case per route with cross-launch state of its own: the fused
quantizer + tile GEMM, the 128-row fused quantizer (rank 128), the int8 route with its pre-pass (atomic maxima in scratch), the
one-launch decode kernel (tagged granules in the workspace), the two-launch decode route.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import lqer_amd

    return lqer_amd


@pytest.mark.parametrize("cfg,M,K,N,r", [("mxint", 1024, 2048, 2048, 32), ("opt", 768, 1024, 1536, 128), ("int", 1024, 1024, 2048, 64),
                                         ("mxint", 4, 2048, 1024, 32), ("mxint", 40, 2048, 1024, 32)])
def test_forwards_on_two_streams_equal_the_forwards_alone(lq, cfg, M, K, N, r):
    import copy

    from bench import INT_Q, MXINT_Q, OPT_Q, make_weights

    qc = {"mxint": MXINT_Q, "opt": OPT_Q, "int": INT_Q}[cfg]
    bias = cfg == "opt"
    g = torch.Generator().manual_seed(M + K + N)
    wts = make_weights(g, K, N, r, bias=bias, quantize_ab=cfg != "int")
    mod = lq.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": wts[0], "A": wts[1], "B": wts[2]}
    if bias:
        sd["bias"] = wts[3]
    mod.load_state_dict(sd)
    mod = mod.to(DEV).half()
    xs = [torch.randn(M, K, generator=g).half().to(DEV) for _ in range(2)]  # a different input per stream
    ref = [mod(x).clone() for x in xs]
    mods = [mod, copy.deepcopy(mod)]  # (own launch caches; same packed values)
    mods[1](xs[1])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    outs = [[], []]
    for it in range(24):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                outs[i].append(mods[i](xs[i]))
    torch.cuda.synchronize()
    for i in (0, 1):
        for y in outs[i]:
            assert torch.equal(y.view(torch.int16), ref[i].view(torch.int16))


def test_forwards_from_two_host_threads(lq):
    """Two Python threads, each with its own stream and module copy, call the C ABI at the same time (ctypes drops the GIL for the
    call): the library keeps no per-call global state (thread-local error string, once-flags, an atomic launch counter), so both
    threads get the bits of the forwards run alone - tile route and one-launch decode route."""
    import copy
    import threading

    from bench import MXINT_Q, make_weights

    K, N, r = 1024, 1024, 32
    g = torch.Generator().manual_seed(99)
    W, A, B = make_weights(g, K, N, r, quantize_ab=True)
    mod = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).half()
    xs = {M: torch.randn(M, K, generator=g).half().to(DEV) for M in (600, 3)}
    ref = {M: mod(x).clone() for M, x in xs.items()}
    mods = [mod, copy.deepcopy(mod)]
    mods[1](xs[3])
    torch.cuda.synchronize()
    errors = []

    def work(i):
        try:
            s = torch.cuda.Stream(DEV)
            with torch.cuda.stream(s):
                for it in range(40):
                    M = 600 if (it + i) % 2 else 3
                    y = mods[i](xs[M])
                    if it % 8 == 7:
                        s.synchronize()
                        if not torch.equal(y, ref[M]):
                            errors.append((i, it, M))
            s.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors

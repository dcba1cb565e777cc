"""Module-level behaviour of the drop-in Linear on the GPU: quantize-once semantics (reference linear.py:149-153),
shared activations under torch.inference_mode(), launches on a non-current device.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import lqer_oracle as O  # the checker

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import lqer_amd

    return lqer_amd


def _non_idempotent_weight(N, K, seed=0, frac=0.51):
    """Weights whose block maxima round DOWN to a power of two under W4 blocks of 16 (|w| in (0.47, 0.53) * 2^k rounds
    to code 4 = 2^(k-1)): a second quantization pass lowers the block exponent and clamps that element to 7/8."""
    g = torch.Generator().manual_seed(seed)
    W = 0.02 * torch.randn(N, K, generator=g)
    blk = W.reshape(N, K // 16, 16)
    amax = blk.abs().amax(-1, keepdim=True)
    e = torch.ceil(torch.log2(amax))
    scale = (frac * 2.0 ** e) / amax  # block maximum -> frac * 2^e
    return (blk * scale).reshape(N, K)


def test_weight_is_quantized_exactly_once(lq):
    from bench import MXINT_Q, make_case

    M, K, N, r = 40, 256, 192, 32
    x, _, A, B = make_case(M, K, N, r, seed=3)
    W = _non_idempotent_weight(N, K)
    wq = O.get_quantizer(MXINT_Q["w_quantizer"])(W)
    assert not torch.equal(O.get_quantizer(MXINT_Q["w_quantizer"])(wq), wq)  # the premise: Q(Q(W)) != Q(W)
    xd = x.to(DEV)
    ref = O.lqer_linear_forward(x, W, None, A, B, MXINT_Q)

    mod = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W}, strict=False)  # the HF checkpoint first (llama_decoder.py:507) ...
    mod = mod.to(DEV)
    mod(xd)  # ... a forward quantizes the weight in place ...
    assert mod.w_is_quantized and torch.equal(mod.weight.detach().cpu(), wq)
    mod.load_state_dict({"A": A, "B": B}, strict=False)  # ... then the low-rank dict (runners.py:220-222)
    assert mod.w_is_quantized  # only A / B were reloaded
    y = mod(xd).cpu()
    assert torch.equal(mod.weight.detach().cpu(), wq)
    assert (y - ref).norm() / ref.norm() <= 1e-5
    # a dtype cast / device round trip rebuilds the A / B images but keeps Q(W)
    m2 = mod.half()
    y2 = m2(xd.half()).float().cpu()
    ref16 = O.lqer_linear_forward(x.half().float(), wq, None, A.half().float(), B.half().float(), MXINT_Q, weight_is_quantized=True)
    assert (y2 - ref16).norm() / ref16.norm() <= 1e-3
    assert torch.equal(m2.weight.detach().float().cpu(), wq.half().float())
    m3 = m2.cpu().to(DEV)
    assert torch.equal(m3(xd.half()).float().cpu(), y2)
    # reloading the dense weight DOES re-quantize
    m3.load_state_dict({"weight": W.half()}, strict=False)
    assert m3.w_is_quantized is False
    y4 = m3(xd.half()).float().cpu()
    ref4 = O.lqer_linear_forward(x.half().float(), W.half().float(), None, A.half().float(), B.half().float(), MXINT_Q)
    assert (y4 - ref4).norm() / ref4.norm() <= 1e-3


def test_new_weights_written_in_place_are_requantized(lq):
    """invalidate_packed() used to mean "the weights changed".  Now (a) an in-place write to the parameter is
    noticed by the next forward on its own (version counter), (b) a write through .data needs
    invalidate_packed(weight_changed=True), and (c) the flag has no default, so an old-style call fails loudly."""
    from bench import MXINT_Q, make_case

    M, K, N, r = 24, 256, 128, 16
    x, W, A, B = make_case(M, K, N, r, seed=31)
    W2 = make_case(M, K, N, r, seed=32)[1]
    xd = x.to(DEV)

    def fresh(Wv):
        m = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
        m.load_state_dict({"weight": Wv, "A": A, "B": B})
        return m.to(DEV)

    want = fresh(W2)(xd)
    m = fresh(W)
    y1 = m(xd)
    assert not torch.equal(y1, want)
    with torch.no_grad():
        m.weight.copy_(W2.to(DEV))  # (a) bumps weight._version
    assert torch.equal(m(xd), want)
    m = fresh(W)
    m(xd)
    m.weight.data.copy_(W2.to(DEV))  # (b) invisible to the version counter
    with pytest.raises(TypeError):
        m.invalidate_packed()  # (c)
    m.invalidate_packed(weight_changed=True)
    assert m.w_is_quantized is False
    assert torch.equal(m(xd), want)
    # and the other direction: dropping only the derived images does not quantize a second time
    m.invalidate_packed(weight_changed=False)
    assert m.w_is_quantized and torch.equal(m(xd), want)


def test_deepcopy_and_pickle_keep_the_quantized_once_weight(lq):
    """copy.deepcopy / pickle of a module that has run: the copy's parameters are new tensors (fresh version counters) - it must
    keep the copied images and the weight quantized ONCE (block_fp quantization is not idempotent on these weights), not take
    the copy for an in-place write and quantize the already quantized weight again."""
    import copy
    import io

    from bench import OPT_Q, make_case

    M, K, N, r = 40, 256, 192, 32
    x, _, A, B, b = make_case(M, K, N, r, seed=5, bias=True)
    W = _non_idempotent_weight(N, K, seed=2)
    mod = lq.LinearFlexibleLqer(K, N, bias=True, q_config=OPT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B, "bias": b})
    mod = mod.to(DEV).half()
    xd = x.half().to(DEV)
    y0 = mod(xd).clone()
    ref = O.lqer_linear_forward(x.half().float(), W.half().float(), b.half().float(), A.half().float(), B.half().float(), OPT_Q)
    assert float((y0.float().cpu() - ref).norm() / ref.norm()) <= 1e-3
    cp = copy.deepcopy(mod)
    assert torch.equal(cp(xd), y0) and torch.equal(cp(xd), y0)
    assert torch.equal(cp.weight, mod.weight)  # (still the once-quantized values)
    buf = io.BytesIO()
    torch.save(mod, buf)
    buf.seek(0)
    ld = torch.load(buf, weights_only=False)
    assert torch.equal(ld(xd), y0)
    # the copy still notices its OWN in-place writes
    with torch.no_grad():
        cp.weight.copy_(torch.zeros_like(cp.weight))
    y_side = cp(xd)
    assert not torch.equal(y_side, y0) and torch.equal(mod(xd), y0)


def test_bias_is_quantized_exactly_once(lq):
    from bench import OPT_Q, make_case

    M, K, N, r = 9, 128, 64, 16
    x, W, A, B, b = make_case(M, K, N, r, seed=4, bias=True)
    b = _non_idempotent_weight(1, N, seed=5, frac=0.501).reshape(N) * 0.5  # 8-bit: maxima in (0.5, 0.5039) 2^e round to 2^(e-1)
    qb = O.get_quantizer(OPT_Q["b_quantizer"])
    assert not torch.equal(qb(qb(b)), qb(b))  # the premise
    ref = O.lqer_linear_forward(x, W, b, A, B, OPT_Q)
    mod = lq.LinearFlexibleLqer(K, N, bias=True, q_config=OPT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "bias": b, "A": A, "B": B})
    mod = mod.to(DEV)
    y = mod(x.to(DEV)).cpu()
    assert (y - ref).norm() / ref.norm() <= 1e-5
    bq = mod.bias.detach().clone()
    mod.load_state_dict({"A": A, "B": B}, strict=False)
    assert torch.equal(mod(x.to(DEV)).cpu(), y) and torch.equal(mod.bias.detach(), bq)


def test_shared_activation_under_inference_mode(lq):
    """q/k/v handed the same tensor under torch.inference_mode(): inference tensors have no version counter - the group
    must still serve the three members from one quantization, give the results of the members run alone, and start a new
    round when a member comes back."""
    from bench import MXINT_Q, make_case
    from lqer_amd.linear import SharedActivation

    M, K, r = 70, 256, 32
    mods = []
    for i, N in enumerate((192, 64, 128)):
        x, W, A, B = make_case(M, K, N, r, seed=30 + i)
        m = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
        m.load_state_dict({"weight": W, "A": A, "B": B})
        mods.append(m.to(DEV).half())
    xd = x.half().to(DEV)
    alone = [m(xd).clone() for m in mods]
    grp = SharedActivation(mods)
    assert grp.enabled
    with torch.inference_mode():
        h = xd * 1.0  # an inference tensor
        assert h.is_inference()
        got = [m(h) for m in mods]
        assert grp._x is None  # released after the last member
        for a, g in zip(alone, got):
            assert (a.float() - g.float()).norm() / a.float().norm() <= 2e-3  # (side-product summation order may differ)
        again = [m(h) for m in mods]  # second round on the same object: images rebuilt, same bits
        for g, a2 in zip(got, again):
            assert torch.equal(g, a2)
        h2 = xd * 2.0
        got2 = mods[0](h2)
        assert not torch.equal(got2, got[0])
    with torch.no_grad():  # ordinary tensors: an in-place change between two members is noticed through the version counter
        h = xd.clone()
        y0 = mods[0](h)
        h.mul_(2.0)
        y1 = mods[1](h)
        assert (y1.float() - 2 * alone[1].float()).norm() / (2 * alone[1].float()).norm() <= 5e-3


def test_shared_activation_groups_share_one_image_pool(lq):
    """Every group takes its images from ONE per-device pool (64 private copies would pin ~1 GiB for a Llama-7B at M = 2048):
    groups that run one after the other reuse the same bytes, and groups whose members are called INTERLEAVED (another
    group has overwritten the pool between two members) re-make their images instead of reading the other group's."""
    from bench import MXINT_Q, make_case
    from lqer_amd.linear import SharedActivation

    M, K, r = 200, 512, 32
    groups, xs, alone = [], [], []
    for gidx in range(2):
        mods = []
        for i, N in enumerate((256, 128)):
            x, W, A, B = make_case(M, K, N, r, seed=50 + 10 * gidx + i)
            m = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
            m.load_state_dict({"weight": W, "A": A, "B": B})
            mods.append(m.to(DEV).half())
        xd = (x * (1.0 + gidx)).half().to(DEV)
        alone.append([m(xd).clone() for m in mods])
        grp = SharedActivation(mods)
        assert grp.enabled
        groups.append(mods)
        xs.append(xd)
    close = lambda a, b: float((a.float() - b.float()).norm() / a.float().norm()) <= 2e-3
    with torch.no_grad():
        # one group after the other (the order of a decoder layer)
        for mods, xd, ref in zip(groups, xs, alone):
            for m, a in zip(mods, ref):
                assert close(a, m(xd))
        pkey = (xs[0].device, torch.cuda.current_stream(xs[0].device).cuda_stream)  # one pool per (device, stream)
        pool = SharedActivation._pool[pkey]
        ptr = pool["xq"].data_ptr()
        # interleaved: g0.m0, g1.m0, g0.m1, g1.m1
        y00 = groups[0][0](xs[0]); y10 = groups[1][0](xs[1]); y01 = groups[0][1](xs[0]); y11 = groups[1][1](xs[1])
        assert close(alone[0][0], y00) and close(alone[1][0], y10) and close(alone[0][1], y01) and close(alone[1][1], y11)
        assert SharedActivation._pool[pkey]["xq"].data_ptr() == ptr  # still the one pool
        # a second stream gets its own images (two streams driving two groups must not race on one pool)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ys = [m(xs[0]) for m in groups[0]]
        side.synchronize()
        assert all(close(a, y) for a, y in zip(alone[0], ys))
        assert (xs[0].device, side.cuda_stream) in SharedActivation._pool and len(SharedActivation._pool) >= 2
        assert SharedActivation._pool[pkey]["xq"].data_ptr() == ptr
    SharedActivation.release_pool()
    assert not SharedActivation._pool


def test_deepcopy_of_a_model_with_shared_activation_groups(lq):
    """A deep copy of a model copies its groups with it (the members point at their group): the copy's group must not keep the
    original's launch plans - raw device pointers of the ORIGINAL members' images - once the original is gone."""
    import copy
    import gc

    from bench import MXINT_Q, make_case
    from lqer_amd.linear import SharedActivation

    M, K, N, r = 96, 256, 192, 32

    class QKV(torch.nn.Module):
        def __init__(self):
            super().__init__()
            mods = []
            for i in range(3):
                _, W, A, B = make_case(M, K, N, r, seed=20 + i)
                m = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
                m.load_state_dict({"weight": W, "A": A, "B": B})
                mods.append(m)
            self.q, self.k, self.v = mods

        def forward(self, x):
            return self.q(x), self.k(x), self.v(x)

    net = QKV().to(DEV).half()
    grp = SharedActivation([net.q, net.k, net.v])
    assert grp.enabled
    x = make_case(M, K, N, r, seed=1)[0].half().to(DEV)
    y0 = [t.clone() for t in net(x)]
    y0b = [t.clone() for t in net(x)]
    assert all(torch.equal(a, b) for a, b in zip(y0, y0b))
    cp = copy.deepcopy(net)
    assert cp.q._group is not grp and cp.q._group is cp.k._group and cp.q._group.members[0] is cp.q
    del net, grp
    gc.collect()
    torch.cuda.empty_cache()
    junk = torch.full((64 << 20,), 0x7F, dtype=torch.uint8, device=DEV)  # (what the freed images' memory may now hold)
    y1 = cp(x)
    assert all(torch.equal(a, b) for a, b in zip(y0, y1))
    assert cp.q._group._cur is not None  # (served through the copy's own group)
    del junk


@pytest.mark.parametrize("r,members,enabled", [(128, 3, False), (128, 2, True), (80, 3, True)])
def test_shared_activation_respects_the_side_gemm_rank_limit(lq, r, members, enabled):
    """The concatenated side GEMM takes a padded rank of at most 256: q/k/v of OPT-6.7B at rank 128 (384) cannot share one
    concatenation - the group is then disabled at construction and its members run one by one (it used to fail at the first
    forward); two such members (256) and three rank-80 members (240) do share."""
    from bench import MXINT_Q, make_case
    from lqer_amd.linear import SharedActivation

    M, K, N = 130, 512, 256
    mods = []
    for i in range(members):
        x, W, A, B = make_case(M, K, N, r, seed=70 + i)
        m = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
        m.load_state_dict({"weight": W, "A": A, "B": B})
        mods.append(m.to(DEV).half())
    xd = x.half().to(DEV)
    alone = [m(xd).clone() for m in mods]
    grp = SharedActivation(mods)
    assert grp.enabled == enabled
    assert all((getattr(m, "_group", None) is grp) == enabled for m in mods)
    with torch.no_grad():
        for m, a in zip(mods, alone):
            y = m(xd)
            assert float((y.float() - a.float()).norm() / a.float().norm()) <= 2e-3


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_forward_on_a_non_current_device(lq):
    from bench import MXINT_Q, make_case

    M, K, N, r = 33, 256, 192, 32
    x, W, A, B = make_case(M, K, N, r, seed=8)
    mod = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    ref = O.lqer_linear_forward(x, W, None, A, B, MXINT_Q)
    torch.cuda.set_device(0)
    mod = mod.to("cuda:1")
    y = mod(x.to("cuda:1"))
    assert torch.cuda.current_device() == 0 and y.device.index == 1
    assert (y.cpu() - ref).norm() / ref.norm() <= 1e-5

"""Module-swap helper: structure on CPU, end-to-end logits parity on the GPU."""
import pytest
import torch
import torch.nn as nn

from bench import A16_Q, MXINT_Q, OPT_Q


def _tiny_llama():
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, vocab_size=320, max_position_embeddings=128)
    return LlamaForCausalLM(cfg).eval()


def _tiny_opt():
    from transformers import OPTConfig, OPTForCausalLM

    torch.manual_seed(0)
    cfg = OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=4, vocab_size=200,
                    max_position_embeddings=64, word_embed_proj_dim=128)
    return OPTForCausalLM(cfg).eval()


def _ab_dict(model, rank, seed=1):
    from lqer_amd import LinearFlexibleLqer
    from oracle import lqer_oracle as O

    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, m in model.named_modules():
        if isinstance(m, LinearFlexibleLqer):
            A = O.mxint_quantize(0.02 * torch.randn(m.in_features, rank, generator=g), width=8, block_size=[16, 1], skip_first_dim=False)
            B = O.mxint_quantize(0.02 * torch.randn(rank, m.out_features, generator=g), width=8, block_size=[16, 1], skip_first_dim=False)
            out[f"{name}.A"], out[f"{name}.B"] = A, B
    return out


def test_swap_structure_cpu():
    from lqer_amd import LinearFlexibleLqer
    from lqer_amd.models import load_low_rank_dict, quantize_model

    model = _tiny_llama()
    ref_sd = {k: v.clone() for k, v in model.state_dict().items()}
    qc = {"linear": MXINT_Q, "model_layer_1": {"self_attn": {"q_proj": dict(MXINT_Q, name="flexible")}}}
    quantize_model(model, qc, {"linear": {"rank": 16}})
    l0, l1 = model.model.layers[0], model.model.layers[1]
    assert isinstance(l0.self_attn.q_proj, LinearFlexibleLqer) and isinstance(l0.mlp.down_proj, LinearFlexibleLqer)
    assert type(l1.self_attn.q_proj).__name__ == "LinearFlexible"  # per-layer override
    assert isinstance(model.lm_head, nn.Linear) and not isinstance(model.lm_head, LinearFlexibleLqer)
    sd = model.state_dict()
    for k, v in ref_sd.items():  # original weights carried over, same keys
        assert torch.equal(sd[k], v), k
    assert sd["model.layers.0.self_attn.q_proj.A"].shape == (256, 16) and sd["model.layers.0.mlp.gate_proj.B"].shape == (16, 512)
    ab = _ab_dict(model, 16)
    assert len(ab) == 2 * (14 - 1)
    assert load_low_rank_dict(model, ab) == []
    assert torch.equal(model.state_dict()["model.layers.1.mlp.up_proj.A"], ab["model.layers.1.mlp.up_proj.A"])
    assert load_low_rank_dict(model, {"model.layers.9.mlp.up_proj.A": torch.zeros(1)}) == ["model.layers.9.mlp.up_proj.A"]
    opt = quantize_model(_tiny_opt(), {"linear": OPT_Q}, {"linear": {"rank": 16}})
    d0 = opt.model.decoder.layers[0]
    assert isinstance(d0.fc1, LinearFlexibleLqer) and isinstance(d0.self_attn.out_proj, LinearFlexibleLqer)
    assert "model.decoder.layers.0.fc2.A" in opt.state_dict()
    with pytest.raises(ValueError):
        quantize_model(nn.Sequential(nn.Linear(4, 4)), {"linear": MXINT_Q}, None)


class _OracleLinear(nn.Module):
    """CPU stand-in with the oracle's forward, for the end-to-end comparison only."""

    def __init__(self, src, q_config):
        super().__init__()
        self.w, self.b, self.A, self.B, self.qc = src.weight.detach().float().cpu(), None if src.bias is None else src.bias.detach().float().cpu(), src.A.detach().float().cpu(), src.B.detach().float().cpu(), q_config

    def forward(self, x):
        from oracle import lqer_oracle as O

        return O.lqer_linear_forward(x, self.w, self.b, self.A, self.B, self.qc)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["llama", "opt", "llama-a16"])
def test_end_to_end_logits_vs_oracle(family):
    from lqer_amd import LinearFlexibleLqer
    from lqer_amd.models import load_low_rank_dict, quantize_model

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # llama-a16: the reference's INT template (llama-7b-int.toml) - pass-through activations, W4 in blocks of 128
    qc = {"llama": MXINT_Q, "opt": OPT_Q, "llama-a16": A16_Q}[family]
    model = quantize_model(_tiny_opt() if family == "opt" else _tiny_llama(), {"linear": qc}, {"linear": {"rank": 16}})
    load_low_rank_dict(model, _ab_dict(model, 16))
    # CPU twin whose projections run the oracle, built from the same parameters before they are quantized in place
    import copy

    twin = copy.deepcopy(model)
    for name, m in list(twin.named_modules()):
        if isinstance(m, LinearFlexibleLqer):
            parent = twin.get_submodule(name.rsplit(".", 1)[0])
            setattr(parent, name.rsplit(".", 1)[1], _OracleLinear(m, qc))
    ids = torch.randint(0, 200, (2, 24), generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref = twin(input_ids=ids).logits
        got = model.to("cuda:0")(input_ids=ids.to("cuda:0")).logits.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).norm() / ref.norm()
    assert err <= 1e-4, float(err)  # fp32 model: only accumulation order differs, layer after layer
    if family == "llama-a16":
        # the same model in fp16: every projection takes the fp16 MFMA route; the bf16-limb route computes the same
        # exact products in another summation order, so the logits agree to fp16 rounding noise
        outs = []
        for native in (True, False):
            m16 = copy.deepcopy(model).half()
            for m in m16.modules():
                if isinstance(m, LinearFlexibleLqer):
                    m.a16_native = native
            with torch.no_grad():
                outs.append(m16(input_ids=ids.to("cuda:0")).logits.float().cpu())
            assert all(m._x_f16 == native for m in m16.modules() if isinstance(m, LinearFlexibleLqer))
        assert torch.isfinite(outs[0]).all()
        assert (outs[0] - outs[1]).norm() / outs[1].norm() <= 3e-3
        assert (outs[0] - got).norm() / got.norm() <= 5e-2  # (fp16 model against the fp32 one: norms, softmax in fp16)


def test_packed_checkpoint_header_mismatch_cpu(tmp_path):
    """load_packed refuses files of another format / for other modules (no GPU needed: failure paths only)."""
    from safetensors.torch import save_file

    from lqer_amd.checkpoint import load_packed
    from lqer_amd.models import quantize_model

    model = _tiny_llama()
    quantize_model(model, {"linear": MXINT_Q}, {"linear": {"rank": 16}})
    bad = tmp_path / "bad.safetensors"
    save_file({"x": torch.zeros(1)}, str(bad), metadata={"format": "something else"})
    with pytest.raises(RuntimeError, match="not a lqer_amd.packed file"):
        load_packed(model, str(bad), device="cpu")
    empty = tmp_path / "empty.safetensors"
    save_file({"x": torch.zeros(1)}, str(empty), metadata={"format": "lqer_amd.packed", "version": "1"})
    with pytest.raises(RuntimeError, match="no packed images"):
        load_packed(model, str(empty), device="cpu")


@pytest.mark.gpu
def test_packed_checkpoint_round_trip_gpu(tmp_path):
    """save_packed -> fresh model with untouched (zero / random) dense weights -> load_packed: same logits bit for
    bit, file about 0.3x the dense fp16 size of the quantized Linears, dense operands not consulted."""
    import os

    from lqer_amd import LinearFlexibleLqer
    from lqer_amd.checkpoint import load_packed, save_packed
    from lqer_amd.models import load_low_rank_dict, quantize_model

    dev = torch.device("cuda:0")
    qc, lc = {"linear": OPT_Q}, {"linear": {"rank": 16}}
    model = _tiny_opt()
    quantize_model(model, qc, lc)
    load_low_rank_dict(model, _ab_dict(model, 16))
    model = model.to(dev)
    ids = torch.randint(0, 200, (2, 24), generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        ref = model(ids).logits.float().cpu()
    path = str(tmp_path / "tiny_opt.packed.safetensors")
    n = save_packed(model, path)
    assert n == sum(isinstance(m, LinearFlexibleLqer) for m in model.modules()) == 12

    fresh = _tiny_opt()
    with torch.no_grad():
        for p in fresh.parameters():
            p.add_(1.0)  # a different model: everything must come from the file
    quantize_model(fresh, qc, lc)
    fresh = fresh.to(dev)
    missing = load_packed(fresh, path, device=dev)
    assert missing == [], missing
    with torch.no_grad():
        for m in fresh.modules():
            if isinstance(m, LinearFlexibleLqer):
                m.weight.zero_()  # the dense operands are not used any more
                m.A.fill_(float("nan"))
        got = fresh(ids).logits.float().cpu()
    assert torch.equal(got, ref)
    dense = sum(m.weight.numel() * 2 + m.A.numel() * 2 + m.B.numel() * 2 for m in model.modules() if isinstance(m, LinearFlexibleLqer))
    packed = sum(v.numel() * v.element_size() for k, v in __import__("lqer_amd.checkpoint", fromlist=["x"]).packed_state_dict(model).items()
                 if k.rsplit(".", 1)[-1] in ("w", "a_t", "b_t", "bias_q", "header"))
    assert packed < dense, (packed, dense)  # 128-wide layers pay the 256-row padding; 0.29x at 4096 x 4096 (CPU test below)
    assert os.path.getsize(path) > 0
    # moving the module keeps the images; reloading dense weights drops them
    fresh.model.decoder.layers[0].fc1.to(dev)
    assert fresh.model.decoder.layers[0].fc1._packed_only and fresh.model.decoder.layers[0].fc1._packed is not None
    fresh.model.decoder.layers[0].fc1.load_state_dict(model.model.decoder.layers[0].fc1.state_dict())
    assert not fresh.model.decoder.layers[0].fc1._packed_only and fresh.model.decoder.layers[0].fc1._packed is None


def test_packed_sizes_cpu():
    """Bytes of the packed images of a 4096 x 4096 rank-32 Linear vs its fp16 operands (the checkpoint's content)."""
    import ctypes as C

    import lqer_amd
    from lqer_amd import _lib, ops

    mod = lqer_amd.LinearFlexibleLqer(4096, 4096, bias=False, q_config=MXINT_Q, l_config={"rank": 32})
    sz = ops.linear_sizes(mod._desc(), 1)
    assert sz.w_packed == 4096 * 4096 // 2 + 4096 * 4096 // 16  # 4-bit codes + one exponent byte per 16 weights
    one_limb = 32 * 4096 * 2
    assert sz.a_t == 3 * one_limb and sz.b_t == 3 * one_limb  # the file keeps only the limbs in use (1 for MXINT8 A, B)
    dense = (4096 * 4096 + 2 * 4096 * 32) * 2
    assert (sz.w_packed + 2 * one_limb) / dense < 0.30


@pytest.mark.gpu
def test_shared_activation_groups_gpu():
    """q/k/v and gate/up quantize their common input once (SharedActivation): same logits as the ungrouped model up to
    the summation order of the side product, the shared images are really used, and a modified or different input is
    never served from them."""
    from lqer_amd import LinearFlexibleLqer
    from lqer_amd.models import load_low_rank_dict, quantize_model

    dev = torch.device("cuda:0")
    qc, lc = {"linear": MXINT_Q}, {"linear": {"rank": 16}}
    shared = quantize_model(_tiny_llama(), qc, lc, share_inputs=True)
    plain = quantize_model(_tiny_llama(), qc, lc, share_inputs=False)
    ab = _ab_dict(shared, 16)
    load_low_rank_dict(shared, ab)
    load_low_rank_dict(plain, ab)
    shared, plain = shared.to(dev), plain.to(dev)
    l0 = shared.model.layers[0]
    grp = l0.self_attn.q_proj._group
    assert grp is not None and grp.enabled and grp is l0.self_attn.k_proj._group is l0.self_attn.v_proj._group
    assert l0.mlp.gate_proj._group is l0.mlp.up_proj._group and l0.mlp.down_proj._group is None
    assert plain.model.layers[0].self_attn.q_proj._group is None
    ids = torch.randint(0, 320, (2, 24), generator=torch.Generator().manual_seed(7)).to(dev)
    with torch.no_grad():
        a = shared(input_ids=ids).logits.float().cpu()
        b = plain(input_ids=ids).logits.float().cpu()
    assert (a - b).norm() / b.norm() <= 2e-5
    # member by member: the same tensor object is served from the shared images, anything else is not
    q, k = l0.self_attn.q_proj, l0.self_attn.k_proj
    pq, pk = plain.model.layers[0].self_attn.q_proj, plain.model.layers[0].self_attn.k_proj
    x = torch.randn(2, 9, 256, device=dev)
    with torch.no_grad():
        yq = q(x)
        made_from = grp._x
        yk = k(x)
        assert grp._x is made_from is x  # k_proj reused q_proj's activation image
        assert (yq - pq(x)).norm() / yq.norm() <= 2e-5 and (yk - pk(x)).norm() / yk.norm() <= 2e-5
        x.mul_(2.0)  # in place: the version counter moves, the images are rebuilt
        yk2 = k(x)
        assert (yk2 - pk(x)).norm() / yk2.norm() <= 2e-5 and not torch.equal(yk2, yk)
        x2 = x.clone()  # an equal tensor that is a different object is quantized afresh (and gives the same result)
        yk3 = k(x2)
        assert grp._x is x2 and torch.equal(yk3, yk2)


@pytest.mark.gpu
def test_shared_activation_groups_int_config_gpu():
    """The INT configuration (one activation block per token, one A_out block per row) also shares q/k/v and gate/up:
    the group re-quantizes x A in blocks of one member's rank, i.e. exactly each member's own row block."""
    from bench import INT_Q
    from lqer_amd.models import load_low_rank_dict, quantize_model

    dev = torch.device("cuda:0")
    qc, lc = {"linear": INT_Q}, {"linear": {"rank": 16}}
    shared = quantize_model(_tiny_llama(), qc, lc, share_inputs=True)
    plain = quantize_model(_tiny_llama(), qc, lc, share_inputs=False)
    ab = _ab_dict(shared, 16)
    load_low_rank_dict(shared, ab)
    load_low_rank_dict(plain, ab)
    shared, plain = shared.to(dev), plain.to(dev)
    grp = shared.model.layers[0].self_attn.q_proj._group
    assert grp is not None and grp.enabled and grp._aout_block == 16
    ids = torch.randint(0, 320, (2, 24), generator=torch.Generator().manual_seed(7)).to(dev)
    with torch.no_grad():
        a = shared(input_ids=ids).logits.float().cpu()
        b = plain(input_ids=ids).logits.float().cpu()
    assert grp._cur is not None and len(grp._served) == len(grp.members)  # the shared images were made and served every member
    assert (a - b).norm() / b.norm() <= 5e-5
    # rank 48 is not a power of two and rank 24 is not a multiple of 16: one block per row cannot be cut per member
    for r in (48, 24):
        odd = quantize_model(_tiny_llama(), qc, {"linear": {"rank": r}}, share_inputs=True)
        assert odd.model.layers[0].self_attn.q_proj._group is None
    r64 = quantize_model(_tiny_llama(), qc, {"linear": {"rank": 64}}, share_inputs=True)
    g64 = r64.model.layers[0].mlp.gate_proj._group
    assert g64 is not None and g64._aout_block == 64
    load_low_rank_dict(r64, _ab_dict(r64, 64))
    p64 = quantize_model(_tiny_llama(), qc, {"linear": {"rank": 64}}, share_inputs=False)
    load_low_rank_dict(p64, _ab_dict(p64, 64))
    with torch.no_grad():
        a = r64.to(dev)(input_ids=ids).logits.float().cpu()
        b = p64.to(dev)(input_ids=ids).logits.float().cpu()
    assert (a - b).norm() / b.norm() <= 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["llama", "opt"])
def test_quantized_attention_end_to_end_gpu(family):
    """enable_quantized_attention: both attention products of every layer go through matmul_flexible (reference
    llama_decoder.py:259-297; opt_decoder.py:125,190 with the "bmm" configuration).  Logits against a CPU twin whose
    projections AND attention products run the oracle."""
    import copy
    import json
    import os

    from transformers import AttentionInterface
    from transformers.masking_utils import AttentionMaskInterface, eager_mask

    from lqer_amd import LinearFlexibleLqer
    from lqer_amd import attention as A
    from lqer_amd.models import load_low_rank_dict, quantize_model
    from oracle import lqer_oracle as O

    mm_cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "matmul_config.json")))
    lin_q = MXINT_Q if family == "llama" else OPT_Q
    qc = {"linear": lin_q, ("matmul" if family == "llama" else "bmm"): mm_cfg}
    model = quantize_model(_tiny_llama() if family == "llama" else _tiny_opt(), qc, {"linear": {"rank": 16}})
    load_low_rank_dict(model, _ab_dict(model, 16))
    twin = copy.deepcopy(model)
    for name, m in list(twin.named_modules()):
        if isinstance(m, LinearFlexibleLqer):
            parent = twin.get_submodule(name.rsplit(".", 1)[0])
            setattr(parent, name.rsplit(".", 1)[1], _OracleLinear(m, lin_q))

    def oracle_attention(module, query, key, value, attention_mask, scaling, dropout=0.0, **kwargs):
        k = A._repeat_kv(key, getattr(module, "num_key_value_groups", 1))
        v = A._repeat_kv(value, getattr(module, "num_key_value_groups", 1))
        b, h, s, d = query.shape
        w = O.matmul_flexible(query.reshape(b * h, s, d), k.reshape(b * h, -1, d).transpose(1, 2), mm_cfg).reshape(b, h, s, -1) * scaling
        if attention_mask is not None:
            w = w + attention_mask
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(query.dtype)
        out = O.matmul_flexible(w.reshape(b * h, s, -1), v.reshape(b * h, -1, d), mm_cfg).reshape(b, h, s, d)
        return out.transpose(1, 2).contiguous(), w

    AttentionInterface.register("lqer_oracle_eager", oracle_attention)
    AttentionMaskInterface.register("lqer_oracle_eager", eager_mask)
    twin.set_attn_implementation("lqer_oracle_eager")
    A.enable_quantized_attention(model, qc)
    assert model.config._attn_implementation == A.IMPLEMENTATION
    ids = torch.randint(0, 200, (2, 20), generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        ref = twin(input_ids=ids).logits
        plain = copy.deepcopy(twin)
        plain.set_attn_implementation("eager")
        ref_plain = plain(input_ids=ids).logits
        got = model.to("cuda:0")(input_ids=ids.to("cuda:0")).logits.float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).norm() / ref.norm() <= 1e-4
    assert (ref - ref_plain).norm() / ref.norm() > 1e-4  # the quantized attention products do change the logits


def test_swap_structure_mistral_cpu():
    """The reference also ships a Mistral decoder (models/mistral_decoder.py): same layer layout as Llama, so the
    by-name swap and the shared-input groups apply unchanged."""
    from transformers import MistralConfig, MistralForCausalLM

    from lqer_amd import LinearFlexibleLqer
    from lqer_amd.models import quantize_model

    torch.manual_seed(0)
    cfg = MistralConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, vocab_size=300, max_position_embeddings=64, sliding_window=32)
    model = MistralForCausalLM(cfg).eval()
    keys = set(model.state_dict().keys())
    quantize_model(model, {"linear": MXINT_Q}, {"linear": {"rank": 16}})
    l0 = model.model.layers[0]
    assert all(isinstance(getattr(l0.self_attn, n), LinearFlexibleLqer) for n in ("q_proj", "k_proj", "v_proj", "o_proj"))
    assert all(isinstance(getattr(l0.mlp, n), LinearFlexibleLqer) for n in ("gate_proj", "up_proj", "down_proj"))
    assert l0.self_attn.k_proj.out_features == 64  # grouped-query attention: k/v are narrower than q
    grp = l0.self_attn.q_proj._group
    assert grp is not None and grp.enabled and len(grp.members) == 3  # different widths share one activation all the same
    assert keys <= set(model.state_dict().keys())
    assert "model.layers.1.mlp.down_proj.A" in model.state_dict()


@pytest.mark.gpu
def test_mistral_gqa_shared_groups_gpu():
    """Grouped-query attention: q (128 wide) and k/v (64 wide) share one activation image; logits equal the
    one-by-one model's up to the side product's summation order."""
    from transformers import MistralConfig, MistralForCausalLM

    from lqer_amd.models import load_low_rank_dict, quantize_model

    def build(share):
        torch.manual_seed(0)
        cfg = MistralConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                            num_key_value_heads=2, vocab_size=300, max_position_embeddings=64, sliding_window=32)
        m = quantize_model(MistralForCausalLM(cfg).eval(), {"linear": MXINT_Q}, {"linear": {"rank": 16}}, share_inputs=share)
        load_low_rank_dict(m, _ab_dict(m, 16))
        return m.to("cuda:0")

    a, b = build(True), build(False)
    ids = torch.randint(0, 300, (2, 16), generator=torch.Generator().manual_seed(2)).to("cuda:0")
    with torch.no_grad():
        la, lb = a(input_ids=ids).logits.float().cpu(), b(input_ids=ids).logits.float().cpu()
    assert torch.isfinite(la).all() and (la - lb).norm() / lb.norm() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tokens", [1, 6, 40])
def test_whole_model_forward_replayed_from_one_graph_gpu(tokens):
    """A serving loop captures the model's token step once and replays it (lqer_amd.graph.GraphedCallable): every quantized
    Linear (shared q/k/v and gate/up inputs), both quantized attention products and the rest of the HF model inside ONE
    hipGraph.  1 and 6 tokens take the one-launch decode kernel (capturable since round 3: its hand-off tag carries the
    launch's dispatch id), 40 tokens the two-launch route.  Replays with new token ids equal the eager logits bit for bit."""
    import json
    import os

    from lqer_amd import attention as A
    from lqer_amd.graph import GraphedCallable
    from lqer_amd.models import load_low_rank_dict, quantize_model

    mm_cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "matmul_config.json")))
    qc = {"linear": MXINT_Q, "matmul": mm_cfg}
    model = quantize_model(_tiny_llama(), qc, {"linear": {"rank": 16}}, share_inputs=True)
    load_low_rank_dict(model, _ab_dict(model, 16))
    A.enable_quantized_attention(model, qc)
    model = model.to("cuda:0").half()
    g = torch.Generator().manual_seed(5)
    ids = [torch.randint(0, 320, (1, tokens), generator=g).to("cuda:0") for _ in range(4)]
    # (the causal mask as a ready 4-D tensor: transformers builds its own with a host-to-device scalar copy, which a capture refuses)
    mask = torch.full((1, 1, tokens, tokens), float("-inf"), dtype=torch.float16, device="cuda:0").triu(1)
    run = lambda t: model(input_ids=t, attention_mask=mask, use_cache=False).logits
    with torch.no_grad():
        eager = [run(i).clone() for i in ids]
        static_ids = ids[0].clone()
        step = GraphedCallable(run, static_ids, warmup=2)
        for i, want in zip(ids, eager):
            got = step(i).clone()
            torch.cuda.synchronize()
            assert torch.isfinite(got).all()
            assert torch.equal(got, want)
    assert not torch.equal(eager[0], eager[1])

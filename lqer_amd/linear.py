"""Drop-in mirror of the reference's quantized Linear modules, running on the HIP library.

Mirrors the reference file src/lqer/quantize/quantized_layers/linear.py:
  _LinearBase        :12-85    constructor signature, q_config / l_config attributes, repr
  LinearFlexible     :88-109   x / w / b quantizers, forward = F.linear(Q_x(x), Q_w(W), Q_b(b))
  LinearFlexibleLqer :112-166  + A [in, rank], B [rank, out], A_out / B_out quantizers defaulting
                               to the x quantizer's config (:115-124)
and the registry of quantized_layers/__init__.py:3-16 (`get_quantized_layer_cls`).

Same parameter names and shapes (weight, bias, A, B), so a stock HF checkpoint and the reference's
`low_rank_dict.pt` load unchanged with `load_state_dict(..., strict=False)` (reference
models/llama_decoder.py:507, runners.py:220-222).  The packed 4-bit weight image, the transposed
bf16 limb images of A and B and the quantized bias are derived, non-persistent buffers built on the
first forward - the counterpart of the reference's in-place first-forward quantization
(linear.py:149-153) - and rebuilt after load_state_dict / .to().

Only the PTQ inference branch exists here (is_ptq=True in every template config); the training
branch (linear.py:158-166) raises.  There is no CPU path: a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from copy import deepcopy

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import LinearDesc, QFmt, check

_SIG_BITS = {torch.float32: 24, torch.float16: 11, torch.bfloat16: 8}  # significand bits incl. the hidden one
_NDEV = []


def _device_count() -> int:
    """torch.cuda.device_count(), asked once (it does not initialise the GPU): a one-GPU process never needs the device guard."""
    if not _NDEV:
        _NDEV.append(torch.cuda.device_count())
    return _NDEV[0]


class _LinearBase(nn.Linear):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 q_config: dict = None, l_config: dict = None) -> None:
        super().__init__(in_features, out_features, bias, device, dtype)
        self.q_config = q_config
        self.l_config = l_config
        self.is_ptq = q_config.get("is_ptq", False)
        self.w_is_quantized = False if self.is_ptq else None
        self._fmt = {}
        self._packed = None
        self._packed_only = False  # True: images came from a packed checkpoint, the dense parameters are not used
        # The weight and the bias are quantized ONCE, like the reference (linear.py:149-153: w_is_quantized stays True;
        # block_fp quantization is not idempotent - a block maximum that rounds down to a power of two lowers the
        # exponent of a second pass).  Their images outlive invalidate_packed() until the dense parameter is reloaded.
        self._w_single = None      # single-copy packed weight image of the current `weight`
        self._bias_q = None        # fp32 [Np] b_quantizer(bias) of the current `bias`
        self._w_ver = None         # (weight._version, bias._version) when the images were built: an in-place write to the
                                   # parameters (`with torch.no_grad(): mod.weight.copy_(W2)`) is seen by the next forward
        self._group = None         # SharedActivation of Linears fed by the same tensor (models.quantize_model)
        self._fw_cache = {}        # token count -> (descriptor, workspace bytes)
        self._x_f16 = False        # pass-through fp16 activations on the fp16 MFMA route (decided when the images are built)
        self.a16_native = True     # False: keep pass-through fp16 activations on the bf16-limb route
        self._x_i8 = False         # per-token 8-bit activations on the int8 MFMA route (decided when the images are built)
        self.i8_a_f16 = True       # int8 route: fp16 A as one fp16 image for the side GEMM (False: the bf16 limb pair)
        self.a8_native = True      # False: keep them on the bf16 route
        self.a16_fused = True      # block-16 MXINT activations: build the image the one-launch activation kernel reads (a_limbs = -2)
        self.tuning = 0            # lqer_linear_desc_t.tuning (_lib.TUNE_*): per-call kernel-variant knobs of tests, same bits
        self._setup_quantizers(q_config)
        self._setup_lqer(l_config)
        # activation quantizers whose blocks can span token rows ([R, L] tiles, skip_first_dim = false): the tile route below
        self._tiles = any(getattr(self._fmt.get(r), "act_tiles", None) is not None for r in ("x", "A_out", "B_out"))
        # ... on EVERY tensor, unless the only such formats are the quantizer's default lone [L] with skip_first_dim = true: per-row blocks
        # of L on a 2-D tensor - the fused kernels' layout -, [1, T, L] tiles on a 3-D one.  Such a module still has packed images.
        self._tiles_only = any(not getattr(self._fmt.get(r), "act_tiles", (0, 0, True, True))[3] for r in ("x", "A_out", "B_out"))
        self.__dict__["_inner"] = None  # (the tile route's main-product Linear: not a registered submodule - it shares this module's parameters)

    # -- configuration -------------------------------------------------------------------------
    def _setup_quantizers(self, q_config: dict):
        raise NotImplementedError

    def _setup_lqer(self, l_config: dict):
        raise NotImplementedError

    @property
    def rank(self) -> int:
        return 0

    def _desc(self, plain: bool = False) -> LinearDesc:
        """The C descriptor (plain = True: the bf16-route descriptor of a Linear whose images also serve the int8 route).  Pass-through x / A_out formats (the *-int.toml templates) carry the number of significand
        bits the HIP path must preserve: those of the module's dtype for x (8 bf16, 11 fp16, 24 fp32 = 1, 2, 3 bf16
        limbs), 16 for x A under a 16-bit dtype (the reference keeps 11 or 8 there), 24 under fp32."""
        f = self._fmt
        none = QFmt(_lib.Q_PASSTHROUGH, 0, 0, 8, 127)
        dt = self.weight.dtype
        if dt not in _SIG_BITS:
            raise TypeError(f"lqer_amd: unsupported dtype {dt} (float32 / float16 / bfloat16)")

        def eff(role, bits):
            q = f.get(role, none)
            if q.kind == _lib.Q_PASSTHROUGH and role == "x" and self._x_f16:
                return QFmt(_lib.Q_PASSTHROUGH_F16, bits, q.block, q.exp_width, q.exp_bias)
            if q.kind == _lib.Q_MXINT and role == "x" and self._x_i8 and not plain:
                return QFmt(_lib.Q_MXINT_I8, q.width, q.block, q.exp_width, q.exp_bias)
            return QFmt(q.kind, bits, q.block, q.exp_width, q.exp_bias) if q.kind == _lib.Q_PASSTHROUGH else q

        return LinearDesc(self.in_features, self.out_features, self.rank, int(self.bias is not None),
                          eff("x", _SIG_BITS[dt]), f["w"], f.get("b", none), eff("A_out", 24 if dt == torch.float32 else 16),
                          f.get("B_out", none), int(getattr(self, "tuning", 0)))

    def _limbs(self):
        """(activation limbs, x A limbs) of the packed images - 1, 1 unless x / A_out are pass-through."""
        return ops.desc_limbs(self._desc())

    def _replicate(self, p: dict) -> dict:
        """Single-copy packed images -> the images the kernels read (no-op for block_fp x / A_out).  Pass-through fp16
        activations take the fp16 MFMA route when the weight block scales and A are exact in fp16 (the activation image
        is then the fp16 tensor itself, A one fp16 image, one copy of W); otherwise, and for bf16 / fp32 tensors, the
        activation is split into bf16 limbs and the images are repeated once per limb."""
        self._x_f16 = False
        self._x_i8 = False
        self._fw_cache = {}
        fx, fw, K = self._fmt["x"], self._fmt["w"], self.in_features
        if (self.a8_native and fx.kind == _lib.Q_MXINT and fw.kind == _lib.Q_MXINT and fx.width <= 8 and (fx.block <= 0 or fx.block >= K)
                and K >= 128 and (fw.block <= 0 or fw.block >= K or fw.block % 128 == 0)):
            # one activation exponent per token, weight blocks of 128 k or more (the W4A8 INT configurations): integer
            # accumulation is exact - the int8 MFMA route, if every weight row's sums provably stay inside i32
            ok, w2 = ops.i8_prepare(p["w"], self.out_features, K, fw)
            if ok:
                self._x_i8 = True
                p = dict(p)
                p["w"] = w2
                # unquantized fp16 A (two bf16 limbs): the side GEMM of the int8 route takes it as ONE fp16 image on the fp16
                # MFMA (int8 mantissas are exact in fp16) - half the A^T bytes and MFMAs of the limb pair.  Round 6: also an A of ONE
                # bf16 limb that fp16 holds exactly (bf16 modules, 8-bit block_fp A): the image carries the fragment-major copy that the
                # one-launch activation kernel reads (act8_fused.hip)
                if self.rank > 0 and int(p.get("a_limbs", 0)) in (1, 2) and self.i8_a_f16:
                    ok16, a16 = ops.a_f16_image(w2, self.out_features, K, p["a_t"], int(p["a_limbs"]), self.rank)
                    if ok16:
                        p["a_t_f16"] = a16
        fa_ = self._fmt.get("A_out")
        if (self.a16_fused and not self._x_i8 and self.rank > 0 and int(p.get("a_limbs", 0)) == 1 and fx.kind == _lib.Q_MXINT and fx.block == 16
                and fx.width <= 9 and fa_ is not None and fa_.kind == _lib.Q_MXINT and fa_.width <= 9
                and _lib.lib().lqer_padded_r(self.rank) in (16, 32, 64)):
            # block-16 MXINT activations with an A of one bf16 limb (the llama-7b.toml / opt-6.7b.toml templates): the bf16 image of A^T with
            # its fragment-major copy - quantizer + x A + A_out then run as ONE launch at prefill sizes (act16_fused.hip, a_limbs = -2)
            p = dict(p)
            p["a_t_b16"] = ops.a_b16_image(p["a_t"], K, self.rank)
        if (self._fmt["x"].kind == _lib.Q_PASSTHROUGH and self.weight.dtype == torch.float16 and self.a16_native
                and fw.kind == _lib.Q_MXINT and fw.width <= 4):
            # (integer weights - two's-complement nibbles - have no fp16 main loop, weights of 5..8 bits travel as three 4-bit limbs
            # over a repeated activation image: the limb route)
            ok, a16 = ops.f16_prepare(p["w"], self.out_features, self.in_features, p.get("a_t"), int(p.get("a_limbs", 0)), self.rank)
            if ok:
                self._x_f16 = True
                p = dict(p)
                if self.rank > 0:
                    p["a_t_limbs"], p["a_t"], p["a_limbs"], p["a_limbs_orig"] = p["a_t"], a16, 1, p["a_limbs"]
        xl, al = self._limbs()
        if xl == 1 and al == 1:
            return p
        L = _lib.lib()
        Kp, Np = L.lqer_padded_k(self.in_features), L.lqer_padded_n(self.out_features)
        q = dict(p)
        q["w"] = ops.replicate_rows(p["w"], (Np // 16) * ops.w_limbs(fw), (Kp // 64) * 576, xl)  # (every weight limb once per activation limb)
        if self.rank > 0:
            rp = L.lqer_padded_r(self.rank)
            q["a_t"] = ops.replicate_rows(p["a_t"], 3 * rp, Kp * 2, xl)
            q["b_t"] = ops.replicate_rows(p["b_t"], 3 * Np, rp * 2, al)
        return q

    def _single_copy(self, name: str) -> torch.Tensor:
        """Copy 0 of a repeated image, as flat bytes (inverse of _replicate)."""
        xl, al = self._limbs()
        L = _lib.lib()
        Kp, Np = L.lqer_padded_k(self.in_features), L.lqer_padded_n(self.out_features)
        rp = L.lqer_padded_r(self.rank) if self.rank > 0 else 0
        rows, rb, c = {"w": ((Np // 16) * ops.w_limbs(self._fmt["w"]), (Kp // 64) * 576, xl), "a_t": (3 * rp, Kp * 2, xl),
                       "b_t": (3 * Np, rp * 2, al)}[name]
        if name == "a_t" and "a_t_limbs" in self._packed:  # fp16 route: the limb image is kept next to the fp16 one
            return self._packed["a_t_limbs"].reshape(-1).view(torch.uint8)
        flat = self._packed[name].reshape(-1).view(torch.uint8)
        if name == "w" and self._x_i8:  # the int8 route's second image lies behind the sign-magnitude one
            return flat[: rows * rb]
        return flat if c == 1 else flat.view(rows, c, rb)[:, 0].contiguous().reshape(-1)

    def __getstate__(self):
        # copy.deepcopy / pickle: the per-token-count launch cache holds raw device pointers of THIS module's images
        st = self.__dict__.copy()
        st["_fw_cache"] = {}
        # the copy's parameters are new tensors with version counters of their own: its first forward adopts them instead of
        # mistaking the copy for an in-place write (which would quantize the already quantized weight a second time)
        st["_w_ver"] = None
        return st

    # -- derived buffers -------------------------------------------------------------------------
    def invalidate_packed(self, *, weight_changed: bool, bias_changed: bool = False) -> None:
        """Drop the derived images; the next forward rebuilds them.  `weight_changed` has no default on purpose: the caller
        says whether `weight` now holds NEW unquantized values (True: it is quantized and packed again, as after
        load_state_dict, which passes it by itself) or the values this module already quantized (False: the
        quantized-once image is kept - .to() / .half() / A, B reloaded; block_fp quantization is not idempotent, reference
        linear.py:149-153 runs it once).  Writes through `weight.data` are invisible to autograd's version counter, so
        after `mod.weight.data.copy_(W2)` call `invalidate_packed(weight_changed=True)`; in-place writes to the parameter
        itself (`mod.weight.copy_(W2)` under no_grad) are noticed by the next forward without any call."""
        inner = self.__dict__.get("_inner")
        if inner is not None:  # (the tile route's twin holds the images of this module's weight / bias)
            inner.invalidate_packed(weight_changed=weight_changed, bias_changed=bias_changed)
        self._packed = None
        self._fw_cache = {}
        self._x_f16 = False
        self._x_i8 = False
        self._w_ver = None
        if weight_changed:
            self._w_single = None
            if self.is_ptq:
                self.w_is_quantized = False
        if bias_changed or weight_changed:  # (the reference quantizes both in its one first-forward step)
            self._bias_q = None
        if getattr(self, "_group", None) is not None:
            self._group.invalidate()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        has = lambda k: (prefix + k) in state_dict
        if any(has(k) for k in ("weight", "bias", "A", "B")):  # dense operands (re)loaded
            self._packed_only = False
            # A, B alone (runners.py:220-222 loads them after the first forward may have run): the weight is NOT
            # quantized a second time
            self.invalidate_packed(weight_changed=has("weight"), bias_changed=has("bias"))

    def _apply(self, fn, recurse=True):
        keep = getattr(self, "_packed_only", False) and self._packed is not None
        dt_before = self.weight.dtype
        singles = None
        if keep and (self._fmt["x"].kind == _lib.Q_PASSTHROUGH or self._limbs() != (1, 1)):
            # pass-through formats: the images depend on the dtype (limb copies, fp16 route) - go back to one copy first
            singles = {k: self._single_copy(k) for k in ("w", "a_t", "b_t") if k in self._packed}
            singles["a_limbs"] = int(self._packed.get("a_limbs_orig", self._packed.get("a_limbs", 0)))
        out = super()._apply(fn, recurse)
        if getattr(self, "_packed_only", False):
            # images loaded from a packed checkpoint are the only copy of the operands: keep them, follow the module
            # to its new device (dtype casts concern them only through the pass-through activation formats)
            dev = self.weight.device
            self._fw_cache = {}
            if dev.type == "cuda" and self._packed is not None:
                p = dict(self._packed)
                if singles is not None and self.weight.dtype != dt_before:
                    p.update(singles)
                    p.pop("a_t_limbs", None), p.pop("a_limbs_orig", None)
                    p = self._replicate({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in p.items()})
                self._packed = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in p.items()}
        else:
            # .to() / .half(): the A / B images follow the new dtype and device; the weight stays quantized once
            self.invalidate_packed(weight_changed=False)
        return out

    # -- packed checkpoint (lqer_amd/checkpoint.py) ------------------------------------------------
    PACKED_FORMAT_VERSION = 1

    def _fmt_digest(self) -> list:
        """Quantizer settings the packed images depend on: [kind, width, block, exp_width, exp_bias] per role."""
        out = []
        for role in ("x", "w", "b", "A_out", "B_out"):
            f = self._fmt.get(role)  # (as configured: the limb counts of pass-through formats follow the loading dtype)
            out += [-1] * 5 if f is None else [f.kind, f.width, f.block, f.exp_width, f.exp_bias]
        return out

    @torch.no_grad()
    def pack(self) -> None:
        """Build the packed images now (normally done by the first forward)."""
        if self._packed is None or self.w_is_quantized is False:
            self._pack()

    @torch.no_grad()
    def packed_state(self) -> dict:
        """The packed operands as flat tensors: 4-bit weight panels with their block exponents, the used bf16 limbs
        of A^T and B^T, the quantized bias, and an int32 header (version, K, N, rank, limb counts, formats)."""
        if self._tiles_only:
            raise NotImplementedError("packed checkpoints hold the fused path's images: a module whose activation blocks span token rows "
                                      "(the tile route) keeps its dense state_dict")
        self.pack()
        p = self._packed
        al, bl = int(p.get("a_limbs_orig", p.get("a_limbs", 0))), int(p.get("b_limbs", 0))
        out = {"w": self._single_copy("w")}  # (a Linear with pass-through activations holds one copy per limb: store one)
        if self.rank > 0:
            Kp, Np = _lib.lib().lqer_padded_k(self.in_features), _lib.lib().lqer_padded_n(self.out_features)
            rp = _lib.lib().lqer_padded_r(self.rank)
            out["a_t"] = self._single_copy("a_t")[: al * rp * Kp * 2].clone()
            out["b_t"] = self._single_copy("b_t")[: bl * Np * rp * 2].clone()
        if self.bias is not None:
            out["bias_q"] = p["bias"].reshape(-1).view(torch.uint8)
        hdr = [self.PACKED_FORMAT_VERSION, self.in_features, self.out_features, self.rank, al, bl,
               int(self.bias is not None)] + self._fmt_digest()
        out["header"] = torch.tensor(hdr, dtype=torch.int32)
        return out

    @torch.no_grad()
    def load_packed_state(self, state: dict, device: torch.device) -> None:
        """Attach images written by packed_state(); the dense weight / A / B parameters are not consulted afterwards
        (they may be left uninitialised).  Raises if the file does not match this module's shape or quantizers."""
        if self._tiles_only:
            raise NotImplementedError("packed checkpoints hold the fused path's images: not for a module on the tile route")
        hdr = [int(v) for v in state["header"].tolist()]
        want = [self.PACKED_FORMAT_VERSION, self.in_features, self.out_features, self.rank]
        if hdr[:4] != want:
            raise RuntimeError(f"packed checkpoint header {hdr[:4]} does not match module {want} (version, in, out, rank)")
        if hdr[6] != int(self.bias is not None) or hdr[7:] != self._fmt_digest():
            raise RuntimeError("packed checkpoint was written with different quantizer settings or bias layout")
        al, bl = hdr[4], hdr[5]
        L = _lib.lib()
        single = self._desc()  # single-copy sizes: those of block_fp activation formats
        single.x_fmt = QFmt(_lib.Q_MXINT, 8, 16, 8, 127)
        single.a_out_fmt = QFmt(_lib.Q_MXINT, 8, 16, 8, 127)
        sz = ops.linear_sizes(single, 1)
        dev = torch.device(device)

        def full(name, nbytes):
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            src = state[name].to(dev).reshape(-1).view(torch.uint8)
            if src.numel() > nbytes:
                raise RuntimeError(f"packed checkpoint tensor {name}: {src.numel()} B > {nbytes} B")
            buf[: src.numel()] = src
            return buf

        p = {"w": full("w", sz.w_packed)}
        if p["w"].numel() != state["w"].numel():
            raise RuntimeError("packed weight size mismatch")
        if self.rank > 0:
            p["a_t"], p["b_t"], p["a_limbs"], p["b_limbs"] = full("a_t", sz.a_t), full("b_t", sz.b_t), al, bl
        if self.bias is not None:
            p["bias"] = full("bias_q", sz.bias_q).view(torch.float32)
        self._packed = self._replicate(p)
        self._packed_only = True
        self.w_is_quantized = True
        self._w_ver = None

    @torch.no_grad()
    def _pack(self) -> None:
        """One-time operand preparation = linear.py:149-153 (weight.copy_(w_quantizer(weight)), same for
        bias) plus the build's packed images.  Like the reference, the weight/bias parameters hold the
        quantized values afterwards, and they are quantized exactly once: when only the derived images were dropped
        (.to(), .half(), A / B reloaded) the kept weight / bias images are reused."""
        W = self.weight.data
        ops._need_gpu(W)
        f = self._fmt
        with torch.cuda.device(W.device):
            if self.w_is_quantized and self._w_single is not None:
                w_img = self._w_single = self._w_single.to(W.device)
            else:
                w_img = self._w_single = ops.pack_weight(W, f["w"])
                # the parameter now carries w_quantizer(W) (|w| <= 1e-8 kept as is, block_fp.py:79-80)
                if int(getattr(f["w"], "block_rows", 1)) == 1:
                    wq = ops.quantize_mxint(W, f["w"], want=("deq",))["deq"]
                else:  # 2-D tiles: read the values back from the image just packed (what ops.quantize_mxint does for them)
                    wq = ops.unpack_weight(w_img, self.out_features, self.in_features, f["w"])
                    wq = torch.where(W.float().abs() <= 1e-8, W.float(), wq)
                self.weight.data.copy_(wq.to(W.dtype))
            p = {"w": w_img}
            if self.bias is not None:
                if self._bias_q is None:
                    self._bias_q = ops.pack_bias(self.bias.data, f["b"])
                    self.bias.data.copy_(self._bias_q[: self.out_features].to(self.bias.dtype))
                p["bias"] = self._bias_q = self._bias_q.to(W.device)
            if self.rank > 0:
                p["a_t"], p["b_t"], p["a_limbs"], p["b_limbs"] = ops.pack_lowrank(self.A.data, self.B.data)
            self._packed = self._replicate(p)
            if self._x_i8:
                # the int8 route's buffer starts with the sign-magnitude image: keep a VIEW of it as the single copy instead of a
                # second full 4.5-bit image for the module's lifetime (several GB on a 7B W4A8-INT model)
                self._w_single = self._single_copy("w")
        self.w_is_quantized = True
        self._w_ver = (self.weight._version, None if self.bias is None else self.bias._version)

    # -- forward -------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # (decode steps are host-bound through eager Python: this path is written for few interpreter operations - no
        # decorator, no repeated nn.Module attribute look-ups; kernels are launched through ctypes, autograd never sees them)
        g = self._group
        if g is not None and x is g._dx:  # an open decode round of this Linear's group (q/k/v, gate/up) for this very tensor
            y = g.take(self, x)
            if y is not None:
                return y
        if not self.is_ptq:
            raise NotImplementedError("lqer_amd implements the PTQ inference branch only (q_config['is_ptq'] = True)")
        if not x.is_cuda:
            raise RuntimeError("lqer_amd runs on the HIP device only: got a CPU tensor (there is no CPU fallback)")
        # the C ABI launches on the calling thread's current device: make x's device current for the call (a module on
        # cuda:1 under a single-process device map, reference experiments/infer_device_map.py:29-37)
        if _device_count() > 1 and x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):
                return self._forward_on_current_device(x)
        return self._forward_on_current_device(x)

    def _written_in_place(self) -> bool:
        """True (after dropping the images concerned) when `weight` / `bias` were written in place since their images were
        built - the version check every forward runs; a decode group runs it for EVERY member before it launches."""
        if self._packed is None or self._packed_only:
            # (images of a packed checkpoint are the only copy of the operands: its dense parameters are never consulted)
            return False
        prm = self._parameters
        bias_p = prm["bias"]
        cur = (prm["weight"]._version, None if bias_p is None else bias_p._version)
        if self._w_ver is None:  # first forward after copy.deepcopy / unpickling: the copied images belong to these parameters
            self._w_ver = cur
            return False
        if cur == self._w_ver:
            return False
        # the dense parameters were written in place since their images were built
        # (a bias that was NOT written already holds b_quantizer(bias): it must not be quantized a second time)
        keep_bias = self._bias_q if cur[1] == self._w_ver[1] else None
        self.invalidate_packed(weight_changed=cur[0] != self._w_ver[0], bias_changed=cur[1] != self._w_ver[1])
        self._bias_q = keep_bias
        return True

    def _forward_tiles(self, x: torch.Tensor) -> torch.Tensor:
        """The forward when an activation quantizer's blocks can span token rows (reference quantizers/utils.py:211-237 for
        [batch, tokens, features] tensors, :161-183 for 2-D tensors with skip_first_dim = false) - a COLD route, statement by
        statement what linear.py:145-157 does, because such blocks cannot live inside the fused kernels (a tile's exponent
        needs the maximum over R token rows: the quantizers become launches of their own over the whole tensor):
            x_q   = Q_x(x)                          HIP tile quantizer (ops.quantize_act_tiles -> lqer_quantize_mxint_tiles)
            main  = x_q W_q^T + b_q                 the fused HIP GEMM of a LinearFlexible twin with a pass-through x_quantizer that
                                                    shares weight / bias (x_q's values are exact in the module's dtype)
            side  = Q_Bout(Q_Aout(x_q A) B)         two rank-r library GEMMs (torch.matmul, as the reference) around the HIP quantizers
        No template configuration of the reference has such blocks."""
        f = self._fmt
        if x.dim() < 2 or x.dim() > 3:
            raise RuntimeError(f"Unsupported x.ndim = {x.dim()}")  # (quantizers/utils.py:284)
        if self._packed_only:
            raise NotImplementedError("this input's activation blocks span token rows (the tile route), which works from the dense "
                                      "parameters - the module was loaded from a packed checkpoint and holds the fused path's images only")

        def q(role, t):
            fm = f.get(role)
            if fm is None or fm.kind == _lib.Q_PASSTHROUGH:
                return t
            if fm.kind == _lib.Q_MXINT:
                return ops.quantize_act_tiles(t, fm)
            return ops.quantize_mxint(t, fm, want=("deq",))["deq"].to(t.dtype)  # integer: elementwise

        w, b = self._parameters["weight"], self._parameters["bias"]
        inner = self.__dict__.get("_inner")
        if inner is None:
            # the twin lives as long as this module and quantizes the weight / bias ONCE (linear.py:149-153), by the base class's own
            # rules: it is handed this module's Parameter objects before every call (.to() across device types replaces them),
            # notices in-place writes through their version counters, and invalidate_packed() is forwarded to it
            qc = dict(name="flexible", is_ptq=True, default=self.q_config.get("default"),
                      x_quantizer=dict(name="passthrough"), w_quantizer=deepcopy(self.q_config.get("w_quantizer", self.q_config.get("default"))))
            if b is not None:
                qc["b_quantizer"] = deepcopy(self.q_config.get("b_quantizer", self.q_config.get("default")))
            inner = self.__dict__["_inner"] = LinearFlexible(self.in_features, self.out_features, bias=b is not None, device="meta", q_config=qc)
        inner._parameters["weight"], inner._parameters["bias"] = w, b
        inner.tuning = self.tuning
        xq = q("x", x)
        y = inner(xq)
        self.w_is_quantized = True
        if self.rank > 0:
            prm = self._parameters
            xaq = q("A_out", torch.matmul(xq, prm["A"]))
            y = y + q("B_out", torch.matmul(xaq, prm["B"]))
        return y

    def _needs_tiles(self, x: torch.Tensor) -> bool:
        """Some activation format of this module MAY span token rows (`_tiles`): does it for THIS tensor?  The reference reads
        block_size against the tensor's rank at call time (quantizers/utils.py:261-284): a lone [L] with skip_first_dim = true means
        per-row blocks of L on a 2-D tensor and on a 3-D tensor with ONE token row (T = 1: decode steps) - those calls keep the fused
        kernels -, [1, T, L] tiles otherwise.  Every other tiled format stays on the tile route whatever the tensor."""
        if self._tiles_only or x.dim() < 2 or x.dim() > 3:
            return True
        f = self._fmt
        try:
            return any(ops.act_rows_per_block(f.get(r), x.shape) != 1 for r in ("x", "A_out", "B_out"))
        except (NotImplementedError, RuntimeError):
            return True

    def _forward_on_current_device(self, x: torch.Tensor) -> torch.Tensor:
        if self._tiles and self._needs_tiles(x):
            return self._forward_tiles(x)
        self._written_in_place()
        if self._packed is None or self.w_is_quantized is False:
            self._pack()
        K, N = self.in_features, self.out_features
        if x.shape[-1] != K:
            raise RuntimeError(f"expected last dim {K}, got {tuple(x.shape)}")
        if self._fmt["x"].kind == _lib.Q_PASSTHROUGH and x.dtype != self._parameters["weight"].dtype:
            # the packed images hold one copy per bf16 limb of the module's dtype (F.linear raises here as well)
            raise RuntimeError(f"expected input dtype {self.weight.dtype} (pass-through x_quantizer), got {x.dtype}")
        x2 = x.reshape(-1, K)
        if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < K):
            x2 = x2.contiguous()
        M = x2.shape[0]
        if self._group is not None and 0 < M <= 8:
            # decode sizes: the whole group (q/k/v, gate/up) in ONE launch; later members are handed the outputs it produced
            yg = self._group.decode_member(self, x, x2)
            if yg is not None:
                return yg  # (already [..., N])
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        if M == 0 or (self._group is not None and self._group.forward_member(self, x, x2, y)):
            return y.reshape(*x.shape[:-1], N)
        # per token count and dtype, built once: descriptor, workspace size and the constant part of the argument list
        # (decode-size forwards are host-bound: the ctypes marshalling of 16 arguments is not free)
        key = (M, x2.dtype, self.tuning)
        ent = self._fw_cache.get(key)
        if ent is None:
            desc = self._desc()
            p = self._packed
            if len(self._fw_cache) > 64:
                self._fw_cache = {}
            # (plain data only: the module must stay deep-copyable and picklable)
            a_t, a_limbs = self._side_image(M, desc, ops.dtype_code(x2))
            ent = self._fw_cache[key] = (desc, ops.linear_sizes(desc, M).workspace, ops.dtype_code(x2),
                                         (p["w"].data_ptr(), a_t, ops._ptr(p.get("b_t")),
                                          a_limbs, p.get("b_limbs", 0), ops._ptr(p.get("bias"))))
        desc, ws_bytes, dt, consts = ent
        dref = C.byref(desc)
        st = ops._stream(x.device)
        ws = ops.workspace_on(x.device, st, ws_bytes)
        rc = _lib.lib().lqer_linear_forward(dref, x2.data_ptr(), dt, M, x2.stride(0) if M > 1 else K, *consts,
                                            y.data_ptr(), N, ws.data_ptr(), ws.numel(), st)
        if rc:
            check(rc, "lqer_linear_forward")
        return y.reshape(*x.shape[:-1], N)

    def _side_image(self, M: int, desc, dt_code: int):
        """(pointer, a_limbs) of the A^T image a forward of M tokens is handed (also what bench.py passes through the C ABI): the limb
        image of lqer_pack_lowrank, or - at the token counts where their kernels run - the int8 route's single fp16 image (a_limbs = -1) /
        the block-16 route's bf16 image with its fragment-major copy (a_limbs = -2; beyond decode sizes only: the one-launch decode kernel
        and the groups take the plain one-limb image)."""
        p = self._packed
        a_t, a_limbs = ops._ptr(p.get("a_t")), p.get("a_limbs", 0)
        if self._x_i8 and "a_t_f16" in p and _lib.lib().lqer_gemm_route(C.byref(desc), M, dt_code) == _lib.ROUTE_I8:
            return p["a_t_f16"].data_ptr(), -1  # (the int8 kernel's token counts only: elsewhere the bf16 kernels run)
        if "a_t_b16" in p and M > 64 and not self._x_i8:
            return p["a_t_b16"].data_ptr(), -2
        return a_t, a_limbs

    def __repr__(self):
        return "{}(in_features={}, out_features={}, bias={}, is_ptq={}, rank={}, backend=hip/gfx950)".format(
            self.__class__.__name__, self.in_features, self.out_features, self.bias is not None, self.is_ptq, self.rank)


class SharedActivation:
    """Linears that receive the SAME input tensor (q/k/v, gate/up: llama_decoder.py:246-248, :176; opt_decoder.py
    q/k/v): the activation is quantized once and one side GEMM over the concatenation of the members' A matrices
    yields every member's x A.  Results are those of the members run one by one (same quantizers, same kernels for
    the main GEMM; the side product's fp32 summation order may differ).

    A member's forward uses the shared images only when it is handed the very tensor object the images were made
    from, unmodified (object identity + version counter) - an equal-looking tensor at a recycled address never hits.
    Requirements checked at construction: same in_features, same block_fp x / A_out quantizers, A_out blocks of 16 or
    one block per row with equal power-of-two ranks (a multiple of 16), rank > 0 for every member, padded ranks summing
    to at most 256 (else `enabled` is False and nothing changes for the members)."""

    _pool = {}  # (device, stream) -> {"xq", "xaq", "scr": uint8 tensors, "owner": (id(group), round)} - see forward_member

    def __init__(self, members):
        self.members = list(members)
        m0 = self.members[0]
        ok = all(isinstance(m, LinearFlexibleLqer) and m.rank > 0 and m.in_features == m0.in_features and not m._tiles for m in self.members)
        key = lambda f: (f.kind, f.width, f.block, f.exp_width, f.exp_bias)
        ok = ok and all(key(m._fmt["x"]) == key(m0._fmt["x"]) and key(m._fmt["A_out"]) == key(m0._fmt["A_out"]) for m in self.members)
        # A_out blocks must not straddle two members' columns of the concatenated x A: blocks of 16 (the padded ranks are
        # multiples of 16), or one block per member row (block_size [1, -1], the INT configurations) with equal ranks -
        # the group then quantizes in blocks of one member's padded rank
        ao = m0._fmt["A_out"] if ok else None
        self._aout_block = 16
        if ok and ao.kind == _lib.Q_MXINT and ao.block != 16:
            rps = {(m.rank + 15) // 16 * 16 for m in self.members}
            whole = ao.block <= 0 or all(ao.block >= m.rank for m in self.members)
            rp0 = next(iter(rps))
            ok = whole and len(rps) == 1 and all(m.rank == rp0 for m in self.members) and (rp0 & (rp0 - 1)) == 0
            self._aout_block = rp0
        ok = ok and ao.kind == _lib.Q_MXINT and len(self.members) > 1
        # the side GEMM kernels take a padded rank of at most 256 (csrc/lowrank_xa.hip): three rank-128 members do not fit
        # in one concatenation - such a group stays disabled and its members run one by one
        ok = ok and sum((m.rank + 15) // 16 * 16 for m in self.members) <= 256
        ok = ok and m0._fmt["x"].kind == _lib.Q_MXINT  # (pass-through activations: every member splits x itself)
        # (weights of 5..8 bits read a three-times repeated activation image: every member makes its own)
        ok = ok and all(ops.w_limbs(m._fmt["w"]) == 1 for m in self.members)
        self.enabled = bool(ok)
        self._cat = None      # concatenated A^T limb image + member offsets
        self._x = None        # the tensor the images below were made from (strong reference: its address stays taken)
        self._ver = -1
        self._round = 0       # rounds started by this group (pool ownership)
        self._cur = None
        self._served = set()  # members served from the current images
        self._plans = {}      # (M, dtype) -> launch constants of the group quantizer and of every member's GEMM
        # decode sizes (M <= 8): ONE launch for the whole group (lqer_linear_forward_group)
        self._dplans = {}     # (M, dtype) -> member table of the group launch, or None when the C ABI refuses the group
        self._dx, self._dver = None, -1   # the tensor the current outputs were computed from
        self._dstream = None  # ... and the stream they were launched on
        self._dys = None      # outputs of the current round, one per member
        self._dserved = set()
        if self.enabled:
            for i, m in enumerate(self.members):
                m._group = self
                m._gidx = i  # (its place in the group: the decode round's output table)

    def invalidate(self):
        self._cat, self._x, self._cur, self._served = None, None, None, set()
        self._plans = {}
        self._dplans, self._dx, self._dys, self._dserved = {}, None, None, set()

    def __getstate__(self):
        # copy.deepcopy / pickle (a whole model is copied with its groups): the launch plans and the current round hold raw device
        # pointers of THIS group's members and of the shared pool - the copy rebuilds them at its first call
        st = self.__dict__.copy()
        st.update(_cat=None, _x=None, _cur=None, _served=set(), _plans={}, _ver=-1, _dplans={}, _dx=None, _dys=None, _dserved=set(),
                  _dver=-1, _dstream=None)
        return st

    @classmethod
    def release_pool(cls):
        """Free the shared image pools of every (device, stream) - they are grow-only and outlive the models that used them.
        (A group's unfinished decode round keeps its activation and output tensors until the next round starts or
        `invalidate()` runs.)"""
        cls._pool.clear()

    @torch.no_grad()
    def _pack_cat(self, dev):
        L = _lib.lib()
        K = self.members[0].in_features
        offs, cols = [], []
        off = 0
        for m in self.members:
            rp = L.lqer_padded_r(m.rank)
            a = torch.zeros(K, rp, dtype=m.A.dtype, device=dev)
            a[:, : m.rank] = m.A.data.to(dev)
            cols.append(a)
            offs.append(off)
            off += rp
        a_cat = torch.cat(cols, dim=1).contiguous()
        dummy_b = torch.zeros(off, 16, dtype=a_cat.dtype, device=dev)
        a_t, _, a_limbs, _ = ops.pack_lowrank(a_cat, dummy_b)
        self._cat = {"a_t": a_t, "a_limbs": a_limbs, "offs": offs, "rp_total": off}
        # members on the int8 route with an fp16 A: the group's side GEMM too takes A as ONE fp16 image (a_limbs = -1)
        m0 = self.members[0]
        if all(m._x_i8 and m.i8_a_f16 for m in self.members) and a_limbs == 2 and m0._packed is not None:
            ok16, a16 = ops.a_f16_image(m0._packed["w"], m0.out_features, K, a_t, 2, off)
            if ok16:
                self._cat["a_t_f16"] = a16

    def take(self, mod, x):
        """A member comes with the tensor of the open decode round: its output, once, if the tensor is unmodified."""
        ys = self._dys
        if ys is None or (None if x.is_inference() else x._version) != self._dver:
            return None
        # the outputs were written on the stream that launched the group: a member called on another stream takes the
        # per-member route (no event, no wait - forward_member keys its pool by (device, stream) for the same reason)
        if ops._stream(x.device) != self._dstream or not mod.is_ptq:
            return None
        idx = mod._gidx
        served = self._dserved
        if idx in served:
            return None
        served.add(idx)
        y = ys[idx]
        if len(served) == len(ys):
            self._dx, self._dys = None, None  # every member served: do not pin the tensors until the next round
        return y

    def decode_member(self, mod, x, x2):
        """Up to 8 tokens: every member's forward in ONE launch (lqer_linear_forward_group: the producers multiply x with the
        concatenated A, each member's weight-streaming workgroups read their rank columns - per member the bits of its own
        forward).  The first member that is handed a tensor launches the group and gets its output; the others are handed
        theirs when they come with the very same tensor object, unmodified, once each.  None = not applicable (the caller
        takes the per-member route)."""
        if not self.enabled:
            return None
        ver = None if x.is_inference() else x._version
        idx = mod._gidx
        M, K = x2.shape
        dtc = ops.dtype_code(x2)
        # the launch reads EVERY member's images: each member's own in-place-write check (the calling member has just run its
        # own; a write to k_proj.weight must not leave k / v on the old images - ADVICE r4).  A member that was written drops
        # its images and, through invalidate_packed, this group's plans.
        for m in self.members:
            if m is not mod:
                if not m.is_ptq:
                    return None
                m._written_in_place()
        plan = self._dplans.get((M, dtc), False)
        if plan is None:
            return None
        L = _lib.lib()
        dev = x2.device
        if plan is False:
            if any(m._packed_only for m in self.members):  # (a packed checkpoint carries no dense A to concatenate)
                self._dplans[(M, dtc)] = None
                return None
            for m in self.members:
                if m._packed is None or m.w_is_quantized is False:
                    m._pack()
            if self._cat is None:
                self._pack_cat(dev)
            n = len(self.members)
            descs = [m._desc(plain=True) for m in self.members]  # (decode sizes never take the int8 tile kernel)
            tab = (_lib.GroupMember * n)()
            for i, (m, d) in enumerate(zip(self.members, descs)):
                pk = m._packed
                tab[i].desc, tab[i].w_packed, tab[i].b_t = C.pointer(d), pk["w"].data_ptr(), pk["b_t"].data_ptr()
                tab[i].b_limbs, tab[i].bias_q, tab[i].ldy = pk["b_limbs"], ops._ptr(pk.get("bias")), m.out_features
            ok = (2 <= n <= 4 and self._cat["a_limbs"] == 1 and all(L.lqer_decode_partials(C.byref(d), M) == 1 for d in descs)
                  and not any(m._x_f16 for m in self.members) and self._cat["rp_total"] <= 128)
            if not ok:
                self._dplans[(M, dtc)] = None
                return None
            if len(self._dplans) > 64:
                self._dplans = {}
            plan = self._dplans[(M, dtc)] = {
                "tab": tab, "descs": descs, "n": n, "Ns": [m.out_features for m in self.members],
                "ws": L.lqer_group_workspace_bytes(K, self._cat["rp_total"]), "a_t": self._cat["a_t"].data_ptr()}
        if x2.stride(0) < K or (x2.data_ptr() & 15) or (x2.stride(0) * x2.element_size()) % 16:
            return None
        Ns = plan["Ns"]
        # one tensor per member: a KV cache that keeps k or v must not keep q|k|v alive
        ys, tab, lead = [], plan["tab"], x.shape[:-1]
        for i, N in enumerate(Ns):
            yi = torch.empty(*lead, N, dtype=x.dtype, device=dev)
            ys.append(yi)
            tab[i].y = yi.data_ptr()
        st = ops._stream(dev)
        ws = ops.workspace_on(dev, st, plan["ws"])
        rc = L.lqer_linear_forward_group(tab, plan["n"], x2.data_ptr(), dtc, M, x2.stride(0) if M > 1 else K, plan["a_t"], 1,
                                         ws.data_ptr(), ws.numel(), st)
        if rc == -2:  # LQER_E_UNSUPPORTED (nothing was launched): this token count / shape stays on the per-member route
            self._dplans[(M, dtc)] = None
            return None
        if rc:
            check(rc, "lqer_linear_forward_group")
        self._dx, self._dver, self._dys, self._dserved, self._dstream = x, ver, ys, {idx}, st
        return ys[idx]

    @torch.no_grad()
    def forward_member(self, mod, x, x2, y) -> bool:
        """Run `mod`'s forward on the shared images; False = not applicable, the caller takes the ordinary route."""
        if not self.enabled or any(m._packed_only for m in self.members):  # (a packed checkpoint carries no dense A)
            return False
        for m in self.members:  # every member packed (the group image needs the final A values)
            if m._packed is None or m.w_is_quantized is False:
                m._pack()
        dev = x2.device
        if self._cat is None:
            self._pack_cat(dev)
        L = _lib.lib()
        M, K = x2.shape
        m0 = self.members[0]
        # the shared images serve a member only for the very tensor object they were made from, unmodified (version
        # counter; inference tensors have none - torch.inference_mode() - and rely on object identity), and only once per
        # member: a member that comes back with the same tensor starts a new round
        ver = None if x.is_inference() else x._version
        idx = self.members.index(mod)
        pkey = (dev, ops._stream(dev))  # like ops.workspace: two streams driving two groups must not share images
        pool = SharedActivation._pool.get(pkey)
        fresh = not (x is self._x and ver == self._ver and self._cur is not None and self._cur["M"] == M and idx not in self._served
                     and pool is not None and pool["owner"] == (id(self), self._round))  # (another group has used the pool since)
        # per token count and dtype, built once (descriptors, route and size queries are host time a 60-us GEMM does not hide):
        # members on the int8 route (per-token activations) share an int8 image only if every member's GEMM takes the int8
        # kernel at this token count, else everybody uses the bf16 image
        dtc = ops.dtype_code(x2)
        plan = self._plans.get((M, dtc))
        if plan is None:
            i8 = all(m._x_i8 for m in self.members) and \
                all(L.lqer_gemm_route(C.byref(m._desc()), M, dtc) == _lib.ROUTE_I8 for m in self.members)
            gdesc = m0._desc(plain=not i8)
            gdesc.rank = self._cat["rp_total"]
            gdesc.a_out_fmt.block = self._aout_block  # (one block per member row -> blocks of a member's rank)
            Mp, Kp = L.lqer_padded_m(M), L.lqer_padded_k(K)
            nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(gdesc), M)
            a_img = (self._cat["a_t_f16"].data_ptr(), -1) if i8 and "a_t_f16" in self._cat \
                else (self._cat["a_t"].data_ptr(), self._cat["a_limbs"])
            mem = []
            for m in self.members:  # (plain data and ctypes structs only)
                d = m._desc(plain=not i8)
                pk = m._packed
                mem.append((d, L.lqer_linear_gemm_scratch_bytes(C.byref(d), M), pk["w"].data_ptr(), pk["b_t"].data_ptr(),
                            pk["b_limbs"], ops._ptr(pk.get("bias")), m.out_features))
            if len(self._plans) > 64:
                self._plans = {}
            plan = self._plans[(M, dtc)] = {
                "gdesc": gdesc, "nscr": nscr, "a_img": a_img, "members": mem,
                "need": {"xq": Mp * Kp * 2, "xaq": Mp * self._cat["rp_total"] * 2, "scr": max(nscr, 16)}}
        st = ops._stream(dev)
        if fresh:
            # the images live in ONE grow-only pool per device, shared by every group (the groups of a model run one after
            # the other; 64 private copies would pin ~1 GiB at M = 2048): a group owns the pool from its first member's call
            # of a round to its last; a member that finds another owner re-makes the images (always correct, only slower)
            if pool is None:
                pool = SharedActivation._pool[pkey] = {"owner": None}
            for name, nbytes in plan["need"].items():
                if name not in pool or pool[name].numel() < nbytes:
                    pool[name] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._round += 1
            pool["owner"] = (id(self), self._round)
            b = {"xq": pool["xq"].data_ptr(), "xaq": pool["xaq"].data_ptr(), "keep": (pool["xq"], pool["xaq"])}
            rc = L.lqer_quantize_act_xa(C.byref(plan["gdesc"]), x2.data_ptr(), dtc, M, x2.stride(0) if M > 1 else K,
                                        *plan["a_img"], b["xq"], b["xaq"], pool["scr"].data_ptr(), plan["nscr"], st)
            if rc:
                check(rc, "lqer_quantize_act_xa (shared input)")
            self._x, self._ver, self._cur, self._served = x, ver, dict(b, M=M), set()
        cur = self._cur
        self._served.add(idx)
        if len(self._served) == len(self.members):
            self._x = None  # every member has been served: do not pin the activation tensor until the next call
        desc, gs, w_ptr, bt_ptr, b_limbs, bias_ptr, N = plan["members"][idx]
        scr = ops.workspace(dev, max(gs, 16))
        rc = L.lqer_linear_gemm_ld(C.byref(desc), cur["xq"], M, w_ptr, cur["xaq"] + 2 * self._cat["offs"][idx],
                                   self._cat["rp_total"], bt_ptr, b_limbs, bias_ptr, y.data_ptr(), dtc, N, scr.data_ptr(), gs, st)
        if rc:
            check(rc, "lqer_linear_gemm_ld (shared input)")
        return True


class LinearFlexible(_LinearBase):
    def _setup_quantizers(self, q_config: dict):
        # q_config["default"] is evaluated eagerly, as in the reference (linear.py:90-91): a config
        # without a "default" key raises KeyError there and here.
        x_cfg = deepcopy(q_config.get("x_quantizer", q_config["default"]))
        w_cfg = deepcopy(q_config.get("w_quantizer", q_config["default"]))
        self._fmt["x"] = ops.make_qfmt(x_cfg, "x")
        self._fmt["w"] = ops.make_qfmt(w_cfg, "w")
        if self.bias is not None:
            self._fmt["b"] = ops.make_qfmt(deepcopy(q_config.get("b_quantizer", q_config["default"])), "b")

    def _setup_lqer(self, l_config: dict):
        pass


class LinearFlexibleLqer(LinearFlexible):
    def _setup_quantizers(self, q_config: dict):
        LinearFlexible._setup_quantizers(self, q_config)
        fall = q_config.get("x_quantizer", q_config["default"])
        self._fmt["B_out"] = ops.make_qfmt(deepcopy(q_config.get("B_out_quantizer", fall)), "B_out")
        self._fmt["A_out"] = ops.make_qfmt(deepcopy(q_config.get("A_out_quantizer", fall)), "A_out")

    def _setup_lqer(self, l_config: dict):
        # y = x_q W_q^T + (x_q A) B ;  A [in, rank], B [rank, out], zeros until loaded (linear.py:134-143)
        self.A = nn.Parameter(torch.zeros(self.weight.shape[1], l_config["rank"]))
        self.B = nn.Parameter(torch.zeros(l_config["rank"], self.weight.shape[0]))

    @property
    def rank(self) -> int:
        return int(self.A.shape[1])


QUANTIZED_MODULE_MAP = {"linear": {"flexible": LinearFlexible, "flexible_lqer": LinearFlexibleLqer}}


def get_quantized_layer_cls(op: str, q_config: dict):
    assert op in QUANTIZED_MODULE_MAP, f"Unsupported quantized op: {op}"
    assert q_config["name"] in QUANTIZED_MODULE_MAP[op], f"Unsupported quantized config: {q_config}"
    return QUANTIZED_MODULE_MAP[op][q_config["name"]]

"""Torch-tensor wrappers over the C ABI (device pointers + the current HIP stream).  Host logic
only: shape checks, buffer allocation, format descriptors.  All arithmetic happens in the HIP
kernels; a CPU tensor is an error, not a fallback."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import LinearDesc, LinearSizes, QFmt, check

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"lqer_amd: unsupported dtype {t.dtype} (float32 / float16 / bfloat16)") from None


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("lqer_amd runs on the HIP device only: got a CPU tensor (there is no CPU fallback)")


def _on_tensor_device(fn):
    """The C ABI launches on the calling thread's CURRENT device (include/lqer_hip.h): run `fn` with the device of its
    first tensor argument current, so that a tensor on cuda:1 is never handed to a launch on cuda:0."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        t = next((a for a in args if torch.is_tensor(a)), None)
        if t is not None and t.is_cuda and t.device.index != torch.cuda.current_device():
            with torch.cuda.device(t.device):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)

    return wrapped


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def make_qfmt(cfg: Optional[dict], role: str = "x") -> QFmt:
    """One q_config entry (reference quantize/__init__.py:1-40 schema) -> lqer_qfmt_t.  `role` = which tensor the
    quantizer is applied to ("x", "A_out", "B_out": activations [.., tokens, features]; "w": the 2-D weight; "b": the
    1-D bias), because the reference right-aligns `block_size` to the tensor and `skip_first_dim` picks the blocking
    routine (quantizers/utils.py:42-67, :261-284).  The HIP path implements blocks that run along the last dim within
    one row: [1, L], or a one-entry [L] / [-1] where the reference reads it per row (activations and weights with
    skip_first_dim = true on 2-D tensors, and the bias), and for the WEIGHT also 2-D tiles [R, L] (skip_first_dim = false; a lone
    [L] then means all rows x L); activation formats whose blocks can span token rows (incl. a lone [L] on a 3-D tensor: all T
    rows x L) are recorded as `act_tiles` and take the module's tile route.  Anything else raises instead of being silently read per row.
    An `integer` WEIGHT (fixed point, signed, width 2..4) is packed as two's-complement nibbles and runs the 128-row tile kernel."""
    if cfg is None:
        raise KeyError("quantizer config missing")
    name = cfg["name"]
    if name == "passthrough":
        return QFmt(_lib.Q_PASSTHROUGH, 0, 0, 8, 127)
    if name == "integer":
        # fixed point (reference quantizers/integer.py:10-43): clamp(rne(x 2^frac_width), lo, hi) / 2^frac_width
        signed = bool(cfg.get("is_signed", True))
        if role == "w" and (not signed or not 2 <= int(cfg["width"]) <= 4):
            # (codes -2^(w-1) .. 2^(w-1)-1 travel as two's-complement nibbles of the packed image: signed, 2..4 bits)
            raise NotImplementedError("an integer weight quantizer must be signed with width 2..4 on the HIP path (4-bit packed image)")
        return QFmt(_lib.Q_INT, int(cfg["width"]), -1, 1 if signed else 0, int(cfg["frac_width"]))
    if name != "block_fp":
        raise NotImplementedError(f"quantizer '{name}' is not implemented on the HIP path (block_fp, integer, passthrough)")
    bs = cfg.get("block_size", [16])
    bs = [bs] if isinstance(bs, int) else list(bs)
    skip = bool(cfg.get("skip_first_dim", True))
    if role == "b":
        assert not skip, "skip_first_dim must be False for bias to be blocked"  # (utils.py:268-271)
        if len(bs) > 1 and any(b != 1 for b in bs[:-1]):
            bs = bs[-1:]  # right-aligned to a 1-D tensor: only the last entry counts
    else:
        block_rows = 1
        act_tiles = None
        if role == "w" and not skip:
            # the 2-D weight with skip_first_dim = false (quantizers/utils.py:161-183): [R, L] = tiles of R rows x L k, a lone [L]
            # is right-aligned to [-1, L] = all rows x L (utils.py:42-67).  One exponent per tile, repeated per row in the image.
            if len(bs) > 2 and any(b != 1 for b in bs[:-2]):
                raise NotImplementedError(f"block_size {bs}: more entries than the weight has dims")
            block_rows = int(bs[-2]) if len(bs) >= 2 else -1
            if block_rows == 0:
                raise ValueError(f"block_size {bs}: a block of 0 rows")
        elif role in ("x", "A_out", "B_out") and (not skip or len(bs) == 1 or bs[-2] != 1):
            # an activation whose blocks can span token rows (quantizers/utils.py:211-237: [R, L] tiles over (tokens, features) of
            # every batch element of a 3-D tensor; :261-270: a 2-D tensor with skip_first_dim = false is blocked like a weight, a
            # lone [L] then means all rows x L).  What the tiles are depends on the tensor's rank at call time (a 2-D tensor with
            # skip_first_dim = true is blocked per row whatever R says, utils.py:127-144): the module's tile route decides there
            # (linear.py `_forward_tiles`, ops.quantize_act_tiles) - the fused kernels never see this format.
            # A ONE-entry [L] with skip_first_dim = true - the reference quantizer's own default, block_fp.py:111-118 - is such a
            # format too: right-aligned to a [batch, tokens, features] tensor it reads [1, T, L], one exponent for ALL token rows x L
            # columns of a batch element (_infer_block_shape prepends -1, utils.py:56-66; _block_3d_activation :211-237); on a 2-D
            # tensor the same entry means per-row blocks of L and runs the fused kernels (the module decides per call).
            # (4th entry: the format is that lone [L] - the only tiled format whose 2-D reading the fused kernels serve)
            act_tiles = (int(bs[-2]) if len(bs) >= 2 else -1, int(bs[-1]), skip, skip and len(bs) == 1)
        elif any(b != 1 for b in bs[:-1]):
            raise NotImplementedError(f"block_size {bs}: only blocks along the last dim are implemented on the HIP path for '{role}'")
    ew = int(cfg.get("exponent_width", 8))
    eb = cfg.get("exponent_bias", None)
    eb = 2 ** (ew - 1) - 1 if eb in (None, "none", "None", "NA") else int(eb)
    fmt = QFmt(_lib.Q_MXINT, int(cfg.get("width", 12)), int(bs[-1]), ew, eb)
    if role != "b" and block_rows != 1:
        fmt.block_rows = block_rows  # (a Python attribute beside the C fields: only the packing call needs it)
    if role != "b" and act_tiles is not None:
        fmt.act_tiles = act_tiles    # (R, L, skip_first_dim) as configured - a Python attribute, see above
    return fmt


def act_rows_per_block(fmt: Optional[QFmt], shape) -> int:
    """Token rows one exponent of an activation format spans on a tensor of `shape` (1: blocks run along the last dim within one row -
    the fused kernels' layout).  Raises what act_tile_shape raises."""
    if fmt is None or getattr(fmt, "act_tiles", None) is None:
        return 1
    R, _ = act_tile_shape(fmt, len(shape))
    rows = int(shape[-2])
    return rows if R <= 0 or R > rows else R


def act_tile_shape(fmt: QFmt, ndim: int):
    """(R, L) of an activation format for a tensor of `ndim` dims, as the reference's dispatch reads it (quantizers/utils.py:261-284;
    R, L <= 0: the whole extent): per-row blocks -> (1, L); `None`-free: raises what the reference raises."""
    t = getattr(fmt, "act_tiles", None)
    if t is None:
        return 1, int(fmt.block)
    R, L, skip = t[:3]
    if ndim == 2:
        return (1, L) if skip else (R, L)  # utils.py:127-144 infers the block against ONE row; :161-183 tiles the matrix
    if ndim == 3:
        if not skip:
            raise NotImplementedError("block 3d weight is not supported.")  # (utils.py:279, verbatim)
        return R, L
    raise RuntimeError(f"Unsupported x.ndim = {ndim}")  # (utils.py:284)


@_on_tensor_device
def quantize_act_tiles(x: torch.Tensor, fmt: QFmt) -> torch.Tensor:
    """block_fp quantizer of an ACTIVATION whose format may have blocks spanning token rows (make_qfmt roles x / A_out / B_out with
    `act_tiles`): x [tokens, features] or [batch, tokens, features] -> the quantized tensor, same shape and dtype (the reference
    quantizer's contract, block_fp.py:111).  Tiles are anchored per batch element (utils.py:211-237)."""
    _need_gpu(x)
    R, L = act_tile_shape(fmt, x.dim())
    xc = x.contiguous()
    rows, cols = xc.shape[-2], xc.shape[-1]
    batches = xc.shape[0] if xc.dim() == 3 else 1
    Re = rows if R <= 0 or R > rows else R
    Le = cols if L <= 0 or L > cols else L
    out = torch.empty(xc.shape, dtype=torch.float32, device=x.device)
    if xc.numel() == 0:
        return out.to(x.dtype)
    amax = torch.empty(batches * (-(-rows // Re)) * (-(-cols // Le)), dtype=torch.float32, device=x.device)
    check(_lib.lib().lqer_quantize_mxint_tiles(xc.data_ptr(), dtype_code(xc), batches, rows, cols, C.byref(fmt), Re, Le, out.data_ptr(),
                                               amax.data_ptr(), _stream(x.device)), "lqer_quantize_mxint_tiles")
    return out.to(x.dtype)


@_on_tensor_device
def quantize_mxint(x: torch.Tensor, fmt: QFmt, want=("deq", "codes", "exps")) -> Dict[str, torch.Tensor]:
    """MXINT quantizer over the last dim of x (any leading dims).  Returns the requested images.  A WEIGHT format with 2-D
    tiles (`fmt.block_rows` != 1, make_qfmt role "w") is honoured too: the values then come from the packed image of
    lqer_pack_weight_mxint_2d (one exponent per tile), |x| <= 1e-8 kept as is (block_fp.py:79-80) - `deq` of a 2-D tensor only."""
    _need_gpu(x)
    if int(getattr(fmt, "block_rows", 1)) != 1:
        if tuple(want) != ("deq",) or x.dim() != 2:
            raise NotImplementedError("quantize_mxint with 2-D weight tiles (block_rows != 1) returns `deq` of a 2-D tensor only")
        xf = x.float()
        deq = unpack_weight(pack_weight(x, fmt), x.shape[0], x.shape[1], fmt)
        return {"deq": torch.where(xf.abs() <= 1e-8, xf, deq)}
    cols = x.shape[-1]
    x2 = x.reshape(-1, cols)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    L = cols if fmt.block <= 0 or fmt.block >= cols else fmt.block
    out: Dict[str, torch.Tensor] = {}
    if "deq" in want:
        out["deq"] = torch.empty(rows, cols, dtype=torch.float32, device=x.device)
    if "codes" in want:
        out["codes"] = torch.empty(rows, cols, dtype=torch.int8, device=x.device)
    if "exps" in want:
        out["exps"] = torch.zeros(rows, -(-cols // L) if L else 0, dtype=torch.int8, device=x.device)
    check(
        _lib.lib().lqer_quantize_mxint(
            x2.data_ptr(), dtype_code(x2), rows, cols, x2.stride(0) if rows > 1 else cols, C.byref(fmt),
            _ptr(out.get("deq")), _ptr(out.get("codes")), _ptr(out.get("exps")), _stream(x.device)),
        "lqer_quantize_mxint",
    )
    for k in ("deq", "codes"):
        if k in out:
            out[k] = out[k].reshape(x.shape)
    return out


@_on_tensor_device
def quantize_act(x2: torch.Tensor, fmt: QFmt, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [M,K] -> exact bf16 image [Mp,Kp] of x_quantizer(x)."""
    _need_gpu(x2)
    M, K = x2.shape
    L = _lib.lib()
    Mp, Kp = L.lqer_padded_m(M), L.lqer_padded_k(K)
    if out is None:
        out = torch.empty(Mp, Kp, dtype=torch.bfloat16, device=x2.device)
    check(
        L.lqer_quantize_act_mxint(x2.data_ptr(), dtype_code(x2), M, K, x2.stride(0) if M > 1 else K, C.byref(fmt), out.data_ptr(), _stream(x2.device)),
        "lqer_quantize_act_mxint",
    )
    return out


def w_limbs(fmt: QFmt) -> int:
    """4-bit limb images of a packed weight: 3 for block_fp weights of 5..8 bits (include/lqer_hip.h "weights of 5..8 bits"), else 1."""
    return 3 if (fmt.kind == _lib.Q_MXINT and fmt.width > 4) else 1


def linear_sizes(desc: LinearDesc, m_max: int) -> LinearSizes:
    sz = LinearSizes()
    check(_lib.lib().lqer_linear_sizes(C.byref(desc), m_max, C.byref(sz)), "lqer_linear_sizes")
    return sz


@_on_tensor_device
def pack_weight(W: torch.Tensor, fmt: QFmt) -> torch.Tensor:
    """W [N,K] -> packed panels.  A weight format with 2-D tiles carries its row count in `fmt.block_rows` (make_qfmt)."""
    _need_gpu(W)
    N, K = W.shape
    if W.stride(-1) != 1:
        W = W.contiguous()
    L = _lib.lib()
    Np, Kp = L.lqer_padded_n(N), L.lqer_padded_k(K)
    packed = torch.empty((Np // 16) * (Kp // 64) * 576 * w_limbs(fmt), dtype=torch.uint8, device=W.device)
    scratch = torch.empty(N * (-(-K // 16)), dtype=torch.int8, device=W.device)
    rows = int(getattr(fmt, "block_rows", 1))
    if rows == 1:
        rc = L.lqer_pack_weight_mxint(W.data_ptr(), dtype_code(W), N, K, W.stride(0), C.byref(fmt), packed.data_ptr(), scratch.data_ptr(),
                                      _stream(W.device))
    else:
        rc = L.lqer_pack_weight_mxint_2d(W.data_ptr(), dtype_code(W), N, K, W.stride(0), C.byref(fmt), rows, packed.data_ptr(),
                                         scratch.data_ptr(), _stream(W.device))
    check(rc, "lqer_pack_weight_mxint")
    return packed


@_on_tensor_device
def replicate_rows(src: torch.Tensor, rows: int, row_bytes: int, copies: int) -> torch.Tensor:
    """[rows][row_bytes] -> [rows][copies * row_bytes] (every row repeated): the packed images of a Linear with
    pass-through activations hold one copy per bf16 limb of the activation (include/lqer_hip.h)."""
    if copies == 1:
        return src
    _need_gpu(src)
    flat = src.reshape(-1).view(torch.uint8)
    if flat.numel() < rows * row_bytes:
        raise ValueError(f"replicate_rows: {flat.numel()} B < {rows} x {row_bytes} B")
    dst = torch.empty(rows * row_bytes * copies, dtype=torch.uint8, device=src.device)
    check(_lib.lib().lqer_replicate_rows(flat.data_ptr(), dst.data_ptr(), rows, row_bytes, copies, _stream(src.device)), "lqer_replicate_rows")
    return dst.view(src.dtype)


@_on_tensor_device
def f16_prepare(w_packed: torch.Tensor, N: int, K: int, a_t_limbs: Optional[torch.Tensor], a_limbs: int, r: int):
    """Eligibility of the fp16 fast path of pass-through fp16 activations (include/lqer_hip.h, lqer_f16_prepare):
    returns (ok, a_t_f16) - ok is False when a weight block scale or an element of A is outside fp16.  Synchronises."""
    _need_gpu(w_packed)
    L = _lib.lib()
    dev = w_packed.device
    flags = torch.zeros(2, dtype=torch.int32, device=dev)
    a16 = torch.empty(L.lqer_a_f16_image_bytes(K, r) // 2, dtype=torch.float16, device=dev) if r > 0 else None  # ([rp][Kp] + fragment-major copy)
    check(L.lqer_f16_prepare(w_packed.data_ptr(), N, K, _ptr(a_t_limbs), a_limbs, r, _ptr(a16), flags.data_ptr(), _stream(dev)), "lqer_f16_prepare")
    fl = flags.tolist()
    return (fl[0] == 0 and fl[1] == 0), a16


@_on_tensor_device
def a_f16_image(w_packed: torch.Tensor, N: int, K: int, a_t_limbs: torch.Tensor, a_limbs: int, r: int):
    """A^T as ONE fp16 image [rp][Kp] for the int8 route's side GEMM (int8 mantissas x fp16 A on the fp16 MFMA, a_limbs = -1
    in lqer_lowrank_xa / lqer_quantize_act_xa / lqer_linear_forward): (ok, image) - ok is False when an element of A is not
    exact in fp16 (only that flag of lqer_f16_prepare matters here).  Synchronises."""
    L = _lib.lib()
    dev = w_packed.device
    flags = torch.zeros(2, dtype=torch.int32, device=dev)
    a16 = torch.empty(L.lqer_a_f16_image_bytes(K, r) // 2, dtype=torch.float16, device=dev)  # ([rp][Kp] + its fragment-major copy)
    check(L.lqer_f16_prepare(w_packed.data_ptr(), N, K, a_t_limbs.data_ptr(), a_limbs, r, a16.data_ptr(), flags.data_ptr(), _stream(dev)),
          "lqer_f16_prepare")
    return flags.tolist()[1] == 0, a16


@_on_tensor_device
def a_b16_image(a_t_limbs: torch.Tensor, K: int, r: int) -> torch.Tensor:
    """Limb 0 of `lqer_pack_lowrank`'s A^T image [rp][Kp] followed by its fragment-major copy (include/lqer_hip.h, lqer_a_b16_prepare):
    what the one-launch activation kernel of block-16 MXINT configurations reads (a_limbs = -2)."""
    L = _lib.lib()
    dev = a_t_limbs.device
    out = torch.empty(L.lqer_a_b16_image_bytes(K, r) // 2, dtype=torch.bfloat16, device=dev)
    check(L.lqer_a_b16_prepare(a_t_limbs.data_ptr(), K, r, out.data_ptr(), _stream(dev)), "lqer_a_b16_prepare")
    return out


@_on_tensor_device
def i8_prepare(w_packed: torch.Tensor, N: int, K: int, w_fmt: QFmt):
    """Eligibility of the int8 MFMA route (include/lqer_hip.h "int8 route") for one packed weight: returns (ok, buffer) -
    `buffer` holds the sign-magnitude image followed by the int8 main loop's image; ok is False when some row's integer
    sums could leave the i32 range (the caller keeps the plain image).  Synchronises."""
    _need_gpu(w_packed)
    L = _lib.lib()
    dev = w_packed.device
    none = QFmt(_lib.Q_PASSTHROUGH, 0, 0, 8, 127)
    desc = LinearDesc(K, N, 0, 0, QFmt(_lib.Q_MXINT_I8, 8, -1, 8, 127), w_fmt, none, none, none)
    sz = linear_sizes(desc, 1)
    buf = torch.zeros(sz.w_packed, dtype=torch.uint8, device=dev)
    flat = w_packed.reshape(-1).view(torch.uint8)
    buf[: flat.numel()] = flat
    flags = torch.zeros(2, dtype=torch.int32, device=dev)
    check(L.lqer_i8_prepare(buf.data_ptr(), N, K, C.byref(w_fmt), flags.data_ptr(), _stream(dev)), "lqer_i8_prepare")
    return flags.tolist()[0] == 0, buf


@_on_tensor_device
def unpack_weight_i8(w_packed: torch.Tensor, N: int, K: int, w_fmt: Optional[QFmt] = None) -> torch.Tensor:
    """Test hook: the int8 route's weight image (inside the buffer of i8_prepare) -> dequantized fp32 [N, K] (`w_fmt`: needed for
    weights of 5..8 bits, whose image of codes lies behind three limb images)."""
    _need_gpu(w_packed)
    out = torch.empty(N, K, dtype=torch.float32, device=w_packed.device)
    if w_fmt is None:
        check(_lib.lib().lqer_unpack_weight_i8(w_packed.data_ptr(), N, K, out.data_ptr(), _stream(w_packed.device)), "lqer_unpack_weight_i8")
    else:
        check(_lib.lib().lqer_unpack_weight_i8_fmt(w_packed.data_ptr(), N, K, C.byref(w_fmt), out.data_ptr(), _stream(w_packed.device)),
              "lqer_unpack_weight_i8_fmt")
    return out


@_on_tensor_device
def quantize_act_i8(x2: torch.Tensor, fmt: QFmt):
    """Test hook: x [M, K] -> (int8 mantissas [Mp, K padded to 128], row scales fp32 [Mp]) of the int8 route's image."""
    _need_gpu(x2)
    M, K = x2.shape
    L = _lib.lib()
    Mp, Kp8 = L.lqer_padded_m(M), -(-K // 128) * 128
    img = (Mp * Kp8 + 255) // 256 * 256
    buf = torch.zeros(img + Mp * 4, dtype=torch.uint8, device=x2.device)
    check(L.lqer_quantize_act_i8(x2.data_ptr(), dtype_code(x2), M, K, x2.stride(0) if M > 1 else K, C.byref(fmt), buf.data_ptr(),
                                 _stream(x2.device)), "lqer_quantize_act_i8")
    return buf[: Mp * Kp8].view(torch.int8).view(Mp, Kp8), buf[img:].view(torch.float32)


def desc_limbs(desc: LinearDesc) -> Tuple[int, int]:
    """(bf16 limbs of the activation image, of the x A image): 1, 1 unless x / A_out are pass-through."""
    a, b = C.c_int(1), C.c_int(1)
    check(_lib.lib().lqer_desc_limbs(C.byref(desc), C.byref(a), C.byref(b)), "lqer_desc_limbs")
    return a.value, b.value


@_on_tensor_device
def unpack_weight(packed: torch.Tensor, N: int, K: int, fmt: QFmt) -> torch.Tensor:
    _need_gpu(packed)
    out = torch.empty(N, K, dtype=torch.float32, device=packed.device)
    check(_lib.lib().lqer_unpack_weight_mxint(packed.data_ptr(), N, K, C.byref(fmt), out.data_ptr(), _stream(packed.device)), "lqer_unpack_weight_mxint")
    return out


@_on_tensor_device
def pack_lowrank(A: torch.Tensor, B: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, int, int]:
    """A [K,r], B [r,N] -> (a_t, b_t, a_limbs, b_limbs).  Synchronises once to read the limb counts."""
    _need_gpu(A, B)
    K, r = A.shape
    r2, N = B.shape
    if r != r2 or A.dtype != B.dtype:
        raise ValueError(f"A {tuple(A.shape)} {A.dtype} and B {tuple(B.shape)} {B.dtype} do not match")
    L = _lib.lib()
    Kp, Np, rp = L.lqer_padded_k(K), L.lqer_padded_n(N), L.lqer_padded_r(r)
    a_t = torch.empty(3 * rp * Kp, dtype=torch.bfloat16, device=A.device)
    b_t = torch.empty(3 * Np * rp, dtype=torch.bfloat16, device=A.device)
    flags = torch.zeros(2, dtype=torch.int32, device=A.device)
    A, B = A.contiguous(), B.contiguous()
    check(
        L.lqer_pack_lowrank(A.data_ptr(), B.data_ptr(), dtype_code(A), K, N, r, a_t.data_ptr(), b_t.data_ptr(), flags.data_ptr(), _stream(A.device)),
        "lqer_pack_lowrank",
    )
    fl = flags.tolist()
    return a_t, b_t, int(fl[0]), int(fl[1])


@_on_tensor_device
def pack_bias(bias: torch.Tensor, fmt: QFmt) -> torch.Tensor:
    _need_gpu(bias)
    N = bias.shape[0]
    out = torch.empty(_lib.lib().lqer_padded_n(N), dtype=torch.float32, device=bias.device)
    bias = bias.contiguous()
    check(_lib.lib().lqer_pack_bias(bias.data_ptr(), dtype_code(bias), N, C.byref(fmt), out.data_ptr(), _stream(bias.device)), "lqer_pack_bias")
    return out


# grow-only per-(device, stream) scratch shared by every Linear (stream-ordered reuse is safe)
_workspaces: Dict[Tuple[int, int], torch.Tensor] = {}


def workspace_on(dev: torch.device, stream: int, nbytes: int) -> torch.Tensor:
    """workspace() for a caller that already holds the stream handle (one torch.cuda.current_stream() less per forward)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws


def workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws

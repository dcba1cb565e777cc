"""ctypes binding of liblqer_hip.so (include/lqer_hip.h).  No CPU fallback: if the shared library
is missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LQER_AMD_LIB: another build of the library - A/B measurements of one bench command across builds on one box)
LIB_PATH = os.environ.get("LQER_AMD_LIB") or os.path.join(_HERE, "liblqer_hip.so")
BUILD_SCRIPT = os.path.join(_HERE, "csrc", "build.sh")

ABI_VERSION = 13
F32, F16, BF16 = 0, 1, 2
Q_PASSTHROUGH, Q_MXINT, Q_PASSTHROUGH_F16, Q_MXINT_I8, Q_INT = 0, 1, 2, 3, 4
K_ALIGN, M_ALIGN, N_ALIGN, R_ALIGN = 64, 256, 256, 16
ROUTE_SMALLM, ROUTE_TILE128, ROUTE_TILE256, ROUTE_I8 = 0, 1, 2, 3
TUNE_TILE_ROWS_128, TUNE_TILE_ROWS_64, TUNE_DECODE_NO_POLL, TUNE_XA_REDUCE_IN_GEMM = 0x1, 0x2, 0x10000, 0x20000
TUNE_I8_ROWS_128, TUNE_I8_ROWS_256, TUNE_AMAX_ATOMIC, TUNE_AMAX_PARTS = 0x4, 0x8, 0x40000, 0x80000
TUNE_ACT16_SPLIT, TUNE_ACT16_FUSED = 0x800000, 0x1000000  # block-16 MXINT activation side: two launches / the one-launch kernel at every M
TUNE_ACT8_SPLIT, TUNE_ACT8_FUSED = 0x200000, 0x400000  # int8 route's activation side: three launches / the one-launch kernel at every M
TUNE_AMAX_NO_MRX = 0x4000000  # int8 route over several rounds of 128-row tiles: the k_bout_amax pre-pass launch instead of the in-GEMM items
TUNE_BOUT_IN_PROLOGUE = 0x2000000  # 128-row tile kernel: the B_out re-quantization in front of the main loop (rounds 1-5) instead of under it
TUNE_AMAX_XCH_MISS = 0x100000  # int8 route's in-GEMM exchange of the B_out row maxima: every workgroup takes its fall-back


def tune_xcd_block(t: int) -> int:
    return (t & 0x3F) << 4



class QFmt(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_int32), ("block", C.c_int32), ("exp_width", C.c_int32), ("exp_bias", C.c_int32)]


class LinearDesc(C.Structure):
    _fields_ = [
        ("in_features", C.c_int32),
        ("out_features", C.c_int32),
        ("rank", C.c_int32),
        ("has_bias", C.c_int32),
        ("x_fmt", QFmt),
        ("w_fmt", QFmt),
        ("b_fmt", QFmt),
        ("a_out_fmt", QFmt),
        ("b_out_fmt", QFmt),
        ("tuning", C.c_int32),  # LQER_TUNE_* (0 = defaults): per-call kernel-variant knobs of tests / measurements
    ]


class GroupMember(C.Structure):
    """lqer_group_member_t: one Linear of a one-launch decode group (lqer_linear_forward_group)."""
    _fields_ = [("desc", C.POINTER(LinearDesc)), ("w_packed", C.c_void_p), ("b_t", C.c_void_p), ("b_limbs", C.c_int32),
                ("bias_q", C.c_void_p), ("y", C.c_void_p), ("ldy", C.c_int64)]


class LinearSizes(C.Structure):
    _fields_ = [("w_packed", C.c_size_t), ("a_t", C.c_size_t), ("b_t", C.c_size_t), ("bias_q", C.c_size_t), ("workspace", C.c_size_t)]


_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_qp, _dp = C.POINTER(QFmt), C.POINTER(LinearDesc)

# name -> (restype, argtypes); must list every symbol declared in include/lqer_hip.h
SIGNATURES = {
    "lqer_version": (_i, []),
    "lqer_last_error": (C.c_char_p, []),
    "lqer_sizeof_qfmt": (_sz, []),
    "lqer_sizeof_linear_desc": (_sz, []),
    "lqer_sizeof_linear_sizes": (_sz, []),
    "lqer_sizeof_group_member": (_sz, []),
    "lqer_padded_k": (_i64, [_i64]),
    "lqer_padded_n": (_i64, [_i64]),
    "lqer_padded_m": (_i64, [_i64]),
    "lqer_padded_r": (_i64, [_i64]),
    "lqer_quantize_mxint": (_i, [_vp, _i, _i64, _i64, _i64, _qp, _vp, _vp, _vp, _vp]),
    "lqer_quantize_mxint_tiles": (_i, [_vp, _i, _i64, _i64, _i64, _qp, _i64, _i64, _vp, _vp, _vp]),
    "lqer_quantize_act_mxint": (_i, [_vp, _i, _i64, _i64, _i64, _qp, _vp, _vp]),
    "lqer_linear_sizes": (_i, [_dp, _i64, C.POINTER(LinearSizes)]),
    "lqer_pack_weight_mxint": (_i, [_vp, _i, _i64, _i64, _i64, _qp, _vp, _vp, _vp]),
    "lqer_pack_weight_mxint_2d": (_i, [_vp, _i, _i64, _i64, _i64, _qp, _i64, _vp, _vp, _vp]),
    "lqer_unpack_weight_mxint": (_i, [_vp, _i64, _i64, _qp, _vp, _vp]),
    "lqer_pack_lowrank": (_i, [_vp, _vp, _i, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "lqer_pack_bias": (_i, [_vp, _i, _i64, _qp, _vp, _vp]),
    "lqer_linear_forward": (_i, [_dp, _vp, _i, _i64, _i64, _vp, _vp, _vp, _i, _i, _vp, _vp, _i64, _vp, _sz, _vp]),
    "lqer_quantize_act_xa": (_i, [_dp, _vp, _i, _i64, _i64, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "lqer_lowrank_xa_scratch_bytes": (_sz, [_dp, _i64]),
    "lqer_act_image_bytes": (_sz, [_dp, _i64]),
    "lqer_lowrank_xa": (_i, [_dp, _vp, _i64, _vp, _i, _vp, _vp, _sz, _vp]),
    "lqer_linear_gemm_scratch_bytes": (_sz, [_dp, _i64]),
    "lqer_linear_gemm": (_i, [_dp, _vp, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _i64, _vp, _sz, _vp]),
    # (ABI 13) the GEMM pre-pass's zero fill written by the one-launch int8 activation kernel in front of it
    "lqer_quantize_act_xa_prep": (_i, [_dp, _vp, _i, _i64, _i64, _vp, _i, _vp, _vp, _vp, _sz, _vp, C.POINTER(_sz), _vp]),
    "lqer_linear_gemm_prepared": (_i, [_dp, _vp, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _i64, _vp, _sz, _sz, _vp]),
    "lqer_linear_gemm_ld": (_i, [_dp, _vp, _i64, _vp, _vp, _i64, _vp, _i, _vp, _vp, _i, _i64, _vp, _sz, _vp]),
    "lqer_desc_limbs": (_i, [_dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lqer_replicate_rows": (_i, [_vp, _vp, _i64, _i64, _i, _vp]),
    "lqer_decode_partials": (_i, [_dp, _i64]),
    "lqer_tile_partials": (_i, [_dp, _i64, _i]),
    "lqer_group_workspace_bytes": (_sz, [_i64, _i64]),
    "lqer_linear_forward_group": (_i, [C.POINTER(GroupMember), _i, _vp, _i, _i64, _i64, _vp, _i, _vp, _sz, _vp]),
    "lqer_gemm_route": (_i, [_dp, _i64, _i]),
    "lqer_gemm_tile_rows": (_i, [_dp, _i64, _i]),
    "lqer_f16_prepare": (_i, [_vp, _i64, _i64, _vp, _i, _i64, _vp, _vp, _vp]),
    "lqer_a_f16_image_bytes": (C.c_size_t, [_i64, _i64]),
    "lqer_a_b16_image_bytes": (C.c_size_t, [_i64, _i64]),
    "lqer_a_b16_prepare": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "lqer_i8_prepare": (_i, [_vp, _i64, _i64, _qp, _vp, _vp]),
    "lqer_unpack_weight_i8": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "lqer_unpack_weight_i8_fmt": (_i, [_vp, _i64, _i64, _qp, _vp, _vp]),
    "lqer_quantize_act_i8": (_i, [_vp, _i, _i64, _i64, _i64, _qp, _vp, _vp]),
    "lqer_clock_probe": (_i, [_vp, _i, _i64, _vp]),
    "lqer_matmul_q_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "lqer_matmul_q_workspace_bytes_fmt": (_sz, [_i64, _i64, _i64, _i64, _qp, _qp]),
    "lqer_matmul_q": (_i, [_vp, _vp, _vp, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _qp, _qp, _vp, _sz, _vp]),
}

_lib: Optional[C.CDLL] = None


class LqerHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["bash", BUILD_SCRIPT], capture_output=True, text=True)
    if res.returncode != 0:
        raise LqerHipError("hipcc build of liblqer_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout.strip())
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LqerHipError(
                f"{LIB_PATH} not found: build it with `bash {BUILD_SCRIPT}` (or __graft_entry__.build()). "
                "lqer_amd has no CPU fallback."
            )
        # RTLD_NOW: an unresolved symbol (a kernel stub the host pass did not emit) fails HERE, not at the first launch
        L = C.CDLL(LIB_PATH, mode=os.RTLD_NOW)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.lqer_version() != ABI_VERSION:
            raise LqerHipError(f"ABI mismatch: library {L.lqer_version()} vs binding {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise LqerHipError(f"{what} failed (code {rc}): {lib().lqer_last_error().decode()}")

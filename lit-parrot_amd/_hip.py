"""ctypes binding of ``libparrot_hip.so`` (the C ABI declared in ``include/parrot_hip.h``).

There is no CPU fallback: if the library is missing, or a call fails, a ``ParrotHipError`` is raised.
PyTorch is only used by the callers for device memory and streams; this module passes raw pointers.
"""
import ctypes as C
from pathlib import Path
from typing import Optional

import torch

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libparrot_hip.so"

EPI_NONE, EPI_RESIDUAL, EPI_GELU, EPI_SWIGLU = 0, 1, 2, 3


class ParrotHipError(RuntimeError):
    pass


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float


class ParrotNorm(C.Structure):
    """``parrot_norm_t`` of include/parrot_hip.h: a norm fused in front of a Linear."""

    _fields_ = [("kind", C.c_int), ("weight", C.c_void_p), ("bias", C.c_void_p), ("eps", C.c_float), ("rsqrt_mode", C.c_int)]


_np = C.POINTER(ParrotNorm)

# ---- stream engine (parrot_eng_op_t / parrot_eng_state_t of include/parrot_hip.h)
ENG_GEMV, ENG_ATTN = 0, 1
ENG_EPI_LOGITS = 4
ENG_WGS = 256
ENG_W_E4, ENG_W_E16, ENG_W_E8 = 0, 1, 2
ENG_W_TWO_LOADERS = 4


class EngOp(C.Structure):  # parrot_eng_op_t
    _fields_ = (
        [(n, C.c_int32) for n in ("type", "epilogue", "K", "nblocks", "nq", "buf", "norm_kind")]
        + [("norm_eps", C.c_float)]
        + [(n, C.c_int32) for n in ("in_embedding", "res_embedding", "wfmt", "res_in", "res_out", "publish", "no_gather", "blk_part", "blk_parts", "acc")]
        + [("threshold", C.c_float), ("reserved", C.c_int32)]
        + [(n, C.c_void_p) for n in ("W", "norm_w", "norm_b", "bias", "norm2_w", "norm2_b", "inp", "out", "part", "k_cache", "v_cache")]
    )


class EngState(C.Structure):  # parrot_eng_state_t
    _fields_ = (
        [("ops", C.c_void_p), ("nops", C.c_int32), ("d", C.c_int32)]
        + [(n, C.c_void_p) for n in ("tokens", "pos", "epoch", "err", "wte", "rope_cos", "rope_sin")]
        + [(n, C.c_int32) for n in ("n_elem", "n_groups", "q_per_kv", "hs", "S", "V", "rsqrt_mode", "nsplit", "greedy",
                                    "lds_buf0_bytes", "lds_buf1_bytes", "kmax", "wfmt", "attn_buf", "vper")]
        + [(n, C.c_void_p) for n in ("arg", "dbg", "dbg_all", "host_words")]
    )


# name -> (restype, argtypes); must list every function include/parrot_hip.h declares
SIGNATURES = {
    "parrot_version": (_i, []),
    "parrot_last_error": (C.c_char_p, []),
    "parrot_prof_begin": (_i, []),
    "parrot_prof_end": (_i, [_i, _vp, _vp, _vp]),
    "parrot_kernel_name": (C.c_char_p, [_i]),
    "parrot_w4_packed_bytes": (_i64, [_i, _i, _i]),
    "parrot_w4_repack": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "parrot_w4_gemv": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _np, _vp]),
    "parrot_w4c_gemv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _np, _vp]),
    "parrot_w4c_gemm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _np, _vp, _vp]),
    "parrot_w4c_dequant": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "parrot_gemm_workspace_floats": (_i64, [_i, _i, _i, _i, _i]),
    "parrot_w4_gemm": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _np, _vp, _vp]),
    "parrot_bf16_gemv": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _np, _vp]),
    "parrot_bf16_gemm": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _np, _vp, _vp]),
    "parrot_w8_quantize_rows": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "parrot_w8_prep_act": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _np, _vp]),
    "parrot_w8_gemv": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "parrot_w8_gemm_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "parrot_w8_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "parrot_w8_gemv_fused": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _np, _vp]),
    "parrot_rmsnorm": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "parrot_layernorm": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "parrot_qkv_rope_kvappend": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "parrot_attn_workspace_floats": (_i64, [_i, _i, _i, _i]),
    "parrot_attn_decode": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "parrot_attn_fused_decode": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "parrot_attn_prefill_scratch_elems": (_i64, [_i, _i, _i]),
    "parrot_attn_prefill": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "parrot_e4_bytes": (_i64, [_i, _i, _i]),
    "parrot_e4_repack": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "parrot_e16_bytes": (_i64, [_i, _i, _i]),
    "parrot_e16_repack": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "parrot_e8_bytes": (_i64, [_i, _i, _i]),
    "parrot_e8_repack": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "parrot_eng_lds_bytes_e8": (_i64, [_i, _i, _i, _i]),
    "parrot_eng_lds_bytes": (_i64, [_i, _i, _i, _i]),
    "parrot_eng_lds_total": (_i64, [_i, _i, _i, _i]),
    "parrot_eng_step": (_i, [C.POINTER(EngState), _vp]),
    "parrot_embedding": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "parrot_gptq_block": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "parrot_stop_check": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "parrot_argmax_advance": (_i, [_vp, _i, _vp, _vp, _vp]),
    "parrot_topk_sample": (_i, [_vp, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (once) and declare the signatures.  Fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ParrotHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the HIP path)"
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise ParrotHipError(f"{LIB_PATH} does not export {name}: rebuild the extension")
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().parrot_last_error().decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise ParrotHipError(f"{what} failed ({rc}): {last_error()}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL).  Non-CUDA tensors are refused: the HIP path has no CPU twin."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ParrotHipError("the HIP path only accepts tensors on a GPU (cuda/HIP device)")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------------------------- profiling sink
def prof_begin() -> None:
    check(load().parrot_prof_begin(), "parrot_prof_begin")


def prof_end() -> dict:
    """Returns {kernel_name: (total_ms, launches)} for the launches since prof_begin()."""
    lib = load()
    cap = 64
    ids = (C.c_int * cap)()
    ms = (C.c_double * cap)()
    cnt = (C.c_int64 * cap)()
    n = lib.parrot_prof_end(cap, C.cast(ids, C.c_void_p), C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p))
    if n < 0:
        raise ParrotHipError(f"parrot_prof_end failed ({n}): {last_error()}")
    return {lib.parrot_kernel_name(ids[i]).decode(): (ms[i], cnt[i]) for i in range(min(n, cap))}

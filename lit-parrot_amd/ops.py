"""Tensor-level wrappers over the C ABI (``_hip.py``): shape checks, pointer extraction, dispatch per Linear class.

Everything here enqueues HIP kernels on the current torch stream and returns immediately; nothing falls back to a
torch implementation.  Activations are bf16 rows ``(M, features)`` of ONE sequence.
"""
import ctypes as C
import os
from typing import Optional

import torch

from . import _hip
from ._hip import EPI_GELU, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU, ParrotHipError, check, ptr, stream

GEMV_MAX_ROWS = 8  # up to this many rows the weight-streaming GEMV kernels are used, above it the GEMM entry points


def _rows(x: torch.Tensor, what: str) -> torch.Tensor:
    if x.dtype != torch.bfloat16:
        raise ParrotHipError(f"{what}: the HIP path computes in bf16 (got {x.dtype}); convert the model with .to(torch.bfloat16)")
    if x.dim() != 2 or x.stride(1) != 1:
        raise ParrotHipError(f"{what}: expected contiguous rows (M, features), got shape {tuple(x.shape)} strides {x.stride()}")
    return x


def _opt_vec(t: Optional[torch.Tensor], n: int, what: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.bfloat16 or t.numel() != n or not t.is_contiguous():
        raise ParrotHipError(f"{what}: expected a contiguous bf16 vector of {n} elements")
    return t


# ------------------------------------------------------------------------------------------------ norms
# 0: rsqrt rounded once (torch GPU semantics, the default); 1: torch's CPU scalar-path double rounding.  Tests flip this to
# compare bit for bit against golden vectors that the reference produced on a CPU.
RMSNORM_RSQRT_MODE = 0


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: torch.Tensor) -> torch.Tensor:
    _rows(x, "rmsnorm"), _rows(out, "rmsnorm")
    M, d = x.shape
    check(_hip.load().parrot_rmsnorm(ptr(x), x.stride(0), ptr(_opt_vec(weight, d, "rmsnorm weight")), ptr(out),
                                     out.stride(0), M, d, float(eps), RMSNORM_RSQRT_MODE, stream()), "parrot_rmsnorm")
    return out


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float,
              out: torch.Tensor) -> torch.Tensor:
    _rows(x, "layernorm"), _rows(out, "layernorm")
    M, d = x.shape
    check(_hip.load().parrot_layernorm(ptr(x), x.stride(0), ptr(_opt_vec(weight, d, "layernorm weight")),
                                       ptr(_opt_vec(bias, d, "layernorm bias")), ptr(out), out.stride(0), M, d,
                                       float(eps), stream()), "parrot_layernorm")
    return out


class Norm:
    """A norm to fuse in front of a Linear: kind 1 = RMSNorm, 2 = LayerNorm (``parrot_norm_t``)."""

    __slots__ = ("kind", "weight", "bias", "eps")

    def __init__(self, kind: int, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float) -> None:
        self.kind, self.weight, self.bias, self.eps = kind, weight, bias, float(eps)

    def c_struct(self, K: int):
        w = _opt_vec(self.weight, K, "fused norm weight")
        b = _opt_vec(self.bias, K, "fused norm bias")
        return _hip.ParrotNorm(self.kind, ptr(w), ptr(b), self.eps, RMSNORM_RSQRT_MODE)


def _prenorm(x: torch.Tensor, norm: Optional["Norm"]):
    """Prefill (more rows than the GEMV kernels take): apply the norm with the stand-alone kernel first."""
    if norm is None or (x.shape[0] <= GEMV_MAX_ROWS and x.shape[0] * x.shape[1] * 2 <= 60000):
        return x, norm  # the fused prologue keeps the normalised rows in LDS (<= 64 KB)
    xn = torch.empty_like(x)
    if norm.kind == 1:
        rmsnorm(x, norm.weight, norm.eps, xn)
    else:
        layernorm(x, norm.weight, norm.bias, norm.eps, xn)
    return xn, None


def _norm_arg(norm: Optional[Norm], K: int):
    return None if norm is None else C.byref(norm.c_struct(K))


# ------------------------------------------------------------------------------------------------ linears
def bf16_linear(weight: torch.Tensor, x: torch.Tensor, out: torch.Tensor, *, bias=None, epilogue=EPI_NONE,
                residual=None, weight2=None, norm: Optional[Norm] = None) -> torch.Tensor:
    _rows(x, "bf16_linear"), _rows(out, "bf16_linear")
    x, norm = _prenorm(x, norm)
    N, K = weight.shape
    if weight.dtype != torch.bfloat16 or not weight.is_contiguous():
        raise ParrotHipError("bf16_linear: weight must be contiguous bf16 (out_features, in_features)")
    M = x.shape[0]
    lib = _hip.load()
    args = (ptr(weight), ptr(weight2), ptr(x), x.stride(0), M, ptr(_opt_vec(bias, N, "bias")),
            ptr(residual), residual.stride(0) if residual is not None else 0, ptr(out), out.stride(0), N, K,
            epilogue, _norm_arg(norm, K))
    if M <= GEMV_MAX_ROWS:
        check(lib.parrot_bf16_gemv(*args, stream()), "parrot_bf16_gemv")
    else:  # prefill: MFMA kernel; short prompts split K and need room for the partial results
        n = lib.parrot_gemm_workspace_floats(M, N, K, 0, epilogue)
        ws = torch.empty((n,), dtype=torch.float32, device=x.device) if n > 0 else None
        check(lib.parrot_bf16_gemm(*args, ptr(ws), stream()), "parrot_bf16_gemm")
    return out


def w4_packed_bytes(N: int, K: int, group: int) -> int:
    n = _hip.load().parrot_w4_packed_bytes(N, K, group)
    if n < 0:
        raise ParrotHipError(f"parrot_w4_packed_bytes({N}, {K}, {group}) failed: {_hip.last_error()}")
    return n


def w4_repack(quant_weight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, N: int, K: int, group: int,
              packed: torch.Tensor, direction: int) -> None:
    """direction 0: reference buffers -> W4K ``packed``; 1: the inverse (writes the reference buffers)."""
    if quant_weight.dtype != torch.uint8 or quant_weight.shape != (N, K // 2) or quant_weight.stride() != (1, N):
        raise ParrotHipError("w4_repack: quant_weight must be uint8 (N, K/2) with strides (1, N) (quantize/gptq.py:216-222)")
    for t, nm in ((scales, "scales"), (zeros, "zeros")):
        if t.dtype != torch.bfloat16 or not t.is_contiguous() or t.shape[0] != N:
            raise ParrotHipError(f"w4_repack: {nm} must be contiguous bf16 (N, groups)")
    if packed.dtype != torch.uint8 or packed.numel() != w4_packed_bytes(N, K, group):
        raise ParrotHipError("w4_repack: packed buffer has the wrong size")
    check(_hip.load().parrot_w4_repack(ptr(quant_weight), ptr(scales), ptr(zeros), N, K, group, ptr(packed), direction,
                                       stream()), "parrot_w4_repack")


W4_GEMM_FUSED_NORM = True  # (False: the stand-alone norm kernels in front of the prompt GEMM - tests compare the two)


def w4_linear(packed: torch.Tensor, N: int, K: int, group: int, x: torch.Tensor, out: torch.Tensor, *, bias=None,
              epilogue=EPI_NONE, residual=None, packed2=None, norm: Optional[Norm] = None) -> torch.Tensor:
    _rows(x, "w4_linear"), _rows(out, "w4_linear")
    # prompts on groups of 64 / 128: parrot_w4_gemm applies the norm itself - one launch writes the normalised rows AND the per-group
    # sums the int4 kernel folds its zero points with (instead of parrot_rmsnorm + the activation-sum pre-pass)
    fused = (norm is not None and W4_GEMM_FUSED_NORM and x.shape[0] > GEMV_MAX_ROWS and group in (64, 128) and K % group == 0
             and K <= 16384 and x.stride(0) % 8 == 0)
    if not fused:
        x, norm = _prenorm(x, norm)
    M = x.shape[0]
    if x.shape[1] != K or out.shape[1] != N:
        raise ParrotHipError(f"w4_linear: x {tuple(x.shape)} / out {tuple(out.shape)} do not match N={N} K={K}")
    lib = _hip.load()
    args = (ptr(packed), ptr(packed2), ptr(x), x.stride(0), M, ptr(_opt_vec(bias, N, "bias")), ptr(residual),
            residual.stride(0) if residual is not None else 0, ptr(out), out.stride(0), N, K, group, epilogue,
            _norm_arg(norm, K))
    if M <= GEMV_MAX_ROWS:
        check(lib.parrot_w4_gemv(*args, stream()), "parrot_w4_gemv")
    else:  # prefill: MFMA kernel; needs the per-group activation sums workspace
        ws = torch.empty((max(1, lib.parrot_gemm_workspace_floats(M, N, K, group if group > 0 else -1, epilogue)),),
                         dtype=torch.float32, device=x.device)
        check(lib.parrot_w4_gemm(*args, ptr(ws), stream()), "parrot_w4_gemm")
    return out


def w4c_dequant(packed: torch.Tensor, code_f32: torch.Tensor, N: int, K: int, block: int, out: torch.Tensor) -> torch.Tensor:
    """Codebook (NF4 / FP4) W4K records -> dense bf16 (N, K) = bf16(code[q] * absmax), bitsandbytes' dequantize_4bit."""
    if out.dtype != torch.bfloat16 or out.shape != (N, K) or out.stride(1) != 1:
        raise ParrotHipError("w4c_dequant: out must be bf16 (N, K) with unit column stride")
    if code_f32.dtype != torch.float32 or code_f32.numel() != 16:
        raise ParrotHipError("w4c_dequant: the codebook is 16 fp32 values")
    check(_hip.load().parrot_w4c_dequant(ptr(packed), ptr(code_f32), ptr(out), out.stride(0), N, K, block, stream()), "parrot_w4c_dequant")
    return out


W4C_PREFILL_FUSED = True  # prompts: the codebook GEMM (GEMV numerics); False = dequantise + dense GEMM (bitsandbytes' order)


def w4c_linear(packed: torch.Tensor, code_words: torch.Tensor, code_f32: torch.Tensor, N: int, K: int, block: int, x: torch.Tensor,
               out: torch.Tensor, *, bias=None, epilogue=EPI_NONE, residual=None, packed2=None, norm: Optional[Norm] = None) -> torch.Tensor:
    """Linear over 4-bit codebook weights.  Up to GEMV_MAX_ROWS rows: the fused dequant-into-GEMV kernel.  More rows
    (prefill): the codebook GEMM on the matrix cores (same numerics as the GEMV), or - W4C_PREFILL_FUSED = False, K not a multiple
    of 64 - dequantise to bf16 and multiply, the order of operations of bitsandbytes' MatMul4Bit."""
    _rows(x, "w4c_linear"), _rows(out, "w4c_linear")
    M = x.shape[0]
    if x.shape[1] != K or out.shape[1] != N:
        raise ParrotHipError(f"w4c_linear: x {tuple(x.shape)} / out {tuple(out.shape)} do not match N={N} K={K}")
    if code_words.dtype != torch.int32 or code_words.numel() != 16:
        raise ParrotHipError("w4c_linear: the bf16 codebook is passed as 16 int32 words")
    x, norm = _prenorm(x, norm)
    if M > GEMV_MAX_ROWS:
        lib = _hip.load()
        if K % 64 == 0 and W4C_PREFILL_FUSED:
            ws = torch.empty((max(1, lib.parrot_gemm_workspace_floats(M, N, K, block, epilogue)),), dtype=torch.float32, device=x.device)
            check(lib.parrot_w4c_gemm(ptr(packed), ptr(packed2), ptr(code_words), ptr(x), x.stride(0), M, ptr(_opt_vec(bias, N, "bias")),
                                      ptr(residual), residual.stride(0) if residual is not None else 0, ptr(out), out.stride(0), N, K,
                                      block, epilogue, None, ptr(ws), stream()), "parrot_w4c_gemm")
            return out
        # bitsandbytes' own order of operations: dequantise to bf16, then a dense GEMM
        dense = w4c_dequant(packed, code_f32, N, K, block, torch.empty((N, K), dtype=torch.bfloat16, device=x.device))
        dense2 = None
        if packed2 is not None:
            dense2 = w4c_dequant(packed2, code_f32, N, K, block, torch.empty((N, K), dtype=torch.bfloat16, device=x.device))
        return bf16_linear(dense, x, out, bias=bias, epilogue=epilogue, residual=residual, weight2=dense2)
    check(_hip.load().parrot_w4c_gemv(ptr(packed), ptr(packed2), ptr(code_words), ptr(x), x.stride(0), M,
                                      ptr(_opt_vec(bias, N, "bias")), ptr(residual), residual.stride(0) if residual is not None else 0,
                                      ptr(out), out.stride(0), N, K, block, epilogue, _norm_arg(norm, K), stream()), "parrot_w4c_gemv")
    return out


def w8_quantize_rows(weight: torch.Tensor, CB: torch.Tensor, SCB: torch.Tensor) -> None:
    N, K = weight.shape
    code = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}.get(weight.dtype)
    if code is None or not weight.is_contiguous():
        raise ParrotHipError("w8_quantize_rows: weight must be contiguous bf16, fp16 or fp32")
    check(_hip.load().parrot_w8_quantize_rows(ptr(weight), code, N, K, ptr(CB), ptr(SCB), stream()), "parrot_w8_quantize_rows")


class W8Act:
    """Quantised activation rows of one LLM.int8 Linear input (buffers are reused across calls)."""

    def __init__(self, M: int, K: int, device) -> None:
        self.M, self.K = M, K
        self.xq = torch.empty((M, K), dtype=torch.int8, device=device)
        self.xout = torch.empty((M, K), dtype=torch.float32, device=device)
        self.sca = torch.empty((M,), dtype=torch.float32, device=device)
        self.nout = torch.empty((M,), dtype=torch.int32, device=device)
        self.oidx = torch.empty((M, K), dtype=torch.int32, device=device)  # compact outlier column list per row
        self.colflag = torch.empty((K,), dtype=torch.int32, device=device) if M > 1 else None  # outlier columns of the call


def w8_prep_act(x: torch.Tensor, threshold: float, act: W8Act, norm: Optional[Norm] = None) -> W8Act:
    _rows(x, "w8_prep_act")
    M, K = x.shape
    assert (M, K) == (act.M, act.K)
    check(_hip.load().parrot_w8_prep_act(ptr(x), x.stride(0), M, K, float(threshold), ptr(act.xq), ptr(act.xout),
                                         ptr(act.sca), ptr(act.nout), ptr(act.oidx), ptr(act.colflag), _norm_arg(norm, K), stream()), "parrot_w8_prep_act")
    return act


W8_PREFILL_GEMM2 = os.environ.get("PARROT_W8_GEMM2", "1") != "0"  # False: the first-generation int8 GEMM inside parrot_w8_gemv (A/B, tests)


def w8_linear(CB: torch.Tensor, SCB: torch.Tensor, N: int, K: int, act: W8Act, out: torch.Tensor, *, bias=None,
              epilogue=EPI_NONE, residual=None) -> torch.Tensor:
    _rows(out, "w8_linear")
    lib = _hip.load()
    nws = int(lib.parrot_w8_gemm_workspace_bytes(act.M, N, K, epilogue)) if W8_PREFILL_GEMM2 else 0
    if nws > 0:  # prompts: int8 MFMA GEMM on the LDS-DMA structure + element-wise dequantise / outlier / epilogue pass
        ws = torch.empty((nws,), dtype=torch.uint8, device=out.device)
        check(lib.parrot_w8_gemm(ptr(CB), ptr(SCB), ptr(act.xq), ptr(act.xout), ptr(act.sca), ptr(act.nout), ptr(act.oidx), act.M,
                                 ptr(_opt_vec(bias, N, "bias")), ptr(residual), residual.stride(0) if residual is not None else 0,
                                 ptr(out), out.stride(0), N, K, epilogue, ptr(ws), stream()), "parrot_w8_gemm")
        return out
    check(_hip.load().parrot_w8_gemv(ptr(CB), ptr(SCB), ptr(act.xq), ptr(act.xout), ptr(act.sca), ptr(act.nout), ptr(act.oidx), act.M,
                                     ptr(_opt_vec(bias, N, "bias")), ptr(residual),
                                     residual.stride(0) if residual is not None else 0, ptr(out), out.stride(0), N, K,
                                     epilogue, stream()), "parrot_w8_gemv")
    return out


W8_FUSED_MAX_K = 16384


def w8_linear_fused(CB: torch.Tensor, SCB: torch.Tensor, N: int, K: int, x: torch.Tensor, threshold: float, out: torch.Tensor,
                    *, bias=None, epilogue=EPI_NONE, residual=None, norm: Optional[Norm] = None) -> torch.Tensor:
    """One token row through an LLM.int8 Linear in one launch (activation quantiser + GEMV + epilogue)."""
    _rows(x, "w8_linear_fused"), _rows(out, "w8_linear_fused")
    if x.shape != (1, K) or out.shape != (1, N) or not x.is_contiguous():
        raise ParrotHipError(f"w8_linear_fused: one contiguous row expected, got x {tuple(x.shape)} out {tuple(out.shape)}")
    check(_hip.load().parrot_w8_gemv_fused(ptr(CB), ptr(SCB), ptr(x), float(threshold), ptr(_opt_vec(bias, N, "bias")),
                                           ptr(residual), ptr(out), N, K, epilogue, _norm_arg(norm, K), stream()),
          "parrot_w8_gemv_fused")
    return out


# ------------------------------------------------------------------------------------------------ attention
def rope_kvappend(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, n_elem: int, pos: torch.Tensor,
                  n_groups: int, q_per_kv: int, hs: int, S: int, q_out: torch.Tensor, k_cache: torch.Tensor,
                  v_cache: torch.Tensor, rope_local: bool = False) -> None:
    _rows(qkv, "rope_kvappend")
    if cos.dtype != torch.float16 or sin.dtype != torch.float16 or not cos.is_contiguous() or not sin.is_contiguous():
        raise ParrotHipError("rope_kvappend: the RoPE tables must be contiguous fp16 (lit_gpt/model.py:325-326)")
    if pos.dtype != torch.int32 or k_cache.dtype != torch.bfloat16 or v_cache.dtype != torch.bfloat16:
        raise ParrotHipError("rope_kvappend: pos must be int32 and the caches bf16")
    if k_cache.numel() != n_groups * S * hs or v_cache.numel() != n_groups * S * hs or not k_cache.is_contiguous():
        raise ParrotHipError("rope_kvappend: cache shape must be (n_groups, S, hs) contiguous")
    check(_hip.load().parrot_qkv_rope_kvappend(ptr(qkv), qkv.stride(0), qkv.shape[0], ptr(cos), ptr(sin), n_elem,
                                               int(rope_local), ptr(pos), n_groups, q_per_kv, hs, S, ptr(q_out), ptr(k_cache),
                                               ptr(v_cache), stream()), "parrot_qkv_rope_kvappend")


# 0: softmax probabilities in fp32 (default); 1: rounded to bf16 before P.V against the key block's maximum, as torch's CPU
# flash-attention kernel computes the bf16 reference (parity runs: tests / tools/parity_table.py flip it);
# windows up to 512 slots, the multi-launch step only (the engine's attention keeps fp32 probabilities)
ATTN_SOFTMAX_MODE = 0
FUSED_ATTN_MAX_Q_PER_KV = 16  # query heads per group the fused single-row kernel is built for
ATTN_SPLIT_KEYS = 1024  # window slots per sequence split of the decode attention (bench.py --attn-split-keys: A/B)


def attn_nsplit(n_groups: int, S: int, q_per_kv: int = 1, M: int = 1) -> int:
    """Sequence splits of the decode-attention kernels: one workgroup walks up to ~1k keys by itself, so windows up to
    1024 take no split at all (no partials / ticket / second pass) and longer ones ceil(S / 1024) splits.  The same rule
    serves the multi-row (prefill) kernel: with one workgroup per (row, head, 32 keys) a 512-token prompt launched 393 K
    mostly empty workgroups per layer and was bound by workgroup dispatch (315 us per layer; 60 us without the split)."""
    chunks = (q_per_kv + 3) // 4 if q_per_kv > 2 else 1
    cap = max(1, 1024 // (n_groups * chunks))
    n = max(1, min(-(-S // ATTN_SPLIT_KEYS), cap, 64))
    if M == 1 and q_per_kv > FUSED_ATTN_MAX_Q_PER_KV:
        # a single row of an MQA model with many query heads (Falcon-7B: 71 on one K/V head) takes the unfused kernels with one
        # workgroup per (group, 4 heads, split): 18 workgroups on 256 CUs took 108 us per layer; aim at ~190, >= 64 keys per split
        n = max(n, min(-(-192 // (n_groups * chunks)), max(1, S // 64), cap, 64))
    return n


def attn_workspace(M: int, n_head: int, hs: int, nsplit: int, device) -> torch.Tensor:
    n = _hip.load().parrot_attn_workspace_floats(M, n_head, hs, nsplit)
    return torch.empty((max(int(n), 1),), dtype=torch.float32, device=device)


def attn_decode(q: torch.Tensor, pos: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, n_groups: int,
                q_per_kv: int, hs: int, S: int, nsplit: int, workspace: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    _rows(y, "attn_decode")
    M = y.shape[0]
    check(_hip.load().parrot_attn_decode(ptr(q), M, ptr(pos), ptr(k_cache), ptr(v_cache), n_groups, q_per_kv, hs, S,
                                         nsplit, ptr(workspace), ptr(y), y.stride(0), ATTN_SOFTMAX_MODE, stream()), "parrot_attn_decode")
    return y


def attn_prefill(q: torch.Tensor, pos: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, n_groups: int, q_per_kv: int,
                 hs: int, S: int, y: torch.Tensor, scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Causal attention of the M prompt rows on the matrix cores (*pos + M <= S: the caller knows there is no ring wrap)."""
    _rows(q, "attn_prefill"), _rows(y, "attn_prefill")
    M = y.shape[0]
    lib = _hip.load()
    n = int(lib.parrot_attn_prefill_scratch_elems(n_groups, hs, S))
    if scratch is None or scratch.numel() < n or scratch.dtype != torch.bfloat16:
        scratch = torch.empty((n,), dtype=torch.bfloat16, device=y.device)
    check(lib.parrot_attn_prefill(ptr(q), M, ptr(pos), ptr(k_cache), ptr(v_cache), ptr(scratch), n_groups, q_per_kv, hs, S, ptr(y),
                                  y.stride(0), stream()), "parrot_attn_prefill")
    return y


# prompts from this many rows take the MFMA kernel when the caller rules out a ring wrap (measured on Llama-2-7B, 128 rows: row-by-row
# kernels 0.38 ms, flash kernel 0.56 ms without / see DESIGN.md 4c with LDS sharing; StableLM-3B, 512 rows: 1.9 vs 0.54 ms)
ATTN_PREFILL_MIN_ROWS = int(os.environ.get("PARROT_ATTN_PREFILL_MIN_ROWS", "160"))


def attn_fused_decode(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, n_elem: int, pos: torch.Tensor,
                      k_cache: torch.Tensor, v_cache: torch.Tensor, n_groups: int, q_per_kv: int, hs: int, S: int,
                      nsplit: int, workspace: torch.Tensor, tickets: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Single new token: split + RoPE + KV append + attention + combine in one launch."""
    _rows(qkv, "attn_fused_decode"), _rows(y, "attn_fused_decode")
    if qkv.shape[0] != 1 or tickets.dtype != torch.int32 or tickets.numel() < n_groups * q_per_kv:
        raise ParrotHipError("attn_fused_decode: one row, int32 tickets[n_groups] expected")
    lib = _hip.load()
    check(lib.parrot_attn_fused_decode(ptr(qkv), ptr(cos), ptr(sin), n_elem, ptr(pos), n_groups, q_per_kv, hs, S,
                                       nsplit, ptr(workspace), ptr(tickets), ptr(k_cache), ptr(v_cache), ptr(y),
                                       ATTN_SOFTMAX_MODE, stream()), "parrot_attn_fused_decode")
    return y


# ------------------------------------------------------------------------------------------------ step glue
def embedding(wte: torch.Tensor, tokens: torch.Tensor, pos: Optional[torch.Tensor], M: int, out: torch.Tensor) -> torch.Tensor:
    _rows(out, "embedding")
    if wte.dtype != torch.bfloat16 or not wte.is_contiguous() or tokens.dtype != torch.int64:
        raise ParrotHipError("embedding: wte must be contiguous bf16 and tokens int64")
    check(_hip.load().parrot_embedding(ptr(wte), wte.shape[1], ptr(tokens), ptr(pos), M, ptr(out), out.stride(0),
                                       stream()), "parrot_embedding")
    return out


def argmax_advance(logits: torch.Tensor, tokens: torch.Tensor, pos: torch.Tensor) -> None:
    if logits.dtype != torch.bfloat16 or tokens.dtype != torch.int64 or pos.dtype != torch.int32:
        raise ParrotHipError("argmax_advance: logits bf16, tokens int64, pos int32 expected")
    check(_hip.load().parrot_argmax_advance(ptr(logits), logits.numel(), ptr(tokens), ptr(pos), stream()),
          "parrot_argmax_advance")


def topk_sample(logits: torch.Tensor, temperature: float, top_k: Optional[int], noise: torch.Tensor, tokens: torch.Tensor,
                pos: torch.Tensor, probs_out: Optional[torch.Tensor] = None) -> None:
    """The reference's sampling step (generate/base.py:136-153) in one launch: ``tokens[pos + 1]`` = the token that
    ``torch.multinomial(softmax(topk-cropped(logits / temperature)), 1)`` draws when its internal noise is ``noise`` -
    a bf16 (V,) buffer the caller fills with ``noise.exponential_(1)`` right before (the call torch.multinomial makes itself,
    so a torch seed gives the tokens the reference's ops give on this device); ``pos += 1``."""
    V = logits.numel()
    if logits.dtype != torch.bfloat16 or noise.dtype != torch.bfloat16 or noise.numel() != V or tokens.dtype != torch.int64 or pos.dtype != torch.int32:
        raise ParrotHipError("topk_sample: logits / noise bf16 (V,), tokens int64, pos int32 expected")
    if probs_out is not None and (probs_out.dtype != torch.bfloat16 or probs_out.numel() != V):
        raise ParrotHipError("topk_sample: probs_out must be bf16 (V,)")
    if not temperature > 0:
        raise ParrotHipError(f"topk_sample: temperature must be positive (got {temperature})")
    check(_hip.load().parrot_topk_sample(ptr(logits), V, float(temperature), int(top_k) if top_k is not None else 0, ptr(noise),
                                         ptr(probs_out), ptr(tokens), ptr(pos), stream()), "parrot_topk_sample")


def stop_check(tokens: torch.Tensor, pos: torch.Tensor, first_gen: torch.Tensor, stop_flat: torch.Tensor,
               stop_off: torch.Tensor, n_stop: int, longest: int, flag: torch.Tensor) -> None:
    """Latch in ``flag`` (int32[2]) the first stop sequence that the generated tokens end with (chat loop)."""
    if tokens.dtype != torch.int64 or stop_flat.dtype != torch.int64 or any(t.dtype != torch.int32 for t in (pos, first_gen, stop_off, flag)):
        raise ParrotHipError("stop_check: tokens / stop_flat int64; pos, first_gen, stop_off, flag int32")
    check(_hip.load().parrot_stop_check(ptr(tokens), ptr(pos), ptr(first_gen), ptr(stop_flat), ptr(stop_off), n_stop, longest,
                                        ptr(flag), stream()), "parrot_stop_check")

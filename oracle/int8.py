"""LLM.int8 restated on the CPU (TEST INFRASTRUCTURE ONLY) — **parity unpinned**.

The reference reaches this arithmetic through bitsandbytes (``bnb.functional.double_quant`` at quantize/bnb.py:55 and
the inherited ``bnb.nn.Linear8bitLt.forward`` -> ``MatMul8bitLt``), a third-party wheel (``bitsandbytes>=0.40.0``,
requirements.txt:5, no lock file) that is not under /root/reference and not installed here, and the reference's tests
never touch it.  This file follows the published algorithm (Dettmers et al., LLM.int8(), arXiv:2208.07339) as
configured at quantize/bnb.py:26-33 (has_fp16_weights=False, threshold=6.0):
  * weights: row-wise absmax int8, CB = rint(127 * W16 / absmax), SCB = absmax;
  * activations per token row: cast to fp16; entries with |a| >= threshold are outliers: zero in the int8 copy and
    excluded from the row absmax; CA = rint(127 * a / absmax), SCA = absmax;
  * C32 = CA @ CB^T in int32; out16 = fp16(C32 * (1/127^2) * SCA * SCB + bias);
  * outlier columns: out16 = fp16(out16 + fp16(A[:, idx] @ fp16(CB[:, idx] * SCB / 127)^T)) accumulated in fp32;
  * cast back to the input dtype.
"""
from typing import Optional, Tuple

import torch

MM_DEQUANT = 6.200012e-05  # 1 / (127 * 127) as the float constant


def quantize_weight_rows(weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    w = weight.half().float()
    absmax = w.abs().amax(dim=1)
    inv = torch.where(absmax > 0, torch.full_like(absmax, 127.0) / absmax, torch.zeros_like(absmax))  # IEEE division
    CB = torch.round(w * inv[:, None]).to(torch.int8)  # rint: half to even
    return CB, absmax


def quantize_act_rows(x: torch.Tensor, threshold: float):
    a = x.half().float()
    outlier = (a.abs() >= threshold) if threshold > 0 else torch.zeros_like(a, dtype=torch.bool)
    kept = torch.where(outlier, torch.zeros_like(a), a)
    absmax = kept.abs().amax(dim=-1)
    inv = torch.where(absmax > 0, torch.full_like(absmax, 127.0) / absmax, torch.zeros_like(absmax))  # (scalar / tensor is reciprocal * scalar in torch)
    CA = torch.round(kept * inv[..., None]).to(torch.int8)
    return CA, absmax, torch.where(outlier, a, torch.zeros_like(a))


def linear(x: torch.Tensor, CB: torch.Tensor, SCB: torch.Tensor, bias: Optional[torch.Tensor], threshold: float = 6.0) -> torch.Tensor:
    shape = x.shape
    rows = x.reshape(-1, shape[-1])
    CA, SCA, xout = quantize_act_rows(rows, threshold)
    C32 = CA.to(torch.int32) @ CB.to(torch.int32).t()  # exact integer accumulate
    v = C32.float() * MM_DEQUANT * SCA[:, None] * SCB[None, :]
    if bias is not None:
        v = v + bias.float()[None, :]
    out16 = v.half()
    has_out = (xout != 0).any(dim=-1)
    if bool(has_out.any()):
        subB = (CB.float() * SCB[:, None] / 127.0).half().float()  # (N, K) dequantised weights in fp16
        add = (xout @ subB.t()).half()  # only the outlier columns are non-zero in xout
        both = (out16.float() + add.float()).half()
        out16 = torch.where(has_out[:, None], both, out16)
    return out16.to(x.dtype).reshape(*shape[:-1], CB.shape[0])

"""LLM.int8 restated on the CPU (TEST INFRASTRUCTURE ONLY) — **parity unpinned**.

The reference reaches this arithmetic through bitsandbytes (``bnb.functional.double_quant`` at quantize/bnb.py:55 and
the inherited ``bnb.nn.Linear8bitLt.forward`` -> ``MatMul8bitLt``), a third-party wheel (``bitsandbytes>=0.40.0``,
requirements.txt:5, no lock file) that is not under /root/reference and not installed here, and the reference's tests
never touch it.  This file follows the published algorithm (Dettmers et al., LLM.int8(), arXiv:2208.07339) as
configured at quantize/bnb.py:26-33 (has_fp16_weights=False, threshold=6.0):
  * weights: row-wise absmax int8, CB = rint(127 * W16 / absmax), SCB = absmax;
  * activations of one call (all its token rows): cast to fp16; entries with |a| >= threshold are outliers: zero in the
    int8 copy and excluded from their row's absmax (double_quant with a threshold); CA = rint(127 * a / absmax),
    SCA = absmax per row;
  * the outlier FEATURE DIMENSIONS are the columns holding at least one outlier in ANY row of the call
    (paper §3.2; MatMul8bitLt: idx = unique(coo_tensorA.colidx)): CA[:, idx] = 0 for EVERY row;
  * C32 = CA @ CB^T in int32; out16 = fp16(C32 * (1/127^2) * SCA * SCB + bias);
  * mixed-precision part over those columns, every row: out16 = fp16(out16 + fp16(A[:, idx] @ fp16(CB[:, idx] * SCB / 127)^T))
    accumulated in fp32 (A[:, idx] holds the fp16 activations of all rows in those columns, outliers or not);
  * cast back to the input dtype.
For a single row (decode, the BASELINE configuration) the column rule and a per-row rule coincide.
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
    """(CA, SCA, subA): int8 rows with the outlier columns of the CALL zeroed, row scales, and the fp16 activations of every
    row in those columns (zero elsewhere)."""
    a = x.half().float()
    outlier = (a.abs() >= threshold) if threshold > 0 else torch.zeros_like(a, dtype=torch.bool)
    kept = torch.where(outlier, torch.zeros_like(a), a)
    absmax = kept.abs().amax(dim=-1)  # a row's own outliers are excluded; entries below the threshold in outlier columns count
    inv = torch.where(absmax > 0, torch.full_like(absmax, 127.0) / absmax, torch.zeros_like(absmax))  # (scalar / tensor is reciprocal * scalar in torch)
    cols = outlier.reshape(-1, a.shape[-1]).any(dim=0)  # outlier feature dimensions of the whole call
    CA = torch.round(kept * inv[..., None]).to(torch.int8)
    CA[..., cols] = 0
    return CA, absmax, torch.where(cols, a, torch.zeros_like(a))


def linear(x: torch.Tensor, CB: torch.Tensor, SCB: torch.Tensor, bias: Optional[torch.Tensor], threshold: float = 6.0) -> torch.Tensor:
    shape = x.shape
    rows = x.reshape(-1, shape[-1])
    CA, SCA, subA = quantize_act_rows(rows, threshold)
    C32 = CA.to(torch.int32) @ CB.to(torch.int32).t()  # exact integer accumulate
    v = C32.float() * MM_DEQUANT * SCA[:, None] * SCB[None, :]
    if bias is not None:
        v = v + bias.float()[None, :]
    out16 = v.half()
    cols = (subA != 0).any(dim=0) if threshold > 0 else torch.zeros(rows.shape[-1], dtype=torch.bool)
    cols = cols | ((rows.half().float().abs() >= threshold).any(dim=0) if threshold > 0 else cols)
    if bool(cols.any()):
        subB = (CB.float() * SCB[:, None] / 127.0).half().float()  # (N, K) dequantised weights in fp16
        add = (subA @ subB.t()).half()  # only the outlier columns are non-zero in subA
        out16 = (out16.float() + add.float()).half()  # every row of the call
    return out16.to(x.dtype).reshape(*shape[:-1], CB.shape[0])

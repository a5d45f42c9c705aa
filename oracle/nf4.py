"""bitsandbytes 4-bit (NF4 / FP4, optional double quantisation) restated on the CPU (TEST INFRASTRUCTURE ONLY) — **parity unpinned**.

The reference reaches this arithmetic through bitsandbytes (``bnb.modules.Linear4bit`` subclassed at quantize/bnb.py:62-75,
selected by lit_gpt/utils.py:36-68 with quant_type nf4 / fp4 and compress_statistics False / True), a third-party wheel
(``bitsandbytes>=0.40.0``, requirements.txt:5, no lock file) that is neither under /root/reference nor installed here, and
the reference's tests never touch it.  This file restates the published algorithm of bitsandbytes 0.40/0.41
(functional.py quantize_4bit / dequantize_4bit / quantize_blockwise / create_dynamic_map; csrc/kernels.cu
kQuantizeBlockwise / kDequantizeBlockwise with dQuantizeNF4, dQuantizeFP4, dDequantizeFP4Tree, dQuantize<0>;
autograd/_functions.py MatMul4Bit = dequantise + F.linear), written with the kernels' own decision trees — deliberately
not the bucketize / table formulation the product's loader uses, so that the two restatements check each other.
The only independent anchor is the NF4 codebook itself, which tests re-derive from its definition (normal quantiles, QLoRA
arXiv:2305.14314 / functional.create_normal_map) with scipy.
"""
from typing import Optional

import numpy as np
import torch

NF4 = np.array([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0],
               dtype=np.float32)


def fp4_value(idx: np.ndarray) -> np.ndarray:
    """dDequantizeFP4Tree: bit 3 = sign, then a tree over bits 2, 1, 0."""
    idx = np.asarray(idx)
    sign = np.where((idx & 0b1000) != 0, -1.0, 1.0).astype(np.float32)
    b2, b1, b0 = (idx & 0b100) != 0, (idx & 0b010) != 0, (idx & 0b001) != 0
    mag = np.where(b2,
                   np.where(b1, np.where(b0, 0.25, 0.16666667), np.where(b0, 0.5, 0.33333333)),
                   np.where(b1, np.where(b0, 1.0, 0.66666667), np.where(b0, 5.208333333e-03, 0.0))).astype(np.float32)
    return sign * mag


FP4 = fp4_value(np.arange(16))


def quantize_nf4(x: np.ndarray) -> np.ndarray:
    """dQuantizeNF4: the kernel's comparison tree (x in [-1, 1])."""
    w = np.where
    return w(x > 0.03979014977812767,
             w(x > 0.3893125355243683,
               w(x > 0.6427869200706482, w(x > 0.8614784181118011, 15, 14), w(x > 0.5016634166240692, 13, 12)),
               w(x > 0.2035212516784668, w(x > 0.2920137718319893, 11, 10), w(x > 0.1202552504837513, 9, 8))),
             w(x > -0.33967943489551544,
               w(x > -0.13791173323988914, w(x > -0.045525018125772476, 7, 6), w(x > -0.23460740596055984, 5, 4)),
               w(x > -0.6106329262256622, w(x > -0.4599952697753906, 3, 2), w(x > -0.8480964004993439, 1, 0)))).astype(np.uint8)


def quantize_fp4(x: np.ndarray) -> np.ndarray:
    """dQuantizeFP4: sign bit, then the tree on |x|."""
    w = np.where
    sign = w(x < 0, 0b1000, 0)
    a = np.abs(x)
    mag = w(a > 0.29166667,
            w(a > 0.583333, w(a > 0.8333333, 0b0011, 0b0010), w(a > 0.4166667, 0b101, 0b100)),
            w(a > 0.0859375, w(a > 0.20833333, 0b0111, 0b0110), w(a > 0.00260417, 0b0001, 0b0000)))
    return (mag + sign).astype(np.uint8)


def create_dynamic_map() -> np.ndarray:
    """functional.create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8): 256 sorted values."""
    data = []
    for i in range(7):
        fraction_items = 2 ** i + 1
        boundaries = torch.linspace(0.1, 1, fraction_items)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-6 + i)) * means).tolist()
        data += (-(10 ** (-6 + i)) * means).tolist()
    data.append(0)
    data.append(1.0)
    data += [0] * (256 - len(data))
    data.sort()
    return np.asarray(data, dtype=np.float32)


def quantize_dynamic_scalar(code: np.ndarray, x: float) -> int:
    """dQuantize<0>(code, 0, x), one value (pure Python: small cases only)."""
    pivot, upper_pivot, lower_pivot = 127, 255, 0
    lower, upper = np.float32(-1.0), np.float32(1.0)
    val = code[pivot]
    i = 64
    while i > 0:
        if x > val:
            lower_pivot, lower = pivot, val
            pivot += i
        else:
            upper_pivot, upper = pivot, val
            pivot -= i
        val = code[pivot]
        i >>= 1
    if upper_pivot == 255:
        upper = code[upper_pivot]
    if lower_pivot == 0:
        lower = code[lower_pivot]
    if x > val:
        return upper_pivot if x > np.float32((upper + val) * np.float32(0.5)) else pivot
    return lower_pivot if x < np.float32((lower + val) * np.float32(0.5)) else pivot


def quantize_4bit(weight: torch.Tensor, quant_type: str, compress_statistics: bool, blocksize: int = 64):
    """-> (packed uint8 (n/2, 1), state dict).  ``weight`` in its own dtype; the kernel reads it as float."""
    a = weight.detach().float().numpy().reshape(-1, blocksize)
    absmax = np.abs(a).max(axis=1).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(absmax > 0, np.float32(1.0) / absmax, np.float32(0.0)).astype(np.float32)  # all-zero block: code of 0.0
    xn = (a * inv[:, None]).astype(np.float32)
    q = (quantize_nf4 if quant_type == "nf4" else quantize_fp4)(xn).reshape(-1)
    packed = ((q[0::2] << 4) | q[1::2]).astype(np.uint8).reshape(-1, 1)
    state = {"shape": tuple(weight.shape), "dtype": weight.dtype, "blocksize": blocksize, "quant_type": quant_type}
    if compress_statistics:
        offset = np.float32(torch.from_numpy(absmax).mean().item())  # torch's fp32 mean, as functional.py computes it
        centred = (absmax - offset).astype(np.float32)
        code = create_dynamic_map()
        nb = -(-centred.size // 256)
        padded = np.concatenate([centred, np.zeros(nb * 256 - centred.size, np.float32)]).reshape(nb, 256)
        absmax2 = np.abs(padded).max(axis=1).astype(np.float32)
        inv2 = np.where(absmax2 > 0, np.float32(1.0) / absmax2, np.float32(0.0)).astype(np.float32)
        normed = (padded * inv2[:, None]).astype(np.float32).reshape(-1)[: centred.size]
        qabs = np.array([quantize_dynamic_scalar(code, v) for v in normed], dtype=np.uint8)
        state.update(qabsmax=qabs, offset=offset, absmax2=absmax2, code8=code)
    else:
        state.update(absmax=absmax)
    return torch.from_numpy(packed), state


def absmax_of(state) -> np.ndarray:
    if "absmax" in state:
        return state["absmax"]
    blocks = np.arange(state["qabsmax"].size) // 256
    return (state["code8"][state["qabsmax"]] * state["absmax2"][blocks] + state["offset"]).astype(np.float32)


def dequantize_4bit(packed: torch.Tensor, state) -> torch.Tensor:
    """kDequantizeBlockwise: code[q] * absmax in fp32, one rounding to the dtype the weight had."""
    b = packed.numpy().reshape(-1)
    q = np.stack([b >> 4, b & 0xF], axis=1).reshape(-1, state["blocksize"])
    code = NF4 if state["quant_type"] == "nf4" else FP4
    w = (code[q] * absmax_of(state)[:, None]).astype(np.float32)
    return torch.from_numpy(w).to(state["dtype"]).reshape(state["shape"])


def linear(x: torch.Tensor, packed: torch.Tensor, state, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """MatMul4Bit.forward: F.linear(x, dequantize_4bit(W).to(x.dtype), bias); evaluated in float64 from the rounded weights and
    rounded once to x.dtype (the bf16 matmul accumulates in fp32 in an implementation-defined order)."""
    w = dequantize_4bit(packed, state).to(x.dtype).double()
    y = x.double() @ w.t()
    if bias is not None:
        y = y + bias.double()
    return y.to(x.dtype)

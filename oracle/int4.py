"""GPTQ int4 format restated on the CPU (TEST INFRASTRUCTURE ONLY).

Follows ``ColBlockQuantizedLinear`` (reference quantize/gptq.py:205-264) and the grid search of
``GPTQQuantizer.find_params_weight`` / ``quantize_weight`` (:312-347).  ``oracle/w4_dequant.c`` restates the
byte-level unpack in plain C; tests check the two against each other and against tests/golden/gptq_*.npz.
"""
from typing import Tuple

import torch


def new_quant_weight(out_features: int, in_features: int) -> torch.Tensor:
    """uint8 (out, in/2) stored column-major, i.e. memory [in/2][out] (gptq.py:216-222)."""
    return torch.empty((out_features, in_features // 2), dtype=torch.uint8).t().contiguous().t()


def pack_weight(weight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, tile_cols: int) -> torch.Tensor:
    """gptq.py:233-241: w/scale + zero, clamp, TRUNCATING uint8 cast, two nibbles per byte (even column low)."""
    weight = weight.clone()
    for j in range(scales.size(1)):
        weight[:, j * tile_cols:(j + 1) * tile_cols] /= scales[:, j:j + 1]
        weight[:, j * tile_cols:(j + 1) * tile_cols] += zeros[:, j:j + 1]
    q = weight.clamp_(min=0, max=15).to(dtype=torch.uint8)
    qw = new_quant_weight(weight.shape[0], weight.shape[1])
    qw.zero_()
    for nr in range(2):
        qw += q[:, nr::2] << (nr * 4)
    return qw


def get_weight(quant_weight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, tile_cols: int,
               dtype: torch.dtype = torch.float) -> torch.Tensor:
    """gptq.py:243-252: nibbles -> dtype, minus zero, times scale, all IN ``dtype`` (bf16 rounds the product)."""
    out_features, half = quant_weight.shape
    weight = torch.empty((out_features, half * 2), dtype=dtype)
    for nr in range(2):
        weight[:, nr::2] = ((quant_weight >> (nr * 4)) & 15).float()
    for j in range(scales.size(1)):
        weight[:, j * tile_cols:(j + 1) * tile_cols] -= zeros[:, j:j + 1]
        weight[:, j * tile_cols:(j + 1) * tile_cols] *= scales[:, j:j + 1]
    return weight


def find_params(x: torch.Tensor, maxq: int = 15) -> Tuple[torch.Tensor, torch.Tensor]:
    """gptq.py:317-347 with perchannel=True, sym=False: per-row min/max including 0 -> scale, integer zero."""
    tmp = torch.zeros(x.shape[0])
    xmin = torch.minimum(x.min(1)[0], tmp)
    xmax = torch.maximum(x.max(1)[0], tmp)
    flat = (xmin == 0) & (xmax == 0)
    xmin[flat] = -1
    xmax[flat] = +1
    scale = (xmax - xmin) / maxq
    zero = torch.round(-xmin / scale)
    return scale.reshape(-1, 1), zero.reshape(-1, 1)


def rtn_quantize(weight: torch.Tensor, tile_cols: int, store_dtype: torch.dtype):
    """Round-to-nearest onto the find_params grid, per (row, group of tile_cols columns).

    Returns (quant_weight in the reference layout, scales, zeros) with scales/zeros in ``store_dtype`` — the grid is
    built from the stored (rounded) parameters so that what is packed is exactly on the grid that get_weight uses.
    """
    out_f, in_f = weight.shape
    if tile_cols == -1:
        tile_cols = in_f
    n_groups = -(-in_f // tile_cols)
    w = weight.float()
    scales = torch.empty((out_f, n_groups))
    zeros = torch.empty((out_f, n_groups))
    q = torch.empty((out_f, in_f), dtype=torch.uint8)
    for j in range(n_groups):
        blk = w[:, j * tile_cols:(j + 1) * tile_cols]
        s, z = find_params(blk)
        s = s.to(store_dtype).float()
        z = z.to(store_dtype).float()
        scales[:, j:j + 1], zeros[:, j:j + 1] = s, z
        q[:, j * tile_cols:(j + 1) * tile_cols] = torch.clamp(torch.round(blk / s) + z, 0, 15).to(torch.uint8)  # gptq.py:313-315
    qw = new_quant_weight(out_f, in_f)
    qw.copy_(q[:, 0::2] | (q[:, 1::2] << 4))
    return qw, scales.to(store_dtype), zeros.to(store_dtype)


def quantize_state_dict(sd, tile_cols: int, is_linear_key):
    """Replace every Linear ``<name>.weight`` by ``<name>.quant_weight/scales/zeros`` (what quantize/gptq.py::main saves)."""
    out = {}
    for k, v in sd.items():
        if is_linear_key(k):
            stem = k[: -len(".weight")]
            qw, s, z = rtn_quantize(v, tile_cols, v.dtype)
            out[stem + ".quant_weight"], out[stem + ".scales"], out[stem + ".zeros"] = qw, s, z
        else:
            out[k] = v
    return out

"""GPTQ int4 Linear replacement with the reference's state-dict format, computed by the W4 HIP kernels.

Interface mirrors ``ColBlockQuantizedLinear`` (reference quantize/gptq.py:205-264): same constructor, same buffers
(``quant_weight`` uint8 logical (out, in/2) stored column-major, ``scales``/``zeros`` (out, ceil(in/tile_cols)),
optional ``bias``), ``pack_weight`` / ``get_weight`` / ``forward``.  Differences, on purpose:
  * ``forward`` runs grouped (tile_cols != -1) and biased layers on the fused kernel too (the reference's Triton path
    is per-channel only and drops the bias, gptq.py:255-264);
  * ``pack_weight`` rounds to nearest before the uint8 cast (the reference truncates, gptq.py:239, which is only
    right for inputs already on the grid);
  * the kernel reads a repacked copy of the weights ("W4K") built once after loading.
``rtn_quantize`` is the round-to-nearest quantiser used for synthetic weights (find_params_weight semantics,
gptq.py:317-347).
"""
from typing import Optional, Tuple

import torch

from .. import ops
from .._hip import EPI_NONE, ParrotHipError


class ColBlockQuantizedLinear(torch.nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool, *, bits: int = 4, tile_cols: int = -1) -> None:
        super().__init__()
        if bits != 4:
            raise NotImplementedError("only 4-bit weights are built for the HIP path")
        if in_features % 2:
            raise ValueError("in_features must be even")
        assert isinstance(bias, bool)
        self.in_features, self.out_features, self.bits = in_features, out_features, bits
        self.tile_cols = in_features if tile_cols == -1 else tile_cols
        self.entries_per_byte = 2
        n_groups = -(-in_features // self.tile_cols)
        # (out, in/2) view of an (in/2, out) row-major buffer: the reference's on-disk layout
        self.register_buffer("quant_weight", torch.empty((in_features // 2, out_features), dtype=torch.uint8).t())
        self.register_buffer("scales", torch.empty((out_features, n_groups)))
        self.register_buffer("zeros", torch.empty((out_features, n_groups)))
        self.register_buffer("bias", torch.empty((out_features,)) if bias else None)
        self._packed: Optional[torch.Tensor] = None  # W4K copy (device), rebuilt when the buffers change

    # -------------------------------------------------------------------------------------- format (torch ops)
    def pack_weight(self, weight: torch.Tensor) -> None:
        """Quantise ``weight`` (out, in) onto the grid given by ``scales``/``zeros`` and store the nibbles."""
        w = weight.to(device=self.quant_weight.device, dtype=torch.float32)
        g = torch.arange(self.in_features, device=w.device) // self.tile_cols
        q = torch.round(w / self.scales.float()[:, g] + self.zeros.float()[:, g]).clamp_(0, 15).to(torch.uint8)
        self.quant_weight.copy_(q[:, 0::2] | (q[:, 1::2] << 4))
        self._packed = None

    def get_weight(self, dtype: torch.dtype = torch.float) -> torch.Tensor:
        """Dequantised (out, in) weight: (q - zero) * scale computed in ``dtype`` like the reference (:243-252)."""
        qw = self.quant_weight
        q = torch.empty((self.out_features, self.in_features), dtype=dtype, device=qw.device)
        q[:, 0::2] = (qw & 0xF).to(dtype)
        q[:, 1::2] = (qw >> 4).to(dtype)
        g = torch.arange(self.in_features, device=qw.device) // self.tile_cols
        q -= self.zeros.to(dtype)[:, g]
        q *= self.scales.to(dtype)[:, g]
        return q

    def _load_from_state_dict(self, *args, **kwargs) -> None:
        super()._load_from_state_dict(*args, **kwargs)
        self._packed = None

    def _apply(self, fn, *args, **kwargs):
        # .to(device) / .to(dtype): keep quant_weight's column-major layout and drop the derived copy
        self._packed = None
        qw = self.quant_weight
        out = super()._apply(fn, *args, **kwargs)
        if self.quant_weight.stride() != (1, self.out_features):
            self.quant_weight = self.quant_weight.t().contiguous().t()
        del qw
        return out

    # -------------------------------------------------------------------------------------- HIP path
    def packed(self) -> torch.Tensor:
        """The kernel-native W4K buffer, built on first use with the repack kernel."""
        if self._packed is None:
            if not self.quant_weight.is_cuda:
                raise ParrotHipError("ColBlockQuantizedLinear: move the module to the GPU before running it")
            nbytes = ops.w4_packed_bytes(self.out_features, self.in_features, self.tile_cols)
            buf = torch.empty((nbytes,), dtype=torch.uint8, device=self.quant_weight.device)
            ops.w4_repack(self.quant_weight, self.scales.to(torch.bfloat16).contiguous(),
                          self.zeros.to(torch.bfloat16).contiguous(), self.out_features, self.in_features,
                          self.tile_cols, buf, 0)
            self._packed = buf
        return self._packed

    def hip_linear(self, x: torch.Tensor, out: torch.Tensor, *, epilogue: int = EPI_NONE, residual=None,
                   partner: Optional["ColBlockQuantizedLinear"] = None, norm=None) -> torch.Tensor:
        """rows (M, in) -> (M, out) with a fused epilogue; ``partner`` is fc_2 for the SwiGLU epilogue."""
        if partner is not None and (partner.tile_cols != self.tile_cols or partner.in_features != self.in_features
                                    or partner.out_features != self.out_features):
            raise ParrotHipError("SwiGLU partner must have the same shape and group size")
        return ops.w4_linear(self.packed(), self.out_features, self.in_features, self.tile_cols, x, out, bias=self.bias,
                             epilogue=epilogue, residual=residual,
                             packed2=partner.packed() if partner is not None else None, norm=norm)

    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        x = inp.reshape(-1, self.in_features)
        out = torch.empty((x.shape[0], self.out_features), dtype=inp.dtype, device=inp.device)
        self.hip_linear(x.contiguous(), out)
        return out.view(*inp.shape[:-1], self.out_features)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bits=4, tile_cols={self.tile_cols}"


def rtn_quantize(weight: torch.Tensor, tile_cols: int = 128, bits: int = 4) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Round-to-nearest asymmetric quantisation per (row, group of ``tile_cols`` columns).

    Grid parameters follow ``GPTQQuantizer.find_params_weight`` (quantize/gptq.py:317-347): the range always
    contains 0, scale = (max - min) / 15, zero = round(-min / scale).  Returns (q uint8 (out, in), scales, zeros)
    with scales/zeros (out, groups) in ``weight.dtype`` — the values the reference would store.
    """
    out_f, in_f = weight.shape
    if tile_cols == -1:
        tile_cols = in_f
    maxq = 2 ** bits - 1
    n_groups = -(-in_f // tile_cols)
    pad = n_groups * tile_cols - in_f
    w = weight.float()
    if pad:  # ragged last group: pad with a value inside the group's range so that min/max are unchanged
        w = torch.cat([w, w[:, -1:].expand(out_f, pad)], dim=1)
    blk = w.view(out_f, n_groups, tile_cols)
    lo = blk.amin(dim=2).clamp(max=0)
    hi = blk.amax(dim=2).clamp(min=0)
    flat = (lo == 0) & (hi == 0)
    lo = torch.where(flat, torch.full_like(lo, -1), lo)
    hi = torch.where(flat, torch.full_like(hi, 1), hi)
    s = (hi - lo) / maxq
    z = torch.round(-lo / s)
    # the stored parameters are rounded to the model dtype first, then used for the grid (as on load)
    scales, zeros = s.to(weight.dtype), z.to(weight.dtype)
    # q = clamp(round(x / scale) + zero, 0, maxq)  (GPTQQuantizer.quantize_weight, gptq.py:313-315)
    q = (torch.round(blk / scales.float()[:, :, None]) + zeros.float()[:, :, None]).clamp_(0, maxq).to(torch.uint8)
    return q.view(out_f, n_groups * tile_cols)[:, :in_f].contiguous(), scales, zeros


def pack_nibbles(q: torch.Tensor) -> torch.Tensor:
    """(out, in) uint8 values 0..15 -> the reference's quant_weight tensor: (out, in/2) with strides (1, out)."""
    return (q[:, 0::2] | (q[:, 1::2] << 4)).t().contiguous().t()

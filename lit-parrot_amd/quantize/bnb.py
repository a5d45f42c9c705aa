"""LLM.int8 Linear replacement (reference quantize/bnb.py:18-60) without bitsandbytes.

``InferenceLinear8bitLt`` keeps the reference class's surface: an ``nn.Linear`` subclass constructed with
``has_fp16_weights=False, threshold=6.0`` (bnb.py:26-33) that quantises its weight when a float ``weight`` arrives
through ``load_state_dict`` (bnb.py:38-50) and afterwards holds ``weight`` = int8 ``CB`` with attributes ``CB`` and
``SCB`` (bnb.py:52-60).  The arithmetic — bitsandbytes' ``double_quant`` / ``MatMul8bitLt`` in the reference — is
done by the W8 HIP kernels (csrc/w8.hip), which restate the published LLM.int8 algorithm.
The 4-bit bitsandbytes variants (``Linear4bit``, nf4/fp4) are not built.
"""
from typing import Optional

import torch

from .. import ops
from .._hip import EPI_NONE, ParrotHipError


FUSED_DECODE = True  # single rows take parrot_w8_gemv_fused (tests switch it off to compare with the two-launch path)


class InferenceLinear8bitLt(torch.nn.Linear):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, *, has_fp16_weights: bool = False,
                 threshold: float = 6.0, device=None, dtype=None) -> None:
        if has_fp16_weights:
            raise NotImplementedError("inference only: has_fp16_weights=False (quantize/bnb.py:30)")
        super().__init__(in_features, out_features, bias, device=device, dtype=dtype)
        self.threshold = float(threshold)
        self.weight.requires_grad_(False)
        if self.bias is not None:
            self.bias.requires_grad_(False)
        self._act: Optional[ops.W8Act] = None
        self._pair = None  # (CB, SCB) of [self; partner] concatenated for the SwiGLU epilogue
        # the reference quantises the freshly initialised weight right away (bnb.py:34-36); that needs the GPU, so
        # here it happens when the module reaches the device (see _apply) or a checkpoint is loaded

    @property
    def is_quantized(self) -> bool:
        return self.weight.dtype == torch.int8

    def _quantize_weight(self, weight: torch.Tensor) -> None:
        """Row-wise absmax int8 of ``weight`` on the current GPU (double_quant semantics, bnb.py:52-60)."""
        if not torch.cuda.is_available():
            raise ParrotHipError("InferenceLinear8bitLt quantises on the GPU: no HIP device is visible")
        dev = weight.device if weight.is_cuda else torch.device("cuda", torch.cuda.current_device())
        w = weight.detach().to(device=dev, dtype=torch.bfloat16).contiguous()
        CB = torch.empty(w.shape, dtype=torch.int8, device=dev)
        SCB = torch.empty((w.shape[0],), dtype=torch.float32, device=dev)
        ops.w8_quantize_rows(w, CB, SCB)
        self.weight = torch.nn.Parameter(CB, requires_grad=False)
        setattr(self.weight, "CB", CB)
        setattr(self.weight, "SCB", SCB)
        self._pair = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs) -> None:
        key = prefix + "weight"
        if key in state_dict:
            self._quantize_weight(state_dict.pop(key))
            # nn.Module would now report the popped key as missing; the reference loads with strict=False
            # (generate/base.py:222).  Put a placeholder so that strict loading also works.
            state_dict[key] = self.weight.data
            super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
            state_dict.pop(key)
            return
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        scb = getattr(self.weight, "SCB", None)
        out = super()._apply(fn, *args, **kwargs)
        if self.is_quantized and scb is not None:
            # nn.Module._apply may have rebuilt the Parameter: carry CB / SCB over, on the same device as CB
            setattr(self.weight, "CB", self.weight.data)
            setattr(self.weight, "SCB", scb.to(self.weight.device))
            self._pair = None
        elif not self.is_quantized and self.weight.is_cuda:
            self._quantize_weight(self.weight.data)
        return out

    # -------------------------------------------------------------------------------------- HIP path
    def prep(self, x: torch.Tensor, norm=None) -> ops.W8Act:
        M, K = x.shape
        if self._act is None or (self._act.M, self._act.K) != (M, K) or self._act.xq.device != x.device:
            self._act = ops.W8Act(M, K, x.device)
        return ops.w8_prep_act(x, self.threshold, self._act, norm)

    def hip_linear(self, x: torch.Tensor, out: torch.Tensor, *, epilogue: int = EPI_NONE, residual=None,
                   partner: Optional["InferenceLinear8bitLt"] = None, act: Optional[ops.W8Act] = None,
                   norm=None) -> torch.Tensor:
        if not self.is_quantized:
            raise ParrotHipError("InferenceLinear8bitLt: weight not quantised yet (move the module to the GPU)")
        CB, SCB = self.weight.data, self.weight.SCB
        if partner is not None:
            if self._pair is None:
                self._pair = (torch.cat([CB, partner.weight.data]).contiguous(), torch.cat([SCB, partner.weight.SCB]).contiguous())
            CB, SCB = self._pair
        if act is None and x.shape[0] == 1 and self.in_features <= ops.W8_FUSED_MAX_K and FUSED_DECODE:
            # decode: activation quantiser (with the fused norm) + GEMV in one launch
            return ops.w8_linear_fused(CB, SCB, self.out_features, self.in_features, x, self.threshold, out, bias=self.bias,
                                       epilogue=epilogue, residual=residual, norm=norm)
        act = act if act is not None else self.prep(x, norm)  # the norm is fused into the activation quantiser
        return ops.w8_linear(CB, SCB, self.out_features, self.in_features, act, out, bias=self.bias, epilogue=epilogue,
                             residual=residual)

    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        x = inp.reshape(-1, self.in_features).contiguous()
        out = torch.empty((x.shape[0], self.out_features), dtype=inp.dtype, device=inp.device)
        self.hip_linear(x, out)
        return out.view(*inp.shape[:-1], self.out_features)

"""``quantization(mode)``: the drop-in boundary of the quantized path (reference lit_gpt/utils.py:26-83).

A context manager that replaces ``torch.nn.Linear`` while a model is being constructed, so that every Linear of
``GPT(config)`` (including ``lm_head``) is built as the quantized class.  Modes:
  * ``"bnb.int8"``      -> ``quantize.bnb.InferenceLinear8bitLt``                      (utils.py:32-35)
  * ``"gptq.int4"``     -> ``quantize.gptq.ColBlockQuantizedLinear(bits=4, tile_cols=-1)`` (utils.py:69-76)
  * ``"gptq.int4-g<N>"``-> the same with ``tile_cols=N`` (e.g. ``gptq.int4-g128``): grouped scales, which the
    reference's class supports (gptq.py:206-226) but its context manager cannot select.
  * ``"bnb.nf4"``, ``"bnb.nf4-dq"``, ``"bnb.fp4"``, ``"bnb.fp4-dq"`` -> ``quantize.bnb.Linear4bit`` with the matching
    ``quant_type`` / ``compress_statistics`` (utils.py:36-68).
Unlike the reference (utils.py:80-83) ``torch.nn.Linear`` is restored even when the body raises.
"""
import re
from contextlib import contextmanager
from typing import Optional

import torch

from .config import find_multiple  # noqa: F401  (re-exported like the reference's utils)

_BNB4 = ("bnb.nf4", "bnb.nf4-dq", "bnb.fp4", "bnb.fp4-dq")


def quantized_linear_class(mode: str):
    if mode == "bnb.int8":
        from .quantize.bnb import InferenceLinear8bitLt

        return InferenceLinear8bitLt
    m = re.fullmatch(r"gptq\.int4(?:-g(\d+))?", mode)
    if m:
        from .quantize.gptq import ColBlockQuantizedLinear

        tile_cols = int(m.group(1)) if m.group(1) else -1

        class QuantizedLinear(ColBlockQuantizedLinear):
            def __init__(self, *args, **kwargs):
                super().__init__(*args, bits=4, tile_cols=tile_cols, **kwargs)

        return QuantizedLinear
    if mode in _BNB4:
        from .quantize.bnb import Linear4bit

        quant_type, compress = mode[4:7], mode.endswith("-dq")

        class QuantizedLinear(Linear4bit):
            def __init__(self, *args, **kwargs):
                super().__init__(*args, quant_type=quant_type, compress_statistics=compress, **kwargs)

        return QuantizedLinear
    raise ValueError(f"Unknown quantization mode: {mode}")


@contextmanager
def quantization(mode: Optional[str] = None):
    if mode is None:
        yield
        return
    quantized_linear_cls = quantized_linear_class(mode)
    torch_linear_cls = torch.nn.Linear
    torch.nn.Linear = quantized_linear_cls
    try:
        yield
    finally:
        torch.nn.Linear = torch_linear_cls
